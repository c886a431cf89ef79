"""Synthetic solver domains (numpy only): the Houdini-free re-expression of the reference's test
domain builders, used by tests/, bench.py and __graft_entry__.smoke().

* build_simple_domain   -- Test.cpp:466-625  (Dirichlet-band box, unit weights)
* build_complex_domain  -- Test.cpp:207-464  (sine free surface, ghost-fluid weights, optional solid)
* expand_domain         -- Test.cpp:170-204 / Ops.h:1328-1644 for a *chosen* level count
                           ("solver grid = N^3, L chosen by the config")
* interior_cube / free_surface_pool -- the BASELINE.json configurations as solver grids.

Grids are indexed [k, j, i] (x fastest in memory).  Labels are uint8 with the reference's values
(Ops.h:11).  Face-weight grid of axis a has one extra entry along a (MG.cpp:167-177).

The solid obstacle is an axis-aligned box with analytic cut-cell area fractions instead of the
reference's SDF sphere (Test.cpp:266-343 uses HDK computeSDFWeightsFace, which is not available).
"""
import numpy as np

INTERIOR, EXTERIOR, DIRICHLET, BOUNDARY = 0, 1, 2, 3
RHS_SEED = 20240501


def face_shape(nz, ny, nx, axis):
    s = [nz, ny, nx]
    s[2 - axis] += 1
    return tuple(s)


def _shift_pair(lab, axis):
    """labels of the (backward, forward) cells of every *interior* face of `axis`"""
    ax = 2 - axis
    n = lab.shape[ax]
    back = np.take(lab, np.arange(0, n - 1), axis=ax)
    fwd = np.take(lab, np.arange(1, n), axis=ax)
    return back, fwd


def _inner_faces(w, axis):
    ax = 2 - axis
    sl = [slice(None)] * 3
    sl[ax] = slice(1, -1)
    return tuple(sl)


def build_simple_domain(grid_size, dirichlet_band=1, dtype=np.float64):
    """Test.cpp:466-625: DIRICHLET band on all six sides, INTERIOR inside, weight 1 on faces that
    touch an INTERIOR cell and no EXTERIOR / out-of-range cell."""
    g, d = grid_size, dirichlet_band
    lab = np.full((g, g, g), EXTERIOR, dtype=np.uint8)
    if d > 0:
        lab[:] = DIRICHLET
    lab[d : g - d, d : g - d, d : g - d] = INTERIOR
    weights = []
    for axis in range(3):
        w = np.zeros(face_shape(g, g, g, axis), dtype=dtype)
        back, fwd = _shift_pair(lab, axis)
        interior = (back == INTERIOR) | (fwd == INTERIOR)
        exterior = (back == EXTERIOR) | (fwd == EXTERIOR)
        w[_inner_faces(w, axis)] = np.where(interior & ~exterior, 1.0, 0.0)
        weights.append(w)
    return lab, weights, 1.0 / g


def ghost_fluid_theta(phi0, phi1):
    """Util.h:25-42, vectorised."""
    theta = np.zeros_like(phi0)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = phi0 / (phi0 - phi1)
        b = phi1 / (phi1 - phi0)
    theta = np.where((phi0 < 0) & (phi1 < 0), 1.0, theta)
    theta = np.where((phi0 < 0) & (phi1 >= 0), a, theta)
    theta = np.where((phi0 >= 0) & (phi1 < 0), b, theta)
    return theta


def _box_face_weights(g, axis, lo, hi, dtype):
    """Open-area fraction of every face of `axis` for a solid box [lo,hi]^3 in the unit cube;
    fractions below 0.01 clamp to 0 (Test.cpp:320)."""
    gz, gy, gx = g if isinstance(g, tuple) else (g, g, g)  # a non-cubic box: coordinates normalised per axis

    def overlap(n):  # covered length fraction of cell intervals [q dx, (q+1) dx]
        dx = 1.0 / n
        q = np.arange(n)
        a = np.maximum(q * dx, lo)
        b = np.minimum((q + 1) * dx, hi)
        return np.clip(b - a, 0.0, None) / dx

    def inside(n):  # face planes q*dx inside the box
        q = np.arange(n + 1) / n
        return ((q >= lo) & (q <= hi)).astype(np.float64)

    per_axis = [overlap(gx), overlap(gy), overlap(gz)]
    per_axis[axis] = inside((gx, gy, gz)[axis])
    covered = per_axis[2][:, None, None] * per_axis[1][None, :, None] * per_axis[0][None, None, :]
    w = 1.0 - covered
    w[w < 0.01] = 0.0
    return w.astype(dtype)


def build_complex_domain(grid_size, use_solid=False, dtype=np.float64, solid_box=(0.4, 0.6)):
    """Test.cpp:207-464: liquid where phi <= 0 with phi = x - .5 + .25 sin(2 pi y + 4 pi z) sampled
    at cell index * dx; wall faces closed; cells without an open face EXTERIOR; liquid/air faces
    divided by clamp(theta, .01, 1); air/air faces 0.  grid_size may be a (gz, gy, gx) tuple for a
    non-cubic box (coordinates are then normalised per axis, the solid box included)."""
    if isinstance(grid_size, tuple):
        return _build_complex_box(grid_size, dtype, use_solid, solid_box)
    g = grid_size
    dx = 1.0 / g
    q = np.arange(g) * dx
    z, y, x = np.meshgrid(q, q, q, indexing="ij")
    phi = x - 0.5 + 0.25 * np.sin(2.0 * np.pi * y + 4.0 * np.pi * z)  # Test.cpp:233-236
    return _complex_from_phi(phi, (g, g, g), use_solid, solid_box, dtype, dx)


def _build_complex_box(shape, dtype, use_solid=False, solid_box=(0.4, 0.6)):
    gz, gy, gx = shape
    z, y, x = np.meshgrid(np.arange(gz) / gz, np.arange(gy) / gy, np.arange(gx) / gx, indexing="ij")
    phi = x - 0.5 + 0.25 * np.sin(2.0 * np.pi * y + 4.0 * np.pi * z)
    return _complex_from_phi(phi, shape, use_solid, solid_box, dtype, 1.0 / max(shape))


def _complex_from_phi(phi, shape, use_solid, solid_box, dtype, dx):
    gz, gy, gx = shape
    g = gx if gz == gy == gx else tuple(shape)
    weights = []
    for axis in range(3):
        if use_solid:
            w = _box_face_weights(g, axis, solid_box[0], solid_box[1], np.float64)
        else:
            w = np.ones(face_shape(gz, gy, gx, axis), dtype=np.float64)
        sl = [slice(None)] * 3
        sl[2 - axis] = 0
        w[tuple(sl)] = 0.0  # Test.cpp:345-360
        sl[2 - axis] = -1
        w[tuple(sl)] = 0.0
        weights.append(w)

    def lo(w, axis):
        return np.take(w, np.arange(0, shape[2 - axis]), axis=2 - axis)

    def hi(w, axis):
        return np.take(w, np.arange(1, shape[2 - axis] + 1), axis=2 - axis)

    open_face = np.zeros(shape, dtype=bool)
    for axis in range(3):
        open_face |= (lo(weights[axis], axis) > 0) | (hi(weights[axis], axis) > 0)
    lab = np.full(shape, EXTERIOR, dtype=np.uint8)  # Test.cpp:382-401
    lab[open_face & (phi > 0)] = DIRICHLET
    lab[open_face & (phi <= 0)] = INTERIOR

    for axis in range(3):  # Test.cpp:406-461
        w = weights[axis]
        inner = _inner_faces(w, axis)
        back, fwd = _shift_pair(lab, axis)
        pb, pf = _shift_pair(phi, axis)
        wi = w[inner]
        both_air = (back == DIRICHLET) & (fwd == DIRICHLET)
        one_air = (back == DIRICHLET) ^ (fwd == DIRICHLET)
        theta = np.clip(ghost_fluid_theta(pb, pf), 0.01, 1.0)
        wi = np.where((wi > 0) & both_air, 0.0, wi)
        wi = np.where((wi > 0) & one_air, wi / theta, wi)
        w[inner] = wi
        weights[axis] = w.astype(dtype)
    return lab, weights, dx


def set_boundary_labels(lab, weights):
    """Ops.h:1574-1644, vectorised: INTERIOR -> BOUNDARY if a face neighbour is DIRICHLET/EXTERIOR
    or one of the six face weights differs from 1.  In place."""
    nz, ny, nx = lab.shape
    interior = lab == INTERIOR
    bad = np.zeros(lab.shape, dtype=bool)
    inactive = (lab == DIRICHLET) | (lab == EXTERIOR)
    for axis in range(3):
        ax = 2 - axis
        n = lab.shape[ax]
        w = weights[axis]
        bad |= np.take(w, np.arange(0, n), axis=ax) != 1
        bad |= np.take(w, np.arange(1, n + 1), axis=ax) != 1
        nb = np.zeros(lab.shape, dtype=bool)
        sl_dst = [slice(None)] * 3
        sl_src = [slice(None)] * 3
        sl_dst[ax], sl_src[ax] = slice(1, None), slice(0, -1)
        nb[tuple(sl_dst)] |= inactive[tuple(sl_src)]
        sl_dst[ax], sl_src[ax] = slice(0, -1), slice(1, None)
        nb[tuple(sl_dst)] |= inactive[tuple(sl_src)]
        bad |= nb
    lab[interior & bad] = BOUNDARY
    return lab


def reference_level_count(base_shape):
    """Ops.h:1340-1345."""
    return int(np.ceil(np.log2(min(base_shape))) - 1)


def expand_domain(base_lab, base_weights, levels=None, solver_shape=None):
    """Pad a base domain with 2^(levels-1) EXTERIOR cells per side (Ops.h:1347-1351) and embed it in
    `solver_shape` (default: every extent rounded up to a power of two, Ops.h:1353-1360); copy
    labels / positive weights at +offset (Ops.h:1408-1453, 1524-1571); label BOUNDARY cells
    (Ops.h:1574).  Returns (labels uint8, weights[3], offset, levels)."""
    bz, by, bx = base_lab.shape
    if levels is None:
        levels = reference_level_count(base_lab.shape)
    pad = 2 ** (levels - 1)
    if solver_shape is None:
        solver_shape = tuple(int(2 ** np.ceil(np.log2(n + 2 * pad))) for n in (bz, by, bx))
    ez, ey, ex = solver_shape
    assert ez >= bz + 2 * pad and ey >= by + 2 * pad and ex >= bx + 2 * pad
    lab = np.full(solver_shape, EXTERIOR, dtype=np.uint8)
    lab[pad : pad + bz, pad : pad + by, pad : pad + bx] = np.where(
        base_lab == EXTERIOR, EXTERIOR, np.where(base_lab == INTERIOR, INTERIOR, DIRICHLET)
    )
    weights = []
    for axis in range(3):
        bw = base_weights[axis]
        w = np.zeros(face_shape(ez, ey, ex, axis), dtype=bw.dtype)
        fz, fy, fx = bw.shape
        w[pad : pad + fz, pad : pad + fy, pad : pad + fx] = np.where(bw > 0, bw, 0)
        weights.append(w)
    set_boundary_labels(lab, weights)
    return lab, weights, pad, levels


def interior_cube_slab(n, levels, z0, z1, dtype=np.float32):
    """The interior-liquid cube of BASELINE configs 1, 2, 4 written down directly (no N^3 float
    temporaries, so it scales to 1024^3): labels of the WHOLE N^3 grid plus the face weights of the
    planes [z0, z1) only.  Identical to buildSimpleDomain(g = N - 2p, band 1) + expansion +
    setBoundaryCellLabels: p = 2^(L-1) EXTERIOR cells per side, a one-cell DIRICHLET shell, the
    outermost liquid layer BOUNDARY (it touches the shell), weight 1 on every face that touches a
    liquid cell (Test.cpp:597-618)."""
    pad = 2 ** (levels - 1)
    assert n - 2 * pad >= 4, "grid too small for this many levels"
    lo, hi = pad + 1, n - pad - 1  # liquid cells: [lo, hi) on every axis
    lab = np.full((n, n, n), EXTERIOR, dtype=np.uint8)
    lab[pad : n - pad, pad : n - pad, pad : n - pad] = DIRICHLET
    lab[lo:hi, lo:hi, lo:hi] = BOUNDARY
    lab[lo + 1 : hi - 1, lo + 1 : hi - 1, lo + 1 : hi - 1] = INTERIOR
    nzl = z1 - z0
    k0, k1 = max(lo, z0) - z0, min(hi, z1) - z0  # liquid planes of the slab, local indices
    weights = []
    for axis in range(3):
        shape = [nzl, n, n]
        shape[2 - axis] += 1
        w = np.zeros(shape, dtype=dtype)
        if axis == 0:
            if k1 > k0:
                w[k0:k1, lo:hi, lo : hi + 1] = 1
        elif axis == 1:
            if k1 > k0:
                w[k0:k1, lo : hi + 1, lo:hi] = 1
        else:  # z faces: global face planes [lo, hi] intersected with the slab's [z0, z1]
            f0, f1 = max(lo, z0) - z0, min(hi + 1, z1 + 1) - z0
            if f1 > f0:
                w[f0:f1, lo:hi, lo:hi] = 1
        weights.append(w)
    return lab, weights, 1.0 / n


def interior_cube(n, levels, dtype=np.float32):
    """BASELINE configs 1, 2, 4: N^3 solver grid, 2^(L-1) EXTERIOR cells per side, a one-cell
    DIRICHLET shell, liquid inside (buildSimpleDomain(g = N - 2p, band 1) + expansion)."""
    return interior_cube_slab(n, levels, 0, n, dtype=dtype)


def free_surface_pool(n, levels, use_solid=True, dtype=np.float32):
    """BASELINE configs 3, 5: N^3 solver grid around buildComplexDomain(g = N - 2p) with the solid
    box; ghost-fluid weights reach 1/0.01."""
    pad = 2 ** (levels - 1)
    g = n - 2 * pad
    base_lab, base_w, _ = build_complex_domain(g, use_solid=use_solid, dtype=dtype)
    lab, w, off, _ = expand_domain(base_lab, base_w, levels=levels, solver_shape=(n, n, n))
    return lab, w, 1.0 / n


def active_mask(lab):
    return (lab == INTERIOR) | (lab == BOUNDARY)


def random_rhs(lab, h, seed=RHS_SEED, dtype=np.float32, z0=0, z1=None):
    """U(0,1) * h^2 on active cells, 0 elsewhere (Test.cpp:1180-1194 with fixed PCG64 seeds).  One
    generator per global z-plane (seed + k), so a Z-slab [z0, z1) of the field can be produced
    without generating the rest; returns only those planes."""
    nz = lab.shape[0]
    z1 = nz if z1 is None else z1
    out = np.zeros((z1 - z0,) + lab.shape[1:], dtype=dtype)
    for k in range(z0, z1):
        act = active_mask(lab[k])
        if act.any():
            plane = np.random.Generator(np.random.PCG64(seed + k)).random(lab.shape[1:]) * (h * h)
            out[k - z0] = np.where(act, plane, 0.0)
    return out


def delta_rhs(lab, grid_size, offset, h, amplitude=1000.0, dtype=np.float32):
    """3x3x3 block of `amplitude` at 10 % of the base grid, scaled by h^2 on active cells
    (Test.cpp:727-742, 793-794)."""
    b = np.zeros(lab.shape, dtype=np.float64)
    p = int(0.1 * grid_size) + offset
    b[p - 1 : p + 2, p - 1 : p + 2, p - 1 : p + 2] = amplitude
    b[active_mask(lab)] *= h * h
    b[~active_mask(lab)] = 0.0  # keeps the zero-outside-active invariant (Ops.h:821-823)
    return b.astype(dtype)


def sine_initial_guess(lab, h, dtype=np.float32):
    """Test.cpp:1918-1920 (the second mode repeats y, as in the reference)."""
    nz, ny, nx = lab.shape
    z, y, x = np.meshgrid(np.arange(nz) * h, np.arange(ny) * h, np.arange(nx) * h, indexing="ij")
    v = np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.sin(2 * np.pi * z) + np.sin(4 * np.pi * x) * np.sin(
        4 * np.pi * y
    ) * np.sin(4 * np.pi * y)
    v[~active_mask(lab)] = 0.0
    return v.astype(dtype)


def projection_scene(shape, seed=7, with_solid_velocity=False, dtype=np.float32, randomize=False):
    """Synthetic inputs of one pressure projection on the BASE grid (Plug.cpp:113-426): a pool with a wavy free
    surface, closed domain walls and an immersed axis-aligned solid box whose faces cut cells (fractional
    cut-cell weights = open area fraction of each face, < 0.01 closed).  Returns a dict of numpy arrays:
    liquid_phi, solid_phi (cell centres; liquid where liquid_phi <= 0, inside the solid where solid_phi >= 0,
    the conventions of Util.cpp:14, 25), cut_weights[3], velocity[3] (face grids) and solid_velocity[3] or None."""
    gz, gy, gx = shape
    dx = 1.0 / max(shape)
    rng = np.random.Generator(np.random.PCG64(seed))
    zc, yc, xc = np.meshgrid((np.arange(gz) + 0.5) * dx, (np.arange(gy) + 0.5) * dx, (np.arange(gx) + 0.5) * dx, indexing="ij")
    top = gz * dx
    level, amp, ph = 0.55, 0.08, 0.0
    lo_f, hi_f = np.array([0.31, 0.27, 0.18]), np.array([0.62, 0.71, 0.44])
    if randomize:  # geometry drawn from the seed too: fill level, wave, box position and size
        level, amp, ph = rng.uniform(0.35, 0.8), rng.uniform(0.0, 0.15), rng.uniform(0, 2 * np.pi)
        lo_f = rng.uniform(0.1, 0.5, 3)
        hi_f = np.minimum(lo_f + rng.uniform(0.15, 0.4, 3), 0.93)
    liquid_phi = zc - top * (level + amp * np.sin(2 * np.pi * xc / (gx * dx) + ph) * np.cos(2 * np.pi * yc / (gy * dx)))
    lo = lo_f * np.array([gx, gy, gz]) * dx  # box corners (x, y, z), off the grid lines
    hi = hi_f * np.array([gx, gy, gz]) * dx
    inside = (xc > lo[0]) & (xc < hi[0]) & (yc > lo[1]) & (yc < hi[1]) & (zc > lo[2]) & (zc < hi[2])
    solid_phi = np.where(inside, dx, -dx)

    def overlap(a0, a1, b0, b1):  # length of [a0, a1] covered by [b0, b1]
        return np.clip(np.minimum(a1, b1) - np.maximum(a0, b0), 0.0, None)

    cut = []
    for axis in range(3):  # 0 = x faces
        fshape = face_shape(gz, gy, gx, axis)
        n = [gx, gy, gz]
        idx = np.meshgrid(*[np.arange(fshape[d]) for d in range(3)], indexing="ij")  # k, j, i
        pos = {0: idx[2], 1: idx[1], 2: idx[0]}  # integer index along x, y, z
        # in-plane extents of the face and its position along the axis
        t = [a for a in range(3) if a != axis]
        area = np.ones(fshape)
        for a in t:
            area = area * overlap(pos[a] * dx, (pos[a] + 1) * dx, lo[a], hi[a]) / dx
        along = pos[axis] * dx
        closed = np.where((along > lo[axis]) & (along < hi[axis]), area, 0.0)
        w = 1.0 - closed
        w[w < 0.01] = 0.0
        wall = (pos[axis] == 0) | (pos[axis] == n[axis])
        w[wall] = 0.0
        cut.append(w.astype(dtype))
    velocity = [(rng.random(c.shape) * 2 - 1).astype(dtype) for c in cut]
    solid_velocity = [(0.2 * (rng.random(c.shape) * 2 - 1)).astype(dtype) for c in cut] if with_solid_velocity else None
    return {
        "liquid_phi": liquid_phi.astype(dtype), "solid_phi": solid_phi.astype(dtype), "cut_weights": cut,
        "velocity": velocity, "solid_velocity": solid_velocity, "dx": dx,
    }
