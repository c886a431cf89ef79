"""CPU emulation of the Z-slab V-cycle: the exchange schedule of csrc/mgps_solver.hip::vcycle restated in
Python, with the fp64 oracle's operators doing the arithmetic on each rank's planes and
torch.distributed (gloo) doing the ghost exchange, gather and scatter.

Purpose: a missing or misplaced exchange, a wrong slab extraction of the band list, a wrong tile
colour offset or a wrong collapse changes the result, so "slab run == whole-grid oracle run to
round-off" pins the DESIGN of the multi-GPU path on machines without a GPU.  (The C++ implementation of
the same schedule is pinned on a GPU by tests/test_distributed.py::test_two_slabs_match_single_gpu.)

Storage per distributed level on a rank: [ghost plane | owned planes | ghost plane].  An operator runs
on a temporary copy padded with inert planes so that the oracle's array-origin-relative rules (16^3
tile colouring, even/odd transfer stencils) coincide with the global ones; only owned planes are
copied back -- exactly what the GPU kernels compute.
"""
import numpy as np
import torch
import torch.distributed as dist

EXTERIOR = 1


class SlabEmulation:
    def __init__(self, oracle, labels, weights, levels, use_gs, group=None):
        self.o, self.group = oracle, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.use_gs = use_gs
        self.w = [np.asarray(a, dtype=np.float64) for a in weights]
        self.glob = oracle.solver(labels, self.w, levels, use_gs)  # global hierarchy on every rank
        self.L = self.glob.levels
        self.lab = [self.glob.level_labels(l) for l in range(self.L)]
        self.band = [self.glob.band(l) for l in range(self.L)]
        nz = labels.shape[0]
        assert nz % self.size == 0
        self.nzl = nz // self.size
        planes, self.D = self.nzl, 0
        while self.D < self.L - 1 and planes % 16 == 0:  # same rule as mgps_create_slab
            self.D += 1
            planes //= 2
        assert self.D >= 1
        self.z0 = [(self.rank * self.nzl) >> l for l in range(self.D + 1)]
        self.z1 = [((self.rank + 1) * self.nzl) >> l for l in range(self.D + 1)]
        self.exchanges = 0
        self.bytes_sent = 0
        if self.rank == 0 and self.L - self.D > 1:
            c = self.lab[self.D]
            cz, cy, cx = c.shape  # unit weights == weight 1 on every face (MG.cpp:572-575)
            ones = [np.ones((cz, cy, cx + 1)), np.ones((cz, cy + 1, cx)), np.ones((cz + 1, cy, cx))]
            self.tail = oracle.solver(c, ones, self.L - self.D, use_gs)

    # -- storage ---------------------------------------------------------------------------------
    def new(self, level):
        nz, ny, nx = self.lab[level].shape
        return np.zeros((self.z1[level] - self.z0[level] + 2, ny, nx))

    def owned(self, a):
        return a[1:-1]

    def _slice_global(self, g, lo, hi, fill):
        """planes [lo, hi) of global array g, padded with `fill` outside the domain"""
        n = g.shape[0]
        out = np.full((hi - lo,) + g.shape[1:], fill, dtype=g.dtype)
        a, b = max(lo, 0), min(hi, n)
        if b > a:
            out[a - lo : b - lo] = g[a:b]
        return out

    def _temp(self, level, a, plo, phi):
        """a = [ghost|owned|ghost] -> zero-padded copy with the first owned plane at index plo"""
        nzo = a.shape[0] - 2
        t = np.zeros((plo + nzo + phi,) + a.shape[1:])
        t[plo - 1 : plo + nzo + 1] = a
        lab = self._slice_global(self.lab[level], self.z0[level] - plo, self.z1[level] + phi, EXTERIOR)
        # planes beyond the ghost planes are padding: inert, or the oracle would run its stencil on their
        # active cells and read past the ends of the temporary
        lab[: plo - 1] = EXTERIOR
        lab[plo + nzo + 1 :] = EXTERIOR
        w = None
        if level == 0:
            w = [self._slice_global(self.w[0], self.z0[0] - plo, self.z1[0] + phi, 0.0),
                 self._slice_global(self.w[1], self.z0[0] - plo, self.z1[0] + phi, 0.0),
                 self._slice_global(self.w[2], self.z0[0] - plo, self.z1[0] + phi + 1, 0.0)]
        return t, np.ascontiguousarray(lab, dtype=np.int32), w

    # -- communication ---------------------------------------------------------------------------
    def _plane_band(self, level, gz):
        """(j, i) of the band cells of global plane gz, reference band order"""
        b = self.band[level]
        sel = b[b[:, 2] == gz]
        return sel[:, 1], sel[:, 0]

    def exchange(self, a, level=None, band_only=False):
        """whole ghost planes, or (band_only) just the band cells of the planes, packed"""
        self.exchanges += 1
        ops, recvs = [], []
        for peer, slot_send, slot_recv, gz_send, gz_recv in (
            (self.rank - 1, 1, 0, None, None),
            (self.rank + 1, -2, -1, None, None),
        ):
            if not (0 <= peer < self.size):
                continue
            if band_only:
                lo_side = peer < self.rank
                gz_send = self.z0[level] if lo_side else self.z1[level] - 1
                gz_recv = self.z0[level] - 1 if lo_side else self.z1[level]
                js, is_ = self._plane_band(level, gz_send)
                jr, ir = self._plane_band(level, gz_recv)
                out = torch.from_numpy(np.ascontiguousarray(a[slot_send][js, is_]))
                inc = torch.empty(len(jr), dtype=out.dtype)
                self.bytes_sent += out.numel() * 8
                recvs.append((slot_recv, inc, (jr, ir)))
            else:
                out = torch.from_numpy(np.ascontiguousarray(a[slot_send]))
                inc = torch.empty_like(out)
                self.bytes_sent += out.numel() * 8
                recvs.append((slot_recv, inc, None))
            if out.numel():
                ops.append(dist.P2POp(dist.isend, out, peer, self.group))
            if inc.numel():
                ops.append(dist.P2POp(dist.irecv, inc, peer, self.group))
        for wk in dist.batch_isend_irecv(ops) if ops else []:
            wk.wait()
        for slot, inc, where in recvs:
            if where is None:
                a[slot] = inc.numpy()
            else:
                a[slot][where] = inc.numpy()

    # -- operators on slab storage -------------------------------------------------------------------
    def _band_cells(self, level, plo):
        b = self.band[level]
        sel = (b[:, 2] >= self.z0[level]) & (b[:, 2] < self.z1[level])
        cells = b[sel].copy()
        cells[:, 2] += plo - self.z0[level]
        return cells

    def band_pass(self, level, x, b):
        tx, lab, w = self._temp(level, x, 2, 2)
        tb, _, _ = self._temp(level, b, 2, 2)
        self.o.boundary_jacobi(tx, tb, lab, self._band_cells(level, 2), w)
        x[1:-1] = tx[2:-2]

    def jacobi(self, level, x, b):
        tx, lab, w = self._temp(level, x, 2, 2)
        tb, _, _ = self._temp(level, b, 2, 2)
        self.o.jacobi(tx, tb, lab, w)
        x[1:-1] = tx[2:-2]

    def gs_half(self, level, x, b, odd, forward):
        plo = 32 if self.z0[level] % 32 == 0 else 16  # keeps the global 16^3 tile grid AND its colours
        tx, lab, w = self._temp(level, x, plo, 16)
        tb, _, _ = self._temp(level, b, plo, 16)
        self.o.tiled_gs(tx, tb, lab, odd, forward, w)
        x[1:-1] = tx[plo:-16]

    def residual(self, level, r, x, b):
        tx, lab, w = self._temp(level, x, 2, 2)
        tb, _, _ = self._temp(level, b, 2, 2)
        tr = np.zeros_like(tx)
        self.o.residual(tr, tx, tb, lab, w)
        r[1:-1] = tr[2:-2]

    def restrict(self, level, coarse_b, fine_r):
        tf, _, _ = self._temp(level, fine_r, 2, 2)
        clab = self._slice_global(self.lab[level + 1], self.z0[level + 1] - 1, self.z1[level + 1] + 1, EXTERIOR).astype(np.int32)
        clab[0] = EXTERIOR  # ghost coarse planes are not computed here
        clab[-1] = EXTERIOR
        tc = np.zeros(clab.shape)
        self.o.downsample(tc, tf, np.ascontiguousarray(clab))
        coarse_b[1:-1] = tc[1:-1]

    def prolong_add(self, level, fine_x, coarse_x):
        tf, lab, _ = self._temp(level, fine_x, 2, 2)
        lab[:2] = EXTERIOR  # only owned fine planes are updated
        lab[-2:] = EXTERIOR
        self.o.upsample_add(tf, np.ascontiguousarray(coarse_x), lab)
        fine_x[1:-1] = tf[2:-2]

    # -- the schedule of mgps_solver.hip::vcycle -------------------------------------------------------
    def band_passes(self, level, x, b, first):
        """first: None (ghosts complete), "full" or "band" -- what the first pass needs"""
        for it in range(3):
            mode = first if it == 0 else "band"
            if mode is not None:
                self.exchange(x, level, band_only=(mode == "band"))
            self.band_pass(level, x, b)

    def smooth_stroke(self, level, x, b, down, fresh):
        self.band_passes(level, x, b, None if fresh else "full")
        if self.use_gs:
            for n, (odd, fwd) in enumerate(((True, True), (False, True)) if down else ((False, False), (True, False))):
                self.exchange(x, level, band_only=(n == 0))  # after band passes only band cells are stale
                self.gs_half(level, x, b, odd, fwd)
        else:
            self.exchange(x, level, band_only=True)
            self.jacobi(level, x, b)
        self.band_passes(level, x, b, "full")  # the full-domain smoother rewrote everything

    def collapsed_tail(self, b_c, x_c):
        mine = torch.from_numpy(np.ascontiguousarray(self.owned(b_c)))
        parts = [torch.empty_like(mine) for _ in range(self.size)] if self.rank == 0 else None
        dist.gather(mine, parts, dst=0, group=self.group)
        out = None
        if self.rank == 0:
            full_b = torch.cat(parts).numpy()
            if self.L - self.D > 1:
                full_x = np.zeros_like(full_b)
                self.tail.apply_vcycle(full_x, full_b, False)
            else:
                full_x = self.glob.coarse_solve(full_b)
            out = [torch.from_numpy(np.ascontiguousarray(p)) for p in np.split(full_x, self.size)]
        got = torch.empty_like(mine)
        dist.scatter(got, out, src=0, group=self.group)
        x_c[1:-1] = got.numpy()

    def vcycle(self, x, b, use_initial_guess):
        D = self.D
        xs = [x] + [self.new(l) for l in range(1, D + 1)]
        bs = [b] + [self.new(l) for l in range(1, D + 1)]
        fresh = False
        if not use_initial_guess:
            x[:] = 0
            fresh = True
        self.smooth_stroke(0, x, b, True, fresh)
        for l in range(D):
            if l > 0:
                xs[l][:] = 0
                self.smooth_stroke(l, xs[l], bs[l], True, True)
            self.exchange(xs[l], l, band_only=True)  # the stroke ended with band passes
            r = self.new(l)
            self.residual(l, r, xs[l], bs[l])
            self.exchange(r)
            self.restrict(l, bs[l + 1], r)
        self.collapsed_tail(bs[D], xs[D])
        for l in range(D - 1, -1, -1):
            self.exchange(xs[l + 1])
            self.prolong_add(l, xs[l], xs[l + 1])
            self.smooth_stroke(l, xs[l], bs[l], False, False)
