"""ctypes front end of the CPU oracle (oracle/mg_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.  PARITY UNPINNED (see the header of mg_oracle.c): the reference
ships no golden vectors and cannot be built here; behaviour is pinned by the reference's own
property checks re-expressed in tests/test_oracle_*.py.

All grids are numpy arrays indexed [k, j, i] (x fastest in memory), labels int32, reals float64
(or float32 with Oracle(f32=True), a same-algorithm single-precision build used to separate
precision effects from logic errors when debugging the HIP path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

INTERIOR, EXTERIOR, DIRICHLET, BOUNDARY = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle shared libraries in place (gcc, OpenMP)."""
    deps = {"libmgoracle.so": ("mg_oracle.c", "mg_fields_oracle.c"), "libmgoracle_f32.so": ("mg_oracle.c",)}

    def stale():
        return force or any(
            not os.path.exists(os.path.join(_HERE, lib))
            or any(os.path.getmtime(os.path.join(_HERE, lib)) < os.path.getmtime(os.path.join(_HERE, src)) for src in srcs)
            for lib, srcs in deps.items()
        )

    if stale():
        import fcntl

        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:  # several test ranks may import at once
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
                force = False


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def effective_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box
    shows every host core but grants a share; spinning one OpenMP thread per visible core there is
    slower than running serially)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


class Oracle:
    def __init__(self, f32=False):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "libmgoracle_f32.so" if f32 else "libmgoracle.so"))
        self.real = np.float32 if f32 else np.float64
        self.creal = C.c_float if f32 else C.c_double
        L = self.lib
        assert L.mgo_real_bytes() == np.dtype(self.real).itemsize
        if "OMP_NUM_THREADS" not in os.environ:
            L.mgo_set_threads(effective_cpus())
        for name in ("mgo_dot", "mgo_squared_l2", "mgo_l2", "mgo_inf_norm", "mgo_ghost_fluid_weight"):
            getattr(L, name).restype = C.c_double
        L.mgo_build_boundary_cells.restype = C.c_int64
        L.mgo_solver_create.restype = C.c_void_p
        L.mgo_solver_band_count.restype = C.c_int64
        L.mgo_solver_band.restype = C.c_void_p
        L.mgo_solver_labels.restype = C.c_void_p

    # -- helpers ---------------------------------------------------------------------------
    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.real)

    @staticmethod
    def lab(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    def set_threads(self, n):
        self.lib.mgo_set_threads(int(n))

    def get_threads(self):
        return self.lib.mgo_get_threads()

    def _wp(self, w):
        """keep-alive list + pointers for optional weights"""
        if w is None:
            return [], (None, None, None)
        ws = [self.arr(a) for a in w]
        return ws, tuple(_ptr(a) for a in ws)

    # -- operators (in place on x / outputs) ---------------------------------------------------
    def jacobi(self, x, b, lab, w=None):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        self.lib.mgo_jacobi(_ptr(x), _ptr(b), _ptr(lab), *wp, nx, ny, nz, None)

    def tiled_gs(self, x, b, lab, odd, forward, w=None):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        self.lib.mgo_tiled_gs(_ptr(x), _ptr(b), _ptr(lab), *wp, nx, ny, nz, int(odd), int(forward))

    def boundary_jacobi(self, x, b, lab, cells, w=None):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.lib.mgo_boundary_jacobi(_ptr(x), _ptr(b), _ptr(lab), _ptr(cells), C.c_int64(len(cells)), *wp, nx, ny, nz)

    def apply_poisson(self, y, x, lab, w=None):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        self.lib.mgo_apply_poisson(_ptr(y), _ptr(x), _ptr(lab), *wp, nx, ny, nz)

    def residual(self, r, x, b, lab, w=None):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        self.lib.mgo_residual(_ptr(r), _ptr(x), _ptr(b), _ptr(lab), *wp, nx, ny, nz)

    def downsample(self, coarse, fine, coarse_lab):
        nz, ny, nx = coarse_lab.shape
        self.lib.mgo_downsample(_ptr(coarse), _ptr(fine), _ptr(coarse_lab), nx, ny, nz)

    def upsample_add(self, fine, coarse, fine_lab):
        nz, ny, nx = fine_lab.shape
        self.lib.mgo_upsample_add(_ptr(fine), _ptr(coarse), _ptr(fine_lab), nx, ny, nz)

    def add_vectors(self, dst, a, bs, s, lab):
        self.lib.mgo_add_vectors(_ptr(dst), _ptr(a), _ptr(bs), self.creal(s), _ptr(lab), C.c_int64(lab.size))

    def add_to_vector(self, dst, a, s, lab):
        self.lib.mgo_add_to_vector(_ptr(dst), _ptr(a), self.creal(s), _ptr(lab), C.c_int64(lab.size))

    def scale_vector(self, v, s, lab):
        self.lib.mgo_scale_vector(_ptr(v), self.creal(s), _ptr(lab), C.c_int64(lab.size))

    def dot(self, a, b, lab):
        nz, ny, nx = lab.shape
        return self.lib.mgo_dot(_ptr(a), _ptr(b), _ptr(lab), nx, ny, nz)

    def squared_l2(self, a, lab):
        nz, ny, nx = lab.shape
        return self.lib.mgo_squared_l2(_ptr(a), _ptr(lab), nx, ny, nz)

    def l2(self, a, lab):
        nz, ny, nx = lab.shape
        return self.lib.mgo_l2(_ptr(a), _ptr(lab), nx, ny, nz)

    def inf_norm(self, a, lab):
        nz, ny, nx = lab.shape
        return self.lib.mgo_inf_norm(_ptr(a), _ptr(lab), nx, ny, nz)

    # -- domain / hierarchy ----------------------------------------------------------------------
    def expanded_layout(self, bnx, bny, bnz):
        dims = (C.c_int * 3)()
        off = (C.c_int * 3)()
        lev = C.c_int()
        self.lib.mgo_expanded_layout(bnx, bny, bnz, dims, off, C.byref(lev))
        return tuple(dims), off[0], lev.value

    def build_expanded_domain(self, base_lab, base_w):
        """buildExpandedDomain of Test.cpp:170-204: labels, 3 weights, (offset, levels)."""
        bnz, bny, bnx = base_lab.shape
        (enx, eny, enz), off, levels = self.expanded_layout(bnx, bny, bnz)
        base_lab = self.lab(base_lab)
        lab = np.empty((enz, eny, enx), dtype=np.int32)
        self.lib.mgo_build_expanded_labels(_ptr(lab), _ptr(base_lab), bnx, bny, bnz, enx, eny, enz, off)
        ws = []
        for axis in range(3):
            shape = [enz, eny, enx]
            shape[2 - axis] += 1
            w = np.empty(shape, dtype=self.real)
            bw = self.arr(base_w[axis])
            self.lib.mgo_build_expanded_weights(_ptr(w), _ptr(bw), axis, bnx, bny, bnz, enx, eny, enz, off)
            ws.append(w)
        self.lib.mgo_set_boundary_labels(_ptr(lab), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), enx, eny, enz)
        return lab, ws, off, levels

    def set_boundary_labels(self, lab, w):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        self.lib.mgo_set_boundary_labels(_ptr(lab), *wp, nx, ny, nz)

    def build_coarse_labels(self, fine):
        nz, ny, nx = fine.shape
        coarse = np.empty((nz // 2, ny // 2, nx // 2), dtype=np.int32)
        self.lib.mgo_build_coarse_labels(_ptr(coarse), _ptr(fine), nx, ny, nz)
        return coarse

    def build_boundary_cells(self, lab, width):
        nz, ny, nx = lab.shape
        out = C.c_void_p()
        n = self.lib.mgo_build_boundary_cells(_ptr(lab), nx, ny, nz, int(width), C.byref(out))
        cells = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(n, 1) * 3,))[: n * 3].copy()
        self.lib.mgo_free(out)
        return cells.reshape(-1, 3)

    def unit_test_exterior(self, lab):
        nz, ny, nx = lab.shape
        return bool(self.lib.mgo_unit_test_exterior(_ptr(lab), nx, ny, nz))

    def unit_test_boundary(self, lab, w=None):
        nz, ny, nx = lab.shape
        keep, wp = self._wp(w)
        return bool(self.lib.mgo_unit_test_boundary(_ptr(lab), *wp, nx, ny, nz))

    def unit_test_coarsening(self, coarse, fine):
        nz, ny, nx = fine.shape
        return bool(self.lib.mgo_unit_test_coarsening(_ptr(coarse), _ptr(fine), nx, ny, nz))

    def ghost_fluid_weight(self, phi0, phi1):
        return self.lib.mgo_ghost_fluid_weight(C.c_double(phi0), C.c_double(phi1))

    def solver(self, lab, w, levels, use_gs, pre_sweeps=1, post_sweeps=1):
        return OracleSolver(self, lab, w, levels, use_gs, pre_sweeps, post_sweeps)


class OracleSolver:
    """GeometricMultigridPoissonSolver (MG.h:10-53) on flat arrays."""

    def __init__(self, orc, lab, w, levels, use_gs, pre_sweeps=1, post_sweeps=1):
        self.o = orc
        self.labels = orc.lab(lab)
        self.w = [orc.arr(a) for a in w]
        nz, ny, nx = self.labels.shape
        self.h = orc.lib.mgo_solver_create(
            _ptr(self.labels), _ptr(self.w[0]), _ptr(self.w[1]), _ptr(self.w[2]), nx, ny, nz, int(levels), int(bool(use_gs))
        )
        if not self.h:
            raise RuntimeError("oracle: multigrid hierarchy could not be built")
        self.h = C.c_void_p(self.h)
        if (pre_sweeps, post_sweeps) != (1, 1):  # benchmark variant; the reference's schedule is 1 / 1
            orc.lib.mgo_solver_set_sweeps(self.h, int(pre_sweeps), int(post_sweeps))

    def close(self):
        if self.h:
            self.o.lib.mgo_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def levels(self):
        return self.o.lib.mgo_solver_levels(self.h)

    def level_dims(self, level):
        d = (C.c_int * 3)()
        self.o.lib.mgo_solver_level_dims(self.h, level, d)
        return tuple(d)

    def level_labels(self, level):
        nx, ny, nz = self.level_dims(level)
        p = self.o.lib.mgo_solver_labels(self.h, level)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(nz, ny, nx)).copy()

    def band(self, level):
        n = self.o.lib.mgo_solver_band_count(self.h, level)
        p = self.o.lib.mgo_solver_band(self.h, level)
        if n == 0:
            return np.zeros((0, 3), dtype=np.int32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n * 3,)).copy().reshape(-1, 3)

    @property
    def coarse_unknowns(self):
        return self.o.lib.mgo_solver_coarse_unknowns(self.h)

    def coarse_solve(self, b):
        nx, ny, nz = self.level_dims(self.levels - 1)
        b = self.o.arr(b)
        x = np.zeros((nz, ny, nx), dtype=self.o.real)
        self.o.lib.mgo_solver_coarse_solve(self.h, _ptr(x), _ptr(b))
        return x

    def apply_vcycle(self, x, b, use_initial_guess=False):
        assert x.dtype == self.o.real and b.dtype == self.o.real and x.flags.c_contiguous
        self.o.lib.mgo_solver_apply_vcycle(self.h, _ptr(x), _ptr(b), int(bool(use_initial_guess)))

    def apply_vcycle_fast(self, x, b, use_initial_guess=False):
        """The optimised CPU comparator (mgo_solver_apply_vcycle_fast): Jacobi smoother, one sweep per stroke only."""
        assert x.dtype == self.o.real and b.dtype == self.o.real and x.flags.c_contiguous
        self.o.lib.mgo_solver_apply_vcycle_fast.restype = C.c_int
        rc = self.o.lib.mgo_solver_apply_vcycle_fast(self.h, _ptr(x), _ptr(b), int(bool(use_initial_guess)))
        if rc != 0:
            raise ValueError("the optimised cycle covers the Jacobi smoother with one sweep per stroke on >= 2 levels")

    def solve_pcg(self, x, b, tol=1e-5, max_iter=2500, use_mg=True):
        stats = (C.c_double * 3)()
        hist = np.zeros(max_iter + 1, dtype=np.float64)
        rc = self.o.lib.mgo_solve_pcg(
            self.h, _ptr(x), _ptr(b), C.c_double(tol), int(max_iter), int(bool(use_mg)), stats, _ptr(hist)
        )
        it = int(stats[0])
        return {"status": rc, "iterations": it, "rel_residual": stats[1], "rel_residual_recomputed": stats[2], "history": hist[: it + 1]}


class FieldsOracle:
    """ctypes front end of oracle/mg_fields_oracle.c (fp64): the plugin-side field pre/post-processing
    (Plug.cpp:716-1207, Util.cpp:5-148).  Grids are numpy [k, j, i] arrays; face grids have one more entry
    along their axis (axis 0 = x = last array dimension)."""

    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "libmgoracle.so"))

    @staticmethod
    def face_shape(shape, axis):
        s = list(shape)
        s[2 - axis] += 1
        return tuple(s)

    @staticmethod
    def _f64(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.float64)

    def material_labels(self, liquid_phi, solid_phi, cw):
        gz, gy, gx = liquid_phi.shape
        out = np.empty(liquid_phi.shape, dtype=np.int32)
        c = [self._f64(a) for a in cw]
        self.lib.mgf_material_labels(_ptr(out), _ptr(self._f64(liquid_phi)), _ptr(self._f64(solid_phi)), _ptr(c[0]), _ptr(c[1]), _ptr(c[2]), gx, gy, gz)
        return out

    def valid_faces(self, material, cw):
        gz, gy, gx = material.shape
        out = []
        for a in range(3):
            v = np.empty(self.face_shape(material.shape, a), dtype=np.uint8)
            self.lib.mgf_valid_faces(a, _ptr(v), _ptr(material), _ptr(self._f64(cw[a])), gx, gy, gz)
            out.append(v)
        return out

    def domain_labels(self, material, eshape, offset):
        gz, gy, gx = material.shape
        ez, ey, ex = eshape
        out = np.empty(eshape, dtype=np.int32)
        self.lib.mgf_domain_labels(_ptr(out), _ptr(material), gx, gy, gz, ex, ey, ez, int(offset))
        return out

    def boundary_weights(self, cw, liquid_phi, valid, material, eshape, offset):
        gz, gy, gx = material.shape
        ez, ey, ex = eshape
        out = []
        phi = self._f64(liquid_phi)
        for a in range(3):
            w = np.empty(self.face_shape(eshape, a), dtype=np.float64)
            self.lib.mgf_boundary_weights(a, _ptr(w), _ptr(self._f64(cw[a])), _ptr(phi), _ptr(valid[a]), _ptr(material), gx, gy, gz, ex, ey, ez, int(offset))
            out.append(w)
        return out

    def rhs(self, material, vel, cw, eshape, offset, solid_vel=None):
        gz, gy, gx = material.shape
        ez, ey, ex = eshape
        out = np.empty(eshape, dtype=np.float64)
        v = [self._f64(a) for a in vel]
        sv = [self._f64(a) for a in solid_vel] if solid_vel is not None else [None] * 3
        c = [self._f64(a) for a in cw]
        self.lib.mgf_rhs(_ptr(out), _ptr(material), *[_ptr(a) for a in v], *[_ptr(a) for a in sv], *[_ptr(a) for a in c], gx, gy, gz, ex, ey, ez, int(offset))
        return out

    def pressure_to_solution(self, pressure, material, eshape, offset):
        gz, gy, gx = material.shape
        ez, ey, ex = eshape
        out = np.empty(eshape, dtype=np.float64)
        self.lib.mgf_pressure_to_solution(_ptr(out), _ptr(self._f64(pressure)), _ptr(material), gx, gy, gz, ex, ey, ez, int(offset))
        return out

    def solution_to_pressure(self, pressure, solution, material, offset):
        gz, gy, gx = material.shape
        ez, ey, ex = solution.shape
        assert pressure.dtype == np.float64 and pressure.flags.c_contiguous
        self.lib.mgf_solution_to_pressure(_ptr(pressure), _ptr(self._f64(solution)), _ptr(material), gx, gy, gz, ex, ey, ez, int(offset))
        return pressure

    def pressure_gradient(self, vel, cw, liquid_phi, pressure, valid, material):
        gz, gy, gx = material.shape
        for a in range(3):
            assert vel[a].dtype == np.float64 and vel[a].flags.c_contiguous
            self.lib.mgf_pressure_gradient(a, _ptr(vel[a]), _ptr(self._f64(cw[a])), _ptr(self._f64(liquid_phi)), _ptr(self._f64(pressure)),
                                           _ptr(valid[a]), _ptr(material), gx, gy, gz)
        return vel

    def divergence(self, material, vel, cw, solid_vel=None):
        gz, gy, gx = material.shape
        out = np.zeros(3)
        v = [self._f64(a) for a in vel]
        sv = [self._f64(a) for a in solid_vel] if solid_vel is not None else [None] * 3
        c = [self._f64(a) for a in cw]
        self.lib.mgf_divergence(_ptr(out), _ptr(material), *[_ptr(a) for a in v], *[_ptr(a) for a in sv], *[_ptr(a) for a in c], gx, gy, gz)
        return float(out[0]), float(out[1]), float(out[2])
