import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the product library is built in-tree by __graft_entry__.build(); a tree that arrives without it (or with
    # sources newer than it) is built here once, with hipcc, before any test imports the package
    csrc = os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "csrc")
    lib = os.path.join(csrc, "libmgps.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs):
        import subprocess

        subprocess.check_call(["make", "-C", csrc, "-j4", "all"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    from oracle.mg_oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def oracle32():
    from oracle.mg_oracle import Oracle

    return Oracle(f32=True)


def make_domain(kind, g, levels=None, solver_shape=None, dtype=np.float32):
    """kind in {'simple', 'complex', 'solid'} -> (labels uint8, weights[3], offset, levels, dx)."""
    from geometricmultigridpressuresolver_amd import domains as D

    if kind == "random":  # blobs of every label, fractional weights everywhere (g = the seed): general rows in every band box, ragged rows
        from test_device_setup import random_domain

        shape, lev = (64, 64, 96), 3
        lab, w = random_domain(shape, lev, g, closed_faces=False)
        return lab, [a.astype(dtype) for a in w], 0, lev, 1.0 / shape[2]
    if kind == "wide":  # non-cubic free-surface box, 256 cells along x (one wavefront per row; with options.stencil_path = 2: the plane-marching sweep)
        bl, bw, dx = D.build_complex_domain((g, g, 248), dtype=dtype)
    elif kind == "widesolid":  # free surface + cut-cell solid box, 264 x 40 x 32 solver grid: ragged in x (256 + 8) and y (2 x 16 + 8)
        bl, bw, dx = D.build_complex_domain((g, g + 8, 256), use_solid=True, dtype=dtype)
    elif kind == "wide512":  # x extent 512 (two wavefronts per row in the quad kernels)
        bl, bw, dx = D.build_complex_domain((g, g, 500), dtype=dtype)
    elif kind == "odd":
        bl, bw, dx = D.build_complex_domain(g, use_solid=True, dtype=dtype)
    elif kind == "simple":
        bl, bw, dx = D.build_simple_domain(g, 1, dtype=dtype)
    else:
        bl, bw, dx = D.build_complex_domain(g, use_solid=(kind == "solid"), dtype=dtype)
    lab, w, off, lev = D.expand_domain(bl, bw, levels=levels, solver_shape=solver_shape)
    return lab, w, off, lev, dx


@pytest.fixture(scope="session")
def domain_factory():
    cache = {}

    def get(kind, g, levels=None, solver_shape=None):
        key = (kind, g, levels, solver_shape)
        if key not in cache:
            cache[key] = make_domain(kind, g, levels, solver_shape)
        return cache[key]

    return get


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
