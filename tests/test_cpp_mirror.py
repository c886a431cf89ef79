"""The C++ host mirror (geometricmultigridpressuresolver_amd/host/mgps_hdk_mirror.hpp) compiles with
g++ against include/mgps.h, links libmgps.so and runs tests/cpp/mirror_smoke.cpp: MG-PCG on the
reference's simple test domain through HDK::solveGeometricConjugateGradient.  Without a HIP device
the program must report MGPS_ERR_NO_DEVICE (exit 77) -- there is no CPU path to fall back to."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "mirror_smoke.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "build", "mirror_smoke")
CSRC = os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "csrc")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [
        "g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
        "-I" + os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "host"), SRC, "-o", OUT,
        "-L" + CSRC, "-lmgps", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib",
    ]
    subprocess.check_call(cmd)


def run():
    return subprocess.run([OUT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)


def test_mirror_compiles_and_fails_loudly_without_gpu():
    import torch

    build()
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: covered by the gpu test")
    res = run()
    assert res.returncode == 77, res.stdout
    assert "no HIP device" in res.stdout and "expanded 64x64x64 offset 8 levels 4" in res.stdout


@pytest.mark.gpu
def test_mirror_solves_on_gpu():
    build()
    res = run()
    assert res.returncode == 0, res.stdout
    assert "iterations" in res.stdout


def test_host_setup_under_sanitizers():
    """The host-side set-up code (hierarchy, tile-local band lists, band groups, slab levels and deep halos)
    compiled with g++ -fsanitize=address,undefined and driven over noisy domains, all band widths and stage
    depths: no report, every group replay bit-exact (tests/cpp/host_setup_sanitize.cpp).  GPU code cannot run
    under a sanitizer on this pool; this is the CPU half."""
    src = os.path.join(ROOT, "tests", "cpp", "host_setup_sanitize.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "build", "host_setup_sanitize")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, os.path.join(CSRC, "mgps_host.cpp"), "-o", out, "-lpthread",
    ])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-4000:]
    assert res.stdout.count(" ok: ") == 6 and "ERROR" not in res.stdout and "runtime error" not in res.stdout, res.stdout[-4000:]


def test_flatten_roundtrip_and_cmake_configures_without_houdini(tmp_path):
    """The HDK-free half of the Houdini shim (geometricmultigridpressuresolver_amd/host/): flattenGrid / unflattenGrid
    round-trip on a 16^3-tiled stand-in for UT_VoxelArray (tests/cpp/flatten_roundtrip.cpp), and the top-level
    CMakeLists.txt configures on a machine without Houdini -- the DOP target is skipped with a message, the library,
    oracle and host-test targets remain (the reference's build fails without $HFS, /root/reference/CMakeLists.txt:10-14)."""
    import shutil

    out = os.path.join(ROOT, "tests", "cpp", "build", "flatten_roundtrip")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "host"),
                           os.path.join(ROOT, "tests", "cpp", "flatten_roundtrip.cpp"), "-o", out, "-lpthread"])
    res = subprocess.run([out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode == 0 and "flatten round trip ok" in res.stdout, res.stdout
    if shutil.which("cmake") is None:
        pytest.skip("no cmake")
    cfg = subprocess.run(["cmake", "-S", ROOT, "-B", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert cfg.returncode == 0, cfg.stdout[-3000:]
    assert "Houdini not found: the DOP plugin target is skipped" in cfg.stdout
    targets = subprocess.run(["cmake", "--build", str(tmp_path), "--target", "help"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    assert "mgps_build" in targets and "mgoracle" in targets and "HDK_GeometricFreeSurfacePressureSolver" not in targets
    # the shim sources keep the reference's node surface: class, node type, description and the twelve parameter tokens
    shim = open(os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "host", "HDK_GeometricFreeSurfacePressureSolver.cpp")).read()
    header = open(os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "host", "HDK_GeometricFreeSurfacePressureSolver.h")).read()
    for token in ('"HDK_GeometricFreeSurfacePressureSolver"', '"HDK Geometric Free Surface Pressure Solver"', "GAS_NAME_SURFACE", "GAS_NAME_VELOCITY",
                  "GAS_NAME_COLLISION", "GAS_NAME_COLLISIONVELOCITY", '"cutCellWeights"', "GAS_NAME_PRESSURE", '"useOldPressure"', "GAS_NAME_DENSITY",
                  '"validFaces"', "SIM_NAME_TOLERANCE", '"maxIterations"', '"useMGPreconditioner"', "initializeSIM", "mgps_project_free_surface"):
        assert token in shim, token
    assert "DECLARE_DATAFACTORY(HDK_GeometricFreeSurfacePressureSolver, GAS_SubSolver" in header and "solveGasSubclass" in header


def test_dop_shim_compiles_against_hdk_mock(tmp_path):
    """The DOP shim (host/HDK_GeometricFreeSurfacePressureSolver.{h,cpp}: the reference's node surface,
    /root/reference/Source/HDK_GeometricFreeSurfacePressureSolver.h:14-55, around ONE mgps_project_free_surface call) goes
    through the compiler front end against tests/hdk_mock/ -- declarations of the slice of the HDK it calls, test
    infrastructure only -- so that a drift of include/mgps_fields.h (mgps_projection's members) or include/mgps.h under the
    shim breaks a test on a machine without Houdini.  The check must bite: the same compile with one member of
    mgps_projection renamed in a copy of the header fails."""
    import shutil

    host = os.path.join(ROOT, "geometricmultigridpressuresolver_amd", "host")
    shim = os.path.join(host, "HDK_GeometricFreeSurfacePressureSolver.cpp")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "hdk_mock"), "-I" + host]
    ok = subprocess.run(base + ["-I" + os.path.join(ROOT, "include"), shim], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert ok.returncode == 0, ok.stdout[-4000:]
    drift = tmp_path / "include"
    shutil.copytree(os.path.join(ROOT, "include"), drift)
    text = (drift / "mgps_fields.h").read_text()
    assert "use_gauss_seidel" in text
    (drift / "mgps_fields.h").write_text(text.replace("use_gauss_seidel", "use_gauss_seidel_renamed"))
    bad = subprocess.run(base + ["-I" + str(drift), shim], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert bad.returncode != 0 and "use_gauss_seidel" in bad.stdout
