"""A host that is not Python: tests/host_gpu_driver.cpp (C++, include/ital_hip.h + the HIP runtime, nothing of ital_amd)
replays one fetch_unlabelled(4) of the golden USPS fixture -- produced by the real reference, tests/golden/make_golden.py --
through libital_hip.so and must return the reference's batch (reference ital/ital.py:84-134).  The drop-in boundary of
SURVEY.md section 8(b) exercised from the language a non-Python maintainer would use."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CXX = os.environ.get("CXX", "g++")      # host-only code: any C++17 compiler (the HIP runtime API is plain C)


def test_cpp_host_replays_the_golden_round(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if shutil.which(CXX) is None:
        pytest.skip("no C++ compiler on this box")
    lib = os.path.join(ROOT, "ital_amd", "libital_hip.so")
    assert os.path.exists(lib), "build the library first (python -m ital_amd.build)"
    exe = str(tmp_path / "host_gpu_driver")
    build = subprocess.run([CXX, "-O1", "-std=c++17", "-w", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "host_gpu_driver.cpp"), "-o", exe, lib,
                            "-Wl,-rpath," + os.path.dirname(lib), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    z = np.load(os.path.join(HERE, "golden", "usps500.npz"))
    X = np.ascontiguousarray(z["X"], dtype=np.float64)
    xfile = str(tmp_path / "X.f64")
    X.tofile(xfile)
    picks = [str(int(i)) for i in z["r0_ret"]]
    run = subprocess.run([exe, xfile, str(X.shape[0]), str(X.shape[1]), repr(float(z["length_scale"])), repr(float(z["var"])),
                          repr(float(z["noise"])), str(int(z["query"])), str(int(z["k"]))] + picks,
                         capture_output=True, text=True, timeout=600)
    print(run.stdout, run.stderr[-2000:])
    assert run.returncode == 0, (run.stdout, run.stderr[-2000:])
    assert "ok (the reference's batch)" in run.stdout
    # ... and through the context-style layer (ital_ctx_*: six calls, no buffer of its own)
    args = [exe, "--ctx"] + run.args[1:]
    run2 = subprocess.run(args, capture_output=True, text=True, timeout=600)
    print(run2.stdout, run2.stderr[-2000:])
    assert run2.returncode == 0 and "picks (context API)" in run2.stdout and "ok (the reference's batch)" in run2.stdout
