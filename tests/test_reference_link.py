"""CPU, this container only: the UNMODIFIED reference tester (src/tester.cpp + test/main.cpp)
compiles against include/kernel.hpp and links against libspmv_launchers.so -- the drop-in claim
at the link level.  Nothing from /root/reference is copied; the files are compiled where they lie
and the binary goes to a temp dir.  Skipped on the GPU box (no /root/reference there)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")


@pytest.mark.skipif(not (REF / "src" / "tester.cpp").exists(), reason="/root/reference not present")
def test_unmodified_reference_tester_links_against_our_launchers(pkg, tmp_path):
    exe = tmp_path / "ref_tester"
    libdir = pkg.capi.LAUNCHERS_PATH.parent
    cmd = ["g++", "-std=c++17", "-O2", "-w",
           f"-I{ROOT / 'include'}",                 # our kernel.hpp shadows the reference's (CUDA) one
           f"-I{REF / 'src' / 'include'}",          # tester.hpp comes from the reference
           str(REF / "src" / "tester.cpp"), str(REF / "test" / "main.cpp"),
           f"-L{libdir}", "-lspmv_launchers", "-lspmv_hip", f"-Wl,-rpath,{libdir}", "-o", str(exe)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    for sym in ("_Z15cublas_gemv_gpuiiPfS_S_", "_Z12wsp_gemv_gpuiiPfS_S_i", "_Z12asp_gemv_gpuiiPfS_S_i",
                "_Z13awsp_gemv_gpuiiPfS_S_i", "_Z17awsp_ref_gemv_gpuiiPfS_S_"):
        assert sym in out
    # without a GPU the first launcher must stop the process the way CUDA_CHECK does (kernel.hpp:21-28)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "start to launch cublas kernel" in r.stdout and "HIP error" in r.stderr
