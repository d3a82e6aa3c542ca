"""GPU (-m gpu): the REFERENCE'S OWN TEST, run against the drop-in.  oracle/_ref/ref_tester is the reference's
src/tester.cpp + test/main.cpp, unmodified, compiled in the build container (oracle/Makefile) against this project's
include/kernel.hpp and linked with libspmv_launchers.so: what a maintainer of the reference gets after the three-line
CMake change of INTEGRATION.md.  The binary draws the reference's random 4096 x 4096 matrix at 50 % zeros
(tester.cpp:93-118, seeded from random_device), runs ITS SgemvCPU (tester.cpp:36-45) and the eight launcher slots
(tester.cpp:47-72), and ITS CompareY (tester.cpp:74-88) prints every |cpu - gpu| > 1e-3 to stderr.
A demonstration of the boundary, not the oracle: parity is pinned by tests/test_oracle.py and the golden fixtures.
Skipped where the binary was not built (no /root/reference at build time)."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

EXE = Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "ref_tester"


@pytest.mark.skipif(not EXE.exists(), reason="oracle/_ref/ref_tester not built (no /root/reference at build time)")
def test_the_reference_tester_passes_against_our_launchers(gpu):
    r = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    out = r.stdout
    assert "=== Sparse SGEMV Test ===" in out and "======== CPU start ======" in out and "========== OK ===========" in out
    for name in ("cublas", "wsp0", "wsp1", "asp2", "awsp0", "awsp1", "awsp2", "awsp_ref"):
        assert f"start to launch {name} kernel" in out, name
    mismatches = [l for l in r.stderr.splitlines() if l.startswith("[GPU kernel")]
    assert not mismatches, mismatches[:5]
