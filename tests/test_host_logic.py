"""CPU: the host-side logic (workloads, partition), the C-ABI library's exports, the launcher
surface's symbols, and the loud failure without a GPU.  No compute runs here."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _no_gpu_early():
    import torch
    return not torch.cuda.is_available()


# ---- C ABI ---------------------------------------------------------------------------------
def _declared_symbols():
    text = (ROOT / "include" / "spmv_hip.h").read_text()
    return sorted(set(re.findall(r"SPMV_API[^;(]*?\b(spmv_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    declared = _declared_symbols()
    assert len(declared) >= 19
    lib = pkg.capi.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/spmv_hip.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(pkg.capi.SIGNATURES) == declared


def test_library_exports_nothing_else(pkg):
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg.capi.LIB_PATH)], capture_output=True, text=True,
                         check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms and all(s.startswith("spmv_") for s in syms), syms


def test_dist_library_exports_every_declared_symbol(pkg):
    """include/spmv_dist.h (the row-block exchange over RCCL for a C++ caller) vs libspmv_dist.so."""
    text = (ROOT / "include" / "spmv_dist.h").read_text()
    declared = sorted(set(re.findall(r"SPMV_API[^;(]*?\b(spmv_dist_\w+)\s*\(", text)))
    assert len(declared) >= 12
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg.capi.DIST_LIB_PATH)], capture_output=True, text=True,
                         check=True).stdout
    syms = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert syms == declared
    # and the Python binding bench.py --backend native drives it with covers exactly the header
    assert sorted(pkg.dist_native.SIGNATURES) == declared


@pytest.mark.skipif(not _no_gpu_early(), reason="only meaningful on a box without a GPU")
def test_dist_selftest_exits_nonzero_without_a_device(pkg):
    p = subprocess.run([str(pkg.capi.DIST_SELFTEST_PATH)], capture_output=True, text=True)
    assert p.returncode != 0 and "HIP error" in p.stderr


def test_launcher_library_has_the_reference_symbols(pkg):
    """The five symbols an unmodified tester.o needs (verified there with nm) and the
    five declared-but-uncalled ones of kernel.hpp:8-17, Itanium-mangled, C++ linkage."""
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg.capi.LAUNCHERS_PATH)], capture_output=True,
                         text=True, check=True).stdout
    need = ["_Z15cublas_gemv_gpuiiPfS_S_", "_Z12wsp_gemv_gpuiiPfS_S_i", "_Z12asp_gemv_gpuiiPfS_S_i",
            "_Z13awsp_gemv_gpuiiPfS_S_i", "_Z17awsp_ref_gemv_gpuiiPfS_S_", "_Z15tiling_gemv_gpuiiPfS_S_",
            "_Z14naive_gemv_gpuiiPfS_S_", "_Z18csr_naive_gemv_gpuiiPfS_S_", "_Z19csr_tiling_gemv_gpuiiPfS_S_",
            "_Z15wsp_sm_gemv_gpuiiPfS_S_"]
    for s in need:
        assert s in out, s


def test_variant_names(pkg):
    lib = pkg.capi.lib()
    for name, v in pkg.capi.ALL_VARIANTS.items():
        assert lib.spmv_variant_name(v).decode() == name
    assert lib.spmv_variant_name(99).decode() == "unknown"


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_compute_entry_points_fail_loudly_without_a_device(pkg):
    capi = pkg.capi
    assert capi.device_count() == 0
    rp = np.array([0, 1], np.int32); ci = np.array([0], np.int32); va = np.array([1.0], np.float32)
    with pytest.raises(capi.SpmvError) as e:
        capi.CsrMatrix.from_host(1, 1, rp, ci, va)
    assert e.value.status == capi.ERR_NO_DEVICE and "no CPU path" in str(e.value)
    with pytest.raises(capi.SpmvError):
        capi.CsrMatrix.from_dense_host(np.ones((2, 2), np.float32))


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_tester_executable_exits_nonzero_without_a_device(pkg):
    p = subprocess.run([str(pkg.capi.TESTER_PATH), "32", "32"], capture_output=True, text=True)
    assert p.returncode != 0
    assert "start to launch cublas kernel" in p.stdout       # tester.cpp:67 banner
    assert "HIP error" in p.stderr                            # kernel.hpp:21-28 convention


def test_product_never_touches_the_oracle():
    """The checker must not be reachable from the product path."""
    for p in list((ROOT / "spmv-test_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in {".py", ".hip", ".hpp", ".cpp", ".h", ""} and "build" not in p.parts \
                and "lib" not in p.parts and "bin" not in p.parts and p.name != "Makefile":
            text = p.read_text(errors="ignore")
            assert "liboracle" not in text and "oracle_" not in text and "load_oracle" not in text, p
    mk = (ROOT / "spmv-test_amd" / "Makefile").read_text()
    assert "oracle" not in mk


# ---- workloads -----------------------------------------------------------------------------
def test_row_lengths_are_exact_and_deterministic(pkg):
    W = pkg.workloads
    for name in ("c2", "c3", "c4"):
        w = W.config(name, scale=1 / 64 if name != "c2" else 1 / 8)
        L = W.row_lengths(w)
        assert L.sum() == w.nnz and L.min() >= 1
        # every 65 536-row block carries exactly its share
        assert np.all(L.reshape(-1, W.BLOCK_ROWS).sum(axis=1) == w.mean * W.BLOCK_ROWS)
        # a shard in the middle equals the same rows of the whole
        r0, n = W.BLOCK_ROWS // 2 + 17, W.BLOCK_ROWS + 1001
        if r0 + n <= w.rows:
            assert np.array_equal(W.row_lengths(w, r0, n), L[r0:r0 + n])
    w4 = W.CONFIGS["c4"]
    assert w4.nnz == 268_435_456 and W.CONFIGS["c3"].nnz == 134_217_728 and W.CONFIGS["c2"].nnz == 16_777_216
    assert W.algorithmic_bytes(w4.rows, w4.cols, w4.nnz) == 2_348_810_244
    assert W.algorithmic_bytes(1 << 20, 1 << 20, 1 << 24) == 146_800_644


def test_mixed_distribution_shape(pkg):
    W = pkg.workloads
    L = W.row_lengths(W.config("c4", scale=1 / 16))
    assert 0.55 < np.mean(L <= 9) < 0.65          # the short class dominates
    assert np.mean(L > 1024) > 5e-5 and L.max() <= 3072
    Lp = W.row_lengths(W.config("c3", scale=1 / 8))
    assert np.median(Lp) < 10 and Lp.max() > 10_000   # heavy tail


@pytest.mark.parametrize("band", [0, 4096])
def test_synthetic_rows_are_sorted_unique_in_range(pkg, oracle, band):
    W = pkg.workloads
    w = W.Workload("t", 1 << 16, 1 << 16, "mixed", 16, band=band)
    rp = W.row_ptr(w)
    ci, va = oracle.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp)
    assert ci.min() >= 0 and ci.max() < w.cols
    d = np.diff(ci.astype(np.int64))
    starts = rp[1:-1]                         # first element of rows 1.. : exempt from the ascending check
    inner = np.ones(len(d), bool)
    inner[starts[starts <= len(d)] - 1] = False
    assert np.all(d[inner] > 0)               # strictly ascending inside every row -> unique
    assert np.all(va != 0) and np.all(np.abs(va) < 1)
    if band:
        rows = np.repeat(np.arange(w.rows), np.diff(rp))
        L = np.repeat(np.diff(rp), np.diff(rp))
        half = np.maximum(band, 8 * L) // 2 + 1
        centre = rows * w.cols // w.rows
        inside = np.abs(ci - centre) <= half
        edge = (centre < half) | (centre > w.cols - half)
        assert np.all(inside | edge)
    # shard == same rows of the whole
    r0, r1 = 1000, 5000
    rps = (rp[r0:r1 + 1] - rp[r0]).astype(np.int32)
    cs, vs = oracle.synth_fill(w.seed, r0, r1, w.rows, w.cols, w.band, rps)
    assert np.array_equal(cs, ci[rp[r0]:rp[r1]]) and np.array_equal(vs, va[rp[r0]:rp[r1]])


# ---- partition -----------------------------------------------------------------------------
def test_balanced_bounds_split_by_nnz(pkg):
    P, W = pkg.partition, pkg.workloads
    w = W.config("c3", scale=1 / 16)
    L = W.row_lengths(w)
    rp = np.concatenate([[0], np.cumsum(L)])
    for parts in (2, 4, 8):
        b = P.balanced_row_bounds(rp, parts)
        assert b[0] == 0 and b[-1] == w.rows and np.all(np.diff(b) >= 0)
        share = np.diff(rp[b])
        assert share.sum() == rp[-1]
        assert share.max() - share.min() <= 2 * L.max()      # equal nnz +- one row
    e = P.equal_row_bounds(10, 4)
    assert list(e) == [0, 2, 5, 7, 10]


def test_shard_rebases_to_int32_and_refuses_overflow(pkg):
    P = pkg.partition
    rp = np.array([0, 5, 5, (1 << 31) + 10, (1 << 31) + 20], np.int64)   # > 2^31 nonzeros overall
    s = P.shard_row_ptr(rp, 0, 2)
    assert s.dtype == np.int32 and list(s) == [0, 5, 5]
    s2 = P.shard_row_ptr(rp, 3, 4)
    assert list(s2) == [0, 10]
    with pytest.raises(ValueError):
        P.shard_row_ptr(rp, 0, 4)


@pytest.mark.skipif(not Path("/root/reference/src/include/kernel.hpp").exists(), reason="/root/reference not present")
def test_launcher_prototypes_match_the_reference_header():
    """Every prototype of the reference's kernel.hpp:8-17 appears in include/kernel.hpp with the same
    name and parameter list (the drop-in is symbol-for-symbol)."""
    def protos(text):
        out = {}
        for m in re.finditer(r"^void\s+(\w+_gemv_gpu)\s*\(([^)]*)\)\s*;", text, flags=re.M):
            params = [re.sub(r"\s+", " ", p.strip()) for p in m.group(2).split(",")]
            out[m.group(1)] = [re.sub(r"\s*\w+$", "", p).replace(" *", "*").strip() for p in params]   # types only
        return out
    ref = protos(Path("/root/reference/src/include/kernel.hpp").read_text())
    ours = protos((ROOT / "include" / "kernel.hpp").read_text())
    assert len(ref) == 10
    for name, types in ref.items():
        assert name in ours, name
        assert ours[name] == types, (name, ours[name], types)


def test_stencil7_helper_matches_a_kron_laplacian(pkg):
    """workloads.stencil7 (host-built 7-point stencil used by bench extras and GPU tests): pattern equals the
    Kronecker-sum Laplacian, columns ascend inside every row, the scaling is per row."""
    import scipy.sparse as sp
    n = 5
    N, rp, ci, va = pkg.workloads.stencil7(n)
    assert N == n ** 3 and len(rp) == N + 1 and rp[-1] == len(ci) == len(va)
    A = sp.csr_matrix((va, ci, rp), shape=(N, N))
    T = sp.diags([1, 1, 1], [-1, 0, 1], shape=(n, n))
    I = sp.identity(n)
    L = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()
    assert (abs(A.sign()) != (L != 0)).nnz == 0
    for r in range(N):
        cols = ci[rp[r]:rp[r + 1]]
        assert np.all(np.diff(cols) > 0)
        row = va[rp[r]:rp[r + 1]]
        d = row[cols == r][0]
        assert d > 0 and np.allclose(row[cols != r], -d / 6.0)
