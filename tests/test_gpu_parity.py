"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bar (task statement): indices bit-exact; fp32 values within 1e-5 (see _util.RTOL for how the
bound is normalised); the SCALAR variant -- same operation order as SgemvCPU -- bit-identical."""
import os
import subprocess

import numpy as np
import pytest

from _util import DeviceProblem, assert_close_to_oracle, synth_problem
from conftest import load_golden

pytestmark = pytest.mark.gpu

VARIANT_NAMES = ["scalar", "wave", "wave_pipe", "vector", "adaptive", "tiled", "panel", "auto", "xskip"]


def _check_all_variants(pkg, oracle, prob, what):
    y_seq = oracle.spmv(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    for name in VARIANT_NAMES:
        try:
            y = prob.run(pkg.capi.ALL_VARIANTS[name])
        except pkg.capi.SpmvError as e:
            if name == "xskip" and "dense-ish" in str(e):      # large sparse matrix: that variant says it is not for it
                continue
            raise
        assert not np.isnan(y).any() or np.isnan(y_seq).any(), f"{what}/{name}: rows left unwritten"
        if name == "scalar":
            assert np.array_equal(y.view(np.uint32), y_seq.view(np.uint32)), f"{what}: scalar not bit-identical"
        assert_close_to_oracle(y, y64, mag, f"{what}/{name}")


# ---- golden fixtures (reference-built CSR) ---------------------------------------------------
def test_golden_fixtures_all_variants(pkg, oracle, gpu, golden):
    prob = DeviceProblem(pkg, gpu, golden.N, golden.M, golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    _check_all_variants(pkg, oracle, prob, golden.name)
    # and against the fixture's own expected output (the dense SgemvCPU loop)
    y = prob.run(pkg.capi.SCALAR)
    assert np.array_equal(y, golden.y)


def test_device_dense_to_csr_is_bit_exact(pkg, gpu, golden):
    """spmv_csr_from_dense_host vs the reference's CSRMatrix arrays (matrix_csr.cpp:5-23)."""
    A = pkg.capi.CsrMatrix.from_dense_host(golden.A)
    assert (A.rows, A.cols, A.nnz) == (golden.N, golden.M, len(golden.vals))
    rp, ci, va = A.download()
    assert np.array_equal(rp[:-1], golden.ref_row_ptrs) and rp[-1] == len(golden.vals)
    assert np.array_equal(ci, golden.col_idx)
    assert np.array_equal(va.view(np.uint32), golden.vals.view(np.uint32))


def test_dense_conversion_keeps_nan_drops_negative_zero(pkg, oracle, gpu):
    A = np.zeros((70, 50), np.float32)
    A[3, 4] = np.nan; A[5, 4] = -0.0; A[6, 4] = 1e-45; A[69, 49] = 2.0
    m = pkg.capi.CsrMatrix.from_dense_host(A)
    rp, ci, va = m.download()
    orp, oci, ova = oracle.csr_from_dense(A)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
    assert np.array_equal(va.view(np.uint32), ova.view(np.uint32))
    assert list(ci[rp[4]:rp[5]]) == [3, 6]



@pytest.mark.parametrize("M,N", [(1, 1), (63, 5), (64, 64), (65, 130), (200, 33), (1000, 257), (4097, 66), (3, 5000)])
def test_device_dense_to_csr_odd_shapes(pkg, oracle, gpu, M, N):
    """The 64 x 64 tile transposition and the multi-workgroup scan on shapes that are not multiples of anything
    (ragged last slab, ragged last column tile, a scan just above and below its single-workgroup limit)."""
    A, _ = pkg.workloads.dense_random(M, N, 0.7, seed=M * 131 + N)
    m = pkg.capi.CsrMatrix.from_dense_host(A)
    rp, ci, va = m.download()
    orp, oci, ova = oracle.csr_from_dense(A)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
    assert np.array_equal(va.view(np.uint32), ova.view(np.uint32))
    m.close()

def test_run_host_path(pkg, oracle, gpu):
    g = load_golden("g256x384_10pct")
    A = pkg.capi.CsrMatrix.from_host(g.N, g.M, g.row_ptr, g.col_idx, g.vals)
    y = np.full(g.N, np.nan, np.float32)
    ms = A.run_host(pkg.capi.SCALAR, g.x, y)
    assert ms > 0 and np.array_equal(y, g.y)
    y2 = np.full(g.N, np.nan, np.float32)
    A.run_host(pkg.capi.ADAPTIVE, g.x, y2)       # plans on demand
    y64, mag = oracle.spmv_f64(g.row_ptr, g.col_idx, g.vals, g.x)
    assert_close_to_oracle(y2, y64, mag, "run_host/adaptive")


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_dense_gemv_slots(pkg, oracle, gpu, mode):
    import torch
    g = load_golden("g256x384_10pct")
    dA = torch.from_numpy(g.A).to(gpu); dx = torch.from_numpy(g.x).to(gpu)
    dy = torch.full((g.N,), float("nan"), device=gpu)
    pkg.capi.dense_gemv(dA, dx, dy, mode)
    torch.cuda.synchronize()
    y = dy.cpu().numpy()
    if mode in (0, 1):       # same order as SgemvCPU, unfused
        assert np.array_equal(y.view(np.uint32), g.y.view(np.uint32))
    y64, mag = oracle.spmv_f64(g.row_ptr, g.col_idx, g.vals, g.x)
    assert_close_to_oracle(y, y64, mag, f"dense mode {mode}")


# ---- synthetic workloads at sizes the oracle finishes in seconds -------------------------------
@pytest.mark.parametrize("name,scale,band", [("c2", 1 / 8, 0), ("c3", 1 / 16, 0), ("c4", 1 / 16, 0),
                                             ("c4", 1 / 16, 4096), ("c3", 1 / 32, 65536)])
def test_synthetic_configs_all_variants(pkg, oracle, gpu, name, scale, band):
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    _check_all_variants(pkg, oracle, prob, w.name)


# ---- edge cases ------------------------------------------------------------------------------------
def _csr(lengths, cols, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = np.asarray(lengths, np.int64)
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(cols, size=int(l), replace=False)) for l in lengths] or
                        [np.zeros(0, np.int64)]).astype(np.int32)
    va = rng.uniform(-1, 1, size=int(rp[-1])).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    return rp, ci, va, x


EDGE_CASES = {
    "one_row_spanning_30_chunks": ([120_000], 200_000),
    "long_row_between_short_rows": ([3, 0, 5] + [20_000] + [1] * 700 + [9000, 2, 0, 0], 50_000),
    "rows_exactly_chunk_aligned": ([4096, 4096, 2048, 2048, 4096], 10_000),
    "chunk_ends_on_row_boundary_then_empties": ([4096, 0, 0, 0, 17], 5_000),
    "all_rows_empty": ([0] * 1000, 64),
    "trailing_empty_rows": ([5, 7] + [0] * 5000, 100),
    "leading_empty_rows": ([0] * 5000 + [5, 7], 100),
    "single_element": ([1], 1),
    "many_tiny_rows": ([1] * 20_000, 3000),
    "lengths_around_short_threshold": ([31, 32, 33, 34, 63, 64, 65, 127, 128, 129] * 40, 4000),
    "wide_matrix_few_rows": ([7000, 1, 6999], 1_000_000),
    "tall_matrix_one_col": ([1] * 9000, 1),
    "not_multiple_of_anything": ([13] * 777 + [0] + [4099], 5003),
}


@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_edge_cases_all_variants(pkg, oracle, gpu, case):
    lengths, cols = EDGE_CASES[case]
    rp, ci, va, x = _csr(lengths, cols, seed=len(case))
    prob = DeviceProblem(pkg, gpu, len(lengths), cols, rp, ci, va, x)
    _check_all_variants(pkg, oracle, prob, case)


def test_zero_rows_and_zero_nnz(pkg, gpu):
    import torch
    capi = pkg.capi
    A = capi.CsrMatrix.from_host(0, 5, np.array([0], np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    x = torch.zeros(8, device=gpu); y = torch.zeros(8, device=gpu)
    for v in capi.VARIANTS.values():
        A.plan(v); A.run(v, x, y)
    torch.cuda.synchronize()


def test_error_conventions(pkg, gpu):
    import torch
    capi = pkg.capi
    g = load_golden("g128_half")
    A = capi.CsrMatrix.from_host(g.N, g.M, g.row_ptr, g.col_idx, g.vals)
    x = torch.zeros(g.M, device=gpu); y = torch.zeros(g.N, device=gpu)
    with pytest.raises(capi.SpmvError) as e:
        A.run(17, x, y)
    assert e.value.status == capi.ERR_VARIANT
    with pytest.raises(capi.SpmvError) as e:
        A.run(capi.ADAPTIVE, x, y)            # not planned yet
    assert e.value.status == capi.ERR_NOT_PLANNED
    with pytest.raises(capi.SpmvError) as e:
        capi.CsrMatrix.from_host(2, 2, np.array([0, 1, 5], np.int32), np.array([0], np.int32),
                                 np.array([1.0], np.float32))
    assert e.value.status == capi.ERR_INVALID


def test_malformed_csr_is_refused_before_any_kernel_sees_it(pkg, gpu):
    """spmv_csr_validate (called by both create entry points): the kernels index x and LDS with these numbers."""
    import torch
    capi = pkg.capi
    va = np.ones(4, np.float32)
    cases = {
        "column index of element 2": (np.array([0, 2, 4], np.int32), np.array([0, 1, 7, 1], np.int32)),    # col >= cols
        "column index of element 0": (np.array([0, 2, 4], np.int32), np.array([-1, 1, 0, 1], np.int32)),   # negative
        "row_ptr decreases or leaves [0, nnz] at row 1": (np.array([0, 3, 2, 4], np.int32), np.array([0, 1, 0, 1], np.int32)),
    }
    for needle, (rp, ci) in cases.items():
        rows = len(rp) - 1
        with pytest.raises(capi.SpmvError) as e:
            capi.CsrMatrix.from_host(rows, 2, rp, ci, va)
        assert e.value.status == capi.ERR_INVALID and needle in str(e.value), str(e.value)
        d_rp, d_ci, d_va = (torch.from_numpy(a).to(gpu) for a in (rp, ci, va))
        with pytest.raises(capi.SpmvError) as e:
            capi.CsrMatrix.from_device(rows, 2, d_rp, d_ci, d_va)
        assert e.value.status == capi.ERR_INVALID and needle in str(e.value), str(e.value)
    # a well-formed one with unsorted and duplicate columns is accepted (the kernels do not need sorted rows)
    rp, ci = np.array([0, 2, 4], np.int32), np.array([1, 0, 1, 1], np.int32)
    A = capi.CsrMatrix.from_host(2, 2, rp, ci, va)
    x = torch.tensor([1.0, 10.0], device=gpu); y = torch.zeros(2, device=gpu)
    for v in capi.VARIANTS.values():
        A.plan(v); A.run(v, x, y)
        torch.cuda.synchronize()
        assert y.tolist() == [11.0, 20.0]



def test_scalar_stays_bit_identical_on_long_rows(pkg, oracle, gpu):
    """Rows of ~1000-2000 nonzeros (the reference's own 50 %-dense regime) take the wavefront-per-row form of the
    SCALAR variant (k_scalar_long); its in-order readlane chain must reproduce the host loop bit for bit, ragged row
    ends and -0.0 / tiny values included."""
    W = pkg.workloads
    for (M, N, zero, seed) in [(2048, 300, 0.5, 5), (4096, 129, 0.25, 6), (1000, 64, 0.0, 7)]:
        A, x = W.dense_random(M, N, zero, seed=seed)
        A[::7, ::3] *= 1e-30
        A[5::11, 1::5] = -0.0
        rp, ci, va = oracle.csr_from_dense(A)
        assert len(ci) > 64 * N
        prob = DeviceProblem(pkg, gpu, N, M, rp, ci, va, x)
        y = prob.run(pkg.capi.SCALAR)
        y_seq = oracle.spmv(rp, ci, va, x)
        assert np.array_equal(y.view(np.uint32), y_seq.view(np.uint32)), (M, N, zero)
        assert np.array_equal(y, oracle.sgemv_dense(A, x))        # and to the dense SgemvCPU loop itself
        prob.A.close()


def test_stencil_matrix_all_variants(pkg, oracle, gpu):
    """A 7-point 3-D stencil (48^3): three column clusters 2*48^2 apart per row -- no contiguous window covers a
    chunk, the tiled plan stages nothing and runs the plain kernel; every variant against the oracle."""
    N, rp, ci, va = pkg.workloads.stencil7(48)
    x = np.random.Generator(np.random.PCG64(48)).uniform(-1, 1, size=N).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, N, N, rp, ci, va, x)
    _check_all_variants(pkg, oracle, prob, "stencil7(48)")
    prob.A.close()

def test_results_are_deterministic_run_to_run(pkg, oracle, gpu):
    w = pkg.workloads.config("c3", scale=1 / 32)
    prob = synth_problem(pkg, oracle, gpu, w)
    for name in ("adaptive", "tiled", "wave_pipe"):
        a = prob.run(pkg.capi.VARIANTS[name]); b = prob.run(pkg.capi.VARIANTS[name])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


# ---- the launcher surface, end to end ----------------------------------------------------------
def test_tester_executable_passes(pkg, gpu):
    env = dict(os.environ, SPMV_SEED="12345")
    p = subprocess.run([str(pkg.capi.TESTER_PATH), "1024", "768"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "========== OK ===========" in p.stdout
    for name in ("cublas", "wsp0", "wsp1", "asp2", "awsp0", "awsp1", "awsp2", "awsp_ref"):
        assert f"start to launch {name} kernel" in p.stdout       # tester.cpp:54-67 order and banner
    assert p.stdout.count(" took ") >= 13 and "[GPU kernel" not in p.stderr


@pytest.mark.parametrize("M,N,zero,seed", [(320, 448, 0.5, 9), (4096, 4096, 0.99, 4096)],
                         ids=["320x448_half", "config1_4096x4096_1pct"])
def test_launchers_from_python_match_oracle(pkg, oracle, gpu, M, N, zero, seed):
    """Call the C++ launchers (mangled names, dense host buffers) the way tester.cpp does.  The second case is
    BASELINE.json configs[0] at its stated size: 4096 x 4096 with 1 % uniform-random nonzeros (and the tester's
    50 %-zero x, tester.cpp:154), through all 13 launcher entries against the oracle."""
    import ctypes
    L = ctypes.CDLL(str(pkg.capi.LAUNCHERS_PATH))
    A, x = pkg.workloads.dense_random(M, N, zero, seed=seed)
    y_ref = oracle.sgemv_dense(A, x)
    rp, ci, va = oracle.csr_from_dense(A)
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    vp, ci_ = ctypes.c_void_p, ctypes.c_int
    calls = [("_Z15cublas_gemv_gpuiiPfS_S_", None), ("_Z12wsp_gemv_gpuiiPfS_S_i", 0), ("_Z12wsp_gemv_gpuiiPfS_S_i", 1),
             ("_Z12asp_gemv_gpuiiPfS_S_i", 2), ("_Z12asp_gemv_gpuiiPfS_S_i", 0), ("_Z12asp_gemv_gpuiiPfS_S_i", 1),
             ("_Z13awsp_gemv_gpuiiPfS_S_i", 0), ("_Z13awsp_gemv_gpuiiPfS_S_i", 1),
             ("_Z13awsp_gemv_gpuiiPfS_S_i", 2), ("_Z17awsp_ref_gemv_gpuiiPfS_S_", None),
             ("_Z18csr_naive_gemv_gpuiiPfS_S_", None), ("_Z19csr_tiling_gemv_gpuiiPfS_S_", None),
             ("_Z14naive_gemv_gpuiiPfS_S_", None), ("_Z15tiling_gemv_gpuiiPfS_S_", None),
             ("_Z15wsp_sm_gemv_gpuiiPfS_S_", None)]
    for sym, ver in calls:
        fn = getattr(L, sym)
        fn.restype = None
        y = np.full(N, np.nan, np.float32)
        if ver is None:
            fn.argtypes = [ci_, ci_, vp, vp, vp]
            fn(M, N, A.ctypes.data, x.ctypes.data, y.ctypes.data)
        else:
            fn.argtypes = [ci_, ci_, vp, vp, vp, ci_]
            fn(M, N, A.ctypes.data, x.ctypes.data, y.ctypes.data, ver)
        assert_close_to_oracle(y, y64, mag, sym)
        assert oracle.L.oracle_compare(N, y_ref.ctypes.data, y.ctypes.data, ctypes.c_float(1e-3)) == 0  # tester.cpp:75
        if "awsp_ref" in sym or "csr_naive" in sym or "naive_gemv" in sym:
            assert np.array_equal(y.view(np.uint32), y_ref.view(np.uint32)), sym


# ---- the reference's tiled bitmap-CSR (f-3) -------------------------------------------------------
def test_tcsr_device_builder_is_bit_exact(pkg, gpu, golden):
    """spmv_tcsr_from_dense_host vs the reference's TCSRMatrix arrays (tcsr.cpp:5-38)."""
    if golden.tcsr is None:
        with pytest.raises(pkg.capi.SpmvError) as e:       # not 32-aligned: refused, like the reference's assert
            pkg.capi.TcsrMatrix.from_dense_host(golden.A)
        assert e.value.status == pkg.capi.ERR_INVALID
        return
    t = pkg.capi.TcsrMatrix.from_dense_host(golden.A)
    bi, bm, va = t.download()
    rbi, rbm, rva = golden.tcsr
    assert np.array_equal(bi, rbi)
    assert np.array_equal(bm, rbm)
    assert np.array_equal(va.view(np.uint32), rva.view(np.uint32))


def test_tcsr_spmv_matches_oracle(pkg, oracle, gpu, golden):
    if golden.tcsr is None:
        pytest.skip("not 32-aligned")
    t = pkg.capi.TcsrMatrix.from_dense_host(golden.A)
    y = np.full(golden.N, np.nan, np.float32)
    ms = t.run_host(golden.x, y)
    assert ms > 0
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    assert_close_to_oracle(y, y64, mag, f"tcsr/{golden.name}")


@pytest.mark.parametrize("M,N,zero", [(4096, 4096, 0.5), (32, 32, 0.0), (2048, 96, 0.9), (64, 4096, 0.97), (320, 1024, 1.0)])
def test_tcsr_tester_regime_and_shapes(pkg, oracle, gpu, M, N, zero):
    """4096^2 at 50 % is the reference tester's own problem (test/main.cpp:4, tester.cpp:106)."""
    import torch
    A, x = pkg.workloads.dense_random(M, N, zero, seed=M + N)
    t = pkg.capi.TcsrMatrix.from_dense_device(torch.from_numpy(A).to(gpu))
    bi, bm, va = t.download()
    obi, obm, ova = oracle.tcsr_from_dense(A)
    assert np.array_equal(bi, obi) and np.array_equal(bm, obm) and np.array_equal(va.view(np.uint32), ova.view(np.uint32))
    dx = torch.from_numpy(x).to(gpu)
    dy = torch.full((N,), float("nan"), device=gpu)
    t.run(dx, dy)
    torch.cuda.synchronize()
    rp, ci, cv = oracle.csr_from_dense(A)
    y64, mag = oracle.spmv_f64(rp, ci, cv, x)
    assert_close_to_oracle(dy.cpu().numpy(), y64, mag, f"tcsr {M}x{N}")
