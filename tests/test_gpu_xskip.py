"""GPU (-m gpu): SPMV_XSKIP -- activation sparsity on the CSR side: the input-major sweep that skips
the segments of the inputs whose x is zero (asp.cu:20-26, awsp.cu:127-134, awsp_ref.cu:52)."""
import numpy as np
import pytest

from _util import DeviceProblem, assert_close_to_oracle

pytestmark = pytest.mark.gpu


def test_xskip_on_golden_fixtures_with_the_testers_sparse_x(pkg, oracle, gpu, golden):
    """The fixtures' x vectors follow tester.cpp:151-167 (half of them 50 % zeros)."""
    prob = DeviceProblem(pkg, gpu, golden.N, golden.M, golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    y = prob.run(pkg.capi.XSKIP)
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    assert_close_to_oracle(y, y64, mag, f"xskip/{golden.name}")
    assert "segments=" in prob.A.plan_describe(pkg.capi.XSKIP)
    prob.A.close()


@pytest.mark.parametrize("M,N,azero,xzero", [(4096, 4096, 0.5, 0.5), (4096, 4096, 0.5, 0.9), (1000, 9001, 0.8, 0.5),
                                             (5000, 300, 0.3, 0.0), (64, 20000, 0.95, 1.0)])
def test_xskip_tester_regime_and_zero_fractions(pkg, oracle, gpu, M, N, azero, xzero):
    """4096^2 at 50 % with a 50 %-zero x is the reference tester's own problem; also 90 % zeros, no zeros, ALL zeros
    (y = 0 without reading a value), several output blocks (N > 1024) and a ragged last block."""
    rng = np.random.Generator(np.random.PCG64(M * 3 + N))
    A = rng.uniform(-1, 1, size=(M, N)).astype(np.float32)
    A[rng.random(size=(M, N)) < azero] = 0.0
    x = rng.uniform(-1, 1, size=M).astype(np.float32)
    x[rng.random(size=M) < xzero] = 0.0
    rp, ci, va = oracle.csr_from_dense(A)
    prob = DeviceProblem(pkg, gpu, N, M, rp, ci, va, x)
    y = prob.run(pkg.capi.XSKIP)
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(y, y64, mag, f"xskip {M}x{N} x_zero={xzero}")
    if xzero == 1.0:
        assert not y.any()
    # deterministic run to run, and -0.0 in x counts as zero like the reference's `x != 0` test
    y2 = prob.run(pkg.capi.XSKIP)
    assert np.array_equal(y.view(np.uint32), y2.view(np.uint32))
    prob.A.close()


def test_xskip_refuses_what_it_is_not_made_for(pkg, oracle, gpu):
    capi, W = pkg.capi, pkg.workloads
    # a large sparse matrix (config 2 itself): 1024 output blocks x 1Mi inputs = 2^30 table entries, beyond 2^27
    w = W.config("c2", band=0)
    rp = W.row_ptr(w)
    ci, va = oracle.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp)
    prob = DeviceProblem(pkg, gpu, w.rows, w.cols, rp, ci, va, oracle.synth_x(w.seed, 0, w.cols))
    with pytest.raises(capi.SpmvError) as e:
        prob.A.plan(capi.XSKIP)
    assert e.value.status == capi.ERR_INVALID and "dense-ish" in str(e.value)
    with pytest.raises(capi.SpmvError) as e:
        prob.A.run(capi.XSKIP, prob.d_x, prob.d_y)
    assert e.value.status == capi.ERR_NOT_PLANNED
    prob.A.close()
    # a row with the same column twice
    rp = np.array([0, 3, 4], np.int32); ci = np.array([1, 1, 2, 0], np.int32); va = np.ones(4, np.float32)
    prob = DeviceProblem(pkg, gpu, 2, 3, rp, ci, va, np.ones(3, np.float32))
    with pytest.raises(capi.SpmvError) as e:
        prob.A.plan(capi.XSKIP)
    assert e.value.status == capi.ERR_INVALID and "duplicate" in str(e.value)
    assert prob.run(capi.SCALAR).tolist() == [3.0, 1.0]      # the other variants take duplicates
    prob.A.close()


def test_xskip_plan_copies_the_values_and_plan_set_refreshes_them(pkg, oracle, gpu):
    import torch
    capi = pkg.capi
    A, x = pkg.workloads.dense_random(512, 640, 0.5, seed=5)
    rp, ci, va = oracle.csr_from_dense(A)
    prob = DeviceProblem(pkg, gpu, 640, 512, rp, ci, va, x)
    y0 = prob.run(capi.XSKIP)
    prob.d_va.mul_(2.0)
    assert np.array_equal(prob.run(capi.XSKIP), y0)                    # spmv_csr_plan is idempotent: the planned copy
    prob.A.plan_set(capi.XSKIP, prob.A.plan_params(capi.XSKIP))       # spmv_csr_plan_set rebuilds it
    assert np.array_equal(prob.run(capi.XSKIP), 2.0 * y0)
    prob.A.close()
