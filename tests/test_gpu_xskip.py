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
    # ... and a duplicate that is NOT adjacent (an unsorted row, which the handle accepts): two lanes of one segment
    # would read-add-write the same LDS word, so the plan refuses unsorted rows outright
    rp = np.array([0, 3, 4], np.int32); ci = np.array([3, 5, 3, 0], np.int32); va = np.ones(4, np.float32)
    prob = DeviceProblem(pkg, gpu, 2, 6, rp, ci, va, np.ones(6, np.float32))
    with pytest.raises(capi.SpmvError) as e:
        prob.A.plan(capi.XSKIP)
    assert e.value.status == capi.ERR_INVALID and "sorted" in str(e.value)
    assert prob.run(capi.SCALAR).tolist() == [3.0, 1.0]
    prob.A.close()
    # unsorted but duplicate-free is refused too (the check cannot tell it from the case above in one pass)
    rp = np.array([0, 3], np.int32); ci = np.array([4, 1, 2], np.int32); va = np.ones(3, np.float32)
    prob = DeviceProblem(pkg, gpu, 1, 6, rp, ci, va, np.ones(6, np.float32))
    with pytest.raises(capi.SpmvError) as e:
        prob.A.plan(capi.XSKIP)
    assert e.value.status == capi.ERR_INVALID and "sorted" in str(e.value)
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


@pytest.mark.parametrize("vname", ["xskip", "panel", "auto"])
def test_plans_that_copy_values_refuse_to_run_stale(pkg, oracle, gpu, vname):
    """a caller that rewrites vals on a borrowed handle says so (spmv_csr_values_changed);
    PANEL / XSKIP (and AUTO where it resolved to PANEL) then fail with SPMV_ERR_STALE_PLAN instead of multiplying
    with the values they copied, until the plan is rebuilt.  The variants that read vals live need nothing."""
    import torch
    capi = pkg.capi
    if vname == "xskip":
        A, x = pkg.workloads.dense_random(512, 640, 0.5, seed=9)
        rp, ci, va = oracle.csr_from_dense(A)
        prob = DeviceProblem(pkg, gpu, 640, 512, rp, ci, va, x)
    else:
        # uniform columns over 2Mi inputs (x = 8 MiB, beyond one L2): SPMV_AUTO resolves to the panel sweep
        w = pkg.workloads.Workload("stale", 1 << 16, 1 << 21, "mixed", 16, band=0)
        rp = pkg.workloads.row_ptr(w)
        ci, va = oracle.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp)
        prob = DeviceProblem(pkg, gpu, w.rows, w.cols, rp, ci, va, oracle.synth_x(w.seed, 0, w.cols))
    v = capi.ALL_VARIANTS[vname]
    y0 = prob.run(v)
    if vname == "auto":
        assert prob.A.plan_describe(v).startswith("auto -> panel")
    y_live0 = prob.run(capi.ADAPTIVE)
    prob.d_va.mul_(2.0)
    torch.cuda.synchronize()
    prob.A.values_changed()
    with pytest.raises(capi.SpmvError) as e:
        prob.A.run(v, prob.d_x, prob.d_y)
    assert e.value.status == capi.ERR_STALE_PLAN and "values_changed" in str(e.value)
    assert np.array_equal(prob.run(capi.ADAPTIVE), 2.0 * y_live0)          # live readers just see the new values
    assert np.array_equal(prob.run(v), 2.0 * y0)                            # run() plans first: the stale copy is rebuilt
    prob.A.close()


def test_check_values_env_catches_an_unannounced_write(pkg, gpu):
    """SPMV_CHECK_VALUES=1 (debug aid): the plan keeps a checksum of vals and every PANEL / XSKIP run recomputes it,
    so a write the caller did NOT announce is caught too.  The variable is read once per process: a child process."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        import __graft_entry__ as ge
        pkg = ge.load_package(); capi = pkg.capi
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(3)
        A = ((rng.random((256, 384)) < 0.5) * rng.standard_normal((256, 384))).astype(np.float32)
        h = capi.CsrMatrix.from_dense_host(A)
        rp, ci, va = h.download(); h.close()
        d = [torch.from_numpy(a).to(dev) for a in (rp, ci, va)]
        h = capi.CsrMatrix.from_device(384, 256, *d)
        x = torch.ones(256, device=dev); y = torch.empty(384, device=dev)
        for v in (capi.XSKIP, capi.PANEL):
            h.plan(v); h.run(v, x, y)
        d[2].mul_(3.0); torch.cuda.synchronize()
        bad = 0
        for v in (capi.XSKIP, capi.PANEL):
            try:
                h.run(v, x, y)
            except capi.SpmvError as e:
                bad += int(e.status == capi.ERR_STALE_PLAN and "SPMV_CHECK_VALUES" in str(e))
        h.run(capi.SCALAR, x, y)
        print("CAUGHT", bad)
    ''') % os.fspath(pkg.capi.PKG_DIR.parent)
    env = dict(os.environ, SPMV_CHECK_VALUES="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "CAUGHT 2" in out.stdout, out.stdout + out.stderr[-1000:]
