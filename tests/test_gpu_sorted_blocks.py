"""GPU (-m gpu): SPMV_PANEL mode 3, "sorted blocks" (csrc/kernels_colsort.hip): row blocks of 4096 rows streamed in
column order, groups of 64 nonzeros with distinct rows, the block's sums in wavefront-private LDS copies.
Role of the reference's tiled format (src/tcsr.cpp:5-38, src/kernels/csr_tiling.cu:24-114) at sparse scale, like the
panel sweep.  Forced here through spmv_csr_plan_set (params[6] = 3) so that every structure goes through it, including
the ones SPMV_AUTO would never send there (long rows: the flagged tail; dense-ish blocks: all flagged)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from _util import DeviceProblem, assert_close_to_oracle, synth_problem

pytestmark = pytest.mark.gpu


def _sorted_params(capi, rows=0, waves=0):
    """params[6] = 3: sorted blocks; params[4] = rows per block (0 = the library's rule | 4096 | 8192), params[5] =
    wavefronts per workgroup (0 | 4 | 8)."""
    return [capi.PANEL, 0, 0, 0, rows, waves, 3, 0]


def _run_sorted(prob, capi, rows=0, waves=0):
    import torch
    prob.A.plan_set(capi.PANEL, _sorted_params(capi, rows, waves))
    prob.d_y.fill_(float("nan"))
    prob.A.run(capi.PANEL, prob.d_x, prob.d_y)
    torch.cuda.synchronize()
    return prob.d_y[:prob.rows].cpu().numpy()


@pytest.mark.parametrize("geometry", [(0, 0), (4096, 8), (4096, 4), (8192, 4)])
@pytest.mark.parametrize("name,band,scale", [("c2", 0, 1 / 4), ("c2", 8192, 1 / 8), ("c4", 200000, 1 / 64), ("c4", 1000000, 1 / 16),
                                             ("c4", 8192, 1 / 64), ("c3", 8192, 1 / 64), ("c4", 0, 1 / 64)])
def test_sorted_blocks_on_the_synthetic_laws(pkg, oracle, gpu, name, band, scale, geometry):
    """Scaled-down configs under several column laws, the device generator checked against the host statement first;
    the three block geometries the plan chooses from, and the plan's own choice."""
    capi = pkg.capi
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    y = _run_sorted(prob, capi, *geometry)
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    assert_close_to_oracle(y, y64, mag, f"sorted blocks {name} band {band}")
    d = prob.A.plan_describe(capi.PANEL)
    assert d.startswith("sorted_blocks="), d
    got = prob.A.plan_params(capi.PANEL)
    assert got[6] == 3 and got[4] in (4096, 8192) and got[5] in (4, 8)
    if geometry != (0, 0):
        assert (got[4], got[5]) == geometry
    # the plan is a pure function of the matrix: a second handle gives the same description and bit-identical y
    import torch
    B = capi.CsrMatrix.from_device(prob.rows, prob.cols, prob.d_rp, prob.d_ci, prob.d_va)
    B.plan_set(capi.PANEL, got)                      # ... planned with the numbers the first one reports
    assert B.plan_describe(capi.PANEL) == d
    yb = torch.full((prob.rows,), float("nan"), device=gpu)
    B.run(capi.PANEL, prob.d_x, yb)
    torch.cuda.synchronize()
    assert np.array_equal(y.view(np.uint32), yb.cpu().numpy().view(np.uint32))
    B.close(); prob.A.close()


def test_sorted_blocks_on_golden_fixtures(pkg, oracle, gpu, golden):
    """The reference-built CSR arrays of the fixtures (dense-ish: most of it lands in the flagged tails, M != N, empty
    first/last rows, -0.0f and denormals)."""
    prob = DeviceProblem(pkg, gpu, golden.N, golden.M, golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    y = _run_sorted(prob, pkg.capi)
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    assert_close_to_oracle(y, y64, mag, f"sorted blocks/{golden.name}")
    prob.A.close()


row_run = st.one_of(
    st.tuples(st.just("const"), st.integers(0, 40), st.integers(1, 3000)),
    st.tuples(st.just("empty"), st.just(0), st.integers(1, 6000)),
    st.tuples(st.just("huge"), st.integers(3000, 70_000), st.integers(1, 2)),
    st.tuples(st.just("ragged"), st.integers(1, 600), st.integers(1, 400)),
    st.tuples(st.just("block"), st.integers(1, 24), st.sampled_from([4095, 4096, 4097, 8191, 8192, 8193])),
)


@settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(runs=st.lists(row_run, min_size=1, max_size=5), cols=st.sampled_from([1, 7, 4096, 70_001, 1 << 18, 1 << 19]),
       dups=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_sorted_blocks_random_structures(pkg, oracle, gpu, runs, cols, dups, seed):
    """Row-length patterns that hit the block cuts (4095 / 4096 / 4097 rows), runs of empty rows, single huge rows (the
    flagged tail), optionally unsorted rows with duplicate columns (the handle accepts them)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = []
    for kind, length, count in runs:
        if kind == "ragged":
            lengths += list(rng.integers(0, length + 1, size=count))
        else:
            lengths += [length] * count
    lengths = np.asarray(lengths, np.int64)
    if not dups:
        lengths = np.minimum(lengths, cols)
    if lengths.sum() > 2_000_000:
        lengths = lengths[: max(1, len(lengths) // 4)]
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = rng.integers(0, cols, size=int(rp[-1])).astype(np.int32)      # unsorted, duplicates possible
    if not dups:
        for r, L in enumerate(lengths):
            if L:
                if L * 4 > cols:
                    ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(cols, size=int(L), replace=False))
                else:
                    c = np.unique(rng.integers(0, cols, size=int(L) * 2))
                    while len(c) < L:
                        c = np.unique(np.concatenate([c, rng.integers(0, cols, size=int(L))]))
                    ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(c, size=int(L), replace=False))
    va = rng.uniform(-1, 1, size=int(rp[-1])).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, len(lengths), cols, rp, ci, va, x)
    capi = pkg.capi
    geometry = [(0, 0), (4096, 8), (4096, 4), (8192, 4)][seed % 4]
    try:
        y = _run_sorted(prob, capi, *geometry)
    except capi.SpmvError as e:
        # the one structure the layout refuses: 256 neighbours in column order spread over 2^19 (2^18) columns or more
        assert e.status == capi.ERR_INVALID and ("2^19" in str(e) or "2^18" in str(e)), str(e)
        assert cols >= (1 << 18)
        prob.A.close()
        return
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(y, y64, mag, "sorted blocks")
    prob.A.close()


def test_sorted_blocks_edge_shapes(pkg, oracle, gpu):
    capi = pkg.capi
    # no rows / no nonzeros / one nonzero / one row of 5000 nonzeros
    cases = [
        (0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32)),
        (3, 5, np.zeros(4, np.int32), np.zeros(0, np.int32)),
        (3, 5, np.array([0, 0, 1, 1], np.int32), np.array([4], np.int32)),
        (1, 6000, np.array([0, 5000], np.int32), np.arange(5000, dtype=np.int32)),
    ]
    for rows, cols, rp, ci in cases:
        va = np.linspace(-1, 1, len(ci), dtype=np.float32) if len(ci) else np.zeros(0, np.float32)
        x = np.linspace(1, 2, cols, dtype=np.float32)
        prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
        y = _run_sorted(prob, capi)
        if rows:
            y64, mag = oracle.spmv_f64(rp, ci, va, x)
            assert_close_to_oracle(y, y64, mag, f"sorted blocks {rows}x{cols}")
        prob.A.close()
    # Inf / NaN in x reach exactly the rows that touch them (the empty slots multiply x[first column of their unit] by 0
    # into a dummy row, never into an output)
    rows, cols = 5000, 9000
    rng = np.random.default_rng(7)
    lengths = rng.integers(0, 20, size=rows)
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(cols, size=int(L), replace=False)) for L in lengths]).astype(np.int32)
    va = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    x[0] = np.inf; x[4000] = np.nan; x[8999] = -np.inf
    prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
    y = _run_sorted(prob, capi)
    y_seq = oracle.spmv(rp, ci, va, x)
    assert np.array_equal(np.isnan(y), np.isnan(y_seq))
    assert np.array_equal(np.isinf(y), np.isinf(y_seq)) and np.array_equal(y[np.isinf(y)], y_seq[np.isinf(y_seq)])
    prob.A.close()
