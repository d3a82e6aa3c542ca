"""GPU (-m gpu): SPMV_PANEL mode 4, "binned" (csrc/kernels_binned.hip): the products x[col] * val streamed in column-panel
order with the panel of x in LDS, then summed per bin of rows from wavefront-private LDS sums -- two streaming launches,
nothing gathered from memory.  Role of the reference's tiled format (src/tcsr.cpp:5-38, src/kernels/csr_tiling.cu:24-114)
at sparse scale, for its own structure law (uniform-random positions, src/tester.cpp:103-121).  Forced here through
spmv_csr_plan_set (params[6] = 4) so that every structure goes through it, including the ones SPMV_AUTO never sends there
(long rows: runs of equal rows folded by the segmented scan; banded columns: a few fat tiles per bin)."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from _util import DeviceProblem, assert_close_to_oracle, synth_problem

pytestmark = pytest.mark.gpu


def _params(capi, rows=0, mode=4):
    """params[6] = 4: binned, 5: binned with the products stored in bin order; params[4] = rows per bin (0 = the library's
    rule | 4096 | 8192)."""
    return [capi.PANEL, 0, 0, 0, rows, 0, mode, 0]


def _run(prob, capi, rows=0, wide=None, mode=4):
    """wide: None = the plan's rule (four products per lane where a tile holds a hundred or more, else one), True / False =
    forced (SPMV_BINNED_WIDE, read when the plan is made)."""
    import os
    import torch
    if wide is not None:
        os.environ["SPMV_BINNED_WIDE"] = "1" if wide else "0"
    try:
        prob.A.plan_set(capi.PANEL, _params(capi, rows, mode))
    finally:
        os.environ.pop("SPMV_BINNED_WIDE", None)
    prob.d_y.fill_(float("nan"))
    prob.A.run(capi.PANEL, prob.d_x, prob.d_y)
    torch.cuda.synchronize()
    return prob.d_y[:prob.rows].cpu().numpy()


@pytest.mark.parametrize("wide", [None, False, True])
@pytest.mark.parametrize("bin_rows", [0, 2048, 4096, 8192])
@pytest.mark.parametrize("name,band,scale", [("c2", 0, 1 / 4), ("c2", 8192, 1 / 8), ("c4", 0, 1 / 16), ("c4", 1000000, 1 / 16),
                                             ("c4", 8192, 1 / 64), ("c3", 0, 1 / 16), ("c3", 8192, 1 / 64)])
def test_binned_on_the_synthetic_laws(pkg, oracle, gpu, name, band, scale, bin_rows, wide):
    """Scaled-down configs under several column laws (the device generator checked against the host statement first), both
    bin sizes and the plan's own choice; a second handle planned with the reported numbers is bit-identical."""
    import torch
    capi = pkg.capi
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    y = _run(prob, capi, bin_rows, wide)
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    assert_close_to_oracle(y, y64, mag, f"binned {name} band {band}")
    if wide is not None:
        assert f"products_per_lane={4 if wide else 2} " in prob.A.plan_describe(capi.PANEL)
    d = prob.A.plan_describe(capi.PANEL)
    assert d.startswith("binned bins="), d
    got = prob.A.plan_params(capi.PANEL)
    assert got[6] == 4 and got[4] in (2048, 4096, 8192)
    if bin_rows:
        assert got[4] == bin_rows
    B = capi.CsrMatrix.from_device(prob.rows, prob.cols, prob.d_rp, prob.d_ci, prob.d_va)
    if wide is not None:
        os.environ["SPMV_BINNED_WIDE"] = "1" if wide else "0"
    try:
        B.plan_set(capi.PANEL, got)
    finally:
        os.environ.pop("SPMV_BINNED_WIDE", None)
    assert B.plan_describe(capi.PANEL) == d
    yb = torch.full((prob.rows,), float("nan"), device=gpu)
    B.run(capi.PANEL, prob.d_x, yb)
    torch.cuda.synchronize()
    assert np.array_equal(y.view(np.uint32), yb.cpu().numpy().view(np.uint32))
    # a second run of the same handle: the scratch products are rewritten, the answer is the same to the bit
    y2 = torch.full((prob.rows,), float("nan"), device=gpu)
    prob.A.run(capi.PANEL, prob.d_x, y2)
    torch.cuda.synchronize()
    assert np.array_equal(y.view(np.uint32), y2.cpu().numpy().view(np.uint32))
    B.close(); prob.A.close()


def test_binned_rows_without_repeats_in_a_panel_match_the_sequential_sum(pkg, oracle, gpu):
    """A row's terms reach its sum in ascending column order, one rounding per product and per add: where no row has two
    nonzeros inside one panel of 32768 columns the result is the host loop's, bit for bit (csr_naive.cu:13-22 order)."""
    rows, cols, per_row = 20000, 16 * 32768, 16
    rng = np.random.default_rng(11)
    # one column per panel and row: 16 panels, 16 nonzeros, ascending
    ci = (np.arange(per_row)[None, :] * 32768 + rng.integers(0, 32768, size=(rows, per_row))).astype(np.int32).ravel()
    rp = (np.arange(rows + 1) * per_row).astype(np.int32)
    va = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
    for wide in (False, True):
        y = _run(prob, pkg.capi, wide=wide)
        y_seq = oracle.spmv(rp, ci, va, x)
        assert np.array_equal(y.view(np.uint32), y_seq.view(np.uint32)), f"wide={wide}"
    prob.A.close()


def test_binned_on_golden_fixtures(pkg, oracle, gpu, golden):
    """The reference-built CSR arrays of the fixtures (dense-ish: one panel, every row a run of equal rows inside its
    tile; M != N, empty first/last rows, -0.0f and denormals)."""
    prob = DeviceProblem(pkg, gpu, golden.N, golden.M, golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    for wide in (False, True):
        y = _run(prob, pkg.capi, wide=wide)
        assert_close_to_oracle(y, y64, mag, f"binned/{golden.name}/wide={wide}")
    prob.A.close()


row_run = st.one_of(
    st.tuples(st.just("const"), st.integers(0, 40), st.integers(1, 3000)),
    st.tuples(st.just("empty"), st.just(0), st.integers(1, 6000)),
    st.tuples(st.just("huge"), st.integers(3000, 70_000), st.integers(1, 2)),
    st.tuples(st.just("ragged"), st.integers(1, 600), st.integers(1, 400)),
    st.tuples(st.just("block"), st.integers(1, 24), st.sampled_from([4095, 4096, 4097, 8191, 8192, 8193])),
)


@settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(runs=st.lists(row_run, min_size=1, max_size=5), cols=st.sampled_from([1, 7, 4096, 32767, 32768, 32769, 70_001, 1 << 19]),
       dups=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_binned_random_structures(pkg, oracle, gpu, runs, cols, dups, seed):
    """Row-length patterns that hit the bin cuts (4095 / 4096 / 4097 rows), runs of empty rows, single huge rows (runs of
    equal rows longer than a wavefront), column counts around the panel width, optionally unsorted rows with duplicate
    columns (the handle accepts them)."""
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
    y = _run(prob, pkg.capi, [0, 2048, 4096, 8192][seed % 4], [None, False, True][(seed >> 2) % 3])
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(y, y64, mag, "binned")
    prob.A.close()


def test_binned_edge_shapes_and_stale_values(pkg, oracle, gpu):
    import torch
    capi = pkg.capi
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
        y = _run(prob, capi)
        if rows:
            y64, mag = oracle.spmv_f64(rp, ci, va, x)
            assert_close_to_oracle(y, y64, mag, f"binned {rows}x{cols}")
        prob.A.close()
    # Inf / NaN in x reach exactly the rows that touch them (the pad slots of a panel multiply x[first column] by 0 into a
    # product nobody reads)
    rows, cols = 5000, 90000
    rng = np.random.default_rng(7)
    lengths = rng.integers(0, 20, size=rows)
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(cols, size=int(L), replace=False)) for L in lengths]).astype(np.int32)
    va = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    x[0] = np.inf; x[40000] = np.nan; x[89999] = -np.inf
    prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
    y = _run(prob, capi)
    y_seq = oracle.spmv(rp, ci, va, x)
    assert np.array_equal(np.isnan(y), np.isnan(y_seq))
    assert np.array_equal(np.isinf(y), np.isinf(y_seq)) and np.array_equal(y[np.isinf(y)], y_seq[np.isinf(y_seq)])
    # the plan holds a copy of the values: announced changes make it stale, spmv_csr_plan rebuilds it as it was planned
    prob.d_va.mul_(2.0)
    prob.A.values_changed()
    with pytest.raises(capi.SpmvError) as ei:
        prob.A.run(capi.PANEL, prob.d_x, prob.d_y)
    assert ei.value.status == capi.ERR_STALE_PLAN
    prob.A.plan(capi.PANEL)
    assert prob.A.plan_describe(capi.PANEL).startswith("binned bins=")
    x2 = np.where(np.isfinite(x), x, 0.5).astype(np.float32)
    prob.d_x.copy_(torch.from_numpy(x2))
    prob.A.run(capi.PANEL, prob.d_x, prob.d_y)
    torch.cuda.synchronize()
    y64, mag = oracle.spmv_f64(rp, ci, 2.0 * va, x2)
    assert_close_to_oracle(prob.d_y[:rows].cpu().numpy(), y64, mag, "binned after re-plan")
    prob.A.close()


def test_auto_resolves_to_binned_where_x_is_beyond_the_caches(pkg, oracle, gpu):
    """Uniform columns over 16 MiB of x (config 3's size): SPMV_AUTO takes the binned layout (its scattered flavour, a bin per
    resident wavefront) and its y passes the oracle."""
    import torch
    capi = pkg.capi
    w = pkg.workloads.config("c3", band=0)
    rp = pkg.workloads.row_ptr(w)
    d_rp = torch.from_numpy(rp).to(gpu)
    d_ci = torch.empty(int(rp[-1]), dtype=torch.int32, device=gpu)
    d_va = torch.empty(int(rp[-1]), dtype=torch.float32, device=gpu)
    d_x = torch.empty(w.cols, dtype=torch.float32, device=gpu)
    d_y = torch.full((w.rows,), float("nan"), device=gpu)
    capi.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
    capi.synth_x(w.seed, 0, w.cols, d_x)
    A = capi.CsrMatrix.from_device(w.rows, w.cols, d_rp, d_ci, d_va)
    A.plan(capi.AUTO)
    d = A.plan_describe(capi.AUTO)
    assert d.startswith("auto -> panel: binned scattered_products bins=1024 "), d
    A.run(capi.AUTO, d_x, d_y)
    torch.cuda.synchronize()
    # a window of rows regenerated on the host
    n = 1 << 17
    r0 = (w.rows // 2 // pkg.workloads.BLOCK_ROWS) * pkg.workloads.BLOCK_ROWS
    k0, k1 = int(rp[r0]), int(rp[r0 + n])
    rps = (rp[r0:r0 + n + 1].astype(np.int64) - k0).astype(np.int32)
    y64, mag = oracle.spmv_f64(rps, d_ci[k0:k1].cpu().numpy(), d_va[k0:k1].cpu().numpy(), d_x.cpu().numpy())
    assert_close_to_oracle(d_y[r0:r0 + n].cpu().numpy(), y64, mag, "auto -> binned, c3 uniform")
    A.close()


# ---- the scattered flavour (params[6] = 5): the product launch stores in bin order, the sum launch streams accumulator numbers
@pytest.mark.parametrize("bin_rows", [0, 4096, 8192, 16384])
@pytest.mark.parametrize("name,band,scale", [("c2", 0, 1 / 4), ("c2", 8192, 1 / 8), ("c4", 0, 1 / 16), ("c4", 1000000, 1 / 16),
                                             ("c4", 8192, 1 / 64), ("c3", 0, 1 / 16), ("c3", 8192, 1 / 64), ("c2", 0, 1.0)])
def test_scattered_on_the_synthetic_laws(pkg, oracle, gpu, name, band, scale, bin_rows):
    import torch
    capi = pkg.capi
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    y = _run(prob, capi, bin_rows, mode=5)
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    assert_close_to_oracle(y, y64, mag, f"binned, scattered products: {name} band {band}")
    d = prob.A.plan_describe(capi.PANEL)
    assert d.startswith("binned scattered_products bins="), d
    got = prob.A.plan_params(capi.PANEL)
    assert got[6] == 5 and got[4] in (4096, 8192, 16384)
    if bin_rows:
        assert got[4] == bin_rows
    B = capi.CsrMatrix.from_device(prob.rows, prob.cols, prob.d_rp, prob.d_ci, prob.d_va)
    B.plan_set(capi.PANEL, got)
    assert B.plan_describe(capi.PANEL) == d
    yb = torch.full((prob.rows,), float("nan"), device=gpu)
    B.run(capi.PANEL, prob.d_x, yb)
    torch.cuda.synchronize()
    assert np.array_equal(y.view(np.uint32), yb.cpu().numpy().view(np.uint32))
    y2 = torch.full((prob.rows,), float("nan"), device=gpu)
    prob.A.run(capi.PANEL, prob.d_x, y2)
    torch.cuda.synchronize()
    assert np.array_equal(y.view(np.uint32), y2.cpu().numpy().view(np.uint32))
    B.close(); prob.A.close()


def test_scattered_on_golden_fixtures(pkg, oracle, gpu, golden):
    """Dense-ish: one panel, every row many times in every step -- spare accumulators for every row (gedge: 1023 of the 1024 a bin
    has, and a last lane whose second entry does not exist: the shuffle that asks it must run on every lane), or the flagged
    bin's atomics (the others)."""
    prob = DeviceProblem(pkg, gpu, golden.N, golden.M, golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    for rows in (4096, 8192, 16384):
        y = _run(prob, pkg.capi, rows, mode=5)
        assert_close_to_oracle(y, y64, mag, f"binned scattered/{golden.name}/{rows}")
    prob.A.close()


@settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(runs=st.lists(row_run, min_size=1, max_size=5), cols=st.sampled_from([1, 7, 4096, 32767, 32768, 32769, 70_001, 1 << 19]),
       dups=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_scattered_random_structures(pkg, oracle, gpu, runs, cols, dups, seed):
    """The structures of test_binned_random_structures: single huge rows fill whole steps with one row (63 spare accumulators),
    many medium rows overflow the spare accumulators of a bin (flagged: LDS atomics)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = []
    for kind, length, count in runs:
        if kind == "ragged":
            lengths += list(rng.integers(0, length + 1, size=count))
        else:
            lengths += [length] * count
    lengths = np.asarray(lengths, np.int64)
    if lengths.sum() > 2_000_000:
        lengths = lengths[: max(1, len(lengths) // 4)]
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = rng.integers(0, cols, size=int(rp[-1])).astype(np.int32)      # unsorted, duplicates possible
    if not dups:
        for r in range(len(lengths)):
            ci[rp[r]:rp[r + 1]].sort()
    va = rng.uniform(-1, 1, size=int(rp[-1])).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, len(lengths), cols, rp, ci, va, x)
    y = _run(prob, pkg.capi, [0, 4096, 8192, 16384][seed % 4], mode=5)
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(y, y64, mag, "binned, scattered products")
    prob.A.close()


def test_scattered_edge_shapes_inf_nan_and_stale_values(pkg, oracle, gpu):
    import torch
    capi = pkg.capi
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
        y = _run(prob, capi, mode=5)
        if rows:
            y64, mag = oracle.spmv_f64(rp, ci, va, x)
            assert_close_to_oracle(y, y64, mag, f"binned scattered {rows}x{cols}")
        prob.A.close()
    rows, cols = 5000, 90000
    rng = np.random.default_rng(7)
    lengths = rng.integers(0, 20, size=rows)
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(cols, size=int(L), replace=False)) for L in lengths]).astype(np.int32)
    va = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    x[0] = np.inf; x[40000] = np.nan; x[89999] = -np.inf
    prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
    y = _run(prob, capi, mode=5)
    y_seq = oracle.spmv(rp, ci, va, x)
    assert np.array_equal(np.isnan(y), np.isnan(y_seq))
    assert np.array_equal(np.isinf(y), np.isinf(y_seq)) and np.array_equal(y[np.isinf(y)], y_seq[np.isinf(y_seq)])
    prob.d_va.mul_(2.0)
    prob.A.values_changed()
    with pytest.raises(capi.SpmvError) as ei:
        prob.A.run(capi.PANEL, prob.d_x, prob.d_y)
    assert ei.value.status == capi.ERR_STALE_PLAN
    prob.A.plan(capi.PANEL)
    assert prob.A.plan_describe(capi.PANEL).startswith("binned scattered_products bins=")
    x2 = np.where(np.isfinite(x), x, 0.5).astype(np.float32)
    prob.d_x.copy_(torch.from_numpy(x2))
    prob.A.run(capi.PANEL, prob.d_x, prob.d_y)
    torch.cuda.synchronize()
    y64, mag = oracle.spmv_f64(rp, ci, 2.0 * va, x2)
    assert_close_to_oracle(prob.d_y[:rows].cpu().numpy(), y64, mag, "binned scattered after re-plan")
    prob.A.close()


def test_scattered_one_pass_fill_gives_the_same_product(pkg, oracle, gpu):
    """SPMV_BS_FILL=1: the plan's one-pass fill (rows ascending inside a tile) instead of the two passes (groups of 64 panels, then
    tiles; the order inside a tile is what the LDS atomics hand out) -- another order of the same sums: both within the bound,
    each one bit-identical to itself on a second handle (the synthetic-law test checks that for the default)."""
    import torch
    capi = pkg.capi
    w = pkg.workloads.config("c4", band=0, scale=1 / 16)          # 32 panels: one group; and c2 at full size: 32 panels too
    for wl in (w, pkg.workloads.config("c3", band=0, scale=1 / 4), pkg.workloads.Workload("wide", 1 << 16, 1 << 23, "const", 16, band=0)):
        prob = synth_problem(pkg, oracle, gpu, wl)
        y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
        ys = []
        for fill in ("0", "1"):
            os.environ["SPMV_BS_FILL"] = fill
            try:
                y = _run(prob, capi, 8192, mode=5)
            finally:
                os.environ.pop("SPMV_BS_FILL", None)
            assert_close_to_oracle(y, y64, mag, f"scattered, SPMV_BS_FILL={fill}: {wl.name}")
            ys.append(y)
        prob.A.close()
