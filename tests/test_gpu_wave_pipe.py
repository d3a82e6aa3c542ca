"""GPU (-m gpu): SPMV_WAVE_PIPE (csrc/kernels_rows.hip: k_wave_bundle / k_wave_pieces / k_wave_combine), the slot of the
reference's pipelined wave kernel (src/kernels/wsp.cu:59-138).  Round 3 gave it a plan -- the rows of more than 512
nonzeros cut into pieces of 1024, and one window of x per block of 512 rows where the columns fit one -- so the cases here
are the ones that plan has to get right: rows on either side of the two thresholds, blocks with and without a window,
the first run of a handle that was never planned, Inf / NaN in x next to the slots past a piece's end."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from _util import DeviceProblem, assert_close_to_oracle, synth_problem

pytestmark = pytest.mark.gpu


def _describe(prob, capi):
    return dict(kv.split("=") for kv in prob.A.plan_describe(capi.WAVE_PIPE).split())


@pytest.mark.parametrize("name,band,scale", [("c3", 8192, 1 / 16), ("c3", 0, 1 / 16), ("c3", 65536, 1 / 16), ("c4", 8192, 1 / 64),
                                             ("c2", 8192, 1 / 4), ("c2", 0, 1 / 4)])
@pytest.mark.parametrize("block", [0, 512, 1024])
def test_wave_pipe_on_the_synthetic_laws(pkg, oracle, gpu, monkeypatch, name, band, scale, block):
    """Scaled-down configs: power-law rows (long rows, pieces), mixed and constant rows; banded columns (every block has a
    window), uniform and wide bands (none has: the kernel without windows).  Both workgroup sizes (the library takes 1024
    rows from 2 Mi rows on; SPMV_WAVE_BLOCK forces one at plan time)."""
    capi = pkg.capi
    if block:
        monkeypatch.setenv("SPMV_WAVE_BLOCK", str(block))
    else:
        monkeypatch.delenv("SPMV_WAVE_BLOCK", raising=False)
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    y = prob.run(capi.WAVE_PIPE)
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    assert_close_to_oracle(y, y64, mag, f"wave_pipe {name} band {band}")
    d = _describe(prob, capi)
    lengths = np.diff(prob.row_ptr)
    assert int(d["long_rows"]) == int((lengths > 512).sum())
    assert int(d["pieces"]) == int(((lengths[lengths > 512] + 1023) // 1024).sum())
    assert int(d["block_rows"]) == (block or (1024 if prob.rows >= (2 << 20) else 512))
    assert int(d["blocks"]) == (prob.rows + int(d["block_rows"]) - 1) // int(d["block_rows"])
    if band == 8192:
        assert int(d["blocks_with_x_window"]) == int(d["blocks"]), d
    if band == 0:
        assert int(d["blocks_with_x_window"]) == 0, d
    # the plan is a function of the matrix, and a handle that was never planned plans on its first run: same bits
    import torch
    B = capi.CsrMatrix.from_device(prob.rows, prob.cols, prob.d_rp, prob.d_ci, prob.d_va)
    assert "not planned" in B.plan_describe(capi.WAVE_PIPE)
    yb = torch.full((prob.rows,), float("nan"), device=gpu)
    B.run(capi.WAVE_PIPE, prob.d_x, yb)
    torch.cuda.synchronize()
    assert B.plan_describe(capi.WAVE_PIPE) == prob.A.plan_describe(capi.WAVE_PIPE)
    assert np.array_equal(y.view(np.uint32), yb.cpu().numpy().view(np.uint32))
    B.close(); prob.A.close()


row_run = st.one_of(
    st.tuples(st.just("const"), st.integers(0, 40), st.integers(1, 1500)),
    st.tuples(st.just("empty"), st.just(0), st.integers(1, 1500)),
    st.tuples(st.just("edge"), st.sampled_from([63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 2049]), st.integers(1, 3)),
    st.tuples(st.just("huge"), st.integers(3000, 70_000), st.integers(1, 2)),
    st.tuples(st.just("ragged"), st.integers(1, 600), st.integers(1, 300)),
)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(runs=st.lists(row_run, min_size=1, max_size=6), cols=st.sampled_from([1, 7, 4096, 8960, 8961, 9472, 9473, 70_001, 1 << 19]),
       local=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_wave_pipe_random_structures(pkg, oracle, gpu, runs, cols, local, seed):
    """Row lengths around the thresholds (a wavefront, a run's 512, a piece's 1024), runs of empty rows, rows that fill
    several pieces; columns either anywhere (blocks without a window) or within 4000 of the row (blocks with one, as long
    as no wide row is in them); unsorted, duplicates allowed (mean row length kept <= 32: above it the variant runs the
    plain wave-per-row kernel, which the parity tests cover)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = []
    for kind, length, count in runs:
        if kind == "ragged":
            lengths += list(rng.integers(0, length + 1, size=count))
        else:
            lengths += [length] * count
    lengths = np.asarray(lengths, np.int64)
    if lengths.sum() > 32 * len(lengths):                   # keep the bundle path: pad with empty rows
        lengths = np.concatenate([lengths, np.zeros(int(lengths.sum() // 32) + 1 - len(lengths), np.int64)])
        lengths = lengths[rng.permutation(len(lengths))]
    rows = len(lengths)
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    nnz = int(rp[-1])
    if local:
        rows_of = np.repeat(np.arange(rows), lengths)
        ci = ((rows_of % cols) + rng.integers(-4000, 4001, size=nnz)) % cols
    else:
        ci = rng.integers(0, cols, size=nnz)
    ci = ci.astype(np.int32)
    va = rng.uniform(-1, 1, size=nnz).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    import os
    os.environ["SPMV_WAVE_BLOCK"] = ["512", "1024"][seed % 2]
    try:
        prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
        y = prob.run(pkg.capi.WAVE_PIPE)
        ys = prob.run(pkg.capi.SCALAR)          # the same kernel with ordered sums: bit-identical to the sequential oracle
    finally:
        del os.environ["SPMV_WAVE_BLOCK"]
    assert not np.isnan(y).any(), "rows left unwritten"
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(y, y64, mag, "wave_pipe")
    assert np.array_equal(ys.view(np.uint32), oracle.spmv(rp, ci, va, x).view(np.uint32)), "scalar is not bit-identical"
    prob.A.close()


def test_wave_pipe_inf_nan_and_piece_ends(pkg, oracle, gpu):
    """Inf / NaN in x reach exactly the rows that touch them.  x[0] is Inf: the slots past the end of a run or a piece
    hold column 0 and value 0, and 0 * Inf must not reach a sum -- rows of 513 .. 1025 + 3 nonzeros leave such slots."""
    capi = pkg.capi
    rows, cols = 6000, 9000
    rng = np.random.default_rng(11)
    lengths = rng.integers(0, 20, size=rows)
    for r, L in ((5, 513), (700, 1024), (701, 1025), (1500, 1028), (3000, 4100), (5999, 600)):
        lengths[r] = L
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(np.arange(1, cols), size=int(L), replace=False)) for L in lengths]).astype(np.int32)
    va = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    x[0] = np.inf                       # no row references column 0
    x[4000] = np.nan
    x[8999] = -np.inf
    prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
    y = prob.run(capi.WAVE_PIPE)
    y_seq = oracle.spmv(rp, ci, va, x)
    assert np.array_equal(np.isnan(y), np.isnan(y_seq))
    assert np.array_equal(np.isinf(y), np.isinf(y_seq)) and np.array_equal(y[np.isinf(y)], y_seq[np.isinf(y_seq)])
    fin = np.isfinite(y_seq)
    y64, mag = oracle.spmv_f64(rp, ci, va, np.where(np.isfinite(x), x, 0).astype(np.float32))
    touched = ~fin
    assert_close_to_oracle(y[~touched], y64[~touched], mag[~touched], "wave_pipe beside Inf/NaN")
    prob.A.close()


def test_wave_pipe_edge_shapes(pkg, oracle, gpu):
    capi = pkg.capi
    cases = [
        (0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32)),                       # no rows
        (3, 5, np.zeros(4, np.int32), np.zeros(0, np.int32)),                       # no nonzeros
        (3, 5, np.array([0, 0, 1, 1], np.int32), np.array([4], np.int32)),          # one nonzero
        (200, 6000, np.concatenate([[0], np.full(200, 5000)]).astype(np.int32), np.arange(5000, dtype=np.int32)),   # one long row, then empty rows
        (513, 600, np.arange(514, dtype=np.int32), (np.arange(513) % 600).astype(np.int32)),                          # a second block of one row
    ]
    for rows, cols, rp, ci in cases:
        va = np.linspace(-1, 1, len(ci), dtype=np.float32) if len(ci) else np.zeros(0, np.float32)
        x = np.linspace(1, 2, cols, dtype=np.float32)
        prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
        y = prob.run(capi.WAVE_PIPE)
        if rows:
            assert not np.isnan(y).any()
            y64, mag = oracle.spmv_f64(rp, ci, va, x)
            assert_close_to_oracle(y, y64, mag, f"wave_pipe {rows}x{cols}")
        prob.A.close()


def test_wave_pipe_x_beyond_a_buffer_descriptor(pkg, oracle, gpu):
    """x of 2^30 + 7 entries (4 GiB): past what a 32-bit descriptor addresses, so the kernels take their plain-load
    instantiations (k_wave_bundle<false, ...>, k_wave_pieces<false>).  Rows with local columns (blocks with a window), rows
    with columns anywhere, rows on either side of the run and piece lengths."""
    import torch
    capi = pkg.capi
    cols = (1 << 30) + 7
    rng = np.random.default_rng(3)
    lengths = rng.integers(0, 24, size=3000)
    for r, L in ((10, 511), (11, 512), (700, 1025), (1900, 5000), (2999, 700)):
        lengths[r] = L
    rows = len(lengths)
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    nnz = int(rp[-1])
    rows_of = np.repeat(np.arange(rows), lengths)
    local = (cols - 9000) + (rows_of % 4000) + rng.integers(0, 4000, size=nnz)      # a window at the far end of x
    anywhere = rng.integers(0, cols, size=nnz)
    ci = np.where(rows_of < 1024, local, anywhere).astype(np.int32)                  # the first blocks local, the rest not
    va = rng.uniform(-1, 1, size=nnz).astype(np.float32)
    x = np.zeros(cols, np.float32)
    x[ci] = rng.uniform(-1, 1, size=nnz).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
    y = prob.run(capi.WAVE_PIPE)
    ys = prob.run(capi.SCALAR)
    assert not np.isnan(y).any()
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(y, y64, mag, "wave_pipe, 4 GiB x")
    assert np.array_equal(ys.view(np.uint32), oracle.spmv(rp, ci, va, x).view(np.uint32)), "scalar is not bit-identical"
    prob.A.close()
    del prob
    torch.cuda.empty_cache()
