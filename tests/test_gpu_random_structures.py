"""GPU (-m gpu): randomly generated CSR structures (hypothesis) through every variant vs the oracle.
Row-length patterns are drawn to hit chunk boundaries, empty-row runs, single huge rows and ragged
tails; columns are random sorted subsets, so windows are arbitrary."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from _util import DeviceProblem, assert_close_to_oracle

pytestmark = pytest.mark.gpu

row_run = st.one_of(
    st.tuples(st.just("const"), st.integers(0, 40), st.integers(1, 3000)),       # (kind, length, count)
    st.tuples(st.just("empty"), st.just(0), st.integers(1, 6000)),
    st.tuples(st.just("huge"), st.integers(3000, 70_000), st.integers(1, 2)),
    st.tuples(st.just("ragged"), st.integers(1, 600), st.integers(1, 400)),
    st.tuples(st.just("aligned"), st.sampled_from([4096, 8192, 16384, 2048]), st.integers(1, 3)),
)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(runs=st.lists(row_run, min_size=1, max_size=6), cols=st.sampled_from([1, 7, 4096, 70_001, 1 << 20]),
       seed=st.integers(0, 2**31 - 1))
def test_random_structures_all_variants(pkg, oracle, gpu, runs, cols, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = []
    for kind, length, count in runs:
        if kind == "ragged":
            lengths += list(rng.integers(0, length + 1, size=count))
        else:
            lengths += [length] * count
    lengths = np.minimum(np.asarray(lengths, np.int64), cols)      # a row cannot hold more than `cols` distinct columns
    if lengths.sum() > 3_000_000:
        lengths = lengths[: max(1, len(lengths) // 4)]
    rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    ci = np.empty(int(rp[-1]), np.int32)
    for r, L in enumerate(lengths):
        if L:
            if L * 4 > cols:
                ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(cols, size=int(L), replace=False))
            else:   # fast path for sparse rows: sorted unique sample by rejection
                c = np.unique(rng.integers(0, cols, size=int(L) * 2))
                while len(c) < L:
                    c = np.unique(np.concatenate([c, rng.integers(0, cols, size=int(L))]))
                ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(c, size=int(L), replace=False))
    va = rng.uniform(-1, 1, size=int(rp[-1])).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, len(lengths), cols, rp, ci, va, x)
    y_seq = oracle.spmv(rp, ci, va, x)
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    for name, v in pkg.capi.ALL_VARIANTS.items():      # xskip included: rows here are sorted and duplicate-free
        try:
            y = prob.run(v)
        except pkg.capi.SpmvError as e:
            if name == "xskip" and "dense-ish" in str(e):
                continue
            raise
        assert not np.isnan(y).any(), f"{name}: rows left unwritten"
        if name == "scalar":
            assert np.array_equal(y.view(np.uint32), y_seq.view(np.uint32))
        assert_close_to_oracle(y, y64, mag, name)
    prob.A.close()
