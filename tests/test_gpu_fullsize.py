"""GPU (-m gpu): BASELINE.json's full sizes, checked through size-independent properties and a
host-regenerated row sample (the oracle cannot walk 256Mi nonzeros in test time)."""
import numpy as np
import pytest

from _util import RTOL

pytestmark = pytest.mark.gpu


# ---- configs 2 and 3 at FULL size: the oracle walks them in a few seconds, so every row is checked ----
@pytest.mark.parametrize("name,band", [("c2", 0), ("c2", 8192), ("c3", 0), ("c3", 8192)])
def test_config_2_and_3_full_size_every_row(pkg, oracle, gpu, name, band):
    """BASELINE configs[1] (1Mi^2, 16/row; the scalar thread-per-row kernel is its named variant) and
    configs[2] (4Mi^2 power-law, mean 32; wave-per-row is its named variant): all variants, all rows."""
    from _util import assert_close_to_oracle, synth_problem
    w = pkg.workloads.config(name, band=band)
    assert w.nnz == {"c2": 1 << 24, "c3": 1 << 27}[name]
    prob = synth_problem(pkg, oracle, gpu, w)          # also checks the device generator bit for bit
    y_seq = oracle.spmv(prob.row_ptr, prob.col_idx, prob.vals, prob.x, threads=8)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    for vname, v in pkg.capi.VARIANTS.items():
        y = prob.run(v)
        if vname == "scalar":
            assert np.array_equal(y.view(np.uint32), y_seq.view(np.uint32)), f"{w.name}: scalar not bit-identical"
        assert_close_to_oracle(y, y64, mag, f"{w.name}/{vname}")


# ---- config 4 at full size, and config 5's per-GPU shard at full size (BASELINE configs[3], configs[4]) ----
# c5 = (128Mi)^2 with 2Gi nonzeros cut into 8 row blocks of 16Mi rows; ONE block is what one MI355X holds:
# 16Mi rows x 128Mi columns, 2^28 nonzeros, the full 512 MiB x (2 818 572 292 bytes).  Blocks
# k = 0 and k = 7 (first and last rows of the global matrix: band clipping at both ends, global row ids up to 2^27).
# ("c4", 0, 8192) is bench.py's headline workload itself (it was only covered by bench.py's parity_sample);
# band 1 000 000 is where SPMV_AUTO takes the sorted blocks of the panel family (round 3).
BIG = [("c4", 0, 8192), ("c4", 0, 0), ("c4", 0, 1 << 16), ("c4", 0, 1_000_000), ("c5", 0, 8192), ("c5", 7, 8192), ("c5", 0, 0),
       ("c5", 7, 0)]


@pytest.fixture(scope="module", params=BIG, ids=[f"{n}-block{k}-band{b}" for n, k, b in BIG])
def c4(request, pkg, gpu):
    import torch
    W, capi = pkg.workloads, pkg.capi
    name, k, band = request.param
    if name == "c4":
        w = W.config("c4", band=band)
        n = w.rows
    else:
        w = W.c5(8, band=band)
        n = 16 << 20
        assert w.rows == w.cols == 134_217_728
    r0 = k * n
    rp = W.row_ptr(w, r0, n)
    nnz = int(rp[-1])
    assert nnz == 268_435_456 and n == 16_777_216
    if name == "c5":
        assert W.algorithmic_bytes(n, w.cols, nnz) == 2_818_572_292        # BASELINE.md section 4, c5 per-GPU shard
    d_rp = torch.from_numpy(rp).to(gpu)
    d_ci = torch.empty(nnz, dtype=torch.int32, device=gpu)
    d_va = torch.empty(nnz, dtype=torch.float32, device=gpu)
    d_x = torch.empty(w.cols, dtype=torch.float32, device=gpu)
    capi.synth_fill(w.seed, r0, n, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
    capi.synth_x(w.seed, 0, w.cols, d_x)
    A = capi.CsrMatrix.from_device(n, w.cols, d_rp, d_ci, d_va)
    Aabs = capi.CsrMatrix.from_device(n, w.cols, d_rp, d_ci, d_va.abs())
    for v in capi.VARIANTS.values():
        A.plan(v)
    Aabs.plan(capi.ADAPTIVE)
    mag = torch.empty(n, dtype=torch.float32, device=gpu)
    Aabs.run(capi.ADAPTIVE, d_x.abs(), mag)          # sum_k |val_k x_k| per row: the error scale
    torch.cuda.synchronize()
    Aabs.close()
    yield dict(w=w, r0=r0, n=n, rp=rp, d_rp=d_rp, d_ci=d_ci, d_va=d_va, d_x=d_x, A=A, mag=mag, name=name)
    A.close()
    del d_rp, d_ci, d_va, d_x, mag
    torch.cuda.empty_cache()


def _run(c4, pkg, variant, x=None):
    import torch
    y = torch.full((c4["n"],), float("nan"), device=c4["d_x"].device)
    c4["A"].run(variant, c4["d_x"] if x is None else x, y)
    torch.cuda.synchronize()
    return y


def test_row_sample_matches_oracle(c4, pkg, oracle):
    """1Mi rows (separated windows incl. the block's first and last rows) regenerated on the host from the GLOBAL
    row ids: device generator bit-exact, SCALAR bit-identical, every variant within 1e-5 * sum|terms|."""
    w, rp, R0, n_local = c4["w"], c4["rp"], c4["r0"], c4["n"]
    x = oracle.synth_x(w.seed, 0, w.cols)
    assert np.array_equal(c4["d_x"].cpu().numpy().view(np.uint32), x.view(np.uint32))
    ys = {n: _run(c4, pkg, v) for n, v in pkg.capi.VARIANTS.items()}
    n = 1 << 18
    for l0 in (0, 7_654_321, n_local - n, 12_000_000):
        l1 = l0 + n
        rps = (rp[l0:l1 + 1].astype(np.int64) - int(rp[l0])).astype(np.int32)
        ci, va = oracle.synth_fill(w.seed, R0 + l0, R0 + l1, w.rows, w.cols, w.band, rps)
        k0, k1 = int(rp[l0]), int(rp[l1])
        assert np.array_equal(c4["d_ci"][k0:k1].cpu().numpy(), ci)                 # indices bit-exact
        assert np.array_equal(c4["d_va"][k0:k1].cpu().numpy().view(np.uint32), va.view(np.uint32))
        if w.band:                                                                  # the band follows the GLOBAL diagonal
            half = max(4 * 3072 * 8, w.band)
            assert ci.min() >= max(0, R0 + l0 - half) and ci.max() <= min(w.cols - 1, R0 + l1 + half)
        y_seq = oracle.spmv(rps, ci, va, x)
        y64, mag = oracle.spmv_f64(rps, ci, va, x)
        for name, y in ys.items():
            got = y[l0:l1].cpu().numpy()
            if name == "scalar":
                assert np.array_equal(got.view(np.uint32), y_seq.view(np.uint32))
            err = np.abs(got.astype(np.float64) - y64)
            assert np.all(err <= RTOL * mag + 1e-37), (name, l0, err.max())


def test_all_variants_agree_on_every_row(c4, pkg):
    """Every variant vs SCALAR (bit-identical to the oracle wherever sampled) on all 16Mi rows."""
    import torch
    ref = _run(c4, pkg, pkg.capi.SCALAR)
    assert not torch.isnan(ref).any()
    for name, v in pkg.capi.VARIANTS.items():
        y = _run(c4, pkg, v)
        assert not torch.isnan(y).any(), f"{name} left rows unwritten"
        bad = ((y - ref).abs() > RTOL * c4["mag"] + 1e-37).sum().item()
        assert bad == 0, f"{name}: {bad} rows differ from scalar beyond {RTOL}*sum|terms|"


@pytest.mark.parametrize("mode", [4, 5])
def test_binned_layout_forced_at_full_size(c4, pkg, mode):
    """The binned layouts of the panel family (round 4) forced on the uniform-column fixtures -- config 4 (512 panels, 83
    nonzeros per tile: SPMV_AUTO takes mode 4, the sum launch fetches the tiles) and config 5's shards (4096 panels of 32768
    columns = the layout's limit, 14-28 nonzeros per tile: AUTO takes mode 5, the product launch stores in bin order) -- each
    mode on both, against SCALAR (bit-identical to the oracle wherever sampled) on every row, and again on the same handle
    (bit-identical: both launches are deterministic)."""
    import torch
    if c4["w"].band != 0:
        pytest.skip("uniform columns only (the banded fixtures' tiles are a few fat ones per bin: covered at small scale)")
    capi = pkg.capi
    ref = _run(c4, pkg, capi.SCALAR)
    B = capi.CsrMatrix.from_device(c4["n"], c4["w"].cols, c4["d_rp"], c4["d_ci"], c4["d_va"])
    B.plan_set(capi.PANEL, [capi.PANEL, 0, 0, 0, 0, 0, mode, 0])
    d = B.plan_describe(capi.PANEL)
    if mode == 4:
        assert d.startswith("binned bins=") and "flagged_tiles=0 " in d, d
    else:
        assert d.startswith("binned scattered_products bins=") and "flagged_bins=0 " in d, d
    y = torch.full((c4["n"],), float("nan"), device=c4["d_x"].device)
    B.run(capi.PANEL, c4["d_x"], y)
    torch.cuda.synchronize()
    assert not torch.isnan(y).any()
    bad = ((y - ref).abs() > RTOL * c4["mag"] + 1e-37).sum().item()
    assert bad == 0, f"binned: {bad} rows differ from scalar beyond {RTOL}*sum|terms|"
    y2 = torch.full((c4["n"],), float("nan"), device=c4["d_x"].device)
    B.run(capi.PANEL, c4["d_x"], y2)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    B.close()


def test_auto_choice_at_full_size(c4, pkg):
    """SPMV_AUTO at the BASELINE sizes: the LDS-tiled kernel on a band of 8192 columns; on uniform columns the binned layout
    of the panel family (round 4: two streaming launches), the flavour whose product launch stores in bin order -- config 4
    (128 nonzeros per 4096 rows x 32768 columns) and config 5's shard (16; 128Mi columns); config 3 too
    (tests/test_gpu_binned.py); the sorted blocks of the panel family on a band of 1M columns (65 536: whichever
    the models price lower)."""
    d = c4["A"].plan_describe(pkg.capi.AUTO)
    band = c4["w"].band
    if band == 0:       # config 4: 128 nonzeros per 4096 rows x 32768 columns; config 5's shard: 16
        assert d.startswith("auto -> panel: binned scattered_products bins="), d
    elif band <= 8192:
        assert d.startswith("auto -> tiled"), d
    elif band >= 1_000_000:
        assert d.startswith("auto -> panel: sorted_blocks="), d
    else:
        assert d.startswith("auto -> tiled") or d.startswith("auto -> panel: sorted_blocks="), d


def test_linearity(c4, pkg):
    """A(a*x1 + x2) == a*A(x1) + A(x2) within the fp32 bound -- no oracle needed at this size."""
    import torch
    dev = c4["d_x"].device
    g = torch.Generator(device=dev).manual_seed(7)
    x1 = c4["d_x"]
    x2 = torch.rand(c4["w"].cols, device=dev, generator=g) * 2 - 1
    a = 0.375
    v = pkg.capi.AUTO
    y1, y2, y12 = _run(c4, pkg, v, x1), _run(c4, pkg, v, x2), _run(c4, pkg, v, a * x1 + x2)
    mag = c4["mag"] * (1 + abs(a)) + 1.0    # |A||x2| is bounded by sum|val| <= mag-ish scale; generous but linear
    bad = ((y12 - (a * y1 + y2)).abs() > 4 * RTOL * mag).sum().item()
    assert bad == 0


def test_ones_vector_gives_row_sums(c4, pkg):
    """x = 1 turns SpMV into per-row sums of vals: compare with a segmented sum done by torch."""
    import torch
    dev = c4["d_x"].device
    ones = torch.ones(c4["w"].cols, device=dev)
    y = _run(c4, pkg, pkg.capi.ADAPTIVE, ones)
    csum = torch.cumsum(c4["d_va"].double(), 0)
    rp = c4["d_rp"].long()
    csum0 = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), csum])
    rowsum = csum0[rp[1:]] - csum0[rp[:-1]]
    lens = (rp[1:] - rp[:-1]).double()
    assert ((y.double() - rowsum).abs() <= 1e-5 * lens + 1e-6).all()
