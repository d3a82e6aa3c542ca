"""GPU (-m gpu): plans are a pure function of the matrix, travel between handles, and SPMV_AUTO picks the
variant; the dense slots are asynchronous again (round-2 items of VERDICT.md)."""
import numpy as np
import pytest

from _util import DeviceProblem, assert_close_to_oracle, synth_problem

pytestmark = pytest.mark.gpu


def _shard(pkg, gpu, prob, r0, r1):
    """Handle over rows [r0, r1) of prob (device views of the same arrays, row_ptr rebased)."""
    import torch
    k0, k1 = int(prob.row_ptr[r0]), int(prob.row_ptr[r1])
    rp = torch.from_numpy((prob.row_ptr[r0:r1 + 1].astype(np.int64) - k0).astype(np.int32)).to(gpu)
    A = pkg.capi.CsrMatrix.from_device(r1 - r0, prob.cols, rp, prob.d_ci[k0:k1], prob.d_va[k0:k1])
    return A


@pytest.mark.parametrize("name,band", [("c4", 8192), ("c3", 8192), ("c4", 0)])
def test_two_plans_of_one_matrix_agree(pkg, oracle, gpu, name, band):
    """ADVICE: 'a test that two plans of the same matrix give the same plan_describe and bit-identical y'."""
    import torch
    capi = pkg.capi
    w = pkg.workloads.config(name, band=band, scale=1 / 16)
    prob = synth_problem(pkg, oracle, gpu, w)
    B = capi.CsrMatrix.from_device(prob.rows, prob.cols, prob.d_rp, prob.d_ci, prob.d_va)
    for v in (capi.TILED, capi.ADAPTIVE, capi.AUTO):
        ya = prob.run(v)
        B.plan(v)
        assert B.plan_describe(v) == prob.A.plan_describe(v)
        assert B.plan_params(v) == prob.A.plan_params(v)
        yb = torch.full((prob.rows,), float("nan"), device=gpu)
        B.run(v, prob.d_x, yb)
        torch.cuda.synchronize()
        assert np.array_equal(ya.view(np.uint32), yb.cpu().numpy().view(np.uint32)), capi.lib().spmv_variant_name(v)
    B.close(); prob.A.close()


def test_row_blocks_planned_alike_are_bit_identical_to_the_whole(pkg, oracle, gpu):
    """VERDICT next-7b: every rank reuses rank 0's (block, maxpass, col16); row blocks that start at a multiple of
    the chunk size (the bench's 65 536-row blocks hold exactly 2^20 nonzeros) then reproduce the single-handle y
    bit for bit, whatever block size each shard would have picked for itself."""
    import torch
    capi, W = pkg.capi, pkg.workloads
    w = W.config("c4", band=8192, scale=1 / 16)          # 1Mi rows, 16 blocks of 65 536 rows
    prob = synth_problem(pkg, oracle, gpu, w)
    for v in (capi.TILED, capi.ADAPTIVE):
        y_whole = prob.run(v)
        cuts = [0, 3 * W.BLOCK_ROWS, 4 * W.BLOCK_ROWS, w.rows]      # unequal blocks
        y = torch.full((w.rows,), float("nan"), device=gpu)
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            S = _shard(pkg, gpu, prob, r0, r1)
            S.plan_like(prob.A, v)
            assert S.plan_params(v) == prob.A.plan_params(v)
            S.run(v, prob.d_x, y[r0:r1])
            torch.cuda.synchronize()
            S.close()
        assert np.array_equal(y.cpu().numpy().view(np.uint32), y_whole.view(np.uint32))
    # a pick made elsewhere (e.g. by SPMV_AUTOTUNE=1 on rank 0) can be imposed too
    prob.A.plan_set(capi.TILED, [capi.TILED, 1024, 6, 1, 0, 0, 0, 0])
    assert "block=1024" in prob.A.plan_describe(capi.TILED) and "maxpass=6" in prob.A.plan_describe(capi.TILED)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    assert_close_to_oracle(prob.run(capi.TILED), y64, mag, "tiled 1024/6 imposed")
    with pytest.raises(capi.SpmvError):
        prob.A.plan_set(capi.TILED, [capi.TILED, 300, 6, 1, 0, 0, 0, 0])
    with pytest.raises(capi.SpmvError):
        prob.A.plan_set(capi.TILED, [capi.PANEL, 512, 4, 1, 0, 0, 0, 0])
    prob.A.close()


def test_auto_picks_tiled_on_a_band_and_panel_on_uniform_columns(pkg, oracle, gpu):
    """VERDICT next-4: 'the plan picks tiled on band 8192, panel on uniform, and matches the oracle on both'."""
    capi, W = pkg.capi, pkg.workloads
    for band, want in ((8192, "auto -> tiled"), (0, "auto -> panel")):
        w = W.config("c4", band=band, scale=1 / 8)      # 2Mi x 2Mi: x is 8 MiB, beyond one XCD's L2
        prob = synth_problem(pkg, oracle, gpu, w)
        y = prob.run(capi.AUTO)
        d = prob.A.plan_describe(capi.AUTO)
        assert d.startswith(want), d
        y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
        assert_close_to_oracle(y, y64, mag, d)
        assert prob.A.plan_params(capi.AUTO)[0] == (capi.TILED if band else capi.PANEL)
        prob.A.close()
    # a small x stays with the row-major kernel even when nothing can be staged
    w = W.config("c2", band=0, scale=1 / 4)             # 256Ki columns = 1 MiB of x
    prob = synth_problem(pkg, oracle, gpu, w)
    prob.run(capi.AUTO)
    assert prob.A.plan_describe(capi.AUTO).startswith("auto -> tiled")
    prob.A.close()


def test_auto_before_plan_is_refused(pkg, oracle, gpu):
    import torch
    capi = pkg.capi
    rp = np.array([0, 1, 2], np.int32); ci = np.array([0, 1], np.int32); va = np.ones(2, np.float32)
    prob = DeviceProblem(pkg, gpu, 2, 2, rp, ci, va, np.ones(2, np.float32))
    with pytest.raises(capi.SpmvError) as e:
        prob.A.run(capi.AUTO, prob.d_x, prob.d_y)
    assert e.value.status == capi.ERR_NOT_PLANNED
    with pytest.raises(capi.SpmvError):
        prob.A.plan_params(capi.AUTO)
    prob.A.close()


def test_handle_is_bound_to_its_device(pkg, gpu):
    """ADVICE (low): plan/run under another current device must be refused, not undefined."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: the refusal needs a second current device")
    capi = pkg.capi
    rp = torch.tensor([0, 1], dtype=torch.int32, device="cuda:0")
    ci = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    va = torch.ones(4, dtype=torch.float32, device="cuda:0")
    A = capi.CsrMatrix.from_device(1, 1, rp, ci[:1], va[:1])
    with torch.cuda.device(1):
        with pytest.raises(capi.SpmvError) as e:
            A.plan(capi.TILED)
        assert e.value.status == capi.ERR_INVALID
    A.close()


# ---- dense slots: asynchronous, workspace handed on in stream order (VERDICT next-6b) ---------------------
def test_dense_modes_back_to_back_on_one_stream(pkg, oracle, gpu):
    """Modes 2 and 3 back to back with different N on ONE stream, no host wait in between, against the oracle --
    the sequence that produced 64 wrong outputs with stream-ordered allocations in round 1 (commit a8cb87e).  The
    partials now live in a workspace that is handed from call to call in stream order."""
    import torch
    capi, W = pkg.capi, pkg.workloads
    shapes = [(1024, 768, 11), (512, 1280, 12), (1024, 768, 13), (96, 4096, 14), (2048, 320, 15)]
    probs = []
    for M, N, seed in shapes:
        A, x = W.dense_random(M, N, 0.5, seed=seed)
        probs.append((A, x, torch.from_numpy(A).to(gpu), torch.from_numpy(x).to(gpu),
                      torch.full((N,), float("nan"), device=gpu), torch.full((N,), float("nan"), device=gpu)))
    torch.cuda.synchronize()
    for rep in range(3):                                   # 30 launches queued without a host wait
        for A, x, dA, dx, y2, y3 in probs:
            capi.dense_gemv(dA, dx, y2, 2)
            capi.dense_gemv(dA, dx, y3, 3)
    torch.cuda.synchronize()
    for A, x, dA, dx, y2, y3 in probs:
        rp, ci, va = oracle.csr_from_dense(A)
        y64, mag = oracle.spmv_f64(rp, ci, va, x)
        assert_close_to_oracle(y2.cpu().numpy(), y64, mag, f"dense mode 2 {A.shape}")
        assert_close_to_oracle(y3.cpu().numpy(), y64, mag, f"dense mode 3 {A.shape}")


def test_dense_workspace_entry_and_cross_stream_handoff(pkg, oracle, gpu):
    import torch
    capi, W = pkg.capi, pkg.workloads
    A, x = W.dense_random(1024, 768, 0.5, seed=21)
    dA, dx = torch.from_numpy(A).to(gpu), torch.from_numpy(x).to(gpu)
    rp, ci, va = oracle.csr_from_dense(A)
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    need = capi.dense_gemv_workspace_bytes(768, 2)
    assert need == 64 * 768 * 4 and capi.dense_gemv_workspace_bytes(768, 0) == 0
    ws = torch.empty(need // 4, dtype=torch.float32, device=gpu)
    y = torch.full((768,), float("nan"), device=gpu)
    capi.dense_gemv(dA, dx, y, 3, workspace=ws)
    torch.cuda.synchronize()
    assert_close_to_oracle(y.cpu().numpy(), y64, mag, "dense mode 3, caller workspace")
    with pytest.raises(capi.SpmvError):                    # too small a workspace is refused, not overrun
        capi.dense_gemv(dA, dx, y, 2, workspace=ws[:100])
    # the library-owned buffer, alternating between two streams
    s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
    ya = torch.full((768,), float("nan"), device=gpu)
    yb = torch.full((768,), float("nan"), device=gpu)
    torch.cuda.synchronize()
    for _ in range(4):
        capi.dense_gemv(dA, dx, ya, 2, stream=s1)
        capi.dense_gemv(dA, dx, yb, 3, stream=s2)
    torch.cuda.synchronize()
    assert_close_to_oracle(ya.cpu().numpy(), y64, mag, "dense mode 2, stream 1")
    assert_close_to_oracle(yb.cpu().numpy(), y64, mag, "dense mode 3, stream 2")


# ---- sorted chunks: x gathered in column order instead of staged slice by slice -----------------------------------
@pytest.mark.parametrize("name,band,scale", [("c4", 65536, 1 / 16), ("c3", 8192, 1 / 8), ("c4", 8192, 1 / 32)])
@pytest.mark.parametrize("block", [256, 512, 1024])
def test_sorted_chunks_reproduce_the_staged_result_bit_for_bit(pkg, oracle, gpu, monkeypatch, name, band, scale, block):
    """The sorted body fetches the same x values and leaves the same products in the same LDS words as the staged
    bodies, so for one workgroup size y must not change by a single bit whether a chunk is staged (SPMV_SORTED_FROM=0),
    sorted from three passes on (the default) or always sorted (=1) -- and must match the oracle."""
    capi = pkg.capi
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    ys, descs = {}, {}
    for sf in ("0", "3", "1", "-1"):
        monkeypatch.setenv("SPMV_SORTED_FROM", sf)
        prob.A.plan_set(capi.TILED, [capi.TILED, block, 8, 1, 0, 0, 0, 0])
        ys[sf] = prob.run(capi.TILED)
        descs[sf] = prob.A.plan_describe(capi.TILED)
        assert_close_to_oracle(ys[sf], y64, mag, f"{w.name} sorted_from={sf}: {descs[sf]}")
    assert " sorted_chunks=0 " in descs["0"]
    n_chunks = int(descs["1"].split("chunks=")[1].split()[0])
    n_sorted = int(descs["1"].split("sorted_chunks=")[1].split()[0])
    # all full chunks but those whose span exceeds the 18-bit column field (the tail of a 65 536-long power-law row
    # next to short rows) or that found a block list
    assert n_sorted >= 0.95 * (n_chunks - 1 - int(descs["1"].split("block_list_chunks=")[1].split()[0])), descs["1"]
    assert np.array_equal(ys["0"].view(np.uint32), ys["1"].view(np.uint32))
    assert np.array_equal(ys["0"].view(np.uint32), ys["3"].view(np.uint32))
    assert np.array_equal(ys["0"].view(np.uint32), ys["-1"].view(np.uint32))      # the default: sorted where modelled cheaper
    prob.A.close()


# ---- the panel sweep with its x panels staged in LDS (small x) vs gathered through L2 --------------------------------
@pytest.mark.parametrize("name,scale,band", [("c2", 1 / 4, 0), ("c3", 1 / 16, 0), ("c4", 1 / 16, 4096), ("c2", 1.0, 0)])
def test_panel_in_lds_mode_matches_oracle_and_l2_mode(pkg, oracle, gpu, name, scale, band):
    """PANEL mode 2 (params[6]): every workgroup copies x through LDS 16 Ki columns at a time; mode 1: the round-1 sweep
    with L2-resident panels.  Both against the oracle; the library's own choice (mode 0) is the L2 sweep everywhere --
    the LDS form measured slower even at config 2 (DESIGN.md section 4 "Uniform columns") and stays as the experiment."""
    capi = pkg.capi
    w = pkg.workloads.config(name, band=band, scale=scale)
    prob = synth_problem(pkg, oracle, gpu, w)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    for mode, tag in ((2, "x_panels_in=LDS"), (1, "x_panels_in=L2")):
        prob.A.plan_set(capi.PANEL, [capi.PANEL, 0, 0, 0, 0, 0, mode, 0])
        d = prob.A.plan_describe(capi.PANEL)
        assert tag in d, d
        y = prob.run(capi.PANEL)
        assert_close_to_oracle(y, y64, mag, f"{w.name} panel {tag}")
        assert prob.A.plan_params(capi.PANEL)[6] == mode
    if name == "c2" and scale == 1.0:
        prob.A.plan_set(capi.PANEL, [capi.PANEL, 0, 0, 0, 0, 0, 0, 0])
        assert "x_panels_in=L2" in prob.A.plan_describe(capi.PANEL)
    prob.A.close()


def test_panel_lds_mode_edge_shapes(pkg, oracle, gpu):
    """Columns not a multiple of the 16 Ki panel, fewer rows than one wavefront block, empty rows, one long row, and a
    row block count that is not a multiple of 16 (idle wavefronts still stage and meet the barriers)."""
    capi = pkg.capi
    rng = np.random.Generator(np.random.PCG64(77))
    for rows, cols, lens in ((5, 100, [3, 0, 7, 0, 1]), (1000, 50_001, None), (7000, 16_384 * 3 + 5, None), (300, 40_000, "long")):
        if lens is None:
            L = rng.integers(0, 40, size=rows)
        elif lens == "long":
            L = rng.integers(0, 6, size=rows); L[17] = 30_000
        else:
            L = np.asarray(lens)
        L = np.minimum(L, cols)
        rp = np.concatenate([[0], np.cumsum(L)]).astype(np.int32)
        ci = np.concatenate([np.sort(rng.choice(cols, size=int(l), replace=False)) for l in L] + [np.zeros(0, np.int64)]).astype(np.int32)
        va = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
        x = rng.uniform(-1, 1, size=cols).astype(np.float32)
        prob = DeviceProblem(pkg, gpu, rows, cols, rp, ci, va, x)
        prob.A.plan_set(capi.PANEL, [capi.PANEL, 0, 0, 0, 0, 0, 2, 0])
        y = prob.run(capi.PANEL)
        y64, mag = oracle.spmv_f64(rp, ci, va, x)
        assert_close_to_oracle(y, y64, mag, f"panel LDS {rows}x{cols}")
        prob.A.close()


def test_clustered_columns_take_the_block_list_plan(pkg, oracle, gpu):
    """a 7-point 3-D stencil (columns r, r+-1, r+-n, r+-n^2: three clusters 2 n^2 apart, no
    contiguous window holds a chunk's span) must get the plan round 1's timed trials picked -- lists of 256-column
    blocks staged in one pass, by the smallest workgroup that holds them -- not the 1024-thread plan the prices of round 2
    chose (0.714 -> 0.638 of peak on 200^3).  Pins the choice; the numbers are in profiles/r03_stencil_plans.jsonl."""
    capi, W = pkg.capi, pkg.workloads
    N, rp, ci, va = W.stencil7(128)
    x = np.random.Generator(np.random.PCG64(128)).uniform(-1, 1, size=N).astype(np.float32)
    prob = DeviceProblem(pkg, gpu, N, N, rp, ci, va, x)
    for v in (capi.TILED, capi.AUTO):
        y = prob.run(v)
        d = prob.A.plan_describe(v)
        fields = dict(f.split("=") for f in d.split(": ")[-1].split() if "=" in f)
        assert fields["block"] == "256", d
        assert int(fields["block_list_chunks"]) >= 0.9 * int(fields["chunks"]), d
        assert int(fields["sorted_chunks"]) == 0, d
        if v == capi.AUTO:
            assert d.startswith("auto -> tiled"), d
        y64, mag = oracle.spmv_f64(rp, ci, va, x)
        assert_close_to_oracle(y, y64, mag, d)
    prob.A.close()
