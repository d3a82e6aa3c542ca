"""GPU (-m gpu): properties of the C ABI beyond plain parity -- graph capture, the N>1 exchange
with real kernels (two gloo ranks sharing the one GPU of the test box), plan selection."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from _util import DeviceProblem, assert_close_to_oracle, synth_problem

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_run_is_graph_capturable(pkg, oracle, gpu):
    """include/spmv_hip.h promises spmv_csr_run allocates nothing and never synchronises."""
    import torch
    w = pkg.workloads.config("c4", band=4096, scale=1 / 64)
    prob = synth_problem(pkg, oracle, gpu, w)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    for name in ("adaptive", "tiled", "vector", "wave_pipe", "scalar"):
        v = pkg.capi.VARIANTS[name]
        prob.A.plan(v)
        prob.A.run(v, prob.d_x, prob.d_y)            # warm (module load, attribute set) outside capture
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                prob.A.run(v, prob.d_x, prob.d_y, stream=s)
        prob.d_y.fill_(float("nan"))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert_close_to_oracle(prob.d_y[:w.rows].cpu().numpy(), y64, mag, f"graph/{name}")
    # the panel family too: the sweep, the sorted blocks and the binned layout (two launches + the scratch products) through
    # spmv_csr_plan_set, and SPMV_WAVE on short rows (the bundles and the long rows' pieces: three launches)
    for name, params in (("panel sweep", [pkg.capi.PANEL, 0, 0, 0, 0, 0, 1, 0]), ("sorted blocks", [pkg.capi.PANEL, 0, 0, 0, 0, 0, 3, 0]),
                         ("binned", [pkg.capi.PANEL, 0, 0, 0, 0, 0, 4, 0]), ("binned, scattered products", [pkg.capi.PANEL, 0, 0, 0, 0, 0, 5, 0]),
                         ("wave", None)):
        v = pkg.capi.PANEL if params else pkg.capi.WAVE
        if params:
            prob.A.plan_set(v, params)
        else:
            prob.A.plan(v)
        prob.A.run(v, prob.d_x, prob.d_y)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                prob.A.run(v, prob.d_x, prob.d_y, stream=s)
        prob.d_y.fill_(float("nan"))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert_close_to_oracle(prob.d_y[:w.rows].cpu().numpy(), y64, mag, f"graph/{name}")


def test_tiled_plan_picks_workgroup_size_from_the_data(pkg, oracle, gpu):
    """Narrow band -> 256-thread workgroups, wider band -> larger ones; results stay within bound."""
    import torch
    W = pkg.workloads
    for band in (1024, 6000, 14000, 100_000):
        w = W.Workload("t", 4 * W.BLOCK_ROWS, 4 * W.BLOCK_ROWS, "const", 16, band=band)
        prob = synth_problem(pkg, oracle, gpu, w)
        y = prob.run(pkg.capi.TILED)
        y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
        assert_close_to_oracle(y, y64, mag, f"band {band}")



@pytest.mark.parametrize("bits,wgs", [("8", "2"), ("11", "6"), ("17", "")])
def test_panel_variant_small_panels_many_launches(pkg, oracle, gpu, monkeypatch, bits, wgs):
    """SPMV_PANEL with 256-column panels (thousands of tiles, instructions that straddle tiles, empty tiles),
    one workgroup per launch (every launch boundary), and the shipped geometry; giant rows take the run-sum
    path (>= 5 nonzeros of one row in one tile).  Results are reproducible bit for bit."""
    monkeypatch.setenv("SPMV_PANEL_BITS", bits)
    if wgs:
        monkeypatch.setenv("SPMV_PANEL_WAVES", wgs)
    W = pkg.workloads
    cases = [W.config("c3", scale=1 / 32), W.config("c4", scale=1 / 64), W.config("c4", band=4096, scale=1 / 64),
             W.Workload("tall", 3 * W.BLOCK_ROWS, 3 * W.BLOCK_ROWS, "const", 16, band=0)]
    for w in cases:
        prob = synth_problem(pkg, oracle, gpu, w)
        y = prob.run(pkg.capi.PANEL)
        y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
        assert not np.isnan(y).any(), w.name
        assert_close_to_oracle(y, y64, mag, f"panel bits={bits} wgs={wgs or 'resident'} {w.name}")
        assert "panels=" in prob.A.plan_describe(pkg.capi.PANEL)
        assert np.array_equal(prob.run(pkg.capi.PANEL).view(np.uint32), y.view(np.uint32))
        prob.A.close()


def test_panel_plan_copies_the_values(pkg, oracle, gpu):
    """include/spmv_hip.h: the PANEL plan re-orders and COPIES vals; the other variants read them live."""
    import torch
    w = pkg.workloads.config("c4", scale=1 / 64)
    prob = synth_problem(pkg, oracle, gpu, w)
    y0 = prob.run(pkg.capi.PANEL)
    prob.d_va.mul_(2.0)
    prob.A.run(pkg.capi.PANEL, prob.d_x, prob.d_y)
    torch.cuda.synchronize()
    assert np.array_equal(prob.d_y[:w.rows].cpu().numpy(), y0)                 # still the planned values
    assert np.array_equal(prob.run(pkg.capi.PANEL), y0)                        # spmv_csr_plan is idempotent: still the copy
    prob.A.plan_set(pkg.capi.PANEL, prob.A.plan_params(pkg.capi.PANEL))       # spmv_csr_plan_set always re-plans
    y1 = prob.run(pkg.capi.PANEL)
    assert np.array_equal(y1, 2.0 * y0)
    assert np.array_equal(prob.run(pkg.capi.TILED).view(np.uint32), prob.run(pkg.capi.TILED).view(np.uint32))


def test_panel_refuses_more_columns_than_its_format_holds(pkg, gpu):
    """packed[] keeps 17 bits of column-in-panel and the plan at most 4096 panels: 2^29 columns per handle."""
    import torch
    capi = pkg.capi
    rp = torch.tensor([0, 1, 2], dtype=torch.int32, device=gpu)
    ci = torch.tensor([5, (1 << 30) - 1, 0, 0], dtype=torch.int32, device=gpu)[:2]
    va = torch.ones(4, dtype=torch.float32, device=gpu)[:2]
    A = capi.CsrMatrix.from_device(2, 1 << 30, rp, ci, va)
    with pytest.raises(capi.SpmvError) as e:
        A.plan(capi.PANEL)
    assert e.value.status == capi.ERR_INVALID and "panels" in str(e.value)
    A.plan(capi.TILED)            # the other variants take it
    A.plan(capi.AUTO)             # ... and SPMV_AUTO falls back to TILED where the panel layout cannot hold the matrix
    assert A.plan_describe(capi.AUTO).startswith("auto -> tiled")
    x = torch.zeros(1 << 30, dtype=torch.float32, device=gpu)
    x[5] = 2.0; x[(1 << 30) - 1] = 3.0
    y = torch.full((2,), float("nan"), device=gpu)
    A.run(capi.AUTO, x, y)
    torch.cuda.synchronize()
    assert y.tolist() == [2.0, 3.0]
    A.close()


def test_block_list_staging_on_a_stencil(pkg, oracle, gpu, monkeypatch):
    """Columns in three clusters 2*96^2 apart (7-point stencil on 96^3): no contiguous window is staged in one pass,
    the plan switches those chunks to a LIST of 256-column blocks (16-bit indices into the staged blocks) and the
    result still matches; with the 16-bit copy refused the same chunks fall back cleanly."""
    N, rp, ci, va = pkg.workloads.stencil7(96)
    x = np.random.Generator(np.random.PCG64(96)).uniform(-1, 1, size=N).astype(np.float32)
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    from _util import DeviceProblem
    for blk, passes in (("512", "8"), ("1024", "6"), ("256", "2")):
        monkeypatch.setenv("SPMV_TILED_BLOCK", blk)
        monkeypatch.setenv("SPMV_MAXPASS", passes)
        prob = DeviceProblem(pkg, gpu, N, N, rp, ci, va, x)
        y = prob.run(pkg.capi.TILED)
        desc = prob.A.plan_describe(pkg.capi.TILED)
        assert_close_to_oracle(y, y64, mag, f"stencil blocks {blk}/{passes}: {desc}")
        if blk != "256":
            assert int(desc.split("block_list_chunks=")[1].split()[0]) > 0, desc
        prob.A.close()
    monkeypatch.setenv("SPMV_BLOCKS", "0")
    prob = DeviceProblem(pkg, gpu, N, N, rp, ci, va, x)
    assert_close_to_oracle(prob.run(pkg.capi.TILED), y64, mag, "stencil, block lists off")
    assert "block_list_chunks=0" in prob.A.plan_describe(pkg.capi.TILED)
    prob.A.close()

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, S, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg, orc = ge.load_package(), ge.load_oracle()
        capi, W = pkg.capi, pkg.workloads
        dev = torch.device("cuda:0")
        sub = W.BLOCK_ROWS
        w = W.Workload("t", S * world * sub, S * world * sub, "mixed", 16, band=8192)
        handles = []
        for s in range(S):
            r0 = (s * world + rank) * sub
            rp = W.row_ptr(w, r0, sub)
            d_rp = torch.from_numpy(rp).to(dev)
            d_ci = torch.empty(int(rp[-1]), dtype=torch.int32, device=dev)
            d_va = torch.empty(int(rp[-1]), dtype=torch.float32, device=dev)
            capi.synth_fill(w.seed, r0, sub, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
            A = capi.CsrMatrix.from_device(sub, w.cols, d_rp, d_ci, d_va)
            A.plan(capi.TILED)
            handles.append(A)
        sh = pkg.dist.PipelinedSpmv(S, sub, w.cols, [(lambda h: lambda x, y: h.run(capi.TILED, x, y))(h) for h in handles], dev)
        if rank == 0:
            capi.synth_x(w.seed, 0, w.cols, sh.x)
        sh.broadcast_x(0)
        for _ in range(3):
            sh.step()
        y = sh.finish()
        torch.cuda.synchronize()
        y = y.cpu().numpy()
        rp_all = W.row_ptr(w)
        ci, va = orc.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp_all)
        x = orc.synth_x(w.seed, 0, w.cols)
        y64, mag = orc.spmv_f64(rp_all, ci, va, x)
        err = np.abs(y.astype(np.float64) - y64)
        q.put((rank, bool(np.all(err <= 1e-5 * mag + 1e-37))))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu_and_exchange_y(built, gpu):
    """The N>1 path of bench.py with real HIP kernels: block-cyclic blocks, all-gather per group
    (through gloo and the host here, RCCL on a multi-GPU node)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=10) for _ in range(2)) == [(0, True), (1, True)]


def test_persistent_launch_knob_gives_the_same_answer(pkg, oracle, gpu, monkeypatch):
    """SPMV_PERSIST=1 (read when the plan is made) walks the chunks with resident workgroups and a
    software-pipelined stream; measured slower but it must stay correct."""
    w = pkg.workloads.Workload("t", 8 * pkg.workloads.BLOCK_ROWS, 8 * pkg.workloads.BLOCK_ROWS, "mixed", 16, band=4096)
    monkeypatch.setenv("SPMV_PERSIST", "1")
    monkeypatch.setenv("SPMV_AUTOTUNE", "0")
    prob = synth_problem(pkg, oracle, gpu, w)
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    for name in ("adaptive", "tiled"):
        assert_close_to_oracle(prob.run(pkg.capi.VARIANTS[name]), y64, mag, f"persist/{name}")
    for blk in ("512", "1024"):
        monkeypatch.setenv("SPMV_TILED_BLOCK", blk)
        p2 = synth_problem(pkg, oracle, gpu, w)
        assert_close_to_oracle(p2.run(pkg.capi.TILED), y64, mag, f"persist/tiled/{blk}")


def test_plan_uses_16bit_columns_only_where_the_span_is_staged(pkg, oracle, gpu):
    """Banded matrix -> (almost) every chunk carries 16-bit offsets; uniform columns -> none; the 32-bit
    and the 16-bit path agree bit for bit (same products, same order)."""
    import re
    W = pkg.workloads
    wb = W.Workload("t", 32 * W.BLOCK_ROWS, 32 * W.BLOCK_ROWS, "mixed", 16, band=8192)
    prob = synth_problem(pkg, oracle, gpu, wb)
    y16 = prob.run(pkg.capi.TILED)
    desc = prob.A.plan_describe(pkg.capi.TILED)
    chunks = int(re.search(r"chunks=(\d+)", desc).group(1))
    n16 = int(re.search(r"col16_chunks=(\d+)", desc).group(1))
    assert n16 >= 0.9 * chunks, desc
    y64, mag = oracle.spmv_f64(prob.row_ptr, prob.col_idx, prob.vals, prob.x)
    assert_close_to_oracle(y16, y64, mag, "col16")

    os.environ["SPMV_COL16"] = "0"
    try:
        p32 = synth_problem(pkg, oracle, gpu, wb)
        os.environ["SPMV_TILED_BLOCK"] = re.search(r"block=(\d+)", desc).group(1)
        os.environ["SPMV_MAXPASS"] = re.search(r"maxpass=(\d+)", desc).group(1)
        y32 = p32.run(pkg.capi.TILED)
        assert "col16_chunks=0 " in p32.A.plan_describe(pkg.capi.TILED)
        assert np.array_equal(y16.view(np.uint32), y32.view(np.uint32))
    finally:
        for k in ("SPMV_COL16", "SPMV_TILED_BLOCK", "SPMV_MAXPASS"):
            os.environ.pop(k, None)

    wu = W.Workload("t", 32 * W.BLOCK_ROWS, 32 * W.BLOCK_ROWS, "mixed", 16, band=0)
    pu = synth_problem(pkg, oracle, gpu, wu)
    pu.run(pkg.capi.TILED)
    assert "col16_chunks=0 " in pu.A.plan_describe(pkg.capi.TILED)


def test_dist_selftest_world_of_one_gpu(pkg, gpu):
    """include/spmv_dist.h end to end from C++ (bin/spmv_dist_selftest): RCCL communicator over the visible GPUs (one on
    this box: the collectives degenerate, the partition / plan hand-over / step path does not), every rank's y
    bit-identical to the whole matrix through a single handle.  On an 8-GPU node the same binary runs 8 ranks."""
    import json
    import subprocess
    import torch
    n = torch.cuda.device_count()
    for extra in ([], ["--unequal", "--variant", "adaptive"]):
        p = subprocess.run([str(pkg.capi.DIST_SELFTEST_PATH), "--ranks", str(n), "--rows-per-rank", str(1 << 18)] + extra,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        out = json.loads(p.stdout.strip().splitlines()[-1])
        assert out["world"] == n and out["rows_differing_from_single_handle"] == 0 and out["step_ms"] > 0


@pytest.mark.parametrize("args", [
    ["--pipeline", "4", "--exchange", "allgather"],                                  # RCCL communicator over the visible GPUs
    ["--pipeline", "4", "--exchange", "p2p", "--variant", "auto", "--band", "0"],
    ["--local", "--ranks", "4", "--pipeline", "4", "--exchange", "peer"],           # world 4 on ONE GPU: peer stores, no RCCL
    ["--local", "--ranks", "3", "--pipeline", "2", "--exchange", "peer", "--variant", "adaptive", "--band", "200000"],
    ["--local", "--ranks", "2", "--pipeline", "1", "--exchange", "peer", "--variant", "auto", "--band", "200000",
     "--rows-per-rank", str(1 << 20)],
    ["--local", "--ranks", "4", "--pipeline", "4", "--exchange", "peer", "--footprint", "--band", "8192"],   # the optional footprint exchange
    ["--pipeline", "2", "--exchange", "p2p", "--footprint"],
    # x doubled from step to step, a slow reader of y_full between the steps, spmv_dist_pipe_release before the next step:
    # the peers' stores of step t+1 wait for the readers of step t
    ["--local", "--ranks", "4", "--pipeline", "4", "--exchange", "peer", "--vary-x"],
    ["--pipeline", "4", "--exchange", "allgather", "--vary-x"],
], ids=["allgather", "p2p-auto", "peer-4ranks", "peer-3ranks", "peer-sorted-blocks", "peer-footprint", "p2p-footprint",
        "peer-vary-x", "allgather-vary-x"])
def test_dist_pipeline_selftest(pkg, gpu, args):
    """The pipelined step of include/spmv_dist.h from C++: S block-cyclic row blocks per rank, the
    exchange of block group s on a side stream under the multiply of block s+1, three steps back to back, every rank's y
    bit-identical to the whole matrix through a single handle planned alike.  With RCCL the world is the visible GPUs (one
    here); with --local the ranks have no communicator and exchange by peer stores, so world > 1 -- block-cyclic offsets,
    plan hand-over, cross-rank event ordering -- runs on this one GPU."""
    import json
    import subprocess
    import torch
    cmd = [str(pkg.capi.DIST_SELFTEST_PATH)] + args
    if "--local" not in args:
        cmd += ["--ranks", str(torch.cuda.device_count())]
    if "--rows-per-rank" not in args:
        cmd += ["--rows-per-rank", str(1 << 18)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["rows_differing_from_single_handle"] == 0 and out["step_ms"] > 0
    if "--vary-x" in args:
        assert out["vary_x_snapshot_words_differing"] == 0
    if "--local" in args:
        assert out["world"] == int(args[args.index("--ranks") + 1]) and out["exchange"] == "peer"
    if "--footprint" in args and out["world"] > 1:
        # a band of 8192 columns: a rank's footprint is its own rows plus 4096 on either side of each of its blocks --
        # most of y is never sent to it (those rows keep the NaN they were preset to)
        assert out["footprint_exchange"] is True
        assert out["rows_never_sent_to_a_rank_that_does_not_need_them"] > 0


def test_column_range(pkg, gpu):
    """spmv_csr_column_range: a row block's x footprint (what the footprint exchange of spmv_dist.h is set up with)."""
    capi = pkg.capi
    rp = np.array([0, 2, 2, 5], np.int32); ci = np.array([7, 3, 11, 4, 9], np.int32); va = np.ones(5, np.float32)
    prob = DeviceProblem(pkg, gpu, 3, 20, rp, ci, va, np.ones(20, np.float32))
    assert prob.A.column_range() == (3, 11)
    prob.A.close()
    prob = DeviceProblem(pkg, gpu, 2, 20, np.zeros(3, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), np.ones(20, np.float32))
    assert prob.A.column_range() == (20, -1)
    prob.A.close()


def test_calibration_kernels_and_marker(pkg, gpu):
    """The measurement aids of include/spmv_hip.h (spmv_calib_*): they run on buffers of the sizes they are told, refuse the
    arguments they cannot take, and the cold first-launch time of a *_run_host call is on record beside the warm one."""
    import torch
    capi = pkg.capi
    table = torch.zeros(1 << 22, dtype=torch.float32, device=gpu)          # 16 MiB = 2^17 lines
    sink = torch.zeros(16, dtype=torch.float32, device=gpu)
    capi.calib_marker(7)
    capi.calib_stream(table, table.numel() * 4, sink)
    for touch in (1, 2, 4):
        capi.calib_gather(table, 1 << 17, 1 << 17, touch, sink)
    capi.calib_store(table, 1 << 20, 4)
    capi.calib_store(table, 1 << 20, 16)
    torch.cuda.synchronize()
    assert float(sink.sum()) == 0.0                                          # the sink is never written
    assert float(table[: (1 << 20) // 4].min()) == 1.0 and float(table[(1 << 20) // 4:].max()) == 0.0
    for bad in (lambda: capi.calib_marker(0), lambda: capi.calib_marker(70000),
                lambda: capi.calib_stream(table, 24, sink),                  # not a multiple of 16
                lambda: capi.calib_gather(table, 3 << 10, 16, 1, sink),      # lines not a power of two
                lambda: capi.calib_gather(table, 1 << 10, 1 << 11, 1, sink),  # more lines than the table has
                lambda: capi.calib_gather(table, 1 << 10, 16, 3, sink),
                lambda: capi.calib_store(table, 1 << 20, 8)):
        with pytest.raises(capi.SpmvError) as e:
            bad()
        assert e.value.status == capi.ERR_INVALID
    # the launchers' figure: first (cold) launch kept beside the warm one
    rp = np.arange(0, 4097, dtype=np.int32) * 8
    ci = (np.arange(4096 * 8, dtype=np.int32) * 7) % 4096
    va = np.ones(4096 * 8, np.float32)
    A = capi.CsrMatrix.from_host(4096, 4096, rp, ci, va)
    y = np.empty(4096, np.float32)
    warm = A.run_host(capi.ADAPTIVE, np.ones(4096, np.float32), y)
    cold = capi.lib().spmv_last_first_launch_ms()
    assert warm > 0 and cold > 0 and np.all(y == 8.0)
    A.close()
