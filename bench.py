#!/usr/bin/env python3
"""bench.py -- fp32 CSR SpMV throughput on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path over the synthetic matrix: y = A x through the C ABI
(spmv_csr_run, all launches on torch's current stream); with N > 1 ranks the step also
concatenates the output slices on every rank with one RCCL all-gather (BASELINE north_star).

  N = 1   config 4 of BASELINE.json: 16Mi x 16Mi, 256Mi nnz, mixed row lengths (the config the
          metric/target is quoted on); inputs resident in HBM before the timed region.  BASELINE
          fixes the row-length law, not the column law; the headline uses banded columns
          (band 8192 = the 2-D 4096 x 4096 mesh coupling of a 16Mi-unknown problem) and the same
          line reports the other column laws, uniform-random included, under "other_workloads".
  N > 1   config 5 generalised: (N*16Mi)^2, one 16Mi-row / 256Mi-nnz block per rank (weak
          scaling: per-GPU work fixed), x (N*64 MiB) on every rank, y all-gathered every step.
          --scaling strong: the SAME (128Mi)^2 / 2Gi-nnz matrix of config 5 for every N, its 32 row
          blocks of 4Mi rows dealt block-cyclically over the N ranks (N = 1 holds all of it: 17.5 GB).

value      = algorithmic bytes of all ranks x K / wall time of the K timed steps  [GB/s]
roofline   = algorithmic bytes of one launch / mean launch time by HIP events on the launch stream
cpu_baseline = the CPU oracle (oracle/, checker code) walking a bounded row sample of the same
             matrix on the host cores, timed in the same run; a baseline, not a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COPY_CEILING_GBS = 6290.0  # same guide: what a float4 copy reaches (79 % of spec) -- no kernel MOVES bytes faster than this
L2_LINE_PEAK_G = 270.0     # profiles/r03_gather_lines_ubench.jsonl: 128-byte line requests per second the L2s serve (G/s)


def touched_x_bytes(torch, d_ci, cols):
    """4 bytes x the DISTINCT columns the matrix references (the CSR-algorithmic count charges all of x: a shard of
    config 5 with banded columns touches 64 MiB of its 512 MiB)."""
    mask = torch.zeros(cols, dtype=torch.bool, device=d_ci.device)
    step = 1 << 25
    for k in range(0, d_ci.numel(), step):
        mask[d_ci[k:k + step].long()] = True
    return 4 * int(mask.sum().item())


def honest_fields(be, ms, nnz, rows, touched_x, plan_text, traffic_entry):
    """What the line says about a kernel time besides 'algorithmic bytes / time / 8 TB/s' (VERDICT round 2, item 2)."""
    out = {}
    touched = 8 * nnz + 4 * (rows + 1) + 4 * rows + touched_x
    out["bytes_touched"] = touched                    # x counted as the distinct columns referenced
    out["frac_of_peak_touched"] = round(touched / ms / 1e6 / HBM_PEAK_GBS, 4)
    if be / ms / 1e6 > COPY_CEILING_GBS:
        out["exceeds_copy_ceiling"] = (f"algorithmic bytes / time = {be / ms / 1e6:.0f} GB/s is above the {COPY_CEILING_GBS:.0f} GB/s "
                                       "a pure copy reaches: the kernel does not move that many bytes (x columns never "
                                       "referenced, 16-bit column copies) -- read frac_of_peak_touched / frac_hbm_counters")
    if traffic_entry is not None:
        out["hbm_bytes_counters"] = traffic_entry["hbm_bytes_per_launch"]
        out["frac_hbm_counters"] = round(traffic_entry["hbm_bytes_per_launch"] / ms / 1e6 / HBM_PEAK_GBS, 4)
        out["counters_from"] = "profiles/" + traffic_entry.get("profile", "?")
    # the panel family is bound by L2 line requests, not by HBM bytes: a second roofline with that ceiling
    if "auto -> panel" in plan_text or plan_text.startswith(("panel_columns", "sorted_blocks")):
        if "lines_per_nonzero=" in plan_text:
            lines = float(plan_text.split("lines_per_nonzero=")[1].split()[0]) * nnz
            what = "lines_per_nonzero of the plan x nonzeros (distinct 128-byte lines of x per block, each requested about once)"
        else:
            lines = float(nnz)
            what = "one line request per nonzero (rows ascend inside a tile: the lanes of an instruction hold different lines)"
        out["roofline_l2_gather"] = {"bound": "l2_gather", "achieved": round(lines / ms / 1e6, 1), "peak": L2_LINE_PEAK_G,
                                     "unit": "G line requests/s", "frac": round(lines / ms / 1e6 / L2_LINE_PEAK_G, 4),
                                     "requests": what,
                                     "peak_from": "tools/ubench_gather_lines.hip: 64 lanes on 64 distinct L2-resident lines"}
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variant", default=os.environ.get("SPMV_BENCH_VARIANT", "auto"),
                    help="auto (default: the library picks tiled or panel from the matrix) | tiled | panel | adaptive | ...")
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5shard"],
                    help="N=1 workload (default: c4); c5shard = rows [0,16Mi) of config 5's (128Mi)^2 matrix with all 128Mi columns "
                         "and the full 512 MiB x: what ONE MI355X of the 8 multiplies")
    ap.add_argument("--band", type=int, default=int(os.environ.get("SPMV_BENCH_BAND", "8192")),
                    help="column law: >0 = diagonal band of that many columns (default 8192), 0 = uniform random")
    ap.add_argument("--no-extras", action="store_true", help="skip the other column laws / configs (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vendor", default=os.environ.get("SPMV_BENCH_VENDOR", "if-ready"), choices=["if-ready", "wait", "off"],
                    help="rocSPARSE's best algorithm beside every workload of other_workloads: librocsparse.so (0.5 GB) is "
                         "loaded by a background thread from the start of the run; 'if-ready' (default) uses it for the workloads "
                         "that begin after it has loaded and never waits -- on a freshly booted box the load alone takes "
                         "minutes --, 'wait' waits for it (profiles/r03_bench_n1_full.json), 'off' skips it")
    ap.add_argument("--cpu-sample-rows", type=int, default=1 << 23)
    ap.add_argument("--rows-per-gpu", type=int, default=16 << 20, help="N>1: rows of each rank's block (default 16Mi)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("SPMV_BENCH_PIPELINE", "4")),
                    help="N>1: row blocks per rank (block-cyclic); the all-gather of one group overlaps the next multiply")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): 16Mi rows per rank, the matrix grows with N; strong: config 5's (128Mi)^2 matrix "
                         "for every N (--total-blocks row blocks dealt over the ranks)")
    ap.add_argument("--total-blocks", type=int, default=32, help="--scaling strong: row blocks of the fixed matrix")
    ap.add_argument("--exchange", default=os.environ.get("SPMV_BENCH_EXCHANGE", "allgather"), choices=["allgather", "p2p"],
                    help="N>1: concatenate y with RCCL's all-gather (default) or with one direct send/recv pair per peer")
    ap.add_argument("--footprint", action="store_true",
                    help="N>1, --backend native --exchange p2p only: the optional footprint exchange of include/spmv_dist.h -- every "
                         "rank receives only the rows of y its own columns reference (a band of 8192: 4096 rows either side of each "
                         "of its blocks) instead of all of y.  NOT the north-star's all-gather: the line says so")
    ap.add_argument("--backend", default=os.environ.get("SPMV_BENCH_BACKEND", "nccl"),
                    help="nccl (= RCCL through torch.distributed, default) | native (RCCL called from C++: libspmv_dist.so's "
                         "pipelined step, include/spmv_dist.h; torch.distributed/gloo only carries the id, the barrier and the "
                         "timing reduction) | gloo (rehearsal of the N>1 path with ranks sharing one GPU)")
    return ap.parse_args()


def main():
    t_main = time.perf_counter()
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: libspmv_hip has no CPU path")
    # rocSPARSE for the comparison fields: the 0.5 GB library is paged in and dlopen'ed in the background, used only once it
    # is there
    vendor_box = {}
    vendor_thread = None

    def _start_vendor_thread():
        nonlocal vendor_thread
        if not (rank == 0 and world == 1 and args.vendor != "off" and not args.no_extras) or vendor_thread is not None:
            return
        import threading

        def _load_vendor():
            try:
                import ctypes
                import subprocess
                # ctypes' dlopen holds the GIL: on a freshly booted box paging in the 0.5 GB library took 140 s during which
                # the main thread stalled at its next Python step.  A child process reads the file first (no GIL involved);
                # the dlopen afterwards finds it in the page cache.
                subprocess.run(["cat", "/opt/rocm/lib/librocsparse.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                ctypes.CDLL("/opt/rocm/lib/librocsparse.so")
                vendor_box["loaded"] = True
            except Exception as ex:
                vendor_box["error"] = str(ex)[:100]
        vendor_thread = threading.Thread(target=_load_vendor, daemon=True)
        vendor_thread.start()
    ndev = torch.cuda.device_count()
    native = args.backend == "native"
    dev_index = local_rank if args.backend in ("nccl", "native") else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo" if native else args.backend, rank=rank, world_size=world)

    _start_vendor_thread()
    variant = capi.VARIANTS[args.variant]

    # ---- the workload: this rank's row block(s), generated on the device ---------------------------
    S = 1 if world == 1 else max(1, args.pipeline)
    t_setup = time.perf_counter()
    strong = args.scaling == "strong"
    if strong:
        # config 5 itself, whatever N: 8 x 16Mi rows, 2Gi nonzeros, cut into --total-blocks equal row blocks
        w = W.c5(8, band=args.band, rows_per_gpu=args.rows_per_gpu)
        if args.total_blocks % world or (w.rows // args.total_blocks) % W.BLOCK_ROWS:
            raise SystemExit("--total-blocks must be a multiple of N and cut the matrix at multiples of 65536 rows")
        S = args.total_blocks // world
        sub_rows = w.rows // args.total_blocks
        owned = [s * world + rank for s in range(S)]
    elif world == 1 and args.config == "c5shard":
        w = W.c5(8, band=args.band)
        sub_rows = 16 << 20
        owned = [0]
    elif world == 1:
        w = W.config(args.config, band=args.band)
        sub_rows = w.rows
        owned = [0]
    else:
        w = W.c5(world, band=args.band, rows_per_gpu=args.rows_per_gpu)
        if (args.rows_per_gpu // S) % W.BLOCK_ROWS:
            raise SystemExit("--rows-per-gpu / --pipeline must be a multiple of 65536 rows")
        sub_rows = args.rows_per_gpu // S
        owned = [s * world + rank for s in range(S)]           # block-cyclic: group s is contiguous in y
    handles, keep, nnz_local = [], [], 0
    plan_params = None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan_ms = 0.0
    for b in owned:
        r0 = b * sub_rows
        rp = W.row_ptr(w, r0, sub_rows)
        nb = int(rp[-1])
        d_rp = torch.from_numpy(rp).to(dev)
        d_ci = torch.empty(nb, dtype=torch.int32, device=dev)
        d_va = torch.empty(nb, dtype=torch.float32, device=dev)
        capi.synth_fill(w.seed, r0, sub_rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
        A = capi.CsrMatrix.from_device(sub_rows, w.cols, d_rp, d_ci, d_va)
        ev0.record()
        if plan_params is None:
            A.plan(variant)
        else:
            A.plan_set(variant, plan_params)       # every block like the first: one chunk size, one summation order
        ev1.record()
        torch.cuda.synchronize()
        plan_ms += ev0.elapsed_time(ev1)
        if plan_params is None:
            plan_params = A.plan_params(variant)
            if world > 1:                          # ... and like rank 0's first block on every rank
                t = torch.tensor(plan_params, dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
                dist.broadcast(t, src=0)
                agreed = [int(v) for v in t.cpu().tolist()]
                if agreed != plan_params:
                    A.plan_set(variant, agreed)
                    plan_params = agreed
        handles.append(A)
        keep.append((rp, d_rp, d_ci, d_va))
        nnz_local += nb
    rows_local = sub_rows * len(owned)

    exchange_only = None
    if world == 1 and not strong and not native:
        A = handles[0]
        rp, d_rp, d_ci, d_va = keep[0]
        d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
        capi.synth_x(w.seed, 0, w.cols, d_x)
        d_y = torch.empty(rows_local, dtype=torch.float32, device=dev)

        def step():
            A.run(variant, d_x, d_y)

        def multiply_only():
            A.run(variant, d_x, d_y)

        def finish():
            pass
    else:
        def bind(h):
            return lambda x, y: h.run(variant, x, y)
        if native:
            # RCCL from C++ (include/spmv_dist.h): rank 0 makes the 128-byte id, gloo hands it round, every rank joins
            idt = torch.zeros(pkg.dist_native.ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                idt = torch.frombuffer(bytearray(pkg.dist_native.unique_id()), dtype=torch.uint8).clone()
            if world > 1:
                dist.broadcast(idt, src=0)
            sh = pkg.dist_native.NativePipeline(world, rank, bytes(idt.numpy().tobytes()), S, sub_rows, w.cols, handles, variant, dev,
                                                exchange=args.exchange)
            if args.footprint:
                if args.exchange != "p2p":
                    raise SystemExit("--footprint needs --exchange p2p")
                mine = torch.tensor([[lo if hi >= 0 else 0, hi + 1 if hi >= 0 else 0] for lo, hi in (h.column_range() for h in handles)],
                                    dtype=torch.int64)
                every = [torch.zeros_like(mine) for _ in range(world)]
                if world > 1:
                    dist.all_gather(every, mine)
                else:
                    every = [mine]
                sh.set_footprint([[int(v) for v in t[:, 0]] for t in every], [[int(v) for v in t[:, 1]] for t in every])
        else:
            sh = pkg.dist.PipelinedSpmv(S, sub_rows, w.cols, [bind(h) for h in handles], dev, exchange=args.exchange)
        if rank == 0:
            capi.synth_x(w.seed, 0, w.cols, sh.x)
        if world > 1:
            sh.broadcast_x(0)                # the one-off distribution of the dense vector
        exchange_only = sh.exchange_only
        d_x = sh.x
        step = sh.step
        finish = sh.finish

        def multiply_only():
            for s_, h in enumerate(handles):
                a_, b_ = sh.block_rows(s_)
                h.run(variant, d_x, sh.y_full[a_:b_])
    # setup ends with ~40 ms of untimed launches: clocks and caches reach their steady state before
    # the W warm-up steps of the contract (the first launches after the host-side setup run ~1-2 % slow)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.04:
        multiply_only()
        torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up, then EXACTLY K timed steps between barrier + synchronize ------------------------
    for _ in range(args.warmup):
        step()
    finish()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    finish()                                  # every all-gather of the K steps has landed
    ev1.record()
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    step_ms_events = ev0.elapsed_time(ev1) / args.steps        # HIP events on the launch stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per step, per rank: vals+col_idx, row_ptr of every owned block, y written, x read once
    bytes_rank = 8 * nnz_local + 4 * (rows_local + len(owned)) + 4 * rows_local + 4 * w.cols
    bytes_all = bytes_rank * world
    value = bytes_all * args.steps / elapsed / 1e9
    gflops = 2.0 * nnz_local * world * args.steps / elapsed / 1e9

    # ---- the kernel alone (no collective): mean launch time by HIP events on its stream -----------
    iters = max(10, args.steps)
    exchange_ms = None
    if world == 1 and not strong and not native:
        kernel_ms = A.time(variant, d_x, d_y, iters)          # spmv_csr_time: events inside the library
    else:
        ev0.record()
        for _ in range(iters):
            multiply_only()
        ev1.record()
        torch.cuda.synchronize()
        kernel_ms = ev0.elapsed_time(ev1) / iters
        if world > 1:                                          # the all-gathers of one step alone, no multiply
            torch.cuda.synchronize(); barrier()
            ev0.record()
            for _ in range(iters):
                exchange_only()
            finish()
            ev1.record()
            torch.cuda.synchronize(); barrier()
            exchange_ms = ev0.elapsed_time(ev1) / iters
    achieved = bytes_rank / (kernel_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        # HBM traffic per SpMV from the PMC passes (tools/profile.sh + summarize_profile.py).  It cannot be measured
        # inside this run (counters need rocprofv3 around the process), so it is REPLAYED from the committed file --
        # and only when that measurement was taken with the very plan this run uses.
        plan_now = handles[0].plan_describe(variant)
        resolved = capi.lib().spmv_variant_name(handles[0].plan_params(variant)[0]).decode()
        traffic, traffic_source = None, None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists() and world == 1 and not strong:
            try:
                ent = json.loads(tfile.read_text()).get(f"{resolved}:{w.name}:band{w.band}", {})
                same_plan = ent.get("plan") is not None and ent["plan"].split(": ")[-1] == plan_now.split(": ")[-1]
                if same_plan:
                    traffic = ent.get("hbm_bytes_per_launch")
                    traffic_source = f"replayed from profiles/{ent.get('profile')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same plan)"
                elif ent:
                    traffic_source = f"profiles/{ent.get('profile')} was taken with another plan ({ent.get('plan')}): not replayed"
            except Exception:
                traffic = None
        def _n(key):
            return int(plan_now.split(key + "=")[1].split()[0]) if key + "=" in plan_now else 0
        if resolved == "panel":
            dom_kernel = "k_colsort" if "sorted_blocks=" in plan_now else "k_panel"
        elif resolved == "tiled" and (_n("col16_chunks") or _n("sorted_chunks")):
            # one launch: all chunks 16-bit -> k_tiled16, all sorted -> k_sorted, otherwise the three bodies in k_tiled_mixed
            dom_kernel = ("k_tiled16" if _n("col16_chunks") == _n("chunks") else
                          "k_sorted" if _n("sorted_chunks") == _n("chunks") else "k_tiled_mixed")
        else:
            dom_kernel = ("k_adaptive" if resolved in ("adaptive", "tiled") else
                          "k_wave_bundle" if "block_rows=" in plan_now else f"k_{resolved}")
        out = {
            "metric": "fp32 CSR SpMV throughput in CSR-algorithmic bytes per second (8/nnz + row_ptr + x + y over time; "
                      "roofline.frac_hbm_counters is the FETCH_SIZE/WRITE_SIZE-based figure)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": w.describe() + (" -- the per-GPU shard: rows [0,16Mi), all columns, the full x"
                                                    if world == 1 and args.config == "c5shard" else ""),
                       "variant": args.variant,
                       "rows_per_gpu": rows_local, "nnz_per_gpu": nnz_local,
                       "parallelism": "single GPU" if world == 1 and not strong and not native else
                       f"{S} block-cyclic row blocks per rank x{world} ranks, all-gather(y) of group s "
                       f"({'RCCL all_gather' if args.exchange == 'allgather' else 'direct send/recv per peer'}) "
                       f"overlapped with the multiply of block s+1, "
                       f"{'RCCL called from C++ (libspmv_dist.so: spmv_dist_pipe_step)' if native else args.backend}"
                       + (" -- FOOTPRINT exchange: every rank receives only the rows of y its columns reference, not the "
                          "all-gather of the north-star" if args.footprint else ""),
                       "devices": (f"{world} ranks on {min(world, max(ndev, 1))} device(s)" +
                                   ("" if ndev >= world and args.backend in ("nccl", "native") else
                                    " -- RANKS SHARE DEVICES: a rehearsal of the plumbing, not a scaling measurement")),
                       "algorithmic_bytes_per_gpu": bytes_rank},
            "pct_of_hbm_peak": round(100.0 * value / world / HBM_PEAK_GBS, 2),
            "gflops": round(gflops, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         # the same launch priced in the bytes the counters saw (16-bit columns: fewer than algorithmic)
                         "achieved_hbm_counters": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9, 2),
                         "frac_hbm_counters": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "definition": "achieved = CSR-algorithmic bytes of one SpMV / kernel_ms; "
                                       "achieved_hbm_counters = FETCH_SIZE+WRITE_SIZE bytes of one SpMV / kernel_ms",
                         "kernel": dom_kernel,
                         "kernel_ms": round(kernel_ms, 5), "timing": "HIP events on the launch stream"},
            "step_ms_events": round(step_ms_events, 5), "multiply_only_ms": round(kernel_ms, 5),
            "exchange_only_ms": None if exchange_ms is None else round(exchange_ms, 5),
            "plan_ms": round(plan_ms, 4), "plan_bytes": sum(h.plan_bytes(variant) for h in handles),
            "plan": handles[0].plan_describe(variant),
            "setup_s": round(setup_s, 2),
        }

    if rank == 0:
        print(f"[bench] headline measured at {time.perf_counter() - t_main:.1f} s", file=sys.stderr, flush=True)
        _start_vendor_thread()
    # ---- CPU baseline: rank 0, N = 1 only, bounded sample of the same matrix ------------------------
    if rank == 0 and world == 1 and not strong and not native and not args.no_cpu_baseline:
        orc = ge.load_oracle()
        n = min(args.cpu_sample_rows, rows_local)
        s0 = ((rows_local - n) // 2 // W.BLOCK_ROWS) * W.BLOCK_ROWS  # a window from the middle of the matrix
        s1 = s0 + n
        k0, k1 = int(rp[s0]), int(rp[s1])
        rps = (rp[s0:s1 + 1].astype(np.int64) - k0).astype(np.int32)
        ci = d_ci[k0:k1].cpu().numpy()
        va = d_va[k0:k1].cpu().numpy()
        x = d_x.cpu().numpy()
        # the GPU box gives one GPU a 16-CPU share of a much larger host: use that many threads
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = int(os.environ.get("SPMV_CPU_THREADS", min(avail, 16)))
        y_cpu = orc.spmv(rps, ci, va, x, threads=cores)            # untimed first touch
        reps, t_cpu = 0, 0.0
        while t_cpu < 10.0 and reps < 5000:
            t1 = time.perf_counter()
            y_cpu = orc.spmv(rps, ci, va, x, threads=cores)
            t_cpu += time.perf_counter() - t1
            reps += 1
        b_sample = W.algorithmic_bytes(n, w.cols, k1 - k0)
        out["cpu_baseline"] = {"value": round(b_sample * reps / t_cpu / 1e9, 3), "unit": "GB/s", "cores": cores,
                               "kind": "port",
                               "sample": f"rows [{s0},{s1}) of the same matrix ({n} rows, {k1 - k0} nnz, full x), "
                                         f"{reps} passes, {t_cpu:.1f} s, oracle_spmv_csr_mt"}
        # the reference's own CPU path as it runs it: the dense SgemvCPU loop (tester.cpp:36-45), one thread,
        # 4096 x 4096 at 50 % zeros (test/main.cpp:4, tester.cpp:106) -- restated in oracle/, timed here
        Ad, xd = W.dense_random(4096, 4096, 0.5, seed=1)
        orc.sgemv_dense(Ad, xd)
        t1 = time.perf_counter()
        for _ in range(3):
            orc.sgemv_dense(Ad, xd)
        dense_ms = (time.perf_counter() - t1) / 3 * 1e3
        out["cpu_baseline"]["reference_dense_loop"] = {
            "what": "SgemvCPU restated (oracle_sgemv_dense), 4096x4096, 50 % zeros, 1 thread",
            "ms": round(dense_ms, 2), "dense_GBs": round(4096 * 4096 * 4 / dense_ms / 1e6, 2)}
        # parity on the sample while both results are at hand (the oracle as checker)
        y64, mag = orc.spmv_f64(rps, ci, va, x)
        got = d_y[s0:s1].cpu().numpy().astype(np.float64)
        out["parity_sample"] = {"rows": n, "max_err_over_1e-5_bound": float(np.max(np.abs(got - y64) / (1e-5 * mag + 1e-37))),
                                "bit_identical_rows_vs_seq_oracle": int(np.sum(d_y[s0:s1].cpu().numpy() == y_cpu))}

    # ---- the other column laws and configs, kernel time only (N = 1) ------------------------------
    if rank == 0 and world == 1 and not strong and not native and not args.no_extras:
        del A, d_rp, d_ci, d_va, d_x, d_y, handles, keep
        torch.cuda.empty_cache()
        extras = []
        try:
            traffic_all = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        except Exception:
            traffic_all = {}
        # rocSPARSE beside every workload (the reference's vendor slot, cublas.cu:33, is a comparison there too); the
        # library is part of the ROCm image -- when it does not load the fields are simply absent
        rocs, vendor = None, None

        def vendor_ready():
            """The rocSPARSE handle once the background load has finished (created here, on the thread that owns the stream)."""
            nonlocal rocs, vendor
            if rocs is not None or args.vendor == "off" or "error" in vendor_box:
                return rocs
            if args.vendor == "wait" and vendor_thread is not None:
                vendor_thread.join()
            if not vendor_box.get("loaded"):
                return None
            try:
                import ctypes
                import importlib.util
                spec = importlib.util.spec_from_file_location("vendor_compare", ROOT / "tools" / "vendor_compare.py")
                vendor = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(vendor)
                rocs = vendor.RocSparse()
                rocs._ok(rocs.L.rocsparse_set_stream(rocs.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")
            except Exception as ex:
                vendor_box["error"] = str(ex)[:100]
                rocs = None
            return rocs
        todo = [("c4", 0), ("c4", 2048), ("c4", 65536), ("c4", 200000), ("c4", 1000000), ("c2", 0), ("c2", 8192), ("c3", 0),
                ("c3", 8192), ("c5", 8192), ("c5", 0)]
        print(f"[bench] cpu baseline done at {time.perf_counter() - t_main:.1f} s", file=sys.stderr, flush=True)
        for cname, band in todo:
            if cname == args.config and band == args.band:
                continue
            t_w = time.perf_counter()
            if cname == "c5":
                # config 5's per-GPU shard: rows [0, 16Mi) of the (128Mi)^2 matrix, all 128Mi columns, the full
                # 512 MiB x (2 818 572 292 algorithmic bytes) -- what ONE MI355X of the 8 multiplies
                we = W.c5(8, band=band)
                n_loc = 16 << 20
                label = "c5 shard (rows [0,16Mi) of " + we.describe() + ")"
            else:
                we = W.config(cname, band=band)
                n_loc = we.rows
                label = we.describe()
            rpe = W.row_ptr(we, 0, n_loc)
            nnz_e = int(rpe[-1])
            e_rp = torch.from_numpy(rpe).to(dev)
            e_ci = torch.empty(nnz_e, dtype=torch.int32, device=dev)
            e_va = torch.empty(nnz_e, dtype=torch.float32, device=dev)
            e_x = torch.empty(we.cols, dtype=torch.float32, device=dev)
            e_y = torch.empty(n_loc, dtype=torch.float32, device=dev)
            capi.synth_fill(we.seed, 0, n_loc, we.rows, we.cols, we.band, e_rp, e_ci, e_va)
            capi.synth_x(we.seed, 0, we.cols, e_x)
            Ae = capi.CsrMatrix.from_device(n_loc, we.cols, e_rp, e_ci, e_va)
            be = W.algorithmic_bytes(n_loc, we.cols, nnz_e)
            # (round 2: "auto" ran first, cold, and read up to 3.5 % slower than the variant it resolves to)
            def timed(v, iters=20, groups=3):
                Ae.plan(v)
                Ae.time(v, e_x, e_y, 10)
                return min(Ae.time(v, e_x, e_y, iters) for _ in range(groups))
            # the variants take turns, three rounds of 20 launches each after a warm round, and keep their best: every one
            # meets the clocks and caches in the same states
            names = ("adaptive", "tiled") + (("panel",) if band == 0 else ()) + ("auto",)
            times = {}
            for vn in names:
                Ae.plan(capi.VARIANTS[vn])
                Ae.time(capi.VARIANTS[vn], e_x, e_y, 10)
            for _round in range(3):
                for vn in names:
                    ms = Ae.time(capi.VARIANTS[vn], e_x, e_y, 20)
                    times[vn] = min(times.get(vn, ms), ms)
            best = min(times.items(), key=lambda kv: kv[1])
            auto = (Ae.plan_describe(capi.VARIANTS["auto"]), times["auto"])
            auto_plan = auto[0]
            resolved = auto_plan.split("auto -> ")[1].split(":")[0] if "auto -> " in auto_plan else "auto"
            # the kernel BASELINE.json's config string names for this config, timed beside the library's choice
            named = {"c2": ("scalar",), "c3": ("wave", "wave_pipe")}.get(cname, ())
            named_out = {}
            for vn in named:
                ms = timed(capi.VARIANTS[vn], iters=10, groups=2)
                named_out[vn] = {"kernel_ms": round(ms, 5), "frac_of_peak": round(be / ms / 1e6 / HBM_PEAK_GBS, 4)}
            tx = touched_x_bytes(torch, e_ci, we.cols)
            tkey = f"{resolved}:{'c5x8' if cname == 'c5' else cname}:band{band}"
            tent = traffic_all.get(tkey)
            if tent is not None and tent.get("plan", "").split(": ", 1)[-1] != auto_plan.split(": ", 1)[-1]:
                tent = None                                      # taken with another plan: not replayed
            entry = {"workload": label, "auto": auto_plan.split(":")[0], "auto_plan": auto_plan,
                     "auto_kernel_ms": round(auto[1], 5),
                     "auto_frac_of_peak": round(be / auto[1] / 1e6 / HBM_PEAK_GBS, 4),
                     "best_variant": best[0], "best_kernel_ms": round(best[1], 5),
                     "best_frac_of_peak": round(be / best[1] / 1e6 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes": be, **honest_fields(be, auto[1], nnz_e, n_loc, tx, auto_plan, tent),
                     **({"config_named_kernel": named_out} if named_out else {})}
            if resolved == "tiled" and resolved in times:
                entry["resolved_variant_kernel_ms"] = round(times[resolved], 5)   # the same plan timed under its own name
            if vendor_ready() is not None:
                try:
                    # (the two algorithms that were rocSPARSE's best on every workload of rounds 1 and 2; all four are in
                    # tools/vendor_compare.py -> profiles/r02_vendor_compare.jsonl)
                    rt = vendor.rocsparse_times(rocs, n_loc, we.cols, nnz_e, e_rp, e_ci, e_va, e_x, e_y, iters=10,
                                                algs={k: vendor.ALGS[k] for k in ("csr_adaptive", "csr_nnzsplit")})
                    if rt:
                        ba = min(rt, key=lambda a: rt[a][0])
                        entry["rocsparse_best_ms"] = round(rt[ba][0], 5)
                        entry["rocsparse_best_algorithm"] = ba
                        entry["speedup_vs_rocsparse_best"] = round(rt[ba][0] / auto[1], 2)
                except Exception as ex:                      # a comparison, never a reason to lose the line
                    entry["rocsparse_error"] = str(ex)[:120]
            extras.append(entry)
            print(f"[bench] {cname} band {band}: {time.perf_counter() - t_w:.1f} s", file=sys.stderr, flush=True)
            Ae.close()
            del e_rp, e_ci, e_va, e_x, e_y
            torch.cuda.empty_cache()
        # a structure the synthetic laws do not cover: a 7-point 3-D stencil (three column clusters 2*200^2 apart)
        print(f"[bench] synthetic extras done at {time.perf_counter() - t_main:.1f} s", file=sys.stderr, flush=True)
        N3, rp3, ci3, va3 = W.stencil7(200)
        t_rp, t_ci, t_va = (torch.from_numpy(a).to(dev) for a in (rp3, ci3, va3))
        t_x = torch.rand(N3, device=dev) * 2 - 1
        t_y = torch.empty(N3, dtype=torch.float32, device=dev)
        A3 = capi.CsrMatrix.from_device(N3, N3, t_rp, t_ci, t_va)
        b3 = W.algorithmic_bytes(N3, N3, len(ci3))
        best, times3 = None, {}
        for vn in ("adaptive", "tiled", "auto"):
            v = capi.VARIANTS[vn]
            A3.plan(v)
            A3.time(v, t_x, t_y, 10)
            ms = min(A3.time(v, t_x, t_y, 20) for _ in range(3))
            times3[vn] = ms
            if vn == "auto":
                auto = (A3.plan_describe(v), ms)
            if best is None or ms < best[1]:
                best = (vn, ms)
        e3 = {"workload": f"stencil7: 7-point stencil on 200^3 = {N3} unknowns, nnz {len(ci3)} (host-built)",
              "auto": auto[0].split(":")[0], "auto_plan": auto[0], "auto_kernel_ms": round(auto[1], 5),
              "auto_frac_of_peak": round(b3 / auto[1] / 1e6 / HBM_PEAK_GBS, 4),
              "best_variant": best[0], "best_kernel_ms": round(best[1], 5),
              "best_frac_of_peak": round(b3 / best[1] / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes": b3,
              **honest_fields(b3, auto[1], len(ci3), N3, touched_x_bytes(torch, t_ci, N3), auto[0], None)}
        if vendor_ready() is not None:
            try:
                rt = vendor.rocsparse_times(rocs, N3, N3, len(ci3), t_rp, t_ci, t_va, t_x, t_y, iters=10,
                                            algs={k: vendor.ALGS[k] for k in ("csr_adaptive", "csr_nnzsplit")})
                if rt:
                    ba = min(rt, key=lambda a: rt[a][0])
                    e3["rocsparse_best_ms"] = round(rt[ba][0], 5)
                    e3["rocsparse_best_algorithm"] = ba
                    e3["speedup_vs_rocsparse_best"] = round(rt[ba][0] / auto[1], 2)
            except Exception as ex:
                e3["rocsparse_error"] = str(ex)[:120]
        extras.append(e3)
        A3.close()
        out["other_workloads"] = extras
        out["vendor_comparison"] = ("rocSPARSE (csr_adaptive, csr_nnzsplit: its best on every workload of rounds 1-2) timed beside "
                                    f"{sum(1 for e in extras if 'rocsparse_best_ms' in e)} of {len(extras)} workloads; --vendor {args.vendor}"
                                    + (": " + vendor_box["error"] if "error" in vendor_box else "")
                                    + ("" if vendor_box.get("loaded") else ": librocsparse.so had not finished loading (run with --vendor wait)"))
        # the headline's config under the other column laws, next to the headline (BASELINE fixes c4's sizes and
        # row-length law, not its column law: the value above holds for the law named in config.workload only)
        laws = {f"band {args.band}" if args.band else "uniform": round(achieved / HBM_PEAK_GBS, 4)}
        for e in extras:
            if e["workload"].startswith(args.config + ":"):
                law = "uniform" if "uniform columns" in e["workload"] else "band " + e["workload"].split("band ")[1].split(")")[0]
                laws[law] = e["auto_frac_of_peak"]
        out["frac_of_peak_by_column_law"] = {"config": args.config, "variant": args.variant, **laws}

    if rank == 0:
        print(json.dumps(out), flush=True)
        print(f"[bench] total {time.perf_counter() - t_main:.1f} s in the process", file=sys.stderr, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
