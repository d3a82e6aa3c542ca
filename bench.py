#!/usr/bin/env python3
"""bench.py -- fp32 CSR SpMV on MI355X: achieved HBM GB/s (BASELINE.json metric), one JSON line on rank 0.

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
          `python3 bench.py --gpus N` starts its own N ranks (one process per GPU) when it was not
          started by a launcher (WORLD_SIZE unset); under torch.distributed.run it is one rank.
          --scaling strong: the SAME (128Mi)^2 / 2Gi-nnz matrix of config 5 for every N.

What the line reports (north_star: "rocprof FETCH_SIZE/WRITE_SIZE reported as achieved HBM GB/s"):
  traffic    = bytes one SpMV moves between the L2s and the memory side, MEASURED IN THIS RUN: before the first
               GPU call this process starts `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --traffic-child ...` and
               the same with WRITE_SIZE (separate passes; role of the reference's profile.sh:18-20); the children run
               every workload of the line once more behind marker dispatches, plus kernels of KNOWN traffic
               (spmv_calib_*) from which the gfx950 corrections are derived in the same pass.
  value      = traffic of all ranks x K / wall time of the K timed steps   [GB/s moved]
  roofline   = traffic of one SpMV / mean launch time by HIP events on the launch stream; frac = that / 8 TB/s.
               effective_* = the same with the CSR-ALGORITHMIC bytes (8/nnz + row_ptr + x + y): the work a CSR
               SpMV stands for; above the traffic figure where 16-bit column copies and untouched x shrink what moves.
  cpu_baseline = the CPU oracle (oracle/, checker code) walking a bounded row sample of the same
               matrix on the host cores, timed in the same run; a baseline, not a target.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time
from collections import defaultdict
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COPY_CEILING_GBS = 6290.0  # same guide: what a float4 copy reaches (79 % of spec)
L2_LINE_PEAK_G = 270.0     # profiles/r03_gather_lines_ubench.jsonl: 128-byte line requests per second the L2s serve (G/s)

# the other column laws / configs of the line (N = 1): (config, band); "c5" = config 5's per-GPU shard
EXTRA_WORKLOADS = [("c4", 0), ("c4", 2048), ("c4", 65536), ("c4", 200000), ("c4", 1000000), ("c2", 0), ("c2", 8192),
                   ("c3", 0), ("c3", 8192), ("c5", 8192), ("c5", 0), ("stencil7", 0)]
CAL_STREAM_BYTES = 1 << 30
CAL_TABLE_LINES = 1 << 24      # 2 GiB: beyond every cache
CAL_STORE_BYTES = 1 << 28
TRAFFIC_RUNS = 2               # launches per workload in a counter pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variant", default=os.environ.get("SPMV_BENCH_VARIANT", "auto"),
                    help="auto (default: the library picks from the matrix) | tiled | panel | adaptive | ...")
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5shard"],
                    help="N=1 workload (default: c4); c5shard = rows [0,16Mi) of config 5's (128Mi)^2 matrix with all 128Mi columns "
                         "and the full 512 MiB x: what ONE MI355X of the 8 multiplies")
    ap.add_argument("--band", type=int, default=int(os.environ.get("SPMV_BENCH_BAND", "8192")),
                    help="column law: >0 = diagonal band of that many columns (default 8192), 0 = uniform random")
    ap.add_argument("--no-extras", action="store_true", help="skip the other column laws / configs (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default=os.environ.get("SPMV_BENCH_TRAFFIC", "measure"), choices=["measure", "replay", "off"],
                    help="HBM traffic per SpMV: 'measure' (default) = two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) of "
                         "this file in --traffic-child mode before the first GPU call, falling back to 'replay' (the figure "
                         "committed in profiles/traffic.json, only under the same plan) when rocprofv3 is missing or fails")
    ap.add_argument("--traffic-timeout", type=float, default=420.0,
                    help="seconds the two counter passes may take together (a cold box spends minutes on the first import of "
                         "torch under the profiler; past the budget the line falls back to the replayed figure)")
    ap.add_argument("--traffic-child", default="", metavar="MANIFEST",
                    help="(internal) run every workload of the line behind marker dispatches for a counter pass, write MANIFEST")
    ap.add_argument("--vendor", default=os.environ.get("SPMV_BENCH_VENDOR", "wait"), choices=["wait", "if-ready", "off"],
                    help="rocSPARSE's best algorithm beside every workload of other_workloads: a child process loads "
                         "librocsparse.so (0.5 GB) and runs it once while this one times the CPU baseline; 'wait' (default) waits "
                         "for that child up to --vendor-wait seconds, 'if-ready' never waits, 'off' skips it")
    ap.add_argument("--vendor-wait", type=float, default=float(os.environ.get("SPMV_BENCH_VENDOR_WAIT", "200")),
                    help="'wait': give up on rocSPARSE this many seconds after its warm-up child was started (default 200; this process's "
                         "own first load of the library comes on top: 0-170 s by box, a thread with libc's dlopen was tried and "
                         "changes nothing -- the interpreter's other loads queue behind it)")
    ap.add_argument("--cpu-sample-rows", type=int, default=1 << 23)
    ap.add_argument("--rows-per-gpu", type=int, default=16 << 20, help="N>1: rows of each rank's block (default 16Mi)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("SPMV_BENCH_PIPELINE", "4")),
                    help="N>1: row blocks per rank (block-cyclic); the all-gather of one group overlaps the next multiply")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): 16Mi rows per rank, the matrix grows with N; strong: config 5's (128Mi)^2 matrix "
                         "for every N (--total-blocks row blocks dealt over the ranks)")
    ap.add_argument("--total-blocks", type=int, default=32, help="--scaling strong: row blocks of the fixed matrix")
    ap.add_argument("--exchange", default=os.environ.get("SPMV_BENCH_EXCHANGE", "all"), choices=["all", "allgather", "p2p"],
                    help="N>1: how y is concatenated.  'allgather' = RCCL's all-gather (the north-star; ALWAYS what `value` is "
                         "measured with unless 'p2p' is forced), 'p2p' = one direct send/recv pair per peer, 'all' (default) = "
                         "the headline with the all-gather, then the other exchanges timed back to back in the same run under "
                         "a watchdog: torch p2p, the C++ pipeline (libspmv_dist.so) with all-gather and p2p, and the "
                         "one-process peer-store pipeline (bin/spmv_dist_selftest) -> exchange_modes{...}")
    ap.add_argument("--modes-timeout", type=float, default=150.0, help="--exchange all: seconds the extra modes may take in total")
    ap.add_argument("--footprint", action="store_true",
                    help="N>1, --backend native --exchange p2p only: the optional footprint exchange of include/spmv_dist.h -- every "
                         "rank receives only the rows of y its own columns reference.  NOT the north-star's all-gather: the line says so")
    ap.add_argument("--backend", default=os.environ.get("SPMV_BENCH_BACKEND", "nccl"),
                    help="nccl (= RCCL through torch.distributed, default) | native (RCCL called from C++: libspmv_dist.so's "
                         "pipelined step, include/spmv_dist.h; torch.distributed/gloo only carries the id, the barrier and the "
                         "timing reduction; with fewer devices than ranks: all ranks in ONE process, peer stores) | gloo (rehearsal "
                         "of the N>1 path with ranks sharing one GPU)")
    return ap.parse_args()


# =====================================================================================================================
# the workloads of the line, shared by the timed process and the counter passes
# =====================================================================================================================

def extra_list(args):
    return [(c, b) for c, b in EXTRA_WORKLOADS if not (c == args.config and b == args.band)]


def wkey(cname, band):
    return f"{cname}:band{band}"


class Built:
    """One workload resident on the device: CSR arrays, handle, x, y."""

    def __init__(self, torch, capi, W, dev, cname, band):
        self.cname, self.band = cname, band
        if cname == "stencil7":
            n3, rp, ci, va = W.stencil7(200)
            self.rows, self.cols, self.nnz = n3, n3, len(ci)
            self.label = f"stencil7: 7-point stencil on 200^3 = {n3} unknowns, nnz {len(ci)} (host-built)"
            self.rp = rp
            self.d_rp, self.d_ci, self.d_va = (torch.from_numpy(a).to(dev) for a in (rp, ci, va))
            self.d_x = torch.empty(n3, dtype=torch.float32, device=dev)
            capi.synth_x(W.DEFAULT_SEED, 0, n3, self.d_x)
        else:
            if cname in ("c5", "c5shard"):
                # config 5's per-GPU shard: rows [0, 16Mi) of the (128Mi)^2 matrix, all 128Mi columns, the full
                # 512 MiB x (2 818 572 292 algorithmic bytes) -- what ONE MI355X of the 8 multiplies
                w = W.c5(8, band=band)
                n_loc = 16 << 20
                self.label = "c5 shard (rows [0,16Mi) of " + w.describe() + ")"
            else:
                w = W.config(cname, band=band)
                n_loc = w.rows
                self.label = w.describe()
            self.w = w
            self.rows, self.cols = n_loc, w.cols
            self.rp = W.row_ptr(w, 0, n_loc)
            self.nnz = int(self.rp[-1])
            self.d_rp = torch.from_numpy(self.rp).to(dev)
            self.d_ci = torch.empty(self.nnz, dtype=torch.int32, device=dev)
            self.d_va = torch.empty(self.nnz, dtype=torch.float32, device=dev)
            self.d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
            capi.synth_fill(w.seed, 0, n_loc, w.rows, w.cols, w.band, self.d_rp, self.d_ci, self.d_va)
            capi.synth_x(w.seed, 0, w.cols, self.d_x)
        self.d_y = torch.empty(self.rows, dtype=torch.float32, device=dev)
        self.A = capi.CsrMatrix.from_device(self.rows, self.cols, self.d_rp, self.d_ci, self.d_va)
        self.algorithmic = W.algorithmic_bytes(self.rows, self.cols, self.nnz)

    def close(self):
        self.A.close()
        self.d_rp = self.d_ci = self.d_va = self.d_x = self.d_y = None


def rank_blocks(args, W, world, rank):
    """(workload, rows per block, pipeline depth S, global ids of the blocks this rank owns) for an N > 1 / strong job."""
    strong = args.scaling == "strong"
    if strong:
        w = W.c5(8, band=args.band, rows_per_gpu=args.rows_per_gpu)
        if args.total_blocks % world or (w.rows // args.total_blocks) % W.BLOCK_ROWS:
            raise SystemExit("--total-blocks must be a multiple of N and cut the matrix at multiples of 65536 rows")
        S = args.total_blocks // world
        return w, w.rows // args.total_blocks, S, [s * world + rank for s in range(S)]
    S = max(1, args.pipeline)
    w = W.c5(world, band=args.band, rows_per_gpu=args.rows_per_gpu)
    if (args.rows_per_gpu // S) % W.BLOCK_ROWS:
        raise SystemExit("--rows-per-gpu / --pipeline must be a multiple of 65536 rows")
    return w, args.rows_per_gpu // S, S, [s * world + rank for s in range(S)]      # block-cyclic: group s is contiguous in y


# =====================================================================================================================
# HBM traffic measured in this run: rocprofv3 --pmc child passes
# =====================================================================================================================

def prefetch(paths):
    """posix_fadvise(WILLNEED): the kernel reads the files into the page cache in the background (no process, no wait).  A
    freshly booted box reads its image at a few MB/s per request: the profiler's libraries (libamd_comgr.so alone is 160 MB)
    are asked for before the first child needs them."""
    for f in paths:
        try:
            fd = os.open(f, os.O_RDONLY)
            try:
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_WILLNEED)
            finally:
                os.close(fd)
        except OSError:
            pass


PROFILER_FILES = ["/opt/rocm/lib/libamd_comgr.so", "/opt/rocm/lib/libamdhip64.so", "/opt/rocm/lib/libhsa-runtime64.so",
                  "/opt/rocm/lib/librocprofiler-sdk.so", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so",
                  "/opt/rocm/lib/librocprofiler-register.so", "/opt/rocm/lib/libhsa-amd-aqlprofile64.so",
                  str(ROOT / "spmv-test_amd" / "lib" / "libspmv_hip.so")]


def under_profiler():
    return "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)


def traffic_child(args):
    """--traffic-child MANIFEST: calibration kernels, then every workload of the line planned and launched TRAFFIC_RUNS
    times behind marker dispatches (spmv_calib_marker: the grid size is the id).  Run under `rocprofv3 --pmc <counter>`;
    the parent cuts the per-dispatch counter file at the markers."""
    t_child = time.perf_counter()
    import torch
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    world = int(os.environ.get("SPMV_TRAFFIC_WORLD", "1"))
    dev_index = int(os.environ.get("SPMV_TRAFFIC_DEVICE", "0"))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    variant = capi.VARIANTS[args.variant]
    manifest = {"runs": TRAFFIC_RUNS, "workloads": [],
                "calibration": {"1": {"what": "16-byte stream", "known_bytes": CAL_STREAM_BYTES},
                                "2": {"what": "4-byte gather, one word per distinct 128-byte line", "known_lines": CAL_TABLE_LINES},
                                "3": {"what": "4-byte gathers, a word in each 64-byte half of every line", "known_lines": CAL_TABLE_LINES},
                                "4": {"what": "4-byte-per-lane stores", "known_bytes": CAL_STORE_BYTES},
                                "5": {"what": "16-byte-per-lane stores", "known_bytes": CAL_STORE_BYTES}}}
    table = torch.zeros(CAL_TABLE_LINES * 32, dtype=torch.float32, device=dev)
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    for ident, fn in ((1, lambda: capi.calib_stream(table, CAL_STREAM_BYTES, sink)),
                      (2, lambda: capi.calib_gather(table, CAL_TABLE_LINES, CAL_TABLE_LINES, 1, sink)),
                      (3, lambda: capi.calib_gather(table, CAL_TABLE_LINES, CAL_TABLE_LINES, 2, sink)),
                      (4, lambda: capi.calib_store(table, CAL_STORE_BYTES, 4)),
                      (5, lambda: capi.calib_store(table, CAL_STORE_BYTES, 16))):
        capi.calib_marker(ident)
        for _ in range(TRAFFIC_RUNS):
            fn()
    capi.calib_marker(99)                                 # (end of the calibration regions)
    torch.cuda.synchronize()
    manifest["seconds"] = {"to_calibration_done": round(time.perf_counter() - t_child, 1)}
    del table
    torch.cuda.empty_cache()

    def one(ident, key, label, handles, run, plan_first=None):
        capi.calib_marker(100 + ident)                    # plan region
        if plan_first is not None:
            plan_first()
        capi.calib_marker(300 + ident)                    # run region
        for _ in range(TRAFFIC_RUNS):
            run()
        capi.calib_marker(500 + ident)                    # ... ends here (what follows builds the next workload)
        torch.cuda.synchronize()
        manifest["workloads"].append({"id": ident, "key": key, "label": label, "plan": handles[0].plan_describe(variant)})

    if world > 1 or args.scaling == "strong":
        # one rank's share of the N > 1 job: rank 0's blocks, planned alike, one launch each per step
        w, sub_rows, S, owned = rank_blocks(args, W, world, 0)
        handles, keep, params = [], [], None
        for b in owned:
            rp = W.row_ptr(w, b * sub_rows, sub_rows)
            d_rp = torch.from_numpy(rp).to(dev)
            d_ci = torch.empty(int(rp[-1]), dtype=torch.int32, device=dev)
            d_va = torch.empty(int(rp[-1]), dtype=torch.float32, device=dev)
            capi.synth_fill(w.seed, b * sub_rows, sub_rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
            handles.append(capi.CsrMatrix.from_device(sub_rows, w.cols, d_rp, d_ci, d_va))
            keep.append((d_rp, d_ci, d_va))
        d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
        capi.synth_x(w.seed, 0, w.cols, d_x)
        d_y = torch.empty(sub_rows * len(owned), dtype=torch.float32, device=dev)

        def plan_all():
            handles[0].plan(variant)
            p = handles[0].plan_params(variant)
            for h in handles[1:]:
                h.plan_set(variant, p)

        def run_all():
            for s_, h in enumerate(handles):
                h.run(variant, d_x, d_y[s_ * sub_rows:(s_ + 1) * sub_rows])
        one(0, "rank", w.describe(), handles, run_all, plan_all)
    else:
        todo = [(args.config, args.band)] + ([] if args.no_extras else extra_list(args))
        for ident, (cname, band) in enumerate(todo):
            B = Built(torch, capi, W, dev, cname, band)
            one(ident, wkey(cname, band), B.label, [B.A], lambda B=B: B.A.run(variant, B.d_x, B.d_y), lambda B=B: B.A.plan(variant))
            B.close()
            del B
            torch.cuda.empty_cache()
    manifest["seconds"]["total"] = round(time.perf_counter() - t_child, 1)
    Path(args.traffic_child).write_text(json.dumps(manifest))


PROBE = ROOT / "spmv-test_amd" / "bin" / "spmv_traffic_probe"


def write_probe_job(args, W, variant_id, world, device, tmp):
    """The job file of bin/spmv_traffic_probe (host/traffic_probe.cpp) for the workloads of this line: row pointers (numpy,
    this process: no GPU needed) as raw files, everything else as numbers.  Returns [(id, key, label)]."""
    lines = [f"V {variant_id} {TRAFFIC_RUNS} {device}"]
    listed = []

    def synth(ident, w, row0, n_loc):
        f = tmp / f"rp_{ident}_{row0}.bin"
        W.row_ptr(w, row0, n_loc).tofile(f)
        lines.append(f"W {ident} {w.seed} {row0} {n_loc} {w.rows} {w.cols} {w.band} {f}")
    if world > 1 or args.scaling == "strong":
        w, sub_rows, S, owned = rank_blocks(args, W, world, 0)
        for b in owned:
            synth(0, w, b * sub_rows, sub_rows)
        listed.append((0, "rank", w.describe()))
    else:
        todo = [(args.config, args.band)] + ([] if args.no_extras else extra_list(args))
        for ident, (cname, band) in enumerate(todo):
            if cname == "stencil7":
                n3, rp, ci, va = W.stencil7(200)
                files = []
                for nm, a in (("rp", rp), ("ci", ci), ("va", va)):
                    f = tmp / f"st_{nm}_{ident}.bin"
                    a.tofile(f)
                    files.append(str(f))
                lines.append(f"F {ident} {n3} {n3} {len(ci)} " + " ".join(files))
                listed.append((ident, wkey(cname, band), "stencil7"))
                continue
            if cname in ("c5", "c5shard"):
                w, n_loc = W.c5(8, band=band), 16 << 20
            else:
                w = W.config(cname, band=band)
                n_loc = w.rows
            synth(ident, w, 0, n_loc)
            listed.append((ident, wkey(cname, band), w.describe()))
    (tmp / "job.txt").write_text("\n".join(lines) + "\n")
    return listed


def cut_counter_file(directory, counter):
    """{marker id: {"kib": sum of Counter_Value, "kernels": {name: kib}, "dispatches": n}} of the newest counter file."""
    files = sorted(glob.glob(str(Path(directory) / "**" / "*_counter_collection.csv"), recursive=True),
                   key=lambda f: Path(f).stat().st_mtime)
    if not files:
        raise RuntimeError(f"no *_counter_collection.csv under {directory}")
    rows = [r for r in csv.DictReader(open(files[-1])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seg = defaultdict(lambda: {"kib": 0.0, "kernels": defaultdict(float), "dispatches": 0})
    cur = None
    for r in rows:
        name = r["Kernel_Name"]
        if "k_marker" in name:
            cur = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            continue
        if cur is None:
            continue
        v = float(r["Counter_Value"])
        seg[cur]["kib"] += v
        seg[cur]["kernels"][name] += v
        seg[cur]["dispatches"] += 1
    return seg


def measure_traffic(args, world=1, device=0):
    """The two counter passes.  Returns (dict or None, source text).  MUST run before this process's first GPU call.
    dict: {"calibration": {...}, "workloads": {key: {"hbm_bytes", "fetch_KiB", "write_KiB", "plan", "dominant_kernel", ...}}}"""
    if args.traffic != "measure":
        return None, f"--traffic {args.traffic}"
    if under_profiler():
        return None, "not measured: this process already runs under a profiler (tools/profile.sh takes the counters itself)"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        return None, "not measured: rocprofv3 not found"
    tmp = Path(tempfile.mkdtemp(prefix="spmv_traffic_", dir=os.environ.get("TMPDIR", "/tmp")))
    t0 = time.perf_counter()
    fwd = ["--variant", args.variant, "--config", args.config, "--band", str(args.band), "--gpus", str(args.gpus),
           "--rows-per-gpu", str(args.rows_per_gpu), "--pipeline", str(args.pipeline), "--scaling", args.scaling,
           "--total-blocks", str(args.total_blocks)] + (["--no-extras"] if args.no_extras else [])
    env = dict(os.environ, TMPDIR="/tmp", SPMV_TRAFFIC_WORLD=str(world), SPMV_TRAFFIC_DEVICE=str(device))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK",
              "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE"):
        env.pop(k, None)
    segs, manifest = {}, None
    try:
        listed = None
        if PROBE.exists() and os.environ.get("SPMV_BENCH_TRAFFIC_CHILD", "native") == "native":
            # the native child (no Python, no torch under the profiler): its job = the workloads as numbers + row-pointer files
            W = ge.load_package().workloads
            vid = {"scalar": 0, "wave": 1, "wave_pipe": 2, "vector": 3, "adaptive": 4, "tiled": 5, "panel": 6, "auto": 7}[args.variant]
            listed = write_probe_job(args, W, vid, world, device, tmp)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = tmp / counter.lower()
            man = tmp / f"manifest_{counter}.json"
            # the program itself directly after `--` (no env/bash hop: the profiler's preloaded library initialises the GPU)
            if listed is not None:
                cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", str(out), "--", str(PROBE), str(tmp / "job.txt"),
                       str(tmp / f"plans_{counter}.txt")]
            else:
                cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", str(out), "--", sys.executable, str(ROOT / "bench.py"),
                       "--traffic-child", str(man)] + fwd
            log = open(tmp / f"{counter}.log", "w")
            p = subprocess.Popen(cmd, cwd=str(tmp), env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                rc = p.wait(timeout=max(5.0, args.traffic_timeout - (time.perf_counter() - t0)))
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)          # the exact process group started above
                p.wait()
                return None, f"not measured: the counter passes exceeded their {args.traffic_timeout:.0f} s (stopped in the {counter} pass)"
            finally:
                log.close()
            if listed is not None and rc == 0 and (tmp / f"plans_{counter}.txt").exists():
                plans = dict(l.split("\t", 1) for l in (tmp / f"plans_{counter}.txt").read_text().splitlines() if "\t" in l)
                manifest = {"runs": TRAFFIC_RUNS, "child": "bin/spmv_traffic_probe",
                            "workloads": [{"id": i, "key": k, "label": lab, "plan": plans.get(str(i), "?")} for i, k, lab in listed]}
            elif rc != 0 or not man.exists():
                tail = (tmp / f"{counter}.log").read_text(errors="replace")[-300:].replace("\n", " | ")
                return None, f"not measured: rocprofv3 --pmc {counter} child rc={rc}: {tail}"
            else:
                manifest = json.loads(man.read_text())
            segs[counter] = cut_counter_file(out, counter)
        runs = manifest["runs"]
        fseg, wseg = segs["FETCH_SIZE"], segs["WRITE_SIZE"]

        def per_run(seg, ident):
            return seg[ident]["kib"] * 1024.0 / runs if ident in seg else 0.0
        tallied_stream = per_run(fseg, 1)
        read_factor = CAL_STREAM_BYTES / tallied_stream if tallied_stream > 0 else 2.0
        g1, g2 = per_run(fseg, 2) / CAL_TABLE_LINES, per_run(fseg, 3) / CAL_TABLE_LINES
        w4, w16 = per_run(wseg, 4), per_run(wseg, 5)
        write_factor = CAL_STORE_BYTES / w4 if w4 > 0 else 1.0
        cal = {"read_factor_16B_stream": round(read_factor, 4),
               "gather_tallied_bytes_per_line": round(g1, 2), "gather_both_halves_tallied_bytes_per_line": round(g2, 2),
               "gather_class": ("a missing 4-byte gather is ONE request for the whole 128-byte line (both halves arrive with it), "
                                "tallied like a stream's: the stream factor applies to gathers too"
                                if g1 > 0 and abs(g2 / g1 - 1.0) < 0.1 else
                                "a gather that touches both halves of a line is tallied twice: sector requests, counted as they are"),
               "gather_factor": round(128.0 / g1, 4) if g1 > 0 and abs(g2 / g1 - 1.0) < 0.1 else 1.0,
               "write_factor_4B_per_lane": round(write_factor, 4),
               "write_factor_16B_per_lane": round(CAL_STORE_BYTES / w16, 4) if w16 > 0 else None,
               "known": {"stream_bytes": CAL_STREAM_BYTES, "gather_lines": CAL_TABLE_LINES, "store_bytes": CAL_STORE_BYTES},
               "how": "spmv_calib_* kernels (csrc/kernels_calib.hip) in the same rocprofv3 passes as the workloads"}
        # one factor for all reads is only right while streams and gathers are tallied alike (they are on gfx950: see
        # gather_class); otherwise the line says so instead of pretending
        gather_same = abs(cal["gather_factor"] / read_factor - 1.0) < 0.05
        cal["one_read_factor_valid"] = bool(gather_same)
        wl = {}
        for m in manifest["workloads"]:
            i = m["id"]
            f_b, w_b = per_run(fseg, 300 + i), per_run(wseg, 300 + i)
            kern = fseg[300 + i]["kernels"] if 300 + i in fseg else {}
            dom = max(kern, key=kern.get) if kern else None
            wl[m["key"]] = {"hbm_bytes": int(round(read_factor * f_b + write_factor * w_b)),
                            "FETCH_SIZE_KiB": round(f_b / 1024.0, 1), "WRITE_SIZE_KiB": round(w_b / 1024.0, 1),
                            "dispatches_per_spmv": (fseg[300 + i]["dispatches"] // runs) if 300 + i in fseg else 0,
                            "dominant_kernel": dom, "plan": m["plan"],
                            "plan_build_hbm_bytes": int(round(read_factor * per_run(fseg, 100 + i) * runs
                                                              + write_factor * per_run(wseg, 100 + i) * runs))}
        res = {"calibration": cal, "workloads": wl, "seconds": round(time.perf_counter() - t0, 1),
               "child_seconds": manifest.get("seconds")}
        who = ("bin/spmv_traffic_probe (the workloads of this line through libspmv_hip.so, no Python under the profiler)"
               if listed is not None else "this file (--traffic-child)")
        return res, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of {who}, "
                     f"{runs} launches per workload behind marker dispatches, corrected with the factors the "
                     "spmv_calib_* kernels gave in the same passes")
    except Exception as ex:                       # a measurement aid must never cost the line
        return None, f"not measured: {type(ex).__name__}: {str(ex)[:200]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def replayed_traffic(key, plan_now):
    """The committed figure of profiles/traffic.json, only when it was taken with the very plan this run uses."""
    tfile = ROOT / "profiles" / "traffic.json"
    try:
        ent = json.loads(tfile.read_text()).get(key, {})
    except Exception:
        return None, None
    if not ent:
        return None, None
    if ent.get("plan") is not None and ent["plan"].split(": ")[-1] == plan_now.split(": ")[-1]:
        return ent.get("hbm_bytes_per_launch"), f"replayed from profiles/{ent.get('profile')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same plan)"
    return None, f"profiles/{ent.get('profile')} was taken with another plan ({ent.get('plan')}): not replayed"


# =====================================================================================================================
# N > 1 without a launcher: this process starts its own ranks
# =====================================================================================================================

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python3 bench.py --gpus N` started directly: one child process per rank (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in its environment), BEFORE this process makes any GPU call; rank 0's JSON line passes through on the
    inherited stdout; non-zero if any rank fails.  (The reference drives everything from one process,
    test/main.cpp:3-7; so does this entry.)"""
    import torch
    n = args.gpus
    ndev = torch.cuda.device_count()              # (does not initialise the GPU on this image)
    if args.backend == "native" and ndev < n:
        return native_local(args, n, ndev)
    traffic_file = None
    if args.traffic == "measure":
        res, src = measure_traffic(args, world=n, device=0)
        fd, traffic_file = tempfile.mkstemp(prefix="spmv_traffic_", suffix=".json")
        os.write(fd, json.dumps({"result": res, "source": src}).encode())
        os.close(fd)
    port = free_port()
    procs = []
    argv = [sys.executable, str(ROOT / "bench.py")] + sys.argv[1:]
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPMV_BENCH_SELF_LAUNCHED="1")
        if traffic_file:
            env["SPMV_BENCH_TRAFFIC_JSON"] = traffic_file
        procs.append(subprocess.Popen(argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:             # one rank failed: the others would wait for it forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if traffic_file:
            try:
                os.unlink(traffic_file)
            except OSError:
                pass
    return rc


def selftest_line(n, per, S, band, steps, exchange, local, verify):
    """bin/spmv_dist_selftest (all ranks in ONE process over include/spmv_dist.h) as a child; its JSON line as a dict."""
    capi_path = ROOT / "spmv-test_amd" / "bin" / "spmv_dist_selftest"
    cmd = [str(capi_path), "--ranks", str(n), "--rows-per-rank", str(per), "--pipeline", str(S), "--exchange", exchange,
           "--band", str(band), "--steps", str(steps)] + (["--local"] if local else []) + ([] if verify else ["--no-verify"])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        raise RuntimeError(f"spmv_dist_selftest rc={p.returncode}: {(p.stderr or p.stdout)[-300:]}")
    return json.loads(line[-1])


def native_local(args, n, ndev):
    """--backend native with fewer devices than ranks: RCCL refuses two ranks on one device, so the C++ pipeline runs with
    all ranks in ONE process (spmv_dist_init_local, peer stores) -- a rehearsal of the plumbing, the line says so."""
    per = min(args.rows_per_gpu, 1 << 20)
    S = max(1, args.pipeline)
    per = max(per // (S << 16), 1) * (S << 16)
    try:
        j = selftest_line(n, per, S, args.band, max(args.steps, 1), "peer", True, True)
    except Exception as ex:
        print(f"bench.py --backend native (local ranks): {ex}", file=sys.stderr)
        return 1
    bytes_all = 8.0 * j["nnz"] + 4.0 * (j["rows"] + n * S) + 4.0 * j["rows"] + 4.0 * j["rows"] * n
    out = {"metric": "fp32 CSR SpMV throughput (CSR-algorithmic bytes per second: no counter pass in this rehearsal)",
           "value": round(bytes_all / j["step_ms"] / 1e6, 2), "unit": "GB/s", "n_gpus": n, "steps": max(args.steps, 1),
           "warmup": 3, "ms_per_step": j["step_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"({n} x {per})^2, 16 nonzeros per row, band {args.band} (bin/spmv_dist_selftest's matrix)",
                      "variant": j["variant"], "parallelism": f"{S} block-cyclic row blocks per rank x{n} ranks, peer stores "
                      "(hipMemcpyPeerAsync) of group s under the multiply of block s+1, libspmv_dist.so: spmv_dist_pipe_step",
                      "devices": f"{n} ranks on {max(ndev, 1)} device(s) -- RANKS SHARE DEVICES: a rehearsal of the plumbing, not a "
                                 "scaling measurement (RCCL refuses two ranks on one device: all ranks in ONE process, no communicator)"},
           "multiply_only_ms": j["multiply_only_ms"], "exchange_only_ms": j["exchange_only_ms"],
           "rows_differing_from_single_handle": j["rows_differing_from_single_handle"], "selftest": j}
    print(json.dumps(out), flush=True)
    return 0


# =====================================================================================================================
# what the line says about a kernel time
# =====================================================================================================================

def touched_x_bytes(torch, d_ci, cols):
    """4 bytes x the DISTINCT columns the matrix references (the CSR-algorithmic count charges all of x: a shard of
    config 5 with banded columns touches 64 MiB of its 512 MiB)."""
    mask = torch.zeros(cols, dtype=torch.bool, device=d_ci.device)
    step = 1 << 25
    for k in range(0, d_ci.numel(), step):
        mask[d_ci[k:k + step].long()] = True
    return 4 * int(mask.sum().item())


def rate_fields(prefix, hbm_bytes, algorithmic, ms):
    """frac fields of one kernel time: `<prefix>frac_hbm` from the measured traffic (None without it), `<prefix>effective_frac`
    from the CSR-algorithmic bytes, and the flags that keep either from being misread."""
    out = {}
    eff = algorithmic / ms / 1e6
    out[prefix + "effective_frac"] = round(eff / HBM_PEAK_GBS, 4)
    if hbm_bytes is not None:
        moved = hbm_bytes / ms / 1e6
        out[prefix + "frac_hbm"] = round(moved / HBM_PEAK_GBS, 4)
        out["hbm_bytes"] = hbm_bytes
        out["traffic_over_algorithmic"] = round(hbm_bytes / algorithmic, 4)
        if moved > COPY_CEILING_GBS:
            out["exceeds_copy_ceiling"] = (f"{moved:.0f} GB/s of counted traffic is above the {COPY_CEILING_GBS:.0f} GB/s a copy "
                                           "reaches: FETCH_SIZE counts L2 <-> fabric requests, those the 256 MiB Infinity Cache "
                                           "serves included (x re-read from it never reaches HBM), and a read-only stream runs "
                                           "above the copy rate (7.0 TB/s measured: profiles/r04_calibration.json)")
    if eff > COPY_CEILING_GBS:
        out["effective_exceeds_copy_ceiling"] = (f"algorithmic bytes / time = {eff:.0f} GB/s is above the {COPY_CEILING_GBS:.0f} GB/s "
                                                 "a copy reaches: the kernel does not move that many bytes (x columns never "
                                                 "referenced, 16-bit column copies) -- read frac_hbm")
    return out


def l2_roofline(plan_text, nnz, ms):
    """The panel family is bound by L2 line requests, not by HBM bytes: a second roofline with that ceiling."""
    if "binned" in plan_text or not ("auto -> panel" in plan_text or plan_text.startswith(("panel_columns", "sorted_blocks"))):
        return {}          # (the binned layout gathers nothing from memory: two streaming launches, HBM-bound)
    if "lines_per_nonzero=" in plan_text:
        lines = float(plan_text.split("lines_per_nonzero=")[1].split()[0]) * nnz
        what = "lines_per_nonzero of the plan x nonzeros (distinct 128-byte lines of x per block, each requested about once)"
    else:
        lines = float(nnz)
        what = "one line request per nonzero (rows ascend inside a tile: the lanes of an instruction hold different lines)"
    return {"roofline_l2_gather": {"bound": "l2_gather", "achieved": round(lines / ms / 1e6, 1), "peak": L2_LINE_PEAK_G,
                                   "unit": "G line requests/s", "frac": round(lines / ms / 1e6 / L2_LINE_PEAK_G, 4),
                                   "requests": what,
                                   "peak_from": "tools/ubench_gather_lines.hip: 64 lanes on 64 distinct L2-resident lines"}}


def dominant_kernel(resolved, plan_now):
    def _n(key):
        return int(plan_now.split(key + "=")[1].split()[0]) if key + "=" in plan_now else 0
    if resolved == "panel":
        return "k_bs_products + k_bs_sums" if "scattered_products" in plan_now else "k_bin_products + k_bin_sums" if "binned" in plan_now else "k_colsort" if "sorted_blocks=" in plan_now else "k_panel"
    if resolved == "tiled" and (_n("col16_chunks") or _n("sorted_chunks")):
        # one launch: all chunks 16-bit -> k_tiled16, all sorted -> k_sorted, otherwise the three bodies in k_tiled_mixed
        return ("k_tiled16" if _n("col16_chunks") == _n("chunks") else
                "k_sorted" if _n("sorted_chunks") == _n("chunks") else "k_tiled_mixed")
    return ("k_adaptive" if resolved in ("adaptive", "tiled") else
            "k_wave_bundle" if "block_rows=" in plan_now else f"k_{resolved}")


# =====================================================================================================================
# one rank
# =====================================================================================================================

def main():
    t_main = time.perf_counter()
    args = parse()
    if args.traffic_child:
        traffic_child(args)
        return 0
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not launched and args.gpus > 1:
        return self_launch(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    strong = args.scaling == "strong"
    native = args.backend == "native"
    single = world == 1 and not strong and not native       # the N = 1 line with extras, CPU baseline, vendor

    # ---- before the first GPU call: the counter passes (rank 0), the vendor library into the page cache ------------
    vendor_cat = None            # the warm-up child of the vendor library (started after the headline is measured)
    traffic, traffic_source = None, None
    if rank == 0 and args.traffic == "measure" and not under_profiler():
        prefetch(PROFILER_FILES)
    if rank == 0:
        pre = os.environ.get("SPMV_BENCH_TRAFFIC_JSON")
        if pre and Path(pre).exists():                       # self_launch measured it before it started the ranks
            j = json.loads(Path(pre).read_text())
            traffic, traffic_source = j["result"], j["source"]
        else:
            traffic, traffic_source = measure_traffic(args, world=world if (world > 1 or strong) else 1, device=local_rank)
        print(f"[bench] traffic passes done at {time.perf_counter() - t_main:.1f} s: {traffic_source[:90]}", file=sys.stderr, flush=True)
    # The vendor library is MAPPED now, while this process has no HIP context yet: a library that registers its code objects
    # with a live context has all of them loaded at once -- 156-180 s of a freshly booted box for librocsparse.so's 0.5 GB,
    # spent inside the first comparison -- while one registered before the context exists loads what is launched, when it is
    vendor_early = None
    if single and rank == 0 and args.vendor != "off" and not args.no_extras:
        try:
            import ctypes
            t_e = time.perf_counter()
            vendor_early = ctypes.CDLL("/opt/rocm/lib/librocsparse.so")
            print(f"[bench] librocsparse.so mapped at {time.perf_counter() - t_main:.1f} s ({time.perf_counter() - t_e:.1f} s)", file=sys.stderr, flush=True)
        except OSError as ex:
            print(f"[bench] librocsparse.so: {ex}", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: libspmv_hip has no CPU path")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend in ("nccl", "native") else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    shared = not (ndev >= world and args.backend in ("nccl", "native"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo" if native else args.backend, rank=rank, world_size=world)
    variant = capi.VARIANTS[args.variant]
    ctl_dev = dev if args.backend == "nccl" else "cpu"       # where the few control numbers of the job travel

    # ---- the workload: this rank's row block(s), generated on the device ---------------------------
    t_setup = time.perf_counter()
    if single:
        S = 1
        if args.config == "c5shard":
            w, sub_rows, owned = W.c5(8, band=args.band), 16 << 20, [0]
        else:
            w = W.config(args.config, band=args.band)
            sub_rows, owned = w.rows, [0]
    else:
        w, sub_rows, S, owned = rank_blocks(args, W, world, rank)
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
                t = torch.tensor(plan_params, dtype=torch.int32, device=ctl_dev)
                dist.broadcast(t, src=0)
                agreed = [int(v) for v in t.cpu().tolist()]
                if agreed != plan_params:
                    A.plan_set(variant, agreed)
                    plan_params = agreed
        handles.append(A)
        keep.append((rp, d_rp, d_ci, d_va))
        nnz_local += nb
    rows_local = sub_rows * len(owned)

    def bind(h):
        return lambda x, y: h.run(variant, x, y)

    def make_pipeline(backend_native, exchange):
        """The N > 1 step: (object with x, y_full, step, finish, exchange_only, block_rows) for one exchange."""
        if backend_native:
            # RCCL from C++ (include/spmv_dist.h): rank 0 makes the 128-byte id, the process group hands it round
            idt = torch.zeros(pkg.dist_native.ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                idt = torch.frombuffer(bytearray(pkg.dist_native.unique_id()), dtype=torch.uint8).clone()
            if world > 1:
                idt = idt.to(ctl_dev)
                dist.broadcast(idt, src=0)
                idt = idt.cpu()
            sh = pkg.dist_native.NativePipeline(world, rank, bytes(idt.numpy().tobytes()), S, sub_rows, w.cols, handles, variant, dev,
                                                exchange=exchange)
            if args.footprint:
                if exchange != "p2p":
                    raise SystemExit("--footprint needs --exchange p2p")
                mine = torch.tensor([[lo if hi >= 0 else 0, hi + 1 if hi >= 0 else 0] for lo, hi in (h.column_range() for h in handles)],
                                    dtype=torch.int64, device=ctl_dev)
                every = [torch.zeros_like(mine) for _ in range(world)]
                if world > 1:
                    dist.all_gather(every, mine)
                else:
                    every = [mine]
                sh.set_footprint([[int(v) for v in t[:, 0]] for t in every], [[int(v) for v in t[:, 1]] for t in every])
            return sh
        return pkg.dist.PipelinedSpmv(S, sub_rows, w.cols, [bind(h) for h in handles], dev, exchange=exchange)

    exchange_only = None
    first_exchange = "p2p" if args.exchange == "p2p" else "allgather"     # 'all': the headline is the north-star's all-gather
    if single:
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
        sh = None
    else:
        sh = make_pipeline(native, first_exchange)
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

    def timed_steps(step_fn, finish_fn):
        """W warm-up steps, then EXACTLY K timed steps between barrier + synchronize; (max-over-ranks seconds, event ms per step)."""
        for _ in range(args.warmup):
            step_fn()
        finish_fn()
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step_fn()
        finish_fn()                               # every all-gather of the K steps has landed
        ev1.record()
        torch.cuda.synchronize(); barrier()
        el = time.perf_counter() - t0
        ms_ev = ev0.elapsed_time(ev1) / args.steps            # HIP events on the launch stream
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ms_ev

    elapsed, step_ms_events = timed_steps(step, finish)

    # per step, per rank: vals+col_idx, row_ptr of every owned block, y written, x read once
    bytes_rank = 8 * nnz_local + 4 * (rows_local + len(owned)) + 4 * rows_local + 4 * w.cols
    gflops = 2.0 * nnz_local * world * args.steps / elapsed / 1e9

    # ---- the kernel alone (no collective): mean launch time by HIP events on its stream -----------
    iters = max(10, args.steps)

    def timed_exchange(xo, fin):
        torch.cuda.synchronize(); barrier()
        ev0.record()
        for _ in range(iters):
            xo()
        fin()
        ev1.record()
        torch.cuda.synchronize(); barrier()
        return ev0.elapsed_time(ev1) / iters
    exchange_ms = None
    if single:
        kernel_ms = A.time(variant, d_x, d_y, iters)          # spmv_csr_time: events inside the library
    else:
        ev0.record()
        for _ in range(iters):
            multiply_only()
        ev1.record()
        torch.cuda.synchronize()
        kernel_ms = ev0.elapsed_time(ev1) / iters
        if world > 1:                                          # the all-gathers of one step alone, no multiply
            exchange_ms = timed_exchange(exchange_only, finish)

    out = None
    if rank == 0:
        plan_now = handles[0].plan_describe(variant)
        resolved = capi.lib().spmv_variant_name(handles[0].plan_params(variant)[0]).decode()
        # ---- traffic of one SpMV of this rank -------------------------------------------------------------------
        hbm_bytes, head_src, tinfo = None, traffic_source, None
        key = wkey(args.config, args.band) if single else "rank"
        if traffic is not None and key in traffic["workloads"]:
            tinfo = traffic["workloads"][key]
            if tinfo["plan"].split(": ")[-1] == plan_now.split(": ")[-1]:
                hbm_bytes = tinfo["hbm_bytes"]
            else:
                head_src = f"measured under another plan ({tinfo['plan']}) than this process made ({plan_now}): not used"
        if hbm_bytes is None and single:
            rb, rs = replayed_traffic(f"{resolved}:{w.name}:band{w.band}", plan_now)
            if rb is not None:
                hbm_bytes, head_src = rb, f"{rs}; in-run measurement: {traffic_source}"
            elif rs is not None:
                head_src = f"{traffic_source}; {rs}"
        basis = hbm_bytes if hbm_bytes is not None else bytes_rank
        value = basis * world * args.steps / elapsed / 1e9
        moved = basis / (kernel_ms * 1e-3) / 1e9
        eff = bytes_rank / (kernel_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(moved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(moved / HBM_PEAK_GBS, 4), "traffic": hbm_bytes, "traffic_source": head_src,
                "frac_basis": ("FETCH_SIZE + WRITE_SIZE bytes of one SpMV (corrected) / kernel_ms / 8 TB/s" if hbm_bytes is not None else
                               "CSR-ALGORITHMIC bytes / kernel_ms / 8 TB/s -- no counter figure in this run (see traffic_source): "
                               "may exceed what the chip can move"),
                "effective_achieved": round(eff, 2), "effective_frac": round(eff / HBM_PEAK_GBS, 4),
                "effective_definition": "CSR-algorithmic bytes of one SpMV (8/nnz + row_ptr + x + y) / kernel_ms: the work done, not "
                                        "the bytes moved (16-bit column copies and x never referenced shrink the traffic)",
                "traffic_over_algorithmic": None if hbm_bytes is None else round(hbm_bytes / bytes_rank, 4),
                "frac_of_copy_ceiling": round(moved / COPY_CEILING_GBS, 4),
                "kernel": dominant_kernel(resolved, plan_now),
                "kernel_ms": round(kernel_ms, 5), "timing": "HIP events on the launch stream"}
        if tinfo is not None:
            roof["counters"] = {k: tinfo[k] for k in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "dispatches_per_spmv", "dominant_kernel")}
        if traffic is not None:
            roof["calibration"] = traffic["calibration"]
            roof["traffic_passes_s"] = traffic["seconds"]
            roof["traffic_child_s"] = traffic.get("child_seconds")
        for flag_rate, name in ((moved if hbm_bytes is not None else 0.0, "exceeds_copy_ceiling"), (eff, "effective_exceeds_copy_ceiling")):
            if flag_rate > COPY_CEILING_GBS:
                roof[name] = (f"{flag_rate:.0f} GB/s is above the {COPY_CEILING_GBS:.0f} GB/s a copy reaches: "
                              + ("the kernel does not move the algorithmic byte count (x columns never referenced, 16-bit column "
                                 "copies) -- read frac" if name.startswith("effective") else
                                 "FETCH_SIZE counts L2 <-> fabric requests, Infinity Cache hits included"))
        out = {
            "metric": ("fp32 CSR SpMV: achieved HBM GB/s (rocprofv3 FETCH_SIZE + WRITE_SIZE bytes of the K steps / wall time)"
                       if hbm_bytes is not None else
                       "fp32 CSR SpMV throughput in CSR-algorithmic bytes per second (no counter figure in this run: see roofline.traffic_source)"),
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": w.describe() + (" -- the per-GPU shard: rows [0,16Mi), all columns, the full x"
                                                    if single and args.config == "c5shard" else ""),
                       "variant": args.variant,
                       "rows_per_gpu": rows_local, "nnz_per_gpu": nnz_local,
                       "parallelism": "single GPU" if single else
                       f"{S} block-cyclic row blocks per rank x{world} ranks, all-gather(y) of group s "
                       f"({'RCCL all_gather' if first_exchange == 'allgather' else 'direct send/recv per peer'}) "
                       f"overlapped with the multiply of block s+1, "
                       f"{'RCCL called from C++ (libspmv_dist.so: spmv_dist_pipe_step)' if native else args.backend}"
                       + (" -- FOOTPRINT exchange: every rank receives only the rows of y its columns reference, not the "
                          "all-gather of the north-star" if args.footprint else ""),
                       "devices": (f"{world} ranks on {min(world, max(ndev, 1))} device(s)" +
                                   ("" if not shared else
                                    " -- RANKS SHARE DEVICES: a rehearsal of the plumbing, not a scaling measurement")),
                       "launch": ("started by a launcher (WORLD_SIZE set)" if launched and not os.environ.get("SPMV_BENCH_SELF_LAUNCHED")
                                  else "bench.py started its own ranks" if launched else "one process"),
                       "algorithmic_bytes_per_gpu": bytes_rank, "hbm_bytes_per_gpu": hbm_bytes},
            "pct_of_hbm_peak": round(100.0 * value / world / HBM_PEAK_GBS, 2),
            "effective_value": round(bytes_rank * world * args.steps / elapsed / 1e9, 2),
            "effective_value_definition": "CSR-algorithmic bytes of all ranks x K / wall time",
            "gflops": round(gflops, 1),
            "roofline": roof,
            "step_ms_events": round(step_ms_events, 5), "multiply_only_ms": round(kernel_ms, 5),
            "exchange_only_ms": None if exchange_ms is None else round(exchange_ms, 5),
            "plan_ms": round(plan_ms, 4), "plan_bytes": sum(h.plan_bytes(variant) for h in handles),
            "plan_amortised_over": round(plan_ms / kernel_ms, 1),
            "plan": plan_now,
            "setup_s": round(setup_s, 2),
        }
        print(f"[bench] headline measured at {time.perf_counter() - t_main:.1f} s", file=sys.stderr, flush=True)
        if single and args.vendor != "off" and not args.no_extras:
            # rocSPARSE for the comparison fields: a child loads the library and runs it once on a tiny matrix while this process
            # is busy on the CPU (the baseline below) -- what a load touches of the 0.5 GB file is then in the page cache and
            # the dlopen here is quick; no GPU work of this process is timed while the child runs
            vendor_cat = subprocess.Popen([sys.executable, str(ROOT / "tools" / "vendor_compare.py"), "--warm"],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t_vendor = time.perf_counter()

    # ---- N > 1, --exchange all: the other exchanges back to back, under a watchdog ---------------------------------
    if world > 1 and args.exchange == "all" and not args.footprint:
        modes = {}
        if rank == 0:
            modes[("native " if native else "torch ") + "allgather"] = {
                "step_ms": round(elapsed / args.steps * 1e3, 5), "exchange_only_ms": None if exchange_ms is None else round(exchange_ms, 5),
                "multiply_only_ms": round(kernel_ms, 5), "this_is": "the headline (value, ms_per_step)"}
        deadline = time.perf_counter() + args.modes_timeout
        done = threading.Event()

        def watchdog():
            # an exchange that has never run at this world size may hang inside RCCL: every rank carries the same deadline;
            # rank 0 prints the line it has, everyone leaves with status 0 (no rank is killed by another)
            while not done.wait(0.5):
                if time.perf_counter() > deadline:
                    if rank == 0:
                        out["exchange_modes"] = dict(modes, watchdog=f"a mode did not finish within {args.modes_timeout:.0f} s: the "
                                                     "line was printed without it and the ranks left")
                        print(json.dumps(out), flush=True)
                    os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        legs = [(native, "p2p")] + ([] if shared else [(not native, "allgather"), (not native, "p2p")])
        for be_native, xch in legs:
            name = ("native " if be_native else "torch ") + xch
            try:
                p2 = make_pipeline(be_native, xch)
                p2.x.copy_(d_x)
                el2, _ = timed_steps(p2.step, p2.finish)
                x_ms = timed_exchange(p2.exchange_only, p2.finish)
                same = bool(torch.equal(p2.y_full, sh.y_full))
                if rank == 0:
                    modes[name] = {"step_ms": round(el2 / args.steps * 1e3, 5), "exchange_only_ms": round(x_ms, 5),
                                   "multiply_only_ms": round(kernel_ms, 5), "y_bit_identical_to_headline": same}
                if hasattr(p2, "close"):
                    p2.close()
                del p2
            except Exception as ex:
                if rank == 0:
                    modes[name] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
        barrier()
        if rank == 0:
            # peer stores need every rank in ONE process: the C++ selftest binary, same sizes, its own 16-per-row matrix
            try:
                j = selftest_line(world, args.rows_per_gpu, S, args.band, min(args.steps, 20), "peer", shared, False)
                modes["native peer_store (one process, bin/spmv_dist_selftest)"] = {
                    "step_ms": j["step_ms"], "exchange_only_ms": j["exchange_only_ms"], "multiply_only_ms": j["multiply_only_ms"],
                    "matrix": "16 nonzeros per row (the selftest's own law), same rows per rank and band", "process_model": j["process_model"]}
            except Exception as ex:
                modes["native peer_store (one process, bin/spmv_dist_selftest)"] = {"error": str(ex)[:300]}
            out["exchange_modes"] = modes
        done.set()

    # ---- CPU baseline: rank 0, N = 1 only, bounded sample of the same matrix ------------------------
    if rank == 0 and single and not args.no_cpu_baseline:
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
                                         f"{reps} passes, {t_cpu:.1f} s, oracle_spmv_csr_mt; CSR-algorithmic bytes of the sample / time"}
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
        print(f"[bench] cpu baseline done at {time.perf_counter() - t_main:.1f} s", file=sys.stderr, flush=True)

    # ---- the other column laws and configs, kernel time only (N = 1) ------------------------------
    if rank == 0 and single and not args.no_extras:
        del A, d_rp, d_ci, d_va, d_x, d_y, handles, keep
        torch.cuda.empty_cache()
        extras = []
        # rocSPARSE beside every workload (the reference's vendor slot, cublas.cu:33, is a comparison there too)
        vendor_box = {}
        rocs, vendor = None, None

        def vendor_ready():
            """The rocSPARSE handle, created on the thread that owns the stream, once the library file is in the page cache."""
            nonlocal rocs, vendor
            if rocs is not None or args.vendor == "off" or "error" in vendor_box:
                return rocs
            if vendor_cat is not None and vendor_cat.poll() is None:
                if args.vendor == "if-ready":
                    return None
                left = args.vendor_wait - (time.perf_counter() - t_vendor)
                try:
                    vendor_cat.wait(timeout=max(left, 0.0))
                except subprocess.TimeoutExpired:
                    vendor_box["error"] = (f"the warm-up child had not loaded librocsparse.so within {args.vendor_wait:.0f} s "
                                           "(a freshly booted box reads its image at a few MB/s)")
                    return None
            if vendor_cat is not None and vendor_cat.returncode not in (0, None):
                vendor_box["error"] = f"the warm-up child of librocsparse.so failed (rc={vendor_cat.returncode})"
                return None
            try:
                import ctypes
                import importlib.util
                t_l = time.perf_counter()
                spec = importlib.util.spec_from_file_location("vendor_compare", ROOT / "tools" / "vendor_compare.py")
                vendor = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(vendor)
                rocs = vendor.RocSparse()
                rocs._ok(rocs.L.rocsparse_set_stream(rocs.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")
                vendor_box["loaded_s"] = round(time.perf_counter() - t_l, 1)
            except Exception as ex:
                vendor_box["error"] = f"{type(ex).__name__}: {str(ex)[:100]}"
                rocs = None
            return rocs

        def vendor_fields(B):
            if vendor_ready() is None:
                return {}
            try:
                # (the two algorithms that were rocSPARSE's best on every workload of rounds 1 and 2; all four are in
                # tools/vendor_compare.py -> profiles/r02_vendor_compare.jsonl)
                rt = vendor.rocsparse_times(rocs, B.rows, B.cols, B.nnz, B.d_rp, B.d_ci, B.d_va, B.d_x, B.d_y, iters=10,
                                            algs={k: vendor.ALGS[k] for k in ("csr_adaptive", "csr_nnzsplit")})
                if rt:
                    ba = min(rt, key=lambda a: rt[a][0])
                    return {"rocsparse_best_ms": round(rt[ba][0], 5), "rocsparse_best_algorithm": ba,
                            "rocsparse_preprocess_ms": round(rt[ba][1], 3)}
            except Exception as ex:                      # a comparison, never a reason to lose the line
                return {"rocsparse_error": str(ex)[:120]}
            return {}

        for cname, band in extra_list(args):
            t_w = time.perf_counter()
            B = Built(torch, capi, W, dev, cname, band)
            be = B.algorithmic
            names = ("adaptive", "tiled") + (("panel",) if band == 0 and cname != "stencil7" else ()) + ("auto",)
            # the variants take turns, three rounds of 20 launches each after a warm round, and keep their best: every one
            # meets the clocks and caches in the same states
            times, plan_cost = {}, {}
            for vn in names:
                ev0.record()
                B.A.plan(capi.VARIANTS[vn])
                ev1.record()
                torch.cuda.synchronize()
                plan_cost[vn] = ev0.elapsed_time(ev1)
                B.A.time(capi.VARIANTS[vn], B.d_x, B.d_y, 10)
            for _round in range(3):
                for vn in names:
                    ms = B.A.time(capi.VARIANTS[vn], B.d_x, B.d_y, 20)
                    times[vn] = min(times.get(vn, ms), ms)
            best = min(times.items(), key=lambda kv: kv[1])
            auto_plan, auto_ms = B.A.plan_describe(capi.VARIANTS["auto"]), times["auto"]
            resolved = auto_plan.split("auto -> ")[1].split(":")[0] if "auto -> " in auto_plan else "auto"
            # AUTO planned after TILED finds TILED's plan made: its own cost = the TILED plan + what it added
            auto_plan_ms = plan_cost["auto"] + plan_cost.get("tiled", 0.0)
            # the kernel BASELINE.json's config string names for this config, timed beside the library's choice
            named_out = {}
            for vn in {"c2": ("scalar",), "c3": ("wave", "wave_pipe")}.get(cname, ()):
                v = capi.VARIANTS[vn]
                B.A.plan(v)
                B.A.time(v, B.d_x, B.d_y, 10)
                ms = min(B.A.time(v, B.d_x, B.d_y, 10) for _ in range(2))
                named_out[vn] = {"kernel_ms": round(ms, 5), "effective_frac": round(be / ms / 1e6 / HBM_PEAK_GBS, 4)}
            hb, hsrc = None, None
            tkey = wkey(cname, band)
            if traffic is not None and tkey in traffic["workloads"]:
                ti = traffic["workloads"][tkey]
                if ti["plan"].split(": ", 1)[-1] == auto_plan.split(": ", 1)[-1]:
                    hb, hsrc = ti["hbm_bytes"], "measured in this run"
                else:
                    hsrc = f"measured under another plan ({ti['plan']}): not used"
            if hb is None and cname != "stencil7":
                tk = f"{resolved}:{'c5x8' if cname == 'c5' else cname}:band{band}"
                rb, rs = replayed_traffic(tk, auto_plan)
                if rb is not None:
                    hb, hsrc = rb, rs
            tx = touched_x_bytes(torch, B.d_ci, B.cols)
            touched = 8 * B.nnz + 4 * (B.rows + 1) + 4 * B.rows + tx
            entry = {"workload": B.label, "auto": auto_plan.split(":")[0], "auto_plan": auto_plan,
                     "auto_kernel_ms": round(auto_ms, 5), **rate_fields("auto_", hb, be, auto_ms),
                     "traffic_source": hsrc,
                     "best_variant": best[0], "best_kernel_ms": round(best[1], 5),
                     "best_effective_frac": round(be / best[1] / 1e6 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes": be, "bytes_touched": touched,
                     "plan_ms": round(auto_plan_ms, 3), "plan_bytes": B.A.plan_bytes(capi.VARIANTS["auto"]),
                     "plan_amortised_over": round(auto_plan_ms / auto_ms, 1),
                     **l2_roofline(auto_plan, B.nnz, auto_ms),
                     **({"config_named_kernel": named_out} if named_out else {})}
            if resolved == "tiled" and resolved in times:
                entry["resolved_variant_kernel_ms"] = round(times[resolved], 5)   # the same plan timed under its own name
            vf = vendor_fields(B)
            if "rocsparse_best_ms" in vf:
                vf["speedup_vs_rocsparse_best"] = round(vf["rocsparse_best_ms"] / auto_ms, 2)
            entry.update(vf)
            extras.append(entry)
            print(f"[bench] {cname} band {band}: {time.perf_counter() - t_w:.1f} s", file=sys.stderr, flush=True)
            B.close()
            del B
            torch.cuda.empty_cache()
        print(f"[bench] extras done at {time.perf_counter() - t_main:.1f} s", file=sys.stderr, flush=True)
        out["other_workloads"] = extras
        n_v = sum(1 for e in extras if "rocsparse_best_ms" in e)
        out["vendor_comparison"] = (f"rocSPARSE (csr_adaptive, csr_nnzsplit: its best on every workload of rounds 1-2) timed beside {n_v} of "
                                    f"{len(extras)} workloads; --vendor {args.vendor}"
                                    + (": " + vendor_box["error"] if "error" in vendor_box else "")
                                    + (f"; library loaded in {vendor_box['loaded_s']} s" if "loaded_s" in vendor_box else ""))
        # the reference's vendor slot proper: cublasSgemv on the dense 4096 x 4096 matrix (cublas.cu:20-35) -- here rocBLAS's
        # sgemv (through torch.mv, which links the ROCm BLAS) beside this library's own kernel in that slot
        try:
            M = N = 4096
            Ad, xd = W.dense_random(M, N, 0.5, seed=1)
            tA, tx_ = torch.from_numpy(Ad).to(dev), torch.from_numpy(xd).to(dev)
            ty = torch.empty(N, dtype=torch.float32, device=dev)
            ws = torch.empty(max(capi.dense_gemv_workspace_bytes(N, 2), 16), dtype=torch.uint8, device=dev)

            def t_of(fn, it=50):
                for _ in range(5):
                    fn()
                best_ms = float("inf")
                for _ in range(3):
                    ev0.record()
                    for _ in range(it):
                        fn()
                    ev1.record()
                    torch.cuda.synchronize()
                    best_ms = min(best_ms, ev0.elapsed_time(ev1) / it)
                return best_ms
            tAt = tA.t()
            ours = t_of(lambda: capi.dense_gemv(tA, tx_, ty, 2, workspace=ws))
            y_ours = ty.clone()
            theirs = t_of(lambda: torch.mv(tAt, tx_, out=ty))
            out["vendor_dense_slot"] = {
                "what": "the reference's first registered slot, cublas_gemv_gpu (src/kernels/cublas.cu:20-35): y = A^T x, dense 4096 x 4096",
                "k_gemv_split_ms": round(ours, 5), "rocblas_sgemv_via_torch_mv_ms": round(theirs, 5),
                "speedup_vs_vendor": round(theirs / ours, 2),
                "dense_GBs_ours": round(M * N * 4 / ours / 1e6, 1), "dense_GBs_vendor": round(M * N * 4 / theirs / 1e6, 1),
                "max_abs_difference": float((y_ours - ty).abs().max().item())}
        except Exception as ex:
            out["vendor_dense_slot"] = {"error": f"{type(ex).__name__}: {str(ex)[:160]}"}
        # the headline's config under the other column laws, next to the headline (BASELINE fixes c4's sizes and
        # row-length law, not its column law: the value above holds for the law named in config.workload only)
        head_law = f"band {args.band}" if args.band else "uniform"
        laws = {head_law: {"frac_hbm": out["roofline"]["frac"] if out["roofline"]["traffic"] is not None else None,
                           "effective_frac": out["roofline"]["effective_frac"]}}
        for e in extras:
            if e["workload"].startswith(args.config + ":"):
                law = "uniform" if "uniform columns" in e["workload"] else "band " + e["workload"].split("band ")[1].split(")")[0]
                laws[law] = {"frac_hbm": e.get("auto_frac_hbm"), "effective_frac": e["auto_effective_frac"]}
        out["frac_by_column_law"] = {"config": args.config, "variant": args.variant,
                                     "frac_hbm": "measured traffic / time / 8 TB/s", "effective_frac": "CSR-algorithmic bytes / time / 8 TB/s "
                                     "(may exceed what the chip moves)", **laws}

    if rank == 0:
        print(json.dumps(out), flush=True)
        print(f"[bench] total {time.perf_counter() - t_main:.1f} s in the process", file=sys.stderr, flush=True)
    if vendor_cat is not None and vendor_cat.poll() is None:
        vendor_cat.kill()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
