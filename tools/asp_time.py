#!/usr/bin/env python3
"""tools/asp_time.py -- activation-sparsity skip (f-2) on the dense slot: mode 2 vs mode 3 on the row-major matrix, and (round 3)
the multiply from the reference's ASP layout (spmv_asp_gemv_ws), with 0 / 50 / 90 %-zero x."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
import torch
pkg = ge.load_package(); capi = pkg.capi
dev = torch.device("cuda:0")
for (M, N) in [(4096, 4096), (16384, 16384)]:
    g = torch.Generator(device=dev).manual_seed(2)
    A = torch.rand((M, N), device=dev, generator=g) * 2 - 1
    A[torch.rand((M, N), device=dev, generator=g) < 0.5] = 0
    for xz in (0.0, 0.5, 0.9):
        x = torch.rand(M, device=dev, generator=g) * 2 - 1
        x[torch.rand(M, device=dev, generator=g) < xz] = 0
        y = torch.empty(N, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rec = dict(M=M, N=N, x_zero=xz)
        for mode in (2, 3):
            for _ in range(5): capi.dense_gemv(A, x, y, mode)
            e0.record()
            for _ in range(50): capi.dense_gemv(A, x, y, mode)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 50
            rec[f"mode{mode}_ms"] = round(ms, 4); rec[f"mode{mode}_dense_GBs"] = round(M * N * 4 / ms / 1e6, 1)
        if M % 32 == 0 and N % 32 == 0:
            asp = torch.empty(M * N, device=dev)
            capi.asp_retile(A, asp)
            ws = torch.empty(capi.dense_gemv_workspace_bytes(N, 3), dtype=torch.uint8, device=dev)
            for _ in range(5): capi.asp_gemv(M, N, asp, x, y, ws)
            e0.record()
            for _ in range(50): capi.asp_gemv(M, N, asp, x, y, ws)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 50
            rec["asp_layout_ms"] = round(ms, 4); rec["asp_layout_dense_GBs"] = round(M * N * 4 / ms / 1e6, 1)
            del asp
        print(json.dumps(rec), flush=True)
