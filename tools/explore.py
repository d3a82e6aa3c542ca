#!/usr/bin/env python3
"""tools/explore.py -- development aid: several (workload, variant, knob) experiments in ONE process
(process start-up on a fresh GPU box is slow).  Edit EXPERIMENTS or pass a JSON list on argv[1]."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

DEFAULT = [
    {"dist": "const", "mean": 16, "band": 256, "variants": ["adaptive", "tiled"], "env": {"SPMV_TILE_CAP": "4224"}},
    {"dist": "mixed", "mean": 16, "band": 256, "variants": ["adaptive", "tiled"], "env": {"SPMV_TILE_CAP": "4224"}},
]


def main():
    import torch
    exps = json.loads(sys.argv[1]) if len(sys.argv) > 1 else DEFAULT
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    cache = {}
    for e in exps:
        rows = e.get("rows", 16 * W.Mi)
        key = (e["dist"], e["mean"], e["band"], rows)
        if key not in cache:
            cache.clear()
            torch.cuda.empty_cache()
            w = W.Workload("x", rows, rows, e["dist"], e["mean"], band=e["band"])
            rp = W.row_ptr(w)
            d_rp = torch.from_numpy(rp).to(dev)
            d_ci = torch.empty(w.nnz, dtype=torch.int32, device=dev)
            d_va = torch.empty(w.nnz, dtype=torch.float32, device=dev)
            d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
            d_y = torch.empty(w.rows, dtype=torch.float32, device=dev)
            capi.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
            capi.synth_x(w.seed, 0, w.cols, d_x)
            cache[key] = (w, d_rp, d_ci, d_va, d_x, d_y)
        w, d_rp, d_ci, d_va, d_x, d_y = cache[key]
        os.environ.update(e.get("env", {}))
        if "lib" in e:          # A/B between builds of the library inside one process
            capi.use_library(ROOT / e["lib"])
        A = capi.CsrMatrix.from_device(w.rows, w.cols, d_rp, d_ci, d_va)
        B = W.algorithmic_bytes(w.rows, w.cols, w.nnz)
        for vname in e["variants"]:
            v = capi.VARIANTS[vname]
            import time as _t
            torch.cuda.synchronize(); _t0 = _t.perf_counter()
            A.plan(v)
            torch.cuda.synchronize(); plan_ms = (_t.perf_counter() - _t0) * 1e3
            A.time(v, d_x, d_y, 3)
            ms = min(A.time(v, d_x, d_y, e.get("iters", 30)) for _ in range(3))
            print(json.dumps(dict(dist=e["dist"], band=e["band"], rows=rows, variant=vname, env=e.get("env", {}), lib=e.get("lib", ""),
                                  ms=round(ms, 4), plan_ms=round(plan_ms, 2), GBs=round(B / ms / 1e6, 1), pct=round(B / ms / 1e6 / 80, 2),
                                  plan=A.plan_describe(v))), flush=True)
        A.close()


if __name__ == "__main__":
    main()
