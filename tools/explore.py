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
        shard = bool(e.get("c5shard"))         # rows [0, 16Mi) of config 5's (128Mi)^2 matrix, all 128Mi columns
        key = (e["dist"], e["mean"], e["band"], rows, shard)
        if key not in cache:
            cache.clear()
            torch.cuda.empty_cache()
            w = W.c5(8, band=e["band"]) if shard else W.Workload("x", rows, rows, e["dist"], e["mean"], band=e["band"])
            n_loc = 16 * W.Mi if shard else w.rows
            rp = W.row_ptr(w, 0, n_loc)
            nnz = int(rp[-1])
            d_rp = torch.from_numpy(rp).to(dev)
            d_ci = torch.empty(nnz, dtype=torch.int32, device=dev)
            d_va = torch.empty(nnz, dtype=torch.float32, device=dev)
            d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
            d_y = torch.empty(n_loc, dtype=torch.float32, device=dev)
            capi.synth_fill(w.seed, 0, n_loc, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
            capi.synth_x(w.seed, 0, w.cols, d_x)
            import dataclasses
            cache[key] = (dataclasses.replace(w, rows=n_loc), d_rp, d_ci, d_va, d_x, d_y, nnz)
        w, d_rp, d_ci, d_va, d_x, d_y, nnz = cache[key]
        os.environ.update(e.get("env", {}))
        if "lib" in e:          # A/B between builds of the library inside one process
            capi.use_library(ROOT / e["lib"])
        A = capi.CsrMatrix.from_device(w.rows, w.cols, d_rp, d_ci, d_va)
        B = W.algorithmic_bytes(w.rows, w.cols, nnz)
        for vname in e["variants"]:
            v = capi.VARIANTS[vname]
            import time as _t
            torch.cuda.synchronize(); _t0 = _t.perf_counter()
            if vname == "panel" and "panel_params" in e:     # [rows or bits, waves, mode]: spmv_csr_plan_set's params[4..6]
                pp = e["panel_params"]
                A.plan_set(v, [v, 0, 0, 0, pp[0], pp[1], pp[2], 0])
            else:
                A.plan(v)
            torch.cuda.synchronize(); plan_ms = (_t.perf_counter() - _t0) * 1e3
            A.time(v, d_x, d_y, 3)
            ms = min(A.time(v, d_x, d_y, e.get("iters", 30)) for _ in range(3))
            print(json.dumps(dict(dist=e["dist"], band=e["band"], rows=w.rows, cols=w.cols, variant=vname, panel_params=e.get("panel_params"), env=e.get("env", {}), lib=e.get("lib", ""),
                                  ms=round(ms, 4), plan_ms=round(plan_ms, 2), GBs=round(B / ms / 1e6, 1), pct=round(B / ms / 1e6 / 80, 2),
                                  plan=A.plan_describe(v))), flush=True)
        A.close()


if __name__ == "__main__":
    main()
