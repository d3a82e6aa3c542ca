#!/usr/bin/env python3
"""tools/sweep.py -- every variant on every BASELINE workload (1 GPU), one table.  Development aid;
the judged number comes from bench.py.  Usage: python tools/sweep.py [--configs c2,c3,c4] [--bands 0,65536]"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2,c3,c4")
    ap.add_argument("--bands", default="0,65536")
    ap.add_argument("--variants", default="scalar,wave,wave_pipe,vector,adaptive,tiled")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    rows_out = []
    for cname in args.configs.split(","):
        for band in [int(b) for b in args.bands.split(",")]:
            w = W.config(cname, band=band)
            rp = W.row_ptr(w)
            d_rp = torch.from_numpy(rp).to(dev)
            d_ci = torch.empty(w.nnz, dtype=torch.int32, device=dev)
            d_va = torch.empty(w.nnz, dtype=torch.float32, device=dev)
            d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
            d_y = torch.empty(w.rows, dtype=torch.float32, device=dev)
            capi.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
            capi.synth_x(w.seed, 0, w.cols, d_x)
            A = capi.CsrMatrix.from_device(w.rows, w.cols, d_rp, d_ci, d_va)
            B = W.algorithmic_bytes(w.rows, w.cols, w.nnz)
            ref = None
            for vname in args.variants.split(","):
                v = capi.VARIANTS[vname]
                A.plan(v)
                A.time(v, d_x, d_y, 3)
                ms = min(A.time(v, d_x, d_y, args.iters) for _ in range(3))
                y = d_y.clone()
                if ref is None:
                    ref = y
                maxdiff = float((y - ref).abs().max())
                gbs = B / ms / 1e6
                rec = dict(config=cname, band=band, variant=vname, ms=round(ms, 4), GBs=round(gbs, 1),
                           pct_peak=round(gbs / 80.0, 2), gflops=round(2 * w.nnz / ms / 1e6, 1), maxdiff_vs_first=maxdiff)
                rows_out.append(rec)
                print(json.dumps(rec), flush=True)
            A.close()
            del d_rp, d_ci, d_va, d_x, d_y
            torch.cuda.empty_cache()
    if args.out:
        Path(args.out).write_text(json.dumps(rows_out, indent=1))


if __name__ == "__main__":
    main()
