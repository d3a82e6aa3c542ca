#!/usr/bin/env python3
"""tools/bitmap_time.py -- the reference's bitmap formats (WSP / AWSP / AWSPRef / TCSR) against the CSR kernels on dense-ish
matrices (the reference's own regime: 4096^2 at 50 %), one process, HIP events.  Prints one JSON line per (size, density)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    import numpy as np
    import torch
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, iters=50):
        for _ in range(5):
            fn()
        best = None
        for _ in range(3):
            ev0.record()
            for _ in range(iters):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / iters
            best = ms if best is None or ms < best else best
        return best

    for n, zero, xzero in ((4096, 0.5, 0.5), (4096, 0.5, 0.0), (4096, 0.9, 0.5), (16384, 0.5, 0.5), (16384, 0.5, 0.9),
                           (16384, 0.9, 0.0)):
        rng = np.random.Generator(np.random.PCG64(n))
        dA = (torch.rand(n, n, device=dev) * 2 - 1) * (torch.rand(n, n, device=dev) >= zero)
        dx = (torch.rand(n, device=dev) * 2 - 1) * (torch.rand(n, device=dev) >= xzero)
        dy = torch.empty(n, device=dev)
        out = {"n": n, "A_zero": zero, "x_zero": xzero}
        csr = capi.CsrMatrix.from_dense_device(dA)
        out["nnz"] = csr.nnz
        out["csr_bytes"] = W.algorithmic_bytes(n, n, csr.nnz)
        for vn in ("wave", "wave_pipe", "adaptive", "tiled", "xskip"):
            v = capi.ALL_VARIANTS[vn]
            csr.plan(v)
            out[f"csr_{vn}_ms"] = round(timed(lambda: csr.run(v, dx, dy)), 5)
        ref = dy.clone()
        t = capi.TcsrMatrix.from_dense_device(dA)
        out["tcsr_ms"] = round(timed(lambda: t.run(dx, dy)), 5)
        t.close()
        for fmt in ("wsp", "awsp", "awsp_ref"):
            B = capi.BitmapMatrix.from_dense_device(fmt, dA)
            out[f"{fmt}_ms"] = round(timed(lambda: B.run(dx, dy)), 5)
            out[f"{fmt}_bytes"] = 4 * (B.n_bitmaps + B.n_vals)
            out[f"{fmt}_max_abs_diff_vs_csr"] = float((dy - ref).abs().max())
            B.close()
        for mode in (2, 3):
            out[f"dense_mode{mode}_ms"] = round(timed(lambda: capi.dense_gemv(dA, dx, dy, mode)), 5)
        csr.close()
        print(json.dumps(out), flush=True)
        del dA
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
