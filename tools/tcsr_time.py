#!/usr/bin/env python3
"""tools/tcsr_time.py -- tiled bitmap-CSR (f-3) against the CSR variants in the reference's own regime
(dense-ish matrices, tester.cpp:106 uses 50 % zeros).  Kernel time by HIP/torch events, 1 GPU."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    import torch
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    for (M, N, zero) in [(4096, 4096, 0.5), (16384, 16384, 0.5), (16384, 16384, 0.9)]:
        g = torch.Generator(device=dev).manual_seed(1)
        A = torch.rand((M, N), device=dev, generator=g) * 2 - 1
        A[torch.rand((M, N), device=dev, generator=g) < zero] = 0
        x = torch.rand(M, device=dev, generator=g) * 2 - 1
        y = torch.empty(N, device=dev)
        t = capi.TcsrMatrix.from_dense_device(A)
        c = capi.CsrMatrix.from_dense_device(A)
        nnz = t.nnz
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timeit(fn, iters=50):
            for _ in range(5):
                fn()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters
        ms_t = timeit(lambda: t.run(x, y))
        y_t = y.clone()
        b_tcsr = 4 * nnz + M * N // 8 + 4 * (t.n_blk_idx) + 4 * M + 4 * N
        b_csr = W.algorithmic_bytes(N, M, nnz)
        rec = dict(M=M, N=N, zero=zero, nnz=nnz, tcsr_ms=round(ms_t, 4), tcsr_format_GBs=round(b_tcsr / ms_t / 1e6, 1),
                   tcsr_csr_equiv_GBs=round(b_csr / ms_t / 1e6, 1), tcsr_bytes=b_tcsr, csr_bytes=b_csr)
        for vn in ("scalar", "wave", "wave_pipe", "vector", "adaptive", "tiled"):
            v = capi.VARIANTS[vn]
            c.plan(v)
            ms = timeit(lambda: c.run(v, x, y))
            rec[f"csr_{vn}_ms"] = round(ms, 4)
            rec[f"maxdiff_{vn}"] = float((y - y_t).abs().max())
        for mode in (2,):
            ms = timeit(lambda: capi.dense_gemv(A, x, y, mode))
            rec["dense_split_ms"] = round(ms, 4)
        print(json.dumps(rec), flush=True)
        t.close(); c.close()


if __name__ == "__main__":
    main()
