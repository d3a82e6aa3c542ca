#!/usr/bin/env python3
"""tools/xskip_traffic.py ZERO_FRACTION -- 20 launches of SPMV_XSKIP (and 20 of SPMV_ADAPTIVE for comparison) on a
16384 x 16384 matrix at 50 % density with a given fraction of zeros in x; run under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/xskip_traffic.py 0.9
to see in the counters that the segments of zero inputs are never read."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    import torch
    zero = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    n = 16384
    g = torch.Generator(device=dev).manual_seed(1)
    dA = (torch.rand(n, n, device=dev, generator=g) * 2 - 1) * (torch.rand(n, n, device=dev, generator=g) >= 0.5)
    dx = (torch.rand(n, device=dev, generator=g) * 2 - 1) * (torch.rand(n, device=dev, generator=g) >= zero)
    dy = torch.empty(n, device=dev)
    A = capi.CsrMatrix.from_dense_device(dA)
    del dA
    out = {"n": n, "nnz": A.nnz, "x_zero_fraction_asked": zero, "x_nonzeros": int((dx != 0).sum().item()),
           "csr_algorithmic_bytes": W.algorithmic_bytes(n, n, A.nnz), "xskip_entry_bytes": 6 * A.nnz}
    for name in ("xskip", "adaptive"):
        v = capi.ALL_VARIANTS[name]
        A.plan(v)
        out[name + "_ms"] = round(A.time(v, dx, dy, 20), 5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
