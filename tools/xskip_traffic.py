#!/usr/bin/env python3
"""tools/xskip_traffic.py ZERO_FRACTION [ROWS NNZ_PER_ROW] -- 20 launches of SPMV_XSKIP (and 20 of SPMV_ADAPTIVE for
comparison) with a given fraction of zeros in x, on a 16384 x 16384 matrix at 50 % density (the reference's regime) or,
with ROWS and NNZ_PER_ROW, on a ROWS x ROWS synthetic CSR with that many nonzeros per row in uniform columns (round 3:
config 2's density -- segments of (1024 outputs, one input) that hold a nonzero or none); run under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/xskip_traffic.py 0.9
to see in the counters what skipping saves."""
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
    g = torch.Generator(device=dev).manual_seed(1)
    if len(sys.argv) > 3:
        n, per_row = int(sys.argv[2]), int(sys.argv[3])
        w = W.Workload("xs", n, n, "const", per_row, band=0)
        rp = W.row_ptr(w)
        d_rp = torch.from_numpy(rp).to(dev)
        d_ci = torch.empty(w.nnz, dtype=torch.int32, device=dev)
        d_va = torch.empty(w.nnz, dtype=torch.float32, device=dev)
        capi.synth_fill(w.seed, 0, n, n, n, 0, d_rp, d_ci, d_va)
        A = capi.CsrMatrix.from_device(n, n, d_rp, d_ci, d_va)
    else:
        n = 16384
        dA = (torch.rand(n, n, device=dev, generator=g) * 2 - 1) * (torch.rand(n, n, device=dev, generator=g) >= 0.5)
        A = capi.CsrMatrix.from_dense_device(dA)
        del dA
    dx = (torch.rand(n, device=dev, generator=g) * 2 - 1) * (torch.rand(n, device=dev, generator=g) >= zero)
    dy = torch.empty(n, device=dev)
    out = {"n": n, "nnz": A.nnz, "nnz_per_segment_of_1024_outputs": round(A.nnz / n * 1024 / n, 4),
           "x_zero_fraction_asked": zero, "x_nonzeros": int((dx != 0).sum().item()),
           "csr_algorithmic_bytes": W.algorithmic_bytes(n, n, A.nnz), "xskip_entry_bytes": 6 * A.nnz}
    for name in ("xskip", "adaptive"):
        v = capi.ALL_VARIANTS[name]
        A.plan(v)
        out[name + "_ms"] = round(A.time(v, dx, dy, 20), 5)
        out[name + "_plan"] = A.plan_describe(v)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
