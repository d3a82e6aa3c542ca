#!/usr/bin/env python3
"""tools/stencil_time.py -- 7-point 3-D stencil matrices (n^3 unknowns, columns r, r+-1, r+-n, r+-n^2: three narrow
clusters of columns 2 n^2 apart) through the CSR variants and rocSPARSE-free: the case a single contiguous window
cannot stage.  Built on the host with numpy, checked against the fp64 oracle on a row sample."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    import numpy as np
    import torch
    pkg = ge.load_package(); orc = ge.load_oracle()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    for n in [int(a) for a in sys.argv[1:]] or [128, 200, 256]:
        N, rp, ci, va = W.stencil7(n)
        x = np.random.Generator(np.random.PCG64(n)).uniform(-1, 1, size=N).astype(np.float32)
        d = [torch.from_numpy(a).to(dev) for a in (rp, ci, va, x)]
        d_y = torch.empty(N, dtype=torch.float32, device=dev)
        A = capi.CsrMatrix.from_device(N, N, d[0], d[1], d[2])
        B = W.algorithmic_bytes(N, N, len(ci))
        s0, s1 = N // 2, N // 2 + (1 << 16)
        rps = (rp[s0:s1 + 1] - rp[s0]).astype(np.int32)
        y64, mag = orc.spmv_f64(rps, ci[rp[s0]:rp[s1]], va[rp[s0]:rp[s1]], x)
        for vn in ("adaptive", "tiled", "panel", "wave_pipe"):
            v = capi.VARIANTS[vn]
            A.plan(v)
            A.time(v, d[3], d_y, 3)
            ms = min(A.time(v, d[3], d_y, 20) for _ in range(3))
            err = np.abs(d_y[s0:s1].cpu().numpy().astype(np.float64) - y64)
            print(json.dumps(dict(stencil=f"{n}^3", rows=N, nnz=len(ci), variant=vn, ms=round(ms, 4), GBs=round(B / ms / 1e6, 1),
                                  pct_of_8TBs=round(B / ms / 1e6 / 80, 2), ok=bool(np.all(err <= 1e-5 * mag + 1e-30)),
                                  plan=A.plan_describe(v))), flush=True)
        A.close()


if __name__ == "__main__":
    main()
