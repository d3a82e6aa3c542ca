"""Helpers shared by the GPU parity tests."""
import numpy as np

# north_star: "within 1e-5 relative on fp32 values".  A reordered fp32 sum cannot meet an
# element-wise relative bound on cancelling rows, so the bound is taken
# relative to the row's magnitude sum_k |val_k * x_col_k|, against the fp64 oracle.
RTOL = 1e-5


def assert_close_to_oracle(y, y64, mag, what=""):
    err = np.abs(y.astype(np.float64) - y64)
    bound = RTOL * mag + 1e-37
    bad = np.flatnonzero(~(err <= bound))
    assert bad.size == 0, (f"{what}: {bad.size} rows outside {RTOL:g}*sum|terms|; first {bad[:5]}, "
                           f"err {err[bad[:5]]}, bound {bound[bad[:5]]}")
    # and plain relative 1e-5 wherever the row does not cancel
    solid = np.abs(y64) >= 0.25 * mag
    solid &= mag > 0
    rel = err[solid] / np.abs(y64[solid])
    assert rel.size == 0 or rel.max() <= RTOL, f"{what}: max relative error {rel.max():.3g}"


class DeviceProblem:
    """A CSR matrix + x resident on the GPU, with the host copies the oracle needs."""

    def __init__(self, pkg, dev, rows, cols, row_ptr, col_idx, vals, x):
        import torch
        self.rows, self.cols = rows, cols
        self.row_ptr, self.col_idx, self.vals, self.x = row_ptr, col_idx, vals, x
        self.d_rp = torch.from_numpy(np.ascontiguousarray(row_ptr, np.int32)).to(dev)
        self.d_ci = torch.from_numpy(np.ascontiguousarray(col_idx, np.int32)).to(dev)
        self.d_va = torch.from_numpy(np.ascontiguousarray(vals, np.float32)).to(dev)
        self.d_x = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        if self.d_x.numel() == 0:
            self.d_x = torch.zeros(4, dtype=torch.float32, device=dev)
        self.d_y = torch.empty(max(rows, 1), dtype=torch.float32, device=dev)
        self.A = pkg.capi.CsrMatrix.from_device(rows, cols, self.d_rp, self.d_ci, self.d_va)

    def run(self, variant):
        import torch
        self.A.plan(variant)
        self.d_y.fill_(float("nan"))        # every row must be overwritten
        self.A.run(variant, self.d_x, self.d_y)
        torch.cuda.synchronize()
        return self.d_y[:self.rows].cpu().numpy()


def synth_problem(pkg, oracle, dev, w):
    """Generate workload w on the DEVICE (spmv_synth_fill) and regenerate it on the host (oracle)."""
    import torch
    W, capi = pkg.workloads, pkg.capi
    rp = W.row_ptr(w)
    nnz = int(rp[-1])
    d_rp = torch.from_numpy(rp).to(dev)
    d_ci = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_va = torch.empty(nnz, dtype=torch.float32, device=dev)
    d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
    capi.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
    capi.synth_x(w.seed, 0, w.cols, d_x)
    torch.cuda.synchronize()
    ci, va = oracle.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp)
    x = oracle.synth_x(w.seed, 0, w.cols)
    assert np.array_equal(d_ci.cpu().numpy(), ci), "device generator differs from the host statement (columns)"
    assert np.array_equal(d_va.cpu().numpy().view(np.uint32), va.view(np.uint32)), "generator differs (values)"
    assert np.array_equal(d_x.cpu().numpy().view(np.uint32), x.view(np.uint32)), "generator differs (x)"
    return DeviceProblem(pkg, dev, w.rows, w.cols, rp, ci, va, x)
