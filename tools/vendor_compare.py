#!/usr/bin/env python3
"""tools/vendor_compare.py -- rocSPARSE csrmv beside this library, same matrices, same process.

SURVEY.md section 8 a-9: the vendor library is the COMPARISON slot at configs 2-4 (the reference's
own vendor slot is cublasSgemv, cublas.cu:33, on the dense matrix).  Nothing here is on the product
path: rocSPARSE (/opt/rocm/lib/librocsparse.so, part of the ROCm image) is loaded with ctypes, its
four CSR SpMV algorithms are preprocessed and timed with HIP events, their y is checked against
ours, and one JSON line per (workload, implementation) is printed.

    python tools/vendor_compare.py [--out FILE] [--iters 30]
"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ALGS = {"csr_adaptive": 2, "csr_rowsplit": 3, "csr_lrb": 7, "csr_nnzsplit": 8}
STAGE_SIZE, STAGE_PREP, STAGE_COMPUTE = 1, 2, 3
I32, F32 = 2, 151          # rocsparse_indextype_i32, rocsparse_datatype_f32_r; 111 below = rocsparse_operation_none

WORKLOADS = [
    ("c2", dict(rows=1 << 20, dist="const", mean=16, band=0)),
    ("c2", dict(rows=1 << 20, dist="const", mean=16, band=8192)),
    ("c3", dict(rows=4 << 20, dist="powerlaw", mean=32, band=0)),
    ("c3", dict(rows=4 << 20, dist="powerlaw", mean=32, band=8192)),
    ("c4", dict(rows=16 << 20, dist="mixed", mean=16, band=0)),
    ("c4", dict(rows=16 << 20, dist="mixed", mean=16, band=8192)),
    ("c4", dict(rows=16 << 20, dist="mixed", mean=16, band=65536)),
]


class RocSparse:
    def __init__(self):
        L = ctypes.CDLL("/opt/rocm/lib/librocsparse.so")
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.rocsparse_create_handle.argtypes = [ctypes.POINTER(vp)]
        L.rocsparse_set_stream.argtypes = [vp, vp]
        L.rocsparse_create_csr_descr.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, vp, i32, i32, i32, i32]
        L.rocsparse_create_dnvec_descr.argtypes = [ctypes.POINTER(vp), i64, vp, i32]
        L.rocsparse_spmv.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.POINTER(ctypes.c_size_t), vp]
        L.rocsparse_destroy_spmat_descr.argtypes = [vp]
        L.rocsparse_destroy_dnvec_descr.argtypes = [vp]
        L.rocsparse_destroy_handle.argtypes = [vp]
        self.L = L
        self.h = vp()
        self._ok(L.rocsparse_create_handle(ctypes.byref(self.h)), "create_handle")

    @staticmethod
    def _ok(st, what):
        if st != 0:
            raise RuntimeError(f"rocsparse {what}: status {st}")

    def close(self):
        self.L.rocsparse_destroy_handle(self.h)


def rocsparse_times(rs, rows, cols, nnz, d_rp, d_ci, d_va, d_x, d_y, iters=20, algs=None):
    """{algorithm: (ms per SpMV, preprocess ms)} of rocsparse_spmv on device CSR arrays (torch tensors), every algorithm
    preprocessed, HIP events on torch's current stream; an algorithm that reports an error is left out.  Used by main()
    below and by bench.py (the vendor slot of the reference, src/kernels/cublas.cu:33, as a COMPARISON)."""
    import torch
    dev = d_x.device
    one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
    vx, vy = ctypes.c_void_p(), ctypes.c_void_p()
    rs._ok(rs.L.rocsparse_create_dnvec_descr(ctypes.byref(vx), cols, d_x.data_ptr(), F32), "dnvec x")
    rs._ok(rs.L.rocsparse_create_dnvec_descr(ctypes.byref(vy), rows, d_y.data_ptr(), F32), "dnvec y")
    out = {}
    for aname, alg in (algs or ALGS).items():
        size = ctypes.c_size_t(0)
        mat = ctypes.c_void_p()   # the analysis of an algorithm is cached INSIDE the matrix descriptor: a fresh one per algorithm
        rs._ok(rs.L.rocsparse_create_csr_descr(ctypes.byref(mat), rows, cols, nnz, d_rp.data_ptr(), d_ci.data_ptr(),
                                               d_va.data_ptr(), I32, I32, 0, F32), "create_csr")

        def call(stage, buf, mat=mat, alg=alg, size=size):
            return rs.L.rocsparse_spmv(rs.h, 111, ctypes.byref(one), mat, vx, ctypes.byref(zero), vy, F32, alg, stage,
                                       ctypes.byref(size), buf)
        ok = call(STAGE_SIZE, None) == 0
        buf = None
        if ok:
            buf = torch.empty(max(int(size.value), 16), dtype=torch.uint8, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ok = call(STAGE_PREP, buf.data_ptr()) == 0
            e1.record()
            torch.cuda.synchronize()
            prep_ms = e0.elapsed_time(e1)
        if ok:
            d_y.zero_()      # (NaN-prefilled y comes back NaN from some rocSPARSE algorithms even with beta = 0)
            for _ in range(3):
                ok = ok and call(STAGE_COMPUTE, buf.data_ptr()) == 0
        if ok:
            best = float("inf")
            for _ in range(2):
                e0.record()
                for _ in range(iters):
                    call(STAGE_COMPUTE, buf.data_ptr())
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters)
            out[aname] = (best, prep_ms)
        del buf
        torch.cuda.synchronize()
        rs.L.rocsparse_destroy_spmat_descr(mat)
    rs.L.rocsparse_destroy_dnvec_descr(vx)
    rs.L.rocsparse_destroy_dnvec_descr(vy)
    return out


def warm():
    """--warm: load librocsparse.so and run its two algorithms once on a tiny matrix, in a process of its own -- what that
    touches of the 0.5 GB file (the relocations, the gfx950 code objects) is then in the page cache when bench.py loads the
    library itself.  (Reading the whole file -- every architecture's code -- took a freshly booted box more than four minutes.)"""
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    rs = RocSparse()
    rs._ok(rs.L.rocsparse_set_stream(rs.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")
    n, k = 4096, 8
    d_rp = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
    d_ci = torch.from_numpy((np.arange(n * k, dtype=np.int64) * 7 % n).astype(np.int32)).to(dev)
    d_va = torch.ones(n * k, dtype=torch.float32, device=dev)
    d_x = torch.ones(n, dtype=torch.float32, device=dev)
    d_y = torch.zeros(n, dtype=torch.float32, device=dev)
    rt = rocsparse_times(rs, n, n, n * k, d_rp, d_ci, d_va, d_x, d_y, iters=1,
                         algs={a: ALGS[a] for a in ("csr_adaptive", "csr_nnzsplit")})
    torch.cuda.synchronize()
    print(json.dumps({"warmed": sorted(rt), "y_ok": bool((d_y == k).all().item())}))
    rs.close()


def main():
    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warm", action="store_true", help="only load the library and run it once on a tiny matrix (bench.py's warm-up child)")
    args = ap.parse_args()
    if args.warm:
        warm()
        return

    pkg = ge.load_package()
    capi, W = pkg.capi, pkg.workloads
    dev = torch.device("cuda:0")
    rs = RocSparse()
    stream = torch.cuda.current_stream().cuda_stream
    rs._ok(rs.L.rocsparse_set_stream(rs.h, ctypes.c_void_p(stream)), "set_stream")
    one = ctypes.c_float(1.0)
    zero = ctypes.c_float(0.0)
    lines = []

    def emit(**kw):
        lines.append(kw)
        print(json.dumps(kw), flush=True)

    for cname, spec in WORKLOADS:
        torch.cuda.empty_cache()
        w = W.Workload(cname, spec["rows"], spec["rows"], spec["dist"], spec["mean"], band=spec["band"])
        rp = W.row_ptr(w)
        d_rp = torch.from_numpy(rp).to(dev)
        d_ci = torch.empty(w.nnz, dtype=torch.int32, device=dev)
        d_va = torch.empty(w.nnz, dtype=torch.float32, device=dev)
        d_x = torch.empty(w.cols, dtype=torch.float32, device=dev)
        d_y = torch.empty(w.rows, dtype=torch.float32, device=dev)
        d_yv = torch.empty(w.rows, dtype=torch.float32, device=dev)
        capi.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, d_rp, d_ci, d_va)
        capi.synth_x(w.seed, 0, w.cols, d_x)
        B = W.algorithmic_bytes(w.rows, w.cols, w.nnz)
        label = f"{cname} band {w.band}" if w.band else f"{cname} uniform"

        A = capi.CsrMatrix.from_device(w.rows, w.cols, d_rp, d_ci, d_va)
        ours = {}
        for vname in ("scalar", "wave", "adaptive", "tiled", "panel", "auto"):
            v = capi.VARIANTS[vname]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            A.plan(v)
            torch.cuda.synchronize()
            plan_ms = (time.perf_counter() - t0) * 1e3
            A.time(v, d_x, d_y, 3)
            ms = min(A.time(v, d_x, d_y, args.iters) for _ in range(3))
            ours[vname] = ms
            emit(workload=label, impl=f"this:{vname}", ms=round(ms, 4), GBs=round(B / ms / 1e6, 1),
                 pct_of_8TBs=round(B / ms / 1e6 / 80, 2), preprocess_ms=round(plan_ms, 3))
        emit(workload=label, impl="this:auto choice", plan=A.plan_describe(capi.AUTO)[:60])
        best_ours = min(ours, key=ours.get)
        ours["best"] = ours[best_ours]
        A.run(capi.TILED, d_x, d_y)
        torch.cuda.synchronize()
        y_ours = d_y.clone()

        vx, vy = ctypes.c_void_p(), ctypes.c_void_p()
        rs._ok(rs.L.rocsparse_create_dnvec_descr(ctypes.byref(vx), w.cols, d_x.data_ptr(), F32), "dnvec x")
        rs._ok(rs.L.rocsparse_create_dnvec_descr(ctypes.byref(vy), w.rows, d_yv.data_ptr(), F32), "dnvec y")
        for aname, alg in ALGS.items():
            size = ctypes.c_size_t(0)
            # the analysis of an algorithm is cached INSIDE the matrix descriptor: a fresh one per algorithm
            mat = ctypes.c_void_p()
            rs._ok(rs.L.rocsparse_create_csr_descr(ctypes.byref(mat), w.rows, w.cols, w.nnz, d_rp.data_ptr(),
                                                   d_ci.data_ptr(), d_va.data_ptr(), I32, I32, 0, F32), "create_csr")

            def call(stage, buf, mat=mat, alg=alg, size=size):
                return rs.L.rocsparse_spmv(rs.h, 111, ctypes.byref(one), mat, vx, ctypes.byref(zero), vy, F32, alg,
                                           stage, ctypes.byref(size), buf)
            st = call(STAGE_SIZE, None)
            if st != 0:
                emit(workload=label, impl=f"rocsparse:{aname}", error=f"buffer_size status {st}")
                continue
            buf = torch.empty(max(int(size.value), 16), dtype=torch.uint8, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st = call(STAGE_PREP, buf.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            if st != 0:
                emit(workload=label, impl=f"rocsparse:{aname}", error=f"preprocess status {st}")
                continue
            prep_ms = e0.elapsed_time(e1)
            d_yv.zero_()      # (NaN-prefilled y comes back NaN from some rocSPARSE algorithms even with beta = 0)
            for _ in range(3):
                st = call(STAGE_COMPUTE, buf.data_ptr())
            if st != 0:
                emit(workload=label, impl=f"rocsparse:{aname}", error=f"compute status {st}")
                continue
            best = float("inf")
            for _ in range(3):
                e0.record()
                for _ in range(args.iters):
                    call(STAGE_COMPUTE, buf.data_ptr())
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.iters)
            diff = (d_yv - y_ours).abs().max().item()
            emit(workload=label, impl=f"rocsparse:{aname}", ms=round(best, 4), GBs=round(B / best / 1e6, 1),
                 pct_of_8TBs=round(B / best / 1e6 / 80, 2), preprocess_ms=round(prep_ms, 3),
                 workspace_bytes=int(size.value), max_abs_diff_vs_this=diff,
                 this_tiled_speedup=round(best / ours["tiled"], 2), this_auto_speedup=round(best / ours["auto"], 2),
                 this_best=best_ours,
                 this_best_speedup=round(best / ours["best"], 2))
            del buf
            torch.cuda.synchronize()
            rs.L.rocsparse_destroy_spmat_descr(mat)
        rs.L.rocsparse_destroy_dnvec_descr(vx)
        rs.L.rocsparse_destroy_dnvec_descr(vy)
        A.close()
    rs.close()
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text("\n".join(json.dumps(l) for l in lines) + "\n")


if __name__ == "__main__":
    main()
