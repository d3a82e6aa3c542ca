"""GPU (-m gpu): the reference's WSP / AWSP / AWSPRef bitmap formats built and multiplied on the device."""
import numpy as np
import pytest

from _util import assert_close_to_oracle

pytestmark = pytest.mark.gpu

FORMATS = ["wsp", "awsp", "awsp_ref"]


@pytest.mark.parametrize("fmt", FORMATS)
def test_bitmap_device_builder_is_bit_exact_vs_reference_arrays(pkg, gpu, golden, fmt):
    """spmv_bitmap_from_dense_host vs the arrays the reference's own WSPMatrix / AWSPMatrix / AWSPRefMatrix produced
    (tests/golden, made through oracle/_ref): bitmaps, padded values and the exposed statistics."""
    capi = pkg.capi
    if fmt not in golden.bitmap:
        with pytest.raises(capi.SpmvError) as e:           # not 32-aligned: refused, like the reference's assert
            capi.BitmapMatrix.from_dense_host(fmt, golden.A)
        assert e.value.status == capi.ERR_INVALID
        return
    rbm, rva, rst = golden.bitmap[fmt]
    B = capi.BitmapMatrix.from_dense_host(fmt, golden.A)
    assert B.stats == [int(v) for v in rst], (B.stats, rst)
    assert (B.n_bitmaps, B.n_vals) == (len(rbm), len(rva))
    bm, va = B.download()
    assert np.array_equal(bm, rbm)
    assert np.array_equal(va.view(np.uint32), rva.view(np.uint32))
    B.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_bitmap_spmv_matches_oracle_on_fixtures(pkg, oracle, gpu, golden, fmt):
    if fmt not in golden.bitmap:
        pytest.skip("not 32-aligned")
    B = pkg.capi.BitmapMatrix.from_dense_host(fmt, golden.A)
    y = np.full(golden.N, np.nan, np.float32)
    ms = B.run_host(golden.x, y)
    assert ms > 0
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    assert_close_to_oracle(y, y64, mag, f"{fmt}/{golden.name}")
    B.close()


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("M,N,zero", [(4096, 4096, 0.5), (32, 32, 0.0), (2048, 96, 0.9), (96, 4096, 0.97), (320, 1024, 1.0),
                                      (1056, 160, 0.3)])
def test_bitmap_tester_regime_and_shapes(pkg, oracle, gpu, fmt, M, N, zero):
    """4096^2 at 50 % is the reference tester's own problem (test/main.cpp:4, tester.cpp:106); the other shapes: one
    block, M/4 not a multiple of 32 (96, 1056), an all-zero matrix, very sparse.  Arrays against the pinned CPU
    restatement, the product against the CSR oracle, with the tester's 50 %-zero x (the x == 0 skip is live)."""
    import torch
    capi = pkg.capi
    A, x = pkg.workloads.dense_random(M, N, zero, seed=M * 7 + N)
    A.flat[::11] = -0.0                                    # dropped by `!= 0.0f`
    B = capi.BitmapMatrix.from_dense_device(fmt, torch.from_numpy(A).to(gpu))
    obm, ova, ost = oracle.bitmap_from_dense(fmt, A)
    assert B.stats == [int(v) for v in ost]
    bm, va = B.download()
    assert np.array_equal(bm, obm) and len(va) == len(ova) and np.array_equal(va.view(np.uint32), ova.view(np.uint32))
    dx = torch.from_numpy(x).to(gpu)
    dy = torch.full((N,), float("nan"), device=gpu)
    B.run(dx, dy)
    torch.cuda.synchronize()
    rp, ci, cv = oracle.csr_from_dense(A)
    y64, mag = oracle.spmv_f64(rp, ci, cv, x)
    assert_close_to_oracle(dy.cpu().numpy(), y64, mag, f"{fmt} {M}x{N}")
    # a dense x as well (no lane skips)
    xd = np.random.Generator(np.random.PCG64(M + N)).uniform(-1, 1, size=M).astype(np.float32)
    B.run(torch.from_numpy(xd).to(gpu), dy)
    torch.cuda.synchronize()
    y64, mag = oracle.spmv_f64(rp, ci, cv, xd)
    assert_close_to_oracle(dy.cpu().numpy(), y64, mag, f"{fmt} {M}x{N} dense x")
    B.close()


def test_bitmap_rejects_bad_arguments(pkg, gpu):
    capi = pkg.capi
    with pytest.raises(capi.SpmvError) as e:
        capi.BitmapMatrix.from_dense_host("wsp", np.ones((33, 32), np.float32))
    assert e.value.status == capi.ERR_INVALID
    import ctypes as C
    h = C.c_void_p()
    A = np.ones((32, 32), np.float32)
    assert capi.lib().spmv_bitmap_from_dense_host(7, 32, 32, A.ctypes.data, 0, C.byref(h)) == capi.ERR_VARIANT
