"""CPU: pin oracle/spmv_oracle.c against the committed golden vectors and, where this container has
it, against the reference's own CSRMatrix build (oracle/_ref)."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
REF_LIB = ROOT / "oracle" / "_ref" / "libref_formats.so"


def test_csr_builder_matches_reference_fixture(oracle, golden):
    """matrix_csr.cpp:5-23 -- bit-exact on indices AND values, reference layout (no sentinel)."""
    rp, ci, va = oracle.csr_from_dense(golden.A)
    assert rp[-1] == len(golden.vals)
    assert np.array_equal(rp[:-1], golden.ref_row_ptrs)
    assert len(golden.ref_row_ptrs) == golden.N            # N entries, not N+1
    assert np.array_equal(ci, golden.col_idx)
    assert np.array_equal(va.view(np.uint32), golden.vals.view(np.uint32))


def test_dense_loop_matches_fixture(oracle, golden):
    y = oracle.sgemv_dense(golden.A, golden.x)
    assert np.array_equal(y.view(np.uint32), golden.y.view(np.uint32))


def test_csr_walk_equals_dense_loop(oracle, golden):
    """the sequential CSR walk is bit-identical to SgemvCPU (zeros add exactly +-0)."""
    y = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    # the only representable difference is the sign of a zero (an empty row gives +0 both ways)
    assert np.array_equal(y, golden.y)
    nz = golden.y != 0
    assert np.array_equal(y[nz].view(np.uint32), golden.y[nz].view(np.uint32))


@pytest.mark.parametrize("threads", [2, 3, 8])
def test_multithreaded_walk_is_bit_identical(oracle, golden, threads):
    y1 = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x, threads=1)
    yt = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x, threads=threads)
    assert np.array_equal(y1.view(np.uint32), yt.view(np.uint32))


def test_f64_walk_bounds_fp32_error(oracle, golden):
    y = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    rows_len = np.diff(golden.row_ptr)
    # sequential fp32 sum of n terms: |err| <= n * eps * sum|terms|
    bound = (rows_len + 1) * 2.0 ** -24 * mag + 1e-45
    assert np.all(np.abs(y.astype(np.float64) - y64) <= bound)


def test_edge_fixture_semantics():
    from conftest import load_golden
    g = load_golden("gedge")
    lens = np.diff(g.row_ptr)
    assert lens[0] == 0 and lens[5] == 0 and lens[79] == 0          # empty CSR rows, incl. first/last
    assert g.A[3, 10] == 0 and np.signbit(g.A[3, 10])                # the -0.0f is in the dense input
    r10 = slice(g.row_ptr[10], g.row_ptr[11])
    assert 3 not in g.col_idx[r10] and 4 in g.col_idx[r10]           # -0.0f dropped, denormal kept
    assert g.y[5] == 0 and g.y[79] == 0


def test_compare_counts_like_reference(oracle):
    a = np.array([0.0, 1.0, 2.0, np.nan], np.float32)
    b = np.array([0.0005, 1.002, 2.0, 0.0], np.float32)
    assert oracle.L.oracle_compare(4, a.ctypes.data, b.ctypes.data, ctypes.c_float(1e-3)) == 2


@pytest.mark.skipif(not REF_LIB.exists(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("shape,zero", [((64, 96), 0.5), ((300, 70), 0.95), ((33, 1), 0.0), ((1, 40), 0.3)])
def test_csr_builder_matches_live_reference(oracle, shape, zero):
    """Run the reference's CSRMatrix (compiled from /root/reference by oracle/Makefile) on fresh inputs."""
    lib = ctypes.CDLL(str(REF_LIB))
    lib.ref_csr_build.restype = ctypes.c_void_p
    lib.ref_csr_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.ref_csr_copy.argtypes = [ctypes.c_void_p] * 4
    lib.ref_csr_free.argtypes = [ctypes.c_void_p]
    rng = np.random.Generator(np.random.PCG64(hash(shape) & 0xFFFF))
    A = rng.uniform(-1, 1, size=shape).astype(np.float32)
    A[rng.random(size=shape) < zero] = 0.0
    A.flat[::7] = -0.0
    M, N = shape
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    h = lib.ref_csr_build(M, N, A.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    rp = np.empty(a.value, np.int32); ci = np.empty(b.value, np.int32); va = np.empty(c.value, np.float32)
    lib.ref_csr_copy(h, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
    lib.ref_csr_free(h)
    orp, oci, ova = oracle.csr_from_dense(A)
    assert a.value == N and np.array_equal(orp[:-1], rp) and orp[-1] == c.value
    assert np.array_equal(oci, ci) and np.array_equal(ova.view(np.uint32), va.view(np.uint32))


def test_tcsr_builder_matches_reference_fixture(oracle, golden):
    """tcsr.cpp:5-38 -- blk_idx, bitmaps and values bit for bit (32-aligned fixtures)."""
    if golden.tcsr is None:
        pytest.skip("fixture is not 32-aligned: the reference's TCSR is undefined for it")
    bi, bm, va = oracle.tcsr_from_dense(golden.A)
    rbi, rbm, rva = golden.tcsr
    assert len(rbi) == (golden.M // 32) * (golden.N // 32) + 1
    assert np.array_equal(bi, rbi) and np.array_equal(bm, rbm)
    assert np.array_equal(va.view(np.uint32), rva.view(np.uint32))
    # the same nonzeros as the CSR of the same matrix, in a different order
    assert len(va) == len(golden.vals) and np.array_equal(np.sort(va), np.sort(golden.vals))


# ---- the bitmap formats of the wsp / awsp / awsp_ref / asp launchers --------------------------------
def decode_bitmap_format(fmt, M, N, bitmaps, vals, stats):
    """Dense A[M][N] back from a reference-layout (bitmaps, vals, stats): the inverse of wsp.cpp:3-40,
    awsp.cpp:3-49, awsp_ref.cpp:4-58 written from the layout description alone (numpy, no oracle code)."""
    bits = np.unpackbits(bitmaps.view(np.uint8), bitorder="little").astype(bool)      # bit k of the stream
    A = np.zeros((M, N), np.float32)
    if fmt == "wsp":                 # bit i*M + j <-> (input j, output i); column i's values at i*nz_max_m
        nzm = int(stats[0])
        occ = bits.reshape(N, M)
        for i in range(N):
            js = np.flatnonzero(occ[i])
            A[js, i] = vals[i * nzm:i * nzm + len(js)]
            assert np.all(vals[i * nzm + len(js):(i + 1) * nzm] == 0)                # zero padding
    elif fmt == "awsp":              # block b = strip*(M/32) + rowblock; word 32b + r, bit c
        nbk = int(stats[0])
        occ = bits.reshape(N // 32, M // 32, 32, 32)
        for s_ in range(N // 32):
            for rb in range(M // 32):
                b = s_ * (M // 32) + rb
                r, c = np.nonzero(occ[s_, rb])
                A[rb * 32 + r, s_ * 32 + c] = vals[b * nbk:b * nbk + len(r)]
                assert np.all(vals[b * nbk + len(r):(b + 1) * nbk] == 0)
    elif fmt == "awsp_ref":          # word s*M + j, bit c; (strip, quarter) streams at s*off[3] + off[q-1]
        off = [int(v) for v in stats]
        occ = bits.reshape(N // 32, M, 32)
        Q = M // 4
        for s_ in range(N // 32):
            for q in range(4):
                r, c = np.nonzero(occ[s_, q * Q:(q + 1) * Q])
                base = s_ * off[3] + (off[q - 1] if q else 0)
                A[q * Q + r, s_ * 32 + c] = vals[base:base + len(r)]
                assert np.all(vals[base + len(r):s_ * off[3] + off[q]] == 0)
    return A


@pytest.mark.parametrize("fmt", ["wsp", "awsp", "awsp_ref"])
def test_bitmap_format_builders_match_reference_fixture(oracle, golden, fmt):
    """Bitmaps, padded values and the exposed statistics (nz_max_m / nz_bk_max_ / warp_nz_offset_[4]) bit for bit
    against the arrays the reference's own classes produced (tests/golden/make_golden.py through oracle/_ref)."""
    if fmt not in golden.bitmap:
        pytest.skip("fixture is not 32-aligned: the reference's bitmap formats are undefined for it")
    rbm, rva, rst = golden.bitmap[fmt]
    bm, va, st = oracle.bitmap_from_dense(fmt, golden.A)
    assert np.array_equal(st, rst), (st, rst)
    assert np.array_equal(bm, rbm)
    assert len(va) == len(rva) and np.array_equal(va.view(np.uint32), rva.view(np.uint32))
    # sizes as the reference defines them
    M, N = golden.M, golden.N
    assert len(rbm) == M * N // 32
    want = {"wsp": N * int(rst[0]), "awsp": (M // 32) * (N // 32) * int(rst[0]), "awsp_ref": (N // 32) * int(rst[3])}[fmt]
    assert len(rva) == want
    # and the layout means what DESIGN says: decoding the reference arrays gives the dense matrix back
    A = decode_bitmap_format(fmt, M, N, rbm, rva, rst)
    keep = golden.A != 0
    assert np.array_equal(A[keep].view(np.uint32), golden.A[keep].view(np.uint32)) and not A[~keep].any()


def test_asp_retiling_matches_reference_checksum(oracle, golden):
    if golden.asp_checksum is None:
        pytest.skip("fixture is not 32-aligned")
    _, av, _ = oracle.bitmap_from_dense("asp", golden.A)
    u = av.view(np.uint32).astype(np.uint64)
    got = [av.size, int(u.sum() & 0xFFFFFFFFFFFF),
           int((u * (np.arange(av.size, dtype=np.uint64) % 65521 + 1)).sum() & 0xFFFFFFFFFFFF)]
    assert got == [int(v) for v in golden.asp_checksum]


@pytest.mark.skipif(not REF_LIB.exists(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("shape,zero", [((64, 96), 0.5), ((128, 32), 0.97), ((32, 160), 0.0), ((256, 64), 1.0)])
def test_bitmap_format_builders_match_live_reference(oracle, shape, zero):
    """The reference's WSPMatrix / AWSPMatrix / AWSPRefMatrix / ASPMatrix run here on fresh inputs."""
    lib = ctypes.CDLL(str(REF_LIB))
    lib.ref_fmt_build.restype = ctypes.c_void_p
    lib.ref_fmt_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    lib.ref_fmt_copy.argtypes = [ctypes.c_void_p] * 3
    lib.ref_fmt_free.argtypes = [ctypes.c_void_p]
    rng = np.random.Generator(np.random.PCG64(shape[0] * 1000 + shape[1]))
    A = rng.uniform(-1, 1, size=shape).astype(np.float32)
    A[rng.random(size=shape) < zero] = 0.0
    A.flat[::5] = -0.0
    M, N = shape
    for kind, fmt in enumerate(["wsp", "awsp", "awsp_ref", "asp"]):
        nb, nv = ctypes.c_int(), ctypes.c_int()
        st = np.zeros(4, np.int32)
        h = lib.ref_fmt_build(kind, M, N, A.ctypes.data, ctypes.byref(nb), ctypes.byref(nv), st.ctypes.data)
        rbm = np.empty(nb.value, np.uint32); rva = np.empty(nv.value, np.float32)
        lib.ref_fmt_copy(h, rbm.ctypes.data, rva.ctypes.data)
        lib.ref_fmt_free(h)
        bm, va, ost = oracle.bitmap_from_dense(fmt, A)
        assert np.array_equal(ost, st), (fmt, ost, st)
        assert np.array_equal(bm, rbm), fmt
        assert len(va) == len(rva) and np.array_equal(va.view(np.uint32), rva.view(np.uint32)), fmt


def test_golden_y_is_not_the_oracle_checking_itself(golden):
    """VERDICT weak-1: y_dense comes from a statement of SgemvCPU that shares no code with oracle/."""
    assert golden.y_source.startswith("numpy_sgemv")
