"""On-disk formats (row f-4): the Matrix Market reader and the binary CSR container of the tester
executable.  Parsing is host code (runs here); the multiply through the C ABI is GPU-marked."""
import subprocess

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp


def _write_mtx(tmp_path, A, name="a.mtx", symmetry="general"):
    p = tmp_path / name
    scipy.io.mmwrite(str(p), A, symmetry=symmetry)
    return p


def _run(pkg, *args):
    return subprocess.run([str(pkg.capi.TESTER_PATH), *map(str, args)], capture_output=True, text=True, timeout=600)


def test_matrix_market_parse_and_binary_round_trip(pkg, tmp_path):
    rng = np.random.Generator(np.random.PCG64(5))
    A = sp.random(300, 200, density=0.05, random_state=rng, dtype=np.float64).tocsr()
    mtx = _write_mtx(tmp_path, A)
    binp = tmp_path / "a.csrbin"
    r = _run(pkg, "--mtx", mtx, "--parse-only", "--save", binp)
    assert r.returncode == 0 and f"matrix 300 x 200, nnz {A.nnz}" in r.stdout
    raw = binp.read_bytes()
    assert raw[:8] == b"SPMVCSR1"
    rows, cols, nnz = np.frombuffer(raw, np.int64, 3, 8)
    assert (rows, cols, nnz) == (300, 200, A.nnz)
    off = 32
    rp = np.frombuffer(raw, np.int32, rows + 1, off); off += 4 * (rows + 1)
    ci = np.frombuffer(raw, np.int32, nnz, off); off += 4 * nnz
    va = np.frombuffer(raw, np.float32, nnz, off)
    A.sort_indices()
    assert np.array_equal(rp, A.indptr) and np.array_equal(ci, A.indices)
    assert np.array_equal(va, A.data.astype(np.float32))
    r2 = _run(pkg, "--csrbin", binp, "--parse-only")
    assert r2.returncode == 0 and f"nnz {A.nnz}" in r2.stdout


def test_symmetric_and_pattern_files(pkg, tmp_path):
    (tmp_path / "s.mtx").write_text("%%MatrixMarket matrix coordinate real symmetric\n% c\n4 4 5\n"
                                    "1 1 2.0\n2 1 -1.0\n3 3 4.0\n4 2 0.5\n4 4 1.5\n")
    assert "matrix 4 x 4, nnz 7" in _run(pkg, "--mtx", tmp_path / "s.mtx", "--parse-only").stdout
    (tmp_path / "p.mtx").write_text("%%MatrixMarket matrix coordinate pattern general\n3 5 4\n1 1\n1 5\n3 2\n3 2\n")
    assert "matrix 3 x 5, nnz 3" in _run(pkg, "--mtx", tmp_path / "p.mtx", "--parse-only").stdout   # duplicate summed
    (tmp_path / "bad.mtx").write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    r = _run(pkg, "--mtx", tmp_path / "bad.mtx", "--parse-only")
    assert r.returncode != 0 and "coordinate" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["scalar", "tiled"])
def test_multiply_a_matrix_from_disk(pkg, gpu, tmp_path, variant):
    rng = np.random.Generator(np.random.PCG64(6))
    A = (sp.random(5000, 7000, density=0.002, random_state=rng, dtype=np.float64) +
         sp.diags(rng.uniform(-1, 1, 5000), 0, shape=(5000, 7000))).tocsr()
    mtx = _write_mtx(tmp_path, A)
    out = tmp_path / "y.txt"
    r = _run(pkg, "--mtx", mtx, "--variant", variant, "--out", out)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "========== OK ===========" in r.stdout
    y = np.loadtxt(out, dtype=np.float64)
    ref = A.astype(np.float32).astype(np.float64) @ np.ones(7000)
    mag = abs(A.astype(np.float32)).astype(np.float64) @ np.ones(7000)
    assert np.all(np.abs(y - ref) <= 1e-5 * mag + 1e-30)
