"""Synthetic workloads: "synthetic CSR of stated (M, N, nnz)" for the BASELINE.json configs.

The reference has no matrix files and no reproducible inputs (its tester seeds mt19937 from
random_device, /root/reference/src/tester.cpp:107,155) and its dense boundary cannot even hold
config 2 (1Mi x 1Mi dense = 4 TiB).  So the workloads are defined here, from integer arithmetic
only, such that

* row LENGTHS are a pure function of (spec, global row) and are computed on the host with numpy
  (this file) -- every 65 536-row block holds exactly ``mean * 65 536`` nonzeros, so totals are
  exact and any row block is a self-contained shard;
* the ELEMENTS of a row (columns, values) are a pure function of (seed, global row, k, length,
  band) and are produced on the device by ``spmv_synth_fill`` (csrc/kernels_synth.hip); the same
  specification is restated on the host in oracle/spmv_oracle.c, so tests regenerate any row
  block on the CPU and compare bit for bit.

Column distributions (DESIGN.md "Synthetic workloads"):
  band == 0   every row draws its columns from all of [0, cols)       ("uniform")
  band  > 0   from a window of max(band, 8*len) columns around the diagonal ("banded")
In both cases column k of a row of length L falls uniformly inside stratum k of L equal strata of
the window, which yields strictly ascending columns (as CSRMatrix emits them,
matrix_csr.cpp:12-20) without a sort.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

import numpy as np

BLOCK_ROWS = 1 << 16          # row lengths are normalised per block of this many rows
DEFAULT_SEED = 20251031       # date of the reference snapshot

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def _row_hash(seed: int, rows: np.ndarray, salt: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x9E3779B97F4A7C15) * (rows.astype(np.uint64) + np.uint64(1))
        z = _mix64(z)
        z = z + np.uint64(0xD1B54A32D192ED03) * np.uint64(salt + 1)
        return _mix64(z)


@dataclass(frozen=True)
class Workload:
    name: str
    rows: int
    cols: int
    dist: str            # "const" | "powerlaw" | "mixed"
    mean: int            # exact mean nonzeros per row (per 65 536-row block)
    band: int = 0        # 0 = uniform columns, >0 = diagonal band of that many columns
    seed: int = DEFAULT_SEED

    @property
    def nnz(self) -> int:
        return self.mean * self.rows

    def describe(self) -> str:
        cols = "uniform columns" if self.band == 0 else f"banded columns (band {self.band})"
        return (f"{self.name}: {self.rows} x {self.cols}, nnz {self.nnz}, row lengths '{self.dist}' "
                f"mean {self.mean}, {cols}, seed {self.seed}")


# Classes of the "mixed" distribution: (probability in 1/10000, shortest, longest).
# Raw mean 15.917; the per-block normalisation tops rows up by one to reach exactly 16.
_MIXED_CLASSES = ((6000, 1, 8), (3000, 9, 32), (950, 33, 96), (49, 97, 203), (1, 1025, 3071))


def _raw_lengths(w: Workload, rows: np.ndarray) -> np.ndarray:
    if w.dist == "const":
        return np.full(rows.shape, w.mean, dtype=np.int64)
    h = _row_hash(w.seed, rows, 0x11)
    if w.dist == "powerlaw":
        # discrete Pareto, alpha = 2: L = floor(Lmin / (1-u)), u = h32 / 2^32, truncated at 65 536.
        # Lmin = mean * 93/1024 keeps the raw mean just below `mean` (E[L] ~ Lmin*(1+ln(Lmax/Lmin)) - 1/2).
        h32 = h >> np.uint64(32)
        num = np.uint64(w.mean * 93) << np.uint64(32)
        den = (np.uint64(1 << 32) - h32) * np.uint64(1024)
        L = (num // den).astype(np.int64)
        return np.clip(L, 1, 65536)
    if w.dist == "mixed":
        if w.mean != 16:
            raise ValueError("the 'mixed' distribution is defined for mean 16")
        sel = (h >> np.uint64(32)) % np.uint64(10000)
        h2 = _row_hash(w.seed, rows, 0x22)
        L = np.zeros(rows.shape, dtype=np.int64)
        lo_p = 0
        for p, lo, hi in _MIXED_CLASSES:
            m = (sel >= np.uint64(lo_p)) & (sel < np.uint64(lo_p + p))
            L[m] = lo + (h2[m] % np.uint64(hi - lo + 1)).astype(np.int64)
            lo_p += p
        return L
    raise ValueError(f"unknown row-length distribution {w.dist!r}")


def _block_lengths(w: Workload, b: int) -> np.ndarray:
    """Lengths of the rows of block b (rows [b*BLOCK_ROWS, ...)), normalised to the exact mean."""
    r0 = b * BLOCK_ROWS
    r1 = min(w.rows, r0 + BLOCK_ROWS)
    rows = np.arange(r0, r1, dtype=np.int64)
    L = _raw_lengths(w, rows)
    if w.dist == "const":
        return L
    target = w.mean * (r1 - r0)
    delta = target - int(L.sum())
    n = r1 - r0
    if delta >= 0:
        # top up: +q on every row, +1 more on the first r rows (q is 0 for the shipped specs)
        q, r = divmod(delta, n)
        L += q
        L[:r] += 1
    else:
        # trim the longest rows first (deterministic order), never below one
        order = np.argsort(-L, kind="stable")
        need = -delta
        for idx in order:
            take = min(need, int(L[idx]) - 1)
            L[idx] -= take
            need -= take
            if need == 0:
                break
        if need:
            raise ValueError("cannot normalise block: mean too small")
    return L


def row_lengths(w: Workload, row0: int = 0, n: int | None = None) -> np.ndarray:
    """int64 lengths of global rows [row0, row0+n)."""
    if n is None:
        n = w.rows - row0
    if not (0 <= row0 and row0 + n <= w.rows):
        raise ValueError("row range outside the matrix")
    if w.dist == "const":
        return np.full(n, w.mean, dtype=np.int64)
    out = np.empty(n, dtype=np.int64)
    b0, b1 = row0 // BLOCK_ROWS, (row0 + n + BLOCK_ROWS - 1) // BLOCK_ROWS
    for b in range(b0, b1):
        L = _block_lengths(w, b)
        g0 = b * BLOCK_ROWS
        s = max(row0, g0)
        e = min(row0 + n, g0 + len(L))
        out[s - row0:e - row0] = L[s - g0:e - g0]
    return out


def row_ptr(w: Workload, row0: int = 0, n: int | None = None) -> np.ndarray:
    """int32 row_ptr (n+1 entries, rebased to 0) of the shard holding global rows [row0, row0+n)."""
    L = row_lengths(w, row0, n)
    rp = np.zeros(len(L) + 1, dtype=np.int64)
    np.cumsum(L, out=rp[1:])
    if rp[-1] >= (1 << 31):
        raise ValueError("shard has >= 2^31 nonzeros: split it into more row blocks")
    return rp.astype(np.int32)


def algorithmic_bytes(rows: int, cols: int, nnz: int) -> int:
    """Bytes one SpMV must move (SURVEY.md section 8d / BASELINE.md section 4):
    vals + col_idx (8/nnz), row_ptr int32 (rows+1), y written once, x read once."""
    return 8 * nnz + 4 * (rows + 1) + 4 * rows + 4 * cols


def flops(nnz: int) -> int:
    return 2 * nnz


def dense_random(M: int, N: int, zero_fraction: float, seed: int):
    """Dense row-major A[M][N] and x[M] in the style of the reference tester
    (tester.cpp:103-121,151-167: P(zero) = zero_fraction, nonzeros ~ U(-1,1)), reproducible."""
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.uniform(-1.0, 1.0, size=(M, N)).astype(np.float32)
    A[rng.random(size=(M, N)) < zero_fraction] = 0.0
    x = rng.uniform(-1.0, 1.0, size=M).astype(np.float32)
    x[rng.random(size=M) < 0.5] = 0.0
    return A, x



def stencil7(n: int):
    """Host-built CSR of a 7-point 3-D stencil on n^3 unknowns (columns r, r+-1, r+-n, r+-n^2 where they exist:
    three narrow clusters of columns 2 n^2 apart -- the structure a single contiguous x window cannot cover).
    Returns (N, row_ptr int32, col_idx int32 ascending per row, vals float32)."""
    N = n ** 3
    r = np.arange(N, dtype=np.int64)
    i, j, k = r // (n * n), (r // n) % n, r % n
    offs = [(-n * n, i > 0), (-n, j > 0), (-1, k > 0), (0, np.ones(N, bool)), (1, k < n - 1), (n, j < n - 1), (n * n, i < n - 1)]
    cnt = sum(m.astype(np.int64) for _, m in offs)
    rp = np.zeros(N + 1, np.int64)
    np.cumsum(cnt, out=rp[1:])
    ci = np.empty(int(rp[-1]), np.int32)
    va = np.empty(int(rp[-1]), np.float32)
    pos = rp[:-1].copy()
    scale = (0.5 + (r * 2654435761 % 1000003).astype(np.float32) / 1000003.0).astype(np.float32)
    for o, m in offs:            # ascending offsets -> ascending columns inside a row
        ci[pos[m]] = (r[m] + o).astype(np.int32)
        va[pos[m]] = (6.0 if o == 0 else -1.0) * scale[m]
        pos[m] += 1
    return N, rp.astype(np.int32), ci, va

# ---- the BASELINE.json configs ------------------------------------------------------------------
Mi = 1 << 20
CONFIGS = {
    # c1 is the dense tester path (dense_random + the launchers); no Workload object
    "c2": Workload("c2", 1 * Mi, 1 * Mi, "const", 16),
    "c3": Workload("c3", 4 * Mi, 4 * Mi, "powerlaw", 32),
    "c4": Workload("c4", 16 * Mi, 16 * Mi, "mixed", 16),
    # c5 is c4's row-length law on (N_gpus * 16Mi)^2, one 16Mi-row block per GPU
}
C4_BAND = 8192     # headline column law: half-bandwidth 4096 = the 2-D 4096 x 4096 mesh of 16Mi unknowns


def config(name: str, band: int | None = None, scale: float = 1.0) -> Workload:
    """A BASELINE config, optionally with banded columns and/or scaled-down rows (tests)."""
    w = CONFIGS[name]
    if scale != 1.0:
        r = max(BLOCK_ROWS, int(w.rows * scale) // BLOCK_ROWS * BLOCK_ROWS)
        w = replace(w, rows=r, cols=r, name=f"{w.name}@{r}")
    if band is not None:
        w = replace(w, band=band)
    return w


def c5(n_gpus: int, band: int = 0, rows_per_gpu: int = 16 * Mi) -> Workload:
    """Config 5 generalised: (n_gpus*16Mi)^2, 256Mi nonzeros per 16Mi-row block (weak scaling)."""
    if rows_per_gpu % BLOCK_ROWS:
        raise ValueError("rows_per_gpu must be a multiple of 65 536")
    r = n_gpus * rows_per_gpu
    return Workload(f"c5x{n_gpus}", r, r, "mixed", 16, band=band)
