"""Shared fixtures.  `-m "not gpu"` runs here on CPU; `-m gpu` runs on a real MI355X."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"
GOLDEN_NAMES = sorted(p.stem for p in GOLDEN_DIR.glob("*.npz"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native libraries exist (they travel prebuilt to the GPU box)."""
    pkg = ge.load_package()
    if not pkg.capi.LIB_PATH.exists() or not (ROOT / "oracle" / "liboracle.so").exists():
        ge.build()
    return pkg


@pytest.fixture(scope="session")
def pkg(built):
    return built


@pytest.fixture(scope="session")
def oracle(built):
    return ge.load_oracle()


class Golden:
    """One committed fixture: reference-built CSR (N row pointers, no sentinel) + inputs + y."""

    def __init__(self, name):
        z = np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False)
        self.name = name
        self.M, self.N = int(z["M"]), int(z["N"])
        self.x = z["x"]
        self.ref_row_ptrs = z["ref_row_ptrs"]
        self.col_idx = z["ref_col_idxs"]
        self.vals = z["ref_vals"]
        self.y = z["y_dense"]
        # our layout: rows+1 entries with the nnz sentinel (csr_naive.cu:15 substitutes it)
        self.row_ptr = np.concatenate([self.ref_row_ptrs, [len(self.vals)]]).astype(np.int32)
        self.tcsr = None
        if "tcsr_blk_idx" in z.files:      # reference TCSRMatrix arrays (32-aligned fixtures only)
            self.tcsr = (z["tcsr_blk_idx"], z["tcsr_bitmaps"], z["tcsr_vals"])
        # reference WSPMatrix / AWSPMatrix / AWSPRefMatrix arrays + the statistics the classes expose (32-aligned only)
        self.bitmap = {}
        for key in ("wsp", "awsp", "awsp_ref"):
            if f"{key}_bitmaps" in z.files:
                self.bitmap[key] = (z[f"{key}_bitmaps"], z[f"{key}_vals"], z[f"{key}_stats"])
        self.asp_checksum = z["asp_checksum"] if "asp_checksum" in z.files else None
        self.y_source = str(z["y_source"])
        if "A" in z.files:
            self.A = z["A"]
        else:
            # exact reconstruction: the reference CSR holds every element with value != 0.0f
            A = np.zeros((self.M, self.N), np.float32)
            rows = np.repeat(np.arange(self.N), np.diff(self.row_ptr))
            A[self.col_idx, rows] = self.vals
            self.A = A


@pytest.fixture(scope="session", params=GOLDEN_NAMES)
def golden(request):
    return Golden(request.param)


def load_golden(name):
    return Golden(name)


@pytest.fixture(scope="session")
def gpu(pkg):
    """torch device for -m gpu tests; the HIP library must be the thing that runs."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test on a box without a GPU"
    assert pkg.capi.device_count() >= 1
    return torch.device("cuda:0")
