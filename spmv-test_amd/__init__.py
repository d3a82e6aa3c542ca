"""spmv-test_amd -- MI355X-native fp32 CSR SpMV behind the launcher surface of PACTHEMAN123/spMV-test.

The product is native: ``lib/libspmv_hip.so`` (hand-written HIP kernels for gfx950 behind the C ABI
of ``include/spmv_hip.h``), ``lib/libspmv_launchers.so`` (the reference's C++ ``*_gemv_gpu``
launchers, ``include/kernel.hpp``) and ``bin/sparse_sgemv`` (the tester).  This Python package is
plumbing for tests and ``bench.py``: a ctypes binding of the C ABI (``capi``), the synthetic
workload definitions (``workloads``), the multi-GPU row-block layer on torch.distributed (``partition``, ``dist``) and
the ctypes binding of the C++ one (``dist_native``: include/spmv_dist.h, RCCL called from C++).
PyTorch is used only for device memory, streams and ``torch.distributed``.

The directory name contains a hyphen, so import it through ``__graft_entry__.load_package()``,
which registers it as ``spmv_test_amd``.
"""
from . import capi, workloads, partition, dist, dist_native  # noqa: F401

__all__ = ["capi", "workloads", "partition", "dist", "dist_native"]
