// kernels_calib.hip -- measurement aids: kernels of KNOWN memory traffic and a marker dispatch.
//
// The north-star's figure is rocprofv3's FETCH_SIZE / WRITE_SIZE over the kernel time (the role of
// /root/reference/profile.sh:18-20: a profiler wrapped around the executable).  On gfx950 FETCH_SIZE tallies the
// 128-byte requests of a 16-byte-per-lane read at 64 bytes; "other access widths are uncalibrated"
// (/opt/skills/guides/MI355X_MICROARCH.md, HBM).  bench.py therefore launches these kernels in the SAME profiled
// process as the SpMV kernels and derives the correction per access class from them:
//   k_calib_stream   reads exactly `bytes` with 16-byte loads, all lanes, consecutive -- the class of every matrix
//                    stream and LDS-DMA window in this library;
//   k_calib_gather   4-byte loads, one lane per DISTINCT 128-byte line of a table far larger than every cache,
//                    neighbouring lanes far apart -- the class of x gathers that miss; `touch` = 1 reads word 0 of
//                    the line, 2 words 0 and 16 (both 64-byte halves), 4 words 0, 8, 16, 24 (all 32-byte sectors):
//                    FETCH_SIZE against the known line count says what one missing gather is tallied as and whether
//                    the memory side fetches lines, halves or sectors;
//   k_calib_store    writes exactly `bytes`, a dword or 16 bytes per lane, consecutive -- the class of the y writes
//                    (WRITE_SIZE against a known count);
//   k_marker         does nothing; its grid size carries an id, so that a counter file (one row per dispatch, in
//                    dispatch order) can be cut into the workloads of one process.
#include "spmv_internal.hpp"

namespace spmv {

using u4 = unsigned __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_calib_stream(const u4 *__restrict__ src, int64_t n16, float *__restrict__ sink)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const u4 v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) sink[0] = 1.0f;    // practically never: keeps the loads
}

__global__ __launch_bounds__(256) void k_calib_gather(const float *__restrict__ table, uint32_t line_mask, int64_t n_lines,
                                                      int touch, float *__restrict__ sink)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_lines) return;
    // multiplication by an odd number is a bijection modulo a power of two: every line exactly once
    const uint32_t line = ((uint32_t)g * 0x9E3779B1u + 0x7F4A7C15u) & line_mask;
    const float *p = table + (int64_t)line * 32;
    float acc = p[0];
    if (touch >= 2) acc += p[16];
    if (touch >= 4) acc += p[8] + p[24];
    if (acc == 12345.678f) sink[0] = acc;
}

// writes exactly `n` words (WIDTH = 1: a dword per lane, the way every kernel here writes y) or `n` 16-byte pieces
template <int WIDTH>
__global__ __launch_bounds__(256) void k_calib_store(float *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (WIDTH == 1) dst[i] = 1.0f;
        else reinterpret_cast<u4 *>(dst)[i] = u4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
    }
}

__global__ void k_marker() {}

}  // namespace spmv

extern "C" {

int spmv_calib_stream(const void *d_src, int64_t bytes, float *d_sink, void *stream)
{
    if (!d_src || !d_sink || bytes < 16 || (bytes & 15) || ((uintptr_t)d_src & 15)) {
        spmv::set_error("spmv_calib_stream: needs a 16-byte aligned buffer of a multiple of 16 bytes");
        return SPMV_ERR_INVALID;
    }
    int dev = 0;
    SPMV_HIP_TRY(hipGetDevice(&dev));
    const int grid = spmv::device_cus(dev) * 8;
    hipLaunchKernelGGL(spmv::k_calib_stream, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const spmv::u4 *)d_src, bytes / 16, d_sink);
    SPMV_HIP_TRY(hipGetLastError());
    return SPMV_OK;
}

int spmv_calib_gather(const float *d_table, int64_t table_lines, int64_t n_lines, int touch, float *d_sink, void *stream)
{
    if (!d_table || !d_sink || table_lines < 1 || (table_lines & (table_lines - 1)) || table_lines > (1ll << 32) ||
        n_lines < 1 || n_lines > table_lines || (touch != 1 && touch != 2 && touch != 4) || ((uintptr_t)d_table & 127)) {
        spmv::set_error("spmv_calib_gather: table_lines must be a power of two <= 2^32, 1 <= n_lines <= table_lines, touch 1|2|4, "
                        "the table 128-byte aligned");
        return SPMV_ERR_INVALID;
    }
    const int64_t grid = (n_lines + 255) / 256;
    hipLaunchKernelGGL(spmv::k_calib_gather, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, d_table,
                       (uint32_t)(table_lines - 1), n_lines, touch, d_sink);
    SPMV_HIP_TRY(hipGetLastError());
    return SPMV_OK;
}

int spmv_calib_store(float *d_dst, int64_t bytes, int width, void *stream)
{
    if (!d_dst || bytes < 16 || (bytes & 15) || ((uintptr_t)d_dst & 15) || (width != 4 && width != 16)) {
        spmv::set_error("spmv_calib_store: needs a 16-byte aligned buffer of a multiple of 16 bytes and width 4|16");
        return SPMV_ERR_INVALID;
    }
    int dev = 0;
    SPMV_HIP_TRY(hipGetDevice(&dev));
    const int grid = spmv::device_cus(dev) * 8;
    if (width == 4) hipLaunchKernelGGL(spmv::k_calib_store<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_dst, bytes / 4);
    else hipLaunchKernelGGL(spmv::k_calib_store<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_dst, bytes / 16);
    SPMV_HIP_TRY(hipGetLastError());
    return SPMV_OK;
}

int spmv_calib_marker(int id, void *stream)
{
    if (id < 1 || id > 65535) {
        spmv::set_error("spmv_calib_marker: id must be in [1, 65535]");
        return SPMV_ERR_INVALID;
    }
    hipLaunchKernelGGL(spmv::k_marker, dim3((unsigned)id), dim3(64), 0, (hipStream_t)stream);
    SPMV_HIP_TRY(hipGetLastError());
    return SPMV_OK;
}

}  // extern "C"
