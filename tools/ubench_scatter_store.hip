// tools/ubench_scatter_store.hip -- development micro-benchmark (not part of the product or the tests): what do stores
// cost on gfx950 when a stream is written out in RUNS of R floats at scattered places (the product launch of the binned
// layout's scattered flavour, csrc/kernels_binned.hip)?  2^28 floats (1 GiB) read linearly and written
//   lin16   linearly, 16 bytes per lane                                  (the fetching flavour's product launch)
//   lin4    linearly, 4 bytes per lane, 64 consecutive floats per store
//   run4    in runs of R floats, run r at place perm(r): 4 bytes per lane, 64 consecutive entries per store (what ships)
//   run16   the same runs, 16 bytes per lane (R a multiple of 4): a lane's four entries lie in one run
//   near    with "near" = 1 the runs of one workgroup land next to each other's neighbours (perm keeps r's low bits): the
//           XCD-contiguous order; 0 = anywhere
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_scatter_store.hip -o tools/bin/ubench_scatter_store
//   run  : tools/bin/ubench_scatter_store            (prints one JSON line per case)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s @%d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

using f4 = float __attribute__((ext_vector_type(4)));
constexpr int BS = 1024;                 // threads per workgroup, 8 floats each and trip
constexpr int64_t N = 1ll << 28;

// place of run r among nruns = 2^k runs: a bijection (odd multiplier modulo 2^k)
__device__ __forceinline__ uint32_t place(uint32_t r, uint32_t mask, int near)
{
    if (near) return ((r >> 5) * 0x9E3779B1u & (mask >> 5)) << 5 | (r & 31u);   // 32 consecutive runs stay together
    return (r * 0x9E3779B1u) & mask;
}

__global__ __launch_bounds__(BS) void k_lin16(const float *__restrict__ in, float *__restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * BS + threadIdx.x) * 8;
    const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + i));
    const f4 b = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + i + 4));
    *reinterpret_cast<f4 *>(out + i) = a * 2.0f;
    *reinterpret_cast<f4 *>(out + i + 4) = b * 2.0f;
}
// interleaved like the product launch: lane l of a wavefront loads 32 bytes of its 512-float block, store j covers floats
// 64 j + l of the block (the values are not transposed: timing only)
template <bool RUNS>
__global__ __launch_bounds__(BS) void k_store4(const float *__restrict__ in, float *__restrict__ out, int R, uint32_t mask, int near)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * BS + threadIdx.x) * 8;
    const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + i));
    const f4 b = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + i + 4));
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const int64_t blk = i & ~511ll;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t e = blk + 64 * j + lane;
        int64_t d = e;
        if (RUNS) {
            const uint32_t r = (uint32_t)(e / R);
            d = (int64_t)place(r, mask, near) * R + (e - (int64_t)r * R);
        }
        out[d] = v[j] * 2.0f;
    }
}
__global__ __launch_bounds__(BS) void k_run16(const float *__restrict__ in, float *__restrict__ out, int R, uint32_t mask, int near)
{
    const int64_t i = ((int64_t)blockIdx.x * BS + threadIdx.x) * 8;
    const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + i));
    const f4 b = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + i + 4));
    const uint32_t r0 = (uint32_t)(i / R), r1 = (uint32_t)((i + 4) / R);
    const int64_t d0 = (int64_t)place(r0, mask, near) * R + (i - (int64_t)r0 * R);
    const int64_t d1 = (int64_t)place(r1, mask, near) * R + (i + 4 - (int64_t)r1 * R);
    *reinterpret_cast<f4 *>(out + d0) = a * 2.0f;
    *reinterpret_cast<f4 *>(out + d1) = b * 2.0f;
}

int main()
{
    float *in = nullptr, *out = nullptr;
    CK(hipMalloc((void **)&in, sizeof(float) * N));
    CK(hipMalloc((void **)&out, sizeof(float) * N));
    CK(hipMemset(in, 0, sizeof(float) * N));
    CK(hipMemset(out, 0, sizeof(float) * N));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, int R, int near, int64_t count, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        printf("{\"case\": \"%s\", \"run_floats\": %d, \"near\": %d, \"ms\": %.4f, \"GBps_read_plus_write\": %.1f}\n", name, R, near, ms,
               2.0 * sizeof(float) * (double)count / ms / 1e6);
    };
    const int grid = (int)(N / (BS * 8));
    timed("lin16", 0, 0, N, [&] { k_lin16<<<grid, BS>>>(in, out); });
    timed("lin4", 0, 0, N, [&] { k_store4<false><<<grid, BS>>>(in, out, 1, 0, 0); });
    // runs of R floats, nruns a power of two with nruns * R <= N (the kernels cover the first nruns * R floats' worth of grid)
    for (int R : {16, 28, 32, 64, 228, 256, 1024}) {
        uint32_t nruns = 1;
        while ((int64_t)nruns * 2 * R <= N) nruns *= 2;
        const int g = (int)((int64_t)nruns * R / (BS * 8));
        for (int near = 0; near < 2; ++near) {
            timed("run4", R, near, (int64_t)nruns * R, [&] { k_store4<true><<<g, BS>>>(in, out, R, nruns - 1, near); });
            if (R % 4 == 0) timed("run16", R, near, (int64_t)nruns * R, [&] { k_run16<<<g, BS>>>(in, out, R, nruns - 1, near); });
        }
    }
    return 0;
}
