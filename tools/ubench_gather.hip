// tools/ubench_gather.hip -- the wall behind uniform-random columns: how many independent 4-byte gathers per second
// does the chip sustain, by where the table lives, and what would staging the table through LDS cost instead?
//   (a) gather: every lane issues `iters` x 16 independent loads x[idx], idx pseudo-random over a table of T bytes that
//       every workgroup shares (T = 32 KiB: L1-sized ... 512 KiB: the panel of SPMV_PANEL, L2-resident ... 64 MiB: x of
//       config 4, Infinity-Cache-resident ... 512 MiB: x of config 5, HBM).  Reported: G gathers/s chip-wide and the
//       L2->L1 bytes that implies at one 128-byte line per gather.
//   (b) stage: every workgroup copies the same T bytes global -> LDS with 16-byte LDS-DMA loads, 128 KiB at a time (what
//       an "x panel in LDS" kernel has to do once per (row block, panel)); reported: TB/s chip-wide.
// From (a): a gather that misses L1 costs one L2 request and moves one line, whatever the cache policy; the chip does
// ~0.2 T of them per second.  268M nonzeros with uniform columns over 64 MiB of x therefore take >= 1.3 ms (22 % of the
// HBM roofline of config 4) unless the nonzeros of one row block hit each line several times -- they do not: a row
// block that fits LDS (<= 40 Ki rows per CU) puts 1.2 nonzeros on a line of x at config 4's density.
// From (b): staging moves T bytes per (row block, panel) to serve rows_in_block x T/4 x density nonzeros, i.e.
// 4/(rows_in_block x density) bytes per nonzero: 250 B at config 4 (16 Ki rows, density 1e-6), more than the 128-byte
// line of the plain gather.  DESIGN.md section 4 "Uniform columns".
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s @%d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_gather(int iters, unsigned mask, const float *__restrict__ x, float *__restrict__ out)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        unsigned idx[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { s = s * 1664525u + 1013904223u; idx[k] = (s >> 4) & mask; }
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = x[idx[k]];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    if (acc == 12345.678f) out[0] = acc;   // never true: keeps the loads
}

__global__ __launch_bounds__(1024) void k_stage(int reps, int64_t table_floats, const float *__restrict__ x, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kSlice = 32768;   // 128 KiB of LDS per pass
    float acc = 0.0f;
    for (int r = 0; r < reps; ++r)
        for (int64_t off = 0; off < table_floats; off += kSlice) {
            for (int i = threadIdx.x * 4; i < kSlice; i += 1024 * 4) __builtin_amdgcn_global_load_lds(x + off + i, lds + i, 16, 0, 0);
            __syncthreads();
            acc += lds[(threadIdx.x * 33 + r) & (kSlice - 1)];
            __syncthreads();
        }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    int dev = 0, cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t max_floats = 128ll << 20;   // 512 MiB
    float *x, *out;
    CK(hipMalloc(&x, sizeof(float) * max_floats));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0, sizeof(float) * max_floats));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int64_t kib : {32ll, 512ll, 4096ll, 65536ll, 524288ll}) {
        const unsigned mask = (unsigned)(kib * 256 - 1);
        for (int wg_per_cu : {4, 8}) {
            const int iters = 64, grid = cus * wg_per_cu;
            hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, iters, mask, x, out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, iters, mask, x, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double g = (double)grid * 256 * iters * 16;
            printf("{\"test\": \"gather\", \"table_KiB\": %lld, \"waves_per_cu\": %d, \"ms\": %.4f, \"G_gathers_per_s\": %.1f, \"TBs_at_128B_per_gather\": %.1f}\n",
                   (long long)kib, wg_per_cu * 4, ms, g / ms / 1e6, g * 128 / ms / 1e9);
        }
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stage), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int64_t kib : {512ll, 4096ll, 65536ll}) {
        const int64_t floats = kib * 256;
        const int reps = kib <= 4096 ? 16 : 1;
        hipLaunchKernelGGL(k_stage, dim3(cus), dim3(1024), 131072, 0, reps, floats, x, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stage, dim3(cus), dim3(1024), 131072, 0, reps, floats, x, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"test\": \"stage_to_lds\", \"table_KiB\": %lld, \"workgroups\": %d, \"ms\": %.4f, \"TBs_chip_wide\": %.2f, \"GBs_per_CU\": %.1f}\n",
               (long long)kib, cus, ms, (double)cus * reps * floats * 4 / ms / 1e9, (double)reps * floats * 4 / ms / 1e6);
    }
    return 0;
}
