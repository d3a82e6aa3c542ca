// tools/ubench_gather_lines.hip -- what does a 4-byte gather instruction cost as a function of how many DISTINCT
// 128-byte lines of x its 64 lanes touch?  (round 3: the question behind column-sorted tiles for SPMV_PANEL.)
//
// tools/ubench_gather.hip measured the worst case: 64 lanes, 64 random lines: 280 G gathers/s from an L2-resident
// table.  Here every wave instruction picks a random window of the table and its lanes touch L lines of it:
//   mode 0 ("run"):    L CONSECUTIVE lines starting at a random line (what a column-sorted stream does), lane l reads a
//                      random word of line base + l*L/64;
//   mode 1 ("spread"): L random lines anywhere in the table, lane l reads a random word of line[l mod L].
// Reported: G gathers/s chip-wide and G line-requests/s (= gathers x L/64).  If the rate in gathers/s scales like 64/L
// the price is per line (TA/L2 request bound) and sorting nonzeros by line pays; if it stays flat the price is per lane.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/ubench_gather_lines tools/ubench_gather_lines.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s @%d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned mix(unsigned s) { s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16; return s; }

template <int MODE>
__global__ __launch_bounds__(256) void k_gather_lines(int iters, unsigned line_mask, int L, const float *__restrict__ x,
                                                      float *__restrict__ out)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    unsigned ws = wave * 2654435761u + 12345u;      // wave-uniform stream: picks the windows
    unsigned ls = (blockIdx.x * 256u + threadIdx.x) * 40503u + 7u;   // per-lane stream: the word inside the line
    const unsigned my = MODE == 0 ? (lane * (unsigned)L) >> 6 : lane % (unsigned)L;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        unsigned idx[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ws = ws * 1664525u + 1013904223u;
            ls = ls * 22695477u + 1u;
            unsigned line;
            if (MODE == 0) line = ((mix(ws) & line_mask) + my) & line_mask;
            else line = mix(ws + my * 0x9e3779b9u) & line_mask;
            idx[k] = line * 32u + ((ls >> 20) & 31u);
        }
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = x[idx[k]];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    if (acc == 12345.678f) out[0] = acc;   // never true: keeps the loads
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
    for (int mode : {0, 1})
        for (int64_t kib : {512ll, 4096ll, 65536ll, 524288ll})
            for (int wg_per_cu : {1, 4})
                for (int L : {64, 32, 16, 8, 4, 2, 1}) {
                    const unsigned line_mask = (unsigned)(kib * 8 - 1);   // 128-byte lines in the table
                    const int iters = 32, grid = cus * wg_per_cu;
                    auto launch = [&]() {
                        if (mode == 0) hipLaunchKernelGGL(k_gather_lines<0>, dim3(grid), dim3(256), 0, 0, iters, line_mask, L, x, out);
                        else hipLaunchKernelGGL(k_gather_lines<1>, dim3(grid), dim3(256), 0, 0, iters, line_mask, L, x, out);
                    };
                    launch();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int r = 0; r < 4; ++r) launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
                    const double g = (double)grid * 256 * iters * 16;
                    printf("{\"test\": \"gather_lines\", \"mode\": \"%s\", \"table_KiB\": %lld, \"waves_per_cu\": %d, \"lines_per_instr\": %d, \"ms\": %.4f, "
                           "\"G_gathers_per_s\": %.1f, \"G_line_requests_per_s\": %.1f}\n",
                           mode == 0 ? "run" : "spread", (long long)kib, wg_per_cu * 4, L, ms, g / ms / 1e6, g * L / 64 / ms / 1e6);
                }
    return 0;
}
