// tools/ubench_lds_atomic.hip -- how fast are LDS float adds?  (ds_add_f32 vs plain read-add-write)
// Every CU runs `waves` wavefronts; each lane performs `iters` updates of a 32 KiB per-workgroup LDS array at
//   mode 0: lane-consecutive addresses (conflict-free)        mode 1: pseudo-random addresses
// with   op 0: atomicAdd (ds_add_f32, no return)   op 1: plain read, add, write   op 2: ds_read_b32 only.
// Prints updates per clock per CU (2.4 GHz nominal) and G updates/s chip-wide.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_lds_atomic tools/ubench_lds_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kWords = 8192;

template <int OP>
__global__ void k(int iters, int mode, float *out)
{
    __shared__ float a[kWords];
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) a[i] = 0.0f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.0f;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) idx = (idx + blockDim.x) & (kWords - 1);
        else { s = s * 1664525u + 1013904223u; idx = (s >> 9) & (kWords - 1); }
        if (OP == 0) atomicAdd(&a[idx], 1.0f);
        else if (OP == 1) a[idx] = a[idx] + 1.0f;
        else acc += a[idx];
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = a[1] + a[77] + acc;
}

int main(int argc, char **argv)
{
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int waves : {4, 8, 16}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int op = 0; op < 3; ++op) {
                auto launch = [&]() {
                    dim3 grid(cus), block(waves * 64);
                    if (op == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, iters, mode, out);
                    else if (op == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, iters, mode, out);
                    else hipLaunchKernelGGL(k<2>, grid, block, 0, 0, iters, mode, out);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < 5; ++r) launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                ms /= 5;
                const double upd = (double)cus * waves * 64 * iters;
                printf("{\"waves_per_cu\": %d, \"addresses\": \"%s\", \"op\": \"%s\", \"ms\": %.4f, \"G_updates_per_s\": %.1f, \"updates_per_clk_per_cu_at_2.4GHz\": %.2f}\n",
                       waves, mode ? "random" : "consecutive", op == 0 ? "ds_add_f32" : (op == 1 ? "read-add-write" : "ds_read_b32"),
                       ms, upd / ms / 1e6, upd / cus / (ms * 1e-3 * 2.4e9));
            }
        }
    }
    return 0;
}
