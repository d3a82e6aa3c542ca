// tools/ubench_panel.hip -- development micro-benchmark: would a 2-D (row block x column panel) sweep
// make uniform-random columns L2-resident?  One workgroup per row block of RB rows keeps its y block in
// LDS and walks the column panels in order; all workgroups of an XCD walk in rough lockstep, so the XCD's
// L2 should hold the panel of x they are all gathering from.  Synthetic tiles: T nonzeros per (row block,
// panel), packed {row_local:14 | col_local:18}, random.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_panel.hip -o tools/bin/ubench_panel
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s @%d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; }
using u4 = unsigned __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

__global__ void k_gen(int64_t n, int rb_rows, int pcols, unsigned* packed, float* val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 7);
    unsigned row = (unsigned)(h % (unsigned)rb_rows);   // any row of the block
    unsigned col = (unsigned)((h >> 20) % (unsigned)pcols);
    packed[i] = (row << 18) | col;
    val[i] = (float)((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f);
}
__global__ void k_fill(int64_t n, float* x) { int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = 1.0f + (float)(i & 7); }

// BLK threads, RB rows per block (y block in LDS), NP panels, T = 4096 nonzeros per tile
template <int BLK, int RB, int T, int MODE = 0>
__global__ __launch_bounds__(BLK) void k_panel(int np, int pcols, const unsigned* __restrict__ packed, const float* __restrict__ val,
                                               const float* __restrict__ x, float* __restrict__ y, int swizzle) {
    __shared__ float ys[RB];
    const int tid = threadIdx.x;
    int rb = blockIdx.x;
    if (swizzle) {  // XCD-contiguous row blocks
        const int n = gridDim.x, q = n / 8, j = rb % 8, idx = rb / 8;
        rb = j * q + idx;
    }
    for (int i = tid; i < RB; i += BLK) ys[i] = 0.f;
    __syncthreads();
    constexpr int V = T / BLK / 4;
    float acc = 0.f;
    for (int p = 0; p < np; ++p) {
        const int64_t base = ((int64_t)rb * np + p) * T;
        const u4* c4 = reinterpret_cast<const u4*>(packed + base);
        const f4* v4 = reinterpret_cast<const f4*>(val + base);
        const float* xp = x + (int64_t)p * pcols;
        u4 cc[V]; f4 vv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { cc[j] = __builtin_nontemporal_load(&c4[j * BLK + tid]); vv[j] = __builtin_nontemporal_load(&v4[j * BLK + tid]); }
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xv = MODE == 2 ? 1.0f : xp[cc[j][q] & 0x3FFFF];
                if (MODE == 1) acc += vv[j][q] * xv;
                else atomicAdd(&ys[cc[j][q] >> 18], vv[j][q] * xv);
            }
    }
    if (MODE == 1) ys[tid] = acc;
    __syncthreads();
    for (int i = tid; i < RB; i += BLK) y[(int64_t)rb * RB + i] = ys[i];
}

template <typename F> float timeit(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError()); return ms / iters;
}

template <int BLK, int RB, int T, int MODE = 0>
void run(int64_t rows, int64_t cols, int64_t nnz, unsigned* packed, float* val, float* x, float* y) {
    const int nrb = rows / RB;
    const int np = (int)(nnz / nrb / T);
    const int pcols = (int)(cols / np);
    k_gen<<<(nnz + 255) / 256, 256>>>(nnz, RB, pcols, packed, val);   // rows sorted inside each T-tile
    CK(hipDeviceSynchronize());
    const double bytes_alg = 8.0 * nnz + 4.0 * (rows + 1) + 4.0 * rows + 4.0 * cols;
    float ms = timeit([&] { k_panel<BLK, RB, T, MODE><<<nrb, BLK>>>(np, pcols, packed, val, x, y, 1); }, 5);
    printf("mode %d  BLK=%4d RB=%5d T=%4d panels=%3d (x slice %4d KiB)  %.4f ms  %.0f GB/s  %.1f%% of peak\n", MODE, BLK, RB, T, np, pcols * 4 / 1024, ms,
           bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80);
}

// v2: the row block's nonzeros as ONE flat stream (tiles are consecutive), 4*V nonzeros per thread per step,
// next step's loads issued before this step's gathers/atomics (the y block in LDS caps occupancy at 4 waves/SIMD,
// so there are 128 VGPRs per lane to spend on loads in flight).
template <int BLK, int RB, int T, int V, int MODE>
__global__ __launch_bounds__(BLK) void k_panel2(int np, int pcols, const unsigned* __restrict__ packed, const float* __restrict__ val,
                                                const float* __restrict__ x, float* __restrict__ y) {
    __shared__ float ys[RB];
    const int tid = threadIdx.x;
    int rb = blockIdx.x;
    { const int n = gridDim.x, q = n / 8, j = rb % 8, idx = rb / 8; rb = j * q + idx; }
    for (int i = tid; i < RB; i += BLK) ys[i] = 0.f;
    __syncthreads();
    const int64_t base = (int64_t)rb * np * T;
    const int total = np * T;
    constexpr int STEP = BLK * 4 * V;
    const u4* c4 = reinterpret_cast<const u4*>(packed + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    u4 ca[V], cb[V]; f4 va[V], vb[V];
    float acc = 0.f;
    auto load = [&](int s, u4* cc, f4* vv) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int i = s / 4 + j * BLK + tid;
            cc[j] = __builtin_nontemporal_load(&c4[i]); vv[j] = __builtin_nontemporal_load(&v4[i]);
        }
    };
    auto work = [&](int s, const u4* cc, const f4* vv) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int k = s + (j * BLK + tid) * 4;
            const float* xp = x + (int64_t)(k / T) * pcols;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xv = MODE == 2 ? 1.0f : xp[cc[j][q] & 0x3FFFF];
                if (MODE == 1) acc += vv[j][q] * xv; else atomicAdd(&ys[cc[j][q] >> 18], vv[j][q] * xv);
            }
        }
    };
    load(0, ca, va);
    for (int s = 0; s < total; s += 2 * STEP) {
        if (s + STEP < total) load(s + STEP, cb, vb);
        work(s, ca, va);
        if (s + 2 * STEP < total) load(s + 2 * STEP, ca, va);
        if (s + STEP < total) work(s + STEP, cb, vb);
    }
    if (MODE == 1) ys[tid] = acc;
    __syncthreads();
    for (int i = tid; i < RB; i += BLK) y[(int64_t)rb * RB + i] = ys[i];
}

template <int BLK, int RB, int T, int V, int MODE = 0>
void run2(int64_t rows, int64_t cols, int64_t nnz, unsigned* packed, float* val, float* x, float* y) {
    const int nrb = rows / RB;
    const int np = (int)(nnz / nrb / T);
    const int pcols = (int)(cols / np);
    k_gen<<<(nnz + 255) / 256, 256>>>(nnz, RB, pcols, packed, val);
    CK(hipDeviceSynchronize());
    const double bytes_alg = 8.0 * nnz + 4.0 * (rows + 1) + 4.0 * rows + 4.0 * cols;
    float ms = timeit([&] { k_panel2<BLK, RB, T, V, MODE><<<nrb, BLK>>>(np, pcols, packed, val, x, y); }, 5);
    printf("v2 mode %d  BLK=%4d RB=%5d T=%4d V=%d panels=%3d (x slice %4d KiB)  %.4f ms  %.0f GB/s  %.1f%% of peak\n", MODE, BLK, RB, T, V, np,
           pcols * 4 / 1024, ms, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80);
}

// v3: every WAVE owns RW rows (its y sub-block in LDS) and sweeps the panels over its own flat stream; tiles are
// row-sorted, so a plain LDS read-add-write replaces the atomic (no two lanes of one instruction share a row unless
// a row holds >= 5 elements of one tile; the real kernel takes a segmented-sum path then).  No barriers at all.
__global__ void k_gen3(int64_t n, int rw, int t, int pcols, unsigned* packed, float* val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 7);
    const int it = (int)(i % t), per = rw / t;                      // element `it` of its tile gets a row bucket of its own
    unsigned row = (unsigned)(it * per + (int)(h % (unsigned)per));
    unsigned col = (unsigned)((h >> 20) % (unsigned)pcols);
    packed[i] = (row << 18) | col;
    val[i] = (float)((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f);
}
template <int RW, int T, int WPB, int V, int MODE>
__global__ __launch_bounds__(64 * WPB) void k_panel3(int np, int pcols, const unsigned* __restrict__ packed, const float* __restrict__ val,
                                                     const float* __restrict__ x, float* __restrict__ y) {
    __shared__ float ys_all[WPB][RW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int b = blockIdx.x;
    { const int n = gridDim.x, q = n / 8, j = b % 8, idx = b / 8; b = j * q + idx; }
    const int wb = b * WPB + w;
    float* ys = ys_all[w];
    for (int i = lane; i < RW; i += 64) ys[i] = 0.f;
    __syncthreads();
    const int64_t base = (int64_t)wb * np * T;
    const int total = np * T;
    constexpr int STEP = 64 * 4 * V;
    const u4* c4 = reinterpret_cast<const u4*>(packed + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    u4 ca[V], cb[V]; f4 va[V], vb[V];
    auto load = [&](int s, u4* cc, f4* vv) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int i = s / 4 + j * 64 + lane;
            cc[j] = __builtin_nontemporal_load(&c4[i]); vv[j] = __builtin_nontemporal_load(&v4[i]);
        }
    };
    auto work = [&](int s, const u4* cc, const f4* vv) {
        float xv[V][4];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int k = s + (j * 64 + lane) * 4;
            const float* xp = x + (int64_t)(k / T) * pcols;
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[j][q] = MODE == 2 ? 1.0f : xp[cc[j][q] & 0x3FFFF];
        }
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = cc[j][q] >> 18;
                ys[r] = ys[r] + vv[j][q] * xv[j][q];
            }
    };
    load(0, ca, va);
    for (int s = 0; s < total; s += 2 * STEP) {
        if (s + STEP < total) load(s + STEP, cb, vb);
        work(s, ca, va);
        if (s + 2 * STEP < total) load(s + 2 * STEP, ca, va);
        if (s + STEP < total) work(s + STEP, cb, vb);
    }
    __syncthreads();
    for (int i = lane; i < RW; i += 64) y[(int64_t)wb * RW + i] = ys[i];
}
template <int RW, int T, int WPB, int V, int MODE = 0>
void run3(int64_t rows, int64_t cols, int64_t nnz, unsigned* packed, float* val, float* x, float* y) {
    const int nwb = rows / RW;
    const int np = (int)(nnz / nwb / T);
    const int pcols = (int)(cols / np);
    k_gen3<<<(nnz + 255) / 256, 256>>>(nnz, RW, T, pcols, packed, val);
    CK(hipDeviceSynchronize());
    const double bytes_alg = 8.0 * nnz + 4.0 * (rows + 1) + 4.0 * rows + 4.0 * cols;
    float ms = timeit([&] { k_panel3<RW, T, WPB, V, MODE><<<nwb / WPB, 64 * WPB>>>(np, pcols, packed, val, x, y); }, 5);
    printf("v3 mode %d  RW=%5d T=%4d waves/blk=%d V=%d panels=%3d (x slice %4d KiB)  %.4f ms  %.0f GB/s  %.1f%% of peak\n", MODE, RW, T, WPB, V, np,
           pcols * 4 / 1024, ms, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80);
}

int main(int argc, char** argv) {
    const int64_t rows = 1ll << 24, cols = rows, nnz = rows * 16;
    unsigned* packed; float *val, *x, *y;
    CK(hipMalloc(&packed, nnz * 4)); CK(hipMalloc(&val, nnz * 4)); CK(hipMalloc(&x, cols * 4)); CK(hipMalloc(&y, rows * 4));
    k_fill<<<(cols + 255) / 256, 256>>>(cols, x);
    run3<4096, 512, 4, 2>(rows, cols, nnz, packed, val, x, y);
    run3<4096, 512, 4, 4>(rows, cols, nnz, packed, val, x, y);
    run3<4096, 512, 2, 2>(rows, cols, nnz, packed, val, x, y);
    run3<8192, 2048, 2, 2>(rows, cols, nnz, packed, val, x, y);
    run3<8192, 2048, 2, 4>(rows, cols, nnz, packed, val, x, y);
    run3<8192, 1024, 2, 4>(rows, cols, nnz, packed, val, x, y);
    run3<8192, 1024, 1, 4>(rows, cols, nnz, packed, val, x, y);
    run3<16384, 4096, 1, 4>(rows, cols, nnz, packed, val, x, y);
    run3<16384, 2048, 1, 4>(rows, cols, nnz, packed, val, x, y);
    run3<16384, 2048, 1, 4, 2>(rows, cols, nnz, packed, val, x, y);
    return 0;
}
