// tools/ubench_stream.hip -- development micro-benchmark (not part of the product or the tests):
// where does the time of the nnz-chunked SpMV go on gfx950?  Stages are added one at a time on
// a 16Mi-row / 256Mi-nnz constant-length matrix with a configurable column band.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_stream.hip -o gpurun_out/ubench
//   run  : gpurun_out/ubench [band=256] [rows_log2=24]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s @%d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}

__global__ void k_gen(int64_t nnz, int per_row, int64_t cols, int64_t band, int* col, float* val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    int64_t row = i / per_row; int k = (int)(i % per_row);
    uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
    int64_t W = band > 0 ? band : cols;
    int64_t w0 = band > 0 ? row - W / 2 : 0;
    if (w0 < 0) w0 = 0; if (w0 > cols - W) w0 = cols - W;
    int64_t lo = (int64_t)k * W / per_row, hi = (int64_t)(k + 1) * W / per_row;
    col[i] = (int)(w0 + lo + (int64_t)((h >> 32) % (uint64_t)(hi - lo)));
    val[i] = (float)((int)(h & 0xFFFFFF) - 0x800000) * (1.0f / 8388608.0f);
}
__global__ void k_fill(int64_t n, float* x) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = (float)((int)(mix64(i + 99) & 0xFFFF) - 32768) * (1.0f / 32768.0f);
}

using i4 = int __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));
constexpr int BS = 256;

// stage 0: stream col+val with 16-B loads, NPT nnz per thread, reduce in registers, 1 store/thread
template <int NPT, bool NT>
__global__ __launch_bounds__(BS) void k_stream(const int* __restrict__ col, const float* __restrict__ val, float* __restrict__ out) {
    constexpr int V = NPT / 4;
    const int64_t base = (int64_t)blockIdx.x * BS * NPT;
    const i4* c4 = reinterpret_cast<const i4*>(col + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    i4 cc[V]; f4 vv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if (NT) { cc[j] = __builtin_nontemporal_load(&c4[j * BS + threadIdx.x]); vv[j] = __builtin_nontemporal_load(&v4[j * BS + threadIdx.x]); }
        else { cc[j] = c4[j * BS + threadIdx.x]; vv[j] = v4[j * BS + threadIdx.x]; }
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j)
        acc += vv[j].x * (float)cc[j].x + vv[j].y * (float)cc[j].y + vv[j].z * (float)cc[j].z + vv[j].w * (float)cc[j].w;
    out[(int64_t)blockIdx.x * BS + threadIdx.x] = acc;
}

// stage 1: + gather x[col]
template <int NPT, bool NT>
__global__ __launch_bounds__(BS) void k_gather(const int* __restrict__ col, const float* __restrict__ val, const float* __restrict__ x, float* __restrict__ out) {
    constexpr int V = NPT / 4;
    const int64_t base = (int64_t)blockIdx.x * BS * NPT;
    const i4* c4 = reinterpret_cast<const i4*>(col + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    i4 cc[V]; f4 vv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if (NT) { cc[j] = __builtin_nontemporal_load(&c4[j * BS + threadIdx.x]); vv[j] = __builtin_nontemporal_load(&v4[j * BS + threadIdx.x]); }
        else { cc[j] = c4[j * BS + threadIdx.x]; vv[j] = v4[j * BS + threadIdx.x]; }
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j)
        acc += vv[j].x * x[cc[j].x] + vv[j].y * x[cc[j].y] + vv[j].z * x[cc[j].z] + vv[j].w * x[cc[j].w];
    out[(int64_t)blockIdx.x * BS + threadIdx.x] = acc;
}

// stage 1b: gather with non-temporal / different-scope loads of x (does the fetch granule change?)
template <int MODE>
__global__ __launch_bounds__(BS) void k_gather_x(const int* __restrict__ col, const float* __restrict__ val, const float* __restrict__ x, float* __restrict__ out) {
    constexpr int V = 4;
    const int64_t base = (int64_t)blockIdx.x * BS * 16;
    const i4* c4 = reinterpret_cast<const i4*>(col + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    i4 cc[V]; f4 vv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { cc[j] = __builtin_nontemporal_load(&c4[j * BS + threadIdx.x]); vv[j] = __builtin_nontemporal_load(&v4[j * BS + threadIdx.x]); }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float xv;
            if (MODE == 0) xv = x[cc[j][q]];
            else if (MODE == 1) xv = __builtin_nontemporal_load(&x[cc[j][q]]);
            else xv = __hip_atomic_load(&x[cc[j][q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += vv[j][q] * xv;
        }
    out[(int64_t)blockIdx.x * BS + threadIdx.x] = acc;
}

// stage 2: + products through LDS, thread-per-row sequential reduce (rows of 16), y store
__device__ __forceinline__ int pad_idx(int i) { return i + (i >> 5); }
template <int NPT, bool NT>
__global__ __launch_bounds__(BS) void k_lds_reduce(const int* __restrict__ col, const float* __restrict__ val, const float* __restrict__ x, float* __restrict__ y, int per_row) {
    constexpr int V = NPT / 4; constexpr int T = BS * NPT;
    __shared__ float prod[T + T / 32];
    const int64_t base = (int64_t)blockIdx.x * T;
    const i4* c4 = reinterpret_cast<const i4*>(col + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    i4 cc[V]; f4 vv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if (NT) { cc[j] = __builtin_nontemporal_load(&c4[j * BS + threadIdx.x]); vv[j] = __builtin_nontemporal_load(&v4[j * BS + threadIdx.x]); }
        else { cc[j] = c4[j * BS + threadIdx.x]; vv[j] = v4[j * BS + threadIdx.x]; }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int p0 = pad_idx((j * BS + threadIdx.x) * 4);
        prod[p0] = vv[j].x * x[cc[j].x]; prod[p0 + 1] = vv[j].y * x[cc[j].y];
        prod[p0 + 2] = vv[j].z * x[cc[j].z]; prod[p0 + 3] = vv[j].w * x[cc[j].w];
    }
    __syncthreads();
    const int rows_here = T / per_row;
    for (int r = threadIdx.x; r < rows_here; r += BS) {
        float acc = 0.f;
        for (int i = r * per_row; i < (r + 1) * per_row; ++i) acc += prod[pad_idx(i)];
        y[base / per_row + r] = acc;
    }
}

// stage 3: x window staged in LDS (aliased with the product buffer), gathers from LDS.
// MODE 0 = full, 1 = window loads skipped (barriers kept), 2 = window via LDS-DMA (global_load_lds)
template <int BLK, int MODE>
__global__ __launch_bounds__(BLK, 8) void k_win(const int* __restrict__ col, const float* __restrict__ val, const float* __restrict__ x,
                                                float* __restrict__ y, int per_row, int band, int64_t cols) {
    constexpr int NPT = 16, V = 4, T = BLK * NPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * T;
    const i4* c4 = reinterpret_cast<const i4*>(col + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    i4 cc[V]; f4 vv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { cc[j] = __builtin_nontemporal_load(&c4[j * BLK + tid]); vv[j] = __builtin_nontemporal_load(&v4[j * BLK + tid]); }
    // window of this chunk: rows [r0, r0 + T/per_row), columns [r0 - band/2, r0 + rows + band/2)
    const int64_t r0 = base / per_row; const int nrows = T / per_row;
    int64_t w0 = r0 - band / 2; if (w0 < 0) w0 = 0; w0 &= ~3ll;
    int64_t w1 = r0 + nrows + band / 2 + 4; if (w1 > cols) w1 = cols;
    const int wlen = (int)(w1 - w0);
    if (MODE == 0) {
        for (int i = tid * 4; i < wlen; i += BLK * 4) *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(x + w0 + i);
    } else if (MODE == 2) {
        // LDS-DMA: the LDS destination is wave-uniform base + lane*16, so the base is the first
        // float of this WAVE's 1-KiB piece of the pass
        const int lane = tid & 63, wave = tid >> 6;
        for (int i0 = 0; i0 < wlen; i0 += BLK * 4) {
            const int piece = i0 + wave * 256;      // floats
            if (piece + lane * 4 < wlen)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(x + w0 + piece + lane * 4),
                                                 (void __attribute__((address_space(3)))*)(smem + piece), 16, 0, 0);
        }
    }
    __syncthreads();
    f4 xv[V];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) { int o = cc[j][q] - (int)w0; o = o < 0 ? 0 : (o >= wlen ? wlen - 1 : o); xv[j][q] = smem[o]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int p0 = pad_idx((j * BLK + tid) * 4);
        smem[p0] = vv[j][0] * xv[j][0]; smem[p0 + 1] = vv[j][1] * xv[j][1]; smem[p0 + 2] = vv[j][2] * xv[j][2]; smem[p0 + 3] = vv[j][3] * xv[j][3];
    }
    __syncthreads();
    for (int r = tid; r < nrows; r += BLK) {
        float acc = 0.f;
        for (int i = r * per_row; i < (r + 1) * per_row; ++i) acc += smem[pad_idx(i)];
        y[r0 + r] = acc;
    }
}

// stage 4: window wider than the LDS region -> NPASS passes of REGION floats, predicated LDS gathers
template <int BLK>
__global__ __launch_bounds__(BLK, 8) void k_winp(const int* __restrict__ col, const float* __restrict__ val, const float* __restrict__ x,
                                                 float* __restrict__ y, int per_row, int band, int64_t cols) {
    constexpr int NPT = 16, V = 4, T = BLK * NPT, REGION = BLK * 18;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * T;
    const i4* c4 = reinterpret_cast<const i4*>(col + base);
    const f4* v4 = reinterpret_cast<const f4*>(val + base);
    i4 cc[V]; f4 vv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { cc[j] = __builtin_nontemporal_load(&c4[j * BLK + tid]); vv[j] = __builtin_nontemporal_load(&v4[j * BLK + tid]); }
    const int64_t r0 = base / per_row; const int nrows = T / per_row;
    int64_t w0 = r0 - band / 2; if (w0 < 0) w0 = 0; w0 &= ~3ll;
    int64_t w1 = r0 + nrows + band / 2 + 4; if (w1 > cols) w1 = cols;
    const int wlen = (int)(w1 - w0);
    const int npass = (wlen + REGION - 1) / REGION;
    f4 xv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) xv[j] = f4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < npass; ++p) {
        const int off = p * REGION;
        const int len = (wlen - off) < REGION ? (wlen - off) : REGION;
        for (int i = tid * 4; i < len; i += BLK * 4) *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(x + w0 + off + i);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) { unsigned o = (unsigned)(cc[j][q] - (int)w0 - off); if (o < (unsigned)len) xv[j][q] = smem[o]; }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int p0 = pad_idx((j * BLK + tid) * 4);
        smem[p0] = vv[j][0] * xv[j][0]; smem[p0 + 1] = vv[j][1] * xv[j][1]; smem[p0 + 2] = vv[j][2] * xv[j][2]; smem[p0 + 3] = vv[j][3] * xv[j][3];
    }
    __syncthreads();
    for (int r = tid; r < nrows; r += BLK) {
        float acc = 0.f;
        for (int i = r * per_row; i < (r + 1) * per_row; ++i) acc += smem[pad_idx(i)];
        y[r0 + r] = acc;
    }
}

template <typename F>
float timeit(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char** argv) {
    int64_t band = argc > 1 ? atoll(argv[1]) : 256;
    int lg = argc > 2 ? atoi(argv[2]) : 24;
    const int per_row = 16;
    const int64_t rows = 1ll << lg, cols = rows, nnz = rows * per_row;
    int* col; float *val, *x, *y;
    CK(hipMalloc(&col, nnz * 4)); CK(hipMalloc(&val, nnz * 4)); CK(hipMalloc(&x, cols * 4)); CK(hipMalloc(&y, nnz / 4 * 4 + 1024));
    k_gen<<<(nnz + 255) / 256, 256>>>(nnz, per_row, cols, band, col, val);
    k_fill<<<(cols + 255) / 256, 256>>>(cols, x);
    CK(hipDeviceSynchronize());
    const double bytes_stream = 8.0 * nnz;
    const double bytes_alg = 8.0 * nnz + 4.0 * (rows + 1) + 4.0 * rows + 4.0 * cols;
    printf("rows=%lld nnz=%lld band=%lld\n", (long long)rows, (long long)nnz, (long long)band);
#define RUN(name, NPT, expr, B) { float ms = timeit([&] { expr; }, 20); printf("%-28s npt=%2d  %.4f ms  %.0f GB/s\n", name, NPT, ms, (B) / ms / 1e6); }
    RUN("stream", 4, (k_stream<4, false><<<nnz / (BS * 4), BS>>>(col, val, y)), bytes_stream)
    RUN("stream", 8, (k_stream<8, false><<<nnz / (BS * 8), BS>>>(col, val, y)), bytes_stream)
    RUN("stream", 16, (k_stream<16, false><<<nnz / (BS * 16), BS>>>(col, val, y)), bytes_stream)
    RUN("stream", 32, (k_stream<32, false><<<nnz / (BS * 32), BS>>>(col, val, y)), bytes_stream)
    RUN("stream nt", 8, (k_stream<8, true><<<nnz / (BS * 8), BS>>>(col, val, y)), bytes_stream)
    RUN("stream nt", 16, (k_stream<16, true><<<nnz / (BS * 16), BS>>>(col, val, y)), bytes_stream)
    RUN("gather", 8, (k_gather<8, false><<<nnz / (BS * 8), BS>>>(col, val, x, y)), bytes_alg)
    RUN("gather", 16, (k_gather<16, false><<<nnz / (BS * 16), BS>>>(col, val, x, y)), bytes_alg)
    RUN("gather nt", 8, (k_gather<8, true><<<nnz / (BS * 8), BS>>>(col, val, x, y)), bytes_alg)
    RUN("gather nt", 16, (k_gather<16, true><<<nnz / (BS * 16), BS>>>(col, val, x, y)), bytes_alg)
    RUN("gather_x plain", 16, (k_gather_x<0><<<nnz / (BS * 16), BS>>>(col, val, x, y)), bytes_alg)
    RUN("gather_x nt", 16, (k_gather_x<1><<<nnz / (BS * 16), BS>>>(col, val, x, y)), bytes_alg)
    RUN("gather_x sc1", 16, (k_gather_x<2><<<nnz / (BS * 16), BS>>>(col, val, x, y)), bytes_alg)
    RUN("lds_reduce", 8, (k_lds_reduce<8, false><<<nnz / (BS * 8), BS>>>(col, val, x, y, per_row)), bytes_alg)
    RUN("lds_reduce", 16, (k_lds_reduce<16, false><<<nnz / (BS * 16), BS>>>(col, val, x, y, per_row)), bytes_alg)
    RUN("lds_reduce nt", 8, (k_lds_reduce<8, true><<<nnz / (BS * 8), BS>>>(col, val, x, y, per_row)), bytes_alg)
    RUN("lds_reduce nt", 16, (k_lds_reduce<16, true><<<nnz / (BS * 16), BS>>>(col, val, x, y, per_row)), bytes_alg)
    if (band <= 300000) {
#define RUNP(BLK, name) { size_t lds = (size_t)BLK * 18 * 4; CK(hipFuncSetAttribute((const void*)&k_winp<BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        RUN(name, BLK, (k_winp<BLK><<<nnz / (BLK * 16), BLK, lds>>>(col, val, x, y, per_row, (int)band, cols)), bytes_alg) }
        RUNP(256, "win multipass blk") RUNP(512, "win multipass blk") RUNP(1024, "win multipass blk")
    }
    if (band <= 16384) {
#define RUNW(BLK, MODE, name) { size_t lds = (size_t)BLK * 18 * 4; CK(hipFuncSetAttribute((const void*)&k_win<BLK, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        if ((size_t)(band + BLK + 8) * 4 <= lds) RUN(name, BLK, (k_win<BLK, MODE><<<nnz / (BLK * 16), BLK, lds>>>(col, val, x, y, per_row, (int)band, cols)), bytes_alg) }
        RUNW(256, 0, "win full blk") RUNW(256, 1, "win noload blk") RUNW(256, 2, "win ldsdma blk")
        RUNW(512, 0, "win full blk") RUNW(512, 1, "win noload blk") RUNW(512, 2, "win ldsdma blk")
        RUNW(1024, 0, "win full blk") RUNW(1024, 1, "win noload blk") RUNW(1024, 2, "win ldsdma blk")
    }
    return 0;
}
