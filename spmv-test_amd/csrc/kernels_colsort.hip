// kernels_colsort.hip -- SPMV_PANEL mode 3: row blocks whose nonzeros are streamed in COLUMN order ("sorted blocks").
//
// Why (round 3, tools/ubench_gather_lines.hip -> profiles/r03_gather_lines_ubench.jsonl): a 4-byte gather instruction is
// priced per DISTINCT 128-byte line of x its 64 lanes touch -- 270 G lines/s chip-wide from an L2-resident table (the
// L2's 34.5 TB/s), 58 G lines/s from the Infinity Cache / HBM -- up to a lane limit of 1.3-2 T gathers/s: 64 lanes on
// 16 lines run at 0.94-1.04 T gathers/s where 64 lanes on 64 lines run at 0.27.  The row-major kernels and the panel
// sweep (kernels_panel.hip, rows ascending inside a tile) put one nonzero of a line into an instruction; a matrix
// whose row block of R rows holds k nonzeros per line of x (k = 32 R nnz_per_row / column span of the block: 2 for
// config 2's uniform columns at R = 4096, 2 for a band of 1M columns, 10 for a band of 200 000) can put k of them
// there -- if the block's nonzeros are streamed sorted by column.  That order breaks the row grouping, so the sums
// cannot be registers or a product buffer; they are the block's outputs themselves, kept in LDS for the whole
// block (the panel sweep's idea), and the plan arranges that the 64 nonzeros of one instruction hold 64 DIFFERENT
// rows, so the add is a plain LDS read-add-write (LDS float atomics: 0.33 lane-updates per clock and CU, measured).
//
// Same role as kernels_panel.hip -- the reference's tiled format (TCSRMatrix src/tcsr.cpp:5-38 + csr_tiling_kernel
// src/kernels/csr_tiling.cu:24-114: a tile of the matrix against a tile of x in shared memory) at sparse scale.
//
// Layout (PanelPlan, sorted_mode):
//   rows are cut into blocks of <= 4096 rows with equal nonzero counts (brow[]);
//   a block's nonzeros, stably sorted by 128-byte line of x, are dealt into GROUPS of 64 with distinct rows: a nonzero
//   whose row is already in the group waits for the next one (at most 16 wait; more go to the block's "flagged" tail,
//   which is added with LDS atomics -- a row of thousands of nonzeros cannot be spread one per group);
//   four groups make a UNIT of 256 slots, slot = lane*4 + group, so a lane's 16-byte load holds one nonzero of each
//   of the unit's groups and one wave instruction works on one group: 64 neighbours in column order, 64 rows;
//     packed[slot] = row_in_block << 19 | column - ubase[unit]       pvals[slot] = value (a COPY, like the panel sweep)
//   empty slots hold the lane's own dummy row (4096 + lane), column offset 0 and value 0: no test in the inner loop;
//   ubeg[b] = first unit of block b, usimple[b] = how many of its units need no atomics.
// Multiply: one workgroup of 8 wavefronts per block, one per CU (130 KiB of LDS): wavefront w takes units w, w+8, ...
// and adds into ITS OWN copy of the block's 4096 sums -- groups are conflict-free inside, wavefronts never share a
// copy, so there is no atomic and no barrier until the 8 copies are added, in wavefront order, into y.  Streams two
// steps ahead, gathers one step ahead of the adds (two wavefronts per SIMD: little else hides the latency).
// Deterministic: the plan is a pure function of the matrix (stable sort, highest lane wins a row), the adds of a
// wavefront happen in instruction order.
#include <climits>
#include "spmv_internal.hpp"

namespace spmv {

namespace {

// Geometry: a block of ROWS rows (4096 | 8192), WAVES wavefronts (= private copies of the block's sums, ROWS + 64 floats
// each) per workgroup -- the plan picks one of
//   4096 rows x 8 wavefronts  (130 KiB: one workgroup per CU)    few blocks (config 2: 256 blocks for 256 CUs)
//   4096 rows x 4 wavefronts  ( 65 KiB: two workgroups per CU)   many blocks: one block's ending overlaps the other's stream
//   8192 rows x 4 wavefronts  (129 KiB: one workgroup per CU)    half the lines of x per nonzero, half the wavefronts: pays
//                                                                 from ~0.3 lines per nonzero (a band of 1M columns)
// packed = row << colbits | column offset, colbits = 19 (4096 + 64 rows: 13 bits) or 18 (8192 + 64 rows: 14 bits).
constexpr int kCbRowsMax = 8192;
__host__ __device__ constexpr int cb_colbits(int rows) { return rows > 4096 ? 18 : 19; }
constexpr int kCbUnit = 4 * kWave;            // slots per unit: four groups of 64
constexpr int kCbQueue = 16;                  // nonzeros that may wait for the next group
constexpr int kCbBins = 32768;                // counters of the plan's counting sort: 128 KiB of LDS
constexpr int kCbVec = 4;                     // units per wavefront and pipeline step (A/B: 4 beats 2 by 4-10 % where gathers cost)

using u4 = unsigned __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));
using us4 = unsigned short __attribute__((ext_vector_type(4)));

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

__device__ __forceinline__ int xcd_block(int bid, int n)   // XCD j = bid % 8 gets a contiguous range of blocks
{
    const int q = n / kXcds, rem = n % kXcds;
    const int j = bid % kXcds, idx = bid / kXcds;
    return j * q + (j < rem ? j : rem) + idx;
}

// ---- plan ---------------------------------------------------------------------------------------------------------
// units a block may use: its nonzeros + 1/16 (groups are not always full) + four units (ragged front, ragged tail, slack)
__global__ void k_cb_units(int nblocks, const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                           int32_t *__restrict__ units)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int64_t n = (int64_t)row_ptr[brow[b + 1]] - row_ptr[brow[b]];
    units[b] = (int32_t)((n + n / 16 + kCbUnit - 1) / kCbUnit + 4);
}

// A row is LONG for its block when it holds more nonzeros than a quarter of the block's groups: it cannot be dealt one
// per group without the others queueing up behind it.  Its nonzeros skip the groups and go to the block's tail.
__device__ __forceinline__ int long_row_limit(int n)
{
    const int g = (n + kWave - 1) / kWave;
    return g / 4 > 16 ? g / 4 : 16;
}

// rank of every lane among the lanes that hold the same key (keys of invalid lanes must be negative), and how many do
__device__ __forceinline__ void rank_same_key(int key, bool valid, int lane, int &rank, int &cnt)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    rank = 0;
    cnt = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int kl = __shfl(key, leader, kWave);
        const unsigned long long m = __ballot(valid && key == kl);
        if (valid && key == kl) {
            rank = __popcll(m & lt);
            cnt = __popcll(m);
        }
        todo &= ~m;
    }
}

// in-place exclusive scan of `count` LDS integers (a multiple of 64) by one wavefront; returns the total
__device__ __forceinline__ int wave_scan_lds(int *a, int count, int lane, int *nonzero)
{
    int carry = 0, used = 0;
    for (int c0 = 0; c0 < count; c0 += kWave) {
        const int v = a[c0 + lane];
        int incl = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(incl, d, kWave);
            if (lane >= d) incl += t;
        }
        a[c0 + lane] = carry + incl - v;
        carry += __shfl(incl, kWave - 1, kWave);
        used += v > 0 ? 1 : 0;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) used += __shfl_down(used, o, kWave);
    if (nonzero) *nonzero = __shfl(used, 0, kWave);
    return carry;
}

// The widest block: max over the blocks of (largest - smallest column) of ALL their nonzeros, for sizing the counters of step 1
__global__ __launch_bounds__(kBlock) void k_cb_span(const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                                                    const int32_t *__restrict__ col_idx, int32_t *__restrict__ widest)
{
    __shared__ int s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
    const int s = row_ptr[brow[blockIdx.x]], e = row_ptr[brow[blockIdx.x + 1]];
    int mn = INT_MAX, mx = -1;
    for (int k = s + tid; k < e; k += kBlock) {
        const int c = col_idx[k];
        mn = c < mn ? c : mn;
        mx = c > mx ? c : mx;
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int a = __shfl_xor(mn, o, kWave), c = __shfl_xor(mx, o, kWave);
        mn = a < mn ? a : mn;
        mx = c > mx ? c : mx;
    }
    if (lane == 0) { s_mn[wv] = mn; s_mx[wv] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBlock / kWave; ++w) {
            mn = s_mn[w] < mn ? s_mn[w] : mn;
            mx = s_mx[w] > mx ? s_mx[w] : mx;
        }
        if (mx >= mn) atomicMax(widest, mx - mn);
    }
}

// Step 1, one workgroup per block: t_* (CSR index space: block b fills [row_ptr[brow[b]], ...)) =
//   [ the nonzeros of its short rows, stably sorted by line of x | the nonzeros of its long rows in CSR order ]
// (bins of 2^shift columns, shift = 5 unless the block spans more than 32768 lines), nshort[b] = where the second part
// begins.  stats[0] += lines of x the block's short rows touch (occupied bins; one per nonzero when shift > 5),
// stats[2] += 1 for a block with shift > 5 (its window of x is beyond 4 MiB: one XCD's L2).
__global__ __launch_bounds__(kBlock) void k_cb_sort(const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                                                    const int32_t *__restrict__ col_idx, const float *__restrict__ vals,
                                                    const uint16_t *__restrict__ rowloc, int32_t *__restrict__ t_col,
                                                    uint16_t *__restrict__ t_row, float *__restrict__ t_val,
                                                    int32_t *__restrict__ nshort, unsigned long long *__restrict__ stats, int nbins)
{
    // nbins counters (a power of two, at most kCbBins): as many as the widest block needs at 32 columns per bin -- a band of
    // 200 000 columns: 8192 counters, 32 KiB, four workgroups per CU where 128 KiB allowed one (the scatter below runs on ONE
    // wavefront per block: round 3's 25 ms of plan time at config 4 were 256 wavefronts on the whole chip)
    extern __shared__ int hist[];   // nbins
    __shared__ unsigned longmask[kCbRowsMax / 32];
    __shared__ int uniq[1024];
    __shared__ int s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
    const int b = blockIdx.x;
    const int row0 = brow[b], nrows = brow[b + 1] - row0;
    const int s = row_ptr[row0], e = row_ptr[row0 + nrows];
    if (s == e) {   // workgroup-uniform
        if (tid == 0) nshort[b] = 0;
        return;
    }
    // ---- all four wavefronts: long rows, column range, histogram (the passes that are only reads)
    const int lmax = long_row_limit(e - s);
    for (int i = tid; i < kCbRowsMax / 32; i += kBlock) longmask[i] = 0u;
    for (int i = tid; i < nbins; i += kBlock) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < nrows; i += kBlock)
        if (row_ptr[row0 + i + 1] - row_ptr[row0 + i] > lmax) atomicOr(&longmask[i >> 5], 1u << (i & 31));
    __syncthreads();
    auto is_long = [&](int rl) { return ((longmask[rl >> 5] >> (rl & 31)) & 1u) != 0u; };
    int mn = INT_MAX, mx = -1;
    for (int k = s + tid; k < e; k += kBlock) {
        if (is_long(rowloc[k])) continue;
        const int c = col_idx[k];
        mn = c < mn ? c : mn;
        mx = c > mx ? c : mx;
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int a = __shfl_xor(mn, o, kWave), c = __shfl_xor(mx, o, kWave);
        mn = a < mn ? a : mn;
        mx = c > mx ? c : mx;
    }
    if (lane == 0) { s_mn[wv] = mn; s_mx[wv] = mx; }
    __syncthreads();
    for (int w = 0; w < kBlock / kWave; ++w) {
        mn = s_mn[w] < mn ? s_mn[w] : mn;
        mx = s_mx[w] > mx ? s_mx[w] : mx;
    }
    if (mx < 0) mn = 0;   // no short row at all
    int shift = 5;
    while (mx >= 0 && ((mx - mn) >> shift) >= nbins) ++shift;
    for (int k = s + tid; k < e; k += kBlock)
        if (!is_long(rowloc[k])) atomicAdd(&hist[(col_idx[k] - mn) >> shift], 1);
    __syncthreads();
    if (wv != 0) return;   // ---- the rest is sequential in the bins' cursors: one wavefront
    int used = 0;
    const int ns = wave_scan_lds(hist, nbins, lane, &used);
    if (lane == 0) {
        nshort[b] = ns;
        atomicAdd(&stats[0], (unsigned long long)(shift == 5 ? used : ns));
        if (shift > 5) atomicAdd(&stats[2], 1ull);
    }
    // stable scatter: 64 nonzeros per trip in CSR order (eight trips' loads in flight at a time: one wavefront has nothing
    // else to hide a memory round trip per trip); the rank of a nonzero among the same-bin nonzeros of its trip comes from
    // ballots (registers only), the bin's cursor is read once before and written once after
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int kAhead = 8;
    int nlong = 0;   // wave-uniform
    for (int base0 = s; base0 < e; base0 += kAhead * kWave) {
        int pc[kAhead], pr[kAhead];
        float pv[kAhead];
#pragma unroll
        for (int t = 0; t < kAhead; ++t) {
            const int k = base0 + t * kWave + lane;
            pc[t] = 0; pr[t] = 0; pv[t] = 0.0f;
            if (k < e) {
                pc[t] = col_idx[k];
                pr[t] = rowloc[k];
                pv[t] = vals[k];
            }
        }
#pragma unroll
        for (int t = 0; t < kAhead; ++t) {
            const int k = base0 + t * kWave + lane;
            if (base0 + t * kWave >= e) break;   // wave-uniform
            const bool valid = k < e;
            const int col = pc[t], rl = pr[t];
            const float v = pv[t];
            const bool lng = valid && is_long(rl);
            const bool shrt = valid && !lng;
            const int bin = shrt ? (col - mn) >> shift : -1;
            const int first = shrt ? hist[bin] : 0;
            // Most nonzeros of a trip are alone in their bin: find the ones that are not with two rounds of "write my
            // lane, read it back" on a small hash table (a lane that reads another lane's id has a rival; the lane that
            // won the first round learns of its rivals in the second, which only they write), and rank only those by
            // ballot (64 ballots per trip otherwise).  Two bins on one hash slot: ranked too, each in its own bin.
            // (volatile: the value read back is another LANE's store -- the compiler must not forward this lane's own)
            volatile int *vu = uniq;
            const int hs = bin & 1023;
            if (shrt) vu[hs] = lane;
            bool rival = shrt && vu[hs] != lane;
            if (rival) vu[hs] = lane;
            rival = rival || (shrt && vu[hs] != lane);
            int rank = 0, cnt = 1;
            if (__ballot(rival)) {
                int rr, cc;
                rank_same_key(bin, rival, lane, rr, cc);
                if (rival) { rank = rr; cnt = cc; }
            }
            if (shrt && rank == cnt - 1) hist[bin] = first + cnt;   // the last of its bin in this trip moves the cursor
            const unsigned long long lm = __ballot(lng);
            const int d = shrt ? s + first + rank : s + ns + nlong + __popcll(lm & lt);
            nlong += __popcll(lm);
            if (valid) {
                t_col[d] = col;
                t_row[d] = (uint16_t)rl;
                t_val[d] = v;
            }
        }
    }
}

// Step 2, one wavefront per block: deal the sorted nonzeros of the short rows into groups of 64 with distinct rows (see the
// file header); what cannot be grouped, and the long rows, form the block's TAIL, sorted by row.  Output in the block's
// units, still with absolute columns (o_col) and the row in a side array (o_row, 0xFFFF = empty slot, preset):
//   units [0, usimple[b])            the groups: slot = unit*256 + lane*4 + (group & 3)
//   units [usimple[b], uend[b]-ubeg) the tail, same slot map, so that one instruction holds 64 CONSECUTIVE tail nonzeros:
//                                    runs of equal rows, folded by a segmented scan at run time
// A block that runs out of units (cannot happen with the long rows taken out, short of adversarial input) becomes all tail.
__global__ __launch_bounds__(kWave) void k_cb_groups(const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ ubeg, const int32_t *__restrict__ nshort,
                                                     int32_t *__restrict__ t_col, uint16_t *__restrict__ t_row,
                                                     float *__restrict__ t_val, int32_t *__restrict__ o_col,
                                                     uint16_t *__restrict__ o_row, float *__restrict__ o_val,
                                                     int32_t *__restrict__ usimple, int32_t *__restrict__ uend,
                                                     unsigned long long *__restrict__ stats)
{
    __shared__ int tag[kCbRowsMax];   // grouping: who holds a row in this round; afterwards: the counters of the tail sort
    __shared__ int q_col[kWave];      // the nonzeros that wait for the next group
    __shared__ int q_row[kWave];
    __shared__ float q_val[kWave];
    constexpr int kBuf = 1024;        // the next nonzeros of the sorted stream, loaded 512 at a time (a ring): one wavefront
    __shared__ int b_col[kBuf];       // has nothing else to hide a memory round trip per group
    __shared__ int b_row[kBuf];
    __shared__ float b_val[kBuf];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int s = row_ptr[brow[b]], n = row_ptr[brow[b + 1]] - s;
    const int ns = nshort[b];
    const int u0 = ubeg[b];
    const int cap = (ubeg[b + 1] - u0) * kCbUnit;
    int32_t *oc = o_col + (int64_t)u0 * kCbUnit;
    uint16_t *orow = o_row + (int64_t)u0 * kCbUnit;
    float *ov = o_val + (int64_t)u0 * kCbUnit;
    for (int i = lane; i < kCbRowsMax; i += kWave) tag[i] = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int nlong = n - ns;
    int front = 0, back = cap, pos = 0, qn = 0, round = 0;
    int c_col = 0, c_row = 0;
    float c_val = 0.0f;
    bool failed = false;
    // One GROUP per trip: 64 candidates -- first the ones that waited (low lanes), then the next of the sorted stream.  The
    // lowest lane that holds a row wins it (atomicMax of a key that grows with the round and falls with the lane); the
    // others wait for the next group, at most kCbQueue of them -- more are set aside for the tail.
    // (Tried: a row at most once per UNIT, so that the multiply can issue a unit's four LDS reads before its first write.
    // No faster -- the add chain is not what bounds the kernel -- and in a narrow band, where 256 neighbours in column
    // order sit on two lines of x, most of the block ends up waiting.)
    int loaded = 0;   // nonzeros of the stream in the ring so far: [pos, loaded) are waiting there
    while (pos < ns || qn > 0) {
        if (loaded - pos < 2 * kWave && loaded < ns) {   // wave-uniform: refill (the slots of [pos - ..., pos) are free)
            int rc_[kBuf / 2 / kWave], rr_[kBuf / 2 / kWave];
            float rv_[kBuf / 2 / kWave];
#pragma unroll
            for (int t = 0; t < kBuf / 2 / kWave; ++t) {
                const int i = loaded + t * kWave + lane;
                rc_[t] = 0; rr_[t] = 0; rv_[t] = 0.0f;
                if (i < ns) { rc_[t] = t_col[s + i]; rr_[t] = t_row[s + i]; rv_[t] = t_val[s + i]; }
            }
#pragma unroll
            for (int t = 0; t < kBuf / 2 / kWave; ++t) {
                const int i = loaded + t * kWave + lane;
                b_col[i & (kBuf - 1)] = rc_[t];
                b_row[i & (kBuf - 1)] = rr_[t];
                b_val[i & (kBuf - 1)] = rv_[t];
            }
            loaded = loaded + kBuf / 2 < ns ? loaded + kBuf / 2 : ns;
        }
        const int room = kWave - qn, left = ns - pos;
        const int take = room < left ? room : left;
        const bool have = lane < qn + take;
        if (lane >= qn && have) {
            const int i = (pos + lane - qn) & (kBuf - 1);
            c_col = b_col[i];
            c_row = b_row[i];
            c_val = b_val[i];
        }
        pos += take;
        ++round;
        const int me = round * kWave + (kWave - 1 - lane);
        if (have) atomicMax(&tag[c_row], me);
        const bool win = have && tag[c_row] == me;
        const bool lose = have && !win;
        const unsigned long long lm = __ballot(lose);
        const int nl = __popcll(lm);
        const bool spill = nl > kCbQueue;
        // room: the unit this group lands in must stay clear of what has been set aside at the back (a group's slots are
        // spread over its whole unit), and the tail must fit behind the groups in the end (it starts on a unit boundary)
        const int fu_next = (front + kWave + kCbUnit - 1) / kCbUnit * kCbUnit;
        const int aside = cap - back + (spill ? nl : kCbQueue);
        if (fu_next + aside + nlong + kCbUnit > cap) { failed = true; break; }
        if (win) {
            const int g = front >> 6;
            const int slot = (g >> 2) * kCbUnit + lane * 4 + (g & 3);
            oc[slot] = c_col;
            orow[slot] = (uint16_t)c_row;
            ov[slot] = c_val;
        }
        front += kWave;
        const int r = __popcll(lm & lt);
        if (spill) {            // too many would wait: they are set aside for the tail
            if (lose) {
                const int slot = back - nl + r;
                oc[slot] = c_col;
                orow[slot] = (uint16_t)c_row;
                ov[slot] = c_val;
            }
            back -= nl;
            qn = 0;
        } else {                // the losers open the next group, in their order
            if (lose) { q_col[r] = c_col; q_row[r] = c_row; q_val[r] = c_val; }
            if (lane < nl) { c_col = q_col[lane]; c_row = q_row[lane]; c_val = q_val[lane]; }
            qn = nl;
        }
    }
    // ---- the tail: [A: what was set aside, in arrival order] + [B: the long rows, in CSR order], stably sorted by row
    int a0 = s, na = cap - back;
    if (failed) {               // everything becomes tail: A = the short rows' nonzeros as sorted, nothing grouped
        front = 0;
        na = ns;
        for (int i = lane; i < cap; i += kWave) { orow[i] = 0xFFFFu; ov[i] = 0.0f; }
    } else {
        // move A out of the units (t_[s, s + ns) has been consumed: na <= ns) and clear where it stood
        for (int i = lane; i < na; i += kWave) {
            t_col[s + i] = oc[back + i];
            t_row[s + i] = orow[back + i];
            t_val[s + i] = ov[back + i];
        }
        for (int i = back + lane; i < cap; i += kWave) { orow[i] = 0xFFFFu; ov[i] = 0.0f; }
    }
    __syncthreads();   // one wavefront: this is the wait that orders the stores above before the loads and stores below
    const int fu = (front + kCbUnit - 1) / kCbUnit;
    const int ntail = na + nlong;
    if (ntail > 0) {
        for (int i = lane; i < kCbRowsMax; i += kWave) tag[i] = 0;
        for (int i = lane; i < ntail; i += kWave) atomicAdd(&tag[t_row[i < na ? a0 + i : s + ns + (i - na)]], 1);
        (void)wave_scan_lds(tag, kCbRowsMax, lane, nullptr);
        for (int base = 0; base < ntail; base += kWave) {
            const int i = base + lane;
            const bool valid = i < ntail;
            const int src = i < na ? a0 + i : s + ns + (i - na);
            int col = 0, row = -1;
            float v = 0.0f;
            if (valid) {
                col = t_col[src];
                row = t_row[src];
                v = t_val[src];
            }
            const int first = valid ? tag[row] : 0;
            int rank, cnt;
            rank_same_key(row, valid, lane, rank, cnt);
            if (valid && rank == cnt - 1) tag[row] = first + cnt;
            if (valid) {
                const int d = first + rank;                 // position in the tail
                const int j = d & (kCbUnit - 1);
                const int slot = (fu + d / kCbUnit) * kCbUnit + (j & (kWave - 1)) * 4 + (j >> 6);
                oc[slot] = col;
                orow[slot] = (uint16_t)row;
                ov[slot] = v;
            }
        }
    }
    if (lane == 0) {
        usimple[b] = fu;
        uend[b] = u0 + fu + (ntail + kCbUnit - 1) / kCbUnit;
        atomicAdd(&stats[1], (unsigned long long)ntail);
    }
}

// tail units of a block = uend - ubeg - usimple (then scanned into tbeg: the block's first slot in the tail-row array)
__global__ void k_cb_tail_units(int nblocks, const int32_t *__restrict__ ubeg, const int32_t *__restrict__ usimple,
                                const int32_t *__restrict__ uend, int32_t *__restrict__ tcount)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) tcount[b] = uend[b] - ubeg[b] - usimple[b];
}

// Step 3: one wavefront per unit, o_col rewritten in place as packed:
//   a unit of groups:  ubase = smallest column of the unit, packed = row << 19 | column - ubase, empty slots = the lane's
//                      dummy row (fail |= 2 when the unit spans 2^19 columns or more);
//   a tail unit:       packed = the column itself (a long row's last nonzeros and the next row's first ones can lie a whole
//                      band apart), ubase = 0, the rows go to trow[] (0xFFFF = empty slot), compact over the tail units.
__global__ __launch_bounds__(kBlock) void k_cb_pack(int64_t units, int rows_cap, int nblocks, const int32_t *__restrict__ ubeg,
                                                    const int32_t *__restrict__ usimple, const int32_t *__restrict__ uend,
                                                    const int32_t *__restrict__ tbeg, uint32_t *__restrict__ packed,
                                                    const uint16_t *__restrict__ o_row, int32_t *__restrict__ ubase,
                                                    uint16_t *__restrict__ trow, int32_t *__restrict__ fail)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t u = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (u >= units) return;   // wave-uniform
    int lo = 0, hi = nblocks;   // the block of unit u: the last b with ubeg[b] <= u
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)ubeg[mid] <= u) lo = mid; else hi = mid;
    }
    const int b = lo;
    const int first_tail = ubeg[b] + usimple[b];
    const bool tail = u >= first_tail && u < uend[b];
    u4 *p4 = reinterpret_cast<u4 *>(packed + u * kCbUnit) + lane;
    const us4 r4 = *(reinterpret_cast<const us4 *>(o_row + u * kCbUnit) + lane);
    u4 c = *p4;
    if (tail) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r4[q] == 0xFFFFu) c[q] = 0u;
        *p4 = c;
        *(reinterpret_cast<us4 *>(trow + ((int64_t)tbeg[b] + (u - first_tail)) * kCbUnit) + lane) = r4;
        if (lane == 0) ubase[u] = 0;
        return;
    }
    int mn = INT_MAX, mx = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (r4[q] != 0xFFFFu) {
            const int col = (int)c[q];
            mn = col < mn ? col : mn;
            mx = col > mx ? col : mx;
        }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int a = __shfl_xor(mn, o, kWave), d = __shfl_xor(mx, o, kWave);
        mn = a < mn ? a : mn;
        mx = d > mx ? d : mx;
    }
    if (mx < 0) mn = 0;   // an empty unit
    const int colbits = cb_colbits(rows_cap);
    if (mx >= 0 && mx - mn >= (1 << colbits) && lane == 0) atomicOr(fail, 2);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        c[q] = r4[q] != 0xFFFFu ? ((unsigned)r4[q] << colbits) | (unsigned)((int)c[q] - mn)
                                : (unsigned)(rows_cap + lane) << colbits;
    *p4 = c;
    if (lane == 0) ubase[u] = mn;
}

// ---- the multiply -----------------------------------------------------------------------------------------------------
struct CbStream {      // one step of a wavefront's stream, as loaded
    u4 c[kCbVec];
    f4 v[kCbVec];
    us4 tr[kCbVec];    // tail units only: the rows
    int base[kCbVec];  // wave-uniform
};
struct CbGather {      // the same step between its gathers and its adds
    float xv[kCbVec][4];
    float val[kCbVec][4];
    int row[kCbVec][4];
};

template <int ROWS, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void k_colsort(int nblocks, const int32_t *__restrict__ brow,
                                                              const int32_t *__restrict__ ubeg,
                                                              const int32_t *__restrict__ usimple,
                                                              const int32_t *__restrict__ uend,
                                                              const int32_t *__restrict__ ubase,
                                                              const int32_t *__restrict__ tbeg,
                                                              const uint16_t *__restrict__ trow,
                                                              const uint32_t *__restrict__ packed,
                                                              const float *__restrict__ pvals,
                                                              const float *__restrict__ x, float *__restrict__ y)
{
    constexpr int kCbRows = ROWS, kCbWaves = WAVES;
    constexpr int kCbYs = ROWS + kWave;        // + one dummy row per lane (what the empty slots add their zeros to)
    constexpr int kCbColBits = cb_colbits(ROWS);
    constexpr unsigned kCbColMask = (1u << kCbColBits) - 1u;
    extern __shared__ __attribute__((aligned(16))) float ys_all[];   // WAVES copies of ROWS + 64 sums
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = xcd_block((int)blockIdx.x, nblocks);
    const int row0 = brow[b], n = brow[b + 1] - row0;
    float *ys = ys_all + w * kCbYs;
    for (int i = lane * 4; i < kCbRows; i += kWave * 4) *reinterpret_cast<f4 *>(ys + i) = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const int u0 = ubeg[b], u1 = uend[b];      // the units the block uses (it owns a few more)
    const int us = u0 + usimple[b];
    const us4 *tr4 = reinterpret_cast<const us4 *>(trow + (int64_t)tbeg[b] * kCbUnit);   // the rows of the block's tail units
    const int mine = (u1 - u0 - w + kCbWaves - 1) / kCbWaves;          // units w, w + 8, ... of the block
    const int nsteps = mine > 0 ? (mine + kCbVec - 1) / kCbVec : 0;
    const u4 *c4 = reinterpret_cast<const u4 *>(packed);
    const f4 *v4 = reinterpret_cast<const f4 *>(pvals);
    auto unit_of = [&](int st, int j) { return u0 + w + kCbWaves * (st * kCbVec + j); };

    auto load = [&](int st, CbStream &r) {
#pragma unroll
        for (int j = 0; j < kCbVec; ++j) {
            const int u = unit_of(st, j);
            if (u < u1) {   // wave-uniform
                r.c[j] = __builtin_nontemporal_load(&c4[(int64_t)u * kWave + lane]);
                r.v[j] = __builtin_nontemporal_load(&v4[(int64_t)u * kWave + lane]);
                r.base[j] = ubase[u];
                if (u >= us) r.tr[j] = tr4[(int64_t)(u - us) * kWave + lane];
            } else {        // past the block: the lane's dummy row, value 0
                const unsigned pad = (unsigned)(kCbRows + lane) << kCbColBits;
                r.c[j] = u4{pad, pad, pad, pad};
                r.v[j] = f4{0.0f, 0.0f, 0.0f, 0.0f};
                r.base[j] = 0;
            }
        }
    };
    auto gather = [&](int st, const CbStream &r, CbGather &g) {
#pragma unroll
        for (int j = 0; j < kCbVec; ++j) {
            const int u = unit_of(st, j);
            if (u >= us && u < u1) {   // wave-uniform: a tail unit -- whole columns, rows beside them
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    g.xv[j][q] = x[r.c[j][q]];
                    g.row[j][q] = r.tr[j][q] == 0xFFFFu ? kCbRows + lane : (int)r.tr[j][q];
                    g.val[j][q] = r.v[j][q];
                }
            } else {
                const float *xb = x + r.base[j];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#ifdef SPMV_CB_NOGATHER   // (A/B builds only: what the kernel costs without its gathers -- results are wrong)
                    g.xv[j][q] = __int_as_float((int)(r.c[j][q] & 0xffu));
#else
                    g.xv[j][q] = xb[r.c[j][q] & kCbColMask];
#endif
                    g.row[j][q] = (int)(r.c[j][q] >> kCbColBits);
                    g.val[j][q] = r.v[j][q];
                }
            }
        }
    };
    auto add = [&](int st, const CbGather &g) {
#pragma unroll
        for (int j = 0; j < kCbVec; ++j) {
            const int u = unit_of(st, j);
            if (u < us || u >= u1) {   // wave-uniform: a group -- every instruction holds 64 distinct rows (dummy rows included)
#ifdef SPMV_CB_NOADD      // (A/B builds only: what the kernel costs without its LDS adds -- results are wrong)
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += g.val[j][q] * g.xv[j][q] + (float)g.row[j][q];
                if (acc == 12345.678f) ys[lane] = acc;
#else
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = ys[g.row[j][q]];
                    ys[g.row[j][q]] = t + g.val[j][q] * g.xv[j][q];
                }
#endif
            } else {
                // the tail of the block: 64 CONSECUTIVE tail nonzeros per instruction, sorted by row -- runs of equal rows
                // (the long rows) are folded by a segmented scan over the lanes, and the last lane of every run adds: those
                // lanes hold distinct rows again (empty slots: the lane's own dummy row), so this is a plain add as well
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = g.row[j][q];
                    float p = g.val[j][q] * g.xv[j][q];
#pragma unroll
                    for (int d = 1; d < kWave; d <<= 1) {
                        const float t = __shfl_up(p, d, kWave);
                        const int r = __shfl_up(row, d, kWave);
                        if (lane >= d && r == row) p += t;
                    }
                    const int next = __shfl_down(row, 1, kWave);
                    if (lane == kWave - 1 || next != row) ys[row] = ys[row] + p;
                }
            }
        }
    };

    CbStream s0, s1;
    CbGather g0, g1;
    if (nsteps > 0) {
        load(0, s0);
        if (nsteps > 1) load(1, s1);
        gather(0, s0, g0);
        if (nsteps > 2) load(2, s0);
    }
    for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) {
            gather(st + 1, s1, g1);
            if (st + 3 < nsteps) load(st + 3, s1);
        }
        add(st, g0);
        if (st + 2 < nsteps) {
            gather(st + 2, s0, g0);
            if (st + 4 < nsteps) load(st + 4, s0);
        }
        if (st + 1 < nsteps) add(st + 1, g1);
    }
    __syncthreads();
    for (int i = tid; i < n; i += kCbWaves * kWave) {
        float acc = ys_all[i];
#pragma unroll
        for (int c = 1; c < kCbWaves; ++c) acc += ys_all[c * kCbYs + i];
        y[row0 + i] = acc;
    }
}

}  // namespace

void destroy_colsort(PanelPlan &p)
{
    if (p.d_ubeg) (void)hipFree(p.d_ubeg);
    if (p.d_usimple) (void)hipFree(p.d_usimple);
    if (p.d_uend) (void)hipFree(p.d_uend);
    if (p.d_tbeg) (void)hipFree(p.d_tbeg);
    if (p.d_trow) (void)hipFree(p.d_trow);
    if (p.d_ubase) (void)hipFree(p.d_ubase);
    p.d_ubeg = p.d_usimple = p.d_uend = p.d_ubase = p.d_tbeg = nullptr;
    p.d_trow = nullptr;
}

// A cheap look at the matrix before SPMV_AUTO pays for this plan (one pass over row_ptr, two columns per row):
//   long_frac = share of the nonzeros that sit in rows of more than 256 (they would go to the tails),
//   wide_frac = share of the 4096-row blocks whose short rows reach over 2^20 columns or more (first and last column of
//               every row: exact for sorted rows) -- their window of x is beyond one XCD's L2 and a line costs a fabric request.
__global__ __launch_bounds__(kBlock) void k_cb_probe(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col_idx, int32_t *__restrict__ bmin,
                                                     int32_t *__restrict__ bmax, unsigned long long *__restrict__ long_nnz)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long mine = 0ull;
    int lo = INT_MAX, hi = -1;
    if (r < rows) {
        const int32_t a = row_ptr[r], b = row_ptr[r + 1];
        if (b - a > 256) {
            mine = (unsigned long long)(b - a);
        } else if (b > a) {
            const int32_t c0 = col_idx[a], c1 = col_idx[b - 1];
            lo = c0 < c1 ? c0 : c1;
            hi = c0 > c1 ? c0 : c1;
        }
    }
    // the 64 rows of a wavefront lie in one 4096-row block (kBlock divides 4096): one atomic per wavefront, not per row
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mine += __shfl_down(mine, o, kWave);
        const int a = __shfl_down(lo, o, kWave), b = __shfl_down(hi, o, kWave);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (mine) atomicAdd(long_nnz, mine);
        if (hi >= 0) {
            const int64_t r_first = r;   // lane 0's row
            atomicMin(&bmin[r_first >> 12], lo);
            atomicMax(&bmax[r_first >> 12], hi);
        }
    }
}
// ... and lines = an estimate of the distinct lines of x the blocks touch: S (1 - exp(-n / S)) for a block of n nonzeros
// whose short rows span S lines (what n uniform draws from S lines occupy)
__global__ void k_cb_probe_wide(int nb, int64_t rows, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ bmin,
                                const int32_t *__restrict__ bmax, unsigned long long *__restrict__ wide)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || bmax[b] < 0) return;
    const int64_t span = (int64_t)bmax[b] - bmin[b];
    if (span >= (1 << 20)) atomicAdd(&wide[0], 1ull);
    const int64_t r0 = (int64_t)b << 12, r1 = r0 + 4096 < rows ? r0 + 4096 : rows;
    const double n = (double)(row_ptr[r1] - row_ptr[r0]), S = (double)(span / 32 + 1);
    atomicAdd(&wide[1], (unsigned long long)(S * (1.0 - exp(-n / S)) + 0.5));
}

// The layout for blocks of at most rows_cap rows (4096 | 8192).  Fills p (brow, packed, pvals, units, tail rows,
// statistics); on failure p owns nothing new.
static int build_colsort(spmv_csr &h, PanelPlan &p, int rows_cap, hipStream_t s)
{
    int rc;
    // blocks of rows_cap rows where the rows are equally long (config 2: 256 blocks of 4096 for 256 CUs); where equal
    // nonzero counts make many cuts longer than that, a lower target so that few cuts have to be split in two
    DevPtr<int32_t> brow;
    int64_t nb0 = (h.rows + rows_cap - 1) / rows_cap;
    if ((rc = panel_row_blocks(h, nb0, rows_cap, s, brow, &p.nblocks))) return rc;
    if ((int64_t)p.nblocks * 100 > nb0 * 101) {
        (void)hipFree(brow.release());
        nb0 = (h.rows + rows_cap * 15 / 16 - 1) / (rows_cap * 15 / 16);
        if ((rc = panel_row_blocks(h, nb0, rows_cap, s, brow, &p.nblocks))) return rc;
    }
    DevPtr<int32_t> ubeg, total, usimple, uend, nshort, fail;
    DevPtr<unsigned long long> stats;
    SPMV_HIP_TRY(ubeg.alloc((size_t)p.nblocks + 1));
    SPMV_HIP_TRY(total.alloc(1));
    SPMV_HIP_TRY(usimple.alloc((size_t)p.nblocks));
    SPMV_HIP_TRY(uend.alloc((size_t)p.nblocks));
    SPMV_HIP_TRY(nshort.alloc((size_t)p.nblocks));
    SPMV_HIP_TRY(fail.alloc(1));
    SPMV_HIP_TRY(stats.alloc(3));
    SPMV_HIP_TRY(hipMemsetAsync(fail.p, 0, sizeof(int32_t), s));
    SPMV_HIP_TRY(hipMemsetAsync(stats.p, 0, 3 * sizeof(unsigned long long), s));
    const unsigned gb = (unsigned)((p.nblocks + 255) / 256);
    k_cb_units<<<dim3(gb), dim3(256), 0, s>>>(p.nblocks, brow.p, h.d_row_ptr, ubeg.p);
    if ((rc = check_launch("k_cb_units"))) return rc;
    if ((rc = exclusive_scan_i32(ubeg.p, p.nblocks, total.p, s))) return rc;
    int32_t units = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&units, total.p, sizeof units, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipMemcpyAsync(ubeg.p + p.nblocks, total.p, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    p.units = units;
    const size_t slots = (size_t)units * kCbUnit;
    DevPtr<uint32_t> packed;
    DevPtr<float> pvals, t_val;
    DevPtr<int32_t> ubase, t_col;
    DevPtr<uint16_t> rowloc, t_row, o_row;
    SPMV_HIP_TRY(packed.alloc(slots));
    SPMV_HIP_TRY(pvals.alloc(slots));
    SPMV_HIP_TRY(ubase.alloc((size_t)units));
    SPMV_HIP_TRY(o_row.alloc(slots));
    SPMV_HIP_TRY(rowloc.alloc((size_t)h.nnz));
    SPMV_HIP_TRY(t_col.alloc((size_t)h.nnz));
    SPMV_HIP_TRY(t_row.alloc((size_t)h.nnz));
    SPMV_HIP_TRY(t_val.alloc((size_t)h.nnz));
    SPMV_HIP_TRY(hipMemsetAsync(packed.p, 0, sizeof(uint32_t) * slots, s));
    SPMV_HIP_TRY(hipMemsetAsync(pvals.p, 0, sizeof(float) * slots, s));
    SPMV_HIP_TRY(hipMemsetAsync(o_row.p, 0xFF, sizeof(uint16_t) * slots, s));
    if (h.nnz > 0) {
        if ((rc = panel_rowloc(h, brow.p, p.nblocks, rowloc.p, s))) return rc;
        // counters for the widest block at 32 columns each (wider blocks than kCbBins counters reach use coarser bins)
        int32_t widest = 0;
        SPMV_HIP_TRY(hipMemsetAsync(total.p, 0, sizeof(int32_t), s));
        k_cb_span<<<dim3((unsigned)p.nblocks), dim3(kBlock), 0, s>>>(brow.p, h.d_row_ptr, h.d_col_idx, total.p);
        if ((rc = check_launch("k_cb_span"))) return rc;
        SPMV_HIP_TRY(hipMemcpyAsync(&widest, total.p, sizeof widest, hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        int nbins = 1024;
        while (nbins < kCbBins && (widest >> 5) >= nbins) nbins <<= 1;
        const size_t lds = sizeof(int) * (size_t)nbins;
        static LdsOptIn optin;
        if ((rc = optin.ensure(reinterpret_cast<const void *>(&k_cb_sort), h.device, (int)(sizeof(int) * (size_t)kCbBins)))) return rc;
        k_cb_sort<<<dim3((unsigned)p.nblocks), dim3(kBlock), lds, s>>>(brow.p, h.d_row_ptr, h.d_col_idx, h.d_vals, rowloc.p,
                                                                        t_col.p, t_row.p, t_val.p, nshort.p, stats.p, nbins);
        if ((rc = check_launch("k_cb_sort"))) return rc;
        k_cb_groups<<<dim3((unsigned)p.nblocks), dim3(kWave), 0, s>>>(brow.p, h.d_row_ptr, ubeg.p, nshort.p, t_col.p, t_row.p,
                                                                       t_val.p, reinterpret_cast<int32_t *>(packed.p), o_row.p,
                                                                       pvals.p, usimple.p, uend.p, stats.p);
        if ((rc = check_launch("k_cb_groups"))) return rc;
    } else {
        SPMV_HIP_TRY(hipMemsetAsync(usimple.p, 0, sizeof(int32_t) * (size_t)p.nblocks, s));
        SPMV_HIP_TRY(hipMemcpyAsync(uend.p, ubeg.p, sizeof(int32_t) * (size_t)p.nblocks, hipMemcpyDeviceToDevice, s));   // no units in use
    }
    // the rows of the tail units, compact: tbeg[b] = the block's first unit in trow
    DevPtr<int32_t> tbeg, ttotal;
    DevPtr<uint16_t> trow;
    SPMV_HIP_TRY(tbeg.alloc((size_t)p.nblocks));
    SPMV_HIP_TRY(ttotal.alloc(1));
    k_cb_tail_units<<<dim3(gb), dim3(256), 0, s>>>(p.nblocks, ubeg.p, usimple.p, uend.p, tbeg.p);
    if ((rc = check_launch("k_cb_tail_units"))) return rc;
    if ((rc = exclusive_scan_i32(tbeg.p, p.nblocks, ttotal.p, s))) return rc;
    int32_t tunits = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&tunits, ttotal.p, sizeof tunits, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    p.tail_units = tunits;
    SPMV_HIP_TRY(trow.alloc((size_t)tunits * kCbUnit));
    k_cb_pack<<<dim3((unsigned)((units + 3) / 4)), dim3(kBlock), 0, s>>>((int64_t)units, rows_cap, p.nblocks, ubeg.p, usimple.p, uend.p,
                                                                         tbeg.p, packed.p, o_row.p, ubase.p, trow.p, fail.p);
    if ((rc = check_launch("k_cb_pack"))) return rc;
    int32_t failed = 0;
    unsigned long long st[3] = {0, 0, 0};
    SPMV_HIP_TRY(hipMemcpyAsync(&failed, fail.p, sizeof failed, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipMemcpyAsync(st, stats.p, sizeof st, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // (the temporaries are freed on return)
    if (failed) {
        set_error("spmv_csr_plan(panel, sorted blocks): 256 neighbouring nonzeros (in column order) of the short rows of a block of "
                  "%d rows span 2^%d columns or more -- this layout is for matrices with at least that much to share in a line of x",
                  rows_cap, cb_colbits(rows_cap));
        return SPMV_ERR_INVALID;
    }
    p.sb_rows = rows_cap;
    p.lines = (int64_t)st[0];
    p.tail = (int64_t)st[1];
    p.wide_blocks = (int64_t)st[2];
    p.d_packed = packed.release();
    p.d_pvals = pvals.release();
    p.d_brow = brow.release();
    p.d_ubeg = ubeg.release();
    p.d_usimple = usimple.release();
    p.d_uend = uend.release();
    p.d_tbeg = tbeg.release();
    p.d_trow = trow.release();
    p.d_ubase = ubase.release();
    return SPMV_OK;
}

int colsort_probe(const spmv_csr &h, hipStream_t s, double *long_frac, double *wide_frac, double *lines_per_nnz)
{
    *long_frac = 0.0;
    *wide_frac = 0.0;
    *lines_per_nnz = 0.0;
    if (h.rows == 0 || h.nnz == 0) return SPMV_OK;
    const int nb = (int)((h.rows + 4095) >> 12);
    DevPtr<int32_t> bmin, bmax;
    DevPtr<unsigned long long> cnt;
    SPMV_HIP_TRY(bmin.alloc((size_t)nb));
    SPMV_HIP_TRY(bmax.alloc((size_t)nb));
    SPMV_HIP_TRY(cnt.alloc(3));
    SPMV_HIP_TRY(hipMemsetAsync(bmin.p, 0x7f, sizeof(int32_t) * (size_t)nb, s));
    SPMV_HIP_TRY(hipMemsetAsync(bmax.p, 0xff, sizeof(int32_t) * (size_t)nb, s));   // -1
    SPMV_HIP_TRY(hipMemsetAsync(cnt.p, 0, 3 * sizeof(unsigned long long), s));
    k_cb_probe<<<dim3((unsigned)((h.rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(h.rows, h.d_row_ptr, h.d_col_idx, bmin.p, bmax.p,
                                                                                         cnt.p);
    int rc = check_launch("k_cb_probe");
    if (rc) return rc;
    k_cb_probe_wide<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s>>>(nb, h.rows, h.d_row_ptr, bmin.p, bmax.p, cnt.p + 1);
    if ((rc = check_launch("k_cb_probe_wide"))) return rc;
    unsigned long long c[3] = {0, 0, 0};
    SPMV_HIP_TRY(hipMemcpyAsync(c, cnt.p, sizeof c, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    *long_frac = (double)c[0] / (double)h.nnz;
    *wide_frac = (double)c[1] / (double)nb;
    *lines_per_nnz = (double)c[2] / (double)h.nnz;
    return SPMV_OK;
}

static void release_colsort(PanelPlan &p)
{
    destroy_colsort(p);
    if (p.d_packed) (void)hipFree(p.d_packed);
    if (p.d_pvals) (void)hipFree(p.d_pvals);
    if (p.d_brow) (void)hipFree(p.d_brow);
    p.d_packed = nullptr;
    p.d_pvals = nullptr;
    p.d_brow = nullptr;
}

// Modelled time per nonzero of a sorted-blocks plan in the units of kernels_adaptive.hip's PlanCost (1 = a chunk that
// streams 8 bytes per nonzero with cache-resident gathers), fitted to config 4 with bands of 8192 ... 1M columns and config 2
// with uniform columns on two MI355X (profiles/r03_sorted_blocks_calibration.jsonl).  SPMV_AUTO compares it with the TILED
// plan's model_cost.
double colsort_cost(int rows_per_block, double lines_per_nnz, double tail_frac)
{
    // the stream and the adds bound the kernel up to ~0.1 lines per nonzero (1.20; 1.6 with 4 wavefronts per CU on 8192-row
    // blocks), the lines of x beyond that: 2.5 per line and nonzero
    const double c = rows_per_block > 4096 ? (1.25 + 2.5 * lines_per_nnz > 1.6 ? 1.25 + 2.5 * lines_per_nnz : 1.6)
                                           : (0.99 + 2.5 * lines_per_nnz > 1.20 ? 0.99 + 2.5 * lines_per_nnz : 1.20);
    return c + tail_frac;
}

double colsort_model_cost(const PanelPlan &p, int64_t nnz)
{
    if (!p.sorted_mode || nnz <= 0) return 0.0;
    return colsort_cost(p.sb_rows, (double)p.lines / (double)nnz, (double)p.tail / (double)nnz);
}

// want_rows: 0 = the rule below | 4096 | 8192; want_waves: 0 = the rule | 4 | 8 (8 only with 4096 rows)
int plan_colsort(spmv_csr &h, PanelPlan &p, int want_rows, int want_waves, hipStream_t s)
{
    p.sorted_mode = true;
    p.lds_mode = false;
    p.pw_bits = 0;
    p.npanels = 0;
    p.waves_per_launch = 0;
    if ((want_rows != 0 && want_rows != 4096 && want_rows != 8192) || (want_waves != 0 && want_waves != 4 && want_waves != 8) ||
        (want_rows == 8192 && want_waves == 8)) {
        set_error("spmv_csr_plan(panel, sorted blocks): rows per block %d / wavefronts %d outside 4096|8192 / 4|8 (8192 x 8 does not fit LDS)",
                  want_rows, want_waves);
        return SPMV_ERR_INVALID;
    }
    if (h.rows == 0) {
        p.sb_rows = want_rows ? want_rows : 4096;
        p.sb_waves = want_waves ? want_waves : 8;
        p.stamp.gen = h.values_gen;     // nothing was copied, but the plan is as fresh as any other made now
        p.stamp.have_sum = false;
        p.ready = true;
        return SPMV_OK;
    }
    const int cus = device_cus(h.device);
    int rc = build_colsort(h, p, want_rows ? want_rows : 4096, s);
    if (rc) { release_colsort(p); return rc; }
    auto waves_for = [&](const PanelPlan &q) {
        if (q.sb_rows > 4096) return 4;
        if (want_waves) return want_waves;
        return q.nblocks >= 4 * cus ? 4 : 8;   // two workgroups of 4 per CU need blocks to go round
    };
    p.sb_waves = waves_for(p);
    // 8192-row blocks: half the lines of x per nonzero, at half the wavefronts per CU -- worth trying from 0.27 lines per
    // nonzero when there are blocks enough to fill the chip with 4 wavefronts per CU (a band of 1M columns at config 4:
    // 0.44 -> 0.25 lines per nonzero, 37 -> 41 % of peak); kept when the model prices it lower.
    if (want_rows == 0 && h.nnz > 0 && (double)p.lines > 0.27 * (double)h.nnz && p.nblocks >= 16 * cus) {
        PanelPlan q = p;
        q.d_packed = nullptr; q.d_pvals = nullptr; q.d_brow = nullptr;
        q.d_ubeg = q.d_usimple = q.d_uend = q.d_ubase = q.d_tbeg = nullptr;
        q.d_trow = nullptr;
        const int rc8 = build_colsort(h, q, 8192, s);
        if (rc8 == SPMV_OK) {
            q.sb_waves = 4;
            if (colsort_model_cost(q, h.nnz) < colsort_model_cost(p, h.nnz)) {
                release_colsort(p);
                p = q;
            } else {
                release_colsort(q);
            }
        } else {
            release_colsort(q);
        }
    }
    if ((rc = stamp_values(h, s, p.stamp))) { release_colsort(p); return rc; }
    p.ready = true;
    return SPMV_OK;
}

template <int ROWS, int WAVES>
static int launch_colsort_t(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s)
{
    const size_t lds = sizeof(float) * (size_t)WAVES * (ROWS + kWave);
    static LdsOptIn optin;
    if (int rc = optin.ensure(reinterpret_cast<const void *>(&k_colsort<ROWS, WAVES>), h.device, (int)lds)) return rc;
    k_colsort<ROWS, WAVES><<<dim3((unsigned)p.nblocks), dim3(WAVES * kWave), lds, s>>>(
        p.nblocks, p.d_brow, p.d_ubeg, p.d_usimple, p.d_uend, p.d_ubase, p.d_tbeg, p.d_trow, p.d_packed, p.d_pvals, x, y);
    return check_launch("k_colsort");
}

int launch_colsort(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s)
{
    if (p.nblocks == 0) return SPMV_OK;
    if (p.sb_rows > 4096) return launch_colsort_t<8192, 4>(h, p, x, y, s);
    return p.sb_waves == 4 ? launch_colsort_t<4096, 4>(h, p, x, y, s) : launch_colsort_t<4096, 8>(h, p, x, y, s);
}

}  // namespace spmv
