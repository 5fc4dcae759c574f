// K10 colours-only backward on the matrix cores (the GAD flow consumes only d loss / d colors:
// scene/gaussian_model.py:192-208):     v_colors[g, :] = sum_px w[px, g] * v_out[px, :],  w = alpha*T.
//
// Per wave the cotangent slab of its pixel block x 128 channels lives in VGPRs as MFMA B operands (K = pixel
// pairs, N = channels); A operands are 32-slot weight tiles (rows = slots); a tile costs (pixels/2) K-steps x
// 4 channel tiles of v_mfma_f32_32x32x2_f32 and yields 32 partial gradient rows of 128 channels.
//
// What happens to those rows is the whole story on this part: float atomics are executed at the memory
// side on a multi-XCD MI355X (TCC_EA0_ATOMIC == TCC_ATOMIC, ~1.2 TB/s measured) while plain stores of the
// same rows are almost free.  So the default ("staged") path has NO atomics and is bit-reproducible:
//   rows    one wave per (tile, 8x8 block, slice of 128 / 64 / 32 channels): weight tile from the forward's
//           scratch (wt), 32 MFMAs per 32 channels, 32 rows stored at fixed addresses (prefix sum of the forward's slot counts);
//   sort    (Gaussian id, row) pairs, radix sort on 32-bit keys; per-Gaussian offsets;
//   reduce  v_colors[g] = sum of its rows, written once (no zero-fill of v_colors needed).
// The single-kernel atomic variant (recomputes alpha itself; needs neither scratch nor the forward's
// slot counts) is kept as the fallback.
#include <cstdlib>
#include <hip/hip_fp16.h>
#include "raster_mfma_common.h"

using namespace gags_mfma;

int64_t gags_sort_u32_scratch_bytes(int64_t n);
int gags_sort_pairs_u32(int64_t n, int nbits, const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out,
                        int32_t *vals_out, void *scratch, int64_t scratch_bytes, hipStream_t st);

namespace {

constexpr int NBB = 4;         // channel tiles per wave
constexpr int CSB = 32 * NBB;  // 128 channels per wave

__device__ __forceinline__ void atomic_add_f32(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// cotangent slab as B operands: V[s][j] = v_out[pixel q = 2s+k][ch0 + 32j + p]
__device__ __forceinline__ void load_slab(float (&V)[16][NBB], const float *__restrict__ v_out, const BlockGeom &g,
                                          int width, int height, int d, int ch0)
{
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int q = 2 * s + g.k;
        const int qj = g.bx0 + (q & 7), qi = g.by0 + (q >> 3);
        const bool ok = (qi < height) && (qj < width);
        const float *src = v_out + ((size_t)(ok ? qi : 0) * width + (ok ? qj : 0)) * d + ch0 + g.p;
#pragma unroll
        for (int j = 0; j < NBB; ++j) V[s][j] = ok ? src[32 * j] : 0.f;
    }
}

__device__ __forceinline__ void tile_mfma(f32x16 (&acc)[NBB], const float (&A)[16], const float (&V)[16][NBB])
{
#pragma unroll
    for (int j = 0; j < NBB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int j = 0; j < NBB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], V[s][j], acc[j], 0, 0, 0);
}

// ---- staged: rows ------------------------------------------------------------------------------------
// One workgroup per (tile, channel slice); wave b of its four owns the tile's 8x8 pixel block b.  K = the block's 64
// pixels in the order of the weight rows raster_weights wrote (32 (upper, lower) pairs): K-step t pairs element t of
// the row's first half (k = 0) with element t of its second half (k = 1).  The wave's cotangent slab (64 px x 32*NBR
// channels) sits in 32*NBR VGPRs as B operands for the whole tile.
//
// A Gaussian that blends into several blocks of the tile leaves ONE gradient row for the tile, not one per block:
// the tile's rows -- numbered by trow[] = exclusive prefix sum of the forward's hit[] flags, i.e. in sorted order --
// are produced in chunks.  A chunk [r0, r1) ends where a block would need a 33rd slot (its run of slots inside the
// chunk is one 32-row MFMA tile) or after CMAX rows.  Per chunk every wave multiplies its run (the 128-MFMA burst
// of one weight tile), parks the partial rows in LDS, and after a barrier the workgroup adds the up to four partial
// rows of every tile row in a fixed order (block 0..3: bit-reproducible) and stores the merged row once, 128 B
// per lane-quad.  Compared with one row per (block, Gaussian) this halves the rows written, sorted and reduced
// (C3: 4.44 M -> ~2 M rows of 2 KB) for ~10 % more MFMA issue (runs are on average 27 of 32 slots long).
constexpr int CMAX = 64;  // merged rows per chunk (bounds pos[] and the merge loop)

template <int NBR>
__global__ __launch_bounds__(256, (NBR == 4 ? 2 : (NBR == 2 ? 3 : 4))) void raster_bwd_rows(
    int d, int width, int height, int tile_w, int n_tiles, int ch_base, int n_slices,
    const float *__restrict__ v_render_colors, const int32_t *__restrict__ offsets, int n_isects,
    const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ trow, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const int32_t *__restrict__ trow_s, float *__restrict__ prow, int prow_pitch,
    uint32_t *__restrict__ row_key, int32_t *__restrict__ row_idx, int rows_cap)
{
    constexpr int CW = 32 * NBR;  // channels per slice; this launch covers channels ch_base .. ch_base + n_slices * CW - 1 (clipped to d)
    constexpr int C4 = CW / 4;    // float4 columns per row of the slice
    __shared__ __attribute__((aligned(16))) float stage[4][32][CW];  // partial rows of the chunk, per block
    __shared__ __attribute__((aligned(4))) uint8_t pos[2][CMAX][4];  // pos[parity][row - r0][b] = slot of block b's run holding that tile row, 0xff: none (one 32-bit read per row)
    __shared__ int cand[2][4];
    __shared__ __attribute__((aligned(16))) float zrow[CW];  // a row of zeros for the merge

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < CW) zrow[threadIdx.x] = 0.f;  // (visible after the first chunk's barrier)
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int tile = gags_tile_of_order(logical / n_slices, tile_w, n_tiles / tile_w);
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int R0 = trow[start], R1 = trow[end];
    if (R1 == R0) return;  // nothing blended in this tile (uniform over the workgroup)
    const int blk = wave;
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    const int sb = gags_slot_base(start, end, tile, blk);
    const int ch0 = ch_base + (logical % n_slices) * CW;
    BlockGeom64 g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;

    // V[t][j] = v_out[pixel t of half k][ch0 + NBR*p + j]  ("strided-NBR" channel tiles: one vector load / store)
    float V[32][NBR];
    if (cnt > 0) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            // K-step t of half-wave k = pixel 16k + t/2 of the 8x4 half t%2: the order of the weight rows
            const int px = 16 * k + (t >> 1);
            const int qj = g.bx0 + (px & 7), qi = g.by0 + 4 * (t & 1) + (px >> 3);
            const bool ok = (qi < height) && (qj < width);
            // NBR == 1 also serves a ragged last slice (D % 32 != 0): lanes past the row are clamped here, masked below
            const float *src = v_render_colors + ((size_t)min(qi, height - 1) * width + min(qj, width - 1)) * d +
                               (NBR == 1 ? min(ch0 + p, d - 1) : ch0 + NBR * p);
            if constexpr (NBR == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src);
                V[t][0] = ok ? v.x : 0.f; V[t][1] = ok ? v.y : 0.f; V[t][2] = ok ? v.z : 0.f; V[t][3] = ok ? v.w : 0.f;
            } else if constexpr (NBR == 2) {
                const float2 v = *reinterpret_cast<const float2 *>(src);
                V[t][0] = ok ? v.x : 0.f; V[t][1] = ok ? v.y : 0.f;
            } else {
                const float v = src[0];
                V[t][0] = ok ? v : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 32; ++t)
#pragma unroll
            for (int j = 0; j < NBR; ++j) V[t][j] = 0.f;
    }

    int pb = 0;  // slots of this block already consumed
    int r0 = R0;
    // tile rows (and Gaussians) of this block's next 33 slots: sorted, hence increasing; 0x7fffffff past the end and
    // for the pad slot of an odd count.  Always fetched one chunk ahead.
    int tr = 0x7fffffff, gid = 0;
    if (lane <= 32 && lane < cnt) {
        tr = trow_s[sb + lane];
        gid = gid_s[sb + lane];
    }
    // weight tile: slots sb+pb .. +31; lane (i = p, k) owns the 32 floats of half k of row i.  Rows past the run (the
    // next chunk's slots, or memory past the block) only produce accumulator rows nobody stores.  Also requested one
    // chunk ahead: right after the previous burst has consumed the registers.
    float A[32];
    auto load_A = [&](int first) {
        const float4 *src = reinterpret_cast<const float4 *>(wt + (size_t)(sb + first + p) * 64 + k * 32);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 v = src[t];
            A[4 * t] = v.x; A[4 * t + 1] = v.y; A[4 * t + 2] = v.z; A[4 * t + 3] = v.w;
        }
    };
    if (cnt > 0) load_A(0);
    for (int it = 0; r0 < R1; ++it) {
        const int par = it & 1;
        if (threadIdx.x < CMAX) reinterpret_cast<uint32_t *>(&pos[par][0][0])[threadIdx.x] = 0xffffffffu;
        const int tr32 = __builtin_amdgcn_readlane(tr, 32);
        if (lane == 0) cand[par][wave] = tr32;
        gags_lds_barrier();  // LDS traffic only: __syncthreads() would also wait for the loads in flight (next chunk's weight tile) and the row stores
        const int r1 = min(min(min(cand[par][0], cand[par][1]), min(cand[par][2], cand[par][3])), min(r0 + CMAX, R1));
        const bool mine = lane < 32 && tr < r1;
        const int run = __popcll(__ballot(mine));  // this block's slots pb .. pb+run-1 fall into [r0, r1)
        const int tr_c = tr, gid_c = gid;
        const int pbn = pb + run;
        if (run > 0) {
            if (mine) pos[par][tr_c - r0][wave] = (uint8_t)lane;
            // The 32*NBR MFMAs of a weight tile are issued as ONE uninterrupted burst: everything they read is waited
            // for up front and nothing else is scheduled into the burst, so that the waves sharing a SIMD alternate
            // (one multiplies while the other loads / merges) instead of stalling and resuming in lock step.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[NBR];
#pragma unroll
            for (int j = 0; j < NBR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int j = 0; j < NBR; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t], V[t][j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (mine && ch0 == 0 && tr_c < rows_cap) {  // row -> Gaussian map for the sort (every block of the row stores the same pair; the call that covers channel 0 writes it)
                row_key[tr_c] = (uint32_t)gid_c;
                row_idx[tr_c] = tr_c;
            }
            // next chunk's operands: in flight while this chunk is parked and merged
            tr = 0x7fffffff;
            if (lane <= 32 && pbn + lane < cnt) {
                tr = trow_s[sb + pbn + lane];
                gid = gid_s[sb + pbn + lane];
            }
            if (pbn < cnt) load_A(pbn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
                if (slot < run) {
                    float *dst = &stage[wave][slot][NBR * p];
                    if constexpr (NBR == 4) *reinterpret_cast<float4 *>(dst) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                    else if constexpr (NBR == 2) *reinterpret_cast<float2 *>(dst) = make_float2(acc[0][r], acc[1][r]);
                    else dst[0] = acc[0][r];
                }
            }
        }
        gags_lds_barrier();  // LDS traffic only: __syncthreads() would also wait for the loads in flight (next chunk's weight tile) and the row stores
        // merged rows of the chunk: sum over the blocks that hold the row, in block order; one float4 per thread and item
        const int items = (r1 - r0) * C4;
        // branch-free, fully unrolled (at most CMAX * C4 / 256 trips): see raster_bwd_rows_pair
        int gt = threadIdx.x;
        asm volatile("" : "+v"(gt));
#pragma unroll
        for (int trip = 0; trip < CMAX * C4 / 256; ++trip) {
            const int item = gt + 256 * trip;
            if (item < items) {
                const int row = item / C4, c4 = item - row * C4;
                // a block that does not hold the row reads a row of zeros instead (one select on the address, not
                // four on the values: x + 0 is exact)
                float4 v[4];
                const uint32_t q4 = *reinterpret_cast<const uint32_t *>(&pos[par][row][0]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int q = (q4 >> (8 * b)) & 0xff;
                    v[b] = *(q != 0xff ? reinterpret_cast<const float4 *>(&stage[b][q][4 * c4]) : reinterpret_cast<const float4 *>(&zrow[4 * c4]));
                }
                float4 sum;
                sum.x = ((v[0].x + v[1].x) + v[2].x) + v[3].x; sum.y = ((v[0].y + v[1].y) + v[2].y) + v[3].y;
                sum.z = ((v[0].z + v[1].z) + v[2].z) + v[3].z; sum.w = ((v[0].w + v[1].w) + v[2].w) + v[3].w;
                float *dst = prow + (size_t)(r0 + row) * prow_pitch + ch0 + 4 * c4;
                if (r0 + row >= rows_cap) {
                    // (capacity-sized scratch and more rows than it holds: nothing is stored; the caller sees the count
                    // afterwards and runs the backward again with the right size)
                } else if (NBR != 1 || ch0 + 4 * c4 + 3 < d) {
                    *reinterpret_cast<float4 *>(dst) = sum;
                } else {  // ragged last slice (D % 32 != 0), possibly D % 4 != 0: never store past the row
                    const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch0 + 4 * c4 + e < d) dst[e] = sv[e];
                }
            }
            if (trip & 1) __builtin_amdgcn_sched_barrier(0);
        }
        pb = pbn;
        r0 = r1;
        // no third barrier: the next chunk's stores into stage[] / pos[par ^ 1] / cand[par ^ 1] come after ITS first
        // barrier, which every thread reaches only after finishing the loop above
    }
}

// ---- staged: rows on the 16-bit matrix cores, fp32-equivalent (the DEFAULT; GAGS_BWD_F32MFMA selects the kernel above) ----
// The same kernel with the contraction on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate).  Operands are written as sums of
// fp16 terms, each obtained by round-to-nearest of what the previous terms left over, after an exact power-of-two scaling:
//   weights    w = (a0 + a1 + a2) / rs   three terms: 33 significand bits >= fp32's 24, i.e. EXACT; rs = one power of two
//                                        per slot row (its largest weight -> [2^14, 2^15): heads and tails of the weights
//                                        that matter are normal fp16 numbers);
//   cotangent  v = (b0 + b1) / cs        two terms: |v cs - b0 - b1| <= 2^-24 |v cs| (each rounding leaves at most half an
//                                        ulp of an 11-bit significand): ONE fp32 rounding of the input; cs = one power of two
//                                        per (pixel block, channel), from the column's largest magnitude;
// and a product as the five terms of order <= 2:   w v ~ a0 b0 + a0 b1 + a1 b0 + a1 b1 + a2 b0   (dropped: a2 b1, 2^-36).
// Every partial product of two fp16 numbers is exact in the fp32 accumulator.  Net effect: each product w v enters the sum
// with a relative error <= 2^-24 -- the cotangent rounded once -- which is below the rounding an fp32 dot product of these
// 64-pixel columns commits in its own additions.  Against float64 (tests/test_fullsize_gpu.py
// ::test_colour_gradient_accuracy_against_float64) it is at least as close as the fp32-MFMA kernel; no atomics, fixed
// order: bit-reproducible.  A burst is 80 MFMAs of 32 cycles instead of 128 of 64.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Phase clocks of the rows kernel's waves (tools/probe/: a SEPARATE probe build, -DGAGS_PROBE; never the shipped library)
#ifdef GAGS_PROBE
}  // namespace
__device__ unsigned long long *gags_probe_buf_bwd = nullptr;
extern "C" __attribute__((visibility("default"))) int gags_probe_set_bwd(void *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(gags_probe_buf_bwd), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
namespace {
#define GAGS_PH_DECL unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tl_ = __builtin_readcyclecounter()
#define GAGS_PH(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pt_[i] += now_ - tl_; tl_ = now_; } while (0)
#define GAGS_PH_WRITE()                                                                                                  \
    do {                                                                                                                 \
        if (gags_probe_buf_bwd && lane == 0)                                                                             \
            for (int i_ = 0; i_ < 8; ++i_) gags_probe_buf_bwd[((size_t)blockIdx.x * 4 + wave) * 8 + i_] = pt_[i_];        \
    } while (0)
#else
#define GAGS_PH_DECL ((void)0)
#define GAGS_PH(i) ((void)0)
#define GAGS_PH_WRITE() ((void)0)
#endif

__device__ __forceinline__ void split8(const float (&x)[8], float scale, f16x8 &hi, f16x8 &lo)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = x[i] * scale;
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
}

// pairs through v_cvt_pk_f16_f32 (round to nearest even, new on gfx950): one conversion instruction per two values and term
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8x3(const float (&x)[8], float scale, f16x8 &t0, f16x8 &t1, f16x8 &t2)
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2v_t v = {x[i] * scale, x[i + 1] * scale};
        const f16x2_t h = __builtin_convertvector(v, f16x2_t);
        const f32x2v_t r = v - __builtin_convertvector(h, f32x2v_t);
        const f16x2_t m = __builtin_convertvector(r, f16x2_t);
        const f32x2v_t q = r - __builtin_convertvector(m, f32x2v_t);
        const f16x2_t l = __builtin_convertvector(q, f16x2_t);
        t0[i] = h[0]; t0[i + 1] = h[1];
        t1[i] = m[0]; t1[i + 1] = m[1];
        t2[i] = l[0]; t2[i + 1] = l[1];
    }
}

__global__ __launch_bounds__(256, 2) void raster_bwd_rows_f16(
    int d, int width, int height, int tile_w, int n_tiles, int ch_base, int n_slices,
    const float *__restrict__ v_render_colors, const int32_t *__restrict__ offsets, int n_isects,
    const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ trow, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const int32_t *__restrict__ trow_s, float *__restrict__ prow, int prow_pitch,
    uint32_t *__restrict__ row_key, int32_t *__restrict__ row_idx, int rows_cap)
{
    constexpr int NBR = 4, CW = 128, C4 = 32;
    __shared__ __attribute__((aligned(16))) float stage[4][32][CW];
    __shared__ __attribute__((aligned(4))) uint8_t pos[2][CMAX][4];
    __shared__ int cand[2][4];
    __shared__ __attribute__((aligned(16))) float zrow[CW];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    GAGS_PH_DECL;
    if (threadIdx.x < CW) zrow[threadIdx.x] = 0.f;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int tile = gags_tile_of_order(logical / n_slices, tile_w, n_tiles / tile_w);
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int R0 = trow[start], R1 = trow[end];
    if (R1 == R0) return;
    const int blk = wave;
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    const int sb = gags_slot_base(start, end, tile, blk);
    const int ch0 = ch_base + (logical % n_slices) * CW;
    BlockGeom64 g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;  // p: channel group (channels ch0 + 4p + j) / slot; k: which 8 of a K-step's 16 pixels

    // cotangent slab as B operands: K element e = 16 s + 8 k + i of the weight rows' order = pixel e >> 1 of the 8x4 half
    // e & 1; Bh / Bl[s][j] = the 8 elements of K-step s for channel ch0 + 4 p + j, head and tail, scaled by cs[j]
    f16x8 Bh[4][NBR], Bl[4][NBR];
    float inv[NBR];
    {
        float raw[4][8][NBR];
        float mx[NBR] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 16 * s4 + 8 * k + i;
                const int pp = e >> 1, hh = e & 1;
                const int qj = g.bx0 + (pp & 7), qi = g.by0 + 4 * hh + (pp >> 3);
                const bool ok = (qi < height) && (qj < width) && cnt > 0;
                const float4 v = *reinterpret_cast<const float4 *>(
                    v_render_colors + ((size_t)min(qi, height - 1) * width + min(qj, width - 1)) * d + ch0 + NBR * p);
                raw[s4][i][0] = ok ? v.x : 0.f; raw[s4][i][1] = ok ? v.y : 0.f;
                raw[s4][i][2] = ok ? v.z : 0.f; raw[s4][i][3] = ok ? v.w : 0.f;
#pragma unroll
                for (int j = 0; j < NBR; ++j) mx[j] = fmaxf(mx[j], fabsf(raw[s4][i][j]));
            }
#pragma unroll
        for (int j = 0; j < NBR; ++j) {
            mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32));  // the column's other 32 pixels live in the other half-wave
            // largest magnitude -> [2^14, 2^15); an all-zero (or non-finite) column keeps scale 1
            const float cs = (mx[j] > 0.f && mx[j] < 3.0e38f) ? ldexpf(1.0f, min(14 - ilogbf(mx[j]), 126)) : 1.0f;  // (clamped: no inf scale for a column below 2^-112)
            inv[j] = 1.0f / cs;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float col[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) col[i] = raw[s4][i][j];
                split8(col, cs, Bh[s4][j], Bl[s4][j]);
            }
        }
    }

    GAGS_PH(0);  // prologue: metadata, cotangent slab loaded, scaled and split
    int pb = 0;
    int r0 = R0;
    int tr = 0x7fffffff, gid = 0;
    if (lane <= 32 && lane < cnt) {
        tr = trow_s[sb + lane];
        gid = gid_s[sb + lane];
    }
    // weight tile, raw: lane (slot p, k) holds elements 16 s + 8 k + i of its slot's row
    float A[32];
    auto load_A = [&](int first) {
        const float4 *src = reinterpret_cast<const float4 *>(wt + (size_t)(sb + first + p) * 64 + k * 8);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 u = src[4 * s4], v = src[4 * s4 + 1];
            A[8 * s4] = u.x; A[8 * s4 + 1] = u.y; A[8 * s4 + 2] = u.z; A[8 * s4 + 3] = u.w;
            A[8 * s4 + 4] = v.x; A[8 * s4 + 5] = v.y; A[8 * s4 + 6] = v.z; A[8 * s4 + 7] = v.w;
        }
    };
    if (cnt > 0) load_A(0);
    for (int it = 0; r0 < R1; ++it) {
        const int par = it & 1;
        if (threadIdx.x < CMAX) reinterpret_cast<uint32_t *>(&pos[par][0][0])[threadIdx.x] = 0xffffffffu;
        const int tr32 = __builtin_amdgcn_readlane(tr, 32);
        if (lane == 0) cand[par][wave] = tr32;
        gags_lds_barrier();  // LDS traffic only: __syncthreads() would also wait for the loads in flight (next chunk's weight tile) and the row stores
        GAGS_PH(1);  // chunk bookkeeping + first barrier
        const int r1 = min(min(min(cand[par][0], cand[par][1]), min(cand[par][2], cand[par][3])), min(r0 + CMAX, R1));
        const bool mine = lane < 32 && tr < r1;
        const int run = __popcll(__ballot(mine));
        const int tr_c = tr, gid_c = gid;
        const int pbn = pb + run;
        if (run > 0) {
            if (mine) pos[par][tr_c - r0][wave] = (uint8_t)lane;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            GAGS_PH(2);  // wait for the weight tile (and everything else in flight: the previous chunk's row stores)
            // power-of-two scale of this lane's slot row: its largest weight (weights are >= 0) -> [2^14, 2^15); exponent
            // arithmetic on the bits.  A row of zeros (pad slot, rows past the run: never stored) keeps scale 1.
            float wmx = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) wmx = fmaxf(wmx, A[i]);
            wmx = fmaxf(wmx, __shfl_xor(wmx, 32));
            const int ebits = (int)((__float_as_uint(wmx) >> 23) & 0xffu);
            const bool sane = ebits >= 15 && ebits <= 200;  // alpha*T lies in (4e-7, 1]; anything else (0, garbage past the block): 1
            const float rs = sane ? __uint_as_float((unsigned)(268 - ebits) << 23) : 1.0f;       // 2^(14 - exponent)
            const unsigned rinv = sane ? ((unsigned)(ebits - 14) << 23) : 0x3f800000u;           // its inverse, as bits
            f32x16 acc[NBR];
#pragma unroll
            for (int j = 0; j < NBR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float a8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a8[i] = A[8 * s4 + i];
                f16x8 a0, a1, a2;
                split8x3(a8, rs, a0, a1, a2);
#pragma unroll
                for (int j = 0; j < NBR; ++j) {  // smallest terms first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, Bh[s4][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, Bl[s4][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, Bh[s4][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, Bl[s4][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, Bh[s4][j], acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // one K-step's terms at a time (hoisted together they spill)
            }
            __builtin_amdgcn_sched_barrier(0);
            GAGS_PH(3);  // row scale, split, 80 MFMAs
            if (mine && ch0 == 0 && tr_c < rows_cap) {
                row_key[tr_c] = (uint32_t)gid_c;
                row_idx[tr_c] = tr_c;
            }
            tr = 0x7fffffff;
            if (lane <= 32 && pbn + lane < cnt) {
                tr = trow_s[sb + pbn + lane];
                gid = gid_s[sb + pbn + lane];
            }
            if (pbn < cnt) load_A(pbn);
            __builtin_amdgcn_sched_barrier(0);
            GAGS_PH(4);  // keys, next chunk's loads issued
            // this lane's four column unscales as two packed pairs: the row's inverse scale multiplies them with two
            // v_pk_mul_f32, the accumulators with two more
            typedef float pk2 __attribute__((ext_vector_type(2)));
            const pk2 inv01 = {inv[0], inv[1]}, inv23 = {inv[2], inv[3]};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // accumulator row r of this lane = slot (r & 3) + 8 (r >> 2) + 4 k: that row's inverse scale sits in lane `slot`
                const int s0 = (r & 3) + 8 * (r >> 2);
                const unsigned i0 = __builtin_amdgcn_readlane(rinv, s0), i1 = __builtin_amdgcn_readlane(rinv, s0 + 4);
                const float ri = __uint_as_float(k ? i1 : i0);
                const int slot = s0 + 4 * k;
                if (slot < run) {
                    const pk2 rr = {ri, ri};
                    const pk2 a01 = {acc[0][r], acc[1][r]}, a23 = {acc[2][r], acc[3][r]};
                    const pk2 o01 = a01 * (inv01 * rr), o23 = a23 * (inv23 * rr);
                    *reinterpret_cast<float4 *>(&stage[wave][slot][NBR * p]) = make_float4(o01[0], o01[1], o23[0], o23[1]);
                }
            }
        }
        GAGS_PH(5);  // unscale + park in LDS
        gags_lds_barrier();  // LDS traffic only: __syncthreads() would also wait for the loads in flight (next chunk's weight tile) and the row stores
        GAGS_PH(6);  // second barrier
        const int items = (r1 - r0) * C4;
        int gt = threadIdx.x;
        asm volatile("" : "+v"(gt));
#pragma unroll
        for (int trip = 0; trip < CMAX * C4 / 256; ++trip) {
            const int item = gt + 256 * trip;
            if (item < items) {
                const int row = item / C4, c4 = item - row * C4;
                float4 v[4];  // (a block that does not hold the row reads the row of zeros: see raster_bwd_rows)
                const uint32_t q4 = *reinterpret_cast<const uint32_t *>(&pos[par][row][0]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int q = (q4 >> (8 * b)) & 0xff;
                    v[b] = *(q != 0xff ? reinterpret_cast<const float4 *>(&stage[b][q][4 * c4]) : reinterpret_cast<const float4 *>(&zrow[4 * c4]));
                }
                float4 sum;
                sum.x = ((v[0].x + v[1].x) + v[2].x) + v[3].x; sum.y = ((v[0].y + v[1].y) + v[2].y) + v[3].y;
                sum.z = ((v[0].z + v[1].z) + v[2].z) + v[3].z; sum.w = ((v[0].w + v[1].w) + v[2].w) + v[3].w;
                if (r0 + row < rows_cap) *reinterpret_cast<float4 *>(prow + (size_t)(r0 + row) * prow_pitch + ch0 + 4 * c4) = sum;
            }
            if (trip & 1) __builtin_amdgcn_sched_barrier(0);
        }
        GAGS_PH(7);  // merge: LDS reads, adds, row stores
        pb = pbn;
        r0 = r1;
    }
    GAGS_PH_WRITE();
}

// capacity-sized row buffers: keys [total, cap) become sentinels (key n_gauss: past every Gaussian), so that the sort and
// the segment offsets can run over the capacity without the host knowing the row count
#include "raster_bwd_rows_cw.h"  // staged rows, channel waves (round 5; the default)

__global__ __launch_bounds__(256) void row_tail_kernel(int64_t cap, const int32_t *__restrict__ total, int n_gauss,
                                                       uint32_t *__restrict__ row_key, int32_t *__restrict__ row_idx)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap || i < (int64_t)total[0]) return;
    row_key[i] = (uint32_t)n_gauss;
    row_idx[i] = 0;
}

// trow_s[slot] = tile row of the slot's intersection (0x7fffffff for the pad slot of an odd count): one coalesced
// stream per block for the rows kernel instead of a dependent sidx -> trow gather.  One wave per (tile, block).
__global__ __launch_bounds__(64) void slot_rows_kernel(int n_tiles, int n_isects, const int32_t *__restrict__ offsets,
                                                       const int32_t *__restrict__ blk_rows,
                                                       const int32_t *__restrict__ sidx_s,
                                                       const int32_t *__restrict__ trow, int32_t *__restrict__ trow_s)
{
    const int tile = blockIdx.x >> 2, blk = blockIdx.x & 3;
    const int cnt = blk_rows[blockIdx.x];
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    for (int j = threadIdx.x; j < cnt; j += 64) {
        const int sx = sidx_s[sb + j];
        trow_s[sb + j] = sx >= 0 ? trow[sx] : 0x7fffffff;
    }
}

// seg[g] = first sorted position whose key is >= g, for g in [0, n_keys]
__global__ __launch_bounds__(256) void seg_offsets_kernel(int n, const uint32_t *__restrict__ sorted_keys, int n_keys,
                                                          int32_t *__restrict__ seg)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cur = min((int)sorted_keys[i], n_keys);
    if (i == 0) {
        for (int g = 0; g <= cur; ++g) seg[g] = 0;
    } else {
        const int prev = min((int)sorted_keys[i - 1], n_keys);
        for (int g = prev + 1; g <= cur; ++g) seg[g] = i;
    }
    if (i == n - 1)
        for (int g = cur + 1; g <= n_keys; ++g) seg[g] = n;
}

__global__ void seg_fill_kernel(int n_keys, int32_t *__restrict__ seg)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g <= n_keys) seg[g] = 0;
}

// v_colors[g, :] = sum over the Gaussian's rows, in sorted (= deterministic) order; VW channels per lane (4: one float4;
// 1: the 1-3 channels a width that is no multiple of 4 leaves over, e.g. the 513th).
// HALF: the sum (formed in fp32) is stored as fp16 -- the gradient of an fp16 feature table in the table's own dtype,
// instead of an fp32 tensor plus a cast pass over it (N x D x 6 bytes of traffic at C5).
constexpr int REDUCE_ITER = 8;
template <bool HALF, int VW>
__global__ __launch_bounds__(256) void reduce_rows_kernel(int n_gauss, int d, int ch_begin, int ch_count,
                                                          const int32_t *__restrict__ seg,
                                                          const int32_t *__restrict__ sorted_rows,
                                                          const float *__restrict__ prow, int prow_pitch,
                                                          void *__restrict__ v_colors_, int sparse,
                                                          const int32_t *__restrict__ wire_pos, float *__restrict__ wire,
                                                          const uint8_t *__restrict__ keep_prev, uint8_t *__restrict__ keep_cur,
                                                          int tail)
{
    // tail (VW == 4 only): the 1-3 channels an odd width leaves behind its float4 columns (the 513th of BASELINE.json configs[4])
    // ride along -- lane t < tail of a Gaussian's group also sums channel ch_begin + ch_count + t -- instead of a second launch
    // with one lane per Gaussian walking the same row lists again (0.22 ms at C5).
    // keep (round 6): v_colors is a PERSISTENT buffer of the caller's whose rows are zero except those the previous backward
    // wrote (keep_prev[g] != 0).  This launch writes the rows that have partial rows now, re-zeroes the rows that had some
    // last time and have none now, leaves every other row alone -- 73 % of the Gaussians blend nothing at C3: 2.2 GB of zero
    // rows per step are not written -- and records keep_cur[g] for the next step (two arrays, swapped by the caller: the lanes
    // of a Gaussian span two waves, a flag cleared in place could be read after it was cleared).
    // wire (by-view multi-GPU step, gags_amd/dist.py): the rows the ranks exchange -- wire_pos[g] >= 0: row wire_pos[g] of the
    // dense [rows, ch_count] fp32 block -- leave from here, next to the gradient itself, instead of being re-read by a pack
    // kernel (a union row this view did not touch gets its zeros here as well: every row of the block is written)
    // (REDUCE_ITER groups of Gaussians per workgroup, one after the other: three of four Gaussians have no rows at C3 -- as one
    // workgroup per group those were 550 k empty workgroups for the dispatcher)
    const int lpg = ch_count / VW;  // lanes per Gaussian (channels ch_begin .. ch_begin + ch_count - 1 of its row)
    const int gpb = 256 / lpg;
    const int gl = threadIdx.x / lpg;
    const int cl = ch_begin + (threadIdx.x % lpg) * VW;
    if (gl >= gpb) return;
    for (int it = 0; it < REDUCE_ITER; ++it) {
    const int g = (blockIdx.x * REDUCE_ITER + it) * gpb + gl;
    if (g >= n_gauss) return;
        const int b = seg[g], e = seg[g + 1];
        // sparse: the caller zero-filled v_colors (on a second stream, under the rows kernel): a Gaussian without rows -- 73 % of
        // them at C3 -- costs nothing here instead of a 4 D-byte row of zeros
        const bool has = b != e;
        bool write_grad = sparse ? has : true;
        const bool on_wire = wire && wire_pos[g] >= 0;
        if (keep_cur) {
            // (a row of the exchanged block will be written by the caller once the ranks' sum is known: it counts as written)
            if (threadIdx.x % lpg == 0) keep_cur[g] = (has || on_wire) ? 1 : 0;
            write_grad = has || keep_prev[g] != 0;
        }
        if (!write_grad && !on_wire) continue;
        const bool skip_grad = !write_grad;  // (only its wire row is due)
        if constexpr (VW == 1) {
            float acc = 0.f;
            for (int i = b; i < e; ++i) acc += prow[(size_t)sorted_rows[i] * prow_pitch + cl];
            if (!skip_grad) {
                if constexpr (HALF) reinterpret_cast<__half *>(v_colors_)[(size_t)g * d + cl] = __float2half_rn(acc);
                else reinterpret_cast<float *>(v_colors_)[(size_t)g * d + cl] = acc;
            }
            if (wire) {
                const int q = wire_pos[g];
                if (q >= 0) wire[(size_t)q * ch_count + (cl - ch_begin)] = acc;
            }
        } else {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int tl = threadIdx.x % lpg;
            const bool has_tail = tl < tail;
            const int ct = ch_begin + ch_count + (has_tail ? tl : 0);  // this lane's tail channel (same order of additions as the body)
            float acc_t = 0.f;
            // Batches of eight rows, every load of a batch requested before the first is added: the row numbers (clamped to the
            // Gaussian's last row: valid addresses, their values unused), then the rows.  A Gaussian has 4.7 rows on average at
            // C3 -- two round trips instead of one per group of four plus one per leftover row.  Same order of additions.
            for (int i = b; i < e; i += 8) {
                int r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = sorted_rows[min(i + j, e - 1)];
                float4 v[8];
                float vt[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = *reinterpret_cast<const float4 *>(prow + (size_t)r[j] * prow_pitch + cl);
                    vt[j] = has_tail ? prow[(size_t)r[j] * prow_pitch + ct] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (i + j < e) {
                        acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
                        acc_t += vt[j];
                    }
                }
            }
            if (!skip_grad && has_tail) {
                if constexpr (HALF) reinterpret_cast<__half *>(v_colors_)[(size_t)g * d + ct] = __float2half_rn(acc_t);
                else reinterpret_cast<float *>(v_colors_)[(size_t)g * d + ct] = acc_t;
            }
            if (!skip_grad) {
                if constexpr (HALF) {
                    const __half2 lo = __floats2half2_rn(acc.x, acc.y), hi = __floats2half2_rn(acc.z, acc.w);
                    uint2 w;
                    w.x = *reinterpret_cast<const unsigned *>(&lo);
                    w.y = *reinterpret_cast<const unsigned *>(&hi);
                    *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(v_colors_) + (size_t)g * d + cl) = w;
                } else {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(nt_f4{acc.x, acc.y, acc.z, acc.w},
                                                reinterpret_cast<nt_f4 *>(reinterpret_cast<float *>(v_colors_) + (size_t)g * d + cl));
                }
            }
            if (wire) {
                const int q = wire_pos[g];
                if (q >= 0) *reinterpret_cast<float4 *>(wire + (size_t)q * ch_count + (cl - ch_begin)) = acc;
            }
        }
    }
}

// ---- fallback: single kernel, float atomics ------------------------------------------------------------
__global__ __launch_bounds__(64, 2) void raster_bwd_atomic(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const GRec *__restrict__ packed,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ v_render_colors, float *__restrict__ v_colors, int by_gauss)
{
    __shared__ __attribute__((aligned(16))) HRec ring[RING];
    __shared__ __attribute__((aligned(16))) float Wt[32 * WT_STRIDE];
    __shared__ int32_t slot_id[32];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8 * n_slices);
    const int slice = logical % n_slices, rest = logical / n_slices;
    const int blk = rest & 7;
    const int tile = gags_tile_of_order(rest >> 3, tile_w, n_tiles / tile_w);
    const int ch0 = slice * CSB;
    const int lane = threadIdx.x;
    BlockGeom g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;

    float V[16][NBB];
    load_slab(V, v_render_colors, g, width, height, d, ch0);

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !g.inside;
    HitStream hs;
    hs.by_gauss = by_gauss != 0;
    hs.init(ring, packed, flatten_ids, start, end, lane, g);

    int nh = 0;
    const int wpos = (p & 1) * 16 + (p >> 1);
    auto flush = [&](int count) {
        float A[16];
        const float4 *rowp = reinterpret_cast<const float4 *>(Wt + p * WT_STRIDE + k * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = rowp[t];
            A[4 * t] = v.x; A[4 * t + 1] = v.y; A[4 * t + 2] = v.z; A[4 * t + 3] = v.w;
        }
        f32x16 acc[NBB];
        tile_mfma(acc, A, V);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
            const int gid = slot_id[slot];
            if (slot < count && gid >= 0) {
                float *dst = v_colors + (size_t)gid * d + ch0 + p;
#pragma unroll
                for (int j = 0; j < NBB; ++j) atomic_add_f32(dst + 32 * j, acc[j][r]);
            }
        }
    };

    hs.refill(6);
    if (!__all(st.done) && hs.rd < hs.nq) {
        bool v_n;
        HRec h_n = hs.at(hs.rd, k, v_n);
        float a_n = eval_alpha(h_n, g.px, g.py, v_n);
        int gid_n = v_n ? h_n.gid : -1;
        bool go = true;
        while (go) {
            const float a_c = a_n;
            const int gid_c = gid_n;
            hs.rd += 2;
            if ((hs.nq - hs.rd) < 6 && hs.pending) hs.refill(6);
            const bool more = hs.rd < hs.nq;
            h_n = hs.at(hs.rd, k, v_n);
            a_n = eval_alpha(h_n, g.px, g.py, v_n);
            gid_n = v_n ? h_n.gid : -1;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
            bool blended;
            const float wgt = step_pair(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
            if (__any(wgt != 0.f)) {
                Wt[(nh + k) * WT_STRIDE + wpos] = wgt;
                if (p == 0) slot_id[nh + k] = gid_c;
                nh += 2;
                if (nh == 32) { flush(32); nh = 0; }
            }
            go = more && !__all(st.done);
        }
    }
    if (nh > 0) flush(nh);
}

struct StagedLayout {
    int64_t key, idx, key_s, idx_s, seg, sort, prow, total;
};
inline int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }
inline StagedLayout staged_layout(int64_t rows, int n_gauss, int d)
{
    StagedLayout L;
    int64_t o = 0;
    L.key = o; o += al256(rows * 4);
    L.idx = o; o += al256(rows * 4);
    L.key_s = o; o += al256(rows * 4);
    L.idx_s = o; o += al256(rows * 4);
    L.seg = o; o += al256(((int64_t)n_gauss + 2) * 4);
    L.sort = o; o += al256(gags_sort_u32_scratch_bytes(rows));
    L.prow = o; o += al256(rows * (int64_t)d * 4);
    L.total = o;
    return L;
}

}  // namespace

int64_t gags_bwd_staged_scratch_bytes_impl(int64_t rows, int n_gauss, int d)
{
    return staged_layout(rows > 0 ? rows : 1, n_gauss, d).total;
}

// mask[g] = 1 for every Gaussian that blended into at least one pixel of the view (the rows of the feature gradient
// that can be non-zero): hit flag of the forward per sorted intersection -> its Gaussian.  Benign race (all write 1).
__global__ __launch_bounds__(256) void blended_mask_kernel(int n_isects, const int32_t *__restrict__ hit,
                                                           const int32_t *__restrict__ flatten_ids, unsigned char *__restrict__ mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_isects && hit[i]) mask[flatten_ids[i]] = 1;
}

int gags_blended_mask_launch(int n_isects, const int32_t *hit, const int32_t *flatten_ids, unsigned char *mask, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    hipLaunchKernelGGL(blended_mask_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, hit, flatten_ids, mask);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

int gags_bwd_slot_rows_launch(int width, int height, int n_isects, const int32_t *offsets, const int32_t *blk_rows,
                              const int32_t *sidx_s, const int32_t *trow, int32_t *trow_s, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    hipLaunchKernelGGL(slot_rows_kernel, dim3(tile_w * tile_h * GAGS_BLOCKS_PER_TILE), dim3(64), 0, st, tile_w * tile_h,
                       n_isects, offsets, blk_rows, sidx_s, trow, trow_s);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// GAGS_BWD_ROWSCALE=1 (experiments; read once): the rows kernel's weight scale per (row, block) instead of the view-wide 2^15
static bool rows_scale_per_block()
{
    static const bool v = [] { const char *e = getenv("GAGS_BWD_ROWSCALE"); return e && e[0] == '1'; }();
    return v;
}

// 1 = width not eligible
int gags_raster_bwd_staged_launch(int d, int width, int height, int n_gauss, const int32_t *offsets, int n_isects,
                                  const float *v_out, const int32_t *blk_rows, const int32_t *trow, int64_t rows,
                                  const float *wt, const int32_t *gid_s, const int32_t *trow_s, void *scratch,
                                  int64_t scratch_bytes, float *v_colors, int stage_flags, int ch_begin, int ch_count,
                                  const int32_t *rows_dev, const int32_t *wire_pos, float *wire, const uint8_t *keep_prev,
                                  uint8_t *keep_cur, hipStream_t st)
{
    // stage: 0 = everything; 1 = rows, 2 = sort + segment offsets, 3 = reduce (per-kernel timing)
    GAGS_CLEAR_ERR();
    const int stage = stage_flags & 15;
    if (!gags_mfma_width(d) || d > 1024) return 1;
    const bool sA = stage == 0 || stage == 1, sS = stage == 0 || stage == 2, sR = stage == 0 || stage == 3;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    // channel range of this call (a by-view step exchanges the gradient range by range while the next range is computed,
    // gags_amd/dist.py); the default is everything.  Ranges start on a multiple of 32 and end on one or at d.
    if (ch_count <= 0 || ch_begin < 0 || ch_begin + ch_count > d || ch_begin % 32 != 0 ||
        (ch_count % 32 != 0 && ch_begin + ch_count != d))
        return GAGS_EINVAL;
    // stage bit 256: the scratch holds partial rows of THIS call's channel range only ([rows, ch_count rounded up to 4]
    // instead of [rows, d]; sized with gags_bwd_staged_scratch_bytes(rows, n, that width)): a wide gradient is then produced
    // range by range through a scratch a quarter (an eighth ...) the size -- what lets heavy views fit (C5H: 80 M rows)
    const bool narrow = (stage_flags & 256) != 0;
    const int pp = narrow ? ((ch_count + 3) & ~3) : d;
    const StagedLayout L = staged_layout(rows > 0 ? rows : 1, n_gauss, pp);
    if (scratch_bytes < L.total) return GAGS_ESCRATCH;
    char *sb = (char *)scratch;
    uint32_t *key = (uint32_t *)(sb + L.key), *key_s = (uint32_t *)(sb + L.key_s);
    int32_t *idx = (int32_t *)(sb + L.idx), *idx_s = (int32_t *)(sb + L.idx_s), *seg = (int32_t *)(sb + L.seg);
    float *prow = (float *)(sb + L.prow) - (narrow ? ch_begin : 0);  // (indexed by absolute channel)
    if (rows > 0) {
        if (sA) {
            // 128-channel slices, then 64, then 32-channel slices (the last one ragged when the range ends at an odd d)
#define GAGS_ROWS_LAUNCH(KERNEL, CH0, NSL)                                                                           \
    hipLaunchKernelGGL(KERNEL, dim3(n_tiles * (NSL)), dim3(256), 0, st, d, width, height, tile_w, n_tiles, (CH0), (NSL), \
                       v_out, offsets, n_isects, blk_rows, trow, wt, gid_s, trow_s, prow, pp, key, idx, (int)rows)
            int c = ch_begin;
            const int ce = ch_begin + ch_count;
            if (ce - c >= 128) {
                const int nsl = (ce - c) / 128;
                if (stage_flags & 32) GAGS_ROWS_LAUNCH(raster_bwd_rows<4>, c, nsl);  // GAGS_BWD_F32MFMA: the fp32 matrix instructions
                else if (stage_flags & 512) GAGS_ROWS_LAUNCH(raster_bwd_rows_f16, c, nsl);  // round 4's shape: a wave per pixel block, rows merged in LDS
                else if (stage_flags & 1024) GAGS_ROWS_LAUNCH((raster_bwd_rows_cw<3, 5>), c, nsl);  // weights as three terms (exact), five product terms
                else if (rows_scale_per_block()) GAGS_ROWS_LAUNCH((raster_bwd_rows_cw<2, 3>), c, nsl);  // (GAGS_BWD_ROWSCALE=1: round 5's scale per (row, block))
                else GAGS_ROWS_LAUNCH((raster_bwd_rows_cw<2, 3, true>), c, nsl);  // default: 16-bit matrix cores, a wave per 32 channels, three product terms, one weight scale
                c += 128 * nsl;
            }
            if (ce - c >= 64) {
                GAGS_ROWS_LAUNCH(raster_bwd_rows<2>, c, 1);
                c += 64;
            }
            if (ce - c > 0) GAGS_ROWS_LAUNCH(raster_bwd_rows<1>, c, (ce - c + 31) / 32);
#undef GAGS_ROWS_LAUNCH
        }
        if (sS) {
            // rows is a CAPACITY when rows_dev is given (the true count lives on the device): sentinel keys past it
            if (rows_dev)
                hipLaunchKernelGGL(row_tail_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rows, rows_dev, n_gauss, key,
                                   idx);
            int nbits = 1;
            while ((1ll << nbits) <= n_gauss) ++nbits;  // keys in [0, n_gauss]
            const int rc = gags_sort_pairs_u32(rows, nbits, key, idx, key_s, idx_s, sb + L.sort, L.prow - L.sort, st);
            if (rc != GAGS_OK) return rc;
            hipLaunchKernelGGL(seg_offsets_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (int)rows,
                               key_s, n_gauss, seg);
        }
    } else if (sS) {
        hipLaunchKernelGGL(seg_fill_kernel, dim3((n_gauss + 1 + 255) / 256), dim3(256), 0, st, n_gauss, seg);
    }
    if (sR) {
        const bool half = (stage_flags & 64) != 0;  // v_colors is an fp16 tensor
        const int sparse = (stage_flags & 128) ? 1 : 0;  // v_colors arrives zero-filled: rows of Gaussians that blended nothing are skipped
        const int c4 = ch_count & ~3, c1 = ch_count & 3;  // float4 lanes + the 1-3 channels an odd width leaves over
        if (wire && (c1 != 0 || !wire_pos)) return GAGS_EINVAL;  // the wire block is [rows, ch_count], ch_count % 4 == 0
        const bool ride = c4 >= 16 && c1 > 0;  // the 1-3 leftover channels ride along with the float4 columns' launch
        if (c4 > 0) {
            const int gpb = 256 / (c4 >> 2);
            const dim3 grid((n_gauss + gpb * REDUCE_ITER - 1) / (gpb * REDUCE_ITER));
            const int tail = ride ? c1 : 0;
            if (half) hipLaunchKernelGGL((reduce_rows_kernel<true, 4>), grid, dim3(256), 0, st, n_gauss, d, ch_begin, c4, seg, idx_s, prow, pp, (void *)v_colors, sparse, wire_pos, wire, keep_prev, keep_cur, tail);
            else hipLaunchKernelGGL((reduce_rows_kernel<false, 4>), grid, dim3(256), 0, st, n_gauss, d, ch_begin, c4, seg, idx_s, prow, pp, (void *)v_colors, sparse, wire_pos, wire, keep_prev, keep_cur, tail);
        }
        if (c1 > 0 && !ride) {
            const int gpb = 256 / c1;
            const dim3 grid((n_gauss + gpb * REDUCE_ITER - 1) / (gpb * REDUCE_ITER));
            if (half) hipLaunchKernelGGL((reduce_rows_kernel<true, 1>), grid, dim3(256), 0, st, n_gauss, d, ch_begin + c4, c1, seg, idx_s, prow, pp, (void *)v_colors, sparse, wire_pos, wire, keep_prev, keep_cur, 0);
            else hipLaunchKernelGGL((reduce_rows_kernel<false, 1>), grid, dim3(256), 0, st, n_gauss, d, ch_begin + c4, c1, seg, idx_s, prow, pp, (void *)v_colors, sparse, wire_pos, wire, keep_prev, keep_cur, 0);
        }
    }
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// =========================================================================================================================
// Geometry gradients at wide D.  The only D-proportional term of d loss / d alpha,
//     v_alpha(g, px) = T_g <c_g, v_px>  -  ra_g SUM_{g' behind g} f_g' <c_g', v_px>  +  T_f ra_g (v_a - <bg, v_px>),
// needs the dot products S[g, px] = <c_g, v_px> and nothing else: the "buffer" term is a running sum of f S over the
// Gaussians behind (linearity).  So:
//   1. raster_bwd_sdot: S[slot][64 px] for every slot of every 8x8 block on the fp32 matrix cores -- the contraction runs
//      over the CHANNELS, 256 per pass (wider tables: accumulating passes); the block's cotangent slab [64 px][256 ch]
//      sits in LDS (66 KB: two workgroups per CU, one loads its slab while the other multiplies), the four waves take
//      the block's 32-slot tiles in turn, the slots' feature rows are streamed from L2 / HBM eight float4 ahead;
//   2. raster_bwd_geom: one wave per block walks its slots back to front, one pixel per lane: alpha again from the packed
//      record, T = f / alpha from the forward's weight, v_alpha from S, the six geometry partials summed over the wave
//      and stored as ONE 32-byte row per slot (no atomics);
//   3. the rows are sorted by Gaussian and summed with the kernels of the colours backward (d = 8).
constexpr int SD_PAD = 4;        // floats of padding per slab row: b128 reads of 32 rows then cover all 64 banks
constexpr int SD_MAXCH = 256;    // channels per pass (LDS); wider tables take several accumulating passes

template <int DCH>  // channels of this pass, compile-time for the full 256 (everything unrolled), 0 = use `dch`
__global__ __launch_bounds__(256, 2) void raster_bwd_sdot(int d, int ch0, int dch_rt, int width, int height, int tile_w, int n_tiles,
                                                       int n_gauss, const float *__restrict__ v_out,
                                                       const float *__restrict__ colors, const float *__restrict__ backgrounds,
                                                       const int32_t *__restrict__ offsets, int n_isects,
                                                       const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ gid_s,
                                                       float *__restrict__ S, float *__restrict__ bgdot, int accumulate,
                                                       const int32_t *__restrict__ row_base)
{
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [64 pixels in wt order e = 2 p + h][dch + SD_PAD]
    __shared__ float bred[4][64];
    __shared__ __attribute__((aligned(16))) float bgs[SD_MAXCH];  // the pass's slice of the background
    const int dch = DCH ? DCH : dch_rt;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE);
    const int blk = logical & 3;
    const int tile = gags_tile_of_order(logical >> 2, tile_w, n_tiles / tile_w);
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    if (cnt == 0) return;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    // S rows: compact numbering (row_base = exclusive prefix sum of blk_rows: S is n_rows x 256 B) or the sparse slot index
    const int srb = row_base ? row_base[tile * GAGS_BLOCKS_PER_TILE + blk] : sb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = dch + SD_PAD;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 8;
    {
        // slab load, eight float4 per thread in flight (a load-then-store loop would wait for every single one)
        const int q4 = dch >> 2;  // float4 per pixel row
        const int total = 64 * q4;
        for (int i0 = tid; i0 < total; i0 += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * 256, total - 1);
                const int e = i / q4, c4 = i - e * q4;
                const int pp = e >> 1, h = e & 1;
                const int pj = bx0 + (pp & 7), pi = by0 + (pp >> 3) + 4 * h;
                const bool in = pi < height && pj < width;
                v[u] = *reinterpret_cast<const float4 *>(v_out + ((size_t)min(pi, height - 1) * width + min(pj, width - 1)) * d + ch0 + 4 * c4);
                if (!in) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < total) {
                    const int e = i / q4, c4 = i - e * q4;
                    *reinterpret_cast<float4 *>(slab + e * pitch + 4 * c4) = v[u];
                }
            }
        }
    }
    if (backgrounds && tid < dch) bgs[tid] = backgrounds[ch0 + tid];
    __syncthreads();
    if (backgrounds) {  // <bg, v_px> per pixel of the block, for the background term of v_alpha (background slice from LDS)
        const int e = tid & 63, part = tid >> 6;
        float acc = 0.f;
        for (int c = 4 * part; c < dch; c += 16) {
            const float4 x = *reinterpret_cast<const float4 *>(slab + e * pitch + c), b = *reinterpret_cast<const float4 *>(bgs + c);
            acc = fmaf(b.x, x.x, acc); acc = fmaf(b.y, x.y, acc); acc = fmaf(b.z, x.z, acc); acc = fmaf(b.w, x.w, acc);
        }
        bred[part][e] = acc;
        __syncthreads();
        if (tid < 64) {
            const int pp = tid >> 1, h = tid & 1;
            const int pj = bx0 + (pp & 7), pi = by0 + (pp >> 3) + 4 * h;
            if (pi < height && pj < width) {
                const float t = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
                float *dst = bgdot + (size_t)pi * width + pj;
                *dst = accumulate ? *dst + t : t;
            }
        }
    }
    const int m = lane & 31, kh = lane >> 5;
    const int nj = dch >> 3;  // steps of 8 channels: lane (m, kh) takes channels 8 j + 4 kh .. + 3 of slot m's row
    // work units = (32-slot tile, 32-pixel half of the block), dealt round-robin to the four waves: with whole tiles as
    // units a block of ~4.75 tiles kept the waves 59 % busy, with half-tiles 79 % (the second reader of a feature row
    // finds it in the CU's L1)
    const int n_units = 2 * ((cnt + 31) >> 5);
    for (int unit = wave; unit < n_units; unit += 4) {
        const int t0 = (unit >> 1) * 32, hh = unit & 1;
        const int slot = sb + min(t0 + m, cnt - 1);
        const int gid = min(gid_s[slot], n_gauss - 1);  // the pad slot's row is computed and never used (its weights are 0)
        const float *crow = colors + (size_t)gid * d + ch0 + 4 * kh;
        const float *b0 = slab + (32 * hh + m) * pitch + 4 * kh;
        f32x16 acc0;
        if (accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(t0 + (r & 3) + 8 * (r >> 2) + 4 * kh, cnt - 1);
                acc0[r] = S[(size_t)(srb + row) * 64 + 32 * hh + m];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        }
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4 *>(crow + 8 * min(u, nj - 1));
        auto step = [&](int j, int u) __attribute__((always_inline)) {
            const float4 av = a[u];
            a[u] = *reinterpret_cast<const float4 *>(crow + 8 * min(j + 8, nj - 1));
            const float4 x0 = *reinterpret_cast<const float4 *>(b0 + 8 * j);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, x0.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, x0.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, x0.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, x0.w, acc0, 0, 0, 0);
        };
        if constexpr (DCH != 0) {  // one basic block: no loop back-edge for the compiler to drain the prefetches at
#pragma unroll
            for (int j = 0; j < DCH / 8; ++j) {
                step(j, j & 7);
                __builtin_amdgcn_sched_barrier(0);  // (or every LDS read of the tile is hoisted to the top: 256 VGPRs + spills)
            }
        } else {
            for (int jb = 0; jb < nj; jb += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (jb + u < nj) step(jb + u, u);
            }
        }
        // accumulator: column = pixel element m of half hh, rows = slots (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = t0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < cnt) S[(size_t)(srb + row) * 64 + 32 * hh + m] = acc0[r];
        }
    }
}

// ---- the same dot pass on the fp16 matrix cores with split operands (round 3; the default) --------------------------------
// S = <c_g, v_px> is a plain contraction over the channels, so the scheme of raster_bwd_rows_f16 applies with the roles
// swapped: the FEATURE rows get one power-of-two scale per Gaussian row (largest magnitude -> [2^14, 2^15)), the COTANGENT
// slab one scale per 8x8 block and pass (the geometry rows sum S over the block's 64 pixels, so an error relative to the
// block's largest cotangent is an error relative to the row's largest term).  Since round 5 both operands are TWO fp16 terms
// (one fp32-level rounding each) and a product is a0 b0 + a0 b1 + a1 b0 (SF_TA = 2; rounds 3-4: feature rows as three terms,
// exact, five product terms: SF_TA = 3) -- against float64 autograd the geometry gradients did not move (v_opacities 2.3e-7,
// v_means2d 2.5e-7; five terms: 2.7e-7 / 2.4e-7), the pass went 1.37 -> 1.16 ms.  Matrix terms at the 16-bit rate
// instead of one at the fp32 rate.
// What bounds this pass is not the arithmetic but the GATHER of the feature rows: every lane of an A operand reads
// another Gaussian's row, and the vector L1 looks up one 128-byte line per cycle -- 32x32 tiles (two lanes per row, 32 B
// per row and instruction) delivered 16-32 B/clk/CU and ran SLOWER with the 16-bit cores than the fp32 kernel (3.0 vs
// 2.1 ms per pass at C3).  Hence v_mfma_f32_16x16x32_f16: FOUR lanes per row, 64 contiguous bytes of one row per
// instruction, 16 lines per wave load.
//   feat_split_kernel: once per pass, every BLENDED Gaussian's 256-channel slice -> SF_TA fp16 terms in operand order
//     ([32-channel step][term][32 halves]) + 1 / scale per row; the split costs ~13 VALU instructions per two values and
//     a row is read by ~16 units, so it is done once, not per unit;
//   raster_bwd_sdot_f16: slab -> registers -> block maximum -> two fp16 planes in LDS (67.6 KB: two workgroups per CU);
//     units of 16 slots x 64 pixels (four 16x16 accumulators), per 32-channel step SF_TA x 16 B from the table (four steps
//     ahead), 8 x 16 B from LDS, 12 MFMAs (20 with three terms), no VALU work in the loop; the accumulators leave scaled back by
//     (row scale x block scale) -- S holds plain dot products, as before.
constexpr int SF_TA = 2;    // fp16 terms of a feature value in the table (2: one fp32-level rounding; 3: exact, five product terms)
constexpr int SF_PADH = 8;  // halves of padding per slab row (one 2-way conflict per ds_read_b128 lane group; 16 is conflict-free and measured the same)
constexpr int SF_PF = 4;    // 32-channel steps of table rows in flight
constexpr int SF_STAGE = 1024;  // slots of a block whose ids and scales are staged in LDS (8 KB; mean block: ~140 slots)
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void feat_split_kernel(int n_gauss, int d, int ch0, int ks, const float *__restrict__ colors,
                                                         const unsigned char *__restrict__ mask, uint4 *__restrict__ table,
                                                         float *__restrict__ rinv)
{
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;  // half a wave per row, 8 channels per lane
    if (row >= n_gauss || !mask[row]) return;
    const int dch = min(32 * ks, d - ch0);
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (8 * l < dch) {
        const float4 *src = reinterpret_cast<const float4 *>(colors + (size_t)row * d + ch0 + 8 * l);
        const float4 u = src[0], v = src[1];
        x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
    }
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(x[i]));
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const int ebits = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    const bool sane = ebits >= 16 && ebits <= 240;  // zero rows, denormals, non-finite values: scale 1
    const float rs = sane ? __uint_as_float((unsigned)(268 - ebits) << 23) : 1.0f;  // 2^(14 - exponent)
    if (l < 4 * ks) {
        f16x8 t0, t1, t2;
        uint4 *dst = table + (size_t)row * (4 * SF_TA * ks) + 4 * SF_TA * (l >> 2) + (l & 3);  // step l / 4, k quarter l % 4; term t: + 4 t
        if constexpr (SF_TA == 3) {
            split8x3(x, rs, t0, t1, t2);
            dst[8] = *reinterpret_cast<const uint4 *>(&t2);
        } else {
            split8x2(x, rs, t0, t1);
        }
        dst[0] = *reinterpret_cast<const uint4 *>(&t0);
        dst[4] = *reinterpret_cast<const uint4 *>(&t1);
    }
    if (l == 0) rinv[row] = sane ? __uint_as_float((unsigned)(ebits - 14) << 23) : 1.0f;
}

template <int KS>  // 32-channel steps of this pass: 8 = the full 256 channels, everything unrolled; 0 = `ks_rt`
__global__ __launch_bounds__(256, 2) void raster_bwd_sdot_f16(int d, int ch0, int ks_rt, int width, int height, int tile_w, int n_tiles,
                                                           int n_gauss, const float *__restrict__ v_out,
                                                           const uint4 *__restrict__ table, const float *__restrict__ rinv,
                                                           const float *__restrict__ backgrounds,
                                                           const int32_t *__restrict__ offsets, int n_isects,
                                                           const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ gid_s,
                                                           float *__restrict__ S, float *__restrict__ bgdot,
                                                           const int32_t *__restrict__ row_base)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 planes[];  // hi [64 pixels e = 2 p + h][pitch], then lo
    __shared__ float bred[4][64];
    __shared__ float wmax[4];
    __shared__ __attribute__((aligned(16))) float bgs[SD_MAXCH];  // the pass's slice of the background
    __shared__ int gids[SF_STAGE];                                 // the block's first SF_STAGE slots: Gaussian id, 1 / row scale
    __shared__ float rinvs[SF_STAGE];
    const int ks = KS ? KS : ks_rt;
    const int dch = min(32 * ks, d - ch0);  // channels of the pass that exist (a multiple of 8); the rest of the last step is 0
    const int pitch = 32 * ks + SF_PADH;
    _Float16 *hi = planes, *lo = planes + 64 * pitch;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE);
    const int blk = logical & 3;
    const int tile = gags_tile_of_order(logical >> 2, tile_w, n_tiles / tile_w);
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    if (cnt == 0) return;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    const int srb = row_base ? row_base[tile * GAGS_BLOCKS_PER_TILE + blk] : sb;  // S rows: compact or sparse numbering
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 8;
    const int m = lane & 15, kq = lane >> 4;
    // work units = 16 slots x the block's 64 pixels, dealt round-robin to the four waves (a block of ~150 slots: ten units)
    const int n_units = (cnt + 15) >> 4;
    // The (unit, step) pairs of a wave form ONE stream: the table rows are requested SF_PF steps ahead ACROSS unit
    // boundaries.  NOTHING else is loaded from global memory inside that stream: loads return in order, so one HBM miss
    // (an id, a scale, an earlier pass's sum) in front of a row request stalls the MFMAs for its whole latency -- the ids
    // and scales of the block's slots are staged in LDS up front (requested before the slab, their latency passes under
    // the slab's), and every pass writes its own S / bgdot buffer (raster_bwd_geom adds them up; no read-modify-write).
    auto slot_gid = [&](int j) __attribute__((always_inline)) { return min(gid_s[sb + min(j, cnt - 1)], n_gauss - 1); };  // (pad slot: clamped, unused)
    const int n_stage = min(cnt, SF_STAGE);
    int sg[SF_STAGE / 256];
#pragma unroll
    for (int q = 0; q < SF_STAGE / 256; ++q) sg[q] = (q * 256 < n_stage) ? slot_gid(tid + q * 256) : 0;
    int gid = 0;
    // Row requests are made in QUAD order -- lane 4 r + q asks for quarter q (16 bytes) of row r's 64-byte step, four
    // consecutive lanes one contiguous 64 bytes -- and the registers are then permuted into the MFMA's operand order
    // (lane 16 q + r): the address coalescer works on neighbouring lanes, and in operand order (neighbours = different
    // Gaussians) a 16-byte-per-lane gather went through at one lane per cycle, which bounded the whole pass.
    const int lr = lane >> 2, lq = lane & 3;
    const int perm_src = 4 * (4 * m + kq);  // ds_bpermute address: operand lane (m, kq) takes from request lane 4 m + kq
    if (wave < n_units) gid = slot_gid(wave * 16 + lr);  // the wave's first unit: its rows are requested before the slab is waited for
    uint4 a[SF_PF][SF_TA];
    float ibs;  // 1 / block scale
    {
        // the whole slab in registers (<= 16 float4 per thread, all requested at once), its largest magnitude, then
        // scaled and split into the two planes
        const int q4 = 8 * ks;  // float4 per padded pixel row
        const int total = 64 * q4;
        float4 v[16];
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u * 256 < total) {  // (uniform)
                const int i = min(tid + u * 256, total - 1);
                const int e = i / q4, c4 = i - e * q4;
                const int pp = e >> 1, h = e & 1;
                const int pj = bx0 + (pp & 7), pi = by0 + (pp >> 3) + 4 * h;
                const bool in = pi < height && pj < width && 4 * c4 < dch && tid + u * 256 < total;
                v[u] = *reinterpret_cast<const float4 *>(v_out + ((size_t)min(pi, height - 1) * width + min(pj, width - 1)) * d +
                                                         ch0 + min(4 * c4, dch - 4));
                if (!in) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wave < n_units) {  // (the ids were requested before the slab: they are here first)
            const uint4 *arow = table + (size_t)gid * (4 * SF_TA * ks) + lq;
#pragma unroll
            for (int u = 0; u < SF_PF; ++u)
#pragma unroll
                for (int t = 0; t < SF_TA; ++t) a[u][t] = arow[4 * SF_TA * min(u, ks - 1) + 4 * t];
        }
        float sr[SF_STAGE / 256];
#pragma unroll
        for (int q = 0; q < SF_STAGE / 256; ++q) sr[q] = (q * 256 < n_stage) ? rinv[sg[q]] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) wmax[wave] = mx;
        if (backgrounds && tid < dch) bgs[tid] = backgrounds[ch0 + tid];
#pragma unroll
        for (int q = 0; q < SF_STAGE / 256; ++q)
            if (q * 256 < n_stage) { gids[tid + q * 256] = sg[q]; rinvs[tid + q * 256] = sr[q]; }
        __syncthreads();
        mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        const int ebits = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        const bool sane = ebits >= 16 && ebits <= 240;  // an all-zero (or non-finite) slab keeps scale 1
        const float bs = sane ? __uint_as_float((unsigned)(268 - ebits) << 23) : 1.0f;  // largest magnitude -> [2^14, 2^15)
        ibs = sane ? __uint_as_float((unsigned)(ebits - 14) << 23) : 1.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            if (i < total) {
                const int e = i / q4, c4 = i - e * q4;
                const f32x2v_t a = {v[u].x * bs, v[u].y * bs}, b = {v[u].z * bs, v[u].w * bs};
                const f16x2_t ah = __builtin_convertvector(a, f16x2_t), bh = __builtin_convertvector(b, f16x2_t);
                const f16x2_t al = __builtin_convertvector(a - __builtin_convertvector(ah, f32x2v_t), f16x2_t);
                const f16x2_t bl = __builtin_convertvector(b - __builtin_convertvector(bh, f32x2v_t), f16x2_t);
                typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
                const f16x4_t h4 = {ah[0], ah[1], bh[0], bh[1]}, l4 = {al[0], al[1], bl[0], bl[1]};
                *reinterpret_cast<f16x4_t *>(hi + e * pitch + 4 * c4) = h4;
                *reinterpret_cast<f16x4_t *>(lo + e * pitch + 4 * c4) = l4;
            }
        }
    }
    __syncthreads();
    if (backgrounds) {  // <bg, v_px> per pixel of the block, for the background term of v_alpha (from the two planes)
        // eight channels at a time, the background slice from LDS (a global load per channel, waited for one by one,
        // was 12.8 of this workgroup's 30 microseconds)
        const int e = tid & 63, part = tid >> 6;
        float acc = 0.f;
        for (int c = 8 * part; c < dch; c += 32) {
            const f16x8 h8 = *reinterpret_cast<const f16x8 *>(hi + e * pitch + c), l8 = *reinterpret_cast<const f16x8 *>(lo + e * pitch + c);
            const float4 b0 = *reinterpret_cast<const float4 *>(bgs + c), b1 = *reinterpret_cast<const float4 *>(bgs + c + 4);
            acc = fmaf(b0.x, (float)h8[0] + (float)l8[0], acc); acc = fmaf(b0.y, (float)h8[1] + (float)l8[1], acc);
            acc = fmaf(b0.z, (float)h8[2] + (float)l8[2], acc); acc = fmaf(b0.w, (float)h8[3] + (float)l8[3], acc);
            acc = fmaf(b1.x, (float)h8[4] + (float)l8[4], acc); acc = fmaf(b1.y, (float)h8[5] + (float)l8[5], acc);
            acc = fmaf(b1.z, (float)h8[6] + (float)l8[6], acc); acc = fmaf(b1.w, (float)h8[7] + (float)l8[7], acc);
        }
        bred[part][e] = acc;
        __syncthreads();
        if (tid < 64) {
            const int pp = tid >> 1, h = tid & 1;
            const int pj = bx0 + (pp & 7), pi = by0 + (pp >> 3) + 4 * h;
            if (pi < height && pj < width) {
                const float t = ((bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid])) * ibs;
                bgdot[(size_t)pi * width + pj] = t;
            }
        }
    }
    auto unit_gid = [&](int unit) __attribute__((always_inline)) {  // staged; a block of more than SF_STAGE slots: the rest from memory
        const int j = min(min(unit, n_units - 1) * 16 + lr, cnt - 1);
        return j < SF_STAGE ? gids[j] : slot_gid(j);
    };
    int gid_n = unit_gid(wave + 4);
    for (int unit = wave; unit < n_units; unit += 4) {
        const int t0 = unit * 16;
        const uint4 *arow = table + (size_t)gid * (4 * SF_TA * ks) + lq;      // step j, term t: + 4 SF_TA j + 4 t
        const uint4 *arow_n = table + (size_t)gid_n * (4 * SF_TA * ks) + lq;  // the wave's next unit (the last one: itself again)
        const int jm = min(t0 + m, cnt - 1);
        const float ri = (jm < SF_STAGE ? rinvs[jm] : rinv[slot_gid(jm)]) * ibs;
        const int gid_nn = unit_gid(unit + 8);
        const _Float16 *bh0 = hi + m * pitch + 8 * kq, *bl0 = lo + m * pitch + 8 * kq;  // pixel group pg: + 16 pg pitch
        f32x4_t acc[4];
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) acc[pg] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto step = [&](int j, int u) __attribute__((always_inline)) {
            uint4 pa[3];
#pragma unroll
            for (int t = 0; t < SF_TA; ++t) {
                pa[t].x = (unsigned)__builtin_amdgcn_ds_bpermute(perm_src, (int)a[u][t].x);
                pa[t].y = (unsigned)__builtin_amdgcn_ds_bpermute(perm_src, (int)a[u][t].y);
                pa[t].z = (unsigned)__builtin_amdgcn_ds_bpermute(perm_src, (int)a[u][t].z);
                pa[t].w = (unsigned)__builtin_amdgcn_ds_bpermute(perm_src, (int)a[u][t].w);
            }
            if constexpr (SF_TA == 2) pa[2] = pa[1];
            const f16x8 a0 = *reinterpret_cast<const f16x8 *>(&pa[0]);
            const f16x8 a1 = *reinterpret_cast<const f16x8 *>(&pa[1]);
            const f16x8 a2 = *reinterpret_cast<const f16x8 *>(&pa[2]);
            {   // slot u next holds step j + SF_PF of this unit, or -- past its end -- step u of the next one
                const uint4 *src = (j + SF_PF < ks) ? arow + 4 * SF_TA * (j + SF_PF) : arow_n + 4 * SF_TA * min(u, ks - 1);
#pragma unroll
                for (int t = 0; t < SF_TA; ++t) a[u][t] = src[4 * t];
            }
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                bh[pg] = *reinterpret_cast<const f16x8 *>(bh0 + 16 * pg * pitch + 32 * j);
                bl[pg] = *reinterpret_cast<const f16x8 *>(bl0 + 16 * pg * pitch + 32 * j);
            }
            // One scheduling fence per step, BETWEEN its reads and its MFMAs: the next step's permutes and LDS reads may
            // rise into this step's MFMAs (and no further; without any fence every read of the unit went to the top: 96 spills).
            __builtin_amdgcn_sched_barrier(0);
            // smallest terms first; the four accumulators in turn (no MFMA waits for the one before it)
            if constexpr (SF_TA == 3) {
#pragma unroll
                for (int pg = 0; pg < 4; ++pg) acc[pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bh[pg], acc[pg], 0, 0, 0);
#pragma unroll
                for (int pg = 0; pg < 4; ++pg) acc[pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl[pg], acc[pg], 0, 0, 0);
            }
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) acc[pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh[pg], acc[pg], 0, 0, 0);
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) acc[pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bl[pg], acc[pg], 0, 0, 0);
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) acc[pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bh[pg], acc[pg], 0, 0, 0);
        };
        if constexpr (KS != 0) {
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                step(j, j % SF_PF);
            }
        } else {
            for (int jb = 0; jb < ks; jb += SF_PF) {
#pragma unroll
                for (int u = 0; u < SF_PF; ++u)
                    if (jb + u < ks) step(jb + u, u);
            }
        }
        gid = gid_n;
        gid_n = gid_nn;
        // accumulator pg: column = pixel element 16 pg + m, rows = slots 4 kq + i; lane m holds the scale of slot m's row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sc = __shfl(ri, 4 * kq + i);
            const int row = t0 + 4 * kq + i;
            if (row < cnt) {
#pragma unroll
                for (int pg = 0; pg < 4; ++pg) S[(size_t)(srb + row) * 64 + 16 * pg + m] = acc[pg][i] * sc;
            }
        }
    }
}

// wave64 sum on the VALU (DPP: quad swaps, row mirrors, row broadcasts; the total lands in lane 63), returned
// wave-uniform -- __shfl_xor goes through the LDS crossbar, and six sums per slot made that the kernel's bound
__device__ __forceinline__ float geom_wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__global__ __launch_bounds__(64) void raster_bwd_geom(int width, int height, int tile_w, int n_tiles, int n_gauss,
                                                      const GRec *__restrict__ packed, const int32_t *__restrict__ offsets,
                                                      int n_isects, const int32_t *__restrict__ blk_rows,
                                                      const float *__restrict__ wt, const int32_t *__restrict__ gid_s,
                                                      const int32_t *__restrict__ sidx_s, const float *__restrict__ S,
                                                      const float *__restrict__ Tbuf, const float *__restrict__ v_alphas,
                                                      const float *__restrict__ bgdot, float *__restrict__ grow,
                                                      uint32_t *__restrict__ key, int32_t *__restrict__ idx, int by_gauss,
                                                      const int32_t *__restrict__ row_base, int n_pass, size_t s_stride,
                                                      size_t bg_stride)
{   // n_pass > 1: the dot pass left one S / bgdot buffer per 256-channel pass (s_stride, bg_stride floats apart): summed here
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE);
    const int blk = logical & 3;
    const int tile = gags_tile_of_order(logical >> 2, tile_w, n_tiles / tile_w);
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    if (cnt == 0) return;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    // row of slot j: compact numbering (row_base = exclusive prefix sum of blk_rows) or the sparse slot index itself
    const int rb = row_base ? row_base[tile * GAGS_BLOCKS_PER_TILE + blk] : sb;
    const int lane = threadIdx.x;  // = element e of the weight rows: pixel p = e >> 1 of the 8x4 half h = e & 1
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int pj = tx * GAGS_TILE + (blk & 1) * 8 + ((lane >> 1) & 7);
    const int pi = ty * GAGS_TILE + (blk >> 1) * 8 + (lane >> 4) + 4 * (lane & 1);
    const bool inside = pi < height && pj < width;
    const size_t pix = inside ? (size_t)pi * width + pj : 0;
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float T_final = inside ? Tbuf[pix] : 1.f;
    // T_f ra (v_a - <bg, v>): everything but ra is per pixel
    float bgd = 0.f;
    if (bgdot && inside) {
        bgd = bgdot[pix];
        for (int p = 1; p < n_pass; ++p) bgd += bgdot[p * bg_stride + pix];
    }
    const float k0 = inside ? T_final * ((v_alphas ? v_alphas[pix] : 0.f) - bgd) : 0.f;
    float behind = 0.f;  // SUM f S over the slots behind the current one
    // two-deep software pipeline over the slots (back to front): the slot's ids and rows are requested two slots ahead,
    // its packed record (addressed by the id) one slot ahead
    struct Ids { int sx, g; float f, sd; };
    auto fetch_ids = [&](int j) __attribute__((always_inline)) {
        Ids q;
        const int slot = sb + max(j, 0);
        q.sx = sidx_s[slot]; q.g = gid_s[slot];
        q.f = wt[(size_t)slot * 64 + lane];
        const size_t so = (size_t)(rb + max(j, 0)) * 64 + lane;  // S is numbered like the per-slot rows (compact or sparse)
        q.sd = S[so];
        if (n_pass > 1) {  // (uniform; at most four passes: D <= 1024)
            q.sd += S[s_stride + so];
            if (n_pass > 2) q.sd += S[2 * s_stride + so];
            if (n_pass > 3) q.sd += S[3 * s_stride + so];
        }
        return q;
    };
    auto fetch_rec = [&](const Ids &q) __attribute__((always_inline)) {  // (the pad slot's gid is n_gauss: clamped, unused)
        return packed[by_gauss ? min(__builtin_amdgcn_readfirstlane(q.g), n_gauss - 1) : max(__builtin_amdgcn_readfirstlane(q.sx), 0)];
    };
    Ids i1 = fetch_ids(cnt - 1), i2 = fetch_ids(cnt - 2);
    GRec r1 = fetch_rec(i1);
    for (int j = cnt - 1; j >= 0; --j) {
        const Ids cur = i1;
        const GRec r = r1;
        i1 = i2;
        r1 = fetch_rec(i1);
        i2 = fetch_ids(j - 2);
        __builtin_amdgcn_sched_barrier(0);  // keep the requests above the arithmetic of the current slot
        const int row = rb + j;
        const int sx = __builtin_amdgcn_readfirstlane(cur.sx);
        if (sx < 0) {  // pad slot: sorts behind every Gaussian
            if (lane == 0) { key[row] = (uint32_t)n_gauss; idx[row] = row; }
            continue;
        }
        const int g = __builtin_amdgcn_readfirstlane(cur.g);
        const float f = cur.f, sdot = cur.sd;
        const float dx = r.x - px, dy = r.y - py;
        const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
        const float vis = gags_exp_neg(sigma);
        const float alpha = fminf(GAGS_ALPHA_MAX, r.o * vis);
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f, g5 = 0.f;
        if (f != 0.f) {  // blended (f = alpha T with alpha >= 1/255 and T > 1e-4)
            const float ra = 1.0f / (1.0f - alpha);
            const float T = f / alpha;
            const float v_alpha = T * sdot - ra * behind + k0 * ra;
            behind = fmaf(f, sdot, behind);
            if (r.o * vis <= GAGS_ALPHA_MAX) {
                const float v_sigma = -r.o * vis * v_alpha;
                g0 = 0.5f * v_sigma * dx * dx;
                g1 = v_sigma * dx * dy;
                g2 = 0.5f * v_sigma * dy * dy;
                g3 = v_sigma * (r.a * dx + r.b * dy);
                g4 = v_sigma * (r.b * dx + r.c * dy);
                g5 = vis * v_alpha;
            }
        }
        g0 = geom_wave_sum(g0); g1 = geom_wave_sum(g1); g2 = geom_wave_sum(g2);
        g3 = geom_wave_sum(g3); g4 = geom_wave_sum(g4); g5 = geom_wave_sum(g5);
        if (lane < 8) {
            const float v = lane == 0 ? g0 : lane == 1 ? g1 : lane == 2 ? g2 : lane == 3 ? g3 : lane == 4 ? g4 : lane == 5 ? g5 : 0.f;
            grow[(size_t)row * 8 + lane] = v;
        }
        if (lane == 0) { key[row] = (uint32_t)g; idx[row] = row; }
    }
}

struct GeomLayout {
    int64_t S, bgdot, grow, key, idx, key_s, idx_s, seg, table, rinv, mask, sort, total;
};
// n_rows < 0: one row per slot of the sparse slot space; else the compact row count (sum of blk_rows)
inline GeomLayout geom_layout(int64_t n_isects, int width, int height, int n_gauss, int d, int64_t n_rows)
{
    const int64_t tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int64_t slots = gags_slot_count(n_isects, tile_w * tile_h);
    const int64_t rows = n_rows < 0 ? slots : (n_rows > 0 ? n_rows : 1);
    GeomLayout L;
    int64_t o = 0;
    const int64_t n_pass = std::max(1, (d + SD_MAXCH - 1) / SD_MAXCH);  // the split-f16 dot pass: one S / bgdot buffer per pass
    L.S = o; o += n_pass * al256((rows + 64) * 256);  // numbered like the per-slot rows: compact (n_rows) or the sparse slot space
    L.bgdot = o; o += n_pass * al256((int64_t)width * height * 4);
    L.grow = o; o += al256(rows * 32);
    L.key = o; o += al256(rows * 4);
    L.idx = o; o += al256(rows * 4);
    L.key_s = o; o += al256(rows * 4);
    L.idx_s = o; o += al256(rows * 4);
    L.seg = o; o += al256(((int64_t)n_gauss + 2) * 4);
    // split feature table of one pass (<= 256 channels: 6 bytes per channel), 1 / row scale, blended mask
    const int64_t ks = (std::min(d, SD_MAXCH) + 31) / 32;  // 32-channel steps, 192 bytes each
    L.table = o; o += al256((int64_t)n_gauss * ks * 192);
    L.rinv = o; o += al256((int64_t)n_gauss * 4);
    L.mask = o; o += al256((int64_t)n_gauss);
    L.sort = o; o += al256(gags_sort_u32_scratch_bytes(rows));
    L.total = o;
    return L;
}

int64_t gags_raster_bwd_geom_scratch_bytes_impl(int64_t n_isects, int width, int height, int n_gauss, int d, int64_t n_rows)
{
    return geom_layout(n_isects, width, height, n_gauss, d, n_rows).total;
}

// 1 = width not eligible (d % 8 != 0 or d < 16)
int gags_raster_bwd_geom_launch(int d, int n_gauss, int width, int height, const float *colors, const float *backgrounds,
                                const int32_t *offsets, int n_isects, const void *packed, const float *v_out,
                                const float *v_alphas, const int32_t *blk_rows, const float *wt, const int32_t *gid_s,
                                const int32_t *sidx_s, const float *Tbuf, void *scratch, int64_t scratch_bytes, float *v_geo,
                                int by_gauss, const int32_t *row_base, int64_t n_rows, const int32_t *hit,
                                const int32_t *flatten_ids, int f32mfma, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (d < 16 || d % 8 != 0) return 1;
    if (!row_base) n_rows = -1;
    const GeomLayout L = geom_layout(n_isects, width, height, n_gauss, d, n_rows);
    if (scratch_bytes < L.total) return GAGS_ESCRATCH;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    const int64_t slots = n_rows < 0 ? gags_slot_count(n_isects, n_tiles) : n_rows;  // rows to sort
    char *sb = (char *)scratch;
    float *S = (float *)(sb + L.S), *bgdot = backgrounds ? (float *)(sb + L.bgdot) : nullptr, *grow = (float *)(sb + L.grow);
    uint32_t *key = (uint32_t *)(sb + L.key), *key_s = (uint32_t *)(sb + L.key_s);
    int32_t *idx = (int32_t *)(sb + L.idx), *idx_s = (int32_t *)(sb + L.idx_s), *seg = (int32_t *)(sb + L.seg);
    // unused slots of the sparse slot space sort behind every Gaussian (the compact numbering has none)
    if (n_rows < 0 && hipMemsetD32Async((hipDeviceptr_t)key, n_gauss, (size_t)slots, st) != hipSuccess) return GAGS_ELAUNCH;
    static bool attr_set = false;
    const int lds_max = 64 * (SD_MAXCH + SD_PAD) * 4;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)raster_bwd_sdot<SD_MAXCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess ||
            hipFuncSetAttribute((const void *)raster_bwd_sdot<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess)
            return GAGS_ELAUNCH;
        attr_set = true;
    }
    const bool split16 = !f32mfma && hit && flatten_ids;
    // the split-f16 dot pass writes one S / bgdot buffer per 256-channel pass (the fp32 one accumulates in the first)
    const int64_t s_rows = (n_rows < 0 ? gags_slot_count(n_isects, n_tiles) : (n_rows > 0 ? n_rows : 1)) + 64;
    const size_t s_stride = (size_t)al256(s_rows * 256) / 4, bg_stride = (size_t)al256((int64_t)width * height * 4) / 4;
    const int n_pass = split16 ? (d + SD_MAXCH - 1) / SD_MAXCH : 1;
    if (split16) {
        static bool attr16_set = false;
        const int lds16_max = 2 * 64 * (SD_MAXCH + SF_PADH) * 2;
        if (!attr16_set) {
            if (hipFuncSetAttribute((const void *)raster_bwd_sdot_f16<SD_MAXCH / 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16_max) != hipSuccess ||
                hipFuncSetAttribute((const void *)raster_bwd_sdot_f16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16_max) != hipSuccess)
                return GAGS_ELAUNCH;
            attr16_set = true;
        }
        unsigned char *mask = (unsigned char *)(sb + L.mask);
        uint4 *table = (uint4 *)(sb + L.table);
        float *rinv = (float *)(sb + L.rinv);
        if (hipMemsetAsync(mask, 0, (size_t)n_gauss, st) != hipSuccess) return GAGS_ELAUNCH;
        if (n_isects > 0) {
            const int rc = gags_blended_mask_launch(n_isects, hit, flatten_ids, mask, st);
            if (rc != GAGS_OK) return rc;
        }
        for (int ch0 = 0; ch0 < d; ch0 += SD_MAXCH) {
            const int ks = (min(SD_MAXCH, d - ch0) + 31) / 32;
            float *S_p = S + (size_t)(ch0 / SD_MAXCH) * s_stride, *bg_p = bgdot ? bgdot + (size_t)(ch0 / SD_MAXCH) * bg_stride : nullptr;
            hipLaunchKernelGGL(feat_split_kernel, dim3((n_gauss + 7) / 8), dim3(256), 0, st, n_gauss, d, ch0, ks, colors, mask, table, rinv);
            const dim3 grid(n_tiles * GAGS_BLOCKS_PER_TILE);
            const size_t lds = (size_t)2 * 64 * (32 * ks + SF_PADH) * 2;
            if (ks == SD_MAXCH / 32)
                hipLaunchKernelGGL(raster_bwd_sdot_f16<SD_MAXCH / 32>, grid, dim3(256), lds, st, d, ch0, ks, width, height, tile_w, n_tiles,
                                   n_gauss, v_out, table, rinv, backgrounds, offsets, n_isects, blk_rows, gid_s, S_p, bg_p, row_base);
            else
                hipLaunchKernelGGL(raster_bwd_sdot_f16<0>, grid, dim3(256), lds, st, d, ch0, ks, width, height, tile_w, n_tiles,
                                   n_gauss, v_out, table, rinv, backgrounds, offsets, n_isects, blk_rows, gid_s, S_p, bg_p, row_base);
        }
    }
    for (int ch0 = 0; ch0 < d && !split16; ch0 += SD_MAXCH) {
        const int dch = min(SD_MAXCH, d - ch0);
        const dim3 grid(n_tiles * GAGS_BLOCKS_PER_TILE);
        const size_t lds = (size_t)64 * (dch + SD_PAD) * 4;
        if (dch == SD_MAXCH)
            hipLaunchKernelGGL(raster_bwd_sdot<SD_MAXCH>, grid, dim3(256), lds, st, d, ch0, dch, width, height, tile_w, n_tiles, n_gauss,
                               v_out, colors, backgrounds, offsets, n_isects, blk_rows, gid_s, S, bgdot, ch0 > 0 ? 1 : 0, row_base);
        else
            hipLaunchKernelGGL(raster_bwd_sdot<0>, grid, dim3(256), lds, st, d, ch0, dch, width, height, tile_w, n_tiles, n_gauss,
                               v_out, colors, backgrounds, offsets, n_isects, blk_rows, gid_s, S, bgdot, ch0 > 0 ? 1 : 0, row_base);
    }
    hipLaunchKernelGGL(raster_bwd_geom, dim3(n_tiles * GAGS_BLOCKS_PER_TILE), dim3(64), 0, st, width, height, tile_w, n_tiles,
                       n_gauss, reinterpret_cast<const GRec *>(packed), offsets, n_isects, blk_rows, wt, gid_s, sidx_s, S, Tbuf,
                       v_alphas, bgdot, grow, key, idx, by_gauss, row_base, n_pass, s_stride, bg_stride);
    GAGS_CHECK_LAUNCH();
    int nbits = 1;
    while ((1ll << nbits) <= n_gauss) ++nbits;  // keys in [0, n_gauss]
    if (slots > 0) {
        const int rc = gags_sort_pairs_u32(slots, nbits, key, idx, key_s, idx_s, sb + L.sort, L.total - L.sort, st);
        if (rc != GAGS_OK) return rc;
        hipLaunchKernelGGL(seg_offsets_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, (int)slots, key_s, n_gauss, seg);
    } else {
        hipLaunchKernelGGL(seg_fill_kernel, dim3((n_gauss + 1 + 255) / 256), dim3(256), 0, st, n_gauss, seg);
    }
    hipLaunchKernelGGL((reduce_rows_kernel<false, 4>), dim3((n_gauss + 128 * REDUCE_ITER - 1) / (128 * REDUCE_ITER)), dim3(256), 0, st, n_gauss, 8, 0, 8, seg, idx_s, grow, 8,
                       (void *)v_geo, 0, (const int32_t *)nullptr, (float *)nullptr, (const uint8_t *)nullptr, (uint8_t *)nullptr, 0);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// 1 = width not eligible (d % 128 != 0)
int gags_raster_bwd_atomic_launch(int d, int width, int height, const void *packed, const int32_t *offsets,
                                  const int32_t *flat, int n_isects, const float *v_out, float *v_colors,
                                  int by_gauss, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (d < CSB || d % CSB != 0) return 1;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / CSB;
    hipLaunchKernelGGL(raster_bwd_atomic, dim3(n_tiles * 8 * n_slices), dim3(64), 0, st, d, width, height, tile_w,
                       n_tiles, n_slices, reinterpret_cast<const GRec *>(packed), offsets, flat, n_isects, v_out,
                       v_colors, by_gauss);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
