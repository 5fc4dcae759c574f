// K9 / K10, MFMA flavour: wide feature rasterization (D a multiple of 32) on the gfx950 matrix
// cores with EXACT fp32 arithmetic (v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain,
// so the forward stays bit-identical to the sequential definition).
//
// Decomposition (DESIGN.md "raster_fwd_mfma"):
//   workgroup = one 16x16 tile x one slice of CS = 32*NB channels, 8 waves;
//   wave w    = the 8x4 pixel block (w&1, w>>1) x all CS channels: NB accumulator tiles of
//               32 px x 32 ch (16 VGPRs each).
//   The tile's depth-sorted Gaussian range is staged through LDS in chunks of GC = 64:
//   (xy, conic, opacity, screen extent) records + the chunk's feature rows [GC][CS] (coalesced
//   16 B/lane global reads).  Each wave COMPACTS the chunk to the Gaussians whose alpha >= 1/255
//   footprint can touch its 32 pixels (conservative extent test -> ballot), then walks the hits
//   two at a time: lane (pixel p = lane&31, k = lane>>5) evaluates alpha for hit 2s+k, one
//   v_permlane32_swap gives every lane both alphas, the transmittance chain runs redundantly in
//   both half-waves, and w[p][k] = alpha*T IS the MFMA A operand (32x2).  B operands (2 x 32
//   feature values) come from LDS with one ds_read_b128 per 4 accumulator tiles: accumulator tile
//   i of a group holds channels {4n+i}, so the epilogue stores 16 B per lane.
//   Zero weights (skipped / terminated pixels) are exact no-ops of the fmaf chain.
// All channels of the slice are composited in ONE walk of the list (no 32-wide re-walks).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GRec {  // 32 B, 16-B aligned
    float x, y, a, b, c, o, ex, ey;
};

__device__ __forceinline__ GRec load_grec(const float *__restrict__ means2d, const float *__restrict__ conics,
                                          const float *__restrict__ opacities, int g)
{
    GRec r;
    const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
    r.x = m.x; r.y = m.y;
    r.a = conics[3 * g]; r.b = conics[3 * g + 1]; r.c = conics[3 * g + 2];
    r.o = opacities[g];
    // conservative half-extent of {alpha >= 1/255}: sigma <= tau = ln(255 o); |dx| <= sqrt(2 tau Sxx)
    const float det = r.a * r.c - r.b * r.b;
    const float tau = __logf(255.0f * r.o) + 0.02f;
    if (!(tau > 0.f)) {
        r.ex = -1.f; r.ey = -1.f;  // can never reach 1/255
    } else if (!(det > 0.f)) {
        r.ex = 3.0e38f; r.ey = 3.0e38f;
    } else {
        const float s = 2.0f * tau / det;
        r.ex = sqrtf(s * r.c) * 1.001f + 0.01f;
        r.ey = sqrtf(s * r.a) * 1.001f + 0.01f;
    }
    return r;
}

// Per-pixel compositing state, replicated in both half-waves (lane p and lane p+32).
struct PixState {
    float T;
    int cur;
    bool done;
};

// alpha of this lane's Gaussian at this lane's pixel (0 when skipped by the A8 rule)
__device__ __forceinline__ float eval_alpha(const GRec &r, float px, float py, bool live)
{
    const float dx = r.x - px, dy = r.y - py;
    const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
    const float alpha = fminf(GAGS_ALPHA_MAX, r.o * gags_exp_neg(sigma));
    return (live && !(sigma < 0.f || alpha < GAGS_ALPHA_MIN)) ? alpha : 0.f;
}

// Packed per-intersection record, written once per view by gags_pack_isects in sorted order so
// that the raster kernels stream it with coalesced 32-B reads instead of three dependent gathers.
//   {x, y, conic a, b, c, opacity, half-extent x, half-extent y} of the alpha >= 1/255 footprint.
__global__ __launch_bounds__(256) void pack_isects_kernel(int n_isects, const int32_t *__restrict__ flatten_ids,
                                                          const float *__restrict__ means2d,
                                                          const float *__restrict__ conics,
                                                          const float *__restrict__ opacities,
                                                          GRec *__restrict__ packed)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_isects) return;
    packed[s] = load_grec(means2d, conics, opacities, flatten_ids[s]);
}

// Record kept in the per-wave LDS ring after the hit test: extents replaced by ids.
struct HRec {
    float x, y, a, b, c, o;
    int gid, sidx;
};

constexpr int RING = 256;  // per-wave ring of compacted hits (8 KB)

// Advance one pixel over the two Gaussians of a K-step (branch-free: selects only).  Both
// half-waves run the same chain; `blended` reports whether THIS lane's slot (k) was composited.
__device__ __forceinline__ float step_pair2(PixState &s, float a0, float a1, int k, bool &blended)
{
    const float t0 = s.T * (1.0f - a0);
    const bool ok0 = !s.done && a0 > 0.f;
    const bool stop0 = ok0 && t0 <= GAGS_T_STOP;
    const bool b0 = ok0 && !stop0;
    const float w0 = b0 ? a0 * s.T : 0.f;
    s.T = b0 ? t0 : s.T;
    s.done = s.done || stop0;
    const float t1 = s.T * (1.0f - a1);
    const bool ok1 = !s.done && a1 > 0.f;
    const bool stop1 = ok1 && t1 <= GAGS_T_STOP;
    const bool b1 = ok1 && !stop1;
    const float w1 = b1 ? a1 * s.T : 0.f;
    s.T = b1 ? t1 : s.T;
    s.done = s.done || stop1;
    blended = k ? b1 : b0;
    return k ? w1 : w0;
}

// Forward, wave-independent: no workgroup barrier anywhere.  A workgroup is ONE wave = one of
// the tile's eight 8x4 pixel blocks x one channel slice (so a finished wave frees its slot at
// once; blocks of a tile are consecutive workgroup ids, i.e. the same XCD / L2); each wave
// walks the tile's sorted range on its own:
//   produce : 64 packed records per pass (register-prefetched one pass ahead), extent test,
//             ballot, compaction of the hits into the wave's private LDS ring;
//   consume : two hits per K-step -- alpha, permlane32 swap, transmittance chain, then NB
//             v_mfma_f32_32x32x2_f32 with B operands (feature rows) loaded straight from
//             global/L2 into VGPRs one K-step ahead (the rows are shared by the 16 waves of a
//             tile and by neighbouring tiles: L2-resident; no LDS staging, no barrier).
template <int NB>
__global__ __launch_bounds__(64, (NB >= 16 ? 1 : 2)) void raster_fwd_mfma(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const GRec *__restrict__ packed,
    const float *__restrict__ colors, const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids, int32_t *__restrict__ blk_rows, int dbg)
{
    constexpr int CS = 32 * NB;
    constexpr int VEC = NB >= 4 ? 4 : NB;
    constexpr int NG = NB / VEC;
    static_assert(NB == 1 || NB == 2 || NB % 4 == 0, "NB in {1,2,4,8,16}");

    __shared__ __attribute__((aligned(16))) HRec ring[RING];
    __shared__ float Tb[32];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8 * n_slices);
    const int slice = logical % n_slices, rest = logical / n_slices;
    const int blk = rest & 7;
    const int tile = (dbg & 16) ? (rest >> 3) : gags_tile_of_order(rest >> 3, tile_w, n_tiles / tile_w);
    const int ch0 = slice * CS;
    const int lane = threadIdx.x;
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 4;
    const int pj = bx0 + (p & 7), pi = by0 + (p >> 3);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float rx0 = (float)bx0 + 0.5f, rx1 = (float)bx0 + 7.5f, ry0 = (float)by0 + 0.5f, ry1 = (float)by0 + 3.5f;

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !inside;

    // ---- producer state: chunk [c, c+64) is in registers (pre), next chunk is at c ----
    int nq = 0, rd = 0;  // hits produced / consumed (wave-uniform); ring index = count & (RING-1)
    int c = start;
    GRec pre;
    int pre_gid = 0, pre_c = start;
    auto issue = [&]() {  // start loading the chunk at c
        pre_c = c;
        const int idx = c + lane;
        if (idx < end) {
            const float4 *src = reinterpret_cast<const float4 *>(packed + idx);
            const float4 u = src[0], v = src[1];
            pre.x = u.x; pre.y = u.y; pre.a = u.z; pre.b = u.w; pre.c = v.x; pre.o = v.y; pre.ex = v.z; pre.ey = v.w;
            pre_gid = flatten_ids[idx];
        } else {
            pre.x = pre.y = 0.f; pre.ex = pre.ey = -1.f; pre.a = pre.b = pre.c = pre.o = 0.f;
        }
        c += 64;
    };
    auto commit = [&]() {  // hit-test the chunk in registers, append the hits to the ring
        const bool hit = (pre_c + lane < end) && (pre.x + pre.ex >= rx0) && (pre.x - pre.ex <= rx1) &&
                         (pre.y + pre.ey >= ry0) && (pre.y - pre.ey <= ry1);
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int pos = nq + __popcll(mask & ((1ull << lane) - 1ull));
            HRec h;
            h.x = pre.x; h.y = pre.y; h.a = pre.a; h.b = pre.b; h.c = pre.c; h.o = pre.o;
            h.gid = pre_gid; h.sidx = pre_c + lane;
            ring[pos & (RING - 1)] = h;
        }
        nq += __popcll(mask);
    };
    bool pending = false;  // a chunk is in flight in `pre`
    if (c < end) { issue(); pending = true; }
    auto refill = [&](int low) {
        while ((nq - rd) < low && pending) {
            commit();
            pending = false;
            if (c < end) { issue(); pending = true; }
        }
    };

    // B operands for the pair at ring position `pos`: lane (p,k) reads VEC floats of the row of slot
    // pos+k.  Always addresses a produced slot (clamped), so it is issued unconditionally one step ahead.
    auto load_b = [&](int pos, float4(&bq)[NG]) {
        const int slot = min(pos + k, nq - 1);
        const int gid = ring[slot & (RING - 1)].gid;
        const float *row = colors + (size_t)gid * d + ch0 + VEC * p;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if constexpr (VEC == 4) bq[gq] = *reinterpret_cast<const float4 *>(row + gq * 128);
            else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2 *>(row); bq[gq] = make_float4(t.x, t.y, 0.f, 0.f); }
            else bq[gq] = make_float4(row[0], 0.f, 0.f, 0.f);
        }
    };
    // alpha of this lane's slot of the pair at `pos` (0 past the end of the list) + its sorted index
    auto eval_at = [&](int pos, int &sidx) {
        const bool valid = pos + k < nq;
        const HRec h = ring[min(pos + k, nq - 1) & (RING - 1)];
        GRec r;
        r.x = h.x; r.y = h.y; r.a = h.a; r.b = h.b; r.c = h.c; r.o = h.o;
        sidx = h.sidx;
        return eval_alpha(r, px, py, valid);
    };

    // Software pipeline, one basic block per K-step: alpha(s+1) and the feature-row loads of
    // step s+1 do not depend on the transmittance chain / MFMAs of step s, so they overlap.
    // Two copies of the step with swapped B buffers avoid register copies.
    refill(6);
    float a_n = 0.f;
    int sidx_n = 0;
    int nrows = 0;  // K-steps that blended anything, x2: the staged backward's row slots for this block
    float4 b0[NG], b1[NG];
    auto kstep = [&](float4(&bc)[NG], float4(&bn)[NG]) -> bool {
        const float a_c = a_n;
        const int sidx_c = sidx_n;
        rd += 2;
        if ((nq - rd) < 6 && pending) refill(6);
        const bool more = rd < nq;
        a_n = eval_at(rd, sidx_n);  // clamped slot; weight forced to 0 past the end
        load_b(rd, bn);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
        bool blended;
        const float wgt = step_pair2(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
        st.cur = blended ? sidx_c : st.cur;
        nrows += __any(wgt != 0.f) ? 2 : 0;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            const float bv[4] = {bc[gq].x, bc[gq].y, bc[gq].z, bc[gq].w};
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                acc[gq * VEC + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wgt, bv[i], acc[gq * VEC + i], 0, 0, 0);
        }
        return more && !__all(st.done);
    };
    if (!__all(st.done) && rd < nq) {
        a_n = eval_at(rd, sidx_n);
        load_b(rd, b0);
        // two K-steps per trip, ONE exit test: a single loop exit keeps the 128 accumulator registers
        // in place (two exits make the register allocator shuffle them); a surplus K-step past the end
        // of the list or past saturation carries zero weights, i.e. is an exact no-op.
        bool go = true;
        while (go) {
            kstep(b0, b1);
            go = kstep(b1, b0);
        }
    }

    // ---- epilogue ----
    {
        const auto cs = __builtin_amdgcn_permlane32_swap((unsigned)st.cur, (unsigned)st.cur, false, false);
        st.cur = max((int)cs[0], (int)cs[1]);  // sorted indices grow along the list
    }
    if (k == 0) Tb[p] = st.T;
    __builtin_amdgcn_wave_barrier();
    if (k == 0 && inside && slice == 0) {
        const size_t pix = (size_t)pi * width + pj;
        render_alphas[pix] = 1.0f - st.T;
        last_ids[pix] = st.cur;
    }
    if (blk_rows && slice == 0 && lane == 0) blk_rows[tile * 8 + blk] = nrows;
    float bgv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bgv[j] = 0.f;
    const bool has_bg = backgrounds != nullptr;  // wave-uniform
    if (has_bg) {
#pragma unroll
        for (int gq = 0; gq < NG; ++gq)
#pragma unroll
            for (int i = 0; i < VEC; ++i) bgv[gq * VEC + i] = backgrounds[ch0 + gq * 32 * VEC + VEC * p + i];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * k;  // accumulator row r <-> pixel q of the block
        const int qj = bx0 + (q & 7), qi = by0 + (q >> 3);
        if (qi >= height || qj >= width) continue;
        const float Tq = Tb[q];
        float *o = render_colors + ((size_t)qi * width + qj) * d + ch0 + VEC * p;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float a = acc[gq * VEC + i][r];
                v[i] = has_bg ? __builtin_fmaf(Tq, bgv[gq * VEC + i], a) : a;
            }
            if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o + gq * 128) = make_float4(v[0], v[1], v[2], v[3]);
            else if constexpr (VEC == 2) *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
            else o[0] = v[0];
        }
    }
}

template <int NB>
int launch_fwd_mfma(int d, int width, int height, const GRec *packed, const float *colors, const float *backgrounds,
                    const int32_t *offsets, const int32_t *flat, int n_isects, float *out, float *alphas,
                    int32_t *last_ids, int32_t *blk_rows, int dbg, hipStream_t st)
{
    constexpr int CS = 32 * NB;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / CS;
    hipLaunchKernelGGL(raster_fwd_mfma<NB>, dim3(n_tiles * 8 * n_slices), dim3(64), 0, st, d, width, height, tile_w,
                       n_tiles, n_slices, packed, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids,
                       blk_rows, dbg);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}


// ------------------------------------------------------------------------------------------
// K10 colours-only backward (the GAD flow: only d loss / d colors is consumed,
// scene/gaussian_model.py:192-208):   v_colors[g, :] += sum_px w[px, g] * v_out[px, :]
// with w = alpha*T recomputed FRONT TO BACK by exactly the forward's arithmetic (same hits,
// same stop decisions, so neither render_alphas nor last_ids nor the features are read).
// One wave per workgroup = one 8x4 pixel block x CSB = 128 channels, no barriers:
//   - the cotangent slab v_out[32 px][128 ch] lives in 64 VGPRs as MFMA B operands
//     (K = pixel pairs, N = channels), loaded once;
//   - producer / consumer exactly as the forward (packed records -> extent test -> LDS ring ->
//     two hits per step, alpha pipelined one step ahead); the weights of a step are transposed
//     through an LDS tile Wt[32 slots][32 px] (row stride 36 dwords: conflict-free
//     ds_write_b32 / ds_read_b128);
//   - every 32 slots: A = Wt^T fragments, 16 K-steps x 4 channel tiles of
//     v_mfma_f32_32x32x2_f32, then one 128-B coalesced float atomic per (Gaussian, channel tile).
// ------------------------------------------------------------------------------------------
constexpr int NBB = 4;            // channel tiles per wave in the backward
constexpr int CSB = 32 * NBB;     // 128 channels per workgroup
constexpr int WT_STRIDE = 36;     // dwords per slot row of the transpose tile

__device__ __forceinline__ void atomic_add_f32(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64, 2) void raster_bwd_colors_mfma(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const GRec *__restrict__ packed,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ v_render_colors, float *__restrict__ v_colors, int dbg)
{
    __shared__ __attribute__((aligned(16))) HRec ring[RING];
    __shared__ __attribute__((aligned(16))) float Wt[32 * WT_STRIDE];
    __shared__ int32_t slot_id[32];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8 * n_slices);
    const int slice = logical % n_slices, rest = logical / n_slices;
    const int blk = rest & 7;
    const int tile = (dbg & 16) ? (rest >> 3) : gags_tile_of_order(rest >> 3, tile_w, n_tiles / tile_w);
    const int ch0 = slice * CSB;
    const int lane = threadIdx.x;
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 4;
    const int pj = bx0 + (p & 7), pi = by0 + (p >> 3);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float rx0 = (float)bx0 + 0.5f, rx1 = (float)bx0 + 7.5f, ry0 = (float)by0 + 0.5f, ry1 = (float)by0 + 3.5f;

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];

    // cotangent slab as B operands: V[s][j] = v_out[pixel q = 2s+k][ch0 + 32j + p]
    float V[16][NBB];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int q = 2 * s + k;
        const int qj = bx0 + (q & 7), qi = by0 + (q >> 3);
        const bool ok = (qi < height) && (qj < width);
        const float *src = v_render_colors + ((size_t)(ok ? qi : 0) * width + (ok ? qj : 0)) * d + ch0 + p;
#pragma unroll
        for (int j = 0; j < NBB; ++j) V[s][j] = ok ? src[32 * j] : 0.f;
    }

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !inside;

    // ---- producer (same as the forward) ----
    int nq = 0, rd = 0;
    int c = start;
    GRec pre;
    int pre_gid = 0, pre_c = start;
    auto issue = [&]() {
        pre_c = c;
        const int idx = c + lane;
        if (idx < end) {
            const float4 *src = reinterpret_cast<const float4 *>(packed + idx);
            const float4 u = src[0], v = src[1];
            pre.x = u.x; pre.y = u.y; pre.a = u.z; pre.b = u.w; pre.c = v.x; pre.o = v.y; pre.ex = v.z; pre.ey = v.w;
            pre_gid = flatten_ids[idx];
        } else {
            pre.x = pre.y = 0.f; pre.ex = pre.ey = -1.f; pre.a = pre.b = pre.c = pre.o = 0.f;
        }
        c += 64;
    };
    auto commit = [&]() {
        const bool hit = (pre_c + lane < end) && (pre.x + pre.ex >= rx0) && (pre.x - pre.ex <= rx1) &&
                         (pre.y + pre.ey >= ry0) && (pre.y - pre.ey <= ry1);
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int pos = nq + __popcll(mask & ((1ull << lane) - 1ull));
            HRec h;
            h.x = pre.x; h.y = pre.y; h.a = pre.a; h.b = pre.b; h.c = pre.c; h.o = pre.o;
            h.gid = pre_gid; h.sidx = pre_c + lane;
            ring[pos & (RING - 1)] = h;
        }
        nq += __popcll(mask);
    };
    bool pending = false;
    if (c < end) { issue(); pending = true; }
    auto refill = [&](int low) {
        while ((nq - rd) < low && pending) {
            commit();
            pending = false;
            if (c < end) { issue(); pending = true; }
        }
    };
    auto eval_at = [&](int pos, int &gid) {
        const bool valid = pos + k < nq;
        const HRec h = ring[min(pos + k, nq - 1) & (RING - 1)];
        GRec r;
        r.x = h.x; r.y = h.y; r.a = h.a; r.b = h.b; r.c = h.c; r.o = h.o;
        gid = valid ? h.gid : -1;
        return eval_alpha(r, px, py, valid);
    };

    int nh = 0;  // filled slots of the current 32-slot block (wave-uniform, even)
    const int wpos = (p & 1) * 16 + (p >> 1);  // position of pixel p inside a slot row: [k][s]

    auto flush = [&](int count) {
        // A[s] = w[slot = p][pixel 2s+k]: 16 consecutive floats of row p at column k*16
        float A[16];
        const float4 *rowp = reinterpret_cast<const float4 *>(Wt + p * WT_STRIDE + k * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = rowp[t];
            A[4 * t] = v.x; A[4 * t + 1] = v.y; A[4 * t + 2] = v.z; A[4 * t + 3] = v.w;
        }
        f32x16 acc[NBB];
#pragma unroll
        for (int j = 0; j < NBB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (!(dbg & 2)) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int j = 0; j < NBB; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], V[s][j], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
            const int gid = slot_id[slot];
            if (slot < count && gid >= 0 && !(dbg & 1)) {
                float *dst = v_colors + (size_t)gid * d + ch0 + p;
                if (dbg & 4) {  // EXPERIMENT: plain stores instead of atomics
#pragma unroll
                    for (int j = 0; j < NBB; ++j)
                        dst[32 * j] = acc[j][r];  // plain store: wrong sums, measures the store path
                } else {
#pragma unroll
                    for (int j = 0; j < NBB; ++j) atomic_add_f32(dst + 32 * j, acc[j][r]);
                }
            }
        }
    };

    refill(6);
    if (!__all(st.done) && rd < nq) {
        int gid_n;
        float a_n = eval_at(rd, gid_n);
        bool go = true;
        while (go) {
            const float a_c = a_n;
            const int gid_c = gid_n;
            rd += 2;
            if ((nq - rd) < 6 && pending) refill(6);
            const bool more = rd < nq;
            a_n = eval_at(rd, gid_n);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
            bool blended;
            const float wgt = step_pair2(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
            if (__any(wgt != 0.f)) {  // pairs nobody blends would only add zeros: skip their atomics
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


// ------------------------------------------------------------------------------------------
// Staged (atomic-free, deterministic) colours-only backward.
// Float atomics on this multi-XCD part are executed at the memory side (~1.2 TB/s measured,
// TCC_EA0_ATOMIC == TCC_ATOMIC) while plain stores of the same rows are ~free, so the partial
// sums are STORED as rows and reduced by a second pass instead:
//   A  one wave per (tile, 8x4 block): alpha once (not once per channel slice), weights kept as
//      A-operand tiles wt[row][32 px] in global (L2-hot for pass B), row -> Gaussian map, and the
//      first 128-channel slice's rows; row slots were counted by the forward (blk_rows) and
//      prefix-summed, so every row has a fixed address: no atomics, bit-reproducible;
//   B  one wave per (tile, block, slice >= 1): pure streaming -- weight tile, 64 MFMAs, 32 rows;
//   C  rows sorted by Gaussian id (radix sort, 32-bit keys), per-Gaussian offsets;
//   D  v_colors[g] = sum of its rows (written once: no zero-fill of v_colors needed).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 2) void raster_bwd_rows_a(
    int d, int width, int height, int tile_w, int n_tiles, int n_gauss, const GRec *__restrict__ packed,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ v_render_colors, const int32_t *__restrict__ blk_rows,
    const int32_t *__restrict__ row_end /* inclusive cumsum of blk_rows */, float *__restrict__ wt,
    uint32_t *__restrict__ row_key, int32_t *__restrict__ row_idx, float *__restrict__ prow)
{
    __shared__ __attribute__((aligned(16))) HRec ring[RING];
    __shared__ __attribute__((aligned(16))) float Wt[32 * WT_STRIDE];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8);
    const int blk = logical & 7;
    const int tile = gags_tile_of_order(logical >> 3, tile_w, n_tiles / tile_w);
    const int cnt = blk_rows[tile * 8 + blk];
    if (cnt == 0) return;
    const int base = row_end[tile * 8 + blk] - cnt;
    const int lane = threadIdx.x;
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 4;
    const int pj = bx0 + (p & 7), pi = by0 + (p >> 3);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float rx0 = (float)bx0 + 0.5f, rx1 = (float)bx0 + 7.5f, ry0 = (float)by0 + 0.5f, ry1 = (float)by0 + 3.5f;

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];

    float V[16][NBB];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int q = 2 * s + k;
        const int qj = bx0 + (q & 7), qi = by0 + (q >> 3);
        const bool ok = (qi < height) && (qj < width);
        const float *src = v_render_colors + ((size_t)(ok ? qi : 0) * width + (ok ? qj : 0)) * d + p;
#pragma unroll
        for (int j = 0; j < NBB; ++j) V[s][j] = ok ? src[32 * j] : 0.f;
    }

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !inside;

    int nq = 0, rd = 0;
    int c = start;
    GRec pre;
    int pre_gid = 0, pre_c = start;
    auto issue = [&]() {
        pre_c = c;
        const int idx = c + lane;
        if (idx < end) {
            const float4 *src = reinterpret_cast<const float4 *>(packed + idx);
            const float4 u = src[0], v = src[1];
            pre.x = u.x; pre.y = u.y; pre.a = u.z; pre.b = u.w; pre.c = v.x; pre.o = v.y; pre.ex = v.z; pre.ey = v.w;
            pre_gid = flatten_ids[idx];
        } else {
            pre.x = pre.y = 0.f; pre.ex = pre.ey = -1.f; pre.a = pre.b = pre.c = pre.o = 0.f;
        }
        c += 64;
    };
    auto commit = [&]() {
        const bool hit = (pre_c + lane < end) && (pre.x + pre.ex >= rx0) && (pre.x - pre.ex <= rx1) &&
                         (pre.y + pre.ey >= ry0) && (pre.y - pre.ey <= ry1);
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int pos = nq + __popcll(mask & ((1ull << lane) - 1ull));
            HRec h;
            h.x = pre.x; h.y = pre.y; h.a = pre.a; h.b = pre.b; h.c = pre.c; h.o = pre.o;
            h.gid = pre_gid; h.sidx = pre_c + lane;
            ring[pos & (RING - 1)] = h;
        }
        nq += __popcll(mask);
    };
    bool pending = false;
    if (c < end) { issue(); pending = true; }
    auto refill = [&](int low) {
        while ((nq - rd) < low && pending) {
            commit();
            pending = false;
            if (c < end) { issue(); pending = true; }
        }
    };
    auto eval_at = [&](int pos, int &gid) {
        const bool valid = pos + k < nq;
        const HRec h = ring[min(pos + k, nq - 1) & (RING - 1)];
        GRec r;
        r.x = h.x; r.y = h.y; r.a = h.a; r.b = h.b; r.c = h.c; r.o = h.o;
        gid = valid ? h.gid : n_gauss;  // the unused slot of a lone last hit sorts behind every Gaussian
        return eval_alpha(r, px, py, valid);
    };

    int nh = 0;    // slots filled in the current 32-slot tile
    int row0 = base;  // first row of the current tile
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
#pragma unroll
        for (int j = 0; j < NBB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int j = 0; j < NBB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], V[s][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
            if (slot < count) {
                float *dst = prow + (size_t)(row0 + slot) * d + p;
#pragma unroll
                for (int j = 0; j < NBB; ++j) dst[32 * j] = acc[j][r];
            }
        }
        row0 += 32;
    };

    refill(6);
    if (!__all(st.done) && rd < nq) {
        int gid_n;
        float a_n = eval_at(rd, gid_n);
        auto kstep = [&]() -> bool {
            const float a_c = a_n;
            const int gid_c = gid_n;
            rd += 2;
            if ((nq - rd) < 6 && pending) refill(6);
            const bool more = rd < nq;
            a_n = eval_at(rd, gid_n);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
            bool blended;
            const float wgt = step_pair2(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
            if (__any(wgt != 0.f)) {  // same predicate as the forward's row count
                const int row = row0 + nh + k;
                Wt[(nh + k) * WT_STRIDE + wpos] = wgt;
                wt[(size_t)row * 32 + wpos] = wgt;  // A-operand image for pass B: 2 x 128 B per step
                if (p == 0) { row_key[row] = (uint32_t)gid_c; row_idx[row] = row; }
                nh += 2;
                if (nh == 32) { flush(32); nh = 0; }
            }
            return more && !__all(st.done);
        };
        bool go = true;
        while (go) {  // mirrors the forward's two-steps-per-trip loop so the row counts agree
            kstep();
            go = kstep();
        }
    }
    if (nh > 0) flush(nh);
}

__global__ __launch_bounds__(64, 2) void raster_bwd_rows_b(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices_b, const float *__restrict__ v_render_colors,
    const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ row_end, const float *__restrict__ wt,
    float *__restrict__ prow)
{
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8 * n_slices_b);
    const int slice = 1 + logical % n_slices_b, rest = logical / n_slices_b;
    const int blk = rest & 7;
    const int tile = gags_tile_of_order(rest >> 3, tile_w, n_tiles / tile_w);
    const int cnt = blk_rows[tile * 8 + blk];
    if (cnt == 0) return;
    const int base = row_end[tile * 8 + blk] - cnt;
    const int ch0 = slice * CSB;
    const int lane = threadIdx.x;
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 4;

    float V[16][NBB];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int q = 2 * s + k;
        const int qj = bx0 + (q & 7), qi = by0 + (q >> 3);
        const bool ok = (qi < height) && (qj < width);
        const float *src = v_render_colors + ((size_t)(ok ? qi : 0) * width + (ok ? qj : 0)) * d + ch0 + p;
#pragma unroll
        for (int j = 0; j < NBB; ++j) V[s][j] = ok ? src[32 * j] : 0.f;
    }
    // weight tile of M-block m: rows base+32m .. +31, lane (i=p, k) owns 16 consecutive floats
    auto load_a = [&](int m, float4(&a)[4]) {
        const float4 *src = reinterpret_cast<const float4 *>(wt + (size_t)(base + 32 * m + p) * 32 + k * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = src[t];
    };
    const int nblocks = (cnt + 31) >> 5;
    float4 an[4];
    load_a(0, an);
    for (int m = 0; m < nblocks; ++m) {
        float A[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) { A[4 * t] = an[t].x; A[4 * t + 1] = an[t].y; A[4 * t + 2] = an[t].z; A[4 * t + 3] = an[t].w; }
        load_a(min(m + 1, nblocks - 1), an);  // prefetch
        f32x16 acc[NBB];
#pragma unroll
        for (int j = 0; j < NBB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int j = 0; j < NBB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], V[s][j], acc[j], 0, 0, 0);
        const int count = min(32, cnt - 32 * m);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
            if (slot < count) {
                float *dst = prow + (size_t)(base + 32 * m + slot) * d + ch0 + p;
#pragma unroll
                for (int j = 0; j < NBB; ++j) dst[32 * j] = acc[j][r];
            }
        }
    }
}

// seg[g] = first sorted position whose key is >= g, for g in [0, n_keys]; seg[n_keys] = n (keys < n_keys valid)
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

// v_colors[g, :] = sum over the Gaussian's rows (in sorted = deterministic order); float4 per lane
__global__ __launch_bounds__(256) void reduce_rows_kernel(int n_gauss, int d, const int32_t *__restrict__ seg,
                                                          const int32_t *__restrict__ sorted_rows,
                                                          const float *__restrict__ prow, float *__restrict__ v_colors)
{
    const int lpg = d >> 2;                       // lanes per Gaussian
    const int gpb = 256 / lpg;                    // Gaussians per block
    const int gl = threadIdx.x / lpg;
    const int g = blockIdx.x * gpb + gl;
    const int cl = (threadIdx.x % lpg) * 4;
    if (gl >= gpb || g >= n_gauss) return;
    const int b = seg[g], e = seg[g + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = b;
    for (; i + 3 < e; i += 4) {
        const int r0 = sorted_rows[i], r1 = sorted_rows[i + 1], r2 = sorted_rows[i + 2], r3 = sorted_rows[i + 3];
        const float4 v0 = *reinterpret_cast<const float4 *>(prow + (size_t)r0 * d + cl);
        const float4 v1 = *reinterpret_cast<const float4 *>(prow + (size_t)r1 * d + cl);
        const float4 v2 = *reinterpret_cast<const float4 *>(prow + (size_t)r2 * d + cl);
        const float4 v3 = *reinterpret_cast<const float4 *>(prow + (size_t)r3 * d + cl);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; i < e; ++i) {
        const float4 v0 = *reinterpret_cast<const float4 *>(prow + (size_t)sorted_rows[i] * d + cl);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
    }
    *reinterpret_cast<float4 *>(v_colors + (size_t)g * d + cl) = acc;
}


// ------------------------------------------------------------------------------------------
// Staged backward, tile-merged flavour (default).  Same idea as rows A/B, but the partial sums of
// the eight 8x4 blocks of a tile are merged in LDS before they are stored, so a row exists per
// (tile, list position) instead of per (block, hit): ~2.1x fewer row bytes written and re-read.
//   W  one wave per (tile, block): alpha ONCE, no MFMA: weight tiles wt[slot][32 px] + the list
//      position of every slot (slots are in list order);
//   M  one 8-wave workgroup per (tile, 64-channel slice): the tile's list is walked in windows
//      of 64 positions; each wave runs the MFMAs of its block's slots inside the window and adds
//      the result rows into an LDS accumulator [64 positions][64 ch] with ds_add_f32; after a
//      barrier the touched rows are stored (coalesced, plain) at prow[sorted index] and flagged;
//   then: intersections sorted by Gaussian id, per-Gaussian sum of its touched rows.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 4) void raster_bwd_weights(
    int width, int height, int tile_w, int n_tiles, const GRec *__restrict__ packed,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int n_isects,
    const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ row_end, float *__restrict__ wt,
    int32_t *__restrict__ row_pos)
{
    __shared__ __attribute__((aligned(16))) HRec ring[RING];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8);
    const int blk = logical & 7;
    const int tile = gags_tile_of_order(logical >> 3, tile_w, n_tiles / tile_w);
    const int cnt = blk_rows[tile * 8 + blk];
    if (cnt == 0) return;
    const int base = row_end[tile * 8 + blk] - cnt;
    const int lane = threadIdx.x;
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (blk & 1) * 8, by0 = ty * GAGS_TILE + (blk >> 1) * 4;
    const int pj = bx0 + (p & 7), pi = by0 + (p >> 3);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float rx0 = (float)bx0 + 0.5f, rx1 = (float)bx0 + 7.5f, ry0 = (float)by0 + 0.5f, ry1 = (float)by0 + 3.5f;

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !inside;

    int nq = 0, rd = 0;
    int c = start;
    GRec pre;
    int pre_c = start;
    auto issue = [&]() {
        pre_c = c;
        const int idx = c + lane;
        if (idx < end) {
            const float4 *src = reinterpret_cast<const float4 *>(packed + idx);
            const float4 u = src[0], v = src[1];
            pre.x = u.x; pre.y = u.y; pre.a = u.z; pre.b = u.w; pre.c = v.x; pre.o = v.y; pre.ex = v.z; pre.ey = v.w;
        } else {
            pre.x = pre.y = 0.f; pre.ex = pre.ey = -1.f; pre.a = pre.b = pre.c = pre.o = 0.f;
        }
        c += 64;
    };
    auto commit = [&]() {
        const bool hit = (pre_c + lane < end) && (pre.x + pre.ex >= rx0) && (pre.x - pre.ex <= rx1) &&
                         (pre.y + pre.ey >= ry0) && (pre.y - pre.ey <= ry1);
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int pos = nq + __popcll(mask & ((1ull << lane) - 1ull));
            HRec h;
            h.x = pre.x; h.y = pre.y; h.a = pre.a; h.b = pre.b; h.c = pre.c; h.o = pre.o;
            h.gid = 0; h.sidx = pre_c + lane - start;  // position inside the tile's list
            ring[pos & (RING - 1)] = h;
        }
        nq += __popcll(mask);
    };
    bool pending = false;
    if (c < end) { issue(); pending = true; }
    auto refill = [&](int low) {
        while ((nq - rd) < low && pending) {
            commit();
            pending = false;
            if (c < end) { issue(); pending = true; }
        }
    };
    // alpha of this lane's slot of the pair at `pos`; a lone last hit's partner slot inherits its position
    // (its weights are all zero, so it adds nothing to that row)
    auto eval_at = [&](int pos, int &lpos) {
        const bool valid = pos + k < nq;
        const HRec h = ring[min(pos + k, nq - 1) & (RING - 1)];
        GRec r;
        r.x = h.x; r.y = h.y; r.a = h.a; r.b = h.b; r.c = h.c; r.o = h.o;
        lpos = h.sidx;
        return eval_alpha(r, px, py, valid);
    };

    int row = base;
    const int wpos = (p & 1) * 16 + (p >> 1);
    refill(6);
    if (!__all(st.done) && rd < nq) {
        int pos_n;
        float a_n = eval_at(rd, pos_n);
        auto kstep = [&]() -> bool {
            const float a_c = a_n;
            const int pos_c = pos_n;
            rd += 2;
            if ((nq - rd) < 6 && pending) refill(6);
            const bool more = rd < nq;
            a_n = eval_at(rd, pos_n);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
            bool blended;
            const float wgt = step_pair2(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
            if (__any(wgt != 0.f)) {  // same predicate as the forward's row count
                wt[(size_t)(row + k) * 32 + wpos] = wgt;  // A-operand image: 2 x 128 B per step
                if (p == 0) row_pos[row + k] = pos_c;
                row += 2;
            }
            return more && !__all(st.done);
        };
        bool go = true;
        while (go) {  // mirrors the forward's two-steps-per-trip loop so the row counts agree
            kstep();
            go = kstep();
        }
    }
}

template <int NBM>
__global__ __launch_bounds__(512, 4) void raster_bwd_merge(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const float *__restrict__ v_render_colors,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ row_end,
    const float *__restrict__ wt, const int32_t *__restrict__ row_pos, float *__restrict__ prow,
    uint8_t *__restrict__ touched)
{
    constexpr int CSM = 32 * NBM;  // channels per workgroup
    constexpr int WIN = 64;        // list positions per window
    __shared__ __attribute__((aligned(16))) float accum[WIN][CSM];
    __shared__ int posb[8][64];
    __shared__ unsigned long long maskw[2];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int slice = logical % n_slices;
    const int tile = gags_tile_of_order(logical / n_slices, tile_w, n_tiles / tile_w);
    const int ch0 = slice * CSM;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, k = lane >> 5;
    const int cnt = blk_rows[tile * 8 + w];
    const int base = row_end[tile * 8 + w] - cnt;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (w & 1) * 8, by0 = ty * GAGS_TILE + (w >> 1) * 4;
    const int tstart = offsets[tile];

    // cotangent slab of this wave's 8x4 block as B operands: V[s][j] = v_out[pixel 2s+k][ch0 + 32j + p]
    float V[16][NBM];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int q = 2 * s + k;
        const int qj = bx0 + (q & 7), qi = by0 + (q >> 3);
        const bool ok = (qi < height) && (qj < width);
        const float *src = v_render_colors + ((size_t)(ok ? qi : 0) * width + (ok ? qj : 0)) * d + ch0 + p;
#pragma unroll
        for (int j = 0; j < NBM; ++j) V[s][j] = ok ? src[32 * j] : 0.f;
    }
    for (int i = threadIdx.x; i < WIN * CSM; i += 512) (&accum[0][0])[i] = 0.f;
    if (threadIdx.x < 2) maskw[threadIdx.x] = 0ull;

    int cursor = 0;
    for (int w0 = 0, it = 0;; w0 += WIN, ++it) {
        if (!__syncthreads_or(cursor < cnt)) break;  // also: previous window's stores / zeroing are done
        const int remaining = cnt - cursor;
        const int mypos = (lane < remaining) ? row_pos[base + cursor + lane] : 0x7fffffff;
        const bool inwin = mypos < w0 + WIN;  // slots are in list order: the in-window ones are a prefix
        const int n_in = __popcll(__ballot(inwin));
        posb[w][lane] = mypos - w0;
        if (n_in > 0) {
            // 64-bit OR over the wave of the window bits this block touches
            unsigned long long bits = inwin ? (1ull << (mypos - w0)) : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) bits |= __shfl_xor(bits, o, 64);
            if (lane == 0) atomicOr(&maskw[it & 1], bits);
        }
        for (int mb = 0; mb * 32 < n_in; ++mb) {
            const int nb = min(32, n_in - 32 * mb);
            float A[16];
            {
                const float4 *src = reinterpret_cast<const float4 *>(wt + (size_t)(base + cursor + 32 * mb + p) * 32 + k * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 v = (p < nb) ? src[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                    A[4 * t] = v.x; A[4 * t + 1] = v.y; A[4 * t + 2] = v.z; A[4 * t + 3] = v.w;
                }
            }
            f32x16 acc[NBM];
#pragma unroll
            for (int j = 0; j < NBM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int j = 0; j < NBM; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], V[s][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
                if (slot < nb) {
                    float *dst = &accum[posb[w][32 * mb + slot]][p];
#pragma unroll
                    for (int j = 0; j < NBM; ++j) atomicAdd(dst + 32 * j, acc[j][r]);  // ds_add_f32
                }
            }
        }
        cursor += n_in;
        __syncthreads();
        // ---- store the touched rows of this window, clear them for the next one ----
        const unsigned long long mask = maskw[it & 1];
        constexpr int F4_PER_ROW = CSM / 4;
#pragma unroll
        for (int i = threadIdx.x; i < WIN * F4_PER_ROW; i += 512) {
            const int row = i / F4_PER_ROW, c4 = i - row * F4_PER_ROW;
            if ((mask >> row) & 1ull) {
                float4 *src = reinterpret_cast<float4 *>(&accum[row][4 * c4]);
                *reinterpret_cast<float4 *>(prow + (size_t)(tstart + w0 + row) * d + ch0 + 4 * c4) = *src;
                *src = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (slice == 0 && threadIdx.x < WIN && ((mask >> threadIdx.x) & 1ull)) touched[tstart + w0 + threadIdx.x] = 1;
        if (threadIdx.x == 0) maskw[(it + 1) & 1] = 0ull;  // nobody touches the other parity until the next barrier
    }
}

__global__ __launch_bounds__(256) void iota_keys_kernel(int n, const int32_t *__restrict__ flatten_ids,
                                                        uint32_t *__restrict__ keys, int32_t *__restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { keys[i] = (uint32_t)flatten_ids[i]; idx[i] = i; }
}

// v_colors[g, :] = sum over the Gaussian's touched rows (sorted = deterministic order); float4 per lane
__global__ __launch_bounds__(256) void reduce_merged_kernel(int n_gauss, int d, const int32_t *__restrict__ seg,
                                                            const int32_t *__restrict__ sorted_pos,
                                                            const uint8_t *__restrict__ touched,
                                                            const float *__restrict__ prow,
                                                            float *__restrict__ v_colors)
{
    const int lpg = d >> 2;
    const int gpb = 256 / lpg;
    const int gl = threadIdx.x / lpg;
    const int g = blockIdx.x * gpb + gl;
    const int cl = (threadIdx.x % lpg) * 4;
    if (gl >= gpb || g >= n_gauss) return;
    const int b = seg[g], e = seg[g + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = b; i < e; i += 4) {
        int sp[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = i + u < e;
            sp[u] = ok[u] ? sorted_pos[i + u] : 0;
            ok[u] = ok[u] && touched[sp[u]];
        }
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = ok[u] ? *reinterpret_cast<const float4 *>(prow + (size_t)sp[u] * d + cl) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    *reinterpret_cast<float4 *>(v_colors + (size_t)g * d + cl) = acc;
}

int launch_bwd_colors_mfma(int d, int width, int height, const GRec *packed, const int32_t *offsets,
                           const int32_t *flat, int n_isects, const float *v_out, float *v_colors, int dbg,
                           hipStream_t st)
{
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / CSB;
    hipLaunchKernelGGL(raster_bwd_colors_mfma, dim3(n_tiles * 8 * n_slices), dim3(64), 0, st, d, width, height, tile_w,
                       n_tiles, n_slices, packed, offsets, flat, n_isects, v_out, v_colors, dbg);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

}  // namespace

int gags_pack_isects_launch(int n_isects, const int32_t *flat, const float *means2d, const float *conics,
                            const float *opacities, void *packed, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (n_isects <= 0) return GAGS_OK;
    hipLaunchKernelGGL(pack_isects_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, flat, means2d,
                       conics, opacities, reinterpret_cast<GRec *>(packed));
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// Returns GAGS_OK when the MFMA path took the call, 1 when d is not eligible (caller falls
// back to the VALU kernels), negative on error.
int gags_raster_fwd_mfma(int d, int width, int height, const void *packed, const float *colors,
                         const float *backgrounds, const int32_t *offsets, const int32_t *flat, int n_isects,
                         float *out, float *alphas, int32_t *last_ids, int32_t *blk_rows, int dbg, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    const GRec *pk = reinterpret_cast<const GRec *>(packed);
#define ARGS d, width, height, pk, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids, blk_rows, dbg, st
    if (d < 32 || d % 32 != 0) return 1;
    if (d % 256 == 0) return launch_fwd_mfma<8>(ARGS);
    if (d % 128 == 0) return launch_fwd_mfma<4>(ARGS);
    if (d % 64 == 0) return launch_fwd_mfma<2>(ARGS);
    return launch_fwd_mfma<1>(ARGS);
#undef ARGS
}

// colours-only backward on the matrix cores; 1 = width not eligible (d % 128 != 0)
int gags_raster_bwd_colors_mfma(int d, int width, int height, const void *packed, const int32_t *offsets,
                                const int32_t *flat, int n_isects, const float *v_out, float *v_colors, int dbg,
                                hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (d < CSB || d % CSB != 0) return 1;
    return launch_bwd_colors_mfma(d, width, height, reinterpret_cast<const GRec *>(packed), offsets, flat, n_isects,
                                  v_out, v_colors, dbg, st);
}

int64_t gags_sort_u32_scratch_bytes(int64_t n);
int gags_sort_pairs_u32(int64_t n, int nbits, const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out,
                        int32_t *vals_out, void *scratch, int64_t scratch_bytes, hipStream_t st);

namespace {
struct StagedLayout {
    int64_t wt, key, idx, key_s, idx_s, seg, sort, prow, pos, touched, total;
};
inline int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }
// merged = 1: rows per sorted intersection (n_isects of them); 0: rows per (block, hit) slot
inline StagedLayout staged_layout(int64_t rows, int64_t n_isects, int n_gauss, int d, bool merged)
{
    StagedLayout L;
    const int64_t nsort = merged ? (n_isects > 0 ? n_isects : 1) : rows;
    int64_t o = 0;
    L.wt = o; o += al256((rows + 64) * 128);          // A-operand tiles, 128 B per slot (+ slack for the prefetch)
    L.pos = o; o += al256((rows + 64) * 4);
    L.key = o; o += al256(nsort * 4);
    L.idx = o; o += al256(nsort * 4);
    L.key_s = o; o += al256(nsort * 4);
    L.idx_s = o; o += al256(nsort * 4);
    L.seg = o; o += al256(((int64_t)n_gauss + 2) * 4);
    L.sort = o; o += al256(gags_sort_u32_scratch_bytes(nsort));
    L.touched = o; o += al256(nsort);
    L.prow = o; o += al256(nsort * (int64_t)d * 4);
    L.total = o;
    return L;
}
}  // namespace

int64_t gags_bwd_staged_scratch_bytes_impl(int64_t rows, int64_t n_isects, int n_gauss, int d)
{
    const int64_t a = staged_layout(rows > 0 ? rows : 1, n_isects, n_gauss, d, true).total;
    const int64_t b = staged_layout(rows > 0 ? rows : 1, n_isects, n_gauss, d, false).total;
    return a > b ? a : b;
}

// 1 = not eligible (d % 128 != 0)
int gags_raster_bwd_colors_staged(int d, int width, int height, int n_gauss, const void *packed,
                                  const int32_t *offsets, const int32_t *flat, int n_isects, const float *v_out,
                                  const int32_t *blk_rows, const int32_t *row_end, int64_t rows, void *scratch,
                                  int64_t scratch_bytes, float *v_colors, int stage_arg, hipStream_t st)
{
    // stage (low 4 bits): 0 = everything; 1 = weights / rows A, 2 = merge / rows B, 3 = sort + segment
    // offsets, 4 = reduce (per-kernel timing).  bit 4 set = per-(block,hit) rows instead of tile-merged rows.
    GAGS_CLEAR_ERR();
    if (d < CSB || d % CSB != 0 || d > 1024) return 1;
    const int stage = stage_arg & 15;
    const bool merged = !(stage_arg & 16);
    const bool sA = stage == 0 || stage == 1, sB = stage == 0 || stage == 2, sS = stage == 0 || stage == 3,
               sR = stage == 0 || stage == 4;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / CSB;
    const StagedLayout L = staged_layout(rows > 0 ? rows : 1, n_isects, n_gauss, d, merged);
    if (scratch_bytes < L.total) return GAGS_ESCRATCH;
    char *sb = (char *)scratch;
    float *wt = (float *)(sb + L.wt);
    uint32_t *key = (uint32_t *)(sb + L.key), *key_s = (uint32_t *)(sb + L.key_s);
    int32_t *idx = (int32_t *)(sb + L.idx), *idx_s = (int32_t *)(sb + L.idx_s), *seg = (int32_t *)(sb + L.seg);
    int32_t *pos = (int32_t *)(sb + L.pos);
    uint8_t *touched = (uint8_t *)(sb + L.touched);
    float *prow = (float *)(sb + L.prow);
    const GRec *pk = reinterpret_cast<const GRec *>(packed);
    int nbits = 1;
    while ((1ll << nbits) <= n_gauss) ++nbits;  // keys in [0, n_gauss]
    const int gpb = 256 / (d >> 2);
    if (merged) {
        constexpr int NBM = 2;
        const int n_slices_m = d / (32 * NBM);
        if (rows > 0 && n_isects > 0) {
            if (sA) {
                if (hipMemsetAsync(touched, 0, (size_t)n_isects, st) != hipSuccess) return GAGS_ELAUNCH;
                hipLaunchKernelGGL(raster_bwd_weights, dim3(n_tiles * 8), dim3(64), 0, st, width, height, tile_w, n_tiles,
                                   pk, offsets, flat, n_isects, blk_rows, row_end, wt, pos);
            }
            if (sB)
                hipLaunchKernelGGL(raster_bwd_merge<NBM>, dim3(n_tiles * n_slices_m), dim3(512), 0, st, d, width, height,
                                   tile_w, n_tiles, n_slices_m, v_out, offsets, blk_rows, row_end, wt, pos, prow,
                                   touched);
            if (sS) {
                hipLaunchKernelGGL(iota_keys_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, flat, key,
                                   idx);
                const int rc = gags_sort_pairs_u32(n_isects, nbits, key, idx, key_s, idx_s, sb + L.sort,
                                                   L.touched - L.sort, st);
                if (rc != GAGS_OK) return rc;
                hipLaunchKernelGGL(seg_offsets_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, key_s,
                                   n_gauss, seg);
            }
        } else if (sS) {
            hipLaunchKernelGGL(seg_fill_kernel, dim3((n_gauss + 1 + 255) / 256), dim3(256), 0, st, n_gauss, seg);
        }
        if (sR)
            hipLaunchKernelGGL(reduce_merged_kernel, dim3((n_gauss + gpb - 1) / gpb), dim3(256), 0, st, n_gauss, d, seg,
                               idx_s, touched, prow, v_colors);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (rows > 0) {
        if (sA)
            hipLaunchKernelGGL(raster_bwd_rows_a, dim3(n_tiles * 8), dim3(64), 0, st, d, width, height, tile_w, n_tiles,
                               n_gauss, pk, offsets, flat, n_isects, v_out, blk_rows, row_end, wt, key, idx, prow);
        if (sB && n_slices > 1)
            hipLaunchKernelGGL(raster_bwd_rows_b, dim3(n_tiles * 8 * (n_slices - 1)), dim3(64), 0, st, d, width, height,
                               tile_w, n_tiles, n_slices - 1, v_out, blk_rows, row_end, wt, prow);
        if (sS) {
            const int rc = gags_sort_pairs_u32(rows, nbits, key, idx, key_s, idx_s, sb + L.sort, L.touched - L.sort, st);
            if (rc != GAGS_OK) return rc;
            hipLaunchKernelGGL(seg_offsets_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (int)rows,
                               key_s, n_gauss, seg);
        }
    } else if (sS) {
        hipLaunchKernelGGL(seg_fill_kernel, dim3((n_gauss + 1 + 255) / 256), dim3(256), 0, st, n_gauss, seg);
    }
    if (sR)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((n_gauss + gpb - 1) / gpb), dim3(256), 0, st, n_gauss, d, seg, idx_s,
                           prow, v_colors);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
