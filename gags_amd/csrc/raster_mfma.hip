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

constexpr int GC = 64;  // Gaussians per LDS chunk

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

// Advance one pixel over the two Gaussians of a K-step; returns the weight of slot k.
__device__ __forceinline__ float step_pair(PixState &s, float a0, float a1, int idx0, int idx1, int k)
{
    float w0 = 0.f, w1 = 0.f;
    if (!s.done && a0 > 0.f) {
        const float nt = s.T * (1.0f - a0);
        if (nt <= GAGS_T_STOP) s.done = true;
        else { w0 = a0 * s.T; s.T = nt; s.cur = idx0; }
    }
    if (!s.done && a1 > 0.f) {
        const float nt = s.T * (1.0f - a1);
        if (nt <= GAGS_T_STOP) s.done = true;
        else { w1 = a1 * s.T; s.T = nt; s.cur = idx1; }
    }
    return k ? w1 : w0;
}

template <int NB>
__global__ __launch_bounds__(512) void raster_fwd_mfma(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const float *__restrict__ means2d,
    const float *__restrict__ conics, const float *__restrict__ opacities, const float *__restrict__ colors,
    const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids)
{
    constexpr int CS = 32 * NB;               // channels per workgroup
    constexpr int VEC = NB >= 4 ? 4 : NB;     // floats per B-operand LDS read
    constexpr int NG = NB / VEC;              // accumulator groups
    static_assert(NB == 1 || NB == 2 || NB % 4 == 0, "NB in {1,2,4,8,...}");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *F = reinterpret_cast<float *>(smem);                       // [GC][CS]
    GRec *P = reinterpret_cast<GRec *>(smem + GC * CS * 4);           // [GC]
    float *Tb = reinterpret_cast<float *>(smem + GC * CS * 4 + GC * 32);  // [8][32]

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int tile = logical / n_slices, slice = logical - tile * n_slices;
    const int ch0 = slice * CS;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (w & 1) * 8, by0 = ty * GAGS_TILE + (w >> 1) * 4;  // wave's 8x4 block
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
    bool wave_done = __all(st.done);

    for (int c0 = start; c0 < end; c0 += GC) {
        if (__syncthreads_count(wave_done) >= 512) break;  // also fences LDS reuse
        const int nc = min(GC, end - c0);
        // ---- stage the chunk: records by the first GC threads, feature rows by all 8 waves ----
        if (threadIdx.x < GC) {
            GRec r;
            if ((int)threadIdx.x < nc) r = load_grec(means2d, conics, opacities, flatten_ids[c0 + threadIdx.x]);
            else { r.x = r.y = r.a = r.b = r.c = r.o = 0.f; r.ex = r.ey = -1.f; }
            P[threadIdx.x] = r;
        }
        {
            constexpr int LPR = CS / 4;           // lanes per row (float4 each)
            constexpr int RPP = 512 / LPR;        // rows per pass
            const int rl = threadIdx.x / LPR, cl = (threadIdx.x - rl * LPR) * 4;
#pragma unroll
            for (int r0 = 0; r0 < GC; r0 += RPP) {
                const int row = r0 + rl;
                if (row < nc) {
                    const int g = flatten_ids[c0 + row];
                    const float4 v = *reinterpret_cast<const float4 *>(colors + (size_t)g * d + ch0 + cl);
                    *reinterpret_cast<float4 *>(F + row * CS + cl) = v;
                }
            }
        }
        __syncthreads();
        if (wave_done) continue;

        // ---- compact: which Gaussians of the chunk can touch this wave's 8x4 pixels ----
        bool hit = false;
        if (lane < nc) {
            const GRec r = P[lane];
            hit = (r.x + r.ex >= rx0) && (r.x - r.ex <= rx1) && (r.y + r.ey >= ry0) && (r.y - r.ey <= ry1);
        }
        unsigned long long mask = __ballot(hit);

        while (mask) {
            const int i0 = __builtin_ctzll(mask);
            mask &= mask - 1;
            int i1 = -1;
            if (mask) { i1 = __builtin_ctzll(mask); mask &= mask - 1; }
            const int mine = k ? (i1 < 0 ? i0 : i1) : i0;
            const GRec r = P[mine];
            const float a_own = eval_alpha(r, px, py, k ? (i1 >= 0) : true);
            // lanes 32-63 of vdst <-> lanes 0-31 of src: r0 = slot-0 alpha, r1 = slot-1 alpha, in every lane
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_own), __float_as_uint(a_own), false, false);
            const float a0 = __uint_as_float(sw[0]), a1 = __uint_as_float(sw[1]);
            const float wgt = step_pair(st, a0, a1, c0 + i0, c0 + i1, k);
            if (__any(wgt != 0.f)) {
                const float *frow = F + mine * CS + VEC * p;
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    float bv[VEC];
                    if constexpr (VEC == 4) {
                        const float4 t = *reinterpret_cast<const float4 *>(frow + gq * 128);
                        bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
                    } else if constexpr (VEC == 2) {
                        const float2 t = *reinterpret_cast<const float2 *>(frow);
                        bv[0] = t.x; bv[1] = t.y;
                    } else {
                        bv[0] = frow[0];
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        acc[gq * VEC + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wgt, bv[i], acc[gq * VEC + i], 0, 0, 0);
                }
            }
            if (__all(st.done)) { wave_done = true; break; }
        }
    }

    // ---- epilogue: out[pix][ch] = acc (+ T*bg); accumulator row r <-> pixel (r&3)+8(r>>2)+4k ----
    if (k == 0) Tb[w * 32 + p] = st.T;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
    if (k == 0 && inside && slice == 0) {
        const size_t pix = (size_t)pi * width + pj;
        render_alphas[pix] = 1.0f - st.T;
        last_ids[pix] = st.cur;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * k;  // pixel index inside the wave's block
        const int qj = bx0 + (q & 7), qi = by0 + (q >> 3);
        if (qi >= height || qj >= width) continue;
        const float Tq = Tb[w * 32 + q];
        float *o = render_colors + ((size_t)qi * width + qj) * d + ch0 + VEC * p;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int ch = ch0 + gq * 32 * VEC + VEC * p + i;
                v[i] = backgrounds ? __builtin_fmaf(Tq, backgrounds[ch], acc[gq * VEC + i][r]) : acc[gq * VEC + i][r];
            }
            if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o + gq * 128) = make_float4(v[0], v[1], v[2], v[3]);
            else if constexpr (VEC == 2) *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
            else o[0] = v[0];
        }
    }
}

template <int NB>
int launch_fwd_mfma(int d, int width, int height, const float *means2d, const float *conics, const float *opacities,
                    const float *colors, const float *backgrounds, const int32_t *offsets, const int32_t *flat,
                    int n_isects, float *out, float *alphas, int32_t *last_ids, hipStream_t st)
{
    constexpr int CS = 32 * NB;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / CS;
    const size_t lds = (size_t)GC * CS * 4 + GC * 32 + 8 * 32 * 4;
    static bool attr_done = false;  // idempotent; a benign race at worst sets it twice
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&raster_fwd_mfma<NB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GAGS_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(raster_fwd_mfma<NB>, dim3(n_tiles * n_slices), dim3(512), lds, st, d, width, height, tile_w,
                       n_tiles, n_slices, means2d, conics, opacities, colors, backgrounds, offsets, flat, n_isects, out,
                       alphas, last_ids);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}


// ------------------------------------------------------------------------------------------
// K10 colours-only backward (the GAD flow: only d loss / d colors is consumed,
// scene/gaussian_model.py:192-208):   v_colors[g, :] += sum_px w[px, g] * v_out[px, :]
// with w = alpha*T recomputed FRONT TO BACK by exactly the forward's arithmetic (same hits,
// same stop decisions, so neither render_alphas nor last_ids nor the features are read).
// Per wave (8x4 pixels x CSB = 128 channels of the slice):
//   - the cotangent slab v_out[32 px][128 ch] lives in 64 VGPRs as MFMA B operands
//     (K = pixel pairs, N = channels), loaded once;
//   - hits are evaluated two per step in the forward's lane layout (pixel, slot) and the
//     weights are transposed through a wave-private LDS tile Wt[32 slots][32 px] (row stride 36
//     dwords: conflict-free ds_write_b32 / ds_read_b128);
//   - every 32 hits: A = Wt^T fragments, 16 K-steps x 4 channel tiles of
//     v_mfma_f32_32x32x2_f32, then one 128-B coalesced float atomic per (Gaussian, channel tile).
// ------------------------------------------------------------------------------------------
constexpr int NBB = 4;            // channel tiles per wave in the backward
constexpr int CSB = 32 * NBB;     // 128 channels per workgroup
constexpr int WT_STRIDE = 36;     // dwords per slot row of the transpose tile

__device__ __forceinline__ void atomic_add_f32(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(512) void raster_bwd_colors_mfma(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const float *__restrict__ means2d,
    const float *__restrict__ conics, const float *__restrict__ opacities, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, const float *__restrict__ v_render_colors,
    float *__restrict__ v_colors)
{
    __shared__ __attribute__((aligned(16))) GRec P[GC];
    __shared__ int32_t ids[GC];
    __shared__ __attribute__((aligned(16))) float Wt_all[8][32 * WT_STRIDE];
    __shared__ int32_t slot_id_all[8][32];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int tile = logical / n_slices, slice = logical - tile * n_slices;
    const int ch0 = slice * CSB;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, k = lane >> 5;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    const int bx0 = tx * GAGS_TILE + (w & 1) * 8, by0 = ty * GAGS_TILE + (w >> 1) * 4;
    const int pj = bx0 + (p & 7), pi = by0 + (p >> 3);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float rx0 = (float)bx0 + 0.5f, rx1 = (float)bx0 + 7.5f, ry0 = (float)by0 + 0.5f, ry1 = (float)by0 + 3.5f;
    float *Wt = Wt_all[w];
    int32_t *slot_id = slot_id_all[w];

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
    bool wave_done = __all(st.done);
    int nh = 0;  // filled slots of the current 32-hit block (wave-uniform)
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
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int j = 0; j < NBB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], V[s][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * k;
            if (slot < count) {
                float *dst = v_colors + (size_t)slot_id[slot] * d + ch0 + p;
#pragma unroll
                for (int j = 0; j < NBB; ++j) atomic_add_f32(dst + 32 * j, acc[j][r]);
            }
        }
    };

    for (int c0 = start; c0 < end; c0 += GC) {
        if (__syncthreads_count(wave_done) >= 512) break;
        const int nc = min(GC, end - c0);
        if (threadIdx.x < GC) {
            GRec r;
            int g = 0;
            if ((int)threadIdx.x < nc) { g = flatten_ids[c0 + threadIdx.x]; r = load_grec(means2d, conics, opacities, g); }
            else { r.x = r.y = r.a = r.b = r.c = r.o = 0.f; r.ex = r.ey = -1.f; }
            P[threadIdx.x] = r;
            ids[threadIdx.x] = g;
        }
        __syncthreads();
        if (wave_done) continue;

        bool hit = false;
        if (lane < nc) {
            const GRec r = P[lane];
            hit = (r.x + r.ex >= rx0) && (r.x - r.ex <= rx1) && (r.y + r.ey >= ry0) && (r.y - r.ey <= ry1);
        }
        unsigned long long mask = __ballot(hit);

        while (mask) {
            if (nh > 30) { flush(nh); nh = 0; }
            const int i0 = __builtin_ctzll(mask);
            mask &= mask - 1;
            int i1 = -1;
            if (mask) { i1 = __builtin_ctzll(mask); mask &= mask - 1; }
            const int mine = k ? (i1 < 0 ? i0 : i1) : i0;
            const GRec r = P[mine];
            const float a_own = eval_alpha(r, px, py, k ? (i1 >= 0) : true);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_own), __float_as_uint(a_own), false, false);
            const float a0 = __uint_as_float(sw[0]), a1 = __uint_as_float(sw[1]);
            const float wgt = step_pair(st, a0, a1, c0 + i0, c0 + i1, k);
            if (__any(wgt != 0.f)) {
                // slots nh (k=0 lanes) and nh+1 (k=1 lanes); a lone hit leaves slot nh+1 unused
                if (k == 0 || i1 >= 0) Wt[(nh + k) * WT_STRIDE + wpos] = wgt;
                if (lane == 0) slot_id[nh] = ids[i0];
                if (lane == 32 && i1 >= 0) slot_id[nh + 1] = ids[i1];
                nh += (i1 >= 0) ? 2 : 1;
            }
            if (__all(st.done)) { wave_done = true; break; }
        }
    }
    if (nh > 0) flush(nh);
}

int launch_bwd_colors_mfma(int d, int width, int height, const float *means2d, const float *conics,
                           const float *opacities, const int32_t *offsets, const int32_t *flat, int n_isects,
                           const float *v_out, float *v_colors, hipStream_t st)
{
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / CSB;
    hipLaunchKernelGGL(raster_bwd_colors_mfma, dim3(n_tiles * n_slices), dim3(512), 0, st, d, width, height, tile_w,
                       n_tiles, n_slices, means2d, conics, opacities, offsets, flat, n_isects, v_out, v_colors);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

}  // namespace

// Returns GAGS_OK when the MFMA path took the call, 1 when d is not eligible (caller falls
// back to the VALU kernels), negative on error.
int gags_raster_fwd_mfma(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, float *out, float *alphas,
                         int32_t *last_ids, hipStream_t st)
{
    GAGS_CLEAR_ERR();
#define ARGS d, width, height, means2d, conics, opacities, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids, st
    if (d < 32 || d % 32 != 0) return 1;
    if (d % 256 == 0) return launch_fwd_mfma<8>(ARGS);
    if (d % 128 == 0) return launch_fwd_mfma<4>(ARGS);
    if (d % 64 == 0) return launch_fwd_mfma<2>(ARGS);
    return launch_fwd_mfma<1>(ARGS);
#undef ARGS
}

// colours-only backward on the matrix cores; 1 = width not eligible (d % 128 != 0)
int gags_raster_bwd_colors_mfma(int d, int width, int height, const float *means2d, const float *conics,
                                const float *opacities, const int32_t *offsets, const int32_t *flat, int n_isects,
                                const float *v_out, float *v_colors, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (d < CSB || d % CSB != 0) return 1;
    return launch_bwd_colors_mfma(d, width, height, means2d, conics, opacities, offsets, flat, n_isects, v_out,
                                  v_colors, st);
}
