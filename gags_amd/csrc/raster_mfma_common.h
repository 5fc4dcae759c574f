// Shared pieces of the matrix-core (wide-D) raster kernels.
//
// Pixel decomposition: a 16x16 tile = eight 8x4 pixel BLOCKS; one wave owns one block.
// Lane l = (pixel p = l & 31, slot k = l >> 5): the MFMA A-operand layout of
// v_mfma_f32_32x32x2_f32 (rows = pixels, K = the two Gaussians of a step).
//
// Per wave, the tile's depth-sorted range is consumed through a HitStream:
//   produce : 64 packed records per pass (register-prefetched one pass ahead), conservative
//             extent test against the block's pixel rectangle, ballot, compaction of the
//             hits into a private LDS ring;
//   consume : two hits per K-step; alpha is evaluated one step ahead of the transmittance
//             chain (software pipeline), one v_permlane32_swap gives every lane both alphas,
//             and the branch-free chain runs redundantly in both half-waves.
#pragma once
#include "common.h"

namespace gags_mfma {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32-byte record per sorted intersection (gags_pack_isects): position, conic, opacity and the
// conservative half-extent of the alpha >= 1/255 footprint.
struct GRec {
    float x, y, a, b, c, o, ex, ey;
};

// record kept in the LDS ring after the extent test: extents replaced by ids
struct HRec {
    float x, y, a, b, c, o;
    int gid, sidx;  // Gaussian id, sorted intersection index
};

constexpr int RING = 128;      // per-wave ring of compacted hits (4 KB; at most 5 + 64 are ever queued)
constexpr int WT_STRIDE = 36;  // dwords per slot row of an LDS weight tile (conflict-free b32 write / b128 read)

__device__ __forceinline__ GRec make_grec_from(float x, float y, float ca, float cb, float cc, float o)
{
    GRec r;
    r.x = x; r.y = y;
    r.a = ca; r.b = cb; r.c = cc;
    r.o = o;
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

__device__ __forceinline__ GRec make_grec(const float *__restrict__ means2d, const float *__restrict__ conics,
                                          const float *__restrict__ opacities, int g)
{
    const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
    return make_grec_from(m.x, m.y, conics[3 * g], conics[3 * g + 1], conics[3 * g + 2], opacities[g]);
}

// Per-pixel compositing state, replicated in both half-waves (lane p and lane p+32).
struct PixState {
    float T;
    int cur;
    bool done;
};

// alpha of a Gaussian at a pixel (0 when skipped by the A8 rule: sigma < 0 or alpha < 1/255)
__device__ __forceinline__ float eval_alpha(const HRec &r, float px, float py, bool live)
{
    const float dx = r.x - px, dy = r.y - py;
    const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
    const float alpha = fminf(GAGS_ALPHA_MAX, r.o * gags_exp_neg(sigma));
    return (live && !(sigma < 0.f || alpha < GAGS_ALPHA_MIN)) ? alpha : 0.f;
}

// Advance one pixel over the two Gaussians of a K-step (branch-free: selects only).
// Returns the weight alpha*T of this lane's slot k; `blended` = this lane's slot was composited.
__device__ __forceinline__ float step_pair(PixState &s, float a0, float a1, int k, bool &blended)
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

// Geometry of the wave's 8x4 pixel block inside its tile.
struct BlockGeom {
    int bx0, by0, pj, pi, p, k;
    bool inside;
    float px, py, rx0, rx1, ry0, ry1;
    __device__ __forceinline__ void init(int tile, int blk, int tile_w, int width, int height, int lane)
    {
        p = lane & 31; k = lane >> 5;
        const int ty = tile / tile_w, tx = tile - ty * tile_w;
        bx0 = tx * GAGS_TILE + (blk & 1) * 8;
        by0 = ty * GAGS_TILE + (blk >> 1) * 4;
        pj = bx0 + (p & 7); pi = by0 + (p >> 3);
        inside = (pi < height) && (pj < width);
        px = (float)pj + 0.5f; py = (float)pi + 0.5f;
        rx0 = (float)bx0 + 0.5f; rx1 = (float)bx0 + 7.5f; ry0 = (float)by0 + 0.5f; ry1 = (float)by0 + 3.5f;
    }
};

// Geometry of a wave's 8x8 pixel block (the split forward / staged backward): a tile = four such blocks.
// Lane (p, k) owns TWO pixels of the block: "A" = pixel p of its upper 8x4 half, "B" = pixel p of its
// lower half (row + 4); they are the two 32-row MFMA blocks of the wave.
struct BlockGeom64 {
    int bx0, by0, pj, piA, piB, p, k;
    bool insideA, insideB;
    float px, pyA, pyB, rx0, rx1, ry0, ry1;
    __device__ __forceinline__ void init(int tile, int blk, int tile_w, int width, int height, int lane)
    {
        p = lane & 31; k = lane >> 5;
        const int ty = tile / tile_w, tx = tile - ty * tile_w;
        bx0 = tx * GAGS_TILE + (blk & 1) * 8;
        by0 = ty * GAGS_TILE + (blk >> 1) * 8;
        pj = bx0 + (p & 7); piA = by0 + (p >> 3); piB = piA + 4;
        insideA = (piA < height) && (pj < width);
        insideB = (piB < height) && (pj < width);
        px = (float)pj + 0.5f; pyA = (float)piA + 0.5f; pyB = (float)piB + 0.5f;
        rx0 = (float)bx0 + 0.5f; rx1 = (float)bx0 + 7.5f; ry0 = (float)by0 + 0.5f; ry1 = (float)by0 + 7.5f;
    }
};

// Producer half of the per-wave pipeline (see the header comment).
struct HitStream {
    HRec *ring;
    const GRec *packed;
    const int32_t *flat;
    int start, end, lane;
    float rx0, rx1, ry0, ry1;
    int nq, rd, c;   // hits produced / consumed, next chunk start (all wave-uniform)
    GRec pre;        // chunk in flight: one record per lane
    int pre_gid, pre_c;
    bool pending;
    bool by_gauss = false;  // records indexed by Gaussian id (gathered here) instead of by sorted intersection

    __device__ __forceinline__ void issue()
    {
        pre_c = c;
        const int idx = c + lane;
        if (idx < end) {
            pre_gid = flat[idx];
            const float4 *src = reinterpret_cast<const float4 *>(packed + (by_gauss ? pre_gid : idx));
            const float4 u = src[0], v = src[1];
            pre.x = u.x; pre.y = u.y; pre.a = u.z; pre.b = u.w; pre.c = v.x; pre.o = v.y; pre.ex = v.z; pre.ey = v.w;
        } else {
            pre.x = pre.y = 0.f; pre.ex = pre.ey = -1.f; pre.a = pre.b = pre.c = pre.o = 0.f;
            pre_gid = 0;
        }
        c += 64;
    }
    __device__ __forceinline__ void commit()
    {
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
    }
    template <typename Geom>
    __device__ __forceinline__ void init(HRec *ring_, const GRec *packed_, const int32_t *flat_, int start_, int end_,
                                         int lane_, const Geom &g)
    {
        ring = ring_; packed = packed_; flat = flat_; start = start_; end = end_; lane = lane_;
        rx0 = g.rx0; rx1 = g.rx1; ry0 = g.ry0; ry1 = g.ry1;
        nq = 0; rd = 0; c = start_; pending = false;
        if (c < end) { issue(); pending = true; }
    }
    // make at least `low` unconsumed hits available (or exhaust the range)
    __device__ __forceinline__ void refill(int low)
    {
        while ((nq - rd) < low && pending) {
            commit();
            pending = false;
            if (c < end) { issue(); pending = true; }
        }
    }
    // record of this lane's slot (k) of the pair at position `pos`; clamped to a produced slot, `valid` says
    // whether the slot exists (a lone last hit has no partner)
    __device__ __forceinline__ HRec at(int pos, int k, bool &valid) const
    {
        valid = pos + k < nq;
        return ring[min(pos + k, nq - 1) & (RING - 1)];
    }
};

}  // namespace gags_mfma
