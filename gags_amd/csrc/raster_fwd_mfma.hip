// K9 on the matrix cores (split path: D >= 16, D % 4 == 0; fused kernel: D % 32 == 0), exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit a
// k-ordered fmaf chain and zero weights are exact no-ops, so both kernels below are bit-identical
// to the sequential definition (and to each other and to the VALU kernels).
//
//  raster_fwd_feat<NB>   default; needs the scratch written by raster_weights.hip.
//      One wave per (tile, 8x8 block, slice of 32*NB <= 128 channels): a pure stream.  Per K-step one
//      8-byte pair of weights per lane (the two 32x2 A operands of the block's upper / lower half), two feature rows as one float4 per lane (the B operands, straight from
//      global/L2 into VGPRs, four steps ahead) and 2*NB MFMAs.  No alpha work, no LDS, no barrier.
//  raster_fwd_fused<NB>  fallback when the caller provides no scratch: same decomposition, but every
//      wave also runs the HitStream / alpha / transmittance pipeline itself (repeated per slice).
//
// Accumulator tile i of a group of 4 holds channels {4n+i}: one float4 load feeds 4 tiles and the
// epilogue stores 16 B per lane.  All D channels of a slice are composited in ONE walk of the list
// (gsplat re-walks it ceil(D/32) times).
#include <hip/hip_fp16.h>
#include <cstdlib>
#include <type_traits>
#include "raster_mfma_common.h"

using namespace gags_mfma;

// Phase stamps of a wave (tools/probe/: a SEPARATE probe build, -DGAGS_PROBE; the shipped library has none of this).
#ifdef GAGS_PROBE
__device__ unsigned long long *gags_probe_buf = nullptr;
extern "C" __attribute__((visibility("default"))) int gags_probe_set(void *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(gags_probe_buf), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#define GAGS_STAMP(i)                                                                                       \
    do {                                                                                                    \
        if (gags_probe_buf && threadIdx.x == 0) gags_probe_buf[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define GAGS_STAMP(i) ((void)0)
#endif

namespace {

template <int NB>
struct FwdCfg {
    static constexpr int CS = 32 * NB;
    static constexpr int VEC = NB >= 4 ? 4 : NB;
    static constexpr int NG = NB / VEC;
    static_assert(NB == 1 || NB == 2 || NB % 4 == 0, "NB in {1,2,4,8,16}");
};

// two feature rows (slot k's row for this half-wave) as B operands.  HALF: the feature table is stored as fp16
// (BASELINE.json configs[4]: half the gather traffic; the values are widened exactly and everything downstream is the
// same fp32 arithmetic, so the render is bit-identical to the fp32 path on the fp16-rounded table)
template <int NB, bool HALF = false>
__device__ __forceinline__ void load_rows(const float *__restrict__ colors, int gid, int d, int ch0, int p,
                                          float4 (&bq)[FwdCfg<NB>::NG])
{
    constexpr int VEC = FwdCfg<NB>::VEC, NG = FwdCfg<NB>::NG;
    // VEC == 1 also serves a ragged last slice (D % 32 != 0): lanes past the row read its last channel and
    // are masked at the store (an output column depends on its own B column only)
    const size_t off = (size_t)gid * d + (VEC == 1 ? min(ch0 + p, d - 1) : ch0 + VEC * p);
    if constexpr (HALF) {
        const __half *row = reinterpret_cast<const __half *>(colors) + off;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if constexpr (VEC == 4) {
                const uint2 u = *reinterpret_cast<const uint2 *>(row + gq * 128);
                const __half2 a = *reinterpret_cast<const __half2 *>(&u.x), b = *reinterpret_cast<const __half2 *>(&u.y);
                bq[gq] = make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
            } else if constexpr (VEC == 2) {
                const __half2 a = *reinterpret_cast<const __half2 *>(row);
                bq[gq] = make_float4(__low2float(a), __high2float(a), 0.f, 0.f);
            } else {
                bq[gq] = make_float4(__half2float(row[0]), 0.f, 0.f, 0.f);
            }
        }
    } else {
        const float *row = colors + off;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if constexpr (VEC == 4) bq[gq] = *reinterpret_cast<const float4 *>(row + gq * 128);
            else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2 *>(row); bq[gq] = make_float4(t.x, t.y, 0.f, 0.f); }
            else bq[gq] = make_float4(row[0], 0.f, 0.f, 0.f);
        }
    }
}

template <int NB>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[NB], float wgt, const float4 (&bc)[FwdCfg<NB>::NG])
{
    constexpr int VEC = FwdCfg<NB>::VEC, NG = FwdCfg<NB>::NG;
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        const float bv[4] = {bc[gq].x, bc[gq].y, bc[gq].z, bc[gq].w};
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            acc[gq * VEC + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wgt, bv[i], acc[gq * VEC + i], 0, 0, 0);
    }
}

// out[pix][ch] = acc (+ T*bg); accumulator row r of lane (p,k) <-> pixel q = (r&3) + 8(r>>2) + 4k of the block.
// Tq[r] = final transmittance of that pixel (fetched by the caller BEFORE this loop, unconditionally:
// a load inside the per-row in-image branch would cost one serialized L2 round trip per row).
// The background slice of this lane's channels (zeros without a background).
template <int NB>
__device__ __forceinline__ void fetch_bg(float (&bgv)[NB], const float *__restrict__ backgrounds, int ch0, int p, int d)
{
    constexpr int VEC = FwdCfg<NB>::VEC, NG = FwdCfg<NB>::NG;
#pragma unroll
    for (int j = 0; j < NB; ++j) bgv[j] = 0.f;
    if (backgrounds != nullptr) {  // wave-uniform
#pragma unroll
        for (int gq = 0; gq < NG; ++gq)
#pragma unroll
            for (int i = 0; i < VEC; ++i) bgv[gq * VEC + i] = backgrounds[min(ch0 + gq * 32 * VEC + VEC * p + i, d - 1)];
    }
}

// Every load of the wave has to have LANDED before the first store: stores count in vmcnt too (gfx9), and a wait for a
// pending load inside a row's in-image branch is a vmcnt(0) -- it also waits for the previous row's STORE to be
// acknowledged.  The 16 stores of a half then went out one write latency apart: 9.5 us per half, 37 % of a feature
// wave's life (timestamps, round 3).  The empty asm makes each value a use HERE, outside the branches.
template <int NB>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[NB], const BlockGeom &g, int width, int height, int d,
                                           int ch0, bool has_bg, float *__restrict__ render_colors, float (&bgv)[NB],
                                           const float (&Tq)[16])
{
    constexpr int VEC = FwdCfg<NB>::VEC, NG = FwdCfg<NB>::NG;
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(bgv[j]));
    float tq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { tq[r] = Tq[r]; asm volatile("" : "+v"(tq[r])); }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * g.k;
        const int qj = g.bx0 + (q & 7), qi = g.by0 + (q >> 3);
        if (qi >= height || qj >= width) continue;
        float *o = render_colors + ((size_t)qi * width + qj) * d + ch0 + VEC * g.p;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float a = acc[gq * VEC + i][r];
                v[i] = has_bg ? __builtin_fmaf(tq[r], bgv[gq * VEC + i], a) : a;
            }
            if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o + gq * 128) = make_float4(v[0], v[1], v[2], v[3]);
            else if constexpr (VEC == 2) *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
            else if (ch0 + g.p < d) o[0] = v[0];
        }
    }
}

template <int NB>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[NB], const BlockGeom &g, int width, int height, int d,
                                         int ch0, const float *__restrict__ backgrounds,
                                         float *__restrict__ render_colors, const float (&Tq)[16])
{
    float bgv[NB];
    fetch_bg<NB>(bgv, backgrounds, ch0, g.p, d);
    store_rows<NB>(acc, g, width, height, d, ch0, backgrounds != nullptr, render_colors, bgv, Tq);
}

// Feature pass over 8x8 pixel blocks: each lane owns two pixels (upper / lower half of the block), i.e. the
// wave holds TWO 32-row accumulator sets of NB channel tiles each (NB = 4: 128 channels, 128 VGPRs).
template <int NB, bool HALF>
__global__ __launch_bounds__(64, 2) void raster_fwd_feat(
    int d, int ch_base, int width, int height, int tile_w, int n_tiles, int n_slices, int n_gauss,
    const float *__restrict__ colors, const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    int n_isects, const int32_t *__restrict__ blk_rows, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const float *__restrict__ Tbuf, float *__restrict__ render_colors)
{
    static_assert(NB == 1 || NB == 2 || NB == 4, "one float4 (or less) of channels per lane");
    constexpr int CS = FwdCfg<NB>::CS, NG = FwdCfg<NB>::NG;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE * n_slices);
    const int slice = logical % n_slices, rest = logical / n_slices;
    const int blk = rest & 3;
    const int tile = gags_tile_of_order(rest >> 2, tile_w, n_tiles / tile_w);
    const int ch0 = ch_base + slice * CS;  // this launch covers channels ch_base .. ch_base + n_slices * CS - 1 (clipped to d)
    const int lane = threadIdx.x;
    BlockGeom64 g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    const int steps = cnt >> 1;
    GAGS_STAMP(0);
#ifdef GAGS_PROBE
    if (gags_probe_buf && threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        gags_probe_buf[(size_t)blockIdx.x * 8 + 6] = hw;
        gags_probe_buf[(size_t)blockIdx.x * 8 + 7] = (unsigned)steps;
    }
#endif

    // The epilogue's inputs -- final transmittance of the lane's 2 x 16 pixels, background of its channels -- are
    // requested FIRST: they are the oldest loads of the wave, long landed when the K loop ends (fetched after it they
    // cost two exposed round trips, ~4 us of a 45 us wave).
    const bool has_bg = backgrounds != nullptr;  // wave-uniform
    float TqA[16], TqB[16], bgv[NB];
    fetch_bg<NB>(bgv, backgrounds, ch0, p, d);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * k;
        const int qj = min(g.bx0 + (q & 7), width - 1);
        TqA[r] = has_bg ? Tbuf[(size_t)min(g.by0 + (q >> 3), height - 1) * width + qj] : 0.f;
        TqB[r] = has_bg ? Tbuf[(size_t)min(g.by0 + 4 + (q >> 3), height - 1) * width + qj] : 0.f;
    }

    f32x16 accA[NB], accB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[j][r] = 0.f; accB[j][r] = 0.f; }

    if (steps > 0) {
        // this lane's slot of step s, clamped to the block's last pair (surplus steps carry weight 0)
        auto slot_of = [&](int s) { return sb + 2 * min(s, steps - 1) + k; };
        const int gmax = n_gauss - 1;
        // PD-deep software pipeline: the feature rows of step s+PD are requested while step s runs its
        // MFMAs (rows mostly come from L2 / Infinity Cache: ~1-2 us), the ids another PD steps earlier.
        // The ids of a whole group of PD steps travel in ONE vector load (lane l < 2*PD holds slot 2s + l) and
        // are picked out with v_readlane when used; ids are clamped only then (a lone hit's partner slot
        // carries N) -- touching a loaded value earlier would make the compiler wait for it right after issue.
        // Vector-memory instructions per K-step: 1 (rows) + 1 (weight pair) + 1/PD (ids); they are not free
        // next to the MFMAs (hardware findings).
        constexpr int PD = 4;
        float aA[PD], aB[PD];
        float4 b[PD][NG];
        auto ids_of = [&](int s) {  // ids of steps s .. s+PD-1, clamped to the block's last pair
            return gid_s[sb + min(2 * s + (lane & (2 * PD - 1)), 2 * steps - 2 + (lane & 1))];
        };
        auto pick = [&](int gv, int i) {  // id of this lane's slot (k) of step i of the group
            const int g0 = __builtin_amdgcn_readlane(gv, 2 * i), g1 = __builtin_amdgcn_readlane(gv, 2 * i + 1);
            return min(k ? g1 : g0, gmax);
        };
        // prologue: issue the loads in exactly the order of the steady state so that the compiler's counted
        // vmcnt waits agree on both edges into the loop
        int gv = 0;
        {
            const int g0v = ids_of(0);
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                load_rows<NB, HALF>(colors, pick(g0v, i), d, ch0, p, b[i]);
                if (i == 0) gv = ids_of(PD);
                const float2 w = *reinterpret_cast<const float2 *>(wt + (size_t)slot_of(i) * 64 + 2 * p);
                aA[i] = w.x; aB[i] = w.y;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        GAGS_STAMP(1);
        for (int s = 0; s < steps; s += PD) {
            const int gv_use = gv;
#ifdef GAGS_PROBE
            if (s == PD) GAGS_STAMP(2);  // the first group's MFMAs are through: its operands had landed
#endif
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                const bool live = s + i < steps;
                mfma_step<NB>(accA, live ? aA[i] : 0.f, b[i]);
                mfma_step<NB>(accB, live ? aB[i] : 0.f, b[i]);
                load_rows<NB, HALF>(colors, pick(gv_use, i), d, ch0, p, b[i]);
                if (i == 0) gv = ids_of(s + 2 * PD);
                const float2 w = *reinterpret_cast<const float2 *>(wt + (size_t)slot_of(s + PD + i) * 64 + 2 * p);
                aA[i] = w.x; aB[i] = w.y;
                // the machine scheduler would otherwise sink these loads to just before their use PD steps
                // later (it minimises register pressure), collapsing the pipeline to depth 1
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    GAGS_STAMP(3);
    // accumulator row r of lane (p,k) <-> pixel q = (r&3) + 8(r>>2) + 4k of a 8x4 half
    BlockGeom half;
    half.p = p; half.k = k; half.bx0 = g.bx0;
    half.by0 = g.by0;
    store_rows<NB>(accA, half, width, height, d, ch0, has_bg, render_colors, bgv, TqA);
    GAGS_STAMP(4);
    half.by0 = g.by0 + 4;
    store_rows<NB>(accB, half, width, height, d, ch0, has_bg, render_colors, bgv, TqB);
    GAGS_STAMP(5);
}

// ---- feature pass on the 16-bit matrix cores (opt-in: GAGS_FWD_F16MFMA; fp16 feature table, D % 128 == 0) -------------
// BASELINE.json configs[4] "fp16 features on CDNA4": v_mfma_f32_32x32x16_f16, K = 16 slots per step.  The features ARE
// fp16, so the B operand is exact; the weights alpha*T (fp32) go in as an fp16 head + tail pair scaled by 2^12 (every
// head and tail is then a normal fp16 number), two MFMAs per tile, products exact in the fp32 accumulator: the render
// differs from the fp32 arithmetic by ~2^-22 relative per term (tests: <= 2e-6 rel-L2), not bit-identical -> opt-in.
// 16 MFMAs of 32 cycles per 16 slots and 128 channels instead of 64 of 64 cycles.
// K runs over slots, the feature table is slot-major: a lane's B operand is 8 slots x 1 channel.  Each lane therefore
// fetches 4 consecutive channels (8 bytes) of each of its 8 slots and re-packs them into the four channel tiles'
// operands (tile j = channels ch0 + 4 n + j: the same "strided" tiles as the fp32 kernel, same float4 epilogue).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(64, 2) void raster_fwd_feat_f16(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, int n_gauss,
    const __half *__restrict__ colors, const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    int n_isects, const int32_t *__restrict__ blk_rows, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const float *__restrict__ Tbuf, float *__restrict__ render_colors)
{
    constexpr int NB = 4, CS = 128;
    constexpr float WSCALE = 4096.0f;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE * n_slices);
    const int slice = logical % n_slices, rest = logical / n_slices;
    const int blk = rest & 3;
    const int tile = gags_tile_of_order(rest >> 2, tile_w, n_tiles / tile_w);
    const int ch0 = slice * CS;
    const int lane = threadIdx.x;
    BlockGeom64 g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;  // A operand: pixel p of a half-block; B operand: channel group p; k: which 8 of the step's 16 slots
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    const int steps = (cnt + 15) >> 4;

    f32x16 accA[NB], accB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[j][r] = 0.f; accB[j][r] = 0.f; }

    if (steps > 0) {
        const int gmax = n_gauss - 1;
        // Raw operands of one step = this lane's 8 slots (8 k + i of the step).  The feature rows are gathers that mostly
        // miss L2: they are requested TWO steps ahead, the ids three, the weight rows (a stream the slices share) one.
        auto ids_of = [&](int s, int (&id)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) id[i] = min(gid_s[sb + min(16 * s + 8 * k + i, cnt - 1)], gmax);
        };
        auto fetch_f = [&](const int (&id)[8], uint2 (&f)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const uint2 *>(colors + (size_t)id[i] * d + ch0 + 4 * p);
        };
        auto fetch_w = [&](int s, float2 (&wv)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int slot = 16 * s + 8 * k + i;
                wv[i] = *reinterpret_cast<const float2 *>(wt + (size_t)(sb + min(slot, cnt - 1)) * 64 + 2 * p);
                if (slot >= cnt) wv[i] = make_float2(0.f, 0.f);
            }
        };
        int idn[8];
        uint2 f0[8], f1[8], f2[8];  // features of steps s, s+1, s+2
        float2 w0[8], w1[8];        // weights of steps s, s+1
        ids_of(0, idn); fetch_f(idn, f0); fetch_w(0, w0);
        ids_of(min(1, steps - 1), idn); fetch_f(idn, f1);
        ids_of(min(2, steps - 1), idn);
        for (int s = 0; s < steps; ++s) {
            fetch_f(idn, f2);                      // step s + 2 (clamped past the end: harmless re-reads)
            ids_of(min(s + 3, steps - 1), idn);
            fetch_w(min(s + 1, steps - 1), w1);
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetches above this step's arithmetic
            h16x8 ahA, alA, ahB, alB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = w0[i].x * WSCALE, b = w0[i].y * WSCALE;
                const _Float16 ha = (_Float16)a, hb = (_Float16)b;
                ahA[i] = ha; alA[i] = (_Float16)(a - (float)ha);
                ahB[i] = hb; alB[i] = (_Float16)(b - (float)hb);
            }
            h16x8 bj[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {  // channel tile j: component j of each slot's four halves
                union { h16x8 v; unsigned u[4]; } t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned lo = (j < 2) ? f0[2 * q].x : f0[2 * q].y, hi = (j < 2) ? f0[2 * q + 1].x : f0[2 * q + 1].y;
                    t.u[q] = (j & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
                }
                bj[j] = t.v;
            }
            // four independent accumulators between two updates of the same one
#pragma unroll
            for (int j = 0; j < NB; ++j) accA[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahA, bj[j], accA[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) accB[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahB, bj[j], accB[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) accA[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alA, bj[j], accA[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) accB[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alB, bj[j], accB[j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { f0[i] = f1[i]; f1[i] = f2[i]; w0[i] = w1[i]; }
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[j][r] *= (1.0f / WSCALE); accB[j][r] *= (1.0f / WSCALE); }
    float Tq[16];
    BlockGeom half;
    half.p = p; half.k = k; half.bx0 = g.bx0;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        half.by0 = g.by0 + 4 * hb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = (r & 3) + 8 * (r >> 2) + 4 * k;
            const int qj = min(half.bx0 + (q & 7), width - 1), qi = min(half.by0 + (q >> 3), height - 1);
            Tq[r] = Tbuf[(size_t)qi * width + qj];
        }
        epilogue<NB>(hb ? accB : accA, half, width, height, d, ch0, backgrounds, render_colors, Tq);
    }
}

// ---- feature pass of an fp32 table on the 16-bit matrix cores, fp32-equivalent (the DEFAULT for D >= 128) -----------------
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32.  Both operands are written as THREE bfloat16
// terms obtained by truncation -- t0 = the top 16 bits of x, t1 = the top 16 bits of x - t0, t2 = x - t0 - t1: every
// subtraction is exact and 3 x 8 significand bits hold fp32's 24, so  x = t0 + t1 + t2  EXACTLY, with fp32's own exponent
// range (no scales, nothing to overflow or flush: the reason bf16 and not fp16 here -- in the forward K runs over SLOTS, so
// a per-Gaussian scale of the feature rows could not be factored out of the accumulator).  A product is issued as its six
// terms of order <= 2 in 2^-8:   w f ~ a2 b0 + a1 b1 + a0 b2 + a1 b0 + a0 b1 + a0 b0   (dropped: a1 b2, a2 b1, a2 b2 <=
// 2^-23 |w f|); every partial product of two bf16 numbers is exact in the fp32 accumulator.  So each product enters the
// sum with a relative error below one fp32 rounding and the sum is accumulated in fp32 by the matrix core: as close to the
// float64 statement as the fp32 matrix instructions (tests/test_parity_gpu.py::test_forward_x16_*), but NOT bit-identical
// to the sequential fmaf chain of the oracle -- GAGS_FWD_EXACT selects raster_fwd_feat above, which is.
// 48 MFMAs of 32 cycles per 16 slots x 64 pixels x 128 channels instead of 64 of 64 cycles.
// One wave per (tile, 8x8 block) and PAIR of 128-channel slices (round 6: see the kernel), K-steps of 16 slots: lane (p, kg) holds, as A operand, the weights of
// pixel p (upper / lower half of the block) for slots 8 kg .. 8 kg + 7 of the step, and as B operand channel 4 p + j of
// the same eight slots (tile j = channels ch0 + 4 n + j: the strided tiles of the fp32 kernel, same float4 epilogue).
// The split is done in registers on the way (per four values 8 v_and + 4 packed fp32 subtractions + 6 v_perm: 216 per step;
// rounds 5's scalar subtractions: 264), then the 48 MFMAs run (four accumulators in rotation) with the next step's requests placed among them: its weights ahead
// of them, its rows in the middle -- into the registers the first two channel tiles' operands have left.  The ids of a step's slots arrive as ONE vector load (each 16-lane row holds its half-wave's eight ids) and
// reach the address arithmetic through DPP row broadcasts: one VALU instruction per gathered row.
typedef short bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2s __attribute__((ext_vector_type(2)));

struct Op3 {  // the three bf16 terms of one MFMA operand (8 K elements per lane)
    bf16x8 t[3];
};

__device__ __forceinline__ void mfma6(f32x16 &acc, const Op3 &a, const Op3 &b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[2], b.t[0], acc, 0, 0, 0);  // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[0], acc, 0, 0, 0);
}

// fp16 feature table (BASELINE.json configs[4]): a half has an 11-bit significand, so TWO bf16 terms by truncation hold it
// exactly (8 + 3 bits; bf16 has fp32's exponent range: no scale, half subnormals included) -- the B operand costs two terms
// instead of three, a product five MFMA terms instead of six (a0 b2 does not exist), the split 8 instead of 11 VALU
// instructions per pair of values, and the gather moves half the bytes.
struct Op2 {
    bf16x8 t[2];
};

__device__ __forceinline__ void mfma5(f32x16 &acc, const Op3 &a, const Op2 &b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[2], b.t[0], acc, 0, 0, 0);  // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[0], acc, 0, 0, 0);
}

// The splits as the K loop issues them (round 6): the residuals of TWO values that sit in adjacent registers -- the (upper,
// lower) weight pair of a slot, two neighbouring channels of a gathered row -- are formed by one packed fp32 subtraction
// (v_pk_add_f32), and the bf16 terms of two consecutive SLOTS are packed afterwards: per four values 8 v_and + 4 v_pk_add +
// 6 v_perm instead of 8 + 8 + 6 (the kernel is bound by its issue slots: 264 -> 216 split instructions per 16-slot step).
__device__ __forceinline__ f32x2s trunc16(f32x2s v)
{
    return f32x2s{__uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u)};
}
struct Res3 {  // a pair of values and what the first / the first two bf16 terms leave over
    f32x2s v, r1, r2;
};
__device__ __forceinline__ Res3 res3(float a, float b)
{
    Res3 o;
    o.v = f32x2s{a, b};
    o.r1 = o.v - trunc16(o.v);
    o.r2 = o.r1 - trunc16(o.r1);
    return o;
}
__device__ __forceinline__ unsigned pack_hi(float a, float b)  // {top 16 bits of a | top 16 bits of b << 16}
{
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
union OpU {
    bf16x8 v;
    unsigned u[4];
};

__device__ __forceinline__ float half_lo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float half_hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

// id of slot I (0..7) of the lane's half of the step, from a vector whose every 16-lane row holds the eight ids of its
// half-wave twice (lane l: slot 8 (l >> 5) + (l & 7)): DPP row broadcast, fused by the compiler into the consuming add
template <int I>
__device__ __forceinline__ unsigned row_bcast_add(unsigned v, unsigned add)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + I, 0xf, 0xf, false) + add;
}

// BIG: the table does not fit 32-bit byte offsets (N D 4 >= 2^32; halves: N D 2): offsets in elements, widened per gather
// HALF: `colors` is an fp16 table (two-term B operands, five product terms: above)
// BG: the view has a background (its rows take background * final transmittance; both parked in LDS during the K loops)
template <bool BIG, bool HALF, bool BG>
__global__ __launch_bounds__(64, 2) void raster_fwd_feat_x16(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, int spw, int n_gauss,
    const float *__restrict__ colors, const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    int n_isects, const int32_t *__restrict__ blk_rows, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const float *__restrict__ Tbuf, float *__restrict__ render_colors)
{
    constexpr int NB = 4, CS = 128;
    GAGS_STAMP(0);
    // A wave serves `spw` consecutive 128-channel slices of its (tile, block) one after the other (round 6): a wave's life
    // opens with three dependent round trips -- list bounds and slot count, the ids of step 0, its gathered rows: ~9 us of a
    // 37 us wave under load (tools/probe/feat16_probe.py) -- and a later slice of the same block needs none of them: its
    // first rows are requested during the last step of the slice before and travel under that slice's stores.
    const int n_groups = n_slices / spw;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE * n_groups);
    const int grp = logical % n_groups, rest = logical / n_groups;
    const int blk = rest & 3;
    const int tile = gags_tile_of_order(rest >> 2, tile_w, n_tiles / tile_w);
    const int ch_first = grp * spw * CS;
    const int lane = threadIdx.x;
    BlockGeom64 g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];  // even
    const int steps = (cnt + 15) >> 4;
#ifdef GAGS_PROBE
    if (gags_probe_buf && threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        gags_probe_buf[(size_t)blockIdx.x * 8 + 6] = hw;
        gags_probe_buf[(size_t)blockIdx.x * 8 + 7] = (unsigned)steps;  // (needs the metadata: stamp 1 = it has arrived)
    }
    GAGS_STAMP(1);
#endif

    // The epilogue's inputs are requested first (see raster_fwd_feat) -- the final transmittance of the lane's OWN two
    // pixels and the background of its channels -- and parked in LDS for the duration of the K loops, which need every
    // register (the epilogue then reads the pixels of its accumulator rows from the lanes that own them).
    constexpr int MAX_SPW = 4;
    __shared__ __attribute__((aligned(16))) float park[BG ? (4 * MAX_SPW + 2) * 64 : 4];
    if constexpr (BG) {
        for (int j = 0; j < spw; ++j) {
            float bgv0[NB];
            fetch_bg<NB>(bgv0, backgrounds, ch_first + j * CS, p, d);
            *reinterpret_cast<float4 *>(park + 256 * j + 4 * lane) = make_float4(bgv0[0], bgv0[1], bgv0[2], bgv0[3]);
        }
        const int pjc = min(g.pj, width - 1);
        const float tA = Tbuf[(size_t)min(g.piA, height - 1) * width + pjc];
        const float tB = Tbuf[(size_t)min(g.piB, height - 1) * width + pjc];
        *reinterpret_cast<float2 *>(park + 256 * MAX_SPW + 2 * lane) = make_float2(tA, tB);
    }

    f32x16 accA[NB], accB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[j][r] = 0.f; accB[j][r] = 0.f; }

    // accumulator row r of lane (p, k) is pixel q = (r & 3) + 8 (r >> 2) + 4 k of a 8x4 half; its transmittance was parked by lane q
    // Stores of a slice: row r of the accumulators is pixel (by0 + 4 hb + (r >> 2), bx0 + 4 k + (r & 3)) of the image -- a
    // wave-uniform address (scalar registers) plus ONE per-lane offset; an image border cuts whole rows (uniform test) and
    // the columns of a half-wave (four lane masks).  The next slice's first rows and weights are in flight meanwhile.
    auto finish_slice = [&](int j) __attribute__((always_inline)) {
        const int ch0 = ch_first + j * CS;
        // (opaque copies: everything below is computed HERE, from three scalars -- left to the optimiser the sixteen row
        // offsets and four column masks are hoisted out of the slice loop and live through the K loops in registers they need)
        // (the lane id from mbcnt on an opaque mask: a thread id kept in a register through the K loops is one register too
        // many -- its reload from scratch would wait for the next slice's rows, which are in flight here)
        int w_ = width, d_ = d;
        unsigned ones = ~0u;
        asm volatile("" : "+s"(w_), "+s"(d_), "+s"(ones));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        const int kk = ln >> 5, pp = ln & 31;
        const unsigned store_off = (unsigned)((4 * kk) * d_ + 4 * pp) * 4u;
        const unsigned row_stride = (unsigned)w_ * (unsigned)d_ * 4u, px_stride = (unsigned)d_ * 4u;
        bool col_ok[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) col_ok[c] = g.bx0 + 4 * kk + c < w_;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float Tq[16], bgv[NB];
            if constexpr (BG) {
                const float4 b4 = *reinterpret_cast<const float4 *>(park + 256 * j + 4 * ln);
                bgv[0] = b4.x; bgv[1] = b4.y; bgv[2] = b4.z; bgv[3] = b4.w;
#pragma unroll
                for (int r = 0; r < 16; ++r) Tq[r] = park[256 * MAX_SPW + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + hb];
            }
            const char *blk_base = reinterpret_cast<const char *>(render_colors) +
                                   (((size_t)(g.by0 + 4 * hb) * w_ + g.bx0) * d_ + ch0) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (g.by0 + 4 * hb + (r >> 2) >= height) continue;  // (uniform)
                const unsigned uo = (unsigned)(r >> 2) * row_stride + (unsigned)(r & 3) * px_stride;  // (< 2^32: four rows of the image)
                float4 v = hb ? make_float4(accB[0][r], accB[1][r], accB[2][r], accB[3][r])
                              : make_float4(accA[0][r], accA[1][r], accA[2][r], accA[3][r]);
                if constexpr (BG) {
                    v.x = __builtin_fmaf(Tq[r], bgv[0], v.x); v.y = __builtin_fmaf(Tq[r], bgv[1], v.y);
                    v.z = __builtin_fmaf(Tq[r], bgv[2], v.z); v.w = __builtin_fmaf(Tq[r], bgv[3], v.w);
                }
                unsigned so = store_off;
                asm volatile("" : "+v"(so));  // (opaque per store: scalar row base + this offset is the store's own addressing mode)
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                if (col_ok[r & 3])
                    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4 *>(const_cast<char *>(blk_base) + (size_t)uo + so));
            }
        }
#pragma unroll
        for (int jj = 0; jj < NB; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA[jj][r] = 0.f; accB[jj][r] = 0.f; }
    };

    if (steps > 0) {
        const unsigned gmax = (unsigned)(n_gauss - 1);
        // The lane's eight slots of step s are sb + 16 s + 8 k + i.  Behind the block's count its region holds ZERO slots
        // up to the next multiple of 16 (raster_weights.hip): weight 0, id N (clamped into the table: a finite row).
        const unsigned id_off = (unsigned)(sb + 8 * k + (lane & 7)) * 4u;  // (slot indices stay below 2^30: GAGS_MAX_ISECTS)
        auto load_ids = [&](int s) {
            return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(gid_s) + (id_off + 64u * (unsigned)s));
        };
        // (a wave-uniform base in scalar registers + ONE 32-bit per-lane offset: a 64-bit per-lane pointer is two registers
        // the K loop does not have)
        const char *wbase = reinterpret_cast<const char *>(wt) + (size_t)sb * 256;
        const unsigned w_off = (unsigned)(8 * k * 64 + 2 * p) * 4u;
        auto load_w = [&](int s, float2 (&w)[8]) {
            const char *src = wbase + (size_t)s * (16 * 256);
            unsigned wo = w_off;
            asm volatile("" : "+v"(wo));  // (opaque: keeps `wbase + w_off` from being hoisted as a 64-bit per-lane pointer)
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = *reinterpret_cast<const float2 *>(src + wo + i * 256);
        };
        constexpr unsigned ESZ = HALF ? 2u : 4u;  // bytes per table element
        using Row = typename std::conditional<HALF, uint2, float4>::type;  // the lane's four channels of one row
        const unsigned lane_off0 = BIG ? (unsigned)(ch_first + 4 * p) : (unsigned)(ch_first + 4 * p) * ESZ;
        const unsigned slice_off = BIG ? (unsigned)CS : (unsigned)CS * ESZ;
        const unsigned row_pitch = BIG ? (unsigned)d : (unsigned)d * ESZ;
        auto load_f = [&](unsigned idv, Row (&f)[8], unsigned lane_off) {  // channels ch0 + 4 p .. + 3 of the eight rows
            const unsigned ro = min(idv, gmax) * row_pitch;  // (the zero slots carry id N)
            const unsigned o[8] = {row_bcast_add<0>(ro, lane_off), row_bcast_add<1>(ro, lane_off), row_bcast_add<2>(ro, lane_off),
                                   row_bcast_add<3>(ro, lane_off), row_bcast_add<4>(ro, lane_off), row_bcast_add<5>(ro, lane_off),
                                   row_bcast_add<6>(ro, lane_off), row_bcast_add<7>(ro, lane_off)};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (BIG) f[i] = *reinterpret_cast<const Row *>(reinterpret_cast<const char *>(colors) + (size_t)o[i] * ESZ);
                else f[i] = *reinterpret_cast<const Row *>(reinterpret_cast<const char *>(colors) + o[i]);
            }
        };
        // software pipeline: rows and weights of step s + 1 are requested during step s, the ids two steps further ahead
        // (loads return in order: an id requested late would be waited for together with the rows issued before it)
        Row F[8];
        float2 W[8];
        unsigned id1, id2;
        // the steps of the wave's slices form ONE sequence: (slice 0, step 0 .. steps - 1), (slice 1, step 0 ..), ...; the
        // ids and weights of a step do not depend on the slice, so the prefetches simply wrap around (behind the last slice:
        // redundant loads of valid addresses, as the clamped ones were)
        load_w(0, W);
        load_f(load_ids(0), F, lane_off0);
        int s3 = 1 % steps;  // step whose ids are requested next
        id1 = load_ids(s3);
        s3 = s3 + 1 == steps ? 0 : s3 + 1;
        id2 = load_ids(s3);
        s3 = s3 + 1 == steps ? 0 : s3 + 1;
        GAGS_STAMP(2);  // ids of step 0 arrived, its rows and weights requested
        auto step = [&](int s, int j) {
#ifdef GAGS_PROBE
            if (s == 1 && j == 0) GAGS_STAMP(3);  // step 0 multiplied: its operands had landed
#endif
            // A operands: the weights of the lane's pixel pair for its eight slots
            Op3 aA, aB;
            {
                OpU oa[3], ob[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const Res3 e = res3(W[2 * q].x, W[2 * q].y), o = res3(W[2 * q + 1].x, W[2 * q + 1].y);  // (upper, lower) of two slots
                    oa[0].u[q] = pack_hi(e.v[0], o.v[0]); oa[1].u[q] = pack_hi(e.r1[0], o.r1[0]); oa[2].u[q] = pack_hi(e.r2[0], o.r2[0]);
                    ob[0].u[q] = pack_hi(e.v[1], o.v[1]); ob[1].u[q] = pack_hi(e.r1[1], o.r1[1]); ob[2].u[q] = pack_hi(e.r2[1], o.r2[1]);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) { aA.t[t] = oa[t].v; aB.t[t] = ob[t].v; }
            }
            // B operands of the four channel tiles; then the rows' registers are free for the next step's
            using OpB = typename std::conditional<HALF, Op2, Op3>::type;
            OpB b[NB];
            {
                if constexpr (HALF) {
                    OpU o[NB][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // channels (0, 1) and (2, 3) of slots 2 q and 2 q + 1; a half leaves at most 3 bits behind its first term
                        const f32x2s e01 = {half_lo(F[2 * q].x), half_hi(F[2 * q].x)}, e23 = {half_lo(F[2 * q].y), half_hi(F[2 * q].y)};
                        const f32x2s o01 = {half_lo(F[2 * q + 1].x), half_hi(F[2 * q + 1].x)}, o23 = {half_lo(F[2 * q + 1].y), half_hi(F[2 * q + 1].y)};
                        const f32x2s re01 = e01 - trunc16(e01), re23 = e23 - trunc16(e23), ro01 = o01 - trunc16(o01), ro23 = o23 - trunc16(o23);
                        o[0][0].u[q] = pack_hi(e01[0], o01[0]); o[0][1].u[q] = pack_hi(re01[0], ro01[0]);
                        o[1][0].u[q] = pack_hi(e01[1], o01[1]); o[1][1].u[q] = pack_hi(re01[1], ro01[1]);
                        o[2][0].u[q] = pack_hi(e23[0], o23[0]); o[2][1].u[q] = pack_hi(re23[0], ro23[0]);
                        o[3][0].u[q] = pack_hi(e23[1], o23[1]); o[3][1].u[q] = pack_hi(re23[1], ro23[1]);
                    }
#pragma unroll
                    for (int j = 0; j < NB; ++j) { b[j].t[0] = o[j][0].v; b[j].t[1] = o[j][1].v; }
                } else {
                    OpU o[NB][3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const Res3 e01 = res3(F[2 * q].x, F[2 * q].y), e23 = res3(F[2 * q].z, F[2 * q].w);
                        const Res3 o01 = res3(F[2 * q + 1].x, F[2 * q + 1].y), o23 = res3(F[2 * q + 1].z, F[2 * q + 1].w);
                        o[0][0].u[q] = pack_hi(e01.v[0], o01.v[0]); o[0][1].u[q] = pack_hi(e01.r1[0], o01.r1[0]); o[0][2].u[q] = pack_hi(e01.r2[0], o01.r2[0]);
                        o[1][0].u[q] = pack_hi(e01.v[1], o01.v[1]); o[1][1].u[q] = pack_hi(e01.r1[1], o01.r1[1]); o[1][2].u[q] = pack_hi(e01.r2[1], o01.r2[1]);
                        o[2][0].u[q] = pack_hi(e23.v[0], o23.v[0]); o[2][1].u[q] = pack_hi(e23.r1[0], o23.r1[0]); o[2][2].u[q] = pack_hi(e23.r2[0], o23.r2[0]);
                        o[3][0].u[q] = pack_hi(e23.v[1], o23.v[1]); o[3][1].u[q] = pack_hi(e23.r1[1], o23.r1[1]); o[3][2].u[q] = pack_hi(e23.r2[1], o23.r2[1]);
                    }
#pragma unroll
                    for (int j = 0; j < NB; ++j) { b[j].t[0] = o[j][0].v; b[j].t[1] = o[j][1].v; b[j].t[2] = o[j][2].v; }
                }
            }
            auto mm = [&](f32x16 &acc, const Op3 &a, const OpB &bb) __attribute__((always_inline)) {
                if constexpr (HALF) mfma5(acc, a, bb);
                else mfma6(acc, a, bb);
            };
            __builtin_amdgcn_sched_barrier(0);
            // the next step's operands, in the order it consumes them: the weights (its A split comes first) ahead of the MFMAs,
            // the rows in the middle of them -- into their own registers and those the first two channel tiles' operands have
            // left (none else are free).  Round 5 had them the other way round: 2 % slower
            load_w(s + 1 == steps ? 0 : s + 1, W);
            __builtin_amdgcn_sched_barrier(0);
            mm(accA[0], aA, b[0]); mm(accB[0], aB, b[0]);
            mm(accA[1], aA, b[1]); mm(accB[1], aB, b[1]);
            __builtin_amdgcn_sched_barrier(0);
            {
                const bool wrap = s + 1 == steps;
                const int jn = (wrap && j + 1 < spw) ? j + 1 : j;  // slice of the next step of the sequence
                load_f(id1, F, lane_off0 + (unsigned)jn * slice_off);
                id1 = id2;
                id2 = load_ids(s3);
                s3 = s3 + 1 == steps ? 0 : s3 + 1;
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(accA[2], aA, b[2]); mm(accB[2], aB, b[2]);
            mm(accA[3], aA, b[3]); mm(accB[3], aB, b[3]);
        };
        for (int j = 0; j < spw; ++j) {
            for (int s = 0; s < steps; ++s) step(s, j);
#ifdef GAGS_PROBE
            if (j == 0) GAGS_STAMP(4);
#endif
            finish_slice(j);
        }
    } else {
        for (int j = 0; j < spw; ++j) finish_slice(j);
    }
    GAGS_STAMP(5);
}

template <int NB>
__global__ __launch_bounds__(64, (NB >= 16 ? 1 : 2)) void raster_fwd_fused(
    int d, int width, int height, int tile_w, int n_tiles, int n_slices, const GRec *__restrict__ packed,
    const float *__restrict__ colors, const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids, int by_gauss)
{
    constexpr int CS = FwdCfg<NB>::CS, NG = FwdCfg<NB>::NG;
    __shared__ __attribute__((aligned(16))) HRec ring[RING];
    __shared__ float Tb[32];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8 * n_slices);
    const int slice = logical % n_slices, rest = logical / n_slices;
    const int blk = rest & 7;
    const int tile = gags_tile_of_order(rest >> 3, tile_w, n_tiles / tile_w);
    const int ch0 = slice * CS;
    const int lane = threadIdx.x;
    BlockGeom g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !g.inside;
    HitStream hs;
    hs.by_gauss = by_gauss != 0;
    hs.init(ring, packed, flatten_ids, start, end, lane, g);
    hs.refill(6);
    if (!__all(st.done) && hs.rd < hs.nq) {
        bool v_n;
        HRec h_n = hs.at(hs.rd, k, v_n);
        float a_n = eval_alpha(h_n, g.px, g.py, v_n);
        int sidx_n = h_n.sidx;
        float4 b0[NG], b1[NG];
        load_rows<NB>(colors, h_n.gid, d, ch0, p, b0);
        auto kstep = [&](float4(&bc)[NG], float4(&bn)[NG]) -> bool {
            const float a_c = a_n;
            const int sidx_c = sidx_n;
            hs.rd += 2;
            if ((hs.nq - hs.rd) < 6 && hs.pending) hs.refill(6);
            const bool more = hs.rd < hs.nq;
            h_n = hs.at(hs.rd, k, v_n);
            a_n = eval_alpha(h_n, g.px, g.py, v_n);
            sidx_n = h_n.sidx;
            load_rows<NB>(colors, h_n.gid, d, ch0, p, bn);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
            bool blended;
            const float wgt = step_pair(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
            st.cur = blended ? sidx_c : st.cur;
            mfma_step<NB>(acc, wgt, bc);
            return more && !__all(st.done);
        };
        // two K-steps per trip, ONE exit test (a second loop exit makes the register allocator shuffle the
        // accumulators); a surplus step past the end / past saturation carries zero weights = exact no-op
        bool go = true;
        while (go) {
            kstep(b0, b1);
            go = kstep(b1, b0);
        }
    }
    {
        const auto cs = __builtin_amdgcn_permlane32_swap((unsigned)st.cur, (unsigned)st.cur, false, false);
        st.cur = max((int)cs[0], (int)cs[1]);
    }
    if (k == 0) Tb[p] = st.T;
    __builtin_amdgcn_wave_barrier();
    if (k == 0 && g.inside && slice == 0) {
        const size_t pix = (size_t)g.pi * width + g.pj;
        render_alphas[pix] = 1.0f - st.T;
        last_ids[pix] = st.cur;
    }
    float Tq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) Tq[r] = Tb[(r & 3) + 8 * (r >> 2) + 4 * k];
    epilogue<NB>(acc, g, width, height, d, ch0, backgrounds, render_colors, Tq);
}

// channels [ch_base, ch_base + ch_count) in slices of 32 * NB (NB == 1: the last slice may be ragged and must end at d)
template <int NB, bool HALF>
int launch_feat(int d, int ch_base, int ch_count, int width, int height, int n_gauss, const float *colors,
                const float *backgrounds, const int32_t *offsets, int n_isects, const int32_t *blk_rows, const float *wt,
                const int32_t *gid_s, const float *Tbuf, float *out, hipStream_t st)
{
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = (ch_count + 32 * NB - 1) / (32 * NB);
    hipLaunchKernelGGL((raster_fwd_feat<NB, HALF>), dim3(n_tiles * GAGS_BLOCKS_PER_TILE * n_slices), dim3(64), 0, st, d, ch_base, width,
                       height, tile_w, n_tiles, n_slices, n_gauss, colors, backgrounds, offsets, n_isects, blk_rows, wt, gid_s,
                       Tbuf, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// Any width D >= 16 in one call: 128-channel slices first, then 64, then 32-channel slices of which the last may be
// ragged (masked lanes) -- 513 = 512 CLIP channels + 1 (BASELINE.json configs[4]) is 4 wide slices + one lane of a
// narrow one, all on the matrix cores and into ONE output tensor.  Rows of an odd width are only 4-byte (fp16 table:
// 2-byte) aligned; vector loads / stores of global memory tolerate that on this part (unaligned access mode).
template <bool BIG, bool HALF, typename... Args>
static void launch_x16(bool bg, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args)
{
    if (bg) hipLaunchKernelGGL((raster_fwd_feat_x16<BIG, HALF, true>), grid, block, lds, st, args...);
    else hipLaunchKernelGGL((raster_fwd_feat_x16<BIG, HALF, false>), grid, block, lds, st, args...);
}

// 128-channel slices a wave of raster_fwd_feat_x16 serves one after the other (1, 2 or 4; must divide the slice count).
// GAGS_FWD_SPW overrides (experiments; read once).
static int fwd_slices_per_wave(int n_slices)
{
    static const int want = [] {
        const char *e = getenv("GAGS_FWD_SPW");
        const int v = e ? atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4) ? v : 0;
    }();
    int spw = want ? want : 2;  // (C3, one box: 1: 2.16 ms, 2: 2.08, 4: 2.13 -- longer waves, longer tail of the launch)
    while (spw > 1 && n_slices % spw != 0) spw >>= 1;
    return spw;
}

template <bool HALF>
int launch_feat_any(int d, int width, int height, int n_gauss, const float *colors, int f16_mfma, int exact,
                    const float *backgrounds, const int32_t *offsets, int n_isects, const int32_t *blk_rows,
                    const float *wt, const int32_t *gid_s, const float *Tbuf, float *out, hipStream_t st)
{
    int done = 0, rc = GAGS_OK;
#define ARGS width, height, n_gauss, colors, backgrounds, offsets, n_isects, blk_rows, wt, gid_s, Tbuf, out, st
    if (d >= 128) {
        done = d / 128 * 128;
        const int64_t table_elems = (int64_t)n_gauss * d;
        if constexpr (HALF) {
            if (!f16_mfma && !exact && table_elems + 1024 < (1ll << 32)) {
                // the default since round 6: the bf16 matrix cores, B = the half as two exact bf16 terms, A = the weight as
                // three (exact), five product terms -- fp32-equivalent like the fp32 table's default (raster_fwd_feat_x16)
                const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
                const int n_tiles = tile_w * tile_h, n_slices = done / 128, spw = fwd_slices_per_wave(n_slices);
                if (table_elems * 2 + 4096 < (1ll << 32))
                    launch_x16<false, true>(backgrounds != nullptr, dim3(n_tiles * GAGS_BLOCKS_PER_TILE * (n_slices / spw)), dim3(64), 0, st, d,
                                       width, height, tile_w, n_tiles, n_slices, spw, n_gauss, colors, backgrounds, offsets, n_isects,
                                       blk_rows, wt, gid_s, Tbuf, out);
                else
                    launch_x16<true, true>(backgrounds != nullptr, dim3(n_tiles * GAGS_BLOCKS_PER_TILE * (n_slices / spw)), dim3(64), 0, st, d,
                                       width, height, tile_w, n_tiles, n_slices, spw, n_gauss, colors, backgrounds, offsets, n_isects,
                                       blk_rows, wt, gid_s, Tbuf, out);
                GAGS_CHECK_LAUNCH();
            } else if (f16_mfma) {  // opt-in: the f16 matrix cores with a fixed weight scale (round 2's kernel)
                const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
                const int n_tiles = tile_w * tile_h, n_slices = done / 128;
                hipLaunchKernelGGL(raster_fwd_feat_f16, dim3(n_tiles * GAGS_BLOCKS_PER_TILE * n_slices), dim3(64), 0, st, d,
                                   width, height, tile_w, n_tiles, n_slices, n_gauss, reinterpret_cast<const __half *>(colors),
                                   backgrounds, offsets, n_isects, blk_rows, wt, gid_s, Tbuf, out);
                GAGS_CHECK_LAUNCH();
            } else {
                rc = launch_feat<4, true>(d, 0, done, ARGS);
            }
        } else if (!exact) {  // the default: 16-bit matrix cores on fp32-equivalent split operands
            const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
            const int n_tiles = tile_w * tile_h, n_slices = done / 128, spw = fwd_slices_per_wave(n_slices);
            if ((int64_t)n_gauss * d + 1024 >= (1ll << 32)) {
                // the BIG instantiation keeps row offsets in 32-bit float units: a table of 2^32 elements or more (16 GiB;
                // 8.4 M x 512) would wrap them.  Such a table takes the fp32 matrix instructions (64-bit row offsets; the
                // oracle's own chain) instead of reading the wrong rows
                rc = launch_feat<4, HALF>(d, 0, done, ARGS);
            } else if ((int64_t)n_gauss * d * 4 + 4096 < (1ll << 32))
                launch_x16<false, false>(backgrounds != nullptr, dim3(n_tiles * GAGS_BLOCKS_PER_TILE * (n_slices / spw)), dim3(64), 0, st, d,
                                   width, height, tile_w, n_tiles, n_slices, spw, n_gauss, colors, backgrounds, offsets, n_isects,
                                   blk_rows, wt, gid_s, Tbuf, out);
            else
                launch_x16<true, false>(backgrounds != nullptr, dim3(n_tiles * GAGS_BLOCKS_PER_TILE * (n_slices / spw)), dim3(64), 0, st, d,
                                   width, height, tile_w, n_tiles, n_slices, spw, n_gauss, colors, backgrounds, offsets, n_isects,
                                   blk_rows, wt, gid_s, Tbuf, out);
            GAGS_CHECK_LAUNCH();
        } else {
            rc = launch_feat<4, HALF>(d, 0, done, ARGS);
        }
        if (rc != GAGS_OK) return rc;
    }
    if (d - done >= 64) {
        rc = launch_feat<2, HALF>(d, done, 64, ARGS);
        if (rc != GAGS_OK) return rc;
        done += 64;
    }
    if (d - done > 0) rc = launch_feat<1, HALF>(d, done, d - done, ARGS);
#undef ARGS
    return rc;
}

template <int NB>
int launch_fused(int d, int width, int height, const GRec *packed, const float *colors, const float *backgrounds,
                 const int32_t *offsets, const int32_t *flat, int n_isects, float *out, float *alphas,
                 int32_t *last_ids, int by_gauss, hipStream_t st)
{
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h, n_slices = d / (32 * NB);
    hipLaunchKernelGGL(raster_fwd_fused<NB>, dim3(n_tiles * 8 * n_slices), dim3(64), 0, st, d, width, height, tile_w,
                       n_tiles, n_slices, packed, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids, by_gauss);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

}  // namespace

// feature pass of the split forward (after gags_raster_weights_launch)
int gags_raster_fwd_feat_launch(int d, int width, int height, int n_gauss, const float *colors, int colors_f16, int exact,
                                const float *backgrounds, const int32_t *offsets, int n_isects,
                                const int32_t *blk_rows, const float *wt, const int32_t *gid_s, const float *Tbuf,
                                float *out, hipStream_t st)
{
    GAGS_CLEAR_ERR();
#define ARGS backgrounds, offsets, n_isects, blk_rows, wt, gid_s, Tbuf, out, st
    if (colors_f16) return launch_feat_any<true>(d, width, height, n_gauss, colors, colors_f16 == 2, exact, ARGS);
    return launch_feat_any<false>(d, width, height, n_gauss, colors, 0, exact, ARGS);
#undef ARGS
}

// single-kernel forward (no scratch)
int gags_raster_fwd_fused_launch(int d, int width, int height, const void *packed, const float *colors,
                                 const float *backgrounds, const int32_t *offsets, const int32_t *flat, int n_isects,
                                 float *out, float *alphas, int32_t *last_ids, int by_gauss, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    const GRec *pk = reinterpret_cast<const GRec *>(packed);
#define ARGS d, width, height, pk, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids, by_gauss, st
    if (d % 256 == 0) return launch_fused<8>(ARGS);
    if (d % 128 == 0) return launch_fused<4>(ARGS);
    if (d % 64 == 0) return launch_fused<2>(ARGS);
    return launch_fused<1>(ARGS);
#undef ARGS
}
