// Shared device helpers for the gfx950 kernels of libgags_hip.so.
// Numerical contract (DESIGN.md "Numerics"): fp32, no implicit FMA contraction (the build
// passes -ffp-contract=off; fmaf() is written where a fused op is part of the algorithm),
// IEEE divide/sqrt, exp(-sigma) by an explicit polynomial so that index tensors and forward
// renders are reproducible bit-for-bit on any IEEE machine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gags_raster.h"

#define GAGS_ALPHA_MAX 0.999f
#define GAGS_ALPHA_MIN (1.0f / 255.0f)
#define GAGS_T_STOP 1e-4f

// torch (or any other HIP user in the process) may leave a benign sticky error behind;
// clear it on entry so that GAGS_CHECK_LAUNCH reports only our own launches.
#define GAGS_CLEAR_ERR() (void)hipGetLastError()

#define GAGS_CHECK_LAUNCH()                       \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return GAGS_ELAUNCH; \
    } while (0)

// exp(-sigma) = 2^t, t = -sigma*log2(e) = n + f, n = rint(t), f in [-.5,.5];
// 2^f by a degree-6 polynomial (max rel. error 1.4 ulp), scaled with ldexp.
// 1 mul, 1 max, 1 rndne, 1 sub, 6 fma, 1 cvt, 1 ldexp -- all full-rate VALU on gfx950.
__device__ __forceinline__ float gags_exp_neg(float sigma)
{
    float t = sigma * -1.44269504088896341f;
    t = fmaxf(t, -125.0f);
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 0x1.444p-13f;
    p = __builtin_fmaf(p, f, 0x1.5f48cp-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2a1cp-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aeccp-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbep-3f);
    p = __builtin_fmaf(p, f, 0x1.62e43p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

__device__ __forceinline__ void gags_tile_aabb(float mx, float my, int radius, int tile_w, int tile_h,
                                               int &x0, int &x1, int &y0, int &y1)
{
    const float tr = (float)radius / (float)GAGS_TILE;
    const float tx = mx / (float)GAGS_TILE, ty = my / (float)GAGS_TILE;
    x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tile_w);
    x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tile_w);
    y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)tile_h);
    y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)tile_h);
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: it also
// waits for every global load in flight -- i.e. it lands the prefetch a double-buffered loop has just issued, once per
// step (round 3: wgrad256_kernel spent a full HBM round trip per 32-pixel step at its barrier).  Use where the waves
// exchange data through LDS only; loads still in flight are waited for where their registers are used.
__device__ __forceinline__ void gags_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// XCD-aware tile remap: the dispatcher places workgroup b on XCD b % 8 (speed only, never
// correctness).  Give each XCD a contiguous run of tiles so neighbouring tiles -- which
// share Gaussians, hence feature rows -- hit the same 4 MiB L2.  Bijective for any count.
__device__ __forceinline__ int gags_xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Tile traversal order used by the raster kernels: the image is cut into 8 horizontal bands (one
// per XCD once gags_xcd_remap has given each XCD a contiguous run of the order) and every band is
// swept COLUMN-major, so the tiles that share a Gaussian (vertical neighbours: next in order,
// horizontal neighbours: one band-height later) are processed close together in time.  This keeps
// re-touched feature / gradient rows resident in L2 / Infinity Cache.  Bijective for any grid.
__device__ __forceinline__ int gags_tile_of_order(int o, int tile_w, int tile_h)
{
    const int bh = (tile_h + 7) >> 3;
    const int per_band = bh * tile_w;
    const int band = o / per_band;
    const int r = o - band * per_band;
    const int y0 = band * bh;
    const int h = min(bh, tile_h - y0);
    const int x = r / h;
    return (y0 + (r - x * h)) * tile_w + x;
}

// Feature widths served by the split matrix-core path (weights pass + feature pass, staged backward): every D >= 16
// (128-channel slices, then 64, then 32-channel slices of which the last may be ragged) -- 16 is the width the
// reference actually rasterizes (train.py:68), 513 = 512 + 1 is BASELINE.json configs[4].  Narrower widths (RGB,
// RGB+ED, depth) run the VALU kernels.
__host__ __device__ __forceinline__ bool gags_mfma_width(int d) { return d >= 16; }

// Slot space of the matrix-core rasterizer (raster_weights.hip): every (tile, 8x8 block) owns a fixed
// region of K-step slots sized by the tile's list length, so no counting pass is needed.
//   base(tile, blk) = 4*start + 64*tile + blk * r16(L),  capacity r16(L) = L rounded up to 16,  L = end - start,  blk in 0..3
// (4 r16(L) <= 4 L + 60: the regions of a tile fit its 4 L + 64 slots.)  A region holds the block's `blk_rows` slots
// (an even count) followed by ZERO slots (weights 0, Gaussian id N) up to the next multiple of 16: kernels that consume
// slots sixteen at a time (the 16-bit matrix-core feature pass) need neither a clamp nor a mask in their last step.
// Total slot count for a view: gags_slot_count().  A slot holds 64 weights (256 B).
#define GAGS_BLOCKS_PER_TILE 4
__host__ __device__ __forceinline__ int gags_slot_base(int start, int end, int tile, int blk)
{
    const int lp = (end - start + 15) & ~15;
    return GAGS_BLOCKS_PER_TILE * start + 64 * tile + blk * lp;
}
__host__ __device__ __forceinline__ int64_t gags_slot_count(int64_t n_isects, int64_t n_tiles)
{
    return GAGS_BLOCKS_PER_TILE * n_isects + 64 * n_tiles + 64;  // (+ slack so that a kernel may read a whole 32-slot tile behind the last region)
}
