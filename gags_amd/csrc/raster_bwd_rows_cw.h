// ---- staged: rows, CHANNEL waves (round 5) -- included by raster_bwd_mfma.hip inside its anonymous namespace ----------
// The contraction of raster_bwd_rows_f16 with the roles of the four waves turned by 90 degrees.  There, wave b owns pixel
// block b of the tile for all 128 channels of the slice and the four waves' partial rows of a tile row meet in LDS (park, two
// barriers whose position the four run lengths negotiate, a 240-instruction merge); here wave w owns 32 CHANNELS of the
// slice for all four blocks of the tile -- its cotangent slab is the same 128 VGPRs: 4 blocks x 4 K-steps x (head, tail) --
// and a tile row's four block contributions meet in the wave's own accumulator: block after block is multiplied into a
// scratch accumulator and folded into the row total with one FMA per element (fixed order 0..3: bit-reproducible).  What the
// waves share instead is the A operand: the weight rows of block b are loaded, scaled and split into their fp16 terms
// ONCE, by wave b, and left in LDS in fragment order (32 KB with two terms: [block][term][K-step][lane] x 16 B) for all four
// waves to read -- the split stays at one block per wave and chunk, exactly what it was.
// A chunk is 32 consecutive tile rows, whatever the blocks' run lengths: row i of the chunk is MFMA row i for every block, a
// block that does not hold the row contributes a row of zeros (scale 0).  So chunks are not negotiated (no candidate
// exchange, no pos[] table), every wave does the same work between the two barriers of a chunk (A terms ready / A terms
// consumed), and the rows leave straight from the accumulators, 128 B per half-wave.  Price: the 32-row tiles are as full as
// the blocks are occupied (2.3 of 4 blocks per tile row at C3: 7.3 chunks of 4 x 12 MFMAs per tile and wave instead of 5.8
// bursts of 80), on a matrix pipe that was a quarter busy.
// Arithmetic per product (template <TA, NM>).  Default <2, 3>: BOTH operands as two fp16 terms after an exact power-of-two
// scaling -- w rs = a0 + a1, v cs = b0 + b1, each with |residual| <= 2^-24 of the value (two round-to-nearest steps of an
// 11-bit significand): ONE fp32-level rounding per operand -- and the three product terms of order <= 1,
//     w v ~ a0 b0 + a0 b1 + a1 b0        (dropped: a1 b1 <= 2^-24 |w v|),
// every partial product exact in the fp32 accumulator: a product enters the sum with a relative error <= 3 * 2^-24, typically
// a third of that, which stays below what an fp32 dot product of these 64-pixel columns commits in its own additions.
// Measured against float64 (tests/test_fullsize_gpu.py::test_colour_gradient_accuracy_against_float64): 1.68e-7 rel-L2, worst
// channel 2.68e-7; the fp32 matrix instructions: 1.90e-7 / 2.71e-7; the exact-weight scheme <3, 5> of raster_bwd_rows_f16
// (weights as three terms, five product terms; stage bit 1024): 1.60e-7 / 2.45e-7 for 1.33x the time.  Scales are powers of
// two: per (tile, channel) for the cotangent, and for the weights ONE constant for the view since round 6 (FIXS below; until
// round 5, and still with three-term weights or GAGS_BWD_ROWSCALE=1, one per (row, block), applied when a block's accumulator
// is folded into the total); both leave when the row is stored.
__device__ __forceinline__ void split8x2(const float (&x)[8], float scale, f16x8 &hi, f16x8 &lo)
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2v_t v = {x[i] * scale, x[i + 1] * scale};
        const f16x2_t h = __builtin_convertvector(v, f16x2_t);
        const f32x2v_t r = v - __builtin_convertvector(h, f32x2v_t);
        const f16x2_t l = __builtin_convertvector(r, f16x2_t);
        hi[i] = h[0]; hi[i + 1] = h[1];
        lo[i] = l[0]; lo[i + 1] = l[1];
    }
}

// TA = fp16 terms of a weight (2, or 3: exact), NM = product terms (3 = a0 b0 + a0 b1 + a1 b0; 5 with TA = 3)
// FIXS (round 6, the default with <2, 3>): ONE weight scale for the whole view, 2^15, instead of a power of two per (row, block).
// A weight is alpha T with alpha in [1/255, 0.99] and T in (1e-4, 1]: it lies in [2^-21.3, 1), so w 2^15 lies in [2^-6.3, 2^15)
// -- inside fp16's normal range (2^-14 .. 65504) without looking at the data.  The head term a0 is a normal half for every
// weight; the tail a1 (<= 2^-11 a0) is a normal half down to w = 2^-18 (two terms = 22 bits, as before) and a subnormal one
// (absolute spacing 2^-24 / 2^15 = 2^-39 in units of the weight) for the 3.3 binades below: those weights, < 3.8e-6, keep
// 19-22 bits.  The error a product can carry is therefore <= 3 2^-24 |w v| + 2^-39 |v|, the second part a millionth of what
// ONE fp32 rounding of a weight near 1 in the same 64-pixel column commits.  What it buys: no row maximum (32 max + shuffle),
// no scale table in LDS, and -- the scale being common to the four blocks -- a block's sums join the row total by a plain
// (packed) addition instead of an FMA with a scale read from LDS; 285 -> ~240 VALU and 63 -> 45 LDS instructions per chunk and
// wave in a kernel whose VALU and MFMA times add (tools/micro/interleave.hip).
template <int TA, int NM, bool FIXS = false>
__global__ __launch_bounds__(256, 2) void raster_bwd_rows_cw(
    int d, int width, int height, int tile_w, int n_tiles, int ch_base, int n_slices,
    const float *__restrict__ v_render_colors, const int32_t *__restrict__ offsets, int n_isects,
    const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ trow, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const int32_t *__restrict__ trow_s, float *__restrict__ prow, int prow_pitch,
    uint32_t *__restrict__ row_key, int32_t *__restrict__ row_idx, int rows_cap)
{
    constexpr int CW = 128;
    __shared__ __attribute__((aligned(16))) uint4 At[4][TA][4][64];  // A terms in fragment order
    __shared__ __attribute__((aligned(16))) float rinv_s[4][32];    // inverse row scales of the chunk, per block

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, k = lane >> 5;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int tile = gags_tile_of_order(logical / n_slices, tile_w, n_tiles / tile_w);
    const int ch0 = ch_base + (logical % n_slices) * CW;
    const int chw = ch0 + 32 * wave + n;  // this lane's channel: column n of the wave's B operands and of its rows
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    // The workgroup's life opens with dependent round trips -- the tile's list bounds, the rows they name, the blocks' slot
    // counts, the first key window, the first weight rows -- and with the 128 KB cotangent slab, whose addresses need nothing
    // but the tile's position.  So the slab is requested FIRST, before anything is known about the tile, and the metadata
    // chain (scalar loads: their own counter) runs underneath it; a tile without rows (returns below) has asked for bytes it
    // does not use.  Round 6: the chain used to run ahead of the slab -- five scalar round trips of 0.5-2 us under load per
    // ~36 us workgroup.
    // cotangent slab: B operands of the wave's 32 channels for the four blocks; K element e = 16 s + 8 k + i of a weight
    // row = pixel e >> 1 of the 8x4 half e & 1 (raster_weights.hip)
    f16x8 Bh[4][4], Bl[4][4];
    float inv_cs;  // ONE column scale per channel for the whole tile (256 pixels): the four blocks share an unscale
    float raw[4][4][8];
    {
        // pixel of K element e = 16 s + 8 k + i: row s (+ 4 for odd i) of the block, column 4 k + (i >> 1): the row is the
        // same for the whole wave, the column differs by the half-wave only -- inside the image the 128 addresses are one
        // per-lane offset plus wave-uniform terms (scalar registers / immediates); tiles cut by the image border clamp
        const bool interior = (tx + 1) * GAGS_TILE <= width && (ty + 1) * GAGS_TILE <= height;
        if (interior) {
            // (a uniform 64-bit base per pixel -- scalar registers, scalar additions -- plus ONE 32-bit lane offset, made opaque per
            // load: formed once as a 64-bit per-lane pointer it costs a 64-bit vector addition per load, 140 per wave)
            const char *tile0 = reinterpret_cast<const char *>(v_render_colors) + ((size_t)(ty * GAGS_TILE) * width + tx * GAGS_TILE) * d * 4;
            const unsigned lane_off = ((unsigned)(4 * k) * (unsigned)d + (unsigned)chw) * 4u;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = (b >> 1) * 8 + 4 * (i & 1) + s4, col = (b & 1) * 8 + (i >> 1);  // compile-time
                        unsigned lo = lane_off;
                        asm volatile("" : "+v"(lo));
                        raw[b][s4][i] = *reinterpret_cast<const float *>(tile0 + ((size_t)row * width + col) * d * 4 + lo);
                    }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int bx0 = tx * GAGS_TILE + (b & 1) * 8, by0 = ty * GAGS_TILE + (b >> 1) * 8;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = 16 * s4 + 8 * k + i;
                        const int pp = e >> 1, hh = e & 1;
                        const int qj = bx0 + (pp & 7), qi = by0 + 4 * hh + (pp >> 3);
                        const bool ok = (qi < height) && (qj < width);
                        const float v = v_render_colors[((size_t)min(qi, height - 1) * width + min(qj, width - 1)) * d + chw];
                        raw[b][s4][i] = ok ? v : 0.f;
                    }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // (the slab's loads stay ahead of the metadata chain)
    const int start = offsets[tile];
    const int end = offsets[tile + 1];
    // the four blocks' slot counts in ONE 16-byte scalar load (tile * 4 ints: aligned)
    const int4 br4 = *reinterpret_cast<const int4 *>(blk_rows + (size_t)tile * GAGS_BLOCKS_PER_TILE);
    const int R0 = trow[start], R1 = trow[end];
    if (R1 == R0) return;
    const int blk = wave;  // the block whose weight rows this wave prepares
    const int cnt = blk == 0 ? br4.x : (blk == 1 ? br4.y : (blk == 2 ? br4.z : br4.w));
    const int sb = gags_slot_base(start, end, tile, blk);
    // a slot of this tile that the forward certainly WROTE: the first slot of its first non-empty block (R1 > R0: some
    // intersection blended, so some block holds its slot).  Rows a block does not hold read it and get the scale 0; an
    // unwritten slot (allocator garbage: the scratch is never cleared) could hold NaN / Inf bit patterns, and 0 * NaN = NaN
    int dummy_sb;
    {
        const int bf = br4.x > 0 ? 0 : (br4.y > 0 ? 1 : (br4.z > 0 ? 2 : 3));
        dummy_sb = gags_slot_base(start, end, tile, bf);
    }

    // bookkeeping of a chunk [r0, r0 + 32): which of the block's slots hold its rows (the block's list is ascending)
    int pb = 0;        // slots of the block consumed so far
    int tr, gid = 0;   // key window: row / Gaussian of slot pb + n (both half-waves hold the same 32 entries)
    auto fetch_keys = [&](int first) __attribute__((always_inline)) {
        tr = 0x7fffffff;
        if (first + n < cnt) {
            tr = trow_s[sb + first + n];
            gid = gid_s[sb + first + n];
        }
    };
    fetch_keys(0);
    float A[32];
    bool present = false;
    auto open_chunk = [&](int r0) __attribute__((always_inline)) {
        // -> present (row n of the chunk is held by the block), A loads issued for it; keys written for the rows
        const int r1 = min(r0 + 32, R1);
        const bool mine = tr < r1;
        // OR over the 32 entries (both half-waves hold the same 32): four DPP steps inside the rows of 16 lanes, lane 15's
        // value into the next row, the result read from lane 31 into a SCALAR register -- a chain of five ds_bpermute
        // (what __shfl_xor compiles to) is ~500 cycles of latency at the head of every chunk
        unsigned m = mine ? (1u << (tr - r0)) : 0u;
        m |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        m |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
        m |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, false);  // row_half_mirror
        m |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, false);  // row_mirror
        m |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x142, 0xa, 0xf, false);  // row_bcast15 into rows 1 and 3
        m = (unsigned)__builtin_amdgcn_readlane((int)m, 31);
        const int run = __popc(m);
        present = (m >> n) & 1u;
        const int src = pb + __popc(m & ((1u << n) - 1u));
        if (mine && k == 0 && ch0 == 0 && tr < rows_cap) {
            row_key[tr] = (uint32_t)gid;
            row_idx[tr] = tr;
        }
        {
            // unconditional (a conditional load would keep A alive through the whole iteration: 32 registers): rows the block
            // does not hold read a slot the forward wrote (finite weights in [0, 1]: dummy_sb above) and get the scale 0
            const float4 *p4 = reinterpret_cast<const float4 *>(wt + (size_t)(present ? sb + src : dummy_sb) * 64 + k * 8);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 u = p4[4 * s4], v = p4[4 * s4 + 1];
                A[8 * s4] = u.x; A[8 * s4 + 1] = u.y; A[8 * s4 + 2] = u.z; A[8 * s4 + 3] = u.w;
                A[8 * s4 + 4] = v.x; A[8 * s4 + 5] = v.y; A[8 * s4 + 6] = v.z; A[8 * s4 + 7] = v.w;
            }
        }
        pb += run;
        fetch_keys(pb);
    };
    open_chunk(R0);  // (the first chunk's weight rows travel under the slab's conversion)
    {
        float mx = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(raw[b][s4][i]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // (exponent clamped: below 2^-112 the scale would overflow to inf -- v * inf, 0 * inf = NaN; such a column keeps 2^126)
        const float cs = (mx > 0.f && mx < 3.0e38f) ? ldexpf(1.0f, min(14 - ilogbf(mx), 126)) : 1.0f;
        inv_cs = (FIXS ? (1.0f / 32768.0f) : 1.0f) / cs;  // (FIXS: the weights' common scale leaves with the column's)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) split8x2(raw[b][s4], cs, Bh[b][s4], Bl[b][s4]);  // (packed conversions: 3 instead of 5 instructions per value, the same terms)
            __builtin_amdgcn_sched_barrier(0);  // block after block, in place: 128 raw values become 128 registers of terms
        }
    }


    const uint4 *At_l = &At[0][0][0][lane];  // this lane's 16 bytes of a (block, term, K-step): 64 uint4 per K-step
    for (int r0 = R0; r0 < R1; r0 += 32) {
        // ---- this wave's block: scale and split the chunk's weight rows, leave the terms in LDS
        {
            float rs;
            if constexpr (FIXS) {
                // a row the block does not hold: scale 0 (its lanes hold the finite weights of a slot the forward wrote)
                rs = present ? 32768.0f : 0.0f;
            } else {
                float wmx = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) wmx = fmaxf(wmx, A[i]);
                wmx = fmaxf(wmx, __shfl_xor(wmx, 32));
                const int ebits = (int)((__float_as_uint(wmx) >> 23) & 0xffu);
                const bool sane = present && ebits >= 15 && ebits <= 200;  // alpha*T lies in (4e-7, 1]
                // a row the block does not hold: scale 0 (its lanes hold the finite weights of a slot the forward wrote)
                rs = sane ? __uint_as_float((unsigned)(268 - ebits) << 23) : (present ? 1.0f : 0.0f);
                const float ri = sane ? __uint_as_float((unsigned)(ebits - 14) << 23) : 1.0f;
                if (k == 0) rinv_s[blk][n] = ri;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float a8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a8[i] = A[8 * s4 + i];
                f16x8 a0, a1, a2;
                if constexpr (TA == 3) {
                    split8x3(a8, rs, a0, a1, a2);
                    At[blk][2][s4][lane] = __builtin_bit_cast(uint4, a2);
                } else {
                    split8x2(a8, rs, a0, a1);
                }
                At[blk][0][s4][lane] = __builtin_bit_cast(uint4, a0);
                At[blk][1][s4][lane] = __builtin_bit_cast(uint4, a1);
            }
        }
        // the next chunk's rows: bookkeeping from the key window fetched a chunk ago, loads issued now: they travel under
        // this chunk's MFMAs
        if (r0 + 32 < R1) open_chunk(r0 + 32);
        __builtin_amdgcn_sched_barrier(0);
        gags_lds_barrier();  // A terms of the four blocks are in LDS
        __builtin_amdgcn_sched_barrier(0);

        // ---- the 16 (block, K-step) products of the chunk; each term's registers are reloaded for the next step as soon as
        // their last MFMA of this step is issued, so the LDS latency sits under the MFMAs in between
        f32x16 tot;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = 0.f;
        uint4 u0 = At_l[(0 * 4 + 0) * 64], u1 = At_l[(1 * 4 + 0) * 64], u2 = At_l[((TA - 1) * 4 + 0) * 64];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // FIXS: one scale for the four blocks -- the first block's products accumulate in the row total itself, the others'
            // in the scratch accumulator, added to the total as they are (the sums stay as short as they were: 64 pixels per
            // chain, four chains per row -- one chain of 256 measured 2.14e-7 against float64 where this order gives 1.7e-7)
            f32x16 acc_;
            f32x16 &acc = (FIXS && b == 0) ? tot : acc_;
            if (!(FIXS && b == 0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int nb = s4 == 3 ? b + 1 : b, ns = s4 == 3 ? 0 : s4 + 1;  // the step after this one
                const bool more = nb < 4;
                if constexpr (TA == 3) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, u2), Bh[b][s4], acc, 0, 0, 0);  // smallest terms first
                    if (more) u2 = At_l[((nb * TA + 2) * 4 + ns) * 64];
                }
                if constexpr (NM != 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, u1), Bl[b][s4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, u1), Bh[b][s4], acc, 0, 0, 0);
                if (more) u1 = At_l[((nb * TA + 1) * 4 + ns) * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, u0), Bl[b][s4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, u0), Bh[b][s4], acc, 0, 0, 0);
                if (more) u0 = At_l[((nb * TA + 0) * 4 + ns) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (FIXS) {
                if (b > 0) {
                    tot += acc_;
                    asm volatile("" : "+v"(tot));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // accumulator row r = chunk row (r & 3) + 8 (r >> 2) + 4 k: unscale by the (row, block) scale and fold into the row total
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 ri4 = *reinterpret_cast<const float4 *>(&rinv_s[b][8 * q4 + 4 * k]);
                    tot[4 * q4 + 0] = fmaf(acc[4 * q4 + 0], ri4.x, tot[4 * q4 + 0]);
                    tot[4 * q4 + 1] = fmaf(acc[4 * q4 + 1], ri4.y, tot[4 * q4 + 1]);
                    tot[4 * q4 + 2] = fmaf(acc[4 * q4 + 2], ri4.z, tot[4 * q4 + 2]);
                    tot[4 * q4 + 3] = fmaf(acc[4 * q4 + 3], ri4.w, tot[4 * q4 + 3]);
                }
                // (pins the fold here: left alone, the optimiser sinks it behind the last block, with four accumulators and four
                // sets of row scales alive -- 100 registers the kernel does not have)
                asm volatile("" : "+v"(tot));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gags_lds_barrier();  // A terms consumed: the next chunk may overwrite them
        __builtin_amdgcn_sched_barrier(0);
        // row addresses: a uniform base per row (scalar registers) + ONE per-lane offset; sixteen 64-bit pointers in
        // vector registers are what pushed the kernel over its 256
        // (byte offsets: a uniform 64-bit row base + ONE 32-bit per-lane offset is the store's own addressing mode; indexed as
        // floats the offset is widened and shifted per store -- eighteen 64-bit vector additions per chunk)
        char *base = reinterpret_cast<char *>(prow + (size_t)r0 * prow_pitch);
        unsigned voff = ((unsigned)(4 * k) * (unsigned)prow_pitch + (unsigned)chw) * 4u;
        const size_t pitch_b = (size_t)prow_pitch * 4;
        if (r0 + 32 <= R1 && r0 + 32 <= rows_cap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("" : "+v"(voff));  // (opaque per store: `base + voff` is not to be formed once as a 64-bit vector value)
                __builtin_nontemporal_store(tot[r] * inv_cs, reinterpret_cast<float *>(base + (size_t)((r & 3) + 8 * (r >> 2)) * pitch_b + voff));
            }
        } else {
            const int nrows = min(min(32, R1 - r0), rows_cap - r0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * k;
                if (row < nrows) __builtin_nontemporal_store(tot[r] * inv_cs, reinterpret_cast<float *>(base + (size_t)((r & 3) + 8 * (r >> 2)) * pitch_b + voff));
            }
        }
    }
}

