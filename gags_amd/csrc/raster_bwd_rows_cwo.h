// ---- staged: rows, channel waves over CLASS-ORDERED tile rows (round 6) -- included by raster_bwd_mfma.hip inside its namespace
// raster_bwd_rows_cw (csrc/raster_bwd_rows_cw.h) with two changes.  (1) A chunk's rows are found through sor[row] = the slot
// offset of the row in each of the tile's four blocks (gags_bwd_rowmap_ordered), not by walking each block's ascending slot
// list: no run lengths, no key windows.  (2) The tile's rows are numbered by class -- upper blocks only, mixed, lower blocks
// only -- so whole chunks at both ends of a tile have rows in two blocks only; a block without a row in the chunk is skipped
// (its split, its 12 MFMAs, its fold): 15 % of the (block, chunk) products at C3.  Same arithmetic per row, same bits.
// TA = fp16 terms of a weight (2, or 3: exact), NM = product terms (3 = a0 b0 + a0 b1 + a1 b0; 5 with TA = 3)
template <int TA, int NM>
__global__ __launch_bounds__(256, 2) void raster_bwd_rows_cwo(
    int d, int width, int height, int tile_w, int n_tiles, int ch_base, int n_slices,
    const float *__restrict__ v_render_colors, const int32_t *__restrict__ offsets, int n_isects,
    const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ trow, const float *__restrict__ wt,
    const int32_t *__restrict__ gid_s, const int32_t *__restrict__ trow_s, float *__restrict__ prow, int prow_pitch,
    uint32_t *__restrict__ row_key, int32_t *__restrict__ row_idx, int rows_cap, const int4 *__restrict__ sor)
{
    constexpr int CW = 128;
    __shared__ __attribute__((aligned(16))) uint4 At[4][TA][4][64];  // A terms in fragment order
    __shared__ __attribute__((aligned(16))) float rinv_s[4][32];    // inverse row scales of the chunk, per block

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, k = lane >> 5;
    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * n_slices);
    const int tile = gags_tile_of_order(logical / n_slices, tile_w, n_tiles / tile_w);
    const int start = offsets[tile];
    const int end = offsets[tile + 1];
    const int R0 = trow[start], R1 = trow[end];
    if (R1 == R0) return;
    const int blk = wave;  // the block whose weight rows this wave prepares
    const int cnt = blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk];
    const int sb = gags_slot_base(start, end, tile, blk);
    // a slot of this tile that the forward certainly WROTE: the first slot of its first non-empty block (R1 > R0: some
    // intersection blended, so some block holds its slot).  Rows a block does not hold read it and get the scale 0; an
    // unwritten slot (allocator garbage: the scratch is never cleared) could hold NaN / Inf bit patterns, and 0 * NaN = NaN
    int dummy_sb;
    {
        const int32_t *br = blk_rows + tile * GAGS_BLOCKS_PER_TILE;
        const int bf = br[0] > 0 ? 0 : (br[1] > 0 ? 1 : (br[2] > 0 ? 2 : 3));
        dummy_sb = gags_slot_base(start, end, tile, bf);
    }
    const int ch0 = ch_base + (logical % n_slices) * CW;
    const int chw = ch0 + 32 * wave + n;  // this lane's channel: column n of the wave's B operands and of its rows
    const int ty = tile / tile_w, tx = tile - ty * tile_w;

    // cotangent slab: B operands of the wave's 32 channels for the four blocks; K element e = 16 s + 8 k + i of a weight
    // row = pixel e >> 1 of the 8x4 half e & 1 (raster_weights.hip)
    f16x8 Bh[4][4], Bl[4][4];
    float inv_cs;  // ONE column scale per channel for the whole tile (256 pixels): the four blocks share an unscale
    {
        float raw[4][4][8];
        // pixel of K element e = 16 s + 8 k + i: row s (+ 4 for odd i) of the block, column 4 k + (i >> 1): the row is the
        // same for the whole wave, the column differs by the half-wave only -- inside the image the 128 addresses are one
        // per-lane offset plus wave-uniform terms (scalar registers / immediates); tiles cut by the image border clamp
        const bool interior = (tx + 1) * GAGS_TILE <= width && (ty + 1) * GAGS_TILE <= height;
        if (interior) {
            const float *lane0 = v_render_colors + ((size_t)(ty * GAGS_TILE) * width + tx * GAGS_TILE + 4 * k) * d + chw;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = (b >> 1) * 8 + 4 * (i & 1) + s4, col = (b & 1) * 8 + (i >> 1);  // compile-time
                        raw[b][s4][i] = lane0[((size_t)row * width + col) * d];
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
        inv_cs = 1.0f / cs;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) split8(raw[b][s4], cs, Bh[b][s4], Bl[b][s4]);
            __builtin_amdgcn_sched_barrier(0);  // block after block, in place: 128 raw values become 128 registers of terms
        }
    }

    // bookkeeping of a chunk [r0, r0 + 32): row r0 + n of the tile is, in block b, the slot at offset sor[row][b] of the block's
    // region (-1: the block does not hold the row) -- no run lengths, no key windows; `any` says which blocks hold a row of
    // the chunk at all: the others are skipped (class-ordered rows: whole chunks at a tile's ends touch two blocks only)
    auto fetch_index = [&](int r0) __attribute__((always_inline)) {
        const int row = r0 + n;
        return row < R1 ? sor[row] : make_int4(-1, -1, -1, -1);
    };
    float A[32];
    bool present = false;
    unsigned any = 0u;      // blocks with a row in the chunk whose A rows are loaded
    int kgid = 0, krow = -1;  // key of this lane's row, stored one iteration later (its gather is a round trip)
    auto open_chunk = [&](int r0, int4 jj) __attribute__((always_inline)) {
        const int jb = blk == 0 ? jj.x : (blk == 1 ? jj.y : (blk == 2 ? jj.z : jj.w));
        present = jb >= 0;
        any = (__ballot(jj.x >= 0) ? 1u : 0u) | (__ballot(jj.y >= 0) ? 2u : 0u) | (__ballot(jj.z >= 0) ? 4u : 0u) |
              (__ballot(jj.w >= 0) ? 8u : 0u);
        krow = -1;
        if (present && k == 0 && ch0 == 0 && r0 + n < rows_cap) {
            kgid = gid_s[sb + jb];
            krow = r0 + n;
        }
        {
            const float4 *p4 = reinterpret_cast<const float4 *>(wt + (size_t)(present ? sb + jb : dummy_sb) * 64 + k * 8);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 u = p4[4 * s4], v = p4[4 * s4 + 1];
                A[8 * s4] = u.x; A[8 * s4 + 1] = u.y; A[8 * s4 + 2] = u.z; A[8 * s4 + 3] = u.w;
                A[8 * s4 + 4] = v.x; A[8 * s4 + 5] = v.y; A[8 * s4 + 6] = v.z; A[8 * s4 + 7] = v.w;
            }
        }
    };
    int4 jnext = fetch_index(R0 + 32);
    open_chunk(R0, fetch_index(R0));

    const uint4 *At_l = &At[0][0][0][lane];  // this lane's 16 bytes of a (block, term, K-step): 64 uint4 per K-step
    for (int r0 = R0; r0 < R1; r0 += 32) {
        const unsigned cur_any = any;  // (of THIS chunk; open_chunk below replaces it with the next chunk's)
        const int sgid = kgid, srow = krow;
        // ---- this wave's block: scale and split the chunk's weight rows, leave the terms in LDS (nothing to do when the
        // block holds no row of the chunk: nobody reads its terms)
        if (cur_any & (1u << blk)) {
            float wmx = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) wmx = fmaxf(wmx, A[i]);
            wmx = fmaxf(wmx, __shfl_xor(wmx, 32));
            const int ebits = (int)((__float_as_uint(wmx) >> 23) & 0xffu);
            const bool sane = present && ebits >= 15 && ebits <= 200;  // alpha*T lies in (4e-7, 1]
            // a row the block does not hold: scale 0 (its lanes hold the finite weights of a slot the forward wrote)
            const float rs = sane ? __uint_as_float((unsigned)(268 - ebits) << 23) : (present ? 1.0f : 0.0f);
            const float ri = sane ? __uint_as_float((unsigned)(ebits - 14) << 23) : 1.0f;
            if (k == 0) rinv_s[blk][n] = ri;
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
        // the next chunk's rows: their slot offsets were fetched a chunk ago, the loads are issued now and travel under this
        // chunk's MFMAs; the offsets of the chunk after that are requested behind them
        if (srow >= 0) {
            row_key[srow] = (uint32_t)sgid;
            row_idx[srow] = srow;
        }
        if (r0 + 32 < R1) {
            open_chunk(r0 + 32, jnext);
            jnext = fetch_index(r0 + 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        gags_lds_barrier();  // A terms of the four blocks are in LDS
        __builtin_amdgcn_sched_barrier(0);

        // ---- the 16 (block, K-step) products of the chunk; each term's registers are reloaded for the next step as soon as
        // their last MFMA of this step is issued, so the LDS latency sits under the MFMAs in between
        f32x16 tot;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = 0.f;
        // the chunk's active blocks in order; each block's first terms are requested by the block before it (or here)
        int nxt[4];  // nxt[b] = the next active block after b, 4 = none (wave-uniform)
        int first = 4;
#pragma unroll
        for (int b = 3; b >= 0; --b) {
            nxt[b] = first;
            if (cur_any & (1u << b)) first = b;
        }
        first &= 3;  // (at least one block is active: the chunk has rows)
        uint4 u0 = At_l[((first * TA + 0) * 4 + 0) * 64], u1 = At_l[((first * TA + 1) * 4 + 0) * 64], u2 = At_l[((first * TA + (TA - 1)) * 4 + 0) * 64];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (!(cur_any & (1u << b))) continue;  // wave-uniform: no row of the chunk lies in this block
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int nb = s4 == 3 ? nxt[b] : b, ns = s4 == 3 ? 0 : s4 + 1;  // the step after this one
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
            // accumulator row r = chunk row (r & 3) + 8 (r >> 2) + 4 k: unscale by the (row, block) scale and fold into the row total
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 ri4 = *reinterpret_cast<const float4 *>(&rinv_s[b][8 * q4 + 4 * k]);
                tot[4 * q4 + 0] = fmaf(acc[4 * q4 + 0], ri4.x, tot[4 * q4 + 0]);
                tot[4 * q4 + 1] = fmaf(acc[4 * q4 + 1], ri4.y, tot[4 * q4 + 1]);
                tot[4 * q4 + 2] = fmaf(acc[4 * q4 + 2], ri4.z, tot[4 * q4 + 2]);
                tot[4 * q4 + 3] = fmaf(acc[4 * q4 + 3], ri4.w, tot[4 * q4 + 3]);
            }
            asm volatile("" : "+v"(tot));
            __builtin_amdgcn_sched_barrier(0);
        }
        gags_lds_barrier();  // A terms consumed: the next chunk may overwrite them
        __builtin_amdgcn_sched_barrier(0);
        // row addresses: a uniform base per row (scalar registers) + ONE per-lane offset; sixteen 64-bit pointers in
        // vector registers are what pushed the kernel over its 256
        float *base = prow + (size_t)r0 * prow_pitch;
        const unsigned voff = (unsigned)(4 * k) * (unsigned)prow_pitch + (unsigned)chw;
        if (r0 + 32 <= R1 && r0 + 32 <= rows_cap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) (base + (size_t)((r & 3) + 8 * (r >> 2)) * prow_pitch)[voff] = tot[r] * inv_cs;
        } else {
            const int nrows = min(min(32, R1 - r0), rows_cap - r0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * k;
                if (row < nrows) (base + (size_t)((r & 3) + 8 * (r >> 2)) * prow_pitch)[voff] = tot[r] * inv_cs;
            }
        }
    }
}

