/*
 * gags_raster.h -- C ABI of the MI355X-native GAGS feature rasterizer (libgags_hip.so).
 *
 * This is the drop-in boundary for ONE path of WHU-USI3DV/GAGS: the operator
 *     gsplat.rasterization(means, quats, scales, opacities, colors, viewmats, Ks,
 *                          backgrounds, width, height, packed=False, sh_degree, render_mode)
 * exactly as the reference calls it at gaussian_renderer/__init__.py:56-70, plus the
 * backward pass autograd runs for it under train.py:174 (`loss.backward()`).
 * Each entry point below is one stage of that operator; the stage <-> reference mapping
 *  is given per function.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *  - every pointer is a DEVICE pointer unless its name ends in _host.
 *  - the caller owns all memory (outputs and scratch); nothing here allocates or frees
 *    caller-visible memory, and there is no global mutable state (re-entrant per stream).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only
 *    enqueue work; they never synchronize, except gags_read_i32 which is the one
 *    explicit device->host readback (n_isects), mirroring gsplat's own sync -- avoidable: gags_tile_emit_cap.
 *  - return value: GAGS_OK (0) or a negative GAGS_E* code; gags_strerror() names it.
 *  - all floating point is fp32; indices are int32; intersection keys are int64
 *    (tile_id << 32 | float_bits(depth)), as in the reference's rasterizer.
 *  - tile size is fixed at 16 (GAGS_TILE), the gsplat default the reference relies on.
 */
#ifndef GAGS_RASTER_H
#define GAGS_RASTER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#pragma GCC visibility push(default)

#define GAGS_TILE 16
/* tile intersections of one view: the raster entries refuse more (GAGS_EINVAL).  4 K-step slots per intersection plus 64 per
 * tile must stay below 2^30 (32-bit byte offsets into the slot tables); 2^28 = 268 M covers a 4 M-Gaussian scene at 40+ tiles
 * per Gaussian.  The slot space is ~1 KB per intersection: the caller's memory is the practical limit. */
#define GAGS_MAX_ISECTS (1ll << 28)

#define GAGS_OK 0
#define GAGS_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported D) */
#define GAGS_ELAUNCH (-2)  /* HIP launch / runtime error (hipGetLastError != success) */
#define GAGS_ESCRATCH (-3) /* scratch buffer too small */
#define GAGS_ENODEV (-4)   /* no usable gfx950 device */

/* raster flags */
#define GAGS_BWD_COLORS_ONLY 1 /* backward: only v_colors (all the GAD flow consumes) */
#define GAGS_FWD_NO_MFMA 2     /* force the VALU kernels even when D allows the MFMA path */
#define GAGS_FEAT_F16 32       /* forward: `colors` points to an fp16 [N,D] table (BASELINE.json configs[4]: fp16 feature
                                  storage); widened exactly, same fp32 arithmetic.  Split matrix-core forward only. */
#define GAGS_FWD_F16MFMA 64    /* with GAGS_FEAT_F16 and D % 128 == 0, opt-in: the feature pass contracts on the 16-bit matrix
                                  cores (features exact, weights as fp16 head + tail: ~2^-22 per term, not bit-identical) */

#define GAGS_FWD_EXACT 2048    /* fp32 table, D >= 128: contract the feature pass with v_mfma_f32_32x32x2_f32 -- bit-for-bit the
                                  sequential fmaf chain of the algorithm (what the oracle computes) -- instead of the default
                                  16-bit matrix cores on operands split into three bf16 terms (exact operands, products to
                                  2^-23: as close to the float64 statement, 3x faster, not bit-identical to the chain) */

#define GAGS_FWD_ONLY_WEIGHTS 512   /* split forward: launch the weights pass only (alphas, last_ids, scratch) ... */
#define GAGS_FWD_ONLY_FEATURES 1024 /* ... or the feature pass only, on the scratch a GAGS_FWD_ONLY_WEIGHTS call left: lets a
                                  caller bracket the two kernels with its own events (bench.py's per-kernel roofline) */

#define GAGS_RECS_BY_GAUSSIAN 256 /* `packed` holds one record per GAUSSIAN (gags_pack_isects with packed = NULL) and the raster
                                  kernels gather it through flatten_ids themselves: no per-intersection copy (32 B x n_isects
                                  written and read back) and no gather kernel */

int gags_abi_version(void);
const char *gags_strerror(int code);
/* number of HIP devices visible, or a negative error; never throws */
int gags_device_count(void);

/* K1 + K4: fused projection and tile counting.
 * Replaces the projection stage of gsplat.rasterization reached from
 * gaussian_renderer/__init__.py:56-63 (means, quats wxyz, scales, viewmats, Ks) and the
 * first pass of its tile intersection.  viewmat: row-major 4x4 world-to-camera (the
 * reference's world_view_transform.transpose(0,1), :55); K: row-major 3x3 (:31-38).
 * Outputs: radii[N] (0 = culled), means2d[N,2], depths[N], conics[N,3],
 * tiles_per_gauss[N].  Culled Gaussians get zeros everywhere. */
int gags_project_fwd(int n, const float *means, const float *quats, const float *scales,
                     const float *viewmat, const float *K, int width, int height,
                     float eps2d, float near_plane, float far_plane, float radius_clip,
                     int32_t *radii, float *means2d, float *depths, float *conics,
                     int32_t *tiles_per_gauss, void *stream);

/* K5: inclusive prefix sum of tiles_per_gauss -> cum[N]; the grand total (n_isects) is
 * also written to total[0], or -1 when it does not fit an int32 (the caller must refuse the view).
 * scratch: gags_scan_scratch_bytes(n) bytes. */
int64_t gags_scan_scratch_bytes(int n);
int gags_cumsum_i32(int n, const int32_t *in, int32_t *cum, int32_t *total,
                    void *scratch, int64_t scratch_bytes, void *stream);

/* K5 over a permuted view: cum[i] = sum_{j <= i} in[idx[j]] (the tile counts in depth order, idx = gags_depth_order's
 * `order`): no gather kernel, no permuted copy.  in != cum. */
int gags_cumsum_gather_i32(int n, const int32_t *in, const int32_t *idx, int32_t *cum, int32_t *total,
                           void *scratch, int64_t scratch_bytes, void *stream);

/* the single device->host readback of the path (n_isects); synchronizes `stream`. */
int gags_read_i32(const int32_t *src, int32_t *dst_host, void *stream);

/* K7a (optional, faster binning): order[N] = stable argsort of the Gaussians by depth bits, and -- when tiles_ordered is
 * not NULL -- tiles_ordered[i] = tiles_per_gauss[order[i]] (prefix-sum THAT, or call gags_cumsum_gather_i32 on
 * tiles_per_gauss and order; emit in that order, and K7 only has to group by tile).
 * scratch: gags_depth_order_scratch_bytes(n) bytes. */
int64_t gags_depth_order_scratch_bytes(int n);
int gags_depth_order(int n, const float *depths, const int32_t *tiles_per_gauss, int32_t *order,
                     int32_t *tiles_ordered, void *scratch, int64_t scratch_bytes, void *stream);

/* K6: emit (key, value) per (Gaussian, tile) intersection, Gaussian after Gaussian in the order given
 * (order == NULL: by index); cum = inclusive prefix sum of the tile counts IN THAT ORDER.
 * isect_ids[n_isects] int64 = tile << 32 | depth bits, flatten_ids[n_isects] int32 = Gaussian. */
int gags_tile_emit(int n, const float *means2d, const int32_t *radii, const float *depths,
                   const int32_t *cum, const int32_t *order, int tile_w, int tile_h,
                   int64_t *isect_ids, int32_t *flatten_ids, void *stream);

/* K6 into CAPACITY-sized buffers, so that the count need not reach the host before the launches: at most `cap` pairs
 * are written, and entries [total[0], cap) become sentinel keys of a tile past the last one (they sort to the end;
 * gags_sort_pairs needs tile_bits = bit length of n_tiles for them).  total = gags_cumsum_i32's device-side total.  Run
 * sort / offsets / raster with `cap` as n_isects; read total afterwards (e.g. from a second stream, once everything is
 * enqueued): if it exceeds cap the results are garbage -- nothing was written out of bounds -- and the view must be run
 * again with a larger capacity. */
int gags_tile_emit_cap(int n, const float *means2d, const int32_t *radii, const float *depths, const int32_t *cum,
                       const int32_t *order, int tile_w, int tile_h, int64_t *isect_ids, int32_t *flatten_ids, int64_t cap,
                       const int32_t *total, void *stream);

/* K7: stable radix sort of the pairs on key bits [0, 32 + tile_bits); with depth_sorted != 0 the input
 * is already in depth order (K7a + ordered K6) and only bits [32, 32 + tile_bits) are sorted -- same
 * result (ties: depth, then Gaussian index), 2 passes instead of 6 at 1080p.
 * scratch: gags_sort_scratch_bytes(n_isects) bytes. */
int64_t gags_sort_scratch_bytes(int64_t n_isects);
int gags_sort_pairs(int64_t n_isects, int tile_bits, int depth_sorted,
                    const int64_t *keys_in, const int32_t *vals_in,
                    int64_t *keys_out, int32_t *vals_out,
                    void *scratch, int64_t scratch_bytes, void *stream);

/* K8: isect_offsets[tile_h*tile_w + 1]: first sorted index of each tile, and in the LAST entry the intersection count
 * (ABI version 2: every raster entry reads a tile's end from isect_offsets[tile + 1], none needs the count from the
 * host).  With capacity-sized inputs (gags_tile_emit_cap) pass the capacity as n_isects: the last entry is where the
 * sentinel keys begin = the true count. */
int gags_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles,
                      int32_t *isect_offsets, void *stream);

/* K8b: gather each intersection's 2-D parameters into sorted order: one 32-byte record
 * {x, y, conic a, b, c, opacity, ex, ey} per sorted intersection, (ex, ey) being the conservative
 * half-extent of the alpha >= 1/255 footprint.  The wide-D (MFMA) raster kernels stream this
 * array instead of gathering means2d/conics/opacities per tile.  packed: n_isects * 32 bytes.
 * grec (optional scratch, n * 32 bytes): when given, the record is built once per Gaussian
 * (radii > 0 only, radii may be NULL = all) and the per-intersection pass is a pure gather.
 * packed == NULL (grec required): only the per-Gaussian table is built; hand `grec` to the raster entries as
 * `packed` together with the flag GAGS_RECS_BY_GAUSSIAN. */
#define GAGS_PACKED_BYTES 32
int gags_pack_isects(int n, int64_t n_isects, const int32_t *flatten_ids, const float *means2d,
                     const float *conics, const float *opacities, const int32_t *radii, void *grec,
                     void *packed, void *stream);

/* K9 (+K11): rasterize forward, any D >= 1 in ONE pass over the sorted lists (no 32-wide
 * re-walks).  Replaces the compositing stage of gsplat.rasterization for
 * colors[N,D] / backgrounds[D] (gaussian_renderer/__init__.py:61,64).
 * backgrounds may be NULL.  Outputs render_colors[H,W,D], render_alphas[H,W],
 * last_ids[H,W] (sorted index of the last blended Gaussian per pixel).
 * Kernel choice: with `packed` given, any D >= 16 runs on the matrix cores as the split weights + feature passes
 * (128-channel slices, then 64, then 32-channel slices, the last one ragged; D = 513 is four wide slices and one lane of a
 * narrow one) when `scratch` (gags_raster_fwd_scratch_bytes) and `blk_rows` ([tile_h*tile_w*4] int32, written: slots per
 * 8x8 pixel block) are provided, else (D % 32 == 0) as one fused kernel; anything else runs the
 * VALU kernels.  The scratch and blk_rows of a split forward are what
 * gags_raster_bwd_colors_staged consumes, so keep them alive until the backward.
 * An fp32 table of exactly 16 channels (the reference's own width) is composited by the weights pass itself (round 6: one
 * kernel, bit-identical to the separate feature pass; not with GAGS_FWD_ONLY_WEIGHTS / _ONLY_FEATURES).
 * The same table with `scratch` and `blk_rows` both NULL (a render nobody differentiates) runs that kernel without writing
 * weight tiles: no scratch needed, same pixels bit for bit, ~20 % less time; there is then nothing for a backward to consume.
 */
int64_t gags_raster_fwd_scratch_bytes(int64_t n_isects, int width, int height);
int gags_raster_fwd(int d, int n, int width, int height, const float *means2d, const float *conics,
                    const float *opacities, const float *colors, const float *backgrounds,
                    const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                    const void *packed /* from gags_pack_isects, or NULL: VALU kernels only */,
                    float *render_colors, float *render_alphas, int32_t *last_ids,
                    void *scratch, int64_t scratch_bytes, int32_t *blk_rows,
                    int flags, void *stream);

/* List trimming (round 6): a heavy view's tiles stop -- every pixel saturated -- after a few hundred of their tens of thousands
 * of sorted entries, yet the split forward's scratch is ~1 KB per LIST ENTRY (C5H: 169 M entries = 173 GB).
 *   gags_raster_list_need: need[tile] (n_tiles int32, written in full) = how many leading entries of the tile's list any of its
 *       pixels reads before the tile is done: the weights pass's own walk (same records, extent test, pairs, stop rule) without
 *       its outputs.  flags: GAGS_RECS_BY_GAUSSIAN as for gags_raster_fwd.
 *   gags_trim_lists: from the INCLUSIVE prefix sums of need (gags_cumsum_i32; its total = the trimmed intersection count, which
 *       sizes flatten_out) the trimmed offsets (n_tiles + 1 entries, the last one = the count) and the trimmed id list
 *       (flatten_out NULL: offsets only).
 * Every raster entry run on (offsets_out, flatten_out, trimmed count) computes bit for bit what it computes on the full
 * lists -- same slots, weights, alphas, renders, gradients -- except that last_ids index the trimmed list:
 *   gags_trim_last_ids: last_ids back to indices of the full sorted list, in place (pixels with alpha == 0 keep their 0). */
int gags_raster_list_need(int n, int width, int height, const int32_t *isect_offsets, const int32_t *flatten_ids,
                          int64_t n_isects, const void *packed, int flags, int32_t *need, void *stream);
int gags_trim_lists(int width, int height, const int32_t *isect_offsets, const int32_t *need_cum,
                    const int32_t *flatten_ids, int32_t *offsets_out, int32_t *flatten_out, void *stream);
int gags_trim_last_ids(int width, int height, const int32_t *isect_offsets, const int32_t *offsets_trimmed,
                       const float *render_alphas, int32_t *last_ids, void *stream);

/* K10: rasterize backward (what autograd runs under train.py:174).
 * v_colors[N,D] (and, unless GAGS_BWD_COLORS_ONLY, v_opacities[N], v_means2d[N,2],
 * v_conics[N,3]) must be zero-filled by the caller; results are accumulated with
 * float atomics.  v_render_alphas may be NULL (treated as zeros). */
int gags_raster_bwd(int d, int width, int height, const float *means2d, const float *conics,
                    const float *opacities, const float *colors, const float *backgrounds,
                    const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                    const void *packed /* from gags_pack_isects, or NULL */,
                    const float *render_alphas, const int32_t *last_ids,
                    const float *v_render_colors, const float *v_render_alphas,
                    float *v_colors, float *v_opacities, float *v_means2d, float *v_conics,
                    int flags, void *stream);

/* K10, staged flavour: colours-only backward WITHOUT atomics (deterministic), 16 <= D <= 1024 (any such D).
 * Consumes the scratch + blk_rows of a split gags_raster_fwd.
 *
 * gags_bwd_rowmap: numbers the (tile, Gaussian) pairs that blended into at least one pixel -- one partial
 * gradient row each -- in sorted order: rowmap[0 .. n_isects] = exclusive prefix sum of the hit flags the
 * forward left in its scratch, followed by the row number of every K-step slot of the forward; total[0] = the
 * row count `rows` (read it with gags_read_i32).  rowmap: gags_bwd_rowmap_elems(...) int32;
 * scratch: gags_bwd_rowmap_scratch_bytes(n_isects) bytes.
 *
 * blk_rows (as gags_raster_fwd wrote it: four int32 per tile) must be 16-byte aligned: a tile's four counts are one load.
 * gags_raster_bwd_colors_staged: per tile the four pixel blocks' partial rows are merged on chip and stored
 * once per (tile, Gaussian), the rows are sorted by Gaussian and reduced; v_colors[N,D] is written in full
 * (no zero-fill needed).  scratch: gags_bwd_staged_scratch_bytes(rows, n, d) bytes.
 * stage: low 4 bits 0 = all, 1..3 = rows, sort, reduce (per-kernel timing, overlap of the zero-fill below).
 * The rows' contraction (128-channel slices) runs on the 16-bit matrix cores with fp32-equivalent split operands (fp16
 * terms after exact power-of-two scalings -- per (tile, channel) for the cotangent; for the weights, which lie in [2^-21.3, 1),
 * the constant 2^15 since round 6 --, fp32 accumulation; see the bits below for the number of terms); atomic-free and
 * bit-reproducible.  Shape since round 5: a wave per 32 channels of the slice, the four pixel blocks'
 * contributions to a tile row meet in its accumulators (csrc/raster_bwd_rows_cw.h).  bit 5 (32): the fp32 matrix
 * instructions instead (rounds 1-2's kernel).  bit 9 (512): round 4's shape (a wave per pixel block, rows merged in LDS;
 * weights as three terms, five product terms).  The default shape multiplies THREE product terms of two-term operands (one
 * fp32-level rounding per operand, the second-order term dropped: <= 3 * 2^-24 per product; 1.68e-7 of float64 against the fp32
 * matrix instructions' 1.90e-7); bit 10 (1024): the default shape with exact three-term weights and five product terms.
 * bit 6 (64): v_colors points to an fp16 [N,D] tensor (the gradient of an fp16 feature table in the table's dtype;
 * sums are formed in fp32 and rounded once).
 * bit 7 (128): v_colors arrives ZERO-FILLED and the reduce stage skips the Gaussians that blended nothing (73 % at C3)
 * instead of writing their rows of zeros -- the caller fills it on a second stream while the rows stage runs.
 * bit 8 (256), with gags_raster_bwd_colors_staged_range / _cap: the scratch holds the partial rows of THIS call's channel
 * range only -- [rows, ch_count] instead of [rows, D]; size it with gags_bwd_staged_scratch_bytes(rows, n, ch_count) -- so
 * that a wide gradient can be produced range by range (stages 1, 2, 3 for the first range, 1 and 3 for the others)
 * through a quarter of the memory: a heavy view's rows (80 M x 2 KB at D = 512) need not exist at once.
 * Returns 1 when D is not eligible. */
int64_t gags_bwd_rowmap_elems(int64_t n_isects, int width, int height);
int64_t gags_bwd_rowmap_scratch_bytes(int64_t n_isects);
int gags_bwd_rowmap(int64_t n_isects, int width, int height, const int32_t *isect_offsets, const int32_t *blk_rows,
                    const void *fwd_scratch, int64_t fwd_scratch_bytes, int32_t *rowmap, int64_t rowmap_elems,
                    int32_t *total, void *scratch, int64_t scratch_bytes, void *stream);
/* K10, geometry part at wide D (D >= 32, D % 8 == 0) after a split gags_raster_fwd: v_geo[N][8] =
 * (v_conics[3], v_means2d[2], v_opacities[1], 0, 0) per Gaussian, written in full, no atomics, deterministic.
 * The D-proportional work -- <colors[g], v_render_colors[px]> for every (slot, pixel) of the forward -- runs on the
 * matrix cores: by default the 16-bit ones with split operands (feature rows and cotangent as two fp16 terms each after
 * exact power-of-two scalings -- one per Gaussian row, one per 8x8 block -- and three product terms: as close to float64 as
 * fp32 matrix arithmetic), with
 * flags bit 5 (32) the fp32 matrix instructions (rounds 1-2's kernel); the per-pair chain uses the forward's own weights (T = weight / alpha: front-to-back quantities,
 * not 1 - render_alpha rebuilt back to front).  Together with gags_raster_bwd_colors_staged this replaces
 * gags_raster_bwd when geometry needs grad at wide D (gsplat's rasterize_to_pixels backward [EXT]).
 * backgrounds / v_render_alphas may be NULL.  row_base (optional, [tile_h*tile_w*4] int32 = exclusive prefix sum of
 * blk_rows) with n_rows = sum of blk_rows numbers the per-slot rows compactly, so that only n_rows keys are sorted;
 * NULL / -1: one row per slot of the sparse slot space (no host-side count needed, ~6x more keys).
 * flatten_ids: the sorted intersections' Gaussian ids (names, with the forward's hit flags, the rows to split).
 * scratch: gags_raster_bwd_geom_scratch_bytes(n_isects, w, h, n, d, n_rows): with row_base the dot products are numbered
 * like the rows (256 B per blended slot and 256-channel pass: 2.3 GB at C3, D = 512), without it they live in a copy of
 * the forward's sparse slot space (~1.1 KB per tile intersection and pass); + 1.5 KB per Gaussian for the split table.
 * Returns 1 when D is not eligible. */
int64_t gags_raster_bwd_geom_scratch_bytes(int64_t n_isects, int width, int height, int n, int d, int64_t n_rows);
int gags_raster_bwd_geom(int d, int n, int width, int height, const float *colors, const float *backgrounds,
                         const int32_t *isect_offsets, int64_t n_isects, const void *packed,
                         const float *v_render_colors, const float *v_render_alphas, const int32_t *blk_rows,
                         const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes,
                         float *v_geo, const int32_t *flatten_ids, const int32_t *row_base, int64_t n_rows, int flags,
                         void *stream);

/* mask[g] (n bytes, written in full) = 1 for every Gaussian that blended into at least one pixel of the view of a
 * split gags_raster_fwd (its scratch): exactly the rows of v_colors that can be non-zero.  A by-view multi-GPU step
 * exchanges only the union of these rows over the ranks (gags_amd/dist.py; SURVEY 8e "gradients are sparse in rows"). */
int gags_blended_mask(int64_t n_isects, int width, int height, int n, const int32_t *flatten_ids,
                      const void *fwd_scratch, int64_t fwd_scratch_bytes, unsigned char *mask, void *stream);
int64_t gags_bwd_staged_scratch_bytes(int64_t rows, int n, int d);
int gags_raster_bwd_colors_staged(int d, int n, int width, int height, const int32_t *isect_offsets,
                                  int64_t n_isects, const float *v_render_colors,
                                  const int32_t *blk_rows, const int32_t *rowmap, int64_t rows,
                                  const void *fwd_scratch, int64_t fwd_scratch_bytes,
                                  void *scratch, int64_t scratch_bytes, float *v_colors, int stage,
                                  void *stream);
/* The same for channels [ch_begin, ch_begin + ch_count) only (whole 128- / 64- / 32-channel slices, as D % 128 / 64
 * allows; the last range may end at D): v_colors[:, ch_begin : ch_begin + ch_count] is written.  A multi-GPU by-view
 * step calls it range after range -- stage 1, 2, 3 for the range that starts at 0 (it writes the row -> Gaussian
 * keys the sort needs), stage 1 and 3 for the others -- and exchanges each range while the next is computed
 * (gags_amd/dist.py).  Same scratch for every range of a view. */
int gags_raster_bwd_colors_staged_range(int d, int n, int width, int height, const int32_t *isect_offsets,
                                        int64_t n_isects, const float *v_render_colors,
                                        const int32_t *blk_rows, const int32_t *rowmap, int64_t rows,
                                        const void *fwd_scratch, int64_t fwd_scratch_bytes,
                                        void *scratch, int64_t scratch_bytes, float *v_colors, int stage,
                                        int ch_begin, int ch_count, void *stream);
/* The same with `rows` as a CAPACITY and the true row count on the device (rows_dev = gags_bwd_rowmap's total; NULL:
 * rows is exact): rows past the capacity are dropped, the keys between the count and the capacity become sentinels.  Read
 * the count afterwards; if it exceeds the capacity run the backward again. */
int gags_raster_bwd_colors_staged_cap(int d, int n, int width, int height, const int32_t *isect_offsets, int64_t n_isects,
                                      const float *v_render_colors, const int32_t *blk_rows, const int32_t *rowmap,
                                      int64_t rows, const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch,
                                      int64_t scratch_bytes, float *v_colors, int stage, int ch_begin, int ch_count,
                                      const int32_t *rows_dev, void *stream);

/* gags_raster_bwd_colors_staged_range whose reduce stage ALSO writes the rows a by-view multi-GPU step exchanges
 * (gags_amd/dist.py): wire[wire_pos[g], :] = the range's gradient row of Gaussian g for every g with
 * wire_pos[g] >= 0 -- wire is a dense fp32 [union rows, ch_count] block (ch_count % 4 == 0), wire_pos the inverse of the
 * union's row list (gags_compact_mask_pos).  Every row of the block is written (a union row this view did not touch gets
 * zeros), so the block needs no clearing and no pack kernel re-reads the gradient.  v_colors is written as always -- or,
 * with keep_prev / keep_cur (both or neither; see gags_raster_bwd_colors_staged_keep below), as a persistent buffer whose
 * rows count as written when they have partial rows OR lie in the exchanged block (the caller writes the ranks' sum there). */
int gags_raster_bwd_colors_staged_wire(int d, int n, int width, int height, const int32_t *isect_offsets, int64_t n_isects,
                                       const float *v_render_colors, const int32_t *blk_rows, const int32_t *rowmap,
                                       int64_t rows, const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch,
                                       int64_t scratch_bytes, float *v_colors, int stage, int ch_begin, int ch_count,
                                       const int32_t *wire_pos, float *wire, const uint8_t *keep_prev, uint8_t *keep_cur,
                                       void *stream);

/* gags_raster_bwd_colors_staged_range into a PERSISTENT gradient buffer (round 6): v_colors is a buffer the caller keeps
 * between steps, all zeros except the rows the previous call wrote -- keep_prev[g] != 0 (n bytes; all zeros for a freshly
 * zeroed buffer).  The reduce stage writes the rows that have partial rows now, re-zeroes the rows that had some last time
 * and none now, touches nothing else (the rows of Gaussians that blend nothing -- 73 % at C3, 2.2 GB of zeros per step --
 * are never written again), and leaves keep_cur[g] (n bytes, another array) for the next call.  Same values as the plain
 * entry, bit for bit.  The caller must know that nobody else wrote the buffer in between (gags_amd/rasterization.py checks
 * the storage's reference count and version counter and falls back to the plain entry otherwise). */
int gags_raster_bwd_colors_staged_keep(int d, int n, int width, int height, const int32_t *isect_offsets, int64_t n_isects,
                                       const float *v_render_colors, const int32_t *blk_rows, const int32_t *rowmap,
                                       int64_t rows, const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch,
                                       int64_t scratch_bytes, float *v_colors, int stage, int ch_begin, int ch_count,
                                       const uint8_t *keep_prev, uint8_t *keep_cur, void *stream);

/* Diagnostics (roofline model): counts[0] += (pixel,Gaussian) pairs evaluated
 * before each pixel's stop, counts[1] += pairs blended.  counts[2] int64, zeroed by caller. */
int gags_raster_stats(int width, int height, const float *means2d, const float *conics,
                      const float *opacities, const int32_t *isect_offsets, const int32_t *flatten_ids,
                      int64_t n_isects, int64_t *counts, void *stream);

/* K1 + K4 on the STORED parameters (R2 folded into R4; SURVEY 8a R2: "4 elementwise kernels/iter over N (fusable into
 * projection)"): rotation[N,4] un-normalised, scaling_log[N,3], opacity_logit[N] exactly as scene/gaussian_model.py:48-61
 * holds them; the kernel applies the getters of :116-139 -- exp (* scaling_modifier, gaussian_renderer/__init__.py:41),
 * F.normalize, sigmoid -- bit for bit as torch evaluates them (tools/micro/actprobe.py), then projects as gags_project_fwd.
 * Extra outputs: opacities[N] (activated: what the raster kernels read); quats_act[N,4] / scales_act[N,3] (activated; NULL
 * = not wanted); grec (N * GAGS_PACKED_BYTES bytes, or NULL): the per-Gaussian record table of K8b for every Gaussian with
 * radii > 0 -- the bytes gags_pack_isects(packed = NULL) would write -- so that the binning needs no record kernel.  The backward takes the gradients back to the stored parameters, v_opacities[N] (from the rasterizer;
 * NULL = zeros) included; v_opacity_logit may be NULL. */
int gags_project_fwd_raw(int n, const float *means, const float *rotation, const float *scaling_log,
                         const float *opacity_logit, float scaling_modifier, const float *viewmat, const float *K,
                         int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                         int32_t *radii, float *means2d, float *depths, float *conics, int32_t *tiles_per_gauss,
                         float *opacities, float *quats_act, float *scales_act, void *grec, void *stream);
int gags_project_bwd_raw(int n, const float *means, const float *rotation, const float *scaling_log,
                         const float *opacity_logit, float scaling_modifier, const float *viewmat, const float *K,
                         int width, int height, float eps2d, const int32_t *radii, const float *v_means2d,
                         const float *v_depths, const float *v_conics, const float *v_opacities, float *v_means,
                         float *v_rotation, float *v_scaling_log, float *v_opacity_logit, void *stream);

/* K2: projection backward: chain rule of gags_project_fwd for Gaussians with radii>0.
 * Inputs v_means2d[N,2], v_depths[N] (may be NULL), v_conics[N,3];
 * outputs (overwritten) v_means[N,3], v_quats[N,4], v_scales[N,3]. */
int gags_project_bwd(int n, const float *means, const float *quats, const float *scales,
                     const float *viewmat, const float *K, int width, int height, float eps2d,
                     const int32_t *radii, const float *conics,
                     const float *v_means2d, const float *v_depths, const float *v_conics,
                     float *v_means, float *v_quats, float *v_scales, void *stream);

/* K3: spherical-harmonics colour, used when feature_mode=False and no override colour
 * (gaussian_renderer/__init__.py:51-53).  coeffs[N,kc,3], campos[3] (device),
 * out[N,3] = max(SH(dir) + 0.5, 0) where radii>0, zeros elsewhere.  Basis and signs as
 * utils/sh_utils.py:57-112. */
int gags_sh_fwd(int n, int kc, int degree, const float *means, const float *campos,
                const float *coeffs, const int32_t *radii, float *out, void *stream);
/* v_coeffs[N,kc,3] overwritten; directions are treated as constants (geometry frozen). */
int gags_sh_bwd(int n, int kc, int degree, const float *means, const float *campos,
                const int32_t *radii, const float *colors_out, const float *v_out,
                float *v_coeffs, void *stream);

/* K3 backward, view-direction part: v_means[n,3] = d loss / d means through dir = normalize(mean - campos) (gsplat
 * propagates this gradient; it matters when the SH colour branch runs with trainable positions, train.py:142 without
 * --feature_mode).  coeffs [n,kc,3], colors_out = gags_sh_fwd's output (clamp mask), v_out [n,3]. */
int gags_sh_bwd_dirs(int n, int kc, int degree, const float *means, const float *campos, const float *coeffs,
                     const int32_t *radii, const float *colors_out, const float *v_out, float *v_means, void *stream);

/* K12: expected-depth normalisation of the last channel: c[...,d-1] /= max(alpha,1e-10)
 * (render_mode "RGB+ED", the only consumer is render.py:118,127-133). */
int gags_ed_normalize(int64_t n_pix, int d, float *render_colors, const float *render_alphas,
                      void *stream);

/* R9: one Adam step of the feature parameter, in place, torch.optim.Adam semantics without weight decay
 * or amsgrad (the reference's optimizer: scene/gaussian_model.py:192-208, eps = 1e-15; stepped at
 * train.py:221-223).  `step` is the 1-based step count; all four arrays have `numel` floats, 16-B aligned. */
int gags_adam_step(int64_t numel, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                   double lr, double beta1, double beta2, double eps, int step, void *stream);

/* Multi-GPU by-view step (north_star "RCCL all-reduce of feature/geometry gradients"): the exchange itself is
 * torch.distributed over RCCL; these two kernels move the rows that can be non-zero between the gradient [N, d] and
 * the dense wire block [n_rows, cw] of one channel range [c0, c0 + cw).  idx: int64 row numbers (device; NULL = all rows,
 * n_rows = N).  Element types: 0 fp32, 1 fp16, 2 bf16 (wire only).
 *   gags_pack_rows  : wire[r, :] = grad[idx[r], c0 : c0 + cw]                     (idx[r] < 0: a padding row, zeros)
 *   gags_unpack_rows: local == NULL: grad[idx[r], c0 : c0 + cw]  = wire[r, :]
 *                     local != NULL: grad[idx[r], c0 : c0 + cw] += wire[r, :] - local[r, :]   (local: same type as wire)
 *                     (idx[r] < 0: skipped)
 *   gags_compact_mask: the row list itself, on the device: idx[0 .. count) = ascending r with mask[r] != 0 (the union of
 *                     the ranks' blended-Gaussian masks, gags_blended_mask), idx[count .. cap) = -1, count[0] = number of
 *                     set rows (may exceed cap: the surplus is not stored).  Replaces a host-synchronising torch.nonzero. */
int64_t gags_compact_mask_scratch_bytes(int n);
int gags_compact_mask(int n, const uint8_t *mask, int64_t cap, int64_t *idx, int32_t *count, void *scratch,
                      int64_t scratch_bytes, void *stream);
/*   gags_compact_mask_pos: the same, and the inverse on the way: pos[r] (n int32, written in full) = the position of row r
 *                     in idx, -1 for rows that are not set or whose position is >= cap (gags_raster_bwd_colors_staged_wire
 *                     writes the exchanged rows through it). */
int gags_compact_mask_pos(int n, const uint8_t *mask, int64_t cap, int64_t *idx, int32_t *pos, int32_t *count, void *scratch,
                          int64_t scratch_bytes, void *stream);
int gags_pack_rows(int64_t n_rows, const int64_t *idx, const void *grad, int grad_type, int d, int c0, int cw,
                   void *wire, int wire_type, void *stream);
int gags_unpack_rows(int64_t n_rows, const int64_t *idx, const void *wire, int wire_type, const void *local,
                     void *grad, int grad_type, int d, int c0, int cw, void *stream);

/* Harness helper (the build's own synthetic step): out[0] = <x, y> over `numel` floats,
 * the terminal loss `(render * G).sum()` of bench.py; reproducible (fixed grid and order).
 * scratch: gags_dot_scratch_bytes() bytes; x, y 16-B aligned. */
int64_t gags_dot_scratch_bytes(void);
int gags_dot_f32(int64_t numel, const float *x, const float *y, float *out, void *scratch,
                 int64_t scratch_bytes, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* GAGS_RASTER_H */
