/*
 * gags_oracle.c -- CPU ORACLE for the GAGS feature-rasterization hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load this library; the shipped path
 * (gags_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: **parity unpinned**.  The arithmetic of this path lives in the
 * third-party pip package `gsplat`, which the reference imports
 * (/root/reference/gaussian_renderer/__init__.py:17,56-70) but does not vendor and does
 * not pin (/root/reference/environment.yml:26).  Its source is absent from
 * /root/reference and the package is not installable here (no network).  This file
 * therefore restates the *published* gsplat-1.4-style algorithm (classic rasterize mode,
 * packed=False, one camera) as declared in SURVEY.md Appendix A (A1-A12), and anchors on
 * the reference's own call site: argument construction and defaults at
 * gaussian_renderer/__init__.py:27-70 (eps2d=0.3, near=0.01, far=1e10, radius_clip=0,
 * tile_size=16 are the gsplat defaults the reference relies on by not passing them).
 * The parts of the path that DO live in the reference tree are pinned by golden vectors
 * generated from the reference's Python (tests/golden/make_golden.py):
 *   - quaternion -> rotation convention  utils/general_utils.py:78-98
 *   - SH basis / signs                   utils/sh_utils.py:57-112
 *   - view matrix / intrinsics           utils/graphics_utils.py:38-49,
 *                                        gaussian_renderer/__init__.py:27-38
 *
 * Numerical contract shared with the HIP kernels (so index tensors and forward renders
 * can be compared BIT-EXACTLY): fp32 everywhere, no FMA contraction except where
 * fmaf() is written, IEEE division / sqrt, and exp(-sigma) evaluated by the explicit
 * polynomial orc_exp_neg() below instead of a libm / hardware exp.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_TILE 16
#define ORC_ALPHA_MAX 0.999f
#define ORC_ALPHA_MIN (1.0f / 255.0f)
#define ORC_T_STOP 1e-4f

int orc_version(void) { return 1; }

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* exp(-sigma): 2^(t), t = -sigma*log2(e), split t = n + f with n = rint(t), f in [-.5,.5],
 * 2^f by a degree-6 polynomial (max rel. error 1.4 ulp), scaled by 2^n with ldexp.
 * Stands in for gsplat's __expf; same few-ulp accuracy class. */
static inline float orc_exp_neg(float sigma)
{
    float t = sigma * -1.44269504088896341f;
    t = fmaxf(t, -125.0f);
    float n = rintf(t);
    float f = t - n;
    float p = 0x1.444p-13f;
    p = fmaf(p, f, 0x1.5f48cp-10f);
    p = fmaf(p, f, 0x1.3b2a1cp-7f);
    p = fmaf(p, f, 0x1.c6aeccp-5f);
    p = fmaf(p, f, 0x1.ebfbep-3f);
    p = fmaf(p, f, 0x1.62e43p-1f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

float orc_exp_neg_scalar(float sigma) { return orc_exp_neg(sigma); }

/* ------------------------------------------------------------------------------------
 * K1  fused projection forward.
 * quaternion convention wxyz as /root/reference/utils/general_utils.py:78-98;
 * viewmat is the row-major world-to-camera matrix the reference passes as
 * world_view_transform.transpose(0,1) (gaussian_renderer/__init__.py:55);
 * K = [[fx,0,cx],[0,fy,cy],[0,0,1]] (gaussian_renderer/__init__.py:31-38).
 * Culled Gaussians get radii=0 and zeros in means2d/depths/conics.
 * ---------------------------------------------------------------------------------- */
void orc_project_fwd(int N, const float *means, const float *quats, const float *scales,
                     const float *viewmat, const float *Kmat, int width, int height,
                     float eps2d, float near_plane, float far_plane, float radius_clip,
                     int32_t *radii, float *means2d, float *depths, float *conics)
{
    const float R00 = viewmat[0], R01 = viewmat[1], R02 = viewmat[2], t0 = viewmat[3];
    const float R10 = viewmat[4], R11 = viewmat[5], R12 = viewmat[6], t1 = viewmat[7];
    const float R20 = viewmat[8], R21 = viewmat[9], R22 = viewmat[10], t2 = viewmat[11];
    const float fx = Kmat[0], cx = Kmat[2], fy = Kmat[4], cy = Kmat[5];
    const float fw = (float)width, fh = (float)height;
    const float tan_fovx = 0.5f * fw / fx;
    const float tan_fovy = 0.5f * fh / fy;
    const float lim_x_pos = (fw - cx) / fx + 0.3f * tan_fovx;
    const float lim_x_neg = cx / fx + 0.3f * tan_fovx;
    const float lim_y_pos = (fh - cy) / fy + 0.3f * tan_fovy;
    const float lim_y_neg = cy / fy + 0.3f * tan_fovy;

#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        means2d[2 * i] = 0.f; means2d[2 * i + 1] = 0.f;
        depths[i] = 0.f;
        conics[3 * i] = 0.f; conics[3 * i + 1] = 0.f; conics[3 * i + 2] = 0.f;

        const float px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
        /* A2: camera space */
        const float x = ((R00 * px + R01 * py) + R02 * pz) + t0;
        const float y = ((R10 * px + R11 * py) + R12 * pz) + t1;
        const float z = ((R20 * px + R21 * py) + R22 * pz) + t2;
        if (z < near_plane || z > far_plane) continue;

        /* A1: covariance from quaternion (wxyz) and scale */
        float qw = quats[4 * i], qx = quats[4 * i + 1], qy = quats[4 * i + 2], qz = quats[4 * i + 3];
        const float inv_norm = 1.0f / sqrtf(((qx * qx + qy * qy) + qz * qz) + qw * qw);
        qw *= inv_norm; qx *= inv_norm; qy *= inv_norm; qz *= inv_norm;
        const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
        const float r00 = 1.f - 2.f * (y2 + z2), r01 = 2.f * (xy - wz), r02 = 2.f * (xz + wy);
        const float r10 = 2.f * (xy + wz), r11 = 1.f - 2.f * (x2 + z2), r12 = 2.f * (yz - wx);
        const float r20 = 2.f * (xz - wy), r21 = 2.f * (yz + wx), r22 = 1.f - 2.f * (x2 + y2);
        const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
        const float m00 = r00 * s0, m01 = r01 * s1, m02 = r02 * s2;
        const float m10 = r10 * s0, m11 = r11 * s1, m12 = r12 * s2;
        const float m20 = r20 * s0, m21 = r21 * s1, m22 = r22 * s2;
        /* Sigma = M M^T (symmetric) */
        const float c00 = (m00 * m00 + m01 * m01) + m02 * m02;
        const float c01 = (m00 * m10 + m01 * m11) + m02 * m12;
        const float c02 = (m00 * m20 + m01 * m21) + m02 * m22;
        const float c11 = (m10 * m10 + m11 * m11) + m12 * m12;
        const float c12 = (m10 * m20 + m11 * m21) + m12 * m22;
        const float c22 = (m20 * m20 + m21 * m21) + m22 * m22;
        /* T = R_cw * Sigma ; Sigma_c = T * R_cw^T (only the 6 unique entries) */
        const float a00 = (R00 * c00 + R01 * c01) + R02 * c02;
        const float a01 = (R00 * c01 + R01 * c11) + R02 * c12;
        const float a02 = (R00 * c02 + R01 * c12) + R02 * c22;
        const float a10 = (R10 * c00 + R11 * c01) + R12 * c02;
        const float a11 = (R10 * c01 + R11 * c11) + R12 * c12;
        const float a12 = (R10 * c02 + R11 * c12) + R12 * c22;
        const float a20 = (R20 * c00 + R21 * c01) + R22 * c02;
        const float a21 = (R20 * c01 + R21 * c11) + R22 * c12;
        const float a22 = (R20 * c02 + R21 * c12) + R22 * c22;
        const float v00 = (a00 * R00 + a01 * R01) + a02 * R02;
        const float v01 = (a00 * R10 + a01 * R11) + a02 * R12;
        const float v02 = (a00 * R20 + a01 * R21) + a02 * R22;
        const float v11 = (a10 * R10 + a11 * R11) + a12 * R12;
        const float v12 = (a10 * R20 + a11 * R21) + a12 * R22;
        const float v22 = (a20 * R20 + a21 * R21) + a22 * R22;

        /* A3: perspective projection with frustum-clamped Jacobian */
        const float rz = 1.f / z;
        const float rz2 = rz * rz;
        const float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
        const float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
        const float j00 = fx * rz, j02 = -fx * tx * rz2;
        const float j11 = fy * rz, j12 = -fy * ty * rz2;
        /* B = J * Sigma_c (2x3), cov2d = B * J^T */
        const float b00 = j00 * v00 + j02 * v02;
        const float b01 = j00 * v01 + j02 * v12;
        const float b02 = j00 * v02 + j02 * v22;
        const float b11 = j11 * v11 + j12 * v12;
        const float b12 = j11 * v12 + j12 * v22;
        float s00 = b00 * j00 + b02 * j02;
        const float s01 = b01 * j11 + b02 * j12;
        float s11 = b11 * j11 + b12 * j12;
        const float m2x = fx * x * rz + cx;
        const float m2y = fy * y * rz + cy;

        /* A4: blur + conic */
        s00 += eps2d; s11 += eps2d;
        const float det = s00 * s11 - s01 * s01;
        if (det <= 0.f) continue;
        const float inv_det = 1.f / det;
        const float ca = s11 * inv_det, cb = -s01 * inv_det, cc = s00 * inv_det;

        /* A5: radius + screen cull */
        const float hb = 0.5f * (s00 + s11);
        const float v1 = hb + sqrtf(fmaxf(0.01f, hb * hb - det));
        const float radius = ceilf(3.f * sqrtf(v1));
        if (radius <= radius_clip) continue;
        if (m2x + radius <= 0.f || m2x - radius >= fw || m2y + radius <= 0.f || m2y - radius >= fh)
            continue;

        radii[i] = (int32_t)radius;
        means2d[2 * i] = m2x; means2d[2 * i + 1] = m2y;
        depths[i] = z;
        conics[3 * i] = ca; conics[3 * i + 1] = cb; conics[3 * i + 2] = cc;
    }
}

/* ------------------------------------------------------------------------------------
 * K4-K8  tile binning.
 * ---------------------------------------------------------------------------------- */
static inline void orc_tile_aabb(float mx, float my, int32_t radius, int tile_w, int tile_h,
                                 int *xmin, int *xmax, int *ymin, int *ymax)
{
    const float tr = (float)radius / (float)ORC_TILE;
    const float tx = mx / (float)ORC_TILE, ty = my / (float)ORC_TILE;
    *xmin = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tile_w);
    *xmax = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tile_w);
    *ymin = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)tile_h);
    *ymax = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)tile_h);
}

void orc_tile_count(int N, const float *means2d, const int32_t *radii, int tile_w, int tile_h,
                    int32_t *tiles_per_gauss)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) { tiles_per_gauss[i] = 0; continue; }
        int x0, x1, y0, y1;
        orc_tile_aabb(means2d[2 * i], means2d[2 * i + 1], radii[i], tile_w, tile_h, &x0, &x1, &y0, &y1);
        tiles_per_gauss[i] = (y1 - y0) * (x1 - x0);
    }
}

/* cum = inclusive cumsum(tiles_per_gauss) (int64); returns n_isects */
int64_t orc_cumsum(int N, const int32_t *tiles_per_gauss, int64_t *cum)
{
    int64_t s = 0;
    for (int i = 0; i < N; ++i) { s += tiles_per_gauss[i]; cum[i] = s; }
    return s;
}

/* key = tile_id << 32 | float_bits(depth)   (camera id is always 0: one camera) */
void orc_tile_emit(int N, const float *means2d, const int32_t *radii, const float *depths,
                   const int64_t *cum, int tile_w, int tile_h, int64_t *isect_ids, int32_t *flatten_ids)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        int x0, x1, y0, y1;
        orc_tile_aabb(means2d[2 * i], means2d[2 * i + 1], radii[i], tile_w, tile_h, &x0, &x1, &y0, &y1);
        int64_t cur = (i == 0) ? 0 : cum[i - 1];
        uint32_t dbits; memcpy(&dbits, &depths[i], 4);
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                const int64_t tile_id = (int64_t)ty * tile_w + tx;
                isect_ids[cur] = (tile_id << 32) | (int64_t)dbits;
                flatten_ids[cur] = i;
                ++cur;
            }
    }
}

/* stable LSD radix sort of (key,value) pairs on key bits [0,nbits) */
void orc_sort_pairs(int64_t n, int nbits, const int64_t *keys_in, const int32_t *vals_in,
                    int64_t *keys_out, int32_t *vals_out)
{
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    uint64_t *kb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    int32_t *va = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int32_t *vb = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    memcpy(ka, keys_in, sizeof(uint64_t) * (size_t)n);
    memcpy(va, vals_in, sizeof(int32_t) * (size_t)n);
    for (int shift = 0; shift < nbits; shift += 8) {
        size_t hist[257]; memset(hist, 0, sizeof(hist));
        for (int64_t i = 0; i < n; ++i) hist[((ka[i] >> shift) & 0xff) + 1]++;
        for (int d = 0; d < 256; ++d) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < n; ++i) {
            const size_t p = hist[(ka[i] >> shift) & 0xff]++;
            kb[p] = ka[i]; vb[p] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        int32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * (size_t)n);
    memcpy(vals_out, va, sizeof(int32_t) * (size_t)n);
    free(ka); free(kb); free(va); free(vb);
}

/* offsets[t] = first sorted index whose tile id is >= t  (== next tile's start when empty) */
void orc_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles, int32_t *offsets)
{
    int64_t i = 0;
    for (int t = 0; t < n_tiles; ++t) {
        while (i < n_isects && (int64_t)(((uint64_t)sorted_ids[i]) >> 32) < t) ++i;
        offsets[t] = (int32_t)i;
    }
}

/* ------------------------------------------------------------------------------------
 * K9  rasterize forward.  colors [N,D], backgrounds [D] or NULL.
 * Outputs: render_colors [H,W,D], render_alphas [H,W], last_ids [H,W].
 * tile_begin/tile_step select a subset of tiles (bench sampling); tests pass 0,1.
 * Returns, through *n_eval / *n_blend (may be NULL), the number of (pixel,Gaussian)
 * pairs evaluated before the stop and the number actually blended.
 * ---------------------------------------------------------------------------------- */
void orc_raster_fwd(int D, int width, int height, int tile_w, int tile_h,
                    const float *means2d, const float *conics, const float *opacities,
                    const float *colors, const float *backgrounds,
                    const int32_t *tile_offsets, const int32_t *flatten_ids, int64_t n_isects,
                    int tile_begin, int tile_step,
                    float *render_colors, float *render_alphas, int32_t *last_ids,
                    int64_t *n_eval, int64_t *n_blend)
{
    const int n_tiles = tile_w * tile_h;
    int64_t tot_eval = 0, tot_blend = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tot_eval, tot_blend)
    for (int tile = tile_begin; tile < n_tiles; tile += tile_step) {
        const int ty = tile / tile_w, tx = tile % tile_w;
        const int64_t start = tile_offsets[tile];
        const int64_t end = (tile == n_tiles - 1) ? n_isects : tile_offsets[tile + 1];
        float *acc = (float *)malloc(sizeof(float) * (size_t)D);
        for (int ly = 0; ly < ORC_TILE; ++ly)
            for (int lx = 0; lx < ORC_TILE; ++lx) {
                const int i = ty * ORC_TILE + ly, j = tx * ORC_TILE + lx;
                if (i >= height || j >= width) continue;
                const float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T = 1.0f;
                int32_t cur = 0;
                for (int k = 0; k < D; ++k) acc[k] = 0.f;
                for (int64_t s = start; s < end; ++s) {
                    const int32_t g = flatten_ids[s];
                    const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    const float alpha = fminf(ORC_ALPHA_MAX, opacities[g] * orc_exp_neg(sigma));
                    ++tot_eval;
                    if (sigma < 0.f || alpha < ORC_ALPHA_MIN) continue;
                    const float next_T = T * (1.0f - alpha);
                    if (next_T <= ORC_T_STOP) break;
                    const float vis = alpha * T;
                    const float *c = colors + (size_t)g * D;
                    for (int k = 0; k < D; ++k) acc[k] = fmaf(c[k], vis, acc[k]);
                    ++tot_blend;
                    cur = (int32_t)s;
                    T = next_T;
                }
                const size_t pix = (size_t)i * width + j;
                render_alphas[pix] = 1.0f - T;
                float *o = render_colors + pix * D;
                if (backgrounds) for (int k = 0; k < D; ++k) o[k] = fmaf(T, backgrounds[k], acc[k]);
                else for (int k = 0; k < D; ++k) o[k] = acc[k];
                last_ids[pix] = cur;
            }
        free(acc);
    }
    if (n_eval) *n_eval = tot_eval;
    if (n_blend) *n_blend = tot_blend;
}

/* ------------------------------------------------------------------------------------
 * K9 with the COLOUR SUMS in double: the same fp32 alpha / transmittance chain, skip and stop
 * decisions as orc_raster_fwd (so the same weights, bit for bit), but every product c * (alpha T)
 * is formed and accumulated in float64 and the result is returned in float64.  This is the
 * yardstick for contraction kernels that are fp32-equivalent without being bit-identical to the
 * sequential fmaf chain (the 16-bit matrix-core feature pass): "how far is each kernel from the
 * exact sum of these fp32 products".  Writes render_colors64 only (tile subset as above).
 * ---------------------------------------------------------------------------------- */
void orc_raster_fwd_acc64(int D, int width, int height, int tile_w, int tile_h,
                          const float *means2d, const float *conics, const float *opacities,
                          const float *colors, const float *backgrounds,
                          const int32_t *tile_offsets, const int32_t *flatten_ids, int64_t n_isects,
                          int tile_begin, int tile_step, double *render_colors64)
{
    const int n_tiles = tile_w * tile_h;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = tile_begin; tile < n_tiles; tile += tile_step) {
        const int ty = tile / tile_w, tx = tile % tile_w;
        const int64_t start = tile_offsets[tile];
        const int64_t end = (tile == n_tiles - 1) ? n_isects : tile_offsets[tile + 1];
        double *acc = (double *)malloc(sizeof(double) * (size_t)D);
        for (int ly = 0; ly < ORC_TILE; ++ly)
            for (int lx = 0; lx < ORC_TILE; ++lx) {
                const int i = ty * ORC_TILE + ly, j = tx * ORC_TILE + lx;
                if (i >= height || j >= width) continue;
                const float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T = 1.0f;
                for (int k = 0; k < D; ++k) acc[k] = 0.0;
                for (int64_t s = start; s < end; ++s) {
                    const int32_t g = flatten_ids[s];
                    const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    const float alpha = fminf(ORC_ALPHA_MAX, opacities[g] * orc_exp_neg(sigma));
                    if (sigma < 0.f || alpha < ORC_ALPHA_MIN) continue;
                    const float next_T = T * (1.0f - alpha);
                    if (next_T <= ORC_T_STOP) break;
                    const double vis = (double)(alpha * T);
                    const float *c = colors + (size_t)g * D;
                    for (int k = 0; k < D; ++k) acc[k] += (double)c[k] * vis;
                    T = next_T;
                }
                double *o = render_colors64 + ((size_t)i * width + j) * D;
                for (int k = 0; k < D; ++k) o[k] = acc[k] + (backgrounds ? (double)T * (double)backgrounds[k] : 0.0);
            }
        free(acc);
    }
}

/* ------------------------------------------------------------------------------------
 * K10  rasterize backward, back-to-front replay from last_ids.
 * v_colors [N,D], v_opacities [N], v_means2d [N,2], v_conics [N,3] must be zeroed by the
 * caller; any of the three geometry outputs may be NULL together (colours-only mode,
 * which is all the GAD flow consumes: scene/gaussian_model.py:192-208).
 * Per-tile sums are taken in double, then added to the fp32 outputs.
 * ---------------------------------------------------------------------------------- */
void orc_raster_bwd(int D, int width, int height, int tile_w, int tile_h,
                    const float *means2d, const float *conics, const float *opacities,
                    const float *colors, const float *backgrounds,
                    const int32_t *tile_offsets, const int32_t *flatten_ids, int64_t n_isects,
                    const float *render_alphas, const int32_t *last_ids,
                    const float *v_render_colors, const float *v_render_alphas,
                    int tile_begin, int tile_step,
                    float *v_colors, float *v_opacities, float *v_means2d, float *v_conics)
{
    const int n_tiles = tile_w * tile_h;
    const int geom = (v_opacities != NULL);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = tile_begin; tile < n_tiles; tile += tile_step) {
        const int ty = tile / tile_w, tx = tile % tile_w;
        const int64_t start = tile_offsets[tile];
        const int64_t end = (tile == n_tiles - 1) ? n_isects : tile_offsets[tile + 1];
        const int64_t cnt = end - start;
        if (cnt <= 0) continue;
        double *lc = (double *)calloc((size_t)cnt * D, sizeof(double));
        double *lg = geom ? (double *)calloc((size_t)cnt * 6, sizeof(double)) : NULL;
        float *buf = (float *)malloc(sizeof(float) * (size_t)D);
        for (int ly = 0; ly < ORC_TILE; ++ly)
            for (int lx = 0; lx < ORC_TILE; ++lx) {
                const int i = ty * ORC_TILE + ly, j = tx * ORC_TILE + lx;
                if (i >= height || j >= width) continue;
                const size_t pix = (size_t)i * width + j;
                const float px = (float)j + 0.5f, py = (float)i + 0.5f;
                const float T_final = 1.0f - render_alphas[pix];
                float T = T_final;
                const float *vc = v_render_colors + pix * D;
                const float va = v_render_alphas ? v_render_alphas[pix] : 0.f;
                float bg_dot = 0.f;
                if (backgrounds) for (int k = 0; k < D; ++k) bg_dot += backgrounds[k] * vc[k];
                for (int k = 0; k < D; ++k) buf[k] = 0.f;
                const int64_t bin_final = last_ids[pix];
                for (int64_t s = (bin_final < end - 1 ? bin_final : end - 1); s >= start; --s) {
                    const int32_t g = flatten_ids[s];
                    const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const float opac = opacities[g];
                    const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    const float vis = orc_exp_neg(sigma);
                    const float alpha = fminf(ORC_ALPHA_MAX, opac * vis);
                    if (sigma < 0.f || alpha < ORC_ALPHA_MIN) continue;
                    const float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    const float fac = alpha * T;
                    const float *c = colors + (size_t)g * D;
                    double *lcg = lc + (size_t)(s - start) * D;
                    float v_alpha = 0.f;
                    for (int k = 0; k < D; ++k) lcg[k] += (double)(fac * vc[k]);
                    if (!geom) continue;
                    for (int k = 0; k < D; ++k) v_alpha += (c[k] * T - buf[k] * ra) * vc[k];
                    v_alpha += T_final * ra * va;
                    if (backgrounds) v_alpha += -T_final * ra * bg_dot;
                    if (opac * vis <= ORC_ALPHA_MAX) {
                        const float v_sigma = -opac * vis * v_alpha;
                        double *l = lg + (size_t)(s - start) * 6;
                        l[0] += (double)(0.5f * v_sigma * dx * dx);
                        l[1] += (double)(v_sigma * dx * dy);
                        l[2] += (double)(0.5f * v_sigma * dy * dy);
                        l[3] += (double)(v_sigma * (ca * dx + cb * dy));
                        l[4] += (double)(v_sigma * (cb * dx + cc * dy));
                        l[5] += (double)(vis * v_alpha);
                    }
                    for (int k = 0; k < D; ++k) buf[k] += c[k] * fac;
                }
            }
        for (int64_t s = 0; s < cnt; ++s) {
            const int32_t g = flatten_ids[start + s];
            float *o = v_colors + (size_t)g * D;
            const double *l = lc + (size_t)s * D;
            for (int k = 0; k < D; ++k) {
                const float add = (float)l[k];
                if (add != 0.f) {
#pragma omp atomic
                    o[k] += add;
                }
            }
            if (geom) {
                const double *q = lg + (size_t)s * 6;
#pragma omp atomic
                v_conics[3 * g] += (float)q[0];
#pragma omp atomic
                v_conics[3 * g + 1] += (float)q[1];
#pragma omp atomic
                v_conics[3 * g + 2] += (float)q[2];
#pragma omp atomic
                v_means2d[2 * g] += (float)q[3];
#pragma omp atomic
                v_means2d[2 * g + 1] += (float)q[4];
#pragma omp atomic
                v_opacities[g] += (float)q[5];
            }
        }
        free(lc); free(buf);
        if (lg) free(lg);
    }
}

/* ------------------------------------------------------------------------------------
 * K3  spherical harmonics colour; basis/signs exactly as
 * /root/reference/utils/sh_utils.py:57-112.  coeffs [N,Kc,3] (Kc >= (deg+1)^2),
 * dirs = mean - campos (normalised here), out [N,3] = max(SH + 0.5, 0) where radii>0
 * (zeros elsewhere), matching the `+0.5, clamp_min(0)` gsplat applies to SH colours.
 * ---------------------------------------------------------------------------------- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

void orc_sh_fwd(int N, int Kc, int deg, const float *means, const float *campos,
                const float *coeffs, const int32_t *radii, float *out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float *o = out + 3 * (size_t)i;
        if (radii && radii[i] <= 0) { o[0] = o[1] = o[2] = 0.f; continue; }
        float x = means[3 * i] - campos[0], y = means[3 * i + 1] - campos[1], z = means[3 * i + 2] - campos[2];
        const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        const float *sh = coeffs + (size_t)i * Kc * 3;
        for (int c = 0; c < 3; ++c) {
#define SH(k) sh[(k) * 3 + c]
            float r = SH_C0 * SH(0);
            if (deg > 0) {
                r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
                if (deg > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                        SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                        SH_C2[4] * (xx - yy) * SH(8);
                    if (deg > 2) {
                        r = r + SH_C3[0] * y * (3.f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                            SH_C3[2] * y * (4.f * zz - xx - yy) * SH(11) +
                            SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SH(12) +
                            SH_C3[4] * x * (4.f * zz - xx - yy) * SH(13) +
                            SH_C3[5] * z * (xx - yy) * SH(14) + SH_C3[6] * x * (xx - 3.f * yy) * SH(15);
                    }
                }
            }
#undef SH
            o[c] = fmaxf(r + 0.5f, 0.f);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * K2  fused projection backward: chain rule of orc_project_fwd for
 * Gaussians with radii > 0; zeros elsewhere.  Never reached in the GAD flow (geometry is
 * frozen, scene/gaussian_model.py:200-205) but part of the rasterization operator.
 * v_conics follows the raster backward's convention: conic b is the single off-diagonal
 * parameter (sigma = .5(a dx^2 + c dy^2) + b dx dy).
 * ---------------------------------------------------------------------------------- */
void orc_project_bwd(int N, const float *means, const float *quats, const float *scales,
                     const float *viewmat, const float *Kmat, int width, int height, float eps2d,
                     const int32_t *radii, const float *v_means2d, const float *v_depths,
                     const float *v_conics, float *v_means, float *v_quats, float *v_scales)
{
    const float Rc[3][3] = {{viewmat[0], viewmat[1], viewmat[2]},
                            {viewmat[4], viewmat[5], viewmat[6]},
                            {viewmat[8], viewmat[9], viewmat[10]}};
    const float tc[3] = {viewmat[3], viewmat[7], viewmat[11]};
    const float fx = Kmat[0], cx = Kmat[2], fy = Kmat[4], cy = Kmat[5];
    const float fw = (float)width, fh = (float)height;
    const float tan_fovx = 0.5f * fw / fx, tan_fovy = 0.5f * fh / fy;
    const float lim_x_pos = (fw - cx) / fx + 0.3f * tan_fovx, lim_x_neg = cx / fx + 0.3f * tan_fovx;
    const float lim_y_pos = (fh - cy) / fy + 0.3f * tan_fovy, lim_y_neg = cy / fy + 0.3f * tan_fovy;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = 0.f; v_scales[3 * i + k] = 0.f; }
        for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = 0.f;
        if (radii[i] <= 0) continue;
        /* ---- recompute forward intermediates ---- */
        const float mu[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        float p[3];
        for (int r = 0; r < 3; ++r) p[r] = ((Rc[r][0] * mu[0] + Rc[r][1] * mu[1]) + Rc[r][2] * mu[2]) + tc[r];
        const float x = p[0], y = p[1], z = p[2];
        const float q0 = quats[4 * i], q1 = quats[4 * i + 1], q2 = quats[4 * i + 2], q3 = quats[4 * i + 3];
        const float inv_norm = 1.0f / sqrtf(((q1 * q1 + q2 * q2) + q3 * q3) + q0 * q0);
        const float qw = q0 * inv_norm, qx = q1 * inv_norm, qy = q2 * inv_norm, qz = q3 * inv_norm;
        float R[3][3];
        R[0][0] = 1.f - 2.f * (qy * qy + qz * qz); R[0][1] = 2.f * (qx * qy - qw * qz); R[0][2] = 2.f * (qx * qz + qw * qy);
        R[1][0] = 2.f * (qx * qy + qw * qz); R[1][1] = 1.f - 2.f * (qx * qx + qz * qz); R[1][2] = 2.f * (qy * qz - qw * qx);
        R[2][0] = 2.f * (qx * qz - qw * qy); R[2][1] = 2.f * (qy * qz + qw * qx); R[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        float M[3][3], S3[3][3], A[3][3], Sc[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] = R[r][c] * s[c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            S3[r][c] = (M[r][0] * M[c][0] + M[r][1] * M[c][1]) + M[r][2] * M[c][2];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            A[r][c] = (Rc[r][0] * S3[0][c] + Rc[r][1] * S3[1][c]) + Rc[r][2] * S3[2][c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            Sc[r][c] = (A[r][0] * Rc[c][0] + A[r][1] * Rc[c][1]) + A[r][2] * Rc[c][2];
        const float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = x * rz, yr = y * rz;
        const int x_in = (xr <= lim_x_pos) && (xr >= -lim_x_neg);
        const int y_in = (yr <= lim_y_pos) && (yr >= -lim_y_neg);
        const float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
        const float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
        const float J[2][3] = {{fx * rz, 0.f, -fx * tx * rz2}, {0.f, fy * rz, -fy * ty * rz2}};
        float B[2][3];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
            B[r][c] = (J[r][0] * Sc[0][c] + J[r][1] * Sc[1][c]) + J[r][2] * Sc[2][c];
        const float s00 = ((B[0][0] * J[0][0] + B[0][1] * J[0][1]) + B[0][2] * J[0][2]) + eps2d;
        const float s01 = (B[0][0] * J[1][0] + B[0][1] * J[1][1]) + B[0][2] * J[1][2];
        const float s11 = ((B[1][0] * J[1][0] + B[1][1] * J[1][1]) + B[1][2] * J[1][2]) + eps2d;
        const float det = s00 * s11 - s01 * s01;
        const float inv_det = 1.f / det;
        const float ca = s11 * inv_det, cb = -s01 * inv_det, cc = s00 * inv_det;
        /* ---- 1. conic -> cov2d:  G = -X Gx X,  X = [[ca,cb],[cb,cc]], Gx = [[va, vb/2],[vb/2, vc]] ---- */
        const float va = v_conics[3 * i], vb = 0.5f * v_conics[3 * i + 1], vc = v_conics[3 * i + 2];
        const float t00 = ca * va + cb * vb, t01 = ca * vb + cb * vc;
        const float t10 = cb * va + cc * vb, t11 = cb * vb + cc * vc;
        const float G00 = -(t00 * ca + t01 * cb), G01 = -(t00 * cb + t01 * cc);
        const float G10 = -(t10 * ca + t11 * cb), G11 = -(t10 * cb + t11 * cc);
        const float G[2][2] = {{G00, G01}, {G10, G11}};
        /* ---- 2. cov2d = J Sc J^T:  G_Sc = J^T G J ; G_J = G J Sc^T + G^T J Sc ---- */
        float GJ[2][3], GSc[3][3], vJ[2][3];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) GJ[r][c] = G[r][0] * J[0][c] + G[r][1] * J[1][c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) GSc[r][c] = J[0][r] * GJ[0][c] + J[1][r] * GJ[1][c];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            const float gt0 = G[0][r], gt1 = G[1][r]; /* G^T row r */
            const float a1 = (GJ[r][0] * Sc[c][0] + GJ[r][1] * Sc[c][1]) + GJ[r][2] * Sc[c][2];
            const float gtj0 = gt0 * J[0][0] + gt1 * J[1][0], gtj1 = gt0 * J[0][1] + gt1 * J[1][1],
                        gtj2 = gt0 * J[0][2] + gt1 * J[1][2];
            const float a2 = (gtj0 * Sc[0][c] + gtj1 * Sc[1][c]) + gtj2 * Sc[2][c];
            vJ[r][c] = a1 + a2;
        }
        /* ---- 3. J and mean2d/depth -> camera-space point ---- */
        const float vm2x = v_means2d[2 * i], vm2y = v_means2d[2 * i + 1];
        float vp[3];
        vp[0] = fx * rz * vm2x;
        vp[1] = fy * rz * vm2y;
        vp[2] = -(fx * x * vm2x + fy * y * vm2y) * rz2;
        if (v_depths) vp[2] += v_depths[i];
        if (x_in) vp[0] += -fx * rz2 * vJ[0][2]; else vp[2] += -fx * rz3 * vJ[0][2] * tx;
        if (y_in) vp[1] += -fy * rz2 * vJ[1][2]; else vp[2] += -fy * rz3 * vJ[1][2] * ty;
        vp[2] += -fx * rz2 * vJ[0][0] - fy * rz2 * vJ[1][1] + 2.f * fx * tx * rz3 * vJ[0][2] + 2.f * fy * ty * rz3 * vJ[1][2];
        /* ---- 4. camera -> world ---- */
        for (int c = 0; c < 3; ++c) v_means[3 * i + c] = (Rc[0][c] * vp[0] + Rc[1][c] * vp[1]) + Rc[2][c] * vp[2];
        float T1[3][3], GS[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            T1[r][c] = (Rc[0][r] * GSc[0][c] + Rc[1][r] * GSc[1][c]) + Rc[2][r] * GSc[2][c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            GS[r][c] = (T1[r][0] * Rc[0][c] + T1[r][1] * Rc[1][c]) + T1[r][2] * Rc[2][c];
        /* ---- 5. Sigma = M M^T: G_M = (G_S + G_S^T) M ---- */
        float GM[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            GM[r][c] = ((GS[r][0] + GS[0][r]) * M[0][c] + (GS[r][1] + GS[1][r]) * M[1][c]) + (GS[r][2] + GS[2][r]) * M[2][c];
        /* ---- 6. M = R diag(s) ---- */
        float GR[3][3];
        for (int c = 0; c < 3; ++c) {
            v_scales[3 * i + c] = (R[0][c] * GM[0][c] + R[1][c] * GM[1][c]) + R[2][c] * GM[2][c];
            for (int r = 0; r < 3; ++r) GR[r][c] = GM[r][c] * s[c];
        }
        /* ---- 7. R(q^) -> q^ ---- */
        const float vqw = 2.f * (-qz * GR[0][1] + qy * GR[0][2] + qz * GR[1][0] - qx * GR[1][2] - qy * GR[2][0] + qx * GR[2][1]);
        const float vqx = 2.f * (qy * GR[0][1] + qz * GR[0][2] + qy * GR[1][0] - 2.f * qx * GR[1][1] - qw * GR[1][2] +
                                 qz * GR[2][0] + qw * GR[2][1] - 2.f * qx * GR[2][2]);
        const float vqy = 2.f * (-2.f * qy * GR[0][0] + qx * GR[0][1] + qw * GR[0][2] + qx * GR[1][0] + qz * GR[1][2] -
                                 qw * GR[2][0] + qz * GR[2][1] - 2.f * qy * GR[2][2]);
        const float vqz = 2.f * (-2.f * qz * GR[0][0] - qw * GR[0][1] + qx * GR[0][2] + qw * GR[1][0] - 2.f * qz * GR[1][1] +
                                 qy * GR[1][2] + qx * GR[2][0] + qy * GR[2][1]);
        /* ---- 8. normalisation q^ = q/|q| ---- */
        const float dotp = ((qw * vqw + qx * vqx) + qy * vqy) + qz * vqz;
        v_quats[4 * i] = (vqw - qw * dotp) * inv_norm;
        v_quats[4 * i + 1] = (vqx - qx * dotp) * inv_norm;
        v_quats[4 * i + 2] = (vqy - qy * dotp) * inv_norm;
        v_quats[4 * i + 3] = (vqz - qz * dotp) * inv_norm;
    }
}

/* ------------------------------------------------------------------------------------
 * Colours-only backward in FORWARD order: v_colors[g] += sum_px (alpha*T) * v_out[px], with
 * alpha*T recomputed front to back exactly as orc_raster_fwd computes it.
 * Mathematically identical to orc_raster_bwd's v_colors; numerically it avoids gsplat's
 * reconstruction T = (1 - render_alpha) * prod 1/(1-alpha), whose fp32 cancellation in
 * 1 - (1 - T) costs up to ~1e-3 relative on nearly saturated pixels (gsplat's own source
 * comments on this).  The HIP colours-only kernel uses this order; tests compare it against
 * this function tightly and against the gsplat-order function loosely.
 * ---------------------------------------------------------------------------------- */
void orc_raster_bwd_colors_fwdorder(int D, int width, int height, int tile_w, int tile_h,
                                    const float *means2d, const float *conics, const float *opacities,
                                    const int32_t *tile_offsets, const int32_t *flatten_ids, int64_t n_isects,
                                    const float *v_render_colors, int tile_begin, int tile_step, float *v_colors)
{
    const int n_tiles = tile_w * tile_h;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = tile_begin; tile < n_tiles; tile += tile_step) {
        const int ty = tile / tile_w, tx = tile % tile_w;
        const int64_t start = tile_offsets[tile];
        const int64_t end = (tile == n_tiles - 1) ? n_isects : tile_offsets[tile + 1];
        const int64_t cnt = end - start;
        if (cnt <= 0) continue;
        double *lc = (double *)calloc((size_t)cnt * D, sizeof(double));
        for (int ly = 0; ly < ORC_TILE; ++ly)
            for (int lx = 0; lx < ORC_TILE; ++lx) {
                const int i = ty * ORC_TILE + ly, j = tx * ORC_TILE + lx;
                if (i >= height || j >= width) continue;
                const float px = (float)j + 0.5f, py = (float)i + 0.5f;
                const float *vc = v_render_colors + ((size_t)i * width + j) * D;
                float T = 1.0f;
                for (int64_t s = start; s < end; ++s) {
                    const int32_t g = flatten_ids[s];
                    const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    const float alpha = fminf(ORC_ALPHA_MAX, opacities[g] * orc_exp_neg(sigma));
                    if (sigma < 0.f || alpha < ORC_ALPHA_MIN) continue;
                    const float next_T = T * (1.0f - alpha);
                    if (next_T <= ORC_T_STOP) break;
                    const float vis = alpha * T;
                    double *l = lc + (size_t)(s - start) * D;
                    for (int k = 0; k < D; ++k) l[k] += (double)vis * (double)vc[k];
                    T = next_T;
                }
            }
        for (int64_t s = 0; s < cnt; ++s) {
            const int32_t g = flatten_ids[start + s];
            float *o = v_colors + (size_t)g * D;
            const double *l = lc + (size_t)s * D;
            for (int k = 0; k < D; ++k) {
                const float add = (float)l[k];
                if (add != 0.f) {
#pragma omp atomic
                    o[k] += add;
                }
            }
        }
        free(lc);
    }
}

/* ------------------------------------------------------------------------------------
 * R9  Adam step of the feature parameter, as the reference configures it:
 * /root/reference/scene/gaussian_model.py:192-208 (`torch.optim.Adam(l, lr=0.0, eps=1e-15)`, one
 * group: `_semantic_feature`, lr = semantic_feature_lr), stepped at /root/reference/train.py:221-223.
 * torch.optim.Adam, single-tensor path, no weight decay, no amsgrad:
 *     m <- m + (1-b1)(g - m);  v <- v*b2 + (1-b2) g g
 *     p <- p - (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
 * The scalars are formed in double (python floats in torch) and applied in fp32.
 * ---------------------------------------------------------------------------------- */
void orc_adam_step(int64_t n, float *p, const float *g, float *m, float *v, double lr, double beta1,
                   double beta2, double eps, int step)
{
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), epsf = (float)eps;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float gi = g[i];
        const float mi = m[i] + w1 * (gi - m[i]);
        const float vi = v[i] * b2 + (w2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + epsf;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}
