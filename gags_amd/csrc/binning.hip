// K5 / K6 / K8: prefix sum, intersection emit, per-tile offsets.  All HBM-bound integer work
// (SURVEY.md 8a R5); coalesced, wave64 scans, no GEMM reshaping.
#include "common.h"
#include "scan.h"

namespace {

// One wave emits the intersections of 64 consecutive Gaussians of the emission order.  Their output positions are ONE
// contiguous range [cum[first - 1], cum[last]) -- so the wave writes that range with consecutive lanes on consecutive
// entries (8-byte keys, 4-byte values: whole cache lines) and finds, per entry, the Gaussian it belongs to by a 6-step
// search over the wave's 64 start offsets in LDS.  (One lane per Gaussian writing its own run of ~5 entries put every
// 12 bytes on a cache line of its own: 0.8 TB/s, 107 us at C3.)  Same entries, same order: tile rows top to bottom,
// left to right inside the Gaussian's tile rectangle.
__global__ __launch_bounds__(256) void tile_emit_kernel(int n, const float *__restrict__ means2d,
                                                        const int32_t *__restrict__ radii,
                                                        const float *__restrict__ depths,
                                                        const int32_t *__restrict__ cum,
                                                        const int32_t *__restrict__ order, int tile_w, int tile_h,
                                                        int64_t *__restrict__ isect_ids,
                                                        int32_t *__restrict__ flatten_ids, int64_t cap)
{
    __shared__ int s_rel[4][64], s_x0[4][64], s_y0[4][64], s_w[4][64], s_gid[4][64];
    __shared__ uint32_t s_dbits[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 256 + threadIdx.x;  // position in the emission order
    const int jw = j - lane;                       // the wave's first position
    if (jw >= n) return;                           // (whole wave)
    const bool live = j < n;
    const int i = live ? (order ? order[j] : j) : 0;  // Gaussian
    const int rad = live ? radii[i] : 0;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    uint32_t dbits = 0;
    if (rad > 0) {
        const float2 m = reinterpret_cast<const float2 *>(means2d)[i];
        gags_tile_aabb(m.x, m.y, rad, tile_w, tile_h, x0, x1, y0, y1);
        dbits = (uint32_t)__float_as_int(depths[i]);
    }
    const int base = (jw == 0) ? 0 : cum[jw - 1];
    const int start = live ? ((j == 0) ? 0 : cum[j - 1]) : 0x7fffffff;
    const int jl = min(jw + 63, n - 1);
    const int total = cum[jl] - base;
    s_rel[w][lane] = live ? start - base : 0x7fffffff;
    s_x0[w][lane] = x0; s_y0[w][lane] = y0; s_w[w][lane] = max(x1 - x0, 1); s_gid[w][lane] = i; s_dbits[w][lane] = dbits;
    __builtin_amdgcn_wave_barrier();  // (LDS traffic of one wave is in order; the barrier keeps the compiler from reordering)
    for (int t = lane; t < total; t += 64) {
        // largest l with s_rel[l] <= t (start offsets are non-decreasing; Gaussians without tiles share their successor's)
        int l = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (s_rel[w][l + step] <= t) l += step;
        const int r = t - s_rel[w][l];
        const int ww = s_w[w][l];
        const int dy = r / ww;
        const int64_t tile_id = (int64_t)(s_y0[w][l] + dy) * tile_w + (s_x0[w][l] + (r - dy * ww));
        const int64_t pos = (int64_t)base + t;
        if (pos < cap) {  // (capacity-sized buffers: a count above the capacity is detected by the caller afterwards)
            isect_ids[pos] = (tile_id << 32) | (int64_t)s_dbits[w][l];
            flatten_ids[pos] = s_gid[w][l];
        }
    }
}

// capacity-sized buffers: entries [total, cap) become sentinels of a tile past the last one (they sort to the end and
// give isect_offsets its final entry = the true count) pointing at Gaussian 0 (never read: no tile's range holds them)
__global__ __launch_bounds__(256) void isect_tail_kernel(int64_t cap, const int32_t *__restrict__ total, int n_tiles,
                                                         int64_t *__restrict__ isect_ids, int32_t *__restrict__ flatten_ids)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap || i < (int64_t)total[0]) return;
    isect_ids[i] = (int64_t)n_tiles << 32;
    flatten_ids[i] = 0;
}

__global__ __launch_bounds__(256) void tile_offsets_kernel(int64_t n_isects, const int64_t *__restrict__ sorted_ids,
                                                           int n_tiles, int32_t *__restrict__ offsets)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_isects) return;
    const int cur = (int)((uint64_t)sorted_ids[idx] >> 32);
    if (idx == 0) {
        for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
        const int prev = (int)((uint64_t)sorted_ids[idx - 1] >> 32);
        for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)idx;
    }
    if (idx == n_isects - 1)
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n_isects;
}

__global__ void fill_i32_kernel(int n, int32_t v, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ __launch_bounds__(256) void ed_normalize_kernel(int64_t n_pix, int d, float *__restrict__ colors,
                                                           const float *__restrict__ alphas)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix) return;
    colors[i * d + (d - 1)] = colors[i * d + (d - 1)] / fmaxf(alphas[i], 1e-10f);
}

}  // namespace

extern "C" int gags_abi_version(void) { return 2; }  // 2: isect_offsets has n_tiles + 1 entries (the last one = n_isects)

extern "C" const char *gags_strerror(int code)
{
    switch (code) {
        case GAGS_OK: return "ok";
        case GAGS_EINVAL: return "invalid argument";
        case GAGS_ELAUNCH: return "HIP launch/runtime error";
        case GAGS_ESCRATCH: return "scratch buffer too small";
        case GAGS_ENODEV: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

extern "C" int gags_device_count(void)
{
    GAGS_CLEAR_ERR();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return GAGS_ENODEV;
    return n;
}

extern "C" int64_t gags_scan_scratch_bytes(int n)
{
    return gags_scan::scratch_bytes(n);
}

extern "C" int gags_cumsum_i32(int n, const int32_t *in, int32_t *cum, int32_t *total, void *scratch,
                               int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (total) hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, 1, 0, total);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (!in || !cum || !scratch) return GAGS_EINVAL;
    if (scratch_bytes < gags_scan_scratch_bytes(n)) return GAGS_ESCRATCH;
    gags_scan::launch<false>(n, in, cum, total, (int32_t *)scratch, st);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_cumsum_gather_i32(int n, const int32_t *in, const int32_t *idx, int32_t *cum, int32_t *total, void *scratch,
                                      int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (total) hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, 1, 0, total);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (!in || !idx || !cum || !scratch || in == cum) return GAGS_EINVAL;  // (a permuted read cannot run in place)
    if (scratch_bytes < gags_scan_scratch_bytes(n)) return GAGS_ESCRATCH;
    gags_scan::launch<false>(n, in, cum, total, (int32_t *)scratch, st, idx);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_read_i32(const int32_t *src, int32_t *dst_host, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!src || !dst_host) return GAGS_EINVAL;
    if (hipMemcpyAsync(dst_host, src, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return GAGS_ELAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GAGS_ELAUNCH;
    return GAGS_OK;
}

extern "C" int gags_tile_emit_cap(int n, const float *means2d, const int32_t *radii, const float *depths, const int32_t *cum,
                                  const int32_t *order, int tile_w, int tile_h, int64_t *isect_ids, int32_t *flatten_ids,
                                  int64_t cap, const int32_t *total, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || tile_w <= 0 || tile_h <= 0 || cap < 0 || cap >= (1ll << 31)) return GAGS_EINVAL;
    if (n == 0 && cap == 0) return GAGS_OK;
    if (!isect_ids || !flatten_ids || (n > 0 && (!means2d || !radii || !depths || !cum))) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n > 0)
        hipLaunchKernelGGL(tile_emit_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, means2d, radii, depths, cum, order, tile_w,
                           tile_h, isect_ids, flatten_ids, cap);
    if (total && cap > 0)
        hipLaunchKernelGGL(isect_tail_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, cap, total, tile_w * tile_h,
                           isect_ids, flatten_ids);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_tile_emit(int n, const float *means2d, const int32_t *radii, const float *depths,
                              const int32_t *cum, const int32_t *order, int tile_w, int tile_h, int64_t *isect_ids,
                              int32_t *flatten_ids, void *stream)
{
    if (n == 0) return GAGS_OK;
    return gags_tile_emit_cap(n, means2d, radii, depths, cum, order, tile_w, tile_h, isect_ids, flatten_ids, (1ll << 31) - 1, nullptr,
                              stream);
}

extern "C" int gags_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles, int32_t *isect_offsets,
                                 void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_isects < 0 || n_tiles <= 0 || !isect_offsets) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // n_tiles + 1 entries: [n_tiles] = the intersection count (with capacity-sized inputs: where the sentinel keys of
    // tile `n_tiles` begin), so that no raster kernel needs the count from the host
    if (n_isects == 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((n_tiles + 1 + 255) / 256), dim3(256), 0, st, n_tiles + 1, 0, isect_offsets);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (!sorted_ids) return GAGS_EINVAL;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((unsigned)((n_isects + 255) / 256)), dim3(256), 0, st, n_isects,
                       sorted_ids, n_tiles + 1, isect_offsets);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_ed_normalize(int64_t n_pix, int d, float *render_colors, const float *render_alphas,
                                 void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || d <= 0) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    if (!render_colors || !render_alphas) return GAGS_EINVAL;
    hipLaunchKernelGGL(ed_normalize_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_pix, d, render_colors, render_alphas);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
