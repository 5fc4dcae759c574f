// N4, second half: what evaluate_iou_loc.py:100-146 does to the relevancy maps of one view after
// OpenCLIPNetwork.get_max_across -- per phrase a 30x30 box mean (the reference: cv2.filter2D on the HOST, one
// device -> host -> device round trip per phrase), blend with the map, min-max normalisation to [-1, 1], clip to
// [0, 1], threshold, and the 7x7 majority filter eval/utils.py:55-64 runs as a Python double loop over every pixel --
// and the box mean + arg-max lerf_localization (:163-176) needs.  All phrases in one call, everything stays on the GPU.
// Plain HBM / L2-bound image kernels: one thread per pixel, coalesced along x.
#include "common.h"
#include "gags_next.h"

namespace {

// cv2.BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), the default border of cv2.filter2D
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// float <-> unsigned key with the same order (min / max of the maps by integer atomics: exact and order-independent)
__device__ __forceinline__ unsigned f2key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// horizontal pass: rowsum[k, y, x] = sum_{dx = -a .. box-1-a} src[k, y, reflect(x + dx)],  a = box / 2 (cv2's anchor).
// Sums in double, rounded once at the end of the vertical pass: the reflected border makes neighbouring windows hold the
// same multiset of pixels, and lerf_localization takes EVERY position that attains the maximum (:174-176) -- fp32 partial
// sums in window order would break such exact ties.
__global__ __launch_bounds__(256) void box_rows_kernel(int h, int w, int box, const float *__restrict__ src,
                                                       double *__restrict__ rowsum)
{
    extern __shared__ float seg[];  // 256 + box values of the row
    const int y = blockIdx.y, k = blockIdx.z, x0 = blockIdx.x * 256, a = box / 2;
    const float *row = src + ((size_t)k * h + y) * w;
    for (int i = threadIdx.x; i < 256 + box; i += 256) seg[i] = row[reflect101(x0 + i - a, w)];
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= w) return;
    double s = 0.0;
    for (int i = 0; i < box; ++i) s += (double)seg[threadIdx.x + i];
    rowsum[((size_t)k * h + y) * w + x] = s;
}

// vertical pass + blend: avg = (sum over the column window) / box^2, blended = 0.5 (avg + src); min / max of the blended
// map and max of avg per phrase (keys[k] = {min blended, max blended, max avg})
__global__ __launch_bounds__(256) void box_cols_kernel(int h, int w, int box, const float *__restrict__ src,
                                                       const double *__restrict__ rowsum, float *__restrict__ avg,
                                                       float *__restrict__ blended, unsigned *__restrict__ keys)
{
    __shared__ unsigned red[3][4];
    const int k = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x, a = box / 2;
    const bool in = x < w;
    float av = 0.f, bl = 0.f;
    if (in) {
        const double *col = rowsum + (size_t)k * h * w + x;
        double s = 0.0;
        for (int i = 0; i < box; ++i) s += col[(size_t)reflect101(y + i - a, h) * w];
        av = (float)(s / (double)(box * box));
        const size_t o = ((size_t)k * h + y) * w + x;
        bl = 0.5f * (av + src[o]);
        avg[o] = av;
        blended[o] = bl;
    }
    unsigned kmin = in ? f2key(bl) : 0xffffffffu, kmax = in ? f2key(bl) : 0u, kavg = in ? f2key(av) : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, off, 64));
        kavg = max(kavg, (unsigned)__shfl_xor((int)kavg, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = kmin; red[1][threadIdx.x >> 6] = kmax; red[2][threadIdx.x >> 6] = kavg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(keys + 3 * k + 0, min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3])));
        atomicMax(keys + 3 * k + 1, max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3])));
        atomicMax(keys + 3 * k + 2, max(max(red[2][0], red[2][1]), max(red[2][2], red[2][3])));
    }
}

__global__ void keys_init_kernel(int n_phrases, unsigned *__restrict__ keys)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < 3 * n_phrases) keys[i] = (i % 3 == 0) ? 0xffffffffu : 0u;
}

// evaluate_iou_loc.py:131-137: output = clip(((v - min) / (max - min + 1e-9)) * 2 - 1, 0, 1); mask = output > thresh
__global__ __launch_bounds__(256) void normalise_kernel(int64_t hw, float thresh, const float *__restrict__ blended,
                                                        const unsigned *__restrict__ keys, float *__restrict__ output,
                                                        unsigned char *__restrict__ mask, float *__restrict__ stats)
{
    const int k = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float mn = key2f(keys[3 * k]), mx = key2f(keys[3 * k + 1]);
    if (p == 0) { stats[3 * k] = mn; stats[3 * k + 1] = mx; stats[3 * k + 2] = key2f(keys[3 * k + 2]); }
    if (p >= hw) return;
    float o = blended[(size_t)k * hw + p] - mn;
    o = o / ((mx - mn) + 1e-9f);
    o = o * 2.0f + -1.0f;
    o = fminf(fmaxf(o, 0.f), 1.f);
    output[(size_t)k * hw + p] = o;
    mask[(size_t)k * hw + p] = o > thresh ? 1 : 0;
}

// eval/utils.py:55-64 `smooth`: majority vote over mask[max(0, i-s) : min(i+s+1, h-1), max(0, j-s) : min(j+s+1, w-1)]
// (the reference's own bounds: the last row / column never votes); argmax(bincount) = 1 iff ones > zeros.
__global__ __launch_bounds__(256) void majority_kernel(int h, int w, int s, const unsigned char *__restrict__ mask,
                                                       unsigned char *__restrict__ out)
{
    const int k = blockIdx.z, i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w) return;
    const int i0 = max(0, i - s), i1 = min(i + s + 1, h - 1), j0 = max(0, j - s), j1 = min(j + s + 1, w - 1);
    int ones = 0, total = 0;
    for (int y = i0; y < i1; ++y)
        for (int x = j0; x < j1; ++x) {
            ones += mask[((size_t)k * h + y) * w + x];
            ++total;
        }
    // an empty window: np.bincount([]) is empty and np.argmax raises in the reference; here the pixel keeps its value
    out[((size_t)k * h + i) * w + j] = total == 0 ? mask[((size_t)k * h + i) * w + j] : (2 * ones > total ? 1 : 0);
}

inline int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t gags_relevancy_activate_scratch_bytes(int n_phrases, int h, int w)
{
    if (n_phrases <= 0 || h <= 0 || w <= 0) return 0;
    return al256((int64_t)n_phrases * h * w * 8) + al256((int64_t)n_phrases * 3 * 4);
}

extern "C" int gags_relevancy_activate(int n_phrases, int h, int w, const float *valid_map, float thresh, int box,
                                       int smooth_scale, float *avg, float *blended, float *output,
                                       unsigned char *mask_pred, unsigned char *mask_smooth, float *stats, void *scratch,
                                       int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_phrases <= 0 || h <= 0 || w <= 0 || box <= 0 || box > 1024 || smooth_scale < 0 || n_phrases > 65535 || h > 65535)
        return GAGS_EINVAL;
    if (!valid_map || !avg || !blended || !output || !mask_pred || !mask_smooth || !stats || !scratch) return GAGS_EINVAL;
    if (scratch_bytes < gags_relevancy_activate_scratch_bytes(n_phrases, h, w)) return GAGS_ESCRATCH;
    hipStream_t st = (hipStream_t)stream;
    double *rowsum = (double *)scratch;
    unsigned *keys = (unsigned *)((char *)scratch + al256((int64_t)n_phrases * h * w * 8));
    const dim3 grid((w + 255) / 256, h, n_phrases);
    hipLaunchKernelGGL(keys_init_kernel, dim3((3 * n_phrases + 63) / 64), dim3(64), 0, st, n_phrases, keys);
    hipLaunchKernelGGL(box_rows_kernel, grid, dim3(256), (size_t)(256 + box) * 4, st, h, w, box, valid_map, rowsum);
    hipLaunchKernelGGL(box_cols_kernel, grid, dim3(256), 0, st, h, w, box, valid_map, rowsum, avg, blended, keys);
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(normalise_kernel, dim3((unsigned)((hw + 255) / 256), n_phrases), dim3(256), 0, st, hw, thresh, blended,
                       keys, output, mask_pred, stats);
    hipLaunchKernelGGL(majority_kernel, grid, dim3(256), 0, st, h, w, smooth_scale, mask_pred, mask_smooth);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
