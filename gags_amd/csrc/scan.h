// Device-wide prefix sum of int32 (three launches: block sums, spine, final).  Shared by the
// tile-count cumsum (K5) and the radix-sort histogram scan (K7).  wave64 shuffles + LDS.
#pragma once
#include "common.h"

namespace gags_scan {
namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;                       // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 per block

__device__ __forceinline__ int wave_incl_scan(int v)
{
    // wave64 inclusive scan by shuffles
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// block-wide inclusive scan of one int per thread (256 threads = 4 waves); returns the
// inclusive prefix and the block total.
__device__ __forceinline__ int block_incl_scan(int v, int &total, int *smem /*>=4 ints*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = wave_incl_scan(v);
    if (lane == 63) smem[w] = s;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = smem[k];
        if (k < w) off += t;
    }
    total = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return s + off;
}

// idx (optional): the scan runs over in[idx[i]] -- a permuted view of `in` (the tile counts in depth order) without
// a gather kernel and a permuted copy in between
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums(int n, const int32_t *__restrict__ in,
                                                                 const int32_t *__restrict__ idx,
                                                                 int32_t *__restrict__ block_sums)
{
    __shared__ int smem[4];
    const int base = blockIdx.x * SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int i = base + k * SCAN_THREADS + threadIdx.x;
        s += (i < n) ? in[idx ? idx[i] : i] : 0;
    }
    int total;
    block_incl_scan(s, total, smem);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of block_sums in place, grand total to total[0]
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine(int nb, int32_t *__restrict__ block_sums,
                                                           int32_t *__restrict__ total_out)
{
    __shared__ int smem[4];
    __shared__ double wide[SCAN_THREADS];
    int carry = 0;
    double dsum = 0.0;  // the same total in a type that cannot wrap: an int32 overflow is reported, not returned
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        const int i = base + threadIdx.x;
        const int v = (i < nb) ? block_sums[i] : 0;
        dsum += (double)v;
        int total;
        const int incl = block_incl_scan(v, total, smem);
        if (i < nb) block_sums[i] = carry + incl - v;
        carry += total;
    }
    wide[threadIdx.x] = dsum;
    __syncthreads();
    if (threadIdx.x == 0 && total_out) {
        double t = 0.0;
        for (int k = 0; k < SCAN_THREADS; ++k) t += wide[k];
        total_out[0] = (t > 2147483647.0) ? -1 : carry;  // -1: the caller must refuse (rasterization.py)
    }
}

template <bool EXCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final(int n, const int32_t *in /* may alias out */,
                                                            const int32_t *__restrict__ idx,
                                                            const int32_t *__restrict__ block_offs,
                                                            int32_t *out)
{
    __shared__ int smem[4];
    // thread t owns SCAN_ITEMS consecutive items (blocked arrangement keeps the scan simple)
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int i = base + k;
        v[k] = (i < n) ? in[idx ? idx[i] : i] : 0;
        s += v[k];
    }
    int total;
    const int incl = block_incl_scan(s, total, smem);
    int run = block_offs[blockIdx.x] + incl - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int i = base + k;
        if (EXCLUSIVE) {
            if (i < n) out[i] = run;
            run += v[k];
        } else {
            run += v[k];
            if (i < n) out[i] = run;
        }
    }
}


}  // namespace

inline int64_t scratch_bytes(int64_t n)
{
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (nb > 0 ? nb : 1) * (int64_t)sizeof(int32_t);
}

// in-place allowed (in == out).  total may be NULL.
template <bool EXCLUSIVE>
inline void launch(int n, const int32_t *in, int32_t *out, int32_t *total, int32_t *block_scratch, hipStream_t st,
                   const int32_t *idx = nullptr)
{
    const int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, n, in, idx, block_scratch);
    hipLaunchKernelGGL(scan_spine, dim3(1), dim3(SCAN_THREADS), 0, st, nb, block_scratch, total);
    hipLaunchKernelGGL(scan_final<EXCLUSIVE>, dim3(nb), dim3(SCAN_THREADS), 0, st, n, in, idx, block_scratch, out);
}

}  // namespace gags_scan
