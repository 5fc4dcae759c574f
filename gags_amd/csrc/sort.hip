// K7: stable LSD radix sort of (int64 key, int32 value) pairs on key bits [0, 32+tile_bits).
//
// Hand-written for gfx950 wave64 (no library): 8-bit digits, three kernels per pass
//   1. histogram   : per-block digit counts (LDS atomics), written digit-major
//   2. row scan    : one workgroup per digit: exclusive prefix of its row over the blocks + the row total
//   3. scatter     : each block re-reads its tile, ranks keys STABLY (wave-level match via
//                    per-digit ballots, wave-ordered LDS counters), stages the pairs in LDS in sorted order and writes
//                    whole digit runs out (digit bases: an exclusive scan of the 256 row totals in the prologue)
// Traffic per pass: read 12 B + write 12 B per pair + one extra 8 B key read for the
// histogram = 32 B/pair; 45-bit keys at 1080p (13 tile bits) need 6 passes.
// HBM-bound integer work: nothing here is reshaped into a GEMM.
#include "common.h"
#include "scan.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 2048 pairs per block
constexpr int RS_RADIX = 256;

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_histogram(int64_t n, int shift, const K *__restrict__ keys,
                                                           uint32_t *__restrict__ hist /*[RADIX][nblocks]*/,
                                                           int nblocks)
{
    __shared__ uint32_t h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int64_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xff], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Scan of the [digit][block] histogram, one workgroup per DIGIT: row d becomes its own exclusive prefix over the blocks
// and totals[d] the row's sum; the scatter kernel adds the digits' bases (an exclusive scan of the 256 totals, done by
// each workgroup in its prologue).  One fully parallel launch instead of the generic three-kernel scan with its
// single-workgroup spine over the flattened matrix.
__global__ __launch_bounds__(RS_THREADS) void rs_rowscan(int nblocks, uint32_t *__restrict__ hist, uint32_t *__restrict__ totals)
{
    __shared__ int smem[4];
    uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += RS_TILE) {
        const int i0 = base + threadIdx.x * RS_ITEMS;
        uint32_t v[RS_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < RS_ITEMS; ++k) {
            v[k] = (i0 + k < nblocks) ? row[i0 + k] : 0u;
            s += v[k];
        }
        int total;
        const int incl = gags_scan::block_incl_scan((int)s, total, smem);
        uint32_t run = carry + (uint32_t)incl - s;
#pragma unroll
        for (int k = 0; k < RS_ITEMS; ++k) {
            if (i0 + k < nblocks) row[i0 + k] = run;
            run += v[k];
        }
        carry += (uint32_t)total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Stable scatter.  Items are laid out blocked-by-wave so that (wave, item, lane) order is the
// input order: wave w owns [w*512, (w+1)*512) of the tile, item k covers 64 consecutive keys.
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter(int64_t n, int shift, const K *__restrict__ keys_in,
                                                         const int32_t *__restrict__ vals_in,
                                                         K *__restrict__ keys_out,
                                                         int32_t *__restrict__ vals_out,
                                                         const uint32_t *__restrict__ hist, int nblocks,
                                                         const uint32_t *__restrict__ totals)
{
    __shared__ uint32_t cnt_s[4][RS_RADIX];  // per-wave digit counts, then per-wave bases
    volatile uint32_t(*cnt)[RS_RADIX] = cnt_s;
    __shared__ uint32_t gbase[RS_RADIX];    // global base of each digit for this block
    __shared__ int scan_tmp[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt[k][threadIdx.x] = 0;
    {
        // base of digit d = keys with a smaller digit (exclusive scan of the 256 row totals) + this digit's keys in
        // earlier blocks (rs_rowscan)
        const uint32_t tot = totals[threadIdx.x];
        int all;
        const int incl = gags_scan::block_incl_scan((int)tot, all, scan_tmp);
        gbase[threadIdx.x] = ((uint32_t)incl - tot) + hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
    }
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * (64 * RS_ITEMS);
    K key[RS_ITEMS];
    int32_t val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];  // rank of the key among same-digit keys of THIS wave (input order)
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int64_t i = wbase + k * 64 + lane;
        const bool ok = i < n;
        key[k] = ok ? keys_in[i] : (K)~(K)0;
        val[k] = ok ? (vals_in ? vals_in[i] : (int32_t)i) : 0;  // vals_in == NULL: the values are the input positions (an argsort)
        const uint32_t dgt = ok ? (uint32_t)((key[k] >> shift) & 0xff) : 0xffffffffu;
        // match-any over the wave: mask of lanes holding the same digit (8 ballots)
        uint64_t m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((dgt >> b) & 1u);
            m &= ((dgt >> b) & 1u) ? bal : ~bal;
        }
        const uint64_t lower = m & ((1ull << lane) - 1ull);
        const uint32_t before = (uint32_t)__popcll(lower);
        uint32_t prev = 0;
        if (ok) {
            prev = cnt[w][dgt];  // same value for all lanes of the group (read before the write below)
        }
        // the read above must complete in every lane before the leader bumps the counter
        __builtin_amdgcn_wave_barrier();
        if (ok && lower == 0) cnt[w][dgt] = prev + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        rank[k] = prev + before;
    }
    __syncthreads();
    // per-digit exclusive scan across the 4 waves -> base of (wave, digit) inside the digit run; and the digit runs' starts
    // inside the block (exclusive scan of the block's digit totals over the 256 digits)
    __shared__ uint32_t dstart[RS_RADIX];
    {
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = cnt[k][d];
            cnt[k][d] = run;
            run += c;
        }
        int total;
        const int incl = gags_scan::block_incl_scan((int)run, total, scan_tmp);  // (two barriers inside)
        dstart[d] = (uint32_t)incl - run;
    }
    __syncthreads();
    // Stage the block's pairs in LDS in their sorted order, then write them out with consecutive threads on consecutive
    // addresses of a digit run.  (Written straight from the ranking registers, neighbouring lanes hold keys of different
    // digits: every 8- / 4-byte store was its own partial cache line -- rs_scatter<u64> ran at 1.8 TB/s.)
    __shared__ K keys_l[RS_TILE];
    __shared__ int32_t vals_l[RS_TILE];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int64_t i = wbase + k * 64 + lane;
        if (i < n) {
            const uint32_t dgt = (uint32_t)((key[k] >> shift) & 0xff);
            const uint32_t lp = dstart[dgt] + cnt[w][dgt] + rank[k];
            keys_l[lp] = key[k];
            vals_l[lp] = val[k];
        }
    }
    __syncthreads();
    const int64_t bbase = (int64_t)blockIdx.x * RS_TILE;
    const int nvalid = (int)((n - bbase) < (int64_t)RS_TILE ? (n - bbase) : (int64_t)RS_TILE);
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int i = k * RS_THREADS + threadIdx.x;
        if (i < nvalid) {
            const K kk = keys_l[i];
            const uint32_t dgt = (uint32_t)((kk >> shift) & 0xff);
            const uint32_t pos = gbase[dgt] + ((uint32_t)i - dstart[dgt]);
            keys_out[pos] = kk;
            vals_out[pos] = vals_l[i];
        }
    }
}

template <typename K>
int64_t sort_scratch_bytes_t(int64_t n_in)
{
    const int64_t n = n_in > 0 ? n_in : 1;
    const int64_t nblocks = (n + RS_TILE - 1) / RS_TILE;
    const int64_t keys = ((n * (int64_t)sizeof(K) + 255) / 256) * 256, vals = ((n * 4 + 255) / 256) * 256;
    const int64_t hist = ((nblocks * RS_RADIX * 4 + 255) / 256) * 256;
    return keys + vals + hist + RS_RADIX * 4;  // + the 256 digit totals
}

// stable LSD sort of (key, value) pairs on key bits [first_bit, first_bit + nbits); ping-pongs through
// `scratch` so that the last pass lands in keys_out / vals_out.
template <typename K>
int sort_pairs_t(int64_t n, int first_bit, int nbits, const K *keys_in, const int32_t *vals_in, K *keys_out,
                 int32_t *vals_out, void *scratch, int64_t scratch_bytes, hipStream_t st)
{
    if (n == 0) return GAGS_OK;
    if (n < 0 || n >= (1ll << 31) || nbits <= 0 || first_bit < 0 || first_bit + nbits > (int)(8 * sizeof(K)))
        return GAGS_EINVAL;
    if (!keys_in || !keys_out || !vals_out || !scratch) return GAGS_EINVAL;  // (vals_in may be NULL: argsort)
    if (scratch_bytes < sort_scratch_bytes_t<K>(n)) return GAGS_ESCRATCH;
    const int nblocks = (int)((n + RS_TILE - 1) / RS_TILE);
    const int64_t keys_b = ((n * (int64_t)sizeof(K) + 255) / 256) * 256, vals_b = ((n * 4 + 255) / 256) * 256;
    K *ktmp = (K *)scratch;
    int32_t *vtmp = (int32_t *)((char *)scratch + keys_b);
    uint32_t *hist = (uint32_t *)((char *)scratch + keys_b + vals_b);
    const int64_t hist_b = (((int64_t)nblocks * RS_RADIX * 4 + 255) / 256) * 256;
    uint32_t *totals = (uint32_t *)((char *)scratch + keys_b + vals_b + hist_b);
    const int passes = (nbits + 7) / 8;
    const K *src_k = keys_in;
    const int32_t *src_v = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        K *dst_k = to_out ? keys_out : ktmp;
        int32_t *dst_v = to_out ? vals_out : vtmp;
        hipLaunchKernelGGL(rs_histogram<K>, dim3(nblocks), dim3(RS_THREADS), 0, st, n, first_bit + p * 8, src_k, hist, nblocks);
        hipLaunchKernelGGL(rs_rowscan, dim3(RS_RADIX), dim3(RS_THREADS), 0, st, nblocks, hist, totals);
        hipLaunchKernelGGL(rs_scatter<K>, dim3(nblocks), dim3(RS_THREADS), 0, st, n, first_bit + p * 8, src_k, src_v, dst_k, dst_v,
                           hist, nblocks, totals);
        src_k = dst_k;
        src_v = dst_v;
    }
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

}  // namespace

// internal: 32-bit-key flavour used by the staged backward (rows sorted by Gaussian id)
int64_t gags_sort_u32_scratch_bytes(int64_t n) { return sort_scratch_bytes_t<uint32_t>(n); }
int gags_sort_pairs_u32(int64_t n, int nbits, const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out,
                        int32_t *vals_out, void *scratch, int64_t scratch_bytes, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    return sort_pairs_t<uint32_t>(n, 0, nbits, keys_in, vals_in, keys_out, vals_out, scratch, scratch_bytes, st);
}

extern "C" int64_t gags_sort_scratch_bytes(int64_t n_isects) { return sort_scratch_bytes_t<uint64_t>(n_isects); }

extern "C" int gags_sort_pairs(int64_t n, int tile_bits, int depth_sorted, const int64_t *keys_in,
                               const int32_t *vals_in, int64_t *keys_out, int32_t *vals_out, void *scratch,
                               int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || tile_bits < 0 || tile_bits > 31) return GAGS_EINVAL;
    // input already in depth order (gags_depth_order + ordered gags_tile_emit): the stable sort only has to
    // group by tile, 2 passes instead of 6 at 1080p
    const int first = depth_sorted ? 32 : 0, nbits = depth_sorted ? (tile_bits > 0 ? tile_bits : 1) : 32 + tile_bits;
    return sort_pairs_t<uint64_t>(n, first, nbits, (const uint64_t *)keys_in, vals_in, (uint64_t *)keys_out, vals_out,
                                  scratch, scratch_bytes, (hipStream_t)stream);
}

namespace {
__global__ __launch_bounds__(256) void gather_i32_kernel(int n, const int32_t *__restrict__ idx,
                                                         const int32_t *__restrict__ src, int32_t *__restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
inline int64_t al256s(int64_t x) { return (x + 255) / 256 * 256; }
}  // namespace

// K7a: depth order of the Gaussians (stable argsort of the depth bits: depths of visible Gaussians are
// positive, so their float bits order like the values; culled ones land anywhere and emit nothing) and the
// tile counts in that order.  N elements instead of n_isects: the per-intersection sort then only groups by tile.
extern "C" int64_t gags_depth_order_scratch_bytes(int n)
{
    const int64_t m = n > 0 ? n : 1;
    return 2 * al256s(m * 4) + sort_scratch_bytes_t<uint32_t>(m);
}

extern "C" int gags_depth_order(int n, const float *depths, const int32_t *tiles_per_gauss, int32_t *order,
                                int32_t *tiles_ordered, void *scratch, int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!depths || !order || !scratch || (tiles_ordered && !tiles_per_gauss)) return GAGS_EINVAL;
    if (scratch_bytes < gags_depth_order_scratch_bytes(n)) return GAGS_ESCRATCH;
    hipStream_t st = (hipStream_t)stream;
    char *sb = (char *)scratch;
    uint32_t *keys_sorted = (uint32_t *)(sb + al256s((int64_t)n * 4));
    void *sort_scratch = sb + 2 * al256s((int64_t)n * 4);
    // argsort: the first pass numbers the inputs itself (no iota kernel, no index array read)
    const int rc = sort_pairs_t<uint32_t>(n, 0, 32, reinterpret_cast<const uint32_t *>(depths), nullptr, keys_sorted, order,
                                          sort_scratch, sort_scratch_bytes_t<uint32_t>(n), st);
    if (rc != GAGS_OK) return rc;
    if (tiles_ordered)  // (optional: gags_cumsum_gather_i32 scans tiles_per_gauss[order[.]] without this copy)
        hipLaunchKernelGGL(gather_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, order, tiles_per_gauss,
                           tiles_ordered);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
