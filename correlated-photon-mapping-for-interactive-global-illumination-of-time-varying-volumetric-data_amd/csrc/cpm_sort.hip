// cpm_sort.hip -- stable LSD radix sort of u32 keys (+ u32 values), ascending (S1).
//
// Replaces clogs::Radixsort::enqueue (ref radixsortcl/ext/clogs/src/radixsort.cpp:169-259,
// kernels radixsortcl/ext/clogs/kernels/radixsort.cl:247-310,322-384,892-944).  clogs sorts
// 4 bits per pass with byte counters in local memory, a single-work-group scan, and 24
// launches for 32-bit keys; its autotuner and sqlite kernel cache are out of scope.
//
// MI355X design: 8 bits per pass, one pass = 3 launches:
//   hist    : per-tile digit histogram, 256 LDS counters per workgroup (ds_add_u32)
//   rowscan : 256 workgroups, one per digit: exclusive prefix of that digit's counts over tiles
//   scatter : per wave a ballot-based multi-split gives every key its stable rank among
//             the keys of its own digit (8 x v_cmp + s_and per key, popcount of the lanes
//             below), per-wave digit counters live in LDS; the tile is then reordered
//             through LDS so that each digit's run leaves the CU as contiguous stores.
// Wave size is 64 (ballots are 64-bit); a tile is 256 threads x ITEMS keys.
#include "cpm_ctx.h"

using namespace cpm;

namespace cpm {

constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;

template <int ITEMS>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                  int shift, uint32_t* __restrict__ hist,
                                                                  uint32_t num_tiles) {
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * (uint32_t)(kSortThreads * ITEMS);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        uint32_t i = base + k * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * num_tiles + tile] = h[threadIdx.x];
}

// Row scan: workgroup d turns row d of the digit-major table hist[256][tiles] into its
// exclusive prefix over tiles and writes the row total to digit_total[d].  256 independent
// coalesced scans, one per CU; the 256-entry scan over digit totals is done by every scatter
// workgroup for itself (it needs a 256-wide block scan anyway).
__global__ __launch_bounds__(kSortThreads) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t num_tiles,
                                                                     uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t wsum[kSortWaves];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t* row = hist + (size_t)blockIdx.x * num_tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_tiles; base += kSortThreads) {
        uint32_t i = base + t;
        uint32_t c = i < num_tiles ? row[i] : 0u;
        uint32_t v = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(v, off, 64);
            if (lane >= (uint32_t)off) v += o;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        uint32_t wave_off = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) { uint32_t x = wsum[w]; wave_off += (w < (int)wave) ? x : 0u; all += x; }
        if (i < num_tiles) row[i] = carry + wave_off + v - c;
        carry += all;
        __syncthreads();
    }
    if (t == 0) digit_total[blockIdx.x] = carry;
}

template <int ITEMS, bool HAS_VALUES>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, uint32_t n,
                                                                     int shift, const uint32_t* __restrict__ hist,
                                                                     const uint32_t* __restrict__ digit_total,
                                                                     uint32_t num_tiles) {
    constexpr int TILE = kSortThreads * ITEMS;
    __shared__ uint32_t wcount[kSortWaves][kRadix];  // per-wave digit counters, later (wave, digit) local starts
    __shared__ uint32_t gofs[kRadix];                // global start of the digit's run minus its local start
    __shared__ uint32_t wsum[kSortWaves];
    __shared__ uint32_t skeys[TILE];
    __shared__ uint32_t svals[HAS_VALUES ? TILE : 1];

    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = t >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t tile_base = tile * (uint32_t)TILE;
    const uint32_t tile_count = (n - tile_base) < (uint32_t)TILE ? (n - tile_base) : (uint32_t)TILE;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wcount[w][t] = 0;
    __syncthreads();

    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const uint32_t wave_base = tile_base + wave * (uint32_t)(64 * ITEMS);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = wave_base + j * 64 + lane;
        bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0xffffffffu;
        if (HAS_VALUES) val[j] = valid ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = wave_base + j * 64 + lane;
        bool valid = i < n;
        uint32_t d = (key[j] >> shift) & (kRadix - 1);
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            bool bit = (d >> b) & 1u;
            uint64_t vote = __ballot(bit);
            m &= bit ? vote : ~vote;
        }
        uint32_t below = (uint32_t)__popcll(m & lt_mask);
        uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t prev = 0;
        if (valid) {
            prev = wcount[wave][d];                        // same address within the group: LDS broadcast
            if (below == 0) wcount[wave][d] = prev + cnt;  // group leader; a wave's LDS ops stay in order
        }
        rank[j] = prev + below;
    }
    __syncthreads();

    // thread t owns digit t: wave prefixes, then an exclusive scan over the 256 digit totals
    uint32_t c[kSortWaves], total = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) { c[w] = wcount[w][t]; total += c[w]; }
    uint32_t v = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(v, off, 64);
        if (lane >= (uint32_t)off) v += o;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    uint32_t wave_off = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wave_off += (w < (int)wave) ? wsum[w] : 0u;
    uint32_t local_start = wave_off + v - total;
    uint32_t run = local_start;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) { wcount[w][t] = run; run += c[w]; }
    // global start of digit t's run = (keys of smaller digits) + (same digit in earlier tiles)
    uint32_t gt = digit_total[t];
    uint32_t gv = gt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(gv, off, 64);
        if (lane >= (uint32_t)off) gv += o;
    }
    __syncthreads();  // wsum is about to be reused
    if (lane == 63) wsum[wave] = gv;
    __syncthreads();
    uint32_t gwave_off = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) gwave_off += (w < (int)wave) ? wsum[w] : 0u;
    gofs[t] = (gwave_off + gv - gt) + hist[(size_t)t * num_tiles + tile] - local_start;
    __syncthreads();

#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = wave_base + j * 64 + lane;
        if (i < n) {
            uint32_t d = (key[j] >> shift) & (kRadix - 1);
            uint32_t lp = wcount[wave][d] + rank[j];
            skeys[lp] = key[j];
            if (HAS_VALUES) svals[lp] = val[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        uint32_t i = k * kSortThreads + t;
        if (i < tile_count) {
            uint32_t kk = skeys[i];
            uint32_t d = (kk >> shift) & (kRadix - 1);
            uint32_t pos = gofs[d] + i;
            keys_out[pos] = kk;
            if (HAS_VALUES) vals_out[pos] = svals[i];
        }
    }
}

// res_keys / res_vals (nullable): when given, the result is NOT copied back after an odd number of
// passes; instead they receive the buffers that hold it (keys/vals or the scratch ping-pong).
template <int ITEMS>
static int sort_passes(cpm_ctx* ctx, uint32_t* keys, uint32_t* vals, uint32_t n, int key_bits, hipStream_t s,
                       uint32_t** res_keys, uint32_t** res_vals) {
    const uint32_t tile = kSortThreads * ITEMS;
    const uint32_t num_tiles = (n + tile - 1) / tile;
    uint32_t* k2 = (uint32_t*)scratch(ctx, CPM_SCR_SORT_KEYS, (size_t)n * 4);
    uint32_t* v2 = vals ? (uint32_t*)scratch(ctx, CPM_SCR_SORT_VALS, (size_t)n * 4) : nullptr;
    uint32_t* hist = (uint32_t*)scratch(ctx, CPM_SCR_SORT_HIST, ((size_t)kRadix * num_tiles + kRadix) * 4);
    if (!k2 || (vals && !v2) || !hist) return CPM_ERR_OUT_OF_MEMORY;
    uint32_t* digit_total = hist + (size_t)kRadix * num_tiles;
    uint32_t *ks = keys, *kd = k2, *vs = vals, *vd = v2;
    for (int shift = 0; shift < key_bits; shift += kRadixBits) {
        CPM_LAUNCH(ctx, radix_hist_kernel<ITEMS>, dim3(num_tiles), dim3(kSortThreads), 0, s, ks, n, shift, hist, num_tiles);
        CPM_LAUNCH(ctx, radix_rowscan_kernel, dim3(kRadix), dim3(kSortThreads), 0, s, hist, num_tiles, digit_total);
        if (vals)
            CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, true>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, vs, kd, vd, n, shift, hist, digit_total, num_tiles);
        else
            CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, false>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, nullptr, kd, nullptr, n, shift, hist, digit_total, num_tiles);
        CPM_LAUNCH_CHECK(ctx, "radix sort pass");
        uint32_t* tmp = ks; ks = kd; kd = tmp;
        tmp = vs; vs = vd; vd = tmp;
    }
    if (res_keys) {
        *res_keys = ks;
        if (res_vals) *res_vals = vs;
        return CPM_OK;
    }
    if (ks != keys) {  // odd number of passes: result back in place (clogs does the same, radixsort.cpp:250-256)
        CPM_HIP_CHECK(ctx, hipMemcpyAsync(keys, ks, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (vals) CPM_HIP_CHECK(ctx, hipMemcpyAsync(vals, vs, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    }
    return CPM_OK;
}

// keys/vals sorted in place; vals may be null (keys only)
int radix_sort(cpm_ctx* ctx, uint32_t* keys, uint32_t* vals, size_t n, int key_bits, hipStream_t s,
               uint32_t** res_keys, uint32_t** res_vals) {
    if (res_keys) *res_keys = keys;
    if (res_vals) *res_vals = vals;
    if (n <= 1) return CPM_OK;
    if (n >= (1ull << 31)) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "radix_sort", "n must be < 2^31");
    if (key_bits <= 0 || key_bits > 32) key_bits = 32;
    // enough workgroups to cover 256 CUs several times over at small n
    if (n <= (1u << 21)) return sort_passes<4>(ctx, keys, vals, (uint32_t)n, key_bits, s, res_keys, res_vals);
    if (n <= (1u << 23)) return sort_passes<8>(ctx, keys, vals, (uint32_t)n, key_bits, s, res_keys, res_vals);
    return sort_passes<16>(ctx, keys, vals, (uint32_t)n, key_bits, s, res_keys, res_vals);
}

}  // namespace cpm

extern "C" {

int cpm_sort_pairs(cpm_ctx* ctx, uint32_t* keys, uint32_t* values, size_t n, int key_bits, cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    CPM_REQUIRE(ctx, (keys && values) || n == 0, "cpm_sort_pairs: null argument");
    return cpm::radix_sort(ctx, keys, values, n, key_bits, (hipStream_t)stream, nullptr, nullptr);
}

int cpm_sort_keys(cpm_ctx* ctx, uint32_t* keys, size_t n, int key_bits, cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    CPM_REQUIRE(ctx, keys || n == 0, "cpm_sort_keys: null argument");
    return cpm::radix_sort(ctx, keys, nullptr, n, key_bits, (hipStream_t)stream, nullptr, nullptr);
}

}  // extern "C"
