// cpm_sort.hip -- stable LSD radix sort of u32 keys (+ u32 values), ascending (S1).
//
// Replaces clogs::Radixsort::enqueue (ref radixsortcl/ext/clogs/src/radixsort.cpp:169-259,
// kernels radixsortcl/ext/clogs/kernels/radixsort.cl:247-310,322-384,892-944).  clogs sorts
// 4 bits per pass with byte counters in local memory, a single-work-group scan, and 24
// launches for 32-bit keys; its autotuner and sqlite kernel cache are out of scope.
//
// MI355X design: 8 bits per pass, one pass = 3 launches (a single-launch onesweep variant with ticketed
// tiles and decoupled look-back is kept as a tested alternative; it measured slower here):
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

// Row scan: WAVE d turns row d of the digit-major table hist[256][tiles] into its exclusive prefix over tiles and
// writes the row total to digit_total[d].  A row of 512 tiles is 8 entries per lane: all 8 loads are issued first,
// the 8 wave scans run in registers and one carry chain joins them -- one round of load latency and no barrier
// (a 256-thread workgroup per row with two barriers per 256 entries took 4.3 us for this 512 KiB table).  The
// 256-entry scan over digit totals is done by every scatter workgroup for itself (it needs a 256-wide block scan
// anyway).
__global__ __launch_bounds__(kSortThreads) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t num_tiles,
                                                                     uint32_t* __restrict__ digit_total) {
    constexpr int U = 8;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t digit = blockIdx.x * kSortWaves + wave;  // 64 workgroups x 4 waves = 256 rows
    uint32_t* row = hist + (size_t)digit * num_tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_tiles; base += 64 * U) {
        uint32_t c[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = base + u * 64 + lane;
            c[u] = i < num_tiles ? row[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = c[u];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(v[u], off, 64);
                if (lane >= (uint32_t)off) v[u] += o;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = base + u * 64 + lane;
            if (i < num_tiles) row[i] = carry + v[u] - c[u];
            carry += __shfl(v[u], 63, 64);
        }
    }
    if (lane == 0) digit_total[digit] = carry;
}

// status word of the look-back: 2 flag bits + 30 value bits in ONE 32-bit word, so publishing needs
// no ordering between separate locations (agent-scope relaxed atomic store / load = sc1 accesses).
constexpr uint32_t kFlagAggregate = 1u << 30, kFlagPrefix = 2u << 30, kValueMask = (1u << 30) - 1u;
constexpr uint32_t kSpinLimit = 1u << 24;  // polls before a wait gives up (sets the error word; never hangs the GPU)

// global digit histograms of every pass in one read of the keys (onesweep's only pre-pass)
__global__ __launch_bounds__(kSortThreads) void radix_global_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                         int passes, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[4][kRadix];
#pragma unroll
    for (int p = 0; p < 4; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * kSortThreads + threadIdx.x; i < n; i += gridDim.x * kSortThreads) {
        const uint32_t k = keys[i];
        for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(k >> (kRadixBits * p)) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * kRadix + threadIdx.x], c);
    }
}

// ONESWEEP = false: tile offsets come from the (hist, rowscan) pre-passes (3 launches per pass).
// ONESWEEP = true : one launch per pass.  Tiles take a ticket (so every earlier tile is already running:
//   forward progress does not depend on dispatch order), publish their per-digit counts, and obtain the
//   count of the same digit in all earlier tiles by decoupled look-back over the status words
//   (Merrill & Garland's single-pass scan, as used by onesweep radix sorts).
// BINSINK (the last pass of cpm_bin's sort): the tile's output loop also does what bin_finalize_kernel would do in
// a launch of its own -- order[pos] = photon index, the compact (pos, power) record fetched from the photon
// array, and the run starts of the cell table.  Inside a tile's digit run the output is final and contiguous,
// so "key differs from its predecessor" is decided in LDS; only the first element of a digit run of a tile
// cannot see its predecessor (it lies in another tile) and takes an atomicMin on the preset table instead.
template <int ITEMS, bool HAS_VALUES, bool ONESWEEP, bool BINSINK>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, uint32_t n,
                                                                     int shift, const uint32_t* __restrict__ hist,
                                                                     const uint32_t* __restrict__ digit_total,
                                                                     uint32_t num_tiles, uint32_t* __restrict__ ticket,
                                                                     uint32_t* __restrict__ status, uint32_t* __restrict__ error,
                                                                     const BinSink sink) {
    constexpr int TILE = kSortThreads * ITEMS;
    __shared__ uint32_t s_tile;
    __shared__ uint32_t wcount[kSortWaves][kRadix];  // per-wave digit counters, later (wave, digit) local starts
    __shared__ uint32_t gofs[kRadix];                // global start of the digit's run minus its local start
    __shared__ uint32_t wsum[kSortWaves];
    __shared__ uint32_t skeys[TILE];
    __shared__ uint32_t svals[HAS_VALUES ? TILE : 1];

    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = t >> 6;
    if (ONESWEEP && t == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wcount[w][t] = 0;
    __syncthreads();
    const uint32_t tile = ONESWEEP ? s_tile : blockIdx.x;
    const uint32_t tile_base = tile * (uint32_t)TILE;
    const uint32_t tile_count = (n - tile_base) < (uint32_t)TILE ? (n - tile_base) : (uint32_t)TILE;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

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
    uint32_t earlier;  // keys with digit t in earlier tiles
    if (ONESWEEP) {
        uint32_t* mine = status + (size_t)tile * kRadix + t;
        earlier = 0;
        if (tile == 0) {
            __hip_atomic_store(mine, total | kFlagPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(mine, total | kFlagAggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t polls = 0;
            for (int p = (int)tile - 1; p >= 0;) {
                const uint32_t w = __hip_atomic_load(status + (size_t)p * kRadix + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((w >> 30) == 0) {  // not published yet
                    if (++polls > kSpinLimit) { atomicOr(error, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                earlier += w & kValueMask;
                if (w & kFlagPrefix) break;
                --p;
            }
            __hip_atomic_store(mine, ((earlier + total) & kValueMask) | kFlagPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        earlier = hist[(size_t)t * num_tiles + tile];
    }
    gofs[t] = (gwave_off + gv - gt) + earlier - local_start;
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
            if (BINSINK) {
                const uint32_t id = svals[i];
                sink.order[pos] = id;
                const float4* q = rec_at(sink.photons, sink.rec, (size_t)id);
                const float4 a = q[0];
                if (sink.channels == 1) {
                    reinterpret_cast<float4*>(sink.sorted)[pos] = a;
                } else {
                    const float4 b = q[sink.rec.b];
                    float4* o = reinterpret_cast<float4*>(sink.sorted) + 2 * (size_t)pos;
                    o[0] = a;
                    o[1] = make_float4(b.x, b.y, 0.f, 0.f);
                }
                const uint32_t pk = i ? skeys[i - 1] : 0u;
                if (i == 0 || ((pk >> shift) & (kRadix - 1)) != d) atomicMin(&sink.cell_start[kk], pos);  // predecessor in another tile
                else if (pk != kk) sink.cell_start[kk] = pos;                                              // run start, decided here
            } else if (HAS_VALUES) {
                vals_out[pos] = svals[i];
            }
        }
    }
}

// res_keys / res_vals (nullable): when given, the result is NOT copied back after an odd number of
// passes; instead they receive the buffers that hold it (keys/vals or the scratch ping-pong).
// 0 = hist + rowscan + scatter (default), 1 = onesweep (cpm_debug_set_sort_mode).  Measured on MI355X at 1 M pairs:
// 20 us per pass for the three short launches vs 39 us for the single onesweep launch (+25 us for its
// global histogram): with all ~1000 tiles in flight at once the per-digit look-back chain is long.

template <int ITEMS>
static int sort_passes(cpm_ctx* ctx, uint32_t* keys, uint32_t* vals, uint32_t n, int key_bits, hipStream_t s,
                       uint32_t** res_keys, uint32_t** res_vals, const BinSink* sink, bool* sink_done, bool first_hist_done) {
    const uint32_t tile = kSortThreads * ITEMS;
    const uint32_t num_tiles = (n + tile - 1) / tile;
    const int passes = (key_bits + kRadixBits - 1) / kRadixBits;
    const bool onesweep = ctx->dbg.sort_mode == 1 && passes <= 4;
    uint32_t* k2 = (uint32_t*)scratch(ctx, CPM_SCR_SORT_KEYS, (size_t)n * 4);
    uint32_t* v2 = vals ? (uint32_t*)scratch(ctx, CPM_SCR_SORT_VALS, (size_t)n * 4) : nullptr;
    // control block: [error | tickets[4] | pad | ghist[4][256] | status[passes][tiles][256]]  or  hist[256][tiles] + totals
    const size_t ctl_words = 8 + 4 * (size_t)kRadix + (size_t)(onesweep ? passes : 1) * num_tiles * kRadix + kRadix;
    uint32_t* ctl = (uint32_t*)scratch(ctx, CPM_SCR_SORT_HIST, ctl_words * 4);
    if (!k2 || (vals && !v2) || !ctl) return CPM_ERR_OUT_OF_MEMORY;
    uint32_t* error = ctl;
    uint32_t* tickets = ctl + 1;
    uint32_t* ghist = ctl + 8;
    uint32_t* status = ghist + 4 * kRadix;       // onesweep: per pass [tiles][256]; else: hist[256][tiles]
    uint32_t* digit_total = status + (size_t)num_tiles * kRadix;  // non-onesweep only
    uint32_t *ks = keys, *kd = k2, *vs = vals, *vd = v2;
    if (onesweep) {
        CPM_HIP_CHECK(ctx, hipMemsetAsync(ctl, 0, (8 + 4 * (size_t)kRadix + (size_t)passes * num_tiles * kRadix) * 4, s));
        const int hist_blocks = (int)(num_tiles < 1024u ? num_tiles : 1024u);
        CPM_LAUNCH(ctx, radix_global_hist_kernel, dim3(hist_blocks), dim3(kSortThreads), 0, s, ks, n, passes, ghist);
    }
    for (int p = 0; p < passes; ++p) {
        const int shift = p * kRadixBits;
        if (onesweep) {
            uint32_t* st = status + (size_t)p * num_tiles * kRadix;
            if (vals)
                CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, true, true, false>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, vs, kd, vd, n, shift,
                           nullptr, ghist + p * kRadix, num_tiles, tickets + p, st, error, BinSink{});
            else
                CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, false, true, false>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, nullptr, kd, nullptr, n,
                           shift, nullptr, ghist + p * kRadix, num_tiles, tickets + p, st, error, BinSink{});
        } else {
            if (!(p == 0 && first_hist_done))  // cpm_bin's key kernel has already counted pass 0's digits per tile
                CPM_LAUNCH(ctx, radix_hist_kernel<ITEMS>, dim3(num_tiles), dim3(kSortThreads), 0, s, ks, n, shift, status, num_tiles);
            CPM_LAUNCH(ctx, radix_rowscan_kernel, dim3(kRadix / kSortWaves), dim3(kSortThreads), 0, s, status, num_tiles, digit_total);
            if (vals && sink && p == passes - 1) {
                CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, true, false, true>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, vs, kd, vd, n, shift,
                           status, digit_total, num_tiles, nullptr, nullptr, nullptr, *sink);
                if (sink_done) *sink_done = true;
            } else if (vals)
                CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, true, false, false>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, vs, kd, vd, n, shift,
                           status, digit_total, num_tiles, nullptr, nullptr, nullptr, BinSink{});
            else
                CPM_LAUNCH(ctx, (radix_scatter_kernel<ITEMS, false, false, false>), dim3(num_tiles), dim3(kSortThreads), 0, s, ks, nullptr, kd, nullptr, n,
                           shift, status, digit_total, num_tiles, nullptr, nullptr, nullptr, BinSink{});
        }
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
// keys per thread of a tile (256 threads) for n keys.  Measured at 1 M pairs, 22 key bits (3 passes + copy-back):
// ITEMS 4: 73 us, 8: 61 us, 16: 70 us -- 2048-key tiles balance table size against workgroup count.
int sort_items_for(const cpm_ctx* ctx, size_t n) {
    const int forced = ctx->dbg.sort_items;  // 0 = by size; 4 / 8 / 16 force the tile size (tuning hook)
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    return n <= (1u << 15) ? 4 : (n <= (1u << 23) ? 8 : 16);
}

// Where pass 0's per-tile digit histogram goes (digit-major [256][tiles]) when the caller counts it itself while
// producing the keys (cpm_bin); nullptr when the sort will not use it (n <= 1, onesweep test mode).
uint32_t* sort_first_hist(cpm_ctx* ctx, size_t n, int key_bits, uint32_t* num_tiles_out) {
    if (n <= 1 || n >= (1ull << 31) || ctx->dbg.sort_mode == 1) return nullptr;
    if (key_bits <= 0 || key_bits > 32) key_bits = 32;
    const uint32_t tile = kSortThreads * (uint32_t)sort_items_for(ctx, n);
    const uint32_t num_tiles = ((uint32_t)n + tile - 1) / tile;
    const size_t ctl_words = 8 + 4 * (size_t)kRadix + (size_t)num_tiles * kRadix + kRadix;  // as sort_passes lays it out
    uint32_t* ctl = (uint32_t*)scratch(ctx, CPM_SCR_SORT_HIST, ctl_words * 4);
    if (!ctl) return nullptr;
    *num_tiles_out = num_tiles;
    return ctl + 8 + 4 * kRadix;
}

// After a sort of n > 1 keys in the default pass structure: digit_total[d] of the LAST pass (keys per digit value),
// device memory inside the sort's scratch block, valid until the next sort on this context.
const uint32_t* sort_last_digit_totals(cpm_ctx* ctx, size_t n) {
    if (n <= 1 || ctx->dbg.sort_mode == 1) return nullptr;
    const uint32_t tile = kSortThreads * (uint32_t)sort_items_for(ctx, n);
    const uint32_t num_tiles = ((uint32_t)n + tile - 1) / tile;
    const uint32_t* ctl = (const uint32_t*)ctx->scratch[CPM_SCR_SORT_HIST];
    return ctl ? ctl + 8 + 4 * kRadix + (size_t)num_tiles * kRadix : nullptr;
}

int radix_sort(cpm_ctx* ctx, uint32_t* keys, uint32_t* vals, size_t n, int key_bits, hipStream_t s,
               uint32_t** res_keys, uint32_t** res_vals, const BinSink* sink, bool* sink_done, bool first_hist_done) {
    if (res_keys) *res_keys = keys;
    if (res_vals) *res_vals = vals;
    if (sink_done) *sink_done = false;
    if (n <= 1) return CPM_OK;
    if (n >= (1ull << 31)) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "radix_sort", "n must be < 2^31");
    if (key_bits <= 0 || key_bits > 32) key_bits = 32;
    switch (sort_items_for(ctx, n)) {
        case 4: return sort_passes<4>(ctx, keys, vals, (uint32_t)n, key_bits, s, res_keys, res_vals, sink, sink_done, first_hist_done);
        case 8: return sort_passes<8>(ctx, keys, vals, (uint32_t)n, key_bits, s, res_keys, res_vals, sink, sink_done, first_hist_done);
        default: return sort_passes<16>(ctx, keys, vals, (uint32_t)n, key_bits, s, res_keys, res_vals, sink, sink_done, first_hist_done);
    }
}

}  // namespace cpm

extern "C" {

// test / measurement hook (include/cpm/cpm_profile.h): 0 = hist + rowscan + scatter (default), 1 = onesweep passes
void cpm_debug_set_sort_mode(cpm_ctx* ctx, int mode) { if (ctx) ctx->dbg.sort_mode = mode; }
void cpm_debug_set_sort_items(cpm_ctx* ctx, int items) { if (ctx) ctx->dbg.sort_items = items; }

int cpm_sort_pairs(cpm_ctx* ctx, uint32_t* keys, uint32_t* values, size_t n, int key_bits, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, (keys && values) || n == 0, "cpm_sort_pairs: null argument");
    return cpm::radix_sort(ctx, keys, values, n, key_bits, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, false);
}

int cpm_sort_keys(cpm_ctx* ctx, uint32_t* keys, size_t n, int key_bits, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, keys || n == 0, "cpm_sort_keys: null argument");
    return cpm::radix_sort(ctx, keys, nullptr, n, key_bits, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, false);
}

}  // extern "C"
