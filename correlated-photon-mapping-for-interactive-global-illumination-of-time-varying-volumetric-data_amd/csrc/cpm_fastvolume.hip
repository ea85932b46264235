// cpm_fastvolume.hip -- photons -> light volume, the MI355X formulation in TOLERANCE MODE
// (cpm_bin_fast + cpm_gather_fast; include/cpm/cpm.h "fast formulation").
//
// What it computes: for every voxel the sum over photons of  power * k * 0.75 * (1 - d^2 / r^2)  for d <= r --
// the terms splatPhoton adds (ref progressivephotonmapping/cl/photonstolightvolume.cl:31-79, kernel
// cl/densityestimationkernel.cl:43-60) with the weight taken from d^2 directly (no sqrt, no division: the
// reference's own sum order is undefined -- CAS float atomics, :15-29 -- so only a tolerance is defined anyway).
// The bit-exact per-voxel sequential contract stays in cpm_bin / cpm_gather (cpm_lightvolume.hip).
//
// How (five short streaming launches instead of eleven latency chains):
//   bin   fast_count_kernel    photon -> brick key (bricks of 8x8x8 voxels, wider along x for big grids: <= 16 Ki
//                              bricks), per-workgroup histogram in LDS (ds_add_rtn_u32 = the photon's rank inside its
//                              (workgroup, brick) run), one returning global atomic per NON-EMPTY (workgroup, brick)
//                              pair = that run's offset inside the brick; max |power| on the way
//         fast_scan_kernel     one workgroup: brick starts, and the list of work items = (brick, chunk of <= 4096
//                              photons): a face brick with 8000 photons becomes 2 items, an empty brick none
//         fast_scatter_kernel  compact 16-byte (pos, power) records to brick_start[key] + rank
//   gather fast_tile_kernel    one workgroup per item: lanes own records, the brick's voxels (+ halo) are an LDS tile of
//                              64-bit FIXED-POINT sums (ds_add_u64): integer addition is associative, so the result does
//                              not depend on the order lanes, waves or items add in -- bitwise reproducible with no
//                              ordering protocol at all.  The tile is stored to the item's slab with plain stores
//         fast_combine_kernel  per voxel: the integer sum of the <= 8 slabs that cover it (own brick + neighbours'
//                              halos, every chunk), ONE rounding to float, plain coalesced store of the light volume
// No global float atomics, no sort passes, no per-voxel ordering: every launch streams its bytes once.
#include "cpm_ctx.h"

using namespace cpm;

namespace {

// tuning experiments build variants with -DCPM_TILE_THREADS=... etc. (tools/build_variant.sh); the defaults are the measured best
#ifndef CPM_TILE_THREADS
#define CPM_TILE_THREADS 1024
#endif
#ifndef CPM_FAST_CHUNK
#define CPM_FAST_CHUNK 4096
#endif
#ifndef CPM_COMBINE_THREADS
#define CPM_COMBINE_THREADS 128
#endif
#ifndef CPM_COUNT_ITEMS
#define CPM_COUNT_ITEMS 2
#endif
#ifndef CPM_TILE_WG_PER_CU
#define CPM_TILE_WG_PER_CU 2
#endif
constexpr int kTileThreads = CPM_TILE_THREADS;
constexpr int kCombineThreads = CPM_COMBINE_THREADS;
constexpr int kFastChunk = CPM_FAST_CHUNK;   // photons per tile-gather work item (1024 threads x 4)
constexpr int kCountItems = CPM_COUNT_ITEMS; // photons per thread of fast_count_kernel (1024 threads)
constexpr int kCountTile = 1024 * kCountItems;
constexpr int kMaxBricks = 16384;  // LDS histogram of fast_count_kernel: 64 KiB

// table layout (u32 entries): [0, nb] brick starts | 4 meta | [nb+5, 2nb+5] item starts | work items, 4 words each
// (brick, first record, end record, 0), 16-byte aligned
constexpr int kMetaMaxPow = 0, kMetaItems = 1;
// accumulator behind the scratch histogram (zero between calls, like the histogram): max |power| bits
constexpr int kAccMaxPow = 0;

struct BrickLayout {
    int lx, ly, lz;        // log2 brick size (voxels)
    int nbx, nby, nbz, nb; // bricks
    int hx, hy, hz;        // halo (voxels a photon of the brick can reach beyond it)
    int tx, ty, tz, tile;  // tile = brick + halo; voxels per tile
    int maxc;              // candidate voxels per axis
};
CPM_DEV uint32_t off_meta(const BrickLayout& L) { return (uint32_t)L.nb + 1u; }
CPM_DEV uint32_t off_item_start(const BrickLayout& L) { return (uint32_t)L.nb + 5u; }
CPM_DEV uint32_t off_items(const BrickLayout& L) { return (2u * (uint32_t)L.nb + 6u + 3u) & ~3u; }

__host__ int make_grid_dev_fast(cpm_ctx* ctx, const cpm_grid_desc* g, GridDev& G) {
    if (!g) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "null grid desc");
    if (g->dims[0] < 1 || g->dims[1] < 1 || g->dims[2] < 1) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "dims < 1");
    if ((unsigned long long)g->dims[0] * g->dims[1] * g->dims[2] >= (1ull << 31))
        return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "more than 2^31 cells");
    if (g->channels != 1 && g->channels != 4) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "channels must be 1 or 4");
    G.dx = g->dims[0]; G.dy = g->dims[1]; G.dz = g->dims[2]; G.channels = g->channels;
    if (!affine_from_matrix(g->texture_to_index, G.t2i) || !affine_from_matrix(g->index_to_texture, G.i2t))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "grid", "texture/index matrices must be scale + translate");
    if (!(G.t2i.sx > 0.f && G.t2i.sy > 0.f && G.t2i.sz > 0.f))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "grid", "fast formulation needs positive texture-to-index scales");
    return CPM_OK;
}

// brick shape from the grid alone: 8 x 8 x 8 voxels, doubled along x, y, z in turn while there are more than 16 Ki bricks
__host__ void brick_shape(const int dims[3], BrickLayout& L) {
    int lg[3] = { 3, 3, 3 };
    auto count = [&](int a) { return (dims[a] + (1 << lg[a]) - 1) >> lg[a]; };
    int axis = 0;
    while ((long long)count(0) * count(1) * count(2) > kMaxBricks) { ++lg[axis]; axis = (axis + 1) % 3; }
    L.lx = lg[0]; L.ly = lg[1]; L.lz = lg[2];
    L.nbx = count(0); L.nby = count(1); L.nbz = count(2);
    L.nb = L.nbx * L.nby * L.nbz;
    L.hx = L.hy = L.hz = 0; L.tx = L.ty = L.tz = L.tile = 0; L.maxc = 0;
}

// halo / tile / candidates from the radius (gather side); false when the tuned kernels do not cover it
__host__ bool brick_reach(const GridDev& G, float radius, BrickLayout& L) {
    const float rx = radius * G.t2i.sx, ry = radius * G.t2i.sy, rz = radius * G.t2i.sz;  // radius in voxels per axis
    L.hx = (int)floorf(rx + 0.501f); L.hy = (int)floorf(ry + 0.501f); L.hz = (int)floorf(rz + 0.501f);
    const float rmax = fmaxf(rx, fmaxf(ry, rz)) + 1e-3f;
    L.maxc = (int)floorf(2.f * rmax) + 1;
    L.tx = (1 << L.lx) + 2 * L.hx; L.ty = (1 << L.ly) + 2 * L.hy; L.tz = (1 << L.lz) + 2 * L.hz;
    L.tile = L.tx * L.ty * L.tz;
    if (!(radius > 0.f) || L.maxc > 4) return false;
    if (2 * L.hx > (1 << L.lx) || 2 * L.hy > (1 << L.ly) || 2 * L.hz > (1 << L.lz)) return false;
    return true;
}

__host__ size_t max_items_for(const BrickLayout& L, int n) {
    return (size_t)(n < L.nb ? n : L.nb) + (size_t)div_up(n > 0 ? n : 1, kFastChunk);
}
__host__ size_t table_entries(const BrickLayout& L, int n) {
    return (((size_t)2 * L.nb + 6 + 3) & ~(size_t)3) + 4 * max_items_for(L, n);
}

// the photon's cell (the voxel whose centre is nearest: floor(index + 0.5), clamped) and brick
CPM_DEV uint32_t brick_key(const GridDev& G, const BrickLayout& L, float4 a) {
    const f3 p = { a.x, a.y, a.z };
    const f3 u = transform_(G.t2i, p);
    const int cx = (int)min_(max_(__builtin_floorf(u.x + 0.5f), 0.0f), (float)(G.dx - 1));
    const int cy = (int)min_(max_(__builtin_floorf(u.y + 0.5f), 0.0f), (float)(G.dy - 1));
    const int cz = (int)min_(max_(__builtin_floorf(u.z + 0.5f), 0.0f), (float)(G.dz - 1));
    return (uint32_t)(cx >> L.lx) + (uint32_t)L.nbx * ((uint32_t)(cy >> L.ly) + (uint32_t)L.nby * (uint32_t)(cz >> L.lz));
}
CPM_DEV bool is_sentinel(float4 a) { return a.x == kFltMax || a.y == kFltMax || a.z == kFltMax; }

// Fixed-point scale 2^sh: every |contribution| <= m = maxpow * |k| * 0.75 < 2^e; sh = 30 - e makes every contribution
// fit a signed 32-bit integer (one v_cvt_i32_f32 -- there is no f32 -> i64 instruction; the generic conversion is 13), and
// fewer than 2^31 of them per voxel keep every partial sum inside int64.  Resolution: 2^-30 of the largest contribution,
// 64 times finer than the last bit of an fp32 sum of that size.  A function of (maxpow, k) only.
CPM_DEV float fixed_scale(float maxpow, float k) {
    const float m = maxpow * __builtin_fabsf(k) * 0.75f;
    if (!(m > 0.f) || m > kFltMax) return 1.0f;
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126;  // m < 2^e
    int sh = 30 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    return __uint_as_float((uint32_t)(sh + 127) << 23);
}
// value * S truncated toward zero, as a 64-bit addend
CPM_DEV unsigned long long to_fixed(float v, float S) {
    const int q = (int)(v * S);  // |v * S| < 2^30
    return (unsigned long long)(long long)q;
}

// Exclusive scan of the brick counts (-> brick starts, left in LDS for the caller's scatter) by one 1024-thread workgroup;
// with `table` (workgroup 0 only) also the scan of the bricks' chunk counts (-> item starts), the work items and the
// totals.  `counts` is the finished global histogram (every workgroup reads it: nb * 4 bytes of L2 traffic each).
CPM_DEV void fast_scan(const uint32_t* __restrict__ counts, const BrickLayout& L, uint32_t* __restrict__ s_start,
                       uint32_t* __restrict__ table, uint32_t* s_c, uint32_t* s_i) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (L.nb + 1023) / 1024;
    const int b0 = t * per, b1 = min(b0 + per, L.nb);
    uint32_t c = 0, it = 0;
    for (int b = b0; b < b1; ++b) { const uint32_t h = counts[b]; s_start[b] = h; c += h; it += (h + kFastChunk - 1) / kFastChunk; }
    uint32_t ci = c, ii = it;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t oc = __shfl_up(ci, off, 64), oi = __shfl_up(ii, off, 64);
        if (lane >= off) { ci += oc; ii += oi; }
    }
    if (lane == 63) { s_c[wave] = ci; s_i[wave] = ii; }
    __syncthreads();
    uint32_t bc = 0, bi = 0, tc = 0, ti = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) { bc += s_c[w]; bi += s_i[w]; } tc += s_c[w]; ti += s_i[w]; }
    uint32_t ac = bc + ci - c, ai = bi + ii - it;  // exclusive prefixes of this thread's first brick
    if (table) {
        uint32_t* __restrict__ bstart = table;
        uint32_t* __restrict__ istart = table + off_item_start(L);
        uint4* __restrict__ items = reinterpret_cast<uint4*>(table + off_items(L));
        for (int b = b0; b < b1; ++b) {
            const uint32_t h = s_start[b];
            bstart[b] = ac; istart[b] = ai; s_start[b] = ac;
            const uint32_t nc = (h + kFastChunk - 1) / kFastChunk;
            for (uint32_t q = 0; q < nc; ++q)
                items[ai + q] = make_uint4((uint32_t)b, ac + q * (uint32_t)kFastChunk, min(ac + (q + 1u) * (uint32_t)kFastChunk, ac + h), 0u);
            ac += h; ai += nc;
        }
        if (t == 0) { bstart[L.nb] = tc; istart[L.nb] = ti; table[off_meta(L) + kMetaItems] = ti; }
    } else {
        for (int b = b0; b < b1; ++b) { const uint32_t h = s_start[b]; s_start[b] = ac; ac += h; }
    }
}

// bin, launch 1 of 2.  Per workgroup (1024 threads, 4096 photons): brick keys, a histogram in LDS whose returning
// ds_add gives every photon its rank inside its (workgroup, brick) run, then ONE returning global atomic per non-empty
// (workgroup, brick) pair: the run's offset inside the brick.  rank = offset + local rank: an unstable counting sort.
template <int CH>
__global__ __launch_bounds__(1024) void fast_count_kernel(const float* __restrict__ photons, int n, GridDev G, BrickLayout L,
                                                          uint32_t* __restrict__ hist, uint32_t* __restrict__ acc,
                                                          uint32_t* __restrict__ rank) {
    extern __shared__ uint32_t s_hist[];
    __shared__ float s_mp[16];
    const int t = threadIdx.x;
    for (int b = t; b < L.nb; b += 1024) s_hist[b] = 0u;
    __syncthreads();
    const float4* __restrict__ ph = reinterpret_cast<const float4*>(photons);
    uint32_t key[kCountItems], lr[kCountItems];
    float mp = 0.f;
    float4 a[kCountItems];
#pragma unroll
    for (int k = 0; k < kCountItems; ++k) {  // the loads first, all in flight together
        const long long i = (long long)blockIdx.x * kCountTile + k * 1024 + t;
        a[k] = make_float4(kFltMax, kFltMax, kFltMax, 0.f);
        if (i < n) a[k] = ph[2 * i];
    }
#pragma unroll
    for (int k = 0; k < kCountItems; ++k) {
        const long long i = (long long)blockIdx.x * kCountTile + k * 1024 + t;
        key[k] = 0xffffffffu; lr[k] = 0u;
        if (i < n && !is_sentinel(a[k])) {
            key[k] = brick_key(G, L, a[k]);
            lr[k] = atomicAdd(&s_hist[key[k]], 1u);
            mp = max_(mp, __builtin_fabsf(a[k].w));
            if (CH == 4) { const float4 b = ph[2 * i + 1]; mp = max_(mp, max_(__builtin_fabsf(b.x), __builtin_fabsf(b.y))); }
        }
    }
    // max |power| of the workgroup (a finite, non-negative float orders like its bit pattern); NaN / inf are ignored
    if (!(mp <= kFltMax)) mp = 0.f;
    for (int off = 32; off > 0; off >>= 1) mp = max_(mp, __shfl_xor(mp, off, 64));
    if ((t & 63) == 0) s_mp[t >> 6] = mp;
    __syncthreads();
    // the runs' offsets: four bins at a time, the atomics of a group issued together
    for (int b = t; b < L.nb; b += 4 * 1024) {
        uint32_t c[4], base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int bb = b + q * 1024; c[q] = bb < L.nb ? s_hist[bb] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { base[q] = 0u; if (c[q]) base[q] = atomicAdd(&hist[b + q * 1024], c[q]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (c[q]) s_hist[b + q * 1024] = base[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCountItems; ++k) {
        const long long i = (long long)blockIdx.x * kCountTile + k * 1024 + t;
        if (i < n) rank[i] = key[k] != 0xffffffffu ? s_hist[key[k]] + lr[k] : 0xffffffffu;
    }
    // one atomic per workgroup at most, and none once the running maximum has reached this workgroup's
    if (t == 0) {
        float m = s_mp[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) m = max_(m, s_mp[w]);
        const uint32_t mb = __float_as_uint(m);
        if (m > 0.f && __hip_atomic_load(&acc[kAccMaxPow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mb) atomicMax(&acc[kAccMaxPow], mb);
    }
}

// bin, launch 2 of 2.  Every workgroup (1024 threads, kScatterItems photons each) turns the finished histogram into the
// brick starts it needs -- an exclusive scan of nb <= 16 Ki counts in LDS, while its photon loads are in flight -- and
// moves its photons' compact records to brick_start[key] + rank.  Workgroup 0 also writes the table the gather reads
// (brick starts, work items, max |power|).  The one-workgroup scan launch this replaces took 5.6 us of pure latency
// between two 10 us kernels.  The histogram is used in turn with a second one: this launch zeroes the OTHER one (idle
// until the next call), so no memset is needed in steady state and no workgroup has to know when the others have read.
constexpr int kScatterItems = 2;
constexpr int kScatterTile = 1024 * kScatterItems;
template <int CH>
__global__ __launch_bounds__(1024) void fast_scatter_kernel(const float* __restrict__ photons, int n, GridDev G, BrickLayout L,
                                                            const uint32_t* __restrict__ rank, const uint32_t* __restrict__ hist,
                                                            uint32_t* __restrict__ hist_next, int hist_words,
                                                            uint32_t* __restrict__ table, float* __restrict__ sorted) {
    extern __shared__ uint32_t s_start[];
    __shared__ uint32_t s_c[16], s_i[16];
    const int t = threadIdx.x;
    const float4* __restrict__ ph = reinterpret_cast<const float4*>(photons);
    const int n_tiles = (int)(((long long)n + kScatterTile - 1) / kScatterTile);
    uint32_t r[kScatterItems];
    float4 a[kScatterItems], b2[kScatterItems];
    auto load = [&](int tile) {
#pragma unroll
        for (int k = 0; k < kScatterItems; ++k) {
            const long long i = (long long)tile * kScatterTile + k * 1024 + t;
            r[k] = 0xffffffffu;
            a[k] = make_float4(0.f, 0.f, 0.f, 0.f); b2[k] = a[k];
            if (i < n) { r[k] = rank[i]; a[k] = ph[2 * i]; if (CH == 4) b2[k] = ph[2 * i + 1]; }
        }
    };
    load(blockIdx.x);  // the first tile's loads: in flight during the scan
    for (int w = blockIdx.x * 1024 + t; w < hist_words; w += gridDim.x * 1024) hist_next[w] = 0u;
    if (blockIdx.x == 0 && t == 0) table[off_meta(L) + kMetaMaxPow] = hist[L.nb + kAccMaxPow];
    fast_scan(hist, L, s_start, blockIdx.x == 0 ? table : nullptr, s_c, s_i);
    __syncthreads();
    // big inputs: a workgroup walks several tiles, so that the scan (nb counts per workgroup) is paid once per 2 Ki+ photons
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (tile != (int)blockIdx.x) load(tile);
#pragma unroll
        for (int k = 0; k < kScatterItems; ++k) {
            if (r[k] == 0xffffffffu) continue;
            const size_t pos = (size_t)s_start[brick_key(G, L, a[k])] + r[k];
            if (CH == 1) {
                // scattered 16-byte records, read next by another launch: streaming stores (14.0 -> 13.4 us; the same hint on the
                // tile launch's slabs costs the combine launch 2.6 us -- those it reads back at once, from the L2)
                typedef float v4 __attribute__((ext_vector_type(4)));
                const v4 q = { a[k].x, a[k].y, a[k].z, a[k].w };
                __builtin_nontemporal_store(q, reinterpret_cast<v4*>(sorted) + pos);
            } else {
                float4* o = reinterpret_cast<float4*>(sorted) + 2 * pos;
                o[0] = a[k];
                o[1] = make_float4(b2[k].x, b2[k].y, 0.f, 0.f);
            }
        }
    }
}

// One record into the LDS tile.  Candidates per axis: the integers within r' (+ slack) of the photon's index-space
// coordinate, clipped to the grid and to the tile; d^2 in texture space with the contract's operands
// (c = indexToTexture * v, d = c - p, d^2 = fma(dz, dz, fma(dy, dy, dx * dx))); weight 0.75 * (1 - d^2 / r^2) for
// d^2 <= r^2; value -> fixed point by truncation.
template <int MAXC, int CH>
CPM_DEV void tile_record(const GridDev& G, float4 a, float pg, float pb, int ox, int oy, int oz, int tx, int ty, int tz, float rgx,
                         float rgy, float rgz, float r2, float inv_r2, float k, float S, long long* __restrict__ tile, int plane) {
    const f3 p = { a.x, a.y, a.z };
    const f3 u = transform_(G.t2i, p);
    const int sx = max(max((int)__builtin_ceilf(u.x - rgx), 0), ox), ex = min(min((int)__builtin_floorf(u.x + rgx), G.dx - 1), ox + tx - 1);
    const int sy = max(max((int)__builtin_ceilf(u.y - rgy), 0), oy), ey = min(min((int)__builtin_floorf(u.y + rgy), G.dy - 1), oy + ty - 1);
    const int sz = max(max((int)__builtin_ceilf(u.z - rgz), 0), oz), ez = min(min((int)__builtin_floorf(u.z + rgz), G.dz - 1), oz + tz - 1);
    const float pk = a.w * k, pkg = pg * k, pkb = pb * k;
    float dxv[MAXC], dyv[MAXC], dzv[MAXC];
    bool okx[MAXC], oky[MAXC], okz[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
        const int vx = sx + q, vy = sy + q, vz = sz + q;
        dxv[q] = fma_(G.i2t.sx, (float)vx, G.i2t.tx) - a.x;
        dyv[q] = fma_(G.i2t.sy, (float)vy, G.i2t.ty) - a.y;
        dzv[q] = fma_(G.i2t.sz, (float)vz, G.i2t.tz) - a.z;
        okx[q] = vx <= ex; oky[q] = vy <= ey; okz[q] = vz <= ez;
    }
    const int base = (sx - ox) + tx * ((sy - oy) + ty * (sz - oz));
#pragma unroll
    for (int qz = 0; qz < MAXC; ++qz)
#pragma unroll
        for (int qy = 0; qy < MAXC; ++qy)
#pragma unroll
            for (int qx = 0; qx < MAXC; ++qx) {
                const float d2 = fma_(dzv[qz], dzv[qz], fma_(dyv[qy], dyv[qy], dxv[qx] * dxv[qx]));
                // the value is formed for every candidate (7 instructions); only the LDS add is conditional
                const float w = 0.75f * (1.0f - d2 * inv_r2);
                const int idx = base + qx + tx * (qy + ty * qz);
                const bool hit = okx[qx] && oky[qy] && okz[qz] && d2 <= r2;
                const unsigned long long q0 = to_fixed(pk * w, S);
                if (hit) atomicAdd(reinterpret_cast<unsigned long long*>(tile + idx), q0);
                if (CH == 4) {
                    const unsigned long long q1 = to_fixed(pkg * w, S), q2 = to_fixed(pkb * w, S);
                    if (hit) {
                        atomicAdd(reinterpret_cast<unsigned long long*>(tile + plane + idx), q1);
                        atomicAdd(reinterpret_cast<unsigned long long*>(tile + 2 * plane + idx), q2);
                    }
                }
            }
}

// gather, launch 1 of 2.  A fixed grid of resident workgroups (two per CU) walks the work items: the number of items
// is only known on the device, and a launch sized for the worst case (bricks + chunks) spent its time dispatching
// thousands of 16-wave workgroups that found nothing to do.  The next item's records are requested before the current
// item's are processed.
template <int MAXC, int CH>
__global__ __launch_bounds__(kTileThreads) void fast_tile_kernel(const float* __restrict__ sorted, const uint32_t* __restrict__ table, GridDev G,
                                                        BrickLayout L, float radius, float k,
                                                        long long* __restrict__ slabs) {
    extern __shared__ long long s_tile[];
    constexpr int CH3 = CH == 4 ? 3 : 1, STRIDE = CH == 4 ? 2 : 1, PER = kFastChunk / kTileThreads;
    const uint32_t n_items = table[off_meta(L) + kMetaItems];
    const int t = threadIdx.x;
    const int words = CH3 * L.tile;
    const float rgx = radius * G.t2i.sx + 1e-3f, rgy = radius * G.t2i.sy + 1e-3f, rgz = radius * G.t2i.sz + 1e-3f;
    const float r2 = radius * radius, inv_r2 = 1.0f / r2;
    const float S = fixed_scale(__uint_as_float(table[off_meta(L) + kMetaMaxPow]), k);
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
    const uint4* __restrict__ items = reinterpret_cast<const uint4*>(table + off_items(L));
    auto fetch = [&](uint32_t item, uint4& desc, float4* a, float4* a2) {
        desc = make_uint4(0u, 0u, 0u, 0u);
        if (item < n_items) desc = items[item];  // (brick, first record, end record, -)
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t j = desc.y + (uint32_t)(q * kTileThreads + t);
            a[q] = make_float4(0.f, 0.f, 0.f, 0.f); a2[q] = a[q];
            if (j < desc.z) { a[q] = rec[STRIDE * (size_t)j]; if (CH == 4) a2[q] = rec[2 * (size_t)j + 1]; }
        }
    };
    uint4 desc, desc_n;
    float4 a[PER], a2[PER], an[PER], an2[PER];
    fetch(blockIdx.x, desc, a, a2);
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        fetch(item + gridDim.x, desc_n, an, an2);  // the next item's records: in flight while this item is processed
        const uint32_t b = desc.x, j0 = desc.y, j1 = desc.z;
        for (int w = t; w < words; w += kTileThreads) s_tile[w] = 0ll;
        const int bx = (int)(b % (uint32_t)L.nbx), by = (int)((b / (uint32_t)L.nbx) % (uint32_t)L.nby), bz = (int)(b / (uint32_t)(L.nbx * L.nby));
        const int ox = (bx << L.lx) - L.hx, oy = (by << L.ly) - L.hy, oz = (bz << L.lz) - L.hz;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t j = j0 + (uint32_t)(q * kTileThreads + t);
            if (j < j1)
                tile_record<MAXC, CH>(G, a[q], a2[q].x, a2[q].y, ox, oy, oz, L.tx, L.ty, L.tz, rgx, rgy, rgz, r2, inv_r2, k, S, s_tile, L.tile);
        }
        __syncthreads();
        long long* __restrict__ slab = slabs + (size_t)item * (size_t)words;
        for (int w = t; w < words; w += kTileThreads) slab[w] = s_tile[w];
        __syncthreads();  // the tile is cleared again at the top
        desc = desc_n;
#pragma unroll
        for (int q = 0; q < PER; ++q) { a[q] = an[q]; a2[q] = an2[q]; }
    }
}

// per voxel: integer sum of every slab that covers it (chunks of the own brick and of the <= 26 neighbours whose halo
// reaches it), one rounding to float
template <int CH>
__global__ __launch_bounds__(kCombineThreads) void fast_combine_kernel(const long long* __restrict__ slabs, const uint32_t* __restrict__ table,
                                                           GridDev G, BrickLayout L, float k, int accumulate,
                                                           float* __restrict__ out) {
    constexpr int CH3 = CH == 4 ? 3 : 1;
    __shared__ uint32_t s_lo[27], s_hi[27];
    __shared__ int s_any;
    const int t = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const int bx = (int)(b % (uint32_t)L.nbx), by = (int)((b / (uint32_t)L.nbx) % (uint32_t)L.nby), bz = (int)(b / (uint32_t)(L.nbx * L.nby));
    if (t == 0) s_any = 0;
    __syncthreads();
    if (t < 27) {
        const int dx = t % 3 - 1, dy = (t / 3) % 3 - 1, dz = t / 9 - 1;
        const int qx = bx + dx, qy = by + dy, qz = bz + dz;
        uint32_t lo = 0, hi = 0;
        if (qx >= 0 && qx < L.nbx && qy >= 0 && qy < L.nby && qz >= 0 && qz < L.nbz) {
            const uint32_t q = (uint32_t)qx + (uint32_t)L.nbx * ((uint32_t)qy + (uint32_t)L.nby * (uint32_t)qz);
            lo = table[off_item_start(L) + q]; hi = table[off_item_start(L) + q + 1];
        }
        s_lo[t] = lo; s_hi[t] = hi;
        if (hi > lo) s_any = 1;
    }
    __syncthreads();
    const bool any = s_any != 0;
    if (!any && accumulate) return;
    const float S = fixed_scale(__uint_as_float(table[off_meta(L) + kMetaMaxPow]), k);
    const float invS = 1.0f / S;  // a power of two: exact
    const int BX = 1 << L.lx, BY = 1 << L.ly, bvox = BX * BY * (1 << L.lz);
    const size_t words = (size_t)CH3 * (size_t)L.tile;
    for (int v = t; v < bvox; v += kCombineThreads) {
        const int lx = v & (BX - 1), ly = (v >> L.lx) & (BY - 1), lz = v >> (L.lx + L.ly);
        const int gx = (bx << L.lx) + lx, gy = (by << L.ly) + ly, gz = (bz << L.lz) + lz;
        if (gx >= G.dx || gy >= G.dy || gz >= G.dz) continue;
        long long sr = 0, sg = 0, sb = 0;
        if (any) {
            // the tiles that cover this voxel: its own brick's, and per axis at most ONE neighbour's halo (2 * halo <= brick):
            // the lower neighbour's when the voxel lies in the brick's first `halo` layers, the upper one's in the last
            const int BZ = 1 << L.lz;
            const int nx = lx < L.hx ? -1 : (lx >= BX - L.hx ? 1 : 0);
            const int ny = ly < L.hy ? -1 : (ly >= BY - L.hy ? 1 : 0);
            const int nz = lz < L.hz ? -1 : (lz >= BZ - L.hz ? 1 : 0);
            // the first chunk of every covering tile: up to eight independent loads in flight; bricks with more chunks
            // (more than 4096 photons) are rare and finish in the loop behind
            long long v0[8], v1[8], v2[8];
            uint32_t lo_[8], hi_[8];
            size_t idx_[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ax = (c & 1) ? nx : 0, ay = (c & 2) ? ny : 0, az = (c & 4) ? nz : 0;
                const bool covers = !((c & 1) && nx == 0) && !((c & 2) && ny == 0) && !((c & 4) && nz == 0);
                const int d = (ax + 1) + 3 * (ay + 1) + 9 * (az + 1);
                lo_[c] = s_lo[d]; hi_[c] = covers ? s_hi[d] : 0u;
                // this voxel inside that brick's tile (origin = brick origin - halo)
                const int ix = lx - ax * BX + L.hx, iy = ly - ay * BY + L.hy, iz = lz - az * BZ + L.hz;
                idx_[c] = (size_t)ix + (size_t)L.tx * ((size_t)iy + (size_t)L.ty * (size_t)iz);
                v0[c] = 0; v1[c] = 0; v2[c] = 0;
                if (hi_[c] > lo_[c]) {
                    const long long* __restrict__ slab = slabs + (size_t)lo_[c] * words;
                    v0[c] = slab[idx_[c]];
                    if (CH == 4) { v1[c] = slab[(size_t)L.tile + idx_[c]]; v2[c] = slab[2 * (size_t)L.tile + idx_[c]]; }
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) { sr += v0[c]; if (CH == 4) { sg += v1[c]; sb += v2[c]; } }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                for (uint32_t item = lo_[c] + 1; item < hi_[c]; ++item) {
                    const long long* __restrict__ slab = slabs + (size_t)item * words;
                    sr += slab[idx_[c]];
                    if (CH == 4) { sg += slab[(size_t)L.tile + idx_[c]]; sb += slab[2 * (size_t)L.tile + idx_[c]]; }
                }
        }
        const size_t o = (size_t)gx + (size_t)G.dx * ((size_t)gy + (size_t)G.dy * (size_t)gz);
        const float fr = (float)sr * invS;
        if (CH == 1) {
            out[o] = accumulate ? out[o] + fr : fr;
        } else {
            const float fg = (float)sg * invS, fb = (float)sb * invS;
            float4* q = reinterpret_cast<float4*>(out) + o;
            if (accumulate) { const float4 tt = *q; *q = make_float4(tt.x + fr, tt.y + fg, tt.z + fb, tt.w); }
            else *q = make_float4(fr, fg, fb, 0.f);
        }
    }
}

template <typename K>
__host__ int allow_lds(cpm_ctx* ctx, K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return CPM_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return set_error(ctx, CPM_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", hipGetErrorString(e));
    return CPM_OK;
}

}  // namespace

extern "C" {

size_t cpm_fast_table_entries(const cpm_grid_desc* grid, int n) {
    if (!grid || n < 0 || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1) return 0;
    BrickLayout L;
    brick_shape(grid->dims, L);
    return table_entries(L, n);
}

int cpm_gather_fast_supported(const cpm_grid_desc* grid, float radius) {
    if (!grid || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1) return 0;
    if (grid->channels != 1 && grid->channels != 4) return 0;
    GridDev G;
    G.dx = grid->dims[0]; G.dy = grid->dims[1]; G.dz = grid->dims[2]; G.channels = grid->channels;
    if (!affine_from_matrix(grid->texture_to_index, G.t2i) || !affine_from_matrix(grid->index_to_texture, G.i2t)) return 0;
    if (!(G.t2i.sx > 0.f && G.t2i.sy > 0.f && G.t2i.sz > 0.f)) return 0;
    BrickLayout L;
    brick_shape(grid->dims, L);
    if (!brick_reach(G, radius, L)) return 0;
    return (size_t)(G.channels == 4 ? 3 : 1) * (size_t)L.tile * 8 <= 160 * 1024 - 1024;
}

int cpm_bin_fast(cpm_ctx* ctx, const float* photons8, int n, const cpm_grid_desc* grid, uint32_t* brick_table,
                 float* sorted_pos_power, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev_fast(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n >= 0, "cpm_bin_fast: n < 0");
    CPM_REQUIRE(ctx, brick_table, "cpm_bin_fast: null brick_table");
    CPM_REQUIRE(ctx, n == 0 || (photons8 && sorted_pos_power), "cpm_bin_fast: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_bin_fast");
    CPM_REQUIRE_ALIGNED16(ctx, sorted_pos_power, "cpm_bin_fast");
    hipStream_t s = (hipStream_t)stream;
    BrickLayout L;
    brick_shape(grid->dims, L);
    // scratch: two histograms (nb brick counts + 4 accumulators each), used in turn -- a call's scatter launch zeroes the
    // one the NEXT call counts into -- then the per-photon ranks
    const size_t hist_words = (size_t)L.nb + 4;
    const size_t arena = (2 * hist_words + (size_t)(n > 0 ? n : 1)) * 4;
    const bool had = ctx->scratch_bytes[CPM_SCR_FAST_BIN] >= arena;
    uint32_t* base = (uint32_t*)scratch(ctx, CPM_SCR_FAST_BIN, arena);
    if (!base) return CPM_ERR_OUT_OF_MEMORY;
    if (!had || ctx->fast_hist_words != hist_words) {  // new arena or another brick count: the zero state is not established
        CPM_HIP_CHECK(ctx, hipMemsetAsync(base, 0, 2 * hist_words * 4, s));
        ctx->fast_hist_parity = 0;
    }
    uint32_t* hist = base + (size_t)ctx->fast_hist_parity * hist_words;
    uint32_t* hist_next = base + (size_t)(ctx->fast_hist_parity ^ 1) * hist_words;
    uint32_t* acc = hist + L.nb;
    uint32_t* rank = base + 2 * hist_words;
    ctx->fast_hist_words = 0;  // re-established below once the kernel that restores the zero state is enqueued
    const size_t lds = (size_t)L.nb * 4;
    if (n > 0) {
        const dim3 cgrid((unsigned)div_up(n, kCountTile));
        if (G.channels == 1) {
            rc = allow_lds(ctx, fast_count_kernel<1>, lds); if (rc) return rc;
            CPM_LAUNCH(ctx, fast_count_kernel<1>, cgrid, dim3(1024), lds, s, photons8, n, G, L, hist, acc, rank);
        } else {
            rc = allow_lds(ctx, fast_count_kernel<4>, lds); if (rc) return rc;
            CPM_LAUNCH(ctx, fast_count_kernel<4>, cgrid, dim3(1024), lds, s, photons8, n, G, L, hist, acc, rank);
        }
        CPM_LAUNCH_CHECK(ctx, "fast_count_kernel");
    }
    // n == 0 still runs one workgroup: the table (all starts 0, no items) is part of the result
    const int stiles = n > 0 ? div_up(n, kScatterTile) : 1, smax = 2 * ctx->num_cus;
    const dim3 sgrid((unsigned)(stiles < smax ? stiles : smax));
    if (G.channels == 1) {
        rc = allow_lds(ctx, fast_scatter_kernel<1>, lds); if (rc) return rc;
        CPM_LAUNCH(ctx, fast_scatter_kernel<1>, sgrid, dim3(1024), lds, s, photons8, n, G, L, rank, hist, hist_next, (int)hist_words, brick_table, sorted_pos_power);
    } else {
        rc = allow_lds(ctx, fast_scatter_kernel<4>, lds); if (rc) return rc;
        CPM_LAUNCH(ctx, fast_scatter_kernel<4>, sgrid, dim3(1024), lds, s, photons8, n, G, L, rank, hist, hist_next, (int)hist_words, brick_table, sorted_pos_power);
    }
    CPM_LAUNCH_CHECK(ctx, "fast_scatter_kernel");
    ctx->fast_hist_parity ^= 1;
    ctx->fast_hist_words = hist_words;
    return CPM_OK;
}

int cpm_gather_fast(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n, const cpm_grid_desc* grid,
                    float radius, float scale, int accumulate, float* grid_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev_fast(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n >= 0 && radius > 0.f, "cpm_gather_fast: bad size or radius");
    CPM_REQUIRE(ctx, brick_table && grid_out && (sorted_pos_power || n == 0), "cpm_gather_fast: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, sorted_pos_power, "cpm_gather_fast");
    if (G.channels == 4) CPM_REQUIRE_ALIGNED16(ctx, grid_out, "cpm_gather_fast");
    BrickLayout L;
    brick_shape(grid->dims, L);
    if (!brick_reach(G, radius, L))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_gather_fast", "radius beyond 2 voxels (or beyond half a brick): use cpm_bin + cpm_gather");
    const int ch3 = G.channels == 4 ? 3 : 1;
    const size_t tile_bytes = (size_t)ch3 * (size_t)L.tile * 8;
    if (tile_bytes > 160 * 1024 - 1024)
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_gather_fast", "tile does not fit the LDS: use cpm_bin + cpm_gather");
    const size_t max_items = max_items_for(L, n);
    long long* slabs = (long long*)scratch(ctx, CPM_SCR_FAST_SLABS, max_items * tile_bytes);
    if (!slabs) return CPM_ERR_OUT_OF_MEMORY;
    const float k = kInv4Pi * scale;
    hipStream_t s = (hipStream_t)stream;
    if (n > 0) {
        // resident workgroups: two of 1024 threads per CU, fewer when the worst case has fewer items
        const size_t resident = (size_t)CPM_TILE_WG_PER_CU * (size_t)ctx->num_cus;
        const dim3 tgrid((unsigned)(max_items < resident ? max_items : resident));
#define CPM_TILE_LAUNCH(MAXC, CH)                                                                                        \
    do {                                                                                                                 \
        rc = allow_lds(ctx, fast_tile_kernel<MAXC, CH>, tile_bytes);                                                     \
        if (rc) return rc;                                                                                               \
        CPM_LAUNCH(ctx, (fast_tile_kernel<MAXC, CH>), tgrid, dim3(kTileThreads), tile_bytes, s, sorted_pos_power, brick_table, G, L, radius, \
                   k, slabs);                                                                                    \
    } while (0)
        if (G.channels == 1) {
            if (L.maxc <= 2) CPM_TILE_LAUNCH(2, 1); else if (L.maxc == 3) CPM_TILE_LAUNCH(3, 1); else CPM_TILE_LAUNCH(4, 1);
        } else {
            if (L.maxc <= 2) CPM_TILE_LAUNCH(2, 4); else if (L.maxc == 3) CPM_TILE_LAUNCH(3, 4); else CPM_TILE_LAUNCH(4, 4);
        }
#undef CPM_TILE_LAUNCH
        CPM_LAUNCH_CHECK(ctx, "fast_tile_kernel");
    }
    if (G.channels == 1)
        CPM_LAUNCH(ctx, fast_combine_kernel<1>, dim3((unsigned)L.nb), dim3(kCombineThreads), 0, s, slabs, brick_table, G, L, k, accumulate, grid_out);
    else
        CPM_LAUNCH(ctx, fast_combine_kernel<4>, dim3((unsigned)L.nb), dim3(kCombineThreads), 0, s, slabs, brick_table, G, L, k, accumulate, grid_out);
    CPM_LAUNCH_CHECK(ctx, "fast_combine_kernel");
    return CPM_OK;
}

}  // extern "C"
