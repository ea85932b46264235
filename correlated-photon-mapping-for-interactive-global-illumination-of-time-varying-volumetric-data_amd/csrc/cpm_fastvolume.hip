// cpm_fastvolume.hip -- photons -> light volume, the MI355X formulation in TOLERANCE MODE
// (cpm_bin_fast + cpm_gather_fast; include/cpm/cpm.h "fast formulation").
//
// What it computes: for every voxel the sum over photons of  power * k * 0.75 * (1 - d^2 / r^2)  for d <= r --
// the terms splatPhoton adds (ref progressivephotonmapping/cl/photonstolightvolume.cl:31-79, kernel
// cl/densityestimationkernel.cl:43-60) with the weight taken from d^2 directly (no sqrt, no division: the
// reference's own sum order is undefined -- CAS float atomics, :15-29 -- so only a tolerance is defined anyway).
// The bit-exact per-voxel sequential contract stays in cpm_bin / cpm_gather (cpm_lightvolume.hip).
//
// How (three short launches):
//   bin    fast_count_kernel    photon -> the bricks (8x8x16 voxels; bigger for grids beyond 8 Ki bricks) its candidate
//                               voxels lie in: its own brick, and a neighbour's when the candidate box straddles a
//                               face (0.3 % of the photons at config 2, where r is half a cell; at most 8 bricks).
//                               Per tile of 4096 photons: histogram in LDS, ONE returning global atomic per non-empty
//                               brick = the offset of the tile's run inside that brick (run_base[tile][brick]).
//          fast_scatter_kernel  every workgroup scans the finished histogram into brick starts (LDS, while its photon
//                               loads are in flight); position of a copy = start + run offset + an LDS counter; a
//                               compact 16-byte (pos, power) record per (photon, brick) copy.  Workgroup 0 also writes
//                               the table: brick starts, the list of non-empty bricks, max |power|, the radius.
//   gather fast_brick_kernel    a brick's voxels receive from exactly the records filed under that brick, so a
//                               workgroup sums a brick on its own: lanes own records, the brick is an LDS tile of
//                               64-bit FIXED-POINT sums (ds_add_u64) -- integer addition is associative, so the result
//                               does not depend on the order lanes, waves or runs add in: bitwise reproducible with no
//                               ordering protocol at all -- then ONE rounding to float and a coalesced store of the
//                               brick (zeros for a brick nothing reaches).  No halos, no slabs, no second launch:
//                               what the records' duplication buys -- for boxes of up to 3 candidates per axis.
//          fast_halo_kernel +   boxes of 4 - 8 candidates along some axis (the workspace's 6 x 6 x 2): duplication stops paying (1.7
//          fast_halo_merge_...  to 2.2 copies per photon, each as dear to a wave as a whole box), so a photon is filed once, under the
//                               brick of its box's low corner; that brick's tile has a halo on its high sides, tiles go to a
//                               staging slot per brick as 64-bit sums, and a second launch adds the <= 8 tiles that cover a
//                               voxel -- still integers, still one rounding, the same bits.
// No global float atomics, no sort passes, no per-voxel ordering.
#include "cpm_ctx.h"

using namespace cpm;

namespace {

// tuning experiments build variants with -DCPM_... (tools/build_variant.sh); the defaults are the measured best
#ifndef CPM_BRICK_THREADS
#define CPM_BRICK_THREADS 1024
#endif
#ifndef CPM_BRICK_WG_PER_CU
#define CPM_BRICK_WG_PER_CU 2
#endif
#ifndef CPM_COUNT_ITEMS
#define CPM_COUNT_ITEMS 4
#endif
#ifndef CPM_BRICK_PER
#define CPM_BRICK_PER 2
#endif
constexpr int kBrickThreads = CPM_BRICK_THREADS;
constexpr int kBrickPer = CPM_BRICK_PER;      // records per lane and batch of fast_brick_kernel
constexpr int kCountItems = CPM_COUNT_ITEMS;  // photons per thread of fast_count_kernel (1024 threads)
constexpr int kCountTile = 1024 * kCountItems;
constexpr int kScatterItems = kCountItems;    // the two launches share the tile decomposition (run_base rows)
constexpr int kScatterTile = 1024 * kScatterItems;
constexpr int kMaxBricks = 8192;              // two LDS words per brick in fast_scatter_kernel: 64 KiB

// table layout (u32 entries): [0, nb] brick starts (table[nb] = records written) | 4 meta | nb: the non-empty bricks in
// brick order (the gather's work items: handed to the workgroups round-robin, so that the bricks of a lit face -- every
// 16th brick of a 128^3 grid -- do not all land on the same few workgroups)
constexpr int kMetaMaxPow = 0, kMetaRadius = 1, kMetaItems = 2;
// accumulators behind every scratch histogram (zero between calls, like the histogram): max |power| bits
constexpr int kAccMaxPow = 0, kAccWords = 4;

struct BrickLayout {
    int lx, ly, lz;         // log2 brick size (voxels)
    int nbx, nby, nbz, nb;  // bricks
    int bvox;               // voxels per brick
    int maxc;               // candidate voxels per axis, the widest axis' (0 until brick_reach)
    int mcx, mcy, mcz;      // ... and per axis: an anisotropic grid (the workspace's 256 x 256 x 48 light volume: r = 2.8 / 2.8 / 0.5
                            // voxels) has a box of 6 x 6 x 2 candidates
    int halo;               // boxes of 4 or more candidates along some axis: a photon is filed under ONE brick, the one of its box's low corner, and that brick's
                            // workgroup sums into a tile with mc - 1 more voxels on the high side of every axis (see fast_halo_kernel)
};
CPM_DEV uint32_t off_meta(const BrickLayout& L) { return (uint32_t)L.nb + 1u; }
// a tile's list of (brick, run offset) pairs in run_base: a head (the count) + at most one pair per brick
__host__ __device__ inline size_t pair_stride(const BrickLayout& L) { return (size_t)L.nb + 1u; }
CPM_DEV uint32_t off_items(const BrickLayout& L) { return (uint32_t)L.nb + 5u; }

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

// candidate voxels per axis for a radius (in texture units): floor(2 r') + 1 with r' = r * textureToIndex + 1e-3
__host__ void candidates_per_axis(const GridDev& G, float radius, int mc[3]) {
    const float r[3] = { radius * G.t2i.sx, radius * G.t2i.sy, radius * G.t2i.sz };  // radius in voxels per axis
    for (int a = 0; a < 3; ++a) {
        const float c = floorf(2.f * (r[a] + 1e-3f)) + 1.f;
        mc[a] = c < 1.f ? 1 : (c > 1e6f ? 1000000 : (int)c);
    }
}

// brick shape: 8 x 8 x 16 voxels; while there are more than 8 Ki bricks, doubled along the axis with the most bricks (ties: x,
// then y, then z).
// (Measured at config 2, 128^3: 4096 bricks of 8^3 -> 80.1 us per frame, 2048 of 16x8x8 -> 76.1, 2048 of 8x8x16 -> 74.9,
// 1024 of 16x16x8 -> 81.0, 8192 of 8x4x8 -> 93.6: fewer bricks make the scans, the per-tile rows and the empty bricks
// cheaper until the heaviest brick -- a serial chain of 1024-record batches in one workgroup -- takes over.)
// A WIDE candidate box (more than 4 candidates along some axis, `mc` given) shapes the brick after itself: per axis 16 voxels where the
// box has 5 or more candidates, 8 otherwise, and never more than 2048 voxels (a box wide along all three axes: 16 x 16 x 8).  Measured at
// the workspace's 256 x 256 x 48 light volume (box 6 x 6 x 2) when a photon was still filed under every brick its box touches: 8 x 8 x 16
// 0.233 ms per frame (2.6 copies per photon), 16 x 16 x 8 0.193 (1.7 copies), 16 x 16 x 4 0.197, 16 x 8 x 8 0.208, 32 x 16 x 4 0.228; and with
// one filing per photon and tiles with a halo (fast_halo_kernel): 16 x 16 x 8 0.154, 16 x 16 x 4 0.157, 8 x 16 x 8 0.160, 16 x 8 x 8 0.164,
// 8 x 8 x 8 0.171 (smaller bricks: more halo to stage and merge, and the gather launch is no faster -- it is not bound by the balance
// of its work items).
__host__ void brick_shape(const int dims[3], BrickLayout& L, const int* mc = nullptr) {
#ifndef CPM_BRICK_LG
#define CPM_BRICK_LG 3, 3, 4
#endif
    int lg[3] = { CPM_BRICK_LG };
    if (mc && (mc[0] > 4 || mc[1] > 4 || mc[2] > 4)) {
#ifdef CPM_WIDE_LG
        const int wl[3] = { CPM_WIDE_LG };
        for (int a = 0; a < 3; ++a) lg[a] = wl[a];
#else
        for (int a = 0; a < 3; ++a) lg[a] = mc[a] >= 5 ? 4 : 3;
        if (lg[0] + lg[1] + lg[2] > 11) lg[2] = 3;
        // a record costs what its box has candidates -- of an axis' floor(2 r') + 1 all but one lie within reach of a photon --: a brick's
        // voxels x that count stay under 56 Ki, the brick halved along its longest axis (x, then y, then z), not below 8 voxels -- or a few
        // heavy bricks are the whole launch (128^3, 1 M photons under the lit face, box 6 x 6 x 6: 16 x 16 x 8 bricks 161 us, 8 x 8 x 8 70;
        // the workspace's 6 x 6 x 2 keeps 16 x 16 x 8: smaller bricks there only add halo to stage and merge)
        long long box = 1;
        for (int a = 0; a < 3; ++a) box *= mc[a] > 8 ? 7 : (mc[a] > 1 ? mc[a] - 1 : 1);
        while ((1ll << (lg[0] + lg[1] + lg[2])) * box > 56 * 1024) {
            int axis = 0;
            if (lg[1] > lg[axis]) axis = 1;
            if (lg[2] > lg[axis]) axis = 2;
            if (lg[axis] <= 3) break;
            --lg[axis];
        }
#endif
    }
    auto count = [&](int a) { return (dims[a] + (1 << lg[a]) - 1) >> lg[a]; };
    while ((long long)count(0) * count(1) * count(2) > kMaxBricks) {
        int axis = 0;
        if (count(1) > count(axis)) axis = 1;
        if (count(2) > count(axis)) axis = 2;
        ++lg[axis];
    }
    L.lx = lg[0]; L.ly = lg[1]; L.lz = lg[2];
    L.nbx = count(0); L.nby = count(1); L.nbz = count(2);
    L.nb = L.nbx * L.nby * L.nbz;
    L.bvox = 1 << (lg[0] + lg[1] + lg[2]);
    L.maxc = L.mcx = L.mcy = L.mcz = 0;
    L.halo = 0;
}
// ... for a grid and a radius (what every entry point that knows the radius uses: count, scatter and gather agree by construction)
__host__ void brick_shape_for(const GridDev& G, float radius, BrickLayout& L) {
    const int dims[3] = { G.dx, G.dy, G.dz };
    int mc[3];
    candidates_per_axis(G, radius, mc);
    brick_shape(dims, L, mc);
}

// candidates per axis from the radius; false when the kernels do not cover it: a candidate box must not span more than
// two bricks per axis (it is at most as wide as a brick, or the brick spans the axis).  Up to 3 candidates per axis a photon is
// filed under every brick its box touches and the record loops are unrolled (fast_brick_kernel<2 / 3>); from 4 it is filed once and
// summed into a tile with a halo, with run-time loops over y and z (fast_halo_kernel<., 4 / 6 / 8>).
constexpr int kMaxCandidates = 8;
__host__ bool brick_reach(const GridDev& G, float radius, BrickLayout& L) {
    int mc[3];
    candidates_per_axis(G, radius, mc);
    L.mcx = mc[0]; L.mcy = mc[1]; L.mcz = mc[2];
    L.maxc = mc[0] > mc[1] ? (mc[0] > mc[2] ? mc[0] : mc[2]) : (mc[1] > mc[2] ? mc[1] : mc[2]);
#ifndef CPM_HALO_FROM
#define CPM_HALO_FROM 3
#endif
    L.halo = L.maxc > CPM_HALO_FROM ? 1 : 0;
    if (!(radius > 0.f) || L.maxc > kMaxCandidates) return false;
    const int lg[3] = { L.lx, L.ly, L.lz }, nbr[3] = { L.nbx, L.nby, L.nbz };
    for (int a = 0; a < 3; ++a)
        if (mc[a] > (1 << lg[a]) && nbr[a] > 1) return false;
    return true;
}
// a tile with a halo: voxels per x row -- the brick's, the halo's, and one more where that makes the count even (the merge reads
// voxels in pairs, 16 bytes; the odd column stays zero)
__host__ __device__ inline int halo_pitch_x(const BrickLayout& L) { return ((1 << L.lx) + L.mcx) & ~1; }
// voxels of a brick's LDS tile: the brick, with its halo where photons are filed once
__host__ __device__ inline int tile_voxels(const BrickLayout& L) {
    return L.halo ? halo_pitch_x(L) * ((1 << L.ly) + L.mcy - 1) * ((1 << L.lz) + L.mcz - 1) : L.bvox;
}
__host__ size_t tile_bytes_for(const GridDev& G, const BrickLayout& L) { return (size_t)(G.channels == 4 ? 3 : 1) * (size_t)tile_voxels(L) * 8; }

__host__ size_t table_entries(const BrickLayout& L) { return 2 * (size_t)L.nb + 5; }

CPM_DEV bool is_sentinel(float4 a) { return a.x == kFltMax || a.y == kFltMax || a.z == kFltMax; }

// The candidate voxels of a photon along one axis: the integers within r' = r * textureToIndex + 1e-3 of its index-space
// coordinate, clipped to the grid; empty (s > e) when none.  The clamps act on floats, so that far-away photons never
// reach an out-of-range float -> int conversion.
// At most `maxc` of them -- the host's floor(2 r') + 1, from which the record capacity and the gather's loops are sized: at
// index coordinates of a hundred or more the rounding of u - r' / u + r' can admit one more integer when 2 r' lies within an
// ulp(u) below an integer; that candidate sits at distance r' > r (weight zero) and is dropped here, so that the count, the
// scatter and the gather agree on a photon's bricks by construction.
CPM_DEV void axis_range(float u, float rg, int dim, int maxc, int& s, int& e) {
    s = (int)min_(max_(__builtin_ceilf(u - rg), 0.0f), (float)dim);
    e = (int)max_(min_(__builtin_floorf(u + rg), (float)(dim - 1)), -1.0f);
    e = min(e, s + maxc - 1);
}
struct Box { int sx, ex, sy, ey, sz, ez; };
CPM_DEV bool candidate_box(const GridDev& G, float4 a, float rgx, float rgy, float rgz, int mcx, int mcy, int mcz, Box& b) {
    const f3 p = { a.x, a.y, a.z };
    const f3 u = transform_(G.t2i, p);
    axis_range(u.x, rgx, G.dx, mcx, b.sx, b.ex);
    axis_range(u.y, rgy, G.dy, mcy, b.sy, b.ey);
    axis_range(u.z, rgz, G.dz, mcz, b.sz, b.ez);
    return b.sx <= b.ex && b.sy <= b.ey && b.sz <= b.ez;
}
// f(key) for every brick the box touches (at most two per axis: a box is at most 4 voxels wide, a brick at least 8
// unless it spans its whole axis)
template <typename F>
CPM_DEV void for_each_brick(const BrickLayout& L, const Box& b, F f) {
    const int bx0 = b.sx >> L.lx, by0 = b.sy >> L.ly, bz0 = b.sz >> L.lz;
    if (L.halo) {  // filed once, under the brick of the box's low corner
        f((uint32_t)bx0 + (uint32_t)L.nbx * ((uint32_t)by0 + (uint32_t)L.nby * (uint32_t)bz0));
        return;
    }
    const int bx1 = b.ex >> L.lx, by1 = b.ey >> L.ly, bz1 = b.ez >> L.lz;
    for (int bz = bz0; bz <= bz1; ++bz)
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx)
                f((uint32_t)bx + (uint32_t)L.nbx * ((uint32_t)by + (uint32_t)L.nby * (uint32_t)bz));
}

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

// Exclusive scan of the brick counts (-> brick starts, left in LDS for the caller's scatter) by one 1024-thread
// workgroup; with `table` (workgroup 0 only) the starts and the total also go to the table.  `counts` is the finished
// global histogram (every workgroup reads it: nb * 4 bytes of L2 traffic each).
CPM_DEV void fast_scan(const uint32_t* __restrict__ counts, const BrickLayout& L, uint32_t* __restrict__ s_start,
                       uint32_t* __restrict__ table, uint32_t* s_c, uint32_t* s_i) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (L.nb + 1023) / 1024;
    const int b0 = min(t * per, L.nb), b1 = min(b0 + per, L.nb);
    uint32_t c = 0, it = 0;
    for (int b = b0; b < b1; ++b) { const uint32_t h = counts[b]; s_start[b] = h; c += h; it += h != 0u; }
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
    for (int b = b0; b < b1; ++b) {
        const uint32_t h = s_start[b];
        s_start[b] = ac;
        if (table) { table[b] = ac; if (h) table[off_items(L) + ai++] = (uint32_t)b; }
        ac += h;
    }
    if (table && t == 0) { table[L.nb] = tc; table[off_meta(L) + kMetaItems] = ti; }
}

// bin, launch 1 of 2.  Per workgroup (1024 threads) = per TILE of 4096 photons: the bricks every photon's candidate box
// touches, counted in an LDS histogram; ONE returning global atomic per non-empty brick reserves the tile's run inside that
// brick -- (brick, offset) goes to the tile's list in run_base (as many pairs as the tile has bricks: a tile's 4 lattice rows
// reach 100 - 200 of config 4's 8192 bricks; a dense row per tile made the scatter launch read 34 MB of them there), the
// list's length to its head, for the scatter launch; max |power| of the stored photons on the way.
template <int CH>
__global__ __launch_bounds__(1024) void fast_count_kernel(const float* __restrict__ photons, uint32_t rec_stride, uint32_t rec_b, int n, GridDev G,
                                                          BrickLayout L, float radius, uint32_t* __restrict__ hist, uint32_t* __restrict__ acc,
                                                          uint32_t* __restrict__ run_base) {
    extern __shared__ uint32_t s_hist[];
    __shared__ float s_mp[16];
    __shared__ uint32_t s_pairs;
    const int t = threadIdx.x;
    for (int b = t; b < L.nb; b += 1024) s_hist[b] = 0u;
    if (t == 0) s_pairs = 0u;
    const float rgx = radius * G.t2i.sx + 1e-3f, rgy = radius * G.t2i.sy + 1e-3f, rgz = radius * G.t2i.sz + 1e-3f;
    const float4* __restrict__ ph = reinterpret_cast<const float4*>(photons);
    float mp = 0.f;
    float4 a[kCountItems], a2[kCountItems];
#pragma unroll
    for (int k = 0; k < kCountItems; ++k) {  // the loads first, all in flight together
        const long long i = (long long)blockIdx.x * kCountTile + k * 1024 + t;
        a[k] = make_float4(kFltMax, kFltMax, kFltMax, 0.f); a2[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) { a[k] = ph[(size_t)rec_stride * i]; if (CH == 4) a2[k] = ph[(size_t)rec_stride * i + rec_b]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCountItems; ++k) {
        if (is_sentinel(a[k])) continue;
        // (finite powers only, photon by photon: one NaN / inf must not hide its neighbours' maximum)
        const float p0 = __builtin_fabsf(a[k].w);
        if (p0 <= kFltMax) mp = max_(mp, p0);
        if (CH == 4) {
            const float p1 = __builtin_fabsf(a2[k].x), p2 = __builtin_fabsf(a2[k].y);
            if (p1 <= kFltMax) mp = max_(mp, p1);
            if (p2 <= kFltMax) mp = max_(mp, p2);
        }
        Box box;
        if (candidate_box(G, a[k], rgx, rgy, rgz, L.mcx, L.mcy, L.mcz, box)) for_each_brick(L, box, [&](uint32_t key) { atomicAdd(&s_hist[key], 1u); });
    }
    // max |power| of the workgroup (a finite, non-negative float orders like its bit pattern)
    for (int off = 32; off > 0; off >>= 1) mp = max_(mp, __shfl_xor(mp, off, 64));
    if ((t & 63) == 0) s_mp[t >> 6] = mp;
    __syncthreads();
    // the runs' places: four bins at a time, the atomics of a group issued together
    uint2* __restrict__ pairs = reinterpret_cast<uint2*>(run_base) + (size_t)blockIdx.x * (size_t)pair_stride(L);
    for (int b = t; b < L.nb; b += 4 * 1024) {
        uint32_t c[4], base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int bb = b + q * 1024; c[q] = bb < L.nb ? s_hist[bb] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { base[q] = 0u; if (c[q]) base[q] = atomicAdd(&hist[b + q * 1024], c[q]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (c[q]) pairs[1u + atomicAdd(&s_pairs, 1u)] = make_uint2((uint32_t)(b + q * 1024), base[q]);
    }
    __syncthreads();
    if (t == 0) pairs[0] = make_uint2(s_pairs, 0u);
    // one atomic per workgroup at most, and none once the running maximum has reached this workgroup's
    if (t == 0) {
        float m = s_mp[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) m = max_(m, s_mp[w]);
        const uint32_t mb = __float_as_uint(m);
        if (m > 0.f && __hip_atomic_load(&acc[kAccMaxPow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mb) atomicMax(&acc[kAccMaxPow], mb);
    }
}

// bin, launch 2 of 2.  Every workgroup (1024 threads) first turns the finished histogram into the brick starts -- an
// exclusive scan of nb <= 8 Ki counts in LDS, while its first photon loads are in flight (the one-workgroup scan launch
// this replaces took 5.6 us of pure latency) -- then, per tile of 4096 photons: next free position of the tile's run in
// every brick = brick start + run_base[tile][brick] (a coalesced row read; entries of bricks the tile does not touch are
// never used), and one LDS atomic per (photon, brick) copy hands out the positions: an unstable counting sort (the order
// inside a brick is irrelevant to integer sums).  Workgroup 0 also writes the table.  The histogram exists twice and is
// used in turn: this launch zeroes the OTHER one (idle until the next call), so no memset is needed in steady state and
// no workgroup has to know when the others have read.  Big inputs: a workgroup walks several tiles, the scan is paid once.
template <int CH>
__global__ __launch_bounds__(1024, 8) void fast_scatter_kernel(const float* __restrict__ photons, uint32_t rec_stride, uint32_t rec_b, int n, GridDev G, BrickLayout L, float radius,
                                                            const uint32_t* __restrict__ hist, const uint32_t* __restrict__ run_base,
                                                            uint32_t* __restrict__ zero_next, int zero_words,
                                                            uint32_t* __restrict__ table, float* __restrict__ sorted) {
    extern __shared__ uint32_t s_lds[];
    uint32_t* s_start = s_lds;         // nb: brick starts
    uint32_t* s_pos = s_lds + L.nb;    // nb: next free position of this tile's run in the brick
    __shared__ uint32_t s_c[16], s_i[16];
    const int t = threadIdx.x;
    const float rgx = radius * G.t2i.sx + 1e-3f, rgy = radius * G.t2i.sy + 1e-3f, rgz = radius * G.t2i.sz + 1e-3f;
    const float4* __restrict__ ph = reinterpret_cast<const float4*>(photons);
    const int n_tiles = (int)(((long long)n + kScatterTile - 1) / kScatterTile);
    float4 a[kScatterItems], b2[kScatterItems];
    // a tile's photons and its list of (brick, run offset) pairs (the first few per lane requested with the photons)
    constexpr int kPairsAhead = 2;
    uint2 pr[kPairsAhead];
    uint32_t n_pairs = 0;
    auto load = [&](int tile) {
#pragma unroll
        for (int k = 0; k < kScatterItems; ++k) {
            const long long i = (long long)tile * kScatterTile + k * 1024 + t;
            a[k] = make_float4(kFltMax, kFltMax, kFltMax, 0.f); b2[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) { a[k] = ph[(size_t)rec_stride * i]; if (CH == 4) b2[k] = ph[(size_t)rec_stride * i + rec_b]; }
        }
        n_pairs = 0;
        if (tile < n_tiles) {
            const uint2* __restrict__ pairs = reinterpret_cast<const uint2*>(run_base) + (size_t)tile * (size_t)pair_stride(L);
            n_pairs = pairs[0].x;
#pragma unroll
            for (int q = 0; q < kPairsAhead; ++q) { const uint32_t j = (uint32_t)(q * 1024 + t); pr[q] = j < n_pairs ? pairs[1u + j] : make_uint2(0u, 0u); }
        }
    };
    // next free position of the tile's run in every brick it has copies in = brick start + run offset (s_pos of other bricks
    // is never read)
    auto place = [&](int tile) {
        const uint2* __restrict__ pairs = reinterpret_cast<const uint2*>(run_base) + (size_t)tile * (size_t)pair_stride(L);
#pragma unroll
        for (int q = 0; q < kPairsAhead; ++q) { const uint32_t j = (uint32_t)(q * 1024 + t); if (j < n_pairs) s_pos[pr[q].x] = s_start[pr[q].x] + pr[q].y; }
        for (uint32_t j = (uint32_t)(kPairsAhead * 1024 + t); j < n_pairs; j += 1024u) { const uint2 p = pairs[1u + j]; s_pos[p.x] = s_start[p.x] + p.y; }
    };
    load(blockIdx.x);
    for (int w = blockIdx.x * 1024 + t; w < zero_words; w += gridDim.x * 1024) zero_next[w] = 0u;
    if (blockIdx.x == 0 && t == 0) {
        table[off_meta(L) + kMetaMaxPow] = hist[L.nb + kAccMaxPow];
        table[off_meta(L) + kMetaRadius] = __float_as_uint(radius);
    }
    fast_scan(hist, L, s_start, blockIdx.x == 0 ? table : nullptr, s_c, s_i);
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (tile != (int)blockIdx.x) load(tile);
        place(tile);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScatterItems; ++k) {
            Box box;
            if (is_sentinel(a[k]) || !candidate_box(G, a[k], rgx, rgy, rgz, L.mcx, L.mcy, L.mcz, box)) continue;
            for_each_brick(L, box, [&](uint32_t key) {
                const size_t pos = (size_t)atomicAdd(&s_pos[key], 1u);
                if (CH == 1) {
                    // scattered 16-byte records, read next by another launch: streaming stores
                    typedef float v4 __attribute__((ext_vector_type(4)));
                    const v4 q = { a[k].x, a[k].y, a[k].z, a[k].w };
                    __builtin_nontemporal_store(q, reinterpret_cast<v4*>(sorted) + pos);
                } else {
                    float4* o = reinterpret_cast<float4*>(sorted) + 2 * pos;
                    o[0] = a[k];
                    o[1] = make_float4(b2[k].x, b2[k].y, 0.f, 0.f);
                }
            });
        }
        __syncthreads();  // s_pos is rewritten for the next tile
    }
}

// One record into the brick's LDS tile.  Candidates per axis: the integers within r' of the photon's index-space
// coordinate, clipped to the grid (axis_range) and to THIS brick -- the same record sits in the neighbouring brick's list
// for the candidates beyond the face; d^2 in texture space with the contract's operands (c = indexToTexture * v,
// d = c - p, d^2 = fma(dz, dz, fma(dy, dy, dx * dx))); weight 0.75 * (1 - d^2 / r^2) for d^2 <= r^2; value -> fixed point
// by truncation.
// The same record for a WIDE box (more than 4 candidates along some axis; fast_halo_kernel): the whole box -- the photon is filed under the
// brick of the box's low corner only, and the tile reaches mc - 1 voxels beyond the brick's high faces -- with run-time loops over z and y
// and the x candidates unrolled.  TX / TY: the tile's row and slice pitch.
template <int CH, int WX /* candidates along x the loop is unrolled for: >= L.mcx */>
CPM_DEV void halo_record(const GridDev& G, const BrickLayout& L, float4 a, float pg, float pb, int ox, int oy, int oz, int TX, int TY,
                         float rgx, float rgy, float rgz, float r2, float inv_r2, float k, float S, long long* __restrict__ tile, int plane) {
    Box c;
    if (!candidate_box(G, a, rgx, rgy, rgz, L.mcx, L.mcy, L.mcz, c)) return;
    // value * S with S a power of two: (p k w) S and (p k S) w round the same real number scaled by 2^sh -- the same bits (a product
    // small enough to be denormal truncates to 0 either way: S <= 2^100)
    const float pk = __builtin_fabsf(a.w) <= kFltMax ? (a.w * k) * S : 0.f;
    const float pkg = __builtin_fabsf(pg) <= kFltMax ? (pg * k) * S : 0.f, pkb = __builtin_fabsf(pb) <= kFltMax ? (pb * k) * S : 0.f;
    float dx2[WX];
#pragma unroll
    for (int q = 0; q < WX; ++q) { const float d = fma_(G.i2t.sx, (float)(c.sx + q), G.i2t.tx) - a.x; dx2[q] = d * d; }
    for (int vz = c.sz; vz <= c.ez; ++vz) {
        const float dz = fma_(G.i2t.sz, (float)vz, G.i2t.tz) - a.z;
        for (int vy = c.sy; vy <= c.ey; ++vy) {
            const float dy = fma_(G.i2t.sy, (float)vy, G.i2t.ty) - a.y;
            // no slot of this row can hit: d^2 >= fma(dz, dz, dy * dy) for every dx (rounding is monotone in the addend) -- the far slice
            // of a photon midway between two slices, the outer rows of one well off its slice
            if (!(fma_(dz, dz, dy * dy) <= r2)) continue;
            const int row = (c.sx - ox) + TX * ((vy - oy) + TY * (vz - oz));
#pragma unroll
            for (int q = 0; q < WX; ++q) {
                const float d2 = fma_(dz, dz, fma_(dy, dy, dx2[q]));   // the contract's operands: dx * dx, then the two fmas
                if (c.sx + q > c.ex || !(d2 <= r2)) continue;
                const float w = 0.75f * (1.0f - d2 * inv_r2);
                atomicAdd(reinterpret_cast<unsigned long long*>(tile + row + q), (unsigned long long)(long long)(int)(pk * w));
                if (CH == 4) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(tile + plane + row + q), (unsigned long long)(long long)(int)(pkg * w));
                    atomicAdd(reinterpret_cast<unsigned long long*>(tile + 2 * plane + row + q), (unsigned long long)(long long)(int)(pkb * w));
                }
            }
        }
    }
}

template <int MAXC, int CH>
CPM_DEV void brick_record(const GridDev& G, float4 a, float pg, float pb, int ox, int oy, int oz, int BX, int BY, int BZ, float rgx,
                          float rgy, float rgz, float r2, float inv_r2, float k, float S, long long* __restrict__ tile, int plane) {
    Box c;
    if (!candidate_box(G, a, rgx, rgy, rgz, MAXC, MAXC, MAXC, c)) return;
    const int sx = max(c.sx, ox), ex = min(c.ex, ox + BX - 1);
    const int sy = max(c.sy, oy), ey = min(c.ey, oy + BY - 1);
    const int sz = max(c.sz, oz), ez = min(c.ez, oz + BZ - 1);
    // (a non-finite power has no fixed-point image: that channel of that photon contributes nothing, as in max |power|)
    const float pk = __builtin_fabsf(a.w) <= kFltMax ? a.w * k : 0.f;
    const float pkg = __builtin_fabsf(pg) <= kFltMax ? pg * k : 0.f, pkb = __builtin_fabsf(pb) <= kFltMax ? pb * k : 0.f;
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
    const int base = (sx - ox) + BX * ((sy - oy) + BY * (sz - oz));
#pragma unroll
    for (int qz = 0; qz < MAXC; ++qz)
#pragma unroll
        for (int qy = 0; qy < MAXC; ++qy)
#pragma unroll
            for (int qx = 0; qx < MAXC; ++qx) {
                const float d2 = fma_(dzv[qz], dzv[qz], fma_(dyv[qy], dyv[qy], dxv[qx] * dxv[qx]));
                // the value is formed for every candidate (7 instructions); only the LDS add is conditional
                const float w = 0.75f * (1.0f - d2 * inv_r2);
                const int idx = base + qx + BX * (qy + BY * qz);
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

// gather: ONE launch.  A fixed grid of resident workgroups (two per CU) strides over the bricks: an empty brick gets its
// zeros (or is left alone in accumulate mode), a brick with records is summed in LDS -- all of its records, 1024 at a
// time, the next batch requested while the current one is added -- rounded once and stored.
#ifndef CPM_BRICK_WAVES
#define CPM_BRICK_WAVES 8
#endif
constexpr int kMaxSubBricks = 512;  // 4x4x4 sub-bricks of a brick (a tile of <= 159 KiB holds <= 320)
// SEG (cpm_gather_fast_segment): nothing is stored to a grid; a gather brick's non-zero 4x4x4 sub-bricks go to slots of a brick-list segment
// (cpm_ctx.h) taken from its device counter -- one atomic per gather brick --, a wave writing one sub-brick's 64 values (256 contiguous
// bytes per channel plane of the slot); no zeros, no marks; the launch's last workgroup writes the segment's header and mailbox word.
template <int MAXC, int CH, bool SEG>
__global__ __launch_bounds__(kBrickThreads, (CH == 4 ? 4 : CPM_BRICK_WAVES)) void fast_brick_kernel(const float* __restrict__ sorted, const uint32_t* __restrict__ table, GridDev G,
                                                                  BrickLayout L, float radius, float k, int accumulate, float* __restrict__ out,
                                                                  uint8_t* __restrict__ marks, SegTarget st) {
    extern __shared__ long long s_tile[];
    __shared__ uint32_t s_slot[SEG ? kMaxSubBricks : 1];   // SEG: a sub-brick's rank among the brick's non-zero ones (0xffffffff: all zero)
    __shared__ uint32_t s_base;
    // marks (nullable): one byte per 4x4x4-voxel brick of the grid (cpm_mark_touched_bricks' numbering), 1 where this launch
    // leaves a non-zero value, 0 elsewhere -- every byte written: what cpm_allreduce_grid_sparse would otherwise read the whole
    // volume again for
    __shared__ uint8_t s_flag[kMaxSubBricks];
    const int nbx4 = (G.dx + 3) >> 2, nby4 = (G.dy + 3) >> 2;
    constexpr int CH3 = CH == 4 ? 3 : 1, STRIDE = CH == 4 ? 2 : 1;
    const int t = threadIdx.x;
    const int BX = 1 << L.lx, BY = 1 << L.ly, BZ = 1 << L.lz;
    const int words = CH3 * L.bvox;
    const float rgx = radius * G.t2i.sx + 1e-3f, rgy = radius * G.t2i.sy + 1e-3f, rgz = radius * G.t2i.sz + 1e-3f;
    const float r2 = radius * radius, inv_r2 = 1.0f / r2;
    const float S = fixed_scale(__uint_as_float(table[off_meta(L) + kMetaMaxPow]), k);
    const float invS = 1.0f / S;  // a power of two: exact
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
    // the zeros of the bricks nothing reaches (not in accumulate mode): one streaming pass over the volume, every lane
    // looks its voxels' brick up itself (two cached loads) -- no per-brick chain of dependent table reads in a workgroup
    if (!SEG && !accumulate) {
        const size_t cells = (size_t)G.dx * G.dy * G.dz;
        if (CH == 1 && (G.dx & 3) == 0) {  // 4 voxels of one x-row (and one brick: bricks are >= 8 wide) per lane and turn
            for (size_t i = ((size_t)blockIdx.x * kBrickThreads + t) * 4; i < cells; i += (size_t)gridDim.x * kBrickThreads * 4) {
                const int x = (int)(i % (size_t)G.dx), y = (int)((i / (size_t)G.dx) % (size_t)G.dy), z = (int)(i / ((size_t)G.dx * G.dy));
                const uint32_t b = (uint32_t)(x >> L.lx) + (uint32_t)L.nbx * ((uint32_t)(y >> L.ly) + (uint32_t)L.nby * (uint32_t)(z >> L.lz));
                if (table[b + 1] == table[b]) {
                    *reinterpret_cast<float4*>(out + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (marks && ((y | z) & 3) == 0) marks[(uint32_t)(x >> 2) + (uint32_t)nbx4 * ((uint32_t)(y >> 2) + (uint32_t)nby4 * (uint32_t)(z >> 2))] = 0;
                }
            }
        } else {
            for (size_t i = (size_t)blockIdx.x * kBrickThreads + t; i < cells; i += (size_t)gridDim.x * kBrickThreads) {
                const int x = (int)(i % (size_t)G.dx), y = (int)((i / (size_t)G.dx) % (size_t)G.dy), z = (int)(i / ((size_t)G.dx * G.dy));
                const uint32_t b = (uint32_t)(x >> L.lx) + (uint32_t)L.nbx * ((uint32_t)(y >> L.ly) + (uint32_t)L.nby * (uint32_t)(z >> L.lz));
                if (table[b + 1] != table[b]) continue;
                if (CH == 1) out[i] = 0.f; else reinterpret_cast<float4*>(out)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (marks && ((x | y | z) & 3) == 0) marks[(uint32_t)(x >> 2) + (uint32_t)nbx4 * ((uint32_t)(y >> 2) + (uint32_t)nby4 * (uint32_t)(z >> 2))] = 0;
            }
        }
    }
    // the bricks with records: the table's list, round-robin over the workgroups
    const uint32_t n_items = table[off_meta(L) + kMetaItems];
    const uint32_t* __restrict__ items = table + off_items(L);
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t b = items[item];
        const uint32_t j0 = table[b], j1 = table[b + 1];
        const int bx = (int)(b % (uint32_t)L.nbx), by = (int)((b / (uint32_t)L.nbx) % (uint32_t)L.nby), bz = (int)(b / (uint32_t)(L.nbx * L.nby));
        const int ox = bx << L.lx, oy = by << L.ly, oz = bz << L.lz;
        // batches of kBrickPer x 1024 records, kBrickPer independent loads per lane; the first batch is requested before the
        // tile is cleared, every next one while the current one is added
        float4 a[kBrickPer], a2[kBrickPer], an[kBrickPer], an2[kBrickPer];
        auto fetch = [&](uint32_t first, float4* x, float4* x2) {
#pragma unroll
            for (int q = 0; q < kBrickPer; ++q) {
                const uint32_t j = first + (uint32_t)(q * kBrickThreads + t);
                x[q] = make_float4(0.f, 0.f, 0.f, 0.f); x2[q] = x[q];
                if (j < j1) { x[q] = rec[STRIDE * (size_t)j]; if (CH == 4) x2[q] = rec[2 * (size_t)j + 1]; }
            }
        };
        fetch(j0, a, a2);
        for (int w = t; w < words; w += kBrickThreads) s_tile[w] = 0ll;
        long long* my_tile = s_tile;
        const int nsub = L.bvox >> 6;
        if (marks || SEG) for (int w = t; w < nsub; w += kBrickThreads) s_flag[w] = 0;  // (the lane that read flag w for the brick before)
        __syncthreads();
        for (uint32_t first = j0; first < j1; first += (uint32_t)(kBrickPer * kBrickThreads)) {  // uniform
            fetch(first + (uint32_t)(kBrickPer * kBrickThreads), an, an2);
#pragma unroll
            for (int q = 0; q < kBrickPer; ++q) {
                const uint32_t j = first + (uint32_t)(q * kBrickThreads + t);
                if (j < j1) {
                    // a box of at most MAXC^3 candidates (MAXC <= 4), all loops unrolled
                    brick_record<MAXC, CH>(G, a[q], a2[q].x, a2[q].y, ox, oy, oz, BX, BY, BZ, rgx, rgy, rgz, r2, inv_r2, k, S, my_tile, L.bvox);
                }
            }
#pragma unroll
            for (int q = 0; q < kBrickPer; ++q) { a[q] = an[q]; a2[q] = an2[q]; }
        }
        __syncthreads();
        if (SEG) {
            // which sub-bricks hold a non-zero value (what the dense launch would store: the rounded sums)
            for (int v = t; v < L.bvox; v += kBrickThreads) {
                const int lx = v & (BX - 1), ly = (v >> L.lx) & (BY - 1), lz = v >> (L.lx + L.ly);
                if (ox + lx >= G.dx || oy + ly >= G.dy || oz + lz >= G.dz) continue;
                bool nonzero = (float)s_tile[v] * invS != 0.f;
                if (CH == 4) nonzero = nonzero || (float)s_tile[L.bvox + v] * invS != 0.f || (float)s_tile[2 * L.bvox + v] * invS != 0.f;
                if (nonzero) s_flag[(lx >> 2) + (BX >> 2) * ((ly >> 2) + (BY >> 2) * (lz >> 2))] = 1;
            }
            __syncthreads();
            // their slots: ranks inside the brick by ballot (wave 0), one atomic on the segment's counter for the brick
            if (t < 64) {
                uint32_t running = 0;
                for (int w0 = 0; w0 < nsub; w0 += 64) {
                    const int w = w0 + t;
                    const bool f = w < nsub && s_flag[w] != 0;
                    const unsigned long long m = __ballot(f);
                    if (w < nsub) s_slot[w] = f ? running + (uint32_t)__popcll(m & ((1ull << t) - 1ull)) : 0xffffffffu;
                    running += (uint32_t)__popcll(m);
                }
                if (t == 0) s_base = running ? atomicAdd(&st.ctl[0], running) : 0u;
            }
            __syncthreads();
            const size_t slot_bytes = seg_slot_bytes(CH);
            for (int idx = t; idx < nsub * 64; idx += kBrickThreads) {
                const int w = idx >> 6, j = idx & 63;
                const uint32_t rel = s_slot[w];
                if (rel == 0xffffffffu) continue;   // (uniform per wave: a wave writes one sub-brick)
                const uint32_t slot = s_base + rel;
                if (slot >= st.room) continue;
                const int sx = w % (BX >> 2), sy = (w / (BX >> 2)) % (BY >> 2), sz = w / ((BX >> 2) * (BY >> 2));
                const int lx = 4 * sx + (j & 3), ly = 4 * sy + ((j >> 2) & 3), lz = 4 * sz + (j >> 4);
                const int v = lx + BX * (ly + BY * lz);
                const bool inside = ox + lx < G.dx && oy + ly < G.dy && oz + lz < G.dz;
                unsigned char* p = st.seg + sizeof(SegHeader) + (size_t)slot * slot_bytes;
                if (j == 0) {
                    const uint32_t id = (uint32_t)((ox >> 2) + sx) + (uint32_t)nbx4 * ((uint32_t)((oy >> 2) + sy) + (uint32_t)nby4 * (uint32_t)((oz >> 2) + sz));
                    *reinterpret_cast<uint4*>(p) = make_uint4(id, 0u, 0u, 0u);
                }
                const float fr = inside ? (float)s_tile[v] * invS : 0.f;
                if (CH == 1) {
                    reinterpret_cast<float*>(p + 16)[j] = fr;
                } else {
                    const float fg = inside ? (float)s_tile[L.bvox + v] * invS : 0.f, fb = inside ? (float)s_tile[2 * L.bvox + v] * invS : 0.f;
                    reinterpret_cast<float4*>(p + 16)[j] = make_float4(fr, fg, fb, 0.f);
                }
            }
            __syncthreads();  // the tile is cleared again for the next brick
            continue;
        }
        for (int v = t; v < L.bvox; v += kBrickThreads) {
            const int lx = v & (BX - 1), ly = (v >> L.lx) & (BY - 1), lz = v >> (L.lx + L.ly);
            const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
            if (gx >= G.dx || gy >= G.dy || gz >= G.dz) continue;
            const size_t o = (size_t)gx + (size_t)G.dx * ((size_t)gy + (size_t)G.dy * (size_t)gz);
            const long long sum_r = s_tile[v];
            const float fr = (float)sum_r * invS;
            bool nonzero = fr != 0.f;
            if (CH == 1) {
                out[o] = accumulate ? out[o] + fr : fr;
            } else {
                const long long sum_g = s_tile[L.bvox + v], sum_b = s_tile[2 * L.bvox + v];
                const float fg = (float)sum_g * invS, fb = (float)sum_b * invS;
                nonzero = nonzero || fg != 0.f || fb != 0.f;
                float4* q = reinterpret_cast<float4*>(out) + o;
                if (accumulate) { const float4 tt = *q; *q = make_float4(tt.x + fr, tt.y + fg, tt.z + fb, tt.w); }
                else *q = make_float4(fr, fg, fb, 0.f);
            }
            if (marks && nonzero) s_flag[(lx >> 2) + (BX >> 2) * ((ly >> 2) + (BY >> 2) * (lz >> 2))] = 1;  // (same value from every writer)
        }
        __syncthreads();  // the tile is cleared again for the next brick
        if (marks)
            for (int w = t; w < nsub; w += kBrickThreads) {
                const int sx = w % (BX >> 2), sy = (w / (BX >> 2)) % (BY >> 2), sz = w / ((BX >> 2) * (BY >> 2));
                const int gx4 = (ox >> 2) + sx, gy4 = (oy >> 2) + sy, gz4 = (oz >> 2) + sz;
                if (4 * gx4 < G.dx && 4 * gy4 < G.dy && 4 * gz4 < G.dz) marks[(uint32_t)gx4 + (uint32_t)nbx4 * ((uint32_t)gy4 + (uint32_t)nby4 * (uint32_t)gz4)] = s_flag[w];
            }
    }
    if (SEG && t == 0) seg_finish(st, gridDim.x);
}

// ---- boxes of 4 or more candidates along some axis: tiles with a halo, no copies ------------------------------------------------------
// (from which width: config 2's photons, 128^3, bin + gather in us with copies | filed once: 3 candidates (r = 1.45 voxels) 53.3 | 54.9,
// 4 (r = 1.9) 86.1 | 71.3, 2 (r = 0.87) 36.4 | 47.4 -- tools/fast_radius_sweep.py)
// With a box 6 wide, 43 % of the records of a 16 x 16 x 8 brick were copies of neighbouring bricks' photons (1.7 copies per photon), and a
// copy costs a wave as much as a whole box: lanes hold unrelated records, the wave runs every row and slot any lane needs
// (docs/EXPERIMENTS.md, round 5).  So a wide-box photon is filed ONCE, under the brick of its box's low corner; that brick's workgroup sums
// whole boxes into a tile that reaches mc - 1 voxels beyond the brick's high faces and leaves the tile's 64-bit sums in `stage` (one
// slot per brick, written by the bricks with records only).  fast_halo_merge_kernel then forms every voxel's sum over the <= 8 tiles
// that cover it -- its own brick's and the halos of the bricks below / left / in front -- as integers, rounds once and stores: the same
// bits as one sum over all photons.  It also writes the zeros of the bricks no tile covers and the non-zero marks.
template <int CH, int WX>
__global__ __launch_bounds__(kBrickThreads, (CH == 4 ? 4 : CPM_BRICK_WAVES)) void fast_halo_kernel(const float* __restrict__ sorted, const uint32_t* __restrict__ table,
                                                                                                    GridDev G, BrickLayout L, float radius, float k,
                                                                                                    long long* __restrict__ stage, int repl) {
    extern __shared__ long long s_tile[];
    constexpr int CH3 = CH == 4 ? 3 : 1, STRIDE = CH == 4 ? 2 : 1;
    const int t = threadIdx.x;
    const int TX = halo_pitch_x(L), TY = (1 << L.ly) + L.mcy - 1;
    const int tvox = tile_voxels(L), words = CH3 * tvox;
    const float rgx = radius * G.t2i.sx + 1e-3f, rgy = radius * G.t2i.sy + 1e-3f, rgz = radius * G.t2i.sz + 1e-3f;
    const float r2 = radius * radius, inv_r2 = 1.0f / r2;
    const float S = fixed_scale(__uint_as_float(table[off_meta(L) + kMetaMaxPow]), k);
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
    const uint32_t n_items = table[off_meta(L) + kMetaItems];
    const uint32_t* __restrict__ items = table + off_items(L);
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t b = items[item];
        const uint32_t j0 = table[b], j1 = table[b + 1];
        const int bx = (int)(b % (uint32_t)L.nbx), by = (int)((b / (uint32_t)L.nbx) % (uint32_t)L.nby), bz = (int)(b / (uint32_t)(L.nbx * L.nby));
        const int ox = bx << L.lx, oy = by << L.ly, oz = bz << L.lz;
        float4 a[kBrickPer], a2[kBrickPer], an[kBrickPer], an2[kBrickPer];
        auto fetch = [&](uint32_t first, float4* x, float4* x2) {
#pragma unroll
            for (int q = 0; q < kBrickPer; ++q) {
                const uint32_t j = first + (uint32_t)(q * kBrickThreads + t);
                x[q] = make_float4(0.f, 0.f, 0.f, 0.f); x2[q] = x[q];
                if (j < j1) { x[q] = rec[STRIDE * (size_t)j]; if (CH == 4) x2[q] = rec[2 * (size_t)j + 1]; }
            }
        };
        fetch(j0, a, a2);
        // `repl` copies of the tile, lane l adds into copy l mod repl: neighbouring lanes hold neighbouring photons, whose adds meet in
        // the same voxels -- same-address LDS atomics of one instruction are served one after the other
        for (int w = t; w < words * repl; w += kBrickThreads) s_tile[w] = 0ll;
        long long* my_tile = s_tile + (size_t)(t & (repl - 1)) * (size_t)words;
        __syncthreads();
        for (uint32_t first = j0; first < j1; first += (uint32_t)(kBrickPer * kBrickThreads)) {  // uniform
            fetch(first + (uint32_t)(kBrickPer * kBrickThreads), an, an2);
#pragma unroll
            for (int q = 0; q < kBrickPer; ++q) {
                const uint32_t j = first + (uint32_t)(q * kBrickThreads + t);
                if (j < j1) halo_record<CH, WX>(G, L, a[q], a2[q].x, a2[q].y, ox, oy, oz, TX, TY, rgx, rgy, rgz, r2, inv_r2, k, S, my_tile, tvox);
            }
#pragma unroll
            for (int q = 0; q < kBrickPer; ++q) { a[q] = an[q]; a2[q] = an2[q]; }
        }
        __syncthreads();
        long long* __restrict__ dst = stage + (size_t)b * (size_t)words;
        for (int w = t; w < words; w += kBrickThreads) {
            long long sum = s_tile[w];
            for (int c = 1; c < repl; ++c) sum += s_tile[(size_t)c * words + w];
            dst[w] = sum;
        }
        __syncthreads();  // the tile is cleared again for the next brick
    }
}

// One workgroup per brick of the grid (see fast_halo_kernel).
// SEG (cpm_gather_fast_segment): a brick's sums go to an LDS copy instead of the grid, its non-zero 4x4x4 sub-bricks from there to slots of a
// brick-list segment (as fast_brick_kernel<., ., true>); bricks no tile covers write nothing.
constexpr int kSegMergeVoxels = 2048;  // the bricks of wide boxes (brick_shape: at most 2^11 voxels)
template <int CH, bool SEG>
__global__ __launch_bounds__(256) void fast_halo_merge_kernel(const long long* __restrict__ stage, const uint32_t* __restrict__ table, GridDev G, BrickLayout L,
                                                              float k, int accumulate, float* __restrict__ out, uint8_t* __restrict__ marks, SegTarget st) {
    __shared__ uint8_t s_flag[kMaxSubBricks];
    __shared__ float s_val[SEG ? kSegMergeVoxels * (CH == 4 ? 3 : 1) : 1];
    __shared__ uint32_t s_slot[SEG ? kSegMergeVoxels / 64 : 1];
    __shared__ uint32_t s_base;
    constexpr int CH3 = CH == 4 ? 3 : 1;
    const int t = threadIdx.x;
    const int BX = 1 << L.lx, BY = 1 << L.ly, BZ = 1 << L.lz;
    const int hx = L.mcx - 1, hy = L.mcy - 1, hz = L.mcz - 1;
    const int TX = halo_pitch_x(L), TY = BY + hy;
    const int tvox = tile_voxels(L), words = CH3 * tvox;
    const int nbx4 = (G.dx + 3) >> 2, nby4 = (G.dy + 3) >> 2, nsub = L.bvox >> 6;
    const float S = fixed_scale(__uint_as_float(table[off_meta(L) + kMetaMaxPow]), k);
    const float invS = 1.0f / S;  // a power of two: exact
    for (uint32_t b = blockIdx.x; b < (uint32_t)L.nb; b += gridDim.x) {
        const int bx = (int)(b % (uint32_t)L.nbx), by = (int)((b / (uint32_t)L.nbx) % (uint32_t)L.nby), bz = (int)(b / (uint32_t)(L.nbx * L.nby));
        const int ox = bx << L.lx, oy = by << L.ly, oz = bz << L.lz;
        // the tiles that cover this brick: its own and those of the 7 bricks one step down along x / y / z (uniform; the 16 table words are
        // requested together -- a brick that is not there stands in as itself)
        const long long* src[8];
        uint32_t lo[8], hi[8], sbv[8];
        int n_src = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int dx = d & 1, dy = (d >> 1) & 1, dz = d >> 2;
            const bool there = !((dx && (bx == 0 || hx == 0)) || (dy && (by == 0 || hy == 0)) || (dz && (bz == 0 || hz == 0)));
            sbv[d] = there ? (uint32_t)(bx - dx) + (uint32_t)L.nbx * ((uint32_t)(by - dy) + (uint32_t)L.nby * (uint32_t)(bz - dz)) : 0xffffffffu;
            const uint32_t sb = there ? sbv[d] : b;
            lo[d] = table[sb]; hi[d] = table[sb + 1];
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const bool lit = sbv[d] != 0xffffffffu && lo[d] != hi[d];
            src[d] = lit ? stage + (size_t)sbv[d] * (size_t)words : nullptr;
            n_src += lit ? 1 : 0;
        }
        if (marks || SEG) for (int w = t; w < nsub; w += 256) s_flag[w] = 0;
        __syncthreads();
        if (n_src == 0) {
            // no tile covers this brick: its zeros (nothing in accumulate mode or for a segment), four voxels of an x row per lane where rows allow it
            if (!SEG && !accumulate) {
                if (CH == 1 && (G.dx & 3) == 0) {
                    for (int v = 4 * t; v < L.bvox; v += 4 * 256) {
                        const int lx = v & (BX - 1), ly = (v >> L.lx) & (BY - 1), lz = v >> (L.lx + L.ly);
                        const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
                        if (gx < G.dx && gy < G.dy && gz < G.dz)
                            *reinterpret_cast<float4*>(out + ((size_t)gx + (size_t)G.dx * ((size_t)gy + (size_t)G.dy * (size_t)gz))) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
                    for (int v = t; v < L.bvox; v += 256) {
                        const int lx = v & (BX - 1), ly = (v >> L.lx) & (BY - 1), lz = v >> (L.lx + L.ly);
                        const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
                        if (gx >= G.dx || gy >= G.dy || gz >= G.dz) continue;
                        const size_t o = (size_t)gx + (size_t)G.dx * ((size_t)gy + (size_t)G.dy * (size_t)gz);
                        if (CH == 1) out[o] = 0.f; else reinterpret_cast<float4*>(out)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        } else {
            // a lane takes PAIRS of voxels along x (16-byte loads: the tile's rows hold an even number of voxels, a pair never straddles the
            // brick's edge or a tile's), kMergePer pairs per turn; every load of a turn is requested before the first is used: one per
            // covering tile that is there (uniform) and reaches the pair's first voxel; the second voxel's value is dropped where that
            // tile does not reach it
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            constexpr int kMergePer = CH == 4 ? 1 : 2;
            const int pairs = L.bvox >> 1, hbx = BX >> 1;
            for (int w0 = t; w0 < pairs; w0 += 256 * kMergePer) {
                ll2 val[kMergePer][8][CH3];
                uint32_t second[kMergePer];   // bit d: tile d reaches the pair's second voxel too
                size_t o[kMergePer];
                bool in[kMergePer];
                int sub[kMergePer], gxs[kMergePer];
#pragma unroll
                for (int u = 0; u < kMergePer; ++u) {
                    const int w = w0 + 256 * u;
                    const int lx = 2 * (w & (hbx - 1)), ly = (w >> (L.lx - 1)) & (BY - 1), lz = w >> (L.lx - 1 + L.ly);
                    const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
                    in[u] = w < pairs && gx < G.dx && gy < G.dy && gz < G.dz;
                    gxs[u] = gx;
                    o[u] = (size_t)gx + (size_t)G.dx * ((size_t)gy + (size_t)G.dy * (size_t)gz);
                    sub[u] = (lx >> 2) + (BX >> 2) * ((ly >> 2) + (BY >> 2) * (lz >> 2));
                    second[u] = 0u;
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const int dx = d & 1, dy = (d >> 1) & 1, dz = d >> 2;
#pragma unroll
                        for (int c = 0; c < CH3; ++c) val[u][d][c] = ll2{ 0ll, 0ll };
                        const bool reach = src[d] && in[u] && !((dx && lx >= hx) || (dy && ly >= hy) || (dz && lz >= hz));
                        if (reach) {
                            // (the voxel's place in a tile one brick down along an axis: that many voxels further up)
                            const int idx = (lx + TX * (ly + TY * lz)) + (dx * BX + TX * (dy * BY + TY * (dz * BZ)));
                            second[u] |= (!dx || lx + 1 < hx) ? (1u << d) : 0u;
#pragma unroll
                            for (int c = 0; c < CH3; ++c) val[u][d][c] = *reinterpret_cast<const ll2*>(src[d] + c * tvox + idx);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kMergePer; ++u) {
                    if (!in[u]) continue;
                    long long sum0[CH3], sum1[CH3];
#pragma unroll
                    for (int c = 0; c < CH3; ++c) { sum0[c] = 0ll; sum1[c] = 0ll; }
#pragma unroll
                    for (int d = 0; d < 8; ++d)
#pragma unroll
                        for (int c = 0; c < CH3; ++c) { sum0[c] += val[u][d][c].x; sum1[c] += ((second[u] >> d) & 1u) ? val[u][d][c].y : 0ll; }
                    const bool two = gxs[u] + 1 < G.dx;
                    const float f0 = (float)sum0[0] * invS, f1 = (float)sum1[0] * invS;
                    bool nonzero = f0 != 0.f || (two && f1 != 0.f);
                    if (SEG) {
                        const int w = w0 + 256 * u;
                        const int v0 = 2 * (w & (hbx - 1)) + BX * (((w >> (L.lx - 1)) & (BY - 1)) + BY * (w >> (L.lx - 1 + L.ly)));
                        s_val[v0] = f0; s_val[v0 + 1] = two ? f1 : 0.f;
                        if (CH == 4) {
                            const float g0 = (float)sum0[CH == 4 ? 1 : 0] * invS, b0 = (float)sum0[CH == 4 ? 2 : 0] * invS;
                            const float g1 = (float)sum1[CH == 4 ? 1 : 0] * invS, b1 = (float)sum1[CH == 4 ? 2 : 0] * invS;
                            nonzero = nonzero || g0 != 0.f || b0 != 0.f || (two && (g1 != 0.f || b1 != 0.f));
                            s_val[L.bvox + v0] = g0; s_val[L.bvox + v0 + 1] = two ? g1 : 0.f;
                            s_val[2 * L.bvox + v0] = b0; s_val[2 * L.bvox + v0 + 1] = two ? b1 : 0.f;
                        }
                        if (nonzero) s_flag[sub[u]] = 1;
                        continue;
                    }
                    if (CH == 1) {
                        if (two && (o[u] & 1) == 0) {
                            float2* q = reinterpret_cast<float2*>(out + o[u]);
                            if (accumulate) { const float2 tt = *q; *q = make_float2(tt.x + f0, tt.y + f1); } else *q = make_float2(f0, f1);
                        } else {
                            out[o[u]] = accumulate ? out[o[u]] + f0 : f0;
                            if (two) out[o[u] + 1] = accumulate ? out[o[u] + 1] + f1 : f1;
                        }
                    } else {
                        const float g0 = (float)sum0[CH == 4 ? 1 : 0] * invS, b0 = (float)sum0[CH == 4 ? 2 : 0] * invS;
                        const float g1 = (float)sum1[CH == 4 ? 1 : 0] * invS, b1 = (float)sum1[CH == 4 ? 2 : 0] * invS;
                        nonzero = nonzero || g0 != 0.f || b0 != 0.f || (two && (g1 != 0.f || b1 != 0.f));
                        float4* q = reinterpret_cast<float4*>(out) + o[u];
                        if (accumulate) {
                            const float4 tt = q[0]; q[0] = make_float4(tt.x + f0, tt.y + g0, tt.z + b0, tt.w);
                            if (two) { const float4 t1 = q[1]; q[1] = make_float4(t1.x + f1, t1.y + g1, t1.z + b1, t1.w); }
                        } else {
                            q[0] = make_float4(f0, g0, b0, 0.f);
                            if (two) q[1] = make_float4(f1, g1, b1, 0.f);
                        }
                    }
                    if (marks && nonzero) s_flag[sub[u]] = 1;  // (same value from every writer; a pair lies in one 4 x 4 x 4 sub-brick)
                }
            }
        }
        __syncthreads();
        if (SEG && n_src != 0) {   // (uniform)
            if (t < 64) {
                uint32_t running = 0;
                for (int w0 = 0; w0 < nsub; w0 += 64) {
                    const int w = w0 + t;
                    const bool f = w < nsub && s_flag[w] != 0;
                    const unsigned long long m = __ballot(f);
                    if (w < nsub) s_slot[w] = f ? running + (uint32_t)__popcll(m & ((1ull << t) - 1ull)) : 0xffffffffu;
                    running += (uint32_t)__popcll(m);
                }
                if (t == 0) s_base = running ? atomicAdd(&st.ctl[0], running) : 0u;
            }
            __syncthreads();
            const size_t slot_bytes = seg_slot_bytes(CH);
            for (int idx = t; idx < nsub * 64; idx += 256) {
                const int w = idx >> 6, j = idx & 63;
                const uint32_t rel = s_slot[w];
                if (rel == 0xffffffffu) continue;
                const uint32_t slot = s_base + rel;
                if (slot >= st.room) continue;
                const int sx = w % (BX >> 2), sy = (w / (BX >> 2)) % (BY >> 2), sz = w / ((BX >> 2) * (BY >> 2));
                const int lx = 4 * sx + (j & 3), ly = 4 * sy + ((j >> 2) & 3), lz = 4 * sz + (j >> 4);
                const int v = lx + BX * (ly + BY * lz);
                const bool inside = ox + lx < G.dx && oy + ly < G.dy && oz + lz < G.dz;   // (voxels beyond the grid were never written to s_val)
                unsigned char* p = st.seg + sizeof(SegHeader) + (size_t)slot * slot_bytes;
                if (j == 0) {
                    const uint32_t id = (uint32_t)((ox >> 2) + sx) + (uint32_t)nbx4 * ((uint32_t)((oy >> 2) + sy) + (uint32_t)nby4 * (uint32_t)((oz >> 2) + sz));
                    *reinterpret_cast<uint4*>(p) = make_uint4(id, 0u, 0u, 0u);
                }
                if (CH == 1) reinterpret_cast<float*>(p + 16)[j] = inside ? s_val[v] : 0.f;
                else reinterpret_cast<float4*>(p + 16)[j] = inside ? make_float4(s_val[v], s_val[L.bvox + v], s_val[2 * L.bvox + v], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (marks)
            for (int w = t; w < nsub; w += 256) {
                const int sx = w % (BX >> 2), sy = (w / (BX >> 2)) % (BY >> 2), sz = w / ((BX >> 2) * (BY >> 2));
                const int gx4 = (ox >> 2) + sx, gy4 = (oy >> 2) + sy, gz4 = (oz >> 2) + sz;
                if (4 * gx4 < G.dx && 4 * gy4 < G.dy && 4 * gz4 < G.dz) marks[(uint32_t)gx4 + (uint32_t)nbx4 * ((uint32_t)gy4 + (uint32_t)nby4 * (uint32_t)gz4)] = s_flag[w];
            }
        __syncthreads();  // s_flag is cleared again for the next brick
    }
    if (SEG && t == 0) seg_finish(st, gridDim.x);
}

// records from one layout to the other: a lane moves one record (two 16-byte loads, two 16-byte stores; one side of each pair is
// contiguous across the wave whatever the direction)
__global__ __launch_bounds__(256) void photons_convert_kernel(const float4* __restrict__ src, uint32_t ss, uint32_t sb, float4* __restrict__ dst,
                                                              uint32_t ds, uint32_t db, size_t n) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float4 a = src[(size_t)ss * j], b = src[(size_t)ss * j + sb];
    dst[(size_t)ds * j] = a;
    dst[(size_t)ds * j + db] = b;
}

// what a workgroup may take of the LDS: the device's limit less 1 KiB for the kernels' static arrays (gfx950: 160 KiB; queried once per
// context -- a part with 64 KiB refuses the wide boxes here instead of failing at the launch)
__host__ size_t lds_limit(const cpm_ctx* ctx) { return (size_t)(ctx && ctx->lds_per_block > 1024 ? ctx->lds_per_block : 64 * 1024) - 1024; }

template <typename K>
__host__ int allow_lds(cpm_ctx* ctx, K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return CPM_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return set_error(ctx, CPM_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", hipGetErrorString(e));
    return CPM_OK;
}

// records a photon can be filed as: one per brick its box touches (two per axis at most); a wide box is filed once (tiles with a halo)
__host__ int copies_per_photon(const BrickLayout& L) { return L.halo ? 1 : (L.maxc > 1 ? 8 : 1); }

}  // namespace

extern "C" {

size_t cpm_fast_table_entries(const cpm_grid_desc* grid, int n) {
    if (!grid || n < 0 || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1) return 0;
    // (the radius is not known here: room for the brick shape of a narrow box and for those of wide ones)
    size_t most = 0;
    for (int w = 0; w < 512; ++w) {  // every box of 1 ... 8 candidates per axis
        const int mc[3] = { 1 + (w & 7), 1 + ((w >> 3) & 7), 1 + (w >> 6) };
        BrickLayout L;
        brick_shape(grid->dims, L, mc);
        most = table_entries(L) > most ? table_entries(L) : most;
    }
    return most;
}

int cpm_gather_fast_supported(const cpm_grid_desc* grid, float radius) { return cpm_gather_fast_supported_on(nullptr, grid, radius); }

int cpm_gather_fast_supported_on(const cpm_ctx* ctx, const cpm_grid_desc* grid, float radius) {
    if (!grid || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1) return 0;
    if (grid->channels != 1 && grid->channels != 4) return 0;
    GridDev G;
    G.dx = grid->dims[0]; G.dy = grid->dims[1]; G.dz = grid->dims[2]; G.channels = grid->channels;
    if (!affine_from_matrix(grid->texture_to_index, G.t2i) || !affine_from_matrix(grid->index_to_texture, G.i2t)) return 0;
    if (!(G.t2i.sx > 0.f && G.t2i.sy > 0.f && G.t2i.sz > 0.f)) return 0;
    BrickLayout L;
    brick_shape_for(G, radius, L);
    if (!brick_reach(G, radius, L)) return 0;
    return tile_bytes_for(G, L) <= (ctx ? lds_limit(ctx) : (size_t)160 * 1024 - 1024);   // (no context: gfx950's 160 KiB)
}

size_t cpm_fast_record_capacity(const cpm_grid_desc* grid, int n, float radius) {
    if (!cpm_gather_fast_supported(grid, radius) || n < 0) return 0;
    GridDev G;
    G.dx = grid->dims[0]; G.dy = grid->dims[1]; G.dz = grid->dims[2]; G.channels = grid->channels;
    (void)affine_from_matrix(grid->texture_to_index, G.t2i); (void)affine_from_matrix(grid->index_to_texture, G.i2t);
    BrickLayout L;
    brick_shape_for(G, radius, L);
    (void)brick_reach(G, radius, L);
    return (size_t)n * (size_t)copies_per_photon(L);
}

int cpm_bin_fast(cpm_ctx* ctx, const float* photons8, int n, const cpm_grid_desc* grid, float radius, uint32_t* brick_table,
                 float* sorted_pos_power, cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    const int described = described_layout(ctx, photons8, nullptr);   // (cpm_records_describe wins over the context's default)
    return cpm_bin_fast_layout(ctx, photons8, described >= 0 ? described : ctx->photon_layout, n, grid, radius, brick_table, sorted_pos_power, stream);
}

int cpm_bin_fast_layout(cpm_ctx* ctx, const float* photons8, int layout, int n, const cpm_grid_desc* grid, float radius, uint32_t* brick_table,
                        float* sorted_pos_power, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, layout == CPM_PHOTONS_INTERLEAVED || layout == CPM_PHOTONS_PLANAR, "cpm_bin_fast: photon layout");
    // half A of record j at float4 j * rs, half B rb behind it (the tracer's rec_stride / rec_b: n here is its N * I)
    // (a described buffer's planes lie its own N * I apart whatever `n` this call bins: cpm_records_describe)
    size_t described_n = 0;
    const bool described_planar = described_layout(ctx, photons8, &described_n) == CPM_PHOTONS_PLANAR;
    CPM_REQUIRE(ctx, !(layout == CPM_PHOTONS_PLANAR && described_planar) || (size_t)(n > 0 ? n : 0) <= described_n, "cpm_bin_fast: more records than the buffer was described with");
    const uint32_t rs = layout == CPM_PHOTONS_PLANAR ? 1u : 2u,
                   rb = layout == CPM_PHOTONS_PLANAR ? (described_planar ? (uint32_t)described_n : (uint32_t)(n > 0 ? n : 0)) : 1u;
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
    brick_shape_for(G, radius, L);
    if (!brick_reach(G, radius, L) || tile_bytes_for(G, L) > lds_limit(ctx))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_bin_fast", "radius beyond 3.5 voxels along some axis (or not positive): use cpm_bin + cpm_gather");
    CPM_REQUIRE(ctx, (long long)n * copies_per_photon(L) < (1ll << 32), "cpm_bin_fast: record positions are 32-bit (n * 8 must stay below 2^32)");
    // scratch: two histograms (nb brick counts + 4 accumulators), used in turn -- a call's scatter launch zeroes the one the
    // NEXT call counts into -- then run_base: per tile of 4096 photons a list of (brick, run offset) pairs behind its length
    // (room for every brick; as many written and read as the tile touches)
    static_assert(kCountTile == kScatterTile, "count and scatter launches share the tile decomposition");
    const size_t hist_words = (size_t)L.nb + kAccWords;
    const size_t tiles = n > 0 ? (size_t)div_up(n, kCountTile) : 1;
    const size_t arena = (2 * hist_words + 2 * tiles * pair_stride(L)) * 4;  // (a pair = 2 words)
    const bool had = ctx->scratch_bytes[CPM_SCR_FAST_BIN] >= arena;
    uint32_t* base = (uint32_t*)scratch(ctx, CPM_SCR_FAST_BIN, arena);
    if (!base) return CPM_ERR_OUT_OF_MEMORY;
    if (!had || ctx->fast_hist_words != hist_words) {  // new arena or another brick count: the zero state is not established
        CPM_HIP_CHECK(ctx, hipMemsetAsync(base, 0, 2 * hist_words * 4, s));
        ctx->fast_hist_parity = 0;
    }
    uint32_t* hist = base + (size_t)ctx->fast_hist_parity * hist_words;
    uint32_t* zero_next = base + (size_t)(ctx->fast_hist_parity ^ 1) * hist_words;
    uint32_t* acc = hist + L.nb;
    uint32_t* run_base = base + 2 * hist_words;
    ctx->fast_hist_words = 0;  // re-established below once the kernel that restores the zero state is enqueued
    const size_t lds = (size_t)L.nb * 4;
    if (n > 0) {
        const dim3 cgrid((unsigned)tiles);
        if (G.channels == 1) {
            rc = allow_lds(ctx, fast_count_kernel<1>, lds); if (rc) return rc;
            CPM_LAUNCH(ctx, fast_count_kernel<1>, cgrid, dim3(1024), lds, s, photons8, rs, rb, n, G, L, radius, hist, acc, run_base);
        } else {
            rc = allow_lds(ctx, fast_count_kernel<4>, lds); if (rc) return rc;
            CPM_LAUNCH(ctx, fast_count_kernel<4>, cgrid, dim3(1024), lds, s, photons8, rs, rb, n, G, L, radius, hist, acc, run_base);
        }
        CPM_LAUNCH_CHECK(ctx, "fast_count_kernel");
    }
    // n == 0 still runs one workgroup: the table (all starts 0) is part of the result
    const int stiles = n > 0 ? div_up(n, kScatterTile) : 1, smax = 2 * ctx->num_cus;
    const dim3 sgrid((unsigned)(stiles < smax ? stiles : smax));
    if (G.channels == 1) {
        rc = allow_lds(ctx, fast_scatter_kernel<1>, 2 * lds); if (rc) return rc;
        CPM_LAUNCH(ctx, fast_scatter_kernel<1>, sgrid, dim3(1024), 2 * lds, s, photons8, rs, rb, n, G, L, radius, hist, run_base, zero_next, (int)hist_words, brick_table, sorted_pos_power);
    } else {
        rc = allow_lds(ctx, fast_scatter_kernel<4>, 2 * lds); if (rc) return rc;
        CPM_LAUNCH(ctx, fast_scatter_kernel<4>, sgrid, dim3(1024), 2 * lds, s, photons8, rs, rb, n, G, L, radius, hist, run_base, zero_next, (int)hist_words, brick_table, sorted_pos_power);
    }
    CPM_LAUNCH_CHECK(ctx, "fast_scatter_kernel");
    ctx->fast_hist_parity ^= 1;
    ctx->fast_hist_words = hist_words;
    ctx->fast_last_table = brick_table;
    ctx->fast_last_radius = radius;
    return CPM_OK;
}

int cpm_photons_convert(cpm_ctx* ctx, const float* src, int src_layout, float* dst, int dst_layout, size_t n_records, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, (src_layout == CPM_PHOTONS_INTERLEAVED || src_layout == CPM_PHOTONS_PLANAR) &&
                         (dst_layout == CPM_PHOTONS_INTERLEAVED || dst_layout == CPM_PHOTONS_PLANAR), "cpm_photons_convert: layout");
    if (n_records == 0) return CPM_OK;
    CPM_REQUIRE(ctx, src && dst && src != dst, "cpm_photons_convert: null or aliased buffers");
    CPM_REQUIRE(ctx, n_records < (1ull << 31), "cpm_photons_convert: more than 2^31 records");
    CPM_REQUIRE_ALIGNED16(ctx, src, "cpm_photons_convert");
    CPM_REQUIRE_ALIGNED16(ctx, dst, "cpm_photons_convert");
    const uint32_t n = (uint32_t)n_records;
    const uint32_t ss = src_layout == CPM_PHOTONS_PLANAR ? 1u : 2u, sb = src_layout == CPM_PHOTONS_PLANAR ? n : 1u;
    const uint32_t ds = dst_layout == CPM_PHOTONS_PLANAR ? 1u : 2u, db = dst_layout == CPM_PHOTONS_PLANAR ? n : 1u;
    CPM_LAUNCH(ctx, photons_convert_kernel, dim3((unsigned)div_up((long long)n_records, 256)), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const float4*>(src), ss, sb, reinterpret_cast<float4*>(dst), ds, db, n_records);
    CPM_LAUNCH_CHECK(ctx, "photons_convert_kernel");
    return CPM_OK;
}

}  // extern "C"

namespace {

// cpm_gather_fast / _marked (seg == nullptr) and cpm_gather_fast_segment (seg: the brick-list segment the non-zero 4x4x4 bricks go to)
int gather_fast_impl(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n, const cpm_grid_desc* grid, float radius, float scale,
                     int accumulate, float* grid_out, uint8_t* nonzero_bricks, const SegTarget* seg, cpm_stream stream) {
    GridDev G;
    int rc = make_grid_dev_fast(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n >= 0 && radius > 0.f, "cpm_gather_fast: bad size or radius");
    CPM_REQUIRE(ctx, brick_table && (grid_out || seg) && (sorted_pos_power || n == 0), "cpm_gather_fast: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, sorted_pos_power, "cpm_gather_fast");
    // (the wide boxes' merge launch stores pairs and quads of voxels, the 4-channel launches float4s)
    if (!seg && G.channels == 4) CPM_REQUIRE_ALIGNED16(ctx, grid_out, "cpm_gather_fast");
    BrickLayout L;
    brick_shape_for(G, radius, L);
    if (!brick_reach(G, radius, L))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_gather_fast", "radius beyond 3.5 voxels along some axis: use cpm_bin + cpm_gather");
    if (!seg && L.halo) CPM_REQUIRE_ALIGNED16(ctx, grid_out, "cpm_gather_fast (boxes of 4 or more candidates)");
    size_t tile_bytes = tile_bytes_for(G, L);
    if (tile_bytes > lds_limit(ctx) || (L.bvox >> 6) > kMaxSubBricks)
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_gather_fast", "brick does not fit the LDS: use cpm_bin + cpm_gather");
    if (seg && L.halo && L.bvox > kSegMergeVoxels)
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_gather_fast_segment", "bricks of more than 2048 voxels: gather into a grid and use cpm_bricklist_pack_grid");
    // the records were filed for ONE radius (a wider one would need copies the bin did not make)
    if (ctx->fast_last_table == brick_table && ctx->fast_last_radius != radius)
        return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_gather_fast", "radius differs from the one given to cpm_bin_fast for this table");
    const float k = kInv4Pi * scale;
    hipStream_t s = (hipStream_t)stream;
    const SegTarget st = seg ? *seg : SegTarget();
    // resident workgroups: two of 1024 threads per CU, fewer when there are fewer bricks
    const size_t resident = (size_t)CPM_BRICK_WG_PER_CU * (size_t)ctx->num_cus;
    const dim3 bgrid((unsigned)((size_t)L.nb < resident ? (size_t)L.nb : resident));
    if (L.halo) {
        // wide boxes: tiles with a halo into the staging slots, then the merge (see fast_halo_kernel).  Copies of the tile for the same-address
        // adds of neighbouring photons: as many (a power of two, at most CPM_BRICK_REPL) as leave two workgroups per CU their LDS.
#ifndef CPM_BRICK_REPL
#define CPM_BRICK_REPL 4
#endif
        int repl = 1;
        while (repl * 2 <= CPM_BRICK_REPL && tile_bytes * (size_t)(repl * 2) <= lds_limit(ctx) * 9 / 20) repl *= 2;
        const size_t stage_bytes = (size_t)L.nb * tile_bytes;
        long long* stage = (long long*)scratch(ctx, CPM_SCR_FAST_STAGE, stage_bytes);
        if (!stage) return CPM_ERR_OUT_OF_MEMORY;
        const size_t lds = tile_bytes * (size_t)repl;
        const int wide = L.mcx <= 4 ? 4 : (L.mcx <= 6 ? 6 : 8);  // what the x loop is unrolled for
#define CPM_HALO_LAUNCH(CH, WX)                                                                                                          \
    do {                                                                                                                                 \
        rc = allow_lds(ctx, fast_halo_kernel<CH, WX>, lds);                                                                              \
        if (rc) return rc;                                                                                                               \
        CPM_LAUNCH(ctx, (fast_halo_kernel<CH, WX>), bgrid, dim3(kBrickThreads), lds, s, sorted_pos_power, brick_table, G, L, radius, k,   \
                   stage, repl);                                                                                                         \
    } while (0)
        if (G.channels == 1) { if (wide == 4) CPM_HALO_LAUNCH(1, 4); else if (wide == 6) CPM_HALO_LAUNCH(1, 6); else CPM_HALO_LAUNCH(1, 8); }
        else { if (wide == 4) CPM_HALO_LAUNCH(4, 4); else if (wide == 6) CPM_HALO_LAUNCH(4, 6); else CPM_HALO_LAUNCH(4, 8); }
#undef CPM_HALO_LAUNCH
        CPM_LAUNCH_CHECK(ctx, "fast_halo_kernel");
        // (a segment: fewer workgroups, each over several bricks -- every workgroup ends with an atomic on the segment's done counter, and returning
        // atomics on one address are served one after the other)
        const size_t mres = (size_t)4 * (size_t)ctx->num_cus;
        const dim3 mgrid((unsigned)(seg && (size_t)L.nb > mres ? mres : (size_t)L.nb));
        if (seg) {
            if (G.channels == 1) CPM_LAUNCH(ctx, (fast_halo_merge_kernel<1, true>), mgrid, dim3(256), 0, s, stage, brick_table, G, L, k, 0, grid_out, nullptr, st);
            else CPM_LAUNCH(ctx, (fast_halo_merge_kernel<4, true>), mgrid, dim3(256), 0, s, stage, brick_table, G, L, k, 0, grid_out, nullptr, st);
        } else {
            if (G.channels == 1) CPM_LAUNCH(ctx, (fast_halo_merge_kernel<1, false>), mgrid, dim3(256), 0, s, stage, brick_table, G, L, k, accumulate, grid_out, nonzero_bricks, st);
            else CPM_LAUNCH(ctx, (fast_halo_merge_kernel<4, false>), mgrid, dim3(256), 0, s, stage, brick_table, G, L, k, accumulate, grid_out, nonzero_bricks, st);
        }
        CPM_LAUNCH_CHECK(ctx, "fast_halo_merge_kernel");
        return CPM_OK;
    }
#define CPM_BRICK_LAUNCH(MAXC, CH, SEG)                                                                                  \
    do {                                                                                                                 \
        rc = allow_lds(ctx, fast_brick_kernel<MAXC, CH, SEG>, tile_bytes);                                               \
        if (rc) return rc;                                                                                               \
        CPM_LAUNCH(ctx, (fast_brick_kernel<MAXC, CH, SEG>), bgrid, dim3(kBrickThreads), tile_bytes, s, sorted_pos_power, brick_table, G, L,  \
                   radius, k, accumulate, grid_out, nonzero_bricks, st);                                                 \
    } while (0)
    if (seg) {
        if (G.channels == 1) { if (L.maxc <= 2) CPM_BRICK_LAUNCH(2, 1, true); else CPM_BRICK_LAUNCH(3, 1, true); }
        else { if (L.maxc <= 2) CPM_BRICK_LAUNCH(2, 4, true); else CPM_BRICK_LAUNCH(3, 4, true); }
    } else if (G.channels == 1) {
        if (L.maxc <= 2) CPM_BRICK_LAUNCH(2, 1, false); else CPM_BRICK_LAUNCH(3, 1, false);
    } else {
        if (L.maxc <= 2) CPM_BRICK_LAUNCH(2, 4, false); else CPM_BRICK_LAUNCH(3, 4, false);
    }
#undef CPM_BRICK_LAUNCH
    CPM_LAUNCH_CHECK(ctx, "fast_brick_kernel");
    return CPM_OK;
}

}  // namespace

extern "C" {

int cpm_gather_fast(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n, const cpm_grid_desc* grid,
                    float radius, float scale, int accumulate, float* grid_out, cpm_stream stream) {
    return cpm_gather_fast_marked(ctx, sorted_pos_power, brick_table, n, grid, radius, scale, accumulate, grid_out, nullptr, stream);
}

int cpm_gather_fast_marked(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n, const cpm_grid_desc* grid,
                           float radius, float scale, int accumulate, float* grid_out, uint8_t* nonzero_bricks, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, !(nonzero_bricks && accumulate), "cpm_gather_fast_marked: the marks describe a volume this launch wrote whole (not with accumulate)");
    CPM_REQUIRE(ctx, grid_out, "cpm_gather_fast: null buffer");
    return gather_fast_impl(ctx, sorted_pos_power, brick_table, n, grid, radius, scale, accumulate, grid_out, nonzero_bricks, nullptr, stream);
}

int cpm_gather_fast_segment(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n, const cpm_grid_desc* grid,
                            float radius, float scale, const cpm_bricklist_segment* segment, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, segment && segment->segment && segment->control && grid, "cpm_gather_fast_segment: null segment (the root, or a communicator of one rank, gathers into its grid)");
    CPM_REQUIRE(ctx, (int)segment->channels == grid->channels, "cpm_gather_fast_segment: the segment was opened for another channel count");
    const unsigned long long nb4 = (unsigned long long)((grid->dims[0] + 3) / 4) * ((grid->dims[1] + 3) / 4) * ((grid->dims[2] + 3) / 4);
    CPM_REQUIRE(ctx, grid->dims[0] >= 1 && grid->dims[1] >= 1 && grid->dims[2] >= 1 && nb4 <= segment->room, "cpm_gather_fast_segment: the segment has no room for every brick of this grid");
    SegTarget st;
    st.seg = static_cast<unsigned char*>(segment->segment);
    st.capacity = segment->capacity; st.room = segment->room; st.ticket = segment->ticket;
    st.ctl = segment->control; st.mailbox = segment->mailbox;
    return gather_fast_impl(ctx, sorted_pos_power, brick_table, n, grid, radius, scale, 0, nullptr, nullptr, &st, stream);
}

}  // extern "C"
