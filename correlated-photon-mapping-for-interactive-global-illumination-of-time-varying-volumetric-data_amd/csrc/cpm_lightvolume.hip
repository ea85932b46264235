// cpm_lightvolume.hip -- photons -> light volume.
//
//  (1) reference formulation: one work-item per photon, Epanechnikov-weighted float atomic
//      adds (G1, G3, G4; ref progressivephotonmapping/cl/photonstolightvolume.cl:31-79,
//      139-202,225-248).  The reference's CAS loop (:15-29) becomes the native
//      global_atomic_add_f32 of gfx950.
//  (2) MI355X formulation (S6 + G1-G3 restated): cell key per photon, stable radix sort,
//      cell-start table, then a per-voxel gather that visits the photons of the cells in
//      reach in a fixed order and adds exactly the terms the splat would add -- plain
//      stores, no atomics, bitwise reproducible.
#include "cpm_ctx.h"

using namespace cpm;

namespace cpm {
}

namespace {

__host__ int make_grid_dev(cpm_ctx* ctx, const cpm_grid_desc* g, GridDev& G) {
    if (!g) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "null grid desc");
    if (g->dims[0] < 1 || g->dims[1] < 1 || g->dims[2] < 1) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "dims < 1");
    if ((unsigned long long)g->dims[0] * g->dims[1] * g->dims[2] >= (1ull << 31))
        return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "more than 2^31 cells");
    if (g->channels != 1 && g->channels != 4) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "grid", "channels must be 1 or 4");
    G.dx = g->dims[0]; G.dy = g->dims[1]; G.dz = g->dims[2]; G.channels = g->channels;
    if (!affine_from_matrix(g->texture_to_index, G.t2i) || !affine_from_matrix(g->index_to_texture, G.i2t))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "grid", "texture/index matrices must be scale + translate");
    return CPM_OK;
}

// cpm_bin's cell keys (floor(p * dims)) and cpm_gather's halo width (floor(r * dims + 0.5) cells) are written for Inviwo's
// own light-volume matrices, textureToIndex = scale(dims) then translate(-0.5) and its inverse -- what
// cpm_grid_desc_default builds and what PhotonToLightVolumeProcessorCL creates (ref processor/
// photontolightvolumeprocessorcl.cpp:137-170).  Any other (scaled / offset) light volume is refused here rather than
// binned against the wrong cells; cpm_splat and the tolerance-mode pair cpm_bin_fast + cpm_gather_fast read the matrices.
__host__ bool default_matrices(const GridDev& G) {
    const float d[3] = { (float)G.dx, (float)G.dy, (float)G.dz };
    const float ts[3] = { G.t2i.sx, G.t2i.sy, G.t2i.sz }, tt[3] = { G.t2i.tx, G.t2i.ty, G.t2i.tz };
    const float is[3] = { G.i2t.sx, G.i2t.sy, G.i2t.sz }, it[3] = { G.i2t.tx, G.i2t.ty, G.i2t.tz };
    for (int a = 0; a < 3; ++a)
        if (ts[a] != d[a] || tt[a] != -0.5f || is[a] != 1.0f / d[a] || it[a] != 0.5f / d[a]) return false;
    return true;
}
#define CPM_REQUIRE_DEFAULT_MATRICES(ctx, G, who)                                                                       \
    do {                                                                                                                \
        if (!default_matrices(G))                                                                                       \
            return set_error((ctx), CPM_ERR_UNSUPPORTED, who,                                                           \
                             "needs the default light-volume matrices (scale(dims), translate(-0.5)); use cpm_bin_fast + cpm_gather_fast or cpm_splat"); \
    } while (0)

struct Box3 { int sx, sy, sz, ex, ey, ez; };

// photonstolightvolume.cl:45-47: convert_int3 truncates; upper bound exclusive, +1 before truncation
CPM_DEV Box3 splat_box(const GridDev& G, f3 p, float radius) {
    f3 lo = { p.x - radius, p.y - radius, p.z - radius };
    f3 hi = { p.x + radius, p.y + radius, p.z + radius };
    f3 a = transform_(G.t2i, lo);
    f3 b = transform_(G.t2i, hi);
    Box3 r;
    r.sx = max((int)a.x, 0); r.sy = max((int)a.y, 0); r.sz = max((int)a.z, 0);
    r.ex = min((int)(b.x + 1.f), G.dx); r.ey = min((int)(b.y + 1.f), G.dy); r.ez = min((int)(b.z + 1.f), G.dz);
    return r;
}

// photonstolightvolume.cl:57-60
CPM_DEV float splat_weight(f3 c, f3 p, float radius) {
    float dx = c.x - p.x, dy = c.y - p.y, dz = c.z - p.z;
    float dist = __builtin_sqrtf(fma_(dz, dz, fma_(dy, dy, dx * dx)));
    return density_kernel_(dist / radius);
}

// splatPhoton (photonstolightvolume.cl:31-79) with the power already scaled; z_first / z_step: the slices
// sz + z_first, sz + z_first + z_step, ... of the box only (0 / 1 = the whole box; lanes sharing a photon split it by slice),
// y_first / y_step: likewise its rows
CPM_DEV void splat_photon(float* __restrict__ out, const GridDev& G, f3 p, f3 pw, float radius, int z_first = 0, int z_step = 1,
                          int y_first = 0, int y_step = 1) {
    if (p.x == kFltMax || p.y == kFltMax || p.z == kFltMax) return;
    Box3 bx = splat_box(G, p, radius);
    for (int z = bx.sz + z_first; z < bx.ez; z += z_step)
        for (int y = bx.sy + y_first; y < bx.ey; y += y_step)
            for (int x = bx.sx; x < bx.ex; ++x) {
                size_t voxel = (size_t)x + (size_t)y * G.dx + (size_t)z * G.dx * G.dy;
                f3 vi = { (float)x, (float)y, (float)z };
                f3 c = transform_(G.i2t, vi);
                float w = splat_weight(c, p, radius);
                if (G.channels == 1) {
                    float v = pw.x * w;
                    if (v != 0.f) unsafeAtomicAdd(&out[voxel], v);
                } else {
                    float vx = pw.x * w, vy = pw.y * w, vz = pw.z * w;
                    if (vx != 0.f) unsafeAtomicAdd(&out[voxel * 4], vx);
                    if (vy != 0.f) unsafeAtomicAdd(&out[voxel * 4 + 1], vy);
                    if (vz != 0.f) unsafeAtomicAdd(&out[voxel * 4 + 2], vz);
                }
            }
}

// splatPhotonsToLightVolumeKernel (photonstolightvolume.cl:139-166)
__global__ __launch_bounds__(256) void splat_kernel(const float* __restrict__ photons, RecLayout R, int n, GridDev G, float radius,
                                                    float k, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* q = rec_at(photons, R, (size_t)i);
    float4 a = q[0], b = q[R.b];
    f3 p = { a.x, a.y, a.z };
    f3 pw = { a.w * k, b.x * k, b.y * k };
    splat_photon(out, G, p, pw, radius);
}

// splatSelectedPhotonsToLightVolumeKernel (photonstolightvolume.cl:168-202)
__global__ __launch_bounds__(256) void splat_selected_kernel(const float* __restrict__ photons, RecLayout R,
                                                             const uint32_t* __restrict__ indices, int n_indices,
                                                             GridDev G, float radius, float k, float multiplier,
                                                             int n_photons, int n_interactions,
                                                             float* __restrict__ out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_indices) return;
    size_t id = indices[j];
    for (int it = 0; it < n_interactions; ++it) {
        const float4* q = rec_at(photons, R, (size_t)it * n_photons + id);
        float4 a = q[0], b = q[R.b];
        f3 p = { a.x, a.y, a.z };
        f3 pw = { a.w * k, b.x * k, b.y * k };
        pw.x *= multiplier; pw.y *= multiplier; pw.z *= multiplier;
        splat_photon(out, G, p, pw, radius);
    }
}

// copyIndexPhotonsKernel (photonstolightvolume.cl:225-248)
__global__ __launch_bounds__(256) void copy_indexed_kernel(const float* __restrict__ photons, RecLayout R,
                                                           const uint32_t* __restrict__ indices, int n_indices,
                                                           float multiplier, int n_photons, int n_interactions,
                                                           float* __restrict__ aligned, int out_offset) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_indices) return;
    size_t id = indices[j];
    for (int it = 0; it < n_interactions; ++it) {
        const float4* q = rec_at(photons, R, (size_t)it * n_photons + id);
        float4 a = q[0], b = q[R.b];
        float4* o = reinterpret_cast<float4*>(aligned) + 2 * ((size_t)out_offset + j + (size_t)it * n_indices);
        o[0] = make_float4(a.x, a.y, a.z, a.w * multiplier);
        o[1] = make_float4(b.x * multiplier, b.y * multiplier, b.z, b.w);
    }
}

// previous-photon snapshot refresh: only the re-traced photons move (two 16-byte accesses each way)
__global__ __launch_bounds__(256) void snapshot_selected_kernel(const float* __restrict__ photons, RecLayout R, const uint32_t* __restrict__ indices,
                                                                int n_indices, int n_photons, int n_interactions,
                                                                float* __restrict__ snapshot) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_indices * n_interactions) return;
    const uint32_t idx = indices[j % n_indices];
    if (idx >= (uint32_t)n_photons) return;
    const size_t id = (size_t)idx + (size_t)(j / n_indices) * (size_t)n_photons;
    const float4* q = rec_at(photons, R, id);
    float4* o = rec_at(snapshot, R, id);
    const float4 a = q[0], b = q[R.b];
    o[0] = a;
    o[R.b] = b;
}

// ---- bin

// cell key (template: cl/hashlightsample.cl:55-64); sentinels get key == cells so that the
// sort needs only bits(cells) key bits and they still land behind every real cell
// One workgroup = one tile of the radix sort that follows (256 x ITEMS keys): while the keys are in registers the
// tile's histogram of the first digit is counted in LDS and written where the sort's first pass expects it
// (digit-major hist[256][tiles]), which saves that pass's histogram launch.
template <int ITEMS>
__global__ __launch_bounds__(256) void bin_keys_kernel(const float* __restrict__ photons, RecLayout R, int n, GridDev G,
                                                       uint32_t cells, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals, uint32_t* __restrict__ cell_start,
                                                       uint32_t* __restrict__ hist, uint32_t num_tiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    // preset the run-start table to "none" on the way (saves a separate fill launch)
    {
        const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, gn = gridDim.x * blockDim.x;
        const uint32_t nvec = (reinterpret_cast<uintptr_t>(cell_start) & 15u) == 0 ? (cells + 1) / 4 : 0;  // 16-byte stores
        for (uint32_t e = gt; e < nvec; e += gn) reinterpret_cast<uint4*>(cell_start)[e] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (uint32_t e = 4 * nvec + gt; e <= cells; e += gn) cell_start[e] = 0xffffffffu;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = (int)blockIdx.x * (256 * ITEMS) + k * 256 + (int)threadIdx.x;
        if (i < n) {
            float4 a = *rec_at(photons, R, (size_t)i);
            uint32_t key = cells;
            if (!(a.x == kFltMax || a.y == kFltMax || a.z == kFltMax)) {
                float fx = min_(max_(__builtin_floorf(a.x * (float)G.dx), 0.0f), (float)(G.dx - 1));
                float fy = min_(max_(__builtin_floorf(a.y * (float)G.dy), 0.0f), (float)(G.dy - 1));
                float fz = min_(max_(__builtin_floorf(a.z * (float)G.dz), 0.0f), (float)(G.dz - 1));
                key = (uint32_t)(int)fx + (uint32_t)G.dx * ((uint32_t)(int)fy + (uint32_t)G.dy * (uint32_t)(int)fz);
            }
            keys[i] = key;
            vals[i] = (uint32_t)i;
            if (hist) atomicAdd(&h[key & 255u], 1u);
        }
    }
    if (hist) {
        __syncthreads();
        hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = h[threadIdx.x];
    }
}

// order[j] = sorted photon index; compact (pos, power) records in cell order; and the start of
// every run of equal keys is dropped into cell_start[key] (the table was preset to 0xffffffff).
// Work is partitioned by photon, so clustered photons do not unbalance it.
__global__ __launch_bounds__(256) void bin_finalize_kernel(const float* __restrict__ photons, RecLayout R,
                                                           const uint32_t* __restrict__ sorted_keys,
                                                           const uint32_t* __restrict__ sorted_vals, int n,
                                                           int channels, uint32_t* __restrict__ order,
                                                           float* __restrict__ sorted, uint32_t* __restrict__ cell_start) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t id = sorted_vals[j];
    uint32_t key = sorted_keys[j];
    if (j == 0 || sorted_keys[j - 1] != key) cell_start[key] = (uint32_t)j;  // key <= cells: inside the table
    order[j] = id;
    const float4* q = rec_at(photons, R, (size_t)id);
    float4 a = q[0];
    if (channels == 1) {
        reinterpret_cast<float4*>(sorted)[j] = a;
    } else {
        float4 b = q[R.b];
        float4* o = reinterpret_cast<float4*>(sorted) + 2 * (size_t)j;
        o[0] = a;
        o[1] = make_float4(b.x, b.y, 0.f, 0.f);
    }
}

// cell_start[c] = first j with key[j] >= c, c = 0..cells: a reverse (suffix) min-scan over the
// table of run starts.  One workgroup owns kCsTile consecutive entries; what lies behind its
// tile is summarised by one (wave-wide, 64-ary) search in the sorted keys (the first j with key >= tile end),
// so the tiles are independent: one coalesced read and one coalesced write of the table.
constexpr int kCsTile = 2048;
constexpr int kCsPer = kCsTile / 256;

__global__ __launch_bounds__(256) void cell_start_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t cells,
                                                         uint32_t* __restrict__ cell_start) {
    __shared__ uint32_t s_hi;
    __shared__ uint32_t wmin[4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t c0 = blockIdx.x * (uint32_t)kCsTile;
    const uint32_t c1 = min(c0 + (uint32_t)kCsTile, cells + 1);  // entries [c0, c1)
    // thread t owns entries [c0 + t*kCsPer, +kCsPer): two 16-byte loads, issued before the search so that their latency
    // hides behind it (the ragged last tile, or a table that is not 16-byte aligned, goes entry by entry)
    static_assert(kCsPer == 8, "two uint4 per thread");
    uint32_t v[kCsPer];
    const uint32_t base = c0 + t * kCsPer;
    const bool vec = base + kCsPer <= c1 && (reinterpret_cast<uintptr_t>(cell_start) & 15u) == 0;
    if (vec) {
        const uint4 a = *reinterpret_cast<const uint4*>(cell_start + base);
        const uint4 b = *reinterpret_cast<const uint4*>(cell_start + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < kCsPer; ++i) v[i] = (base + i < c1) ? cell_start[base + i] : 0xffffffffu;
    }
    if (wave == 0) {
        // first j with keys[j] >= c1 by a 64-ary search: 64 probes per round trip instead of one (a binary search is 20
        // DEPENDENT loads, ~10 us of latency that was the whole kernel; 1 M keys -> 16 K -> 256 -> 4 -> done)
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t step = (hi - lo + 63u) >> 6;
            const uint32_t pos = lo + lane * step;
            const bool below = pos < hi && keys[pos] < c1;           // monotone in lane: keys are sorted
            const uint32_t cnt = (uint32_t)__popcll(__ballot(below));
            if (cnt == 0) break;                                      // keys[lo] >= c1
            const uint32_t last_below = lo + (cnt - 1) * step;
            hi = min(hi, last_below + step);                          // first probe that was not below (or the old end)
            lo = last_below + 1;
        }
        if (lane == 0) s_hi = lo;
    }
    uint32_t m = 0xffffffffu;
#pragma unroll
    for (int i = kCsPer - 1; i >= 0; --i) { m = min(m, v[i]); v[i] = m; }
    uint32_t sfx = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_down(sfx, off, 64);
        if (lane + off < 64) sfx = min(sfx, o);
    }
    if (lane == 0) wmin[wave] = sfx;
    __syncthreads();
    uint32_t after = s_hi;
#pragma unroll
    for (int w = 3; w >= 0; --w) if (w > (int)wave) after = min(after, wmin[w]);
    uint32_t next_lane = __shfl_down(sfx, 1, 64);
    uint32_t behind = (lane == 63) ? after : min(next_lane, after);
    if (vec) {
        *reinterpret_cast<uint4*>(cell_start + base) = make_uint4(min(v[0], behind), min(v[1], behind), min(v[2], behind), min(v[3], behind));
        *reinterpret_cast<uint4*>(cell_start + base + 4) = make_uint4(min(v[4], behind), min(v[5], behind), min(v[6], behind), min(v[7], behind));
    } else {
#pragma unroll
        for (int i = 0; i < kCsPer; ++i)
            if (base + i < c1) cell_start[base + i] = min(v[i], behind);
    }
}

// ---- gather
//
// One work-item per voxel; a wave owns a compact 4x4x4 brick of voxels (photons cluster on
// surfaces: with row-shaped waves every wave would carry a few heavy lanes, with bricks whole
// waves are heavy or idle).  Per neighbour row the records of cells [x-R, x+R] are one
// contiguous run of the cell-sorted array; rows and records are visited in ascending order,
// i.e. per voxel in ascending sorted index: the sequential fp32 sum the contract defines.
constexpr int kGW = 4;  // brick edge (voxels)

template <int CH>
CPM_DEV void gather_pair(const GridDev& G, float4 a, float pg, float pb, f3 c, int x, int y, int z, float radius,
                         float d2, float k, float& sr, float& sg, float& sb) {
    f3 p = { a.x, a.y, a.z };
    Box3 bb = splat_box(G, p, radius);
    if (x < bb.sx || x >= bb.ex || y < bb.sy || y >= bb.ey || z < bb.sz || z >= bb.ez) return;
    float w = density_kernel_(__builtin_sqrtf(d2) / radius);
    float vr = (a.w * k) * w;
    if (vr != 0.f) sr += vr;
    if (CH == 4) {
        float vg = (pg * k) * w, vb = (pb * k) * w;
        if (vg != 0.f) sg += vg;
        if (vb != 0.f) sb += vb;
    }
}

// The kernel:
//   * a brick-level early-out: 36 lanes fetch the two cell-start boundaries of the brick's halo
//     rows; an empty halo (the common case) costs one load latency and one store;
//   * compaction: the cheap exact reject runs per record, survivors are parked as (j, d^2) in a
//     per-lane LDS queue ([slot][thread]: conflict-free), and the reference's box test, sqrt and
//     division run over the queues when some lane's queue is full -- so the expensive path
//     executes with most lanes busy instead of once per record with one lane in ten active.
//     A lane drains its queue in the order it filled it: the summation order is unchanged.
constexpr int kQ = 8;
template <int CH, int BATCH>
__global__ __launch_bounds__(256) void gather_voxel_kernel(const float* __restrict__ sorted,
                                                            const uint32_t* __restrict__ cell_start, GridDev G, float radius,
                                                            float r2max, float k, int Rx, int Ry, int Rz, int accumulate,
                                                            int bxn, int byn, float* __restrict__ out,
                                                            unsigned long long* __restrict__ dbg,
                                                            const uint8_t* __restrict__ bmask) {
    constexpr int STRIDE = (CH == 1 ? 1 : 2);
    __shared__ uint32_t q_j[kQ][256];
    __shared__ float q_d2[kQ][256];
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned tests = 0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int gb = blockIdx.x * 4 + wave;
    const int by = (gb / bxn) % byn, bz = gb / (bxn * byn);
    const int bx = (gb % bxn + 4 * (by + bz)) % bxn;
    // Workgroups are dealt round-robin to the 8 XCDs by index.  Photons pile up on surfaces, so the
    // brick a slot gets is rotated along x by 4 (by + bz): every XCD then owns an even share of each
    // face of the volume instead of one XCD owning the whole x = 0 face (measured: 859 -> 180 us).
    const int x0 = bx * kGW, y0 = by * kGW, z0 = bz * kGW;
    if (z0 >= G.dz) return;
    if (bmask && !bmask[(size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz)]) return;  // cpm_gather_bricks: brick not selected
    const int x = x0 + (lane & 3), y = y0 + ((lane >> 2) & 3), z = z0 + (lane >> 4);
    const bool valid = x < G.dx && y < G.dy && z < G.dz;
    // ---- brick-level early-out
    const int nry = kGW + 2 * Ry, nrows = nry * (kGW + 2 * Rz);
    bool any = false;
    for (int r = lane; r < nrows; r += 64) {
        const int cy = y0 - Ry + (r % nry), cz = z0 - Rz + (r / nry);
        if (cy >= 0 && cy < G.dy && cz >= 0 && cz < G.dz) {
            const uint32_t row = (uint32_t)G.dx * ((uint32_t)cy + (uint32_t)G.dy * (uint32_t)cz);
            any |= cell_start[row + (uint32_t)min(x0 + kGW - 1 + Rx, G.dx - 1) + 1] != cell_start[row + (uint32_t)max(x0 - Rx, 0)];
        }
    }
    float sr = 0.f, sg = 0.f, sb = 0.f;
    if (__any(any) && valid) {
        f3 vi = { (float)x, (float)y, (float)z };
        const f3 c = transform_(G.i2t, vi);
        const int xlo = max(x - Rx, 0), xhi = min(x + Rx, G.dx - 1);
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
        int qn = 0;
        auto drain = [&]() {
            for (int s = 0; s < kQ; ++s) {
                if (!__any(s < qn)) break;
                if (s < qn) {
                    const uint32_t jj = q_j[s][t];
                    const float d2 = q_d2[s][t];
                    float4 a = rec[STRIDE * (size_t)jj];
                    float pg = 0.f, pb = 0.f;
                    if (CH == 4) { float4 q = rec[2 * (size_t)jj + 1]; pg = q.x; pb = q.y; }
                    gather_pair<CH>(G, a, pg, pb, c, x, y, z, radius, d2, k, sr, sg, sb);
                }
            }
            qn = 0;
        };
        for (int cz = z - Rz; cz <= z + Rz; ++cz) {
            if (cz < 0 || cz >= G.dz) continue;
            for (int cy = y - Ry; cy <= y + Ry; ++cy) {
                if (cy < 0 || cy >= G.dy) continue;
                uint32_t row = (uint32_t)G.dx * ((uint32_t)cy + (uint32_t)G.dy * (uint32_t)cz);
                uint32_t j = cell_start[row + xlo];
                const uint32_t je = cell_start[row + xhi + 1];
                while (j < je) {
                    float4 a[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) a[u] = rec[STRIDE * (size_t)min(j + u, je - 1)];
                    if (__any(qn > kQ - BATCH)) drain();  // wave-uniform: everyone drains together
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        float ddx = c.x - a[u].x, ddy = c.y - a[u].y, ddz = c.z - a[u].z;
                        float d2 = fma_(ddz, ddz, fma_(ddy, ddy, ddx * ddx));
                        if (j + u < je) {
                            ++tests;
                            if (d2 <= r2max) { q_j[qn][t] = j + u; q_d2[qn][t] = d2; ++qn; }  // else: exactly no contribution
                        }
                    }
                    j += BATCH;
                }
            }
        }
        drain();
    }
    if (dbg) {
        unsigned mx = tests, sm = tests;
        for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (unsigned)__shfl_xor(mx, off, 64)); sm += __shfl_xor(sm, off, 64); }
        if (lane == 0) {
            unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
            const size_t gbs = (size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz);
            dbg[4 * gbs + 0] = t_start;
            dbg[4 * gbs + 1] = __builtin_amdgcn_s_memrealtime();
            dbg[4 * gbs + 2] = ((unsigned long long)mx << 32) | sm;
            dbg[4 * gbs + 3] = xcc;
        }
    }
    if (!valid) return;
    const uint32_t v = (uint32_t)x + (uint32_t)G.dx * ((uint32_t)y + (uint32_t)G.dy * (uint32_t)z);
    if (CH == 1) {
        out[v] = accumulate ? out[v] + sr : sr;
    } else {
        float4* o = reinterpret_cast<float4*>(out) + v;
        if (accumulate) { float4 tt = *o; *o = make_float4(tt.x + sr, tt.y + sg, tt.z + sb, tt.w); }
        else *o = make_float4(sr, sg, sb, 0.f);
    }
}

// ---- record-major gather (single channel, <= 3 candidate voxels per axis) ---------------------------
//
// The voxel-major kernel above makes every lane stream its own window of records: a wave-load
// touches up to 64 cache lines and the heaviest voxel (hundreds of candidates) sets the kernel
// time.  Here lanes own RECORDS instead.  One wave = one 4x4x4 brick of voxels:
//   1. halo row runs (jb, len) -> exclusive prefix: the brick's records form one flat sequence,
//      ascending in sorted index; it is consumed 64 records per step, lane l = record base + l
//      (coalesced 16-byte loads);
//   2. a record can only reach voxels whose centre lies within r (+1e-3 slack) of it per axis:
//      <= 2 per axis at r < 1 voxel, i.e. <= 8 candidates (<= 27 in general here).  The box test
//      is per record; per candidate: centre, d^2, exact cheap reject, sqrt, division, kernel.
//      A non-zero contribution goes to the LDS slot s_val[candidate][lane] and the lane's bit is
//      OR-ed into the target voxel's 64-bit mask (ds_or_b64: order-free, hence deterministic);
//   3. lane v then walks the set bits of its voxel's mask in ascending lane order and adds the
//      slots -- ascending lanes and ascending steps are ascending sorted index, so each voxel
//      performs exactly the sequential fp32 sum the contract defines.
// No global-memory divergence, no per-lane chains, no workgroup barrier.
template <int MAXC>
__global__ __launch_bounds__(256) void gather_records_kernel(const float* __restrict__ sorted,
                                                             const uint32_t* __restrict__ cell_start, GridDev G, float radius,
                                                             float r2max, float k, int Rx, int Ry, int Rz, int accumulate,
                                                             int bxn, int byn, float* __restrict__ out,
                                                             unsigned long long* __restrict__ dbg,
                                                             const uint8_t* __restrict__ bmask) {
    constexpr int NC = MAXC * MAXC * MAXC;
    constexpr int MAXROWS = 64;
    __shared__ float s_val_all[4][NC][64];
    __shared__ unsigned long long s_mask_all[4][64];
    __shared__ uint32_t s_base_all[4][64];
    __shared__ uint32_t s_off_all[4][MAXROWS + 1];
    __shared__ uint32_t s_jb_all[4][MAXROWS];
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*s_val)[64] = s_val_all[wave];
    unsigned long long* s_mask = s_mask_all[wave];
    uint32_t* s_base = s_base_all[wave];
    uint32_t* s_off = s_off_all[wave];
    uint32_t* s_jb = s_jb_all[wave];

    const int gb = blockIdx.x * 4 + wave;
    const int by = (gb / bxn) % byn, bz = gb / (bxn * byn);
    const int bx = (gb % bxn + 4 * (by + bz)) % bxn;  // XCD-balancing rotation (see gather_voxel_kernel)
    const int x0 = bx * kGW, y0 = by * kGW, z0 = bz * kGW;
    if (z0 >= G.dz) return;
    if (bmask && !bmask[(size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz)]) return;  // cpm_gather_bricks
    const int x = x0 + (lane & 3), y = y0 + ((lane >> 2) & 3), z = z0 + (lane >> 4);
    const bool valid = x < G.dx && y < G.dy && z < G.dz;

    // ---- 1. halo rows
    const int nry = kGW + 2 * Ry, nrows = nry * (kGW + 2 * Rz);
    uint32_t jb = 0, len = 0;
    if (lane < nrows) {
        const int cy = y0 - Ry + (lane % nry), cz = z0 - Rz + (lane / nry);
        if (cy >= 0 && cy < G.dy && cz >= 0 && cz < G.dz) {
            const uint32_t row = (uint32_t)G.dx * ((uint32_t)cy + (uint32_t)G.dy * (uint32_t)cz);
            jb = cell_start[row + (uint32_t)max(x0 - Rx, 0)];
            len = cell_start[row + (uint32_t)min(x0 + kGW - 1 + Rx, G.dx - 1) + 1] - jb;
        }
    }
    uint32_t incl = len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    float sum = 0.f;
    if (total != 0) {  // wave-uniform
        if (lane < nrows) { s_off[lane] = incl - len; s_jb[lane] = jb; }
        if (lane == 0) s_off[nrows] = total;
        s_mask[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
        const float rgx = radius * (float)G.dx + 1e-3f, rgy = radius * (float)G.dy + 1e-3f, rgz = radius * (float)G.dz + 1e-3f;
        const int x1 = min(x0 + kGW - 1, G.dx - 1), y1 = min(y0 + kGW - 1, G.dy - 1), z1 = min(z0 + kGW - 1, G.dz - 1);
        for (uint32_t base = 0; base < total; base += 64) {
            const uint32_t i = base + lane;
            const bool have = i < total;
            // row of record i: the last r with s_off[r] <= i
            int lo = 0, hi = nrows - 1;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= i) lo = mid; else hi = mid - 1;
            }
            const uint32_t j = s_jb[lo] + (i - s_off[lo]);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (have) a = rec[j];
            const f3 p = { a.x, a.y, a.z };
            // candidate voxels: centres within r (+ slack) per axis, inside this brick
            const f3 u = transform_(G.t2i, p);  // index space: voxel v has its centre at u == v
            int sx = max((int)__builtin_ceilf(u.x - rgx), x0), ex = min((int)__builtin_floorf(u.x + rgx), x1);
            int sy = max((int)__builtin_ceilf(u.y - rgy), y0), ey = min((int)__builtin_floorf(u.y + rgy), y1);
            int sz = max((int)__builtin_ceilf(u.z - rgz), z0), ez = min((int)__builtin_floorf(u.z + rgz), z1);
            const int nx = max(ex - sx + 1, 0), ny = max(ey - sy + 1, 0), nz = max(ez - sz + 1, 0);
            const int ncand = have ? nx * ny * nz : 0;
            s_base[lane] = (uint32_t)(sx - x0) | ((uint32_t)(sy - y0) << 2) | ((uint32_t)(sz - z0) << 4) | ((uint32_t)nx << 6) | ((uint32_t)ny << 8);
            const Box3 bb = splat_box(G, p, radius);
            const float pk = a.w * k;
            int cx = 0, cy = 0, cz = 0;
            for (int c = 0; c < NC; ++c) {
                if (!__any(c < ncand)) break;
                if (c < ncand) {
                    const int vx = sx + cx, vy = sy + cy, vz = sz + cz;
                    f3 vi = { (float)vx, (float)vy, (float)vz };
                    const f3 cc = transform_(G.i2t, vi);
                    const float ddx = cc.x - a.x, ddy = cc.y - a.y, ddz = cc.z - a.z;
                    const float d2 = fma_(ddz, ddz, fma_(ddy, ddy, ddx * ddx));
                    if (d2 <= r2max &&  // otherwise exactly no contribution
                        vx >= bb.sx && vx < bb.ex && vy >= bb.sy && vy < bb.ey && vz >= bb.sz && vz < bb.ez) {
                        const float w = density_kernel_(__builtin_sqrtf(d2) / radius);
                        const float val = pk * w;
                        if (val != 0.f) {
                            s_val[c][lane] = val;
                            const int vl = (vx - x0) + 4 * (vy - y0) + 16 * (vz - z0);
                            atomicOr(&s_mask[vl], 1ull << lane);
                        }
                    }
                    if (++cx == nx) { cx = 0; if (++cy == ny) { cy = 0; ++cz; } }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- 3. lane = voxel: add this step's contributors in ascending lane (= sorted index) order
            unsigned long long m = s_mask[lane];
            if (m) {
                s_mask[lane] = 0ull;
                const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
                do {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t b = s_base[l];
                    const int bsx = b & 3, bsy = (b >> 2) & 3, bsz = (b >> 4) & 3, bnx = (b >> 6) & 3, bny = (b >> 8) & 3;
                    const int c = ((lz - bsz) * bny + (ly - bsy)) * bnx + (lx - bsx);
                    sum += s_val[c][l];
                } while (m);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (dbg && lane == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
        const size_t gbs = (size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz);
        dbg[4 * gbs + 0] = t_start;
        dbg[4 * gbs + 1] = __builtin_amdgcn_s_memrealtime();
        dbg[4 * gbs + 2] = total;
        dbg[4 * gbs + 3] = xcc;
    }
    if (!valid) return;
    const uint32_t v = (uint32_t)x + (uint32_t)G.dx * ((uint32_t)y + (uint32_t)G.dy * (uint32_t)z);
    out[v] = accumulate ? out[v] + sum : sum;
}

// ---- the record phase shared by the tuned record-major kernels ---------------------------------------
// MAXC = candidate voxels per axis a record can reach (2 at r < 1 cell, 3 at r < 1.5 cells).  A record's candidates
// are consecutive integers per axis, so they differ in their residues mod MAXC: the slot a contribution is parked
// in is the residue triple of the TARGET voxel -- which the voxel's own lane knows without looking anything up.
//   * per axis the MAXC centre offsets d = c - p and box verdicts are computed once per record, so the exact cheap
//     test of a candidate costs two fma and a compare (d^2 = fma(dz, dz, fma(dy, dy, dx*dx)), the contract's operands);
//   * about 2 % (MAXC = 2) of the (record, candidate) pairs pass; the weight (sqrt, division, kernel: ~40
//     instructions) is evaluated per surviving candidate RANK -- first survivor of every lane together, then the
//     second, ... -- i.e. a few times per step instead of once per candidate with one or two lanes active.
template <int MAXC> struct CandSlots { static constexpr int N = MAXC * MAXC * MAXC; };
CPM_DEV int mod3_(int v) { return v - 3 * ((v * 43691) >> 17); }  // v in [0, 2^15): grid coordinates

template <int MAXC>
CPM_DEV int voxel_slot(int x, int y, int z) {
    if (MAXC == 2) return (x & 1) | ((y & 1) << 1) | ((z & 1) << 2);
    if (MAXC == 4) return (x & 3) | ((y & 3) << 2) | ((z & 3) << 4);
    return mod3_(x) + 3 * mod3_(y) + 9 * mod3_(z);
}

// CH = 1: one value per contribution (power.r); CH = 4: three (r, g, b), parked in three planes of NC slots each.  A
// contribution is kept when any channel is non-zero and the drain adds all of its channels: adding a zero is the
// identity here (sums are never -0), so this equals the per-channel `if (v != 0) s += v` of the contract.
template <int MAXC, int CH>
CPM_DEV void gather_record_phase(const GridDev& G, float4 a, float pg, float pb, bool have, int x0, int y0, int z0, int x1,
                                 int y1, int z1, float rgx, float rgy, float rgz, float radius, float r2max, float k, int lane,
                                 float (*s_val)[64], unsigned long long* s_mask) {
    constexpr int NC = MAXC * MAXC * MAXC;
    const f3 p = { a.x, a.y, a.z };
    const f3 u = transform_(G.t2i, p);  // index space: voxel v has its centre at u == v
    const int sx = max((int)__builtin_ceilf(u.x - rgx), x0), ex = min((int)__builtin_floorf(u.x + rgx), x1);
    const int sy = max((int)__builtin_ceilf(u.y - rgy), y0), ey = min((int)__builtin_floorf(u.y + rgy), y1);
    const int sz = max((int)__builtin_ceilf(u.z - rgz), z0), ez = min((int)__builtin_floorf(u.z + rgz), z1);
    const int nx = have ? max(ex - sx + 1, 0) : 0, ny = max(ey - sy + 1, 0), nz = max(ez - sz + 1, 0);
    const Box3 bb = splat_box(G, p, radius);
    const float pk = a.w * k, pkg = pg * k, pkb = pb * k;
    float dxv[MAXC], dyv[MAXC], dzv[MAXC];
    bool okx[MAXC], oky[MAXC], okz[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
        const int vx = sx + q, vy = sy + q, vz = sz + q;
        dxv[q] = fma_(G.i2t.sx, (float)vx, G.i2t.tx) - a.x;
        dyv[q] = fma_(G.i2t.sy, (float)vy, G.i2t.ty) - a.y;
        dzv[q] = fma_(G.i2t.sz, (float)vz, G.i2t.tz) - a.z;
        okx[q] = q < nx && vx >= bb.sx && vx < bb.ex;
        oky[q] = q < ny && vy >= bb.sy && vy < bb.ey;
        okz[q] = q < nz && vz >= bb.sz && vz < bb.ez;
    }
    unsigned long long hits = 0;  // one bit per candidate (64 at MAXC = 4)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int qx = c % MAXC, qy = (c / MAXC) % MAXC, qz = c / (MAXC * MAXC);
        const float d2 = fma_(dzv[qz], dzv[qz], fma_(dyv[qy], dyv[qy], dxv[qx] * dxv[qx]));
        if (okx[qx] && oky[qy] && okz[qz] && d2 <= r2max) hits |= 1ull << c;  // d2 > r2max: exactly no contribution
    }
    const int vl0 = (sx - x0) + 4 * (sy - y0) + 16 * (sz - z0);
    // residues of the first candidate; candidate q's residue is (r0 + q) mod MAXC
    const int rx0 = MAXC == 3 ? mod3_(sx) : (sx & (MAXC - 1)), ry0 = MAXC == 3 ? mod3_(sy) : (sy & (MAXC - 1)),
              rz0 = MAXC == 3 ? mod3_(sz) : (sz & (MAXC - 1));
    while (__any(hits != 0)) {
        if (hits != 0) {
            const int c = __builtin_ctzll(hits);
            hits &= hits - 1;
            int qx, qy, qz;
            if (MAXC == 2) { qx = c & 1; qy = (c >> 1) & 1; qz = c >> 2; }
            else if (MAXC == 4) { qx = c & 3; qy = (c >> 2) & 3; qz = c >> 4; }
            else { qz = (c * 57) >> 9; const int r = c - 9 * qz; qy = (r * 11) >> 5; qx = r - 3 * qy; }  // c / 9, (c % 9) / 3, c % 3 for c < 27
            float ddx = dxv[0], ddy = dyv[0], ddz = dzv[0];
#pragma unroll
            for (int q = 1; q < MAXC; ++q) {
                ddx = (qx == q) ? dxv[q] : ddx;
                ddy = (qy == q) ? dyv[q] : ddy;
                ddz = (qz == q) ? dzv[q] : ddz;
            }
            const float d2 = fma_(ddz, ddz, fma_(ddy, ddy, ddx * ddx));  // the same operands as above: the same value
            const float wgt = density_kernel_(__builtin_sqrtf(d2) / radius);
            const float val = pk * wgt;
            const float valg = CH == 4 ? pkg * wgt : 0.f, valb = CH == 4 ? pkb * wgt : 0.f;
            if (val != 0.f || (CH == 4 && (valg != 0.f || valb != 0.f))) {
                int slot;
                if (MAXC == 2) {
                    slot = (rx0 ^ qx) | ((ry0 ^ qy) << 1) | ((rz0 ^ qz) << 2);
                } else if (MAXC == 4) {
                    slot = ((rx0 + qx) & 3) | (((ry0 + qy) & 3) << 2) | (((rz0 + qz) & 3) << 4);
                } else {
                    int rx = rx0 + qx, ry = ry0 + qy, rz = rz0 + qz;
                    rx -= rx >= 3 ? 3 : 0; ry -= ry >= 3 ? 3 : 0; rz -= rz >= 3 ? 3 : 0;
                    slot = rx + 3 * ry + 9 * rz;
                }
                s_val[slot][lane] = val;
                if (CH == 4) { s_val[NC + slot][lane] = valg; s_val[2 * NC + slot][lane] = valb; }
                atomicOr(&s_mask[vl0 + qx + 4 * qy + 16 * qz], 1ull << lane);
            }
        }
    }
}

// lane = voxel: add a step's contributors (bits of m) in ascending lane (= sorted index) order to the voxel's sums
template <int NC, int CH>
CPM_DEV void gather_drain(unsigned long long m, const float (*s_val)[64], int my_slot, float& sr, float& sg, float& sb) {
    const float* mine = s_val[my_slot];
    do {  // up to four contributors are fetched together, then added in order
        const int l0 = __builtin_ctzll(m); m &= m - 1;
        const int l1 = m ? __builtin_ctzll(m) : l0; const bool h1 = m != 0; m &= m - 1;
        const int l2 = m ? __builtin_ctzll(m) : l0; const bool h2 = m != 0; m &= m - 1;
        const int l3 = m ? __builtin_ctzll(m) : l0; const bool h3 = m != 0; m &= m - 1;
        const float v0 = mine[l0], v1 = mine[l1], v2 = mine[l2], v3 = mine[l3];
        sr += v0;
        if (h1) sr += v1;
        if (h2) sr += v2;
        if (h3) sr += v3;
        if (CH == 4) {
            const float* mg = s_val[NC + my_slot];
            const float* mb = s_val[2 * NC + my_slot];
            const float g0 = mg[l0], g1 = mg[l1], g2 = mg[l2], g3 = mg[l3];
            const float b0 = mb[l0], b1 = mb[l1], b2 = mb[l2], b3 = mb[l3];
            sg += g0; sb += b0;
            if (h1) { sg += g1; sb += b1; }
            if (h2) { sg += g2; sb += b2; }
            if (h3) { sg += g3; sb += b3; }
        }
    } while (m);
}

// ---- tuned record-major gather: one wave per brick (MAXC = 2, 3 or 4 candidate voxels per axis, halo of <= 2 cells)
// Same algorithm and the same per-voxel summation order as gather_records_kernel, tuned:
//   * the record phase above (per-axis terms once per record, survivors weighted per rank, residue slots);
//   * the next step's records are fetched before the current step is processed;
//   * the flattened-index -> row lookup is a popcount over a bitmask of row starts instead of a binary search;
//   * z-slabs are visited from both faces inwards (photons pile up where light enters the volume, i.e. on faces):
//     the heaviest bricks are dispatched first instead of last;
//   * 3-D launch, no integer division on the way to the brick (four fifths of the waves find an empty halo).
// Used above 64 Ki bricks, where the hardware's wave scheduling balances the load by itself; below that
// gather_coop_kernel shares bricks between waves.
template <int MAXC, int CH>
__global__ __launch_bounds__(256) void gather_records2_kernel(const float* __restrict__ sorted,
                                                              const uint32_t* __restrict__ cell_start, GridDev G, float radius,
                                                              float r2max, float k, int Rx, int Ry, int Rz, int accumulate,
                                                              int bxn, int byn, int bzn, float* __restrict__ out,
                                                              unsigned long long* __restrict__ dbg,
                                                              const uint8_t* __restrict__ bmask) {
    constexpr int MAXROWS = 64;
    constexpr int MAXWORDS = 64;  // row-start bitmask: up to 4096 records per brick halo, else the generic kernel's path
    constexpr int NC = MAXC * MAXC * MAXC, CH3 = CH == 4 ? 3 : 1, STRIDE = CH == 4 ? 2 : 1;
    __shared__ float s_val_all[4][CH3 * NC][64];
    __shared__ unsigned long long s_mask_all[4][64];
    __shared__ unsigned long long s_start_all[4][MAXWORDS];
    __shared__ uint32_t s_rowjb_all[4][MAXROWS];   // per NON-EMPTY row (compacted): jb - exclusive offset
    __shared__ uint32_t s_rowoff_all[4][MAXROWS];  // per non-empty row: exclusive offset
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*s_val)[64] = s_val_all[wave];
    unsigned long long* s_mask = s_mask_all[wave];
    unsigned long long* s_start = s_start_all[wave];
    uint32_t* s_rowjb = s_rowjb_all[wave];
    uint32_t* s_rowoff = s_rowoff_all[wave];

    // 3-D launch (x: 4 bricks per workgroup, y, z): no integer divisions on the way to the brick -- four fifths of
    // the waves find an empty halo and exit, so their prologue is a sixth of the kernel's instructions
    const int bxr = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave);
    if (bxr >= bxn) return;
    const int by = (int)blockIdx.y, bzi = (int)blockIdx.z;
    const int bz = (bzi & 1) ? (bzi >> 1) : (bzn - 1 - (bzi >> 1));  // bzn-1, 0, bzn-2, 1, ...
    const int bx = (bxr + 4 * (by + bzi)) % bxn;                     // XCD-balancing rotation (scalar)
    if (bmask && !bmask[(size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz)]) return;  // cpm_gather_bricks: brick not selected
    const int x0 = bx * kGW, y0 = by * kGW, z0 = bz * kGW;
    const int x = x0 + (lane & 3), y = y0 + ((lane >> 2) & 3), z = z0 + (lane >> 4);
    const bool valid = x < G.dx && y < G.dy && z < G.dz;
    const int my_par = voxel_slot<MAXC>(x, y, z);

    const int nry = kGW + 2 * Ry, nrows = nry * (kGW + 2 * Rz);  // Ry <= 1: 4 or 6 rows per slab
    uint32_t jb = 0, len = 0;
    if (lane < nrows) {
        const int rz = nry == 4 ? lane >> 2 : (nry == 6 ? (lane * 43) >> 8 : lane >> 3);  // lane / nry (exact for lane < 64)
        const int cy = y0 - Ry + (lane - rz * nry), cz = z0 - Rz + rz;
        if (cy >= 0 && cy < G.dy && cz >= 0 && cz < G.dz) {
            const uint32_t row = (uint32_t)G.dx * ((uint32_t)cy + (uint32_t)G.dy * (uint32_t)cz);
            jb = cell_start[row + (uint32_t)max(x0 - Rx, 0)];
            len = cell_start[row + (uint32_t)min(x0 + kGW - 1 + Rx, G.dx - 1) + 1] - jb;
        }
    }
    float sum = 0.f, sumg = 0.f, sumb = 0.f;
    uint32_t incl = len, total = 0;
    if (__any(len != 0)) {  // wave-uniform; an empty halo skips the prefix sum as well
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        total = __shfl(incl, 63, 64);
    }
    if (total != 0) {  // wave-uniform
        const int nwords = (int)((total + 63) >> 6);
        s_mask[lane] = 0ull;
        if (lane < MAXWORDS) s_start[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        // compacted non-empty rows: row k starts at flattened index off_k; lookup value = jb - off
        const unsigned long long ne = __ballot(len != 0);
        if (len != 0) {
            const uint32_t off = incl - len;
            const int kk = __popcll(ne & ((1ull << lane) - 1ull));
            s_rowjb[kk] = jb - off;
            s_rowoff[kk] = off;
            if (total <= (uint32_t)MAXWORDS * 64u) atomicOr(&s_start[off >> 6], 1ull << (off & 63));
        }
        __builtin_amdgcn_wave_barrier();
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
        const float rgx = radius * (float)G.dx + 1e-3f, rgy = radius * (float)G.dy + 1e-3f, rgz = radius * (float)G.dz + 1e-3f;
        const int x1 = min(x0 + kGW - 1, G.dx - 1), y1 = min(y0 + kGW - 1, G.dy - 1), z1 = min(z0 + kGW - 1, G.dz - 1);
        const bool fast_rows = total <= (uint32_t)MAXWORDS * 64u;

        // record index of flattened position i (step `w`, lane l): the row is found by counting the row
        // starts at or before i -- a popcount over the start bitmask (or, for > 4096 records around one
        // brick, a binary search over the compacted row offsets)
        const int nne = __popcll(ne);
        uint32_t rows_before = 0;  // non-empty rows that start before the current step (wave-uniform)
        auto locate = [&](int w, uint32_t& rb_next) -> uint32_t {
            const uint32_t i = (uint32_t)w * 64u + lane;
            if (fast_rows) {
                const unsigned long long sb = s_start[w];
                const int kk = (int)rows_before + __popcll(sb & ((2ull << lane) - 1ull)) - 1;
                rb_next = rows_before + (uint32_t)__popcll(sb);
                return s_rowjb[max(kk, 0)] + i;
            }
            int lo = 0, hi = nne - 1;
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_rowoff[mid] <= i) lo = mid; else hi = mid - 1;
            }
            rb_next = rows_before;
            return s_rowjb[lo] + i;
        };

        uint32_t rb_next = 0;
        uint32_t jcur = locate(0, rb_next);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a;
        if ((uint32_t)lane < total) { a = rec[STRIDE * (size_t)jcur]; if (CH == 4) a2 = rec[2 * (size_t)jcur + 1]; }
        for (int w = 0; w < nwords; ++w) {
            const bool have = (uint32_t)w * 64u + lane < total;
            rows_before = rb_next;
            // prefetch the next step's records
            float4 an = make_float4(0.f, 0.f, 0.f, 0.f), an2 = an;
            if (w + 1 < nwords) {
                const uint32_t jn = locate(w + 1, rb_next);
                if ((uint32_t)(w + 1) * 64u + lane < total) { an = rec[STRIDE * (size_t)jn]; if (CH == 4) an2 = rec[2 * (size_t)jn + 1]; }
            }
            gather_record_phase<MAXC, CH>(G, a, a2.x, a2.y, have, x0, y0, z0, x1, y1, z1, rgx, rgy, rgz, radius, r2max, k, lane, s_val,
                                          s_mask);
            __builtin_amdgcn_wave_barrier();
            // lane = voxel: add this step's contributors in ascending lane (= sorted index) order
            unsigned long long m = s_mask[lane];
            if (m) {
                s_mask[lane] = 0ull;
                gather_drain<NC, CH>(m, s_val, my_par, sum, sumg, sumb);
            }
            __builtin_amdgcn_wave_barrier();
            a = an; a2 = an2;
        }
    }
    if (dbg && lane == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
        const size_t gbs = (size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz);
        dbg[4 * gbs + 0] = t_start;
        dbg[4 * gbs + 1] = __builtin_amdgcn_s_memrealtime();
        dbg[4 * gbs + 2] = total;
        dbg[4 * gbs + 3] = xcc;
    }
    if (!valid) return;
    const uint32_t v = (uint32_t)x + (uint32_t)G.dx * ((uint32_t)y + (uint32_t)G.dy * (uint32_t)z);
    if (CH == 1) {
        out[v] = accumulate ? out[v] + sum : sum;
    } else {
        float4* o = reinterpret_cast<float4*>(out) + v;
        if (accumulate) { float4 tt = *o; *o = make_float4(tt.x + sum, tt.y + sumg, tt.z + sumb, tt.w); }
        else *o = make_float4(sum, sumg, sumb, 0.f);
    }
}

// ---- cooperative record-major gather --------------------------------------------------------------------
// gather_records2_kernel ends when its heaviest bricks do: photons pile up on faces, a face brick holds 1000+
// records, and ONE wave walks them 64 at a time -- 16+ dependent steps of ~650 instructions (45-68 us of the
// kernel's 80) while most SIMDs have run dry.  Here a workgroup's four waves own four bricks TOGETHER:
//   * the bricks of a workgroup are taken from four z-slabs a quarter of the grid apart (one near a face, three
//     further in), so workgroups carry similar amounts of work;
//   * phase 0: wave b builds brick b's row tables in LDS (as before); barrier;
//   * phase 1: the steps of the four bricks form one sequence g = 0, 1, ...; wave w takes g = w, w+4, ...:
//     it loads the 64 records of step g and evaluates their candidates into ITS OWN slot matrix and masks --
//     the expensive part, now four steps at a time per workgroup whatever brick they belong to;
//   * the drains must stay in record order per voxel (the summation contract), so they take turns: a wave waits
//     until the LDS turn counter reaches g, its voxel lanes add the step's contributions to the brick's running
//     sums in LDS, and it passes the turn on.  All four waves are resident (one workgroup), the holder of the
//     turn never waits: no deadlock;
//   * phase 2: wave b writes brick b's sums.
// Same additions in the same order as gather_records2_kernel: bit-identical results.
template <int NB, int MAXC, int CH>  // NB bricks = waves per workgroup; MAXC candidate voxels per axis; CH channels
__global__ __launch_bounds__(64 * NB) void gather_coop_kernel(const float* __restrict__ sorted,
                                                          const uint32_t* __restrict__ cell_start, GridDev G, float radius,
                                                          float r2max, float k, int Rx, int Ry, int Rz, int accumulate,
                                                          int bxn, int byn, int bzn, int zq, float* __restrict__ out,
                                                          unsigned long long* __restrict__ dbg,
                                                          const uint8_t* __restrict__ bmask) {
    constexpr int MAXROWS = 64;
    constexpr int MAXWORDS = 64;  // row-start bitmask: up to 4096 records per brick halo, else a binary search
    constexpr int NC = MAXC * MAXC * MAXC, CH3 = CH == 4 ? 3 : 1, STRIDE = CH == 4 ? 2 : 1;
    __shared__ float s_val_all[NB][CH3 * NC][64];               // per WAVE: this step's contributions by target parity
    __shared__ unsigned long long s_mask_all[NB][64];    // per WAVE: this step's contributor lanes per voxel
    __shared__ unsigned long long s_start_all[NB][MAXWORDS];  // per BRICK
    __shared__ uint32_t s_rb_all[NB][MAXWORDS];          // per BRICK: non-empty rows that start before step w
    __shared__ uint32_t s_rowjb_all[NB][MAXROWS];        // per BRICK, compacted non-empty rows: jb - exclusive offset
    __shared__ uint32_t s_rowoff_all[NB][MAXROWS];       // per BRICK: exclusive offset
    __shared__ float s_sum_all[NB][CH3][64];             // per BRICK: running per-voxel sums
    __shared__ uint32_t s_total[NB], s_nne[NB], s_first[NB + 1];
    __shared__ uint32_t s_turn;
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int byw = (int)blockIdx.y, zj = (int)blockIdx.z;
    // workgroups go to XCDs round-robin by linear index = blockIdx.x mod 8 here (bxn % 8 == 0): let one XCD own a
    // contiguous eighth of the x range, so that bricks sharing halo rows (x, y and z neighbours) share an L2
    const int bxr = (bxn & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7u) * (bxn >> 3) + (int)(blockIdx.x >> 3);
    // Brick b of this workgroup lies a quarter of the grid further along y AND along z (z in faces-first order) than
    // brick b - 1: photons pile up on the faces light enters through, and a workgroup should not own four bricks of
    // one face.  x is rotated by (y + z) so that the bricks of the x = 0 face meet every residue of blockIdx.x mod 8,
    // i.e. every XCD (workgroups are dealt to XCDs round-robin by linear index).  For fixed b the map from
    // (blockIdx.x, .y, .z) to bricks is a bijection onto the b-th quarter of the z-slabs: every brick exactly once.
    const int yq = (byn + NB - 1) / NB;
    auto brick_origin = [&](int b, int& x0, int& y0, int& z0) -> bool {
        const int bzi = zj + b * zq;
        if (bzi >= bzn) return false;
        const int bz = (bzi & 1) ? (bzi >> 1) : (bzn - 1 - (bzi >> 1));  // bzn-1, 0, bzn-2, 1, ...
        const int by = (byw + b * yq) % byn;
        const int bx = (bxr + by + bzi) % bxn;
        x0 = bx * kGW; y0 = by * kGW; z0 = bz * kGW;
        // cpm_gather_bricks: a brick that is not selected has no records here and is not written
        return !bmask || bmask[(size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz)] != 0;
    };

    // ---- phase 0: wave b prepares brick b
    {
        int x0 = 0, y0 = 0, z0 = 0;
        const bool exists = brick_origin(wave, x0, y0, z0);
        const int nry = kGW + 2 * Ry, nrows = nry * (kGW + 2 * Rz);
        uint32_t jb = 0, len = 0;
        if (exists && lane < nrows) {
            const int rz = nry == 4 ? lane >> 2 : (nry == 6 ? (lane * 43) >> 8 : lane >> 3);  // lane / nry (exact for lane < 64)
            const int cy = y0 - Ry + (lane - rz * nry), cz = z0 - Rz + rz;
            if (cy >= 0 && cy < G.dy && cz >= 0 && cz < G.dz) {
                const uint32_t row = (uint32_t)G.dx * ((uint32_t)cy + (uint32_t)G.dy * (uint32_t)cz);
                jb = cell_start[row + (uint32_t)max(x0 - Rx, 0)];
                len = cell_start[row + (uint32_t)min(x0 + kGW - 1 + Rx, G.dx - 1) + 1] - jb;
            }
        }
        uint32_t incl = len, total = 0;
        const unsigned long long ne = __ballot(len != 0);
        if (ne) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            total = __shfl(incl, 63, 64);
        }
#pragma unroll
        for (int ch = 0; ch < CH3; ++ch) s_sum_all[wave][ch][lane] = 0.f;
        s_mask_all[wave][lane] = 0ull;
        s_start_all[wave][lane] = 0ull;
        if (lane == 0) { s_total[wave] = total; s_nne[wave] = (uint32_t)__popcll(ne); if (wave == 0) s_turn = 0u; }
        __builtin_amdgcn_wave_barrier();
        if (len != 0) {
            const uint32_t off = incl - len;
            const int kk = __popcll(ne & ((1ull << lane) - 1ull));
            s_rowjb_all[wave][kk] = jb - off;
            s_rowoff_all[wave][kk] = off;
            if (total <= (uint32_t)MAXWORDS * 64u) atomicOr(&s_start_all[wave][off >> 6], 1ull << (off & 63));
        }
        __builtin_amdgcn_wave_barrier();
        if (total != 0) {  // (four bricks in five are empty: they skip this)
            // exclusive prefix over the words of the start mask: rows that start before step w
            uint32_t pc = (uint32_t)__popcll(s_start_all[wave][lane]), pin = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(pin, off, 64);
                if (lane >= off) pin += o;
            }
            s_rb_all[wave][lane] = pin - pc;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // first step of every brick in the workgroup's step sequence
        uint32_t acc = 0;
        for (int b = 0; b < NB; ++b) { s_first[b] = acc; acc += (s_total[b] + 63) >> 6; }
        s_first[NB] = acc;
    }
    __syncthreads();
    const uint32_t GS = s_first[NB];

    // ---- phase 1: steps g = wave, wave + 4, ... of the workgroup's sequence
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted);
    const float rgx = radius * (float)G.dx + 1e-3f, rgy = radius * (float)G.dy + 1e-3f, rgz = radius * (float)G.dz + 1e-3f;
    float (*s_val)[64] = s_val_all[wave];
    unsigned long long* s_mask = s_mask_all[wave];
    for (uint32_t g = (uint32_t)wave; g < GS; g += NB) {
        int b = 0;
#pragma unroll
        for (int q = 1; q < NB; ++q) b += (g >= s_first[q]) ? 1 : 0;
        const uint32_t w = g - s_first[b];
        const uint32_t total = s_total[b];
        int x0 = 0, y0 = 0, z0 = 0;
        brick_origin(b, x0, y0, z0);
        const int x1 = min(x0 + kGW - 1, G.dx - 1), y1 = min(y0 + kGW - 1, G.dy - 1), z1 = min(z0 + kGW - 1, G.dz - 1);
        const uint32_t i = w * 64u + lane;
        const bool have = i < total;
        uint32_t j;
        if (total <= (uint32_t)MAXWORDS * 64u) {
            const unsigned long long sb = s_start_all[b][w];
            const int kk = max((int)s_rb_all[b][w] + (int)__popcll(sb & ((2ull << lane) - 1ull)) - 1, 0);  // (int): unsigned + int picks max(double, double)
            j = s_rowjb_all[b][kk] + i;
        } else {
            int lo = 0, hi = (int)s_nne[b] - 1;
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_rowoff_all[b][mid] <= i) lo = mid; else hi = mid - 1;
            }
            j = s_rowjb_all[b][lo] + i;
        }
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a;
        if (have) { a = rec[STRIDE * (size_t)j]; if (CH == 4) a2 = rec[2 * (size_t)j + 1]; }
        gather_record_phase<MAXC, CH>(G, a, a2.x, a2.y, have, x0, y0, z0, x1, y1, z1, rgx, rgy, rgz, radius, r2max, k, lane, s_val,
                                      s_mask);
        const int my_par = voxel_slot<MAXC>(x0 + (lane & 3), y0 + ((lane >> 2) & 3), z0 + (lane >> 4));
        // ---- the drain of step g, in turn
        while (__hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != g) __builtin_amdgcn_s_sleep(1);
        unsigned long long m = s_mask[lane];
        if (m) {
            s_mask[lane] = 0ull;
            float sum = s_sum_all[b][0][lane], sumg = CH == 4 ? s_sum_all[b][CH3 - 2][lane] : 0.f, sumb = CH == 4 ? s_sum_all[b][CH3 - 1][lane] : 0.f;
            gather_drain<NC, CH>(m, s_val, my_par, sum, sumg, sumb);
            s_sum_all[b][0][lane] = sum;
            if (CH == 4) { s_sum_all[b][CH3 - 2][lane] = sumg; s_sum_all[b][CH3 - 1][lane] = sumb; }
        }
        if (lane == 0) __hip_atomic_store(&s_turn, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();

    // ---- phase 2: wave b writes brick b
    int x0 = 0, y0 = 0, z0 = 0;
    if (!brick_origin(wave, x0, y0, z0)) return;
    const int x = x0 + (lane & 3), y = y0 + ((lane >> 2) & 3), z = z0 + (lane >> 4);
    if (dbg && lane == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
        const size_t gbs = (size_t)(x0 / kGW) + (size_t)bxn * ((size_t)(y0 / kGW) + (size_t)byn * (size_t)(z0 / kGW));
        dbg[4 * gbs + 0] = t_start;
        dbg[4 * gbs + 1] = __builtin_amdgcn_s_memrealtime();
        dbg[4 * gbs + 2] = s_total[wave];
        dbg[4 * gbs + 3] = xcc;
    }
    if (x < G.dx && y < G.dy && z < G.dz) {
        const float sum = s_sum_all[wave][0][lane];
        const uint32_t v = (uint32_t)x + (uint32_t)G.dx * ((uint32_t)y + (uint32_t)G.dy * (uint32_t)z);
        if (CH == 1) {
            out[v] = accumulate ? out[v] + sum : sum;
        } else {
            const float sumg = s_sum_all[wave][CH3 - 2][lane], sumb = s_sum_all[wave][CH3 - 1][lane];
            float4* o = reinterpret_cast<float4*>(out) + v;
            if (accumulate) { float4 tt = *o; *o = make_float4(tt.x + sum, tt.y + sumg, tt.z + sumb, tt.w); }
            else *o = make_float4(sum, sumg, sumb, 0.f);
        }
    }
}

// splat boxes of the selected photons (all interactions) -> the 4x4x4 voxel bricks they overlap
__global__ __launch_bounds__(256) void mark_bricks_kernel(const float* __restrict__ photons, RecLayout R, const uint32_t* __restrict__ indices,
                                                          int n_indices, int n_photons, int n_interactions, GridDev G,
                                                          float radius, int bxn, int byn, uint8_t* __restrict__ mask) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_indices * n_interactions) return;
    const size_t id = (size_t)indices[j % n_indices] + (size_t)(j / n_indices) * (size_t)n_photons;
    const float4 a = *rec_at(photons, R, id);
    if (a.x == kFltMax || a.y == kFltMax || a.z == kFltMax) return;
    const f3 p = { a.x, a.y, a.z };
    const Box3 bb = splat_box(G, p, radius);
    if (bb.ex <= bb.sx || bb.ey <= bb.sy || bb.ez <= bb.sz) return;
    for (int bz = bb.sz >> 2; bz <= (bb.ez - 1) >> 2; ++bz)
        for (int by = bb.sy >> 2; by <= (bb.ey - 1) >> 2; ++by)
            for (int bx = bb.sx >> 2; bx <= (bb.ex - 1) >> 2; ++bx)
                mask[(size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz)] = 1;  // same value from every writer
}

CPM_DEV void mark_bricks_of(const GridDev& G, f3 p, float radius, int bxn, int byn, uint8_t* __restrict__ mask) {
    if (p.x == kFltMax || p.y == kFltMax || p.z == kFltMax) return;
    const Box3 bb = splat_box(G, p, radius);
    if (bb.ex <= bb.sx || bb.ey <= bb.sy || bb.ez <= bb.sz) return;
    for (int bz = bb.sz >> 2; bz <= (bb.ez - 1) >> 2; ++bz)
        for (int by = bb.sy >> 2; by <= (bb.ey - 1) >> 2; ++by)
            for (int bx = bb.sx >> 2; bx <= (bb.ex - 1) >> 2; ++bx)
                mask[(size_t)bx + (size_t)bxn * ((size_t)by + (size_t)byn * (size_t)bz)] = 1;
}

// splatSelectedPhotonsToLightVolumeKernel twice (photonstolightvolume.cl:168-202 as called at
// processor/photontolightvolumeprocessorcl.cpp:268-274) in one launch over a device-side count.  Six lanes per selected
// photon j: (remove the record it had before its re-trace, old_photons[k * old_stride + j] | add the one it has now) x the
// z slices sz + {0, 1, 2}, sz + 3 + ..., of the splat box (x `yparts` groups of its rows where the box is wide: the workspace's
// 7 x 7 x 3 voxels) -- a few thousand photons make a latency-bound launch, and a lane's chain of dependent sqrt / divide / atomic
// steps (up to 27; 49 for a wide box before its rows were split) is what it lasts.
__global__ __launch_bounds__(256) void splat_delta_kernel(const float* __restrict__ old_photons, uint32_t old_stride, RecLayout RO,
                                                          const float* __restrict__ photons, RecLayout R, const uint32_t* __restrict__ indices,
                                                          const int32_t* __restrict__ n_dev, int max_n, int apply_below, GridDev G,
                                                          float radius, float k, int n_photons, int n_interactions, int bxn, int byn,
                                                          uint8_t* __restrict__ mask, float* __restrict__ out, int yparts) {
    const int n = min(*n_dev, max_n);
    if (apply_below > 0 && n >= apply_below) return;
    const int parts = 6 * yparts;
    // (the launch is sized for a few thousand photons, not for the budget: a grid over 6 x max_n lanes is thousands of
    // workgroups that read the count and leave -- 4 us of dispatch at 1 M photons; larger counts stride)
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)parts * n; t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t / parts), part = (int)(t - (long long)parts * j);
    const bool add = part & 1;
    const int zpart = (part >> 1) % 3, ypart = (part >> 1) / 3;
    const size_t id = indices[j];
    for (int it = 0; it < n_interactions; ++it) {
        // old_stride == 0: the old records sit at the photons' own indices (cpm_photon_importance_retrace's old_photons8)
        // (RO: the compact copy of cpm_trace_selected keeps float8 records; the old records at the photons' own indices lie as the photons do)
        const float4* qo = rec_at(old_photons, RO, old_stride ? (size_t)it * old_stride + (size_t)j : (size_t)it * n_photons + id);
        const float4* qn = rec_at(photons, R, (size_t)it * n_photons + id);
        const float4 oa = qo[0], ob = qo[RO.b], na = qn[0], nb = qn[R.b];
        // the same record before and after: its two splats cancel term by term
        if (__float_as_uint(oa.x) == __float_as_uint(na.x) && __float_as_uint(oa.y) == __float_as_uint(na.y) && __float_as_uint(oa.z) == __float_as_uint(na.z) &&
            __float_as_uint(oa.w) == __float_as_uint(na.w) && __float_as_uint(ob.x) == __float_as_uint(nb.x) && __float_as_uint(ob.y) == __float_as_uint(nb.y))
            continue;
        const float4 a = add ? na : oa, b = add ? nb : ob;
        const float m = add ? 1.f : -1.f;
        f3 p = { a.x, a.y, a.z };
        f3 pw = { a.w * k, b.x * k, b.y * k };
        pw.x *= m; pw.y *= m; pw.z *= m;
        splat_photon(out, G, p, pw, radius, zpart, 3, ypart, yparts);
        if (mask && zpart == 0 && ypart == 0) mark_bricks_of(G, p, radius, bxn, byn, mask);
    }
    }
}

int key_bits_for(uint32_t max_key) {  // bits needed to represent max_key
    int b = 1;
    while (b < 32 && (max_key >> b) != 0) ++b;
    return b;
}

}  // namespace

extern "C" {

// diagnostic hook (include/cpm/cpm_profile.h): 4 x u64 per 4x4x4 brick = (start, end [100 MHz ticks], records, XCC id)
void cpm_debug_set_gather_stamps(cpm_ctx* ctx, unsigned long long* dev) { if (ctx) ctx->dbg.gather_stamps = dev; }
// test hook: 1 = voxel-major kernel for every gather, 2 = generic record-major kernel instead of the r < 1 specialisation
// test / measurement hook: 1 (default) = the last sort pass finalises the bin, 0 = separate bin_finalize_kernel
void cpm_debug_set_bin_fused(cpm_ctx* ctx, int on) { if (ctx) ctx->dbg.bin_fused = on; }
// 1 (default): by launch size; 0: always one wave per brick; 2 / 4 / 8: always that many waves sharing as many bricks
void cpm_debug_set_gather_coop(cpm_ctx* ctx, int on) { if (ctx) ctx->dbg.gather_coop = on; }
void cpm_debug_force_voxel_gather(cpm_ctx* ctx, int on) { if (ctx) ctx->dbg.gather_force_voxel = on; }

int cpm_splat(cpm_ctx* ctx, const float* photons8, int total_photons, const cpm_grid_desc* grid, float radius,
              float scale, float* grid_out, cpm_stream stream) {
    return cpm_splat_records(ctx, photons8, total_photons, total_photons, grid, radius, scale, grid_out, stream);
}

int cpm_splat_records(cpm_ctx* ctx, const float* photons8, int n_records, int total_photons, const cpm_grid_desc* grid, float radius,
                      float scale, float* grid_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_records >= total_photons, "cpm_splat_records: fewer records than photons");
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, total_photons >= 0 && radius > 0.f, "cpm_splat: bad size or radius");
    if (total_photons == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && grid_out, "cpm_splat: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_splat");
    float k = kInv4Pi * scale;
    CPM_LAUNCH(ctx, splat_kernel, dim3(div_up(total_photons, 256)), dim3(256), 0, (hipStream_t)stream, photons8, rec_layout(ctx, photons8, (size_t)n_records),
                       total_photons, G, radius, k, grid_out);
    CPM_LAUNCH_CHECK(ctx, "splat_kernel");
    return CPM_OK;
}

int cpm_splat_selected(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                       const cpm_grid_desc* grid, float radius, float scale, float multiplier, int n_photons,
                       int n_interactions, float* grid_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n_indices >= 0 && n_photons >= 0 && n_interactions >= 1 && radius > 0.f, "cpm_splat_selected: bad size");
    if (n_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && indices && grid_out, "cpm_splat_selected: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_splat_selected");
    float k = kInv4Pi * scale;
    CPM_LAUNCH(ctx, splat_selected_kernel, dim3(div_up(n_indices, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
                       rec_layout(ctx, photons8, (size_t)n_photons * n_interactions), indices, n_indices, G, radius, k, multiplier, n_photons, n_interactions, grid_out);
    CPM_LAUNCH_CHECK(ctx, "splat_selected_kernel");
    return CPM_OK;
}

int cpm_splat_delta(cpm_ctx* ctx, const float* old_photons8, int old_stride, const float* photons8, const uint32_t* indices,
                    const int32_t* n_indices_dev, int max_indices, int apply_below, const cpm_grid_desc* grid, float radius,
                    float scale, int n_photons, int n_interactions, uint8_t* brick_mask, float* grid_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, max_indices >= 0 && (old_stride == 0 || old_stride >= max_indices) && n_photons >= 0 && n_interactions >= 1 && radius > 0.f,
                "cpm_splat_delta: bad size");
    CPM_REQUIRE(ctx, max_indices < (1 << 28), "cpm_splat_delta: too many indices");
    if (max_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, old_photons8 && photons8 && indices && n_indices_dev && grid_out, "cpm_splat_delta: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_splat_delta");
    CPM_REQUIRE_ALIGNED16(ctx, old_photons8, "cpm_splat_delta");
    const float k = kInv4Pi * scale;
    // a box more than 4 rows high: its rows in 4 groups (a lane's chain of dependent splat steps is what the launch lasts)
    const int yparts = 2.f * radius * G.t2i.sy + 1.f > 4.f ? 4 : 1;
    const long long wgs = div_up(6ll * yparts * max_indices, 256);
    const RecLayout R = rec_layout(ctx, photons8, (size_t)n_photons * n_interactions);
    CPM_LAUNCH(ctx, splat_delta_kernel, dim3((unsigned)(wgs < 2048 ? wgs : 2048)), dim3(256), 0, (hipStream_t)stream, old_photons8,
               (uint32_t)old_stride, old_stride ? rec_interleaved() : R, photons8, R, indices, n_indices_dev, max_indices, apply_below, G, radius, k, n_photons, n_interactions,
               div_up(G.dx, 4), div_up(G.dy, 4), brick_mask, grid_out, yparts);
    CPM_LAUNCH_CHECK(ctx, "splat_delta_kernel");
    return CPM_OK;
}

int cpm_copy_indexed_photons(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                             float multiplier, int n_photons, int n_interactions, float* aligned8, int out_offset,
                             cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_indices >= 0 && n_photons >= 0 && n_interactions >= 1 && out_offset >= 0, "cpm_copy_indexed_photons: bad size");
    if (n_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && indices && aligned8, "cpm_copy_indexed_photons: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_copy_indexed_photons");
    CPM_REQUIRE_ALIGNED16(ctx, aligned8, "cpm_copy_indexed_photons");
    CPM_LAUNCH(ctx, copy_indexed_kernel, dim3(div_up(n_indices, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
                       rec_layout(ctx, photons8, (size_t)n_photons * n_interactions), indices, n_indices, multiplier, n_photons, n_interactions, aligned8, out_offset);
    CPM_LAUNCH_CHECK(ctx, "copy_indexed_kernel");
    return CPM_OK;
}

int cpm_snapshot_selected_photons(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices, int n_photons,
                                  int n_interactions, float* snapshot8, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_indices >= 0 && n_photons >= 0 && n_interactions >= 1, "cpm_snapshot_selected_photons: bad sizes");
    CPM_REQUIRE(ctx, (long long)n_indices * n_interactions < (1ll << 31), "cpm_snapshot_selected_photons: too many photons");
    if (n_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && indices && snapshot8, "cpm_snapshot_selected_photons: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_snapshot_selected_photons");
    CPM_REQUIRE_ALIGNED16(ctx, snapshot8, "cpm_snapshot_selected_photons");
    const int threads = n_indices * n_interactions;
    CPM_LAUNCH(ctx, snapshot_selected_kernel, dim3((unsigned)div_up(threads, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
               rec_layout(ctx, photons8, (size_t)n_photons * n_interactions), indices,
               n_indices, n_photons, n_interactions, snapshot8);
    CPM_LAUNCH_CHECK(ctx, "snapshot_selected_kernel");
    return CPM_OK;
}

int cpm_bin(cpm_ctx* ctx, const float* photons8, int n, const cpm_grid_desc* grid, uint32_t* order,
            uint32_t* cell_start, float* sorted_pos_power, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE_DEFAULT_MATRICES(ctx, G, "cpm_bin");
    CPM_REQUIRE(ctx, n >= 0, "cpm_bin: n < 0");
    CPM_REQUIRE(ctx, cell_start, "cpm_bin: null cell_start");
    CPM_REQUIRE(ctx, n == 0 || (photons8 && order && sorted_pos_power), "cpm_bin: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_bin");
    CPM_REQUIRE_ALIGNED16(ctx, sorted_pos_power, "cpm_bin");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t cells = (uint32_t)G.dx * G.dy * G.dz;
    uint32_t* keys = (uint32_t*)scratch(ctx, CPM_SCR_BIN_KEYS, (size_t)(n > 0 ? n : 1) * 8);
    if (!keys) return CPM_ERR_OUT_OF_MEMORY;
    uint32_t* vals = keys + (n > 0 ? n : 1);
    bool finalized = false;
    if (n > 0) {
        const int kb = key_bits_for(cells);
        const int items = cpm::sort_items_for(ctx, (size_t)n);
        uint32_t num_tiles = 0;
        uint32_t* hist = ctx->dbg.bin_fused ? cpm::sort_first_hist(ctx, (size_t)n, kb, &num_tiles) : nullptr;
        const dim3 kgrid((unsigned)div_up(n, 256 * items));
        switch (items) {
            case 4: CPM_LAUNCH(ctx, bin_keys_kernel<4>, kgrid, dim3(256), 0, s, photons8, rec_layout(ctx, photons8, (size_t)n), n, G, cells, keys, vals, cell_start, hist, num_tiles); break;
            case 8: CPM_LAUNCH(ctx, bin_keys_kernel<8>, kgrid, dim3(256), 0, s, photons8, rec_layout(ctx, photons8, (size_t)n), n, G, cells, keys, vals, cell_start, hist, num_tiles); break;
            default: CPM_LAUNCH(ctx, bin_keys_kernel<16>, kgrid, dim3(256), 0, s, photons8, rec_layout(ctx, photons8, (size_t)n), n, G, cells, keys, vals, cell_start, hist, num_tiles); break;
        }
        CPM_LAUNCH_CHECK(ctx, "bin_keys_kernel");
        // no copy-back after an odd number of passes: the cell-start kernel reads the keys wherever the
        // ping-pong left them; the last scatter pass writes order / records / run starts itself (BinSink)
        BinSink sink;
        sink.photons = photons8; sink.rec = rec_layout(ctx, photons8, (size_t)n); sink.channels = G.channels; sink.order = order; sink.sorted = sorted_pos_power;
        sink.cell_start = cell_start;
        rc = cpm::radix_sort(ctx, keys, vals, (size_t)n, kb, s, &keys, &vals, ctx->dbg.bin_fused ? &sink : nullptr, &finalized, hist != nullptr);
        if (rc) return rc;
    }
    // run starts -> table (preset to "none" by bin_keys_kernel), then the suffix-min scan turns it into cell starts
    if (n == 0) CPM_HIP_CHECK(ctx, hipMemsetAsync(cell_start, 0xff, ((size_t)cells + 1) * sizeof(uint32_t), s));
    if (n > 0 && !finalized) {  // n == 1, the onesweep test mode, or cpm_debug_set_bin_fused(0)
        CPM_LAUNCH(ctx, bin_finalize_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, photons8, rec_layout(ctx, photons8, (size_t)n), keys, vals, n, G.channels,
                           order, sorted_pos_power, cell_start);
        CPM_LAUNCH_CHECK(ctx, "bin_finalize_kernel");
    }
    CPM_LAUNCH(ctx, cell_start_kernel, dim3(div_up((long long)cells + 1, kCsTile)), dim3(256), 0, s, keys, (uint32_t)n,
                       cells, cell_start);
    CPM_LAUNCH_CHECK(ctx, "cell_start_kernel");
    return CPM_OK;
}

static int gather_impl(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
                       const cpm_grid_desc* grid, float radius, float scale, int accumulate, float* grid_out,
                       const uint8_t* brick_mask, cpm_stream stream);

int cpm_gather(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
               const cpm_grid_desc* grid, float radius, float scale, int accumulate, float* grid_out,
               cpm_stream stream) {
    return gather_impl(ctx, sorted_pos_power, cell_start, n, grid, radius, scale, accumulate, grid_out, nullptr, stream);
}

int cpm_gather_bricks(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
                      const cpm_grid_desc* grid, float radius, float scale, const uint8_t* brick_mask, float* grid_out,
                      cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, brick_mask, "cpm_gather_bricks: null mask");
    return gather_impl(ctx, sorted_pos_power, cell_start, n, grid, radius, scale, 0, grid_out, brick_mask, stream);
}

int cpm_mark_touched_bricks(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices, int n_photons,
                            int n_interactions, const cpm_grid_desc* grid, float radius, uint8_t* brick_mask, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n_indices >= 0 && n_photons >= 0 && n_interactions >= 1 && radius > 0.f, "cpm_mark_touched_bricks: bad size or radius");
    if (n_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && indices && brick_mask, "cpm_mark_touched_bricks: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_mark_touched_bricks");
    const int bxn = div_up(G.dx, 4), byn = div_up(G.dy, 4);
    const long long threads = (long long)n_indices * n_interactions;
    CPM_LAUNCH(ctx, mark_bricks_kernel, dim3((unsigned)div_up(threads, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
               rec_layout(ctx, photons8, (size_t)n_photons * n_interactions), indices,
               n_indices, n_photons, n_interactions, G, radius, bxn, byn, brick_mask);
    CPM_LAUNCH_CHECK(ctx, "mark_bricks_kernel");
    return CPM_OK;
}

static int gather_impl(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
                       const cpm_grid_desc* grid, float radius, float scale, int accumulate, float* grid_out,
                       const uint8_t* brick_mask, cpm_stream stream) {
    CPM_ENTER(ctx);
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE_DEFAULT_MATRICES(ctx, G, "cpm_gather");
    CPM_REQUIRE(ctx, n >= 0 && radius > 0.f, "cpm_gather: bad size or radius");
    CPM_REQUIRE(ctx, cell_start && grid_out && (sorted_pos_power || n == 0), "cpm_gather: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, sorted_pos_power, "cpm_gather");
    if (G.channels == 4) CPM_REQUIRE_ALIGNED16(ctx, grid_out, "cpm_gather");   // float4 stores
    const uint32_t cells = (uint32_t)G.dx * G.dy * G.dz;
    // cells whose photons can reach a voxel: |cell - voxel| <= floor(r * dim + 0.5) per axis,
    // with 1e-3 of slack for the fp32 rounding of the box / distance tests
    int Rx = (int)floorf(fmaf(radius, (float)G.dx, 0.501f));
    int Ry = (int)floorf(fmaf(radius, (float)G.dy, 0.501f));
    int Rz = (int)floorf(fmaf(radius, (float)G.dz, 0.501f));
    float k = kInv4Pi * scale;
    // cheap exact reject: d^2 > r^2 (1 + 1e-5)  =>  fl(fl(sqrt(d^2)) / r) > 1  =>  weight 0
    float r2max = (radius * radius) * 1.00001f;
    // any radius: the voxel-major kernel takes the halo widths at run time (the tuned record-major kernels need R <= 2)
    int bxn = div_up(G.dx, 4), byn = div_up(G.dy, 4), bzn = div_up(G.dz, 4);
    dim3 gridDim((unsigned)div_up((long long)bxn * byn * bzn, 4)), block(256);
    hipStream_t hs = (hipStream_t)stream;
    // candidates per axis a record can reach: floor(2 (r' + 1e-3)) + 1
    const float rmax = fmaxf(radius * (float)G.dx, fmaxf(radius * (float)G.dy, radius * (float)G.dz)) + 1e-3f;
    const int cand_axis = (int)floorf(2.f * rmax) + 1;
    // Sharing bricks between the waves of a workgroup pays when the launch is small enough for a few heavy bricks to
    // set its length (128^3 grid, 32 K bricks: 80 -> 61 us); with 8x the bricks the hardware's own wave scheduling
    // balances the load and the turn-taking only costs (256^3 grid: 203 us one wave per brick, 224 us shared).
    const bool coop = ctx->dbg.gather_coop > 1 || (ctx->dbg.gather_coop == 1 && (long long)bxn * byn * bzn <= 65536);
    // the tuned record-major kernels: one channel, halo of <= 2 cells, 2 to 4 candidate voxels per axis (r < 2 cells)
    // Measured on config 2's photons (128^3 grid): r = 0.87 / 1.0 / 1.2 / 1.45 / 1.73 cells -> tuned cooperative kernel
    // 64 / 105 / 118 / 147 / 390 us against 157 / 194 / 258 / 351 / 722 us for the better of the generic kernels.  At 4
    // candidates per axis (64 per record, ~20 survivors) the record-major form only wins where photons are dense:
    // 256^3 grid, 0.06 photons per cell: 742 us against 577 us voxel-major -- hence the density condition.
    const bool dense = (long long)n * 4 >= (long long)cells;
    // the tuned record-major kernels: halo of <= 2 cells, 2 to 4 candidate voxels per axis (r < 2 cells); 4-channel
    // light volumes up to 3 candidates (the slot planes of 4 x 4 x 4 x 3 would not fit a workgroup's LDS)
    // (mod3_ of the 3-candidate kernels is exact for coordinates below 2^15 only: longer axes take the generic kernels)
    const bool mod3_ok = cand_axis != 3 || (G.dx <= 32768 && G.dy <= 32768 && G.dz <= 32768);
    const bool tuned = (cand_axis <= 3 || (cand_axis == 4 && dense && G.channels == 1)) && Rx <= 2 && Ry <= 2 && Rz <= 2 &&
                       ctx->dbg.gather_force_voxel == 0 && mod3_ok;
#define CPM_COOP_LAUNCH(NB, MAXC, CH)                                                                                          \
    do {                                                                                                                       \
        const int zq = div_up(bzn, NB);                                                                                        \
        CPM_LAUNCH(ctx, (gather_coop_kernel<NB, MAXC, CH>), dim3((unsigned)bxn, (unsigned)byn, (unsigned)zq), dim3(64 * NB), 0, \
                   hs, sorted_pos_power, cell_start, G, radius, r2max, k, Rx, Ry, Rz, accumulate, bxn, byn, bzn, zq, grid_out, \
                   ctx->dbg.gather_stamps, brick_mask);                                                                               \
    } while (0)
#define CPM_REC2_LAUNCH(MAXC, CH)                                                                                              \
    CPM_LAUNCH(ctx, (gather_records2_kernel<MAXC, CH>), dim3((unsigned)div_up(bxn, 4), (unsigned)byn, (unsigned)bzn), block, 0, \
               hs, sorted_pos_power, cell_start, G, radius, r2max, k, Rx, Ry, Rz, accumulate, bxn, byn, bzn, grid_out,         \
               ctx->dbg.gather_stamps, brick_mask)
    if (tuned && coop) {
        if (G.channels == 4) {
            if (cand_axis <= 2) CPM_COOP_LAUNCH(4, 2, 4); else CPM_COOP_LAUNCH(4, 3, 4);
        } else if (cand_axis <= 2) {
            if (ctx->dbg.gather_coop == 2) CPM_COOP_LAUNCH(2, 2, 1);
            else if (ctx->dbg.gather_coop == 8) CPM_COOP_LAUNCH(8, 2, 1);
            else CPM_COOP_LAUNCH(4, 2, 1);
        } else if (cand_axis == 3) {
            CPM_COOP_LAUNCH(4, 3, 1);
        } else {
            CPM_COOP_LAUNCH(4, 4, 1);
        }
    } else if (tuned) {
        if (G.channels == 4) {
            if (cand_axis <= 2) CPM_REC2_LAUNCH(2, 4); else CPM_REC2_LAUNCH(3, 4);
        } else if (cand_axis <= 2) {
            CPM_REC2_LAUNCH(2, 1);
        } else if (cand_axis == 3) {
            CPM_REC2_LAUNCH(3, 1);
        } else {
            CPM_REC2_LAUNCH(4, 1);
        }
    }
#undef CPM_COOP_LAUNCH
#undef CPM_REC2_LAUNCH
    else if (G.channels == 1 && cand_axis <= 2 && ctx->dbg.gather_force_voxel != 1)
        CPM_LAUNCH(ctx, gather_records_kernel<2>, gridDim, block, 0, hs, sorted_pos_power, cell_start, G, radius, r2max, k, Rx, Ry, Rz,
                   accumulate, bxn, byn, grid_out, ctx->dbg.gather_stamps, brick_mask);
    else if (G.channels == 1 && cand_axis <= 3 && ctx->dbg.gather_force_voxel != 1)
        CPM_LAUNCH(ctx, gather_records_kernel<3>, gridDim, block, 0, hs, sorted_pos_power, cell_start, G, radius, r2max, k, Rx, Ry, Rz,
                   accumulate, bxn, byn, grid_out, ctx->dbg.gather_stamps, brick_mask);
    else if (G.channels == 1)
        CPM_LAUNCH(ctx, (gather_voxel_kernel<1, 4>), gridDim, block, 0, hs, sorted_pos_power, cell_start, G, radius, r2max, k, Rx, Ry, Rz,
                   accumulate, bxn, byn, grid_out, ctx->dbg.gather_stamps, brick_mask);
    else
        CPM_LAUNCH(ctx, (gather_voxel_kernel<4, 4>), gridDim, block, 0, hs, sorted_pos_power, cell_start, G, radius, r2max, k, Rx, Ry, Rz,
                   accumulate, bxn, byn, grid_out, ctx->dbg.gather_stamps, brick_mask);
    CPM_LAUNCH_CHECK(ctx, "gather_kernel");
    return CPM_OK;
}

}  // extern "C"
