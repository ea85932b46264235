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
int radix_sort(cpm_ctx* ctx, uint32_t* keys, uint32_t* vals, size_t n, int key_bits, hipStream_t s);
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

// splatPhoton (photonstolightvolume.cl:31-79) with the power already scaled
CPM_DEV void splat_photon(float* __restrict__ out, const GridDev& G, f3 p, f3 pw, float radius) {
    if (p.x == kFltMax || p.y == kFltMax || p.z == kFltMax) return;
    Box3 bx = splat_box(G, p, radius);
    for (int z = bx.sz; z < bx.ez; ++z)
        for (int y = bx.sy; y < bx.ey; ++y)
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
__global__ __launch_bounds__(256) void splat_kernel(const float* __restrict__ photons, int n, GridDev G, float radius,
                                                    float k, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* q = reinterpret_cast<const float4*>(photons) + 2 * (size_t)i;
    float4 a = q[0], b = q[1];
    f3 p = { a.x, a.y, a.z };
    f3 pw = { a.w * k, b.x * k, b.y * k };
    splat_photon(out, G, p, pw, radius);
}

// splatSelectedPhotonsToLightVolumeKernel (photonstolightvolume.cl:168-202)
__global__ __launch_bounds__(256) void splat_selected_kernel(const float* __restrict__ photons,
                                                             const uint32_t* __restrict__ indices, int n_indices,
                                                             GridDev G, float radius, float k, float multiplier,
                                                             int n_photons, int n_interactions,
                                                             float* __restrict__ out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_indices) return;
    size_t id = indices[j];
    for (int it = 0; it < n_interactions; ++it) {
        const float4* q = reinterpret_cast<const float4*>(photons) + 2 * ((size_t)it * n_photons + id);
        float4 a = q[0], b = q[1];
        f3 p = { a.x, a.y, a.z };
        f3 pw = { a.w * k, b.x * k, b.y * k };
        pw.x *= multiplier; pw.y *= multiplier; pw.z *= multiplier;
        splat_photon(out, G, p, pw, radius);
    }
}

// copyIndexPhotonsKernel (photonstolightvolume.cl:225-248)
__global__ __launch_bounds__(256) void copy_indexed_kernel(const float* __restrict__ photons,
                                                           const uint32_t* __restrict__ indices, int n_indices,
                                                           float multiplier, int n_photons, int n_interactions,
                                                           float* __restrict__ aligned, int out_offset) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_indices) return;
    size_t id = indices[j];
    for (int it = 0; it < n_interactions; ++it) {
        const float4* q = reinterpret_cast<const float4*>(photons) + 2 * ((size_t)it * n_photons + id);
        float4 a = q[0], b = q[1];
        float4* o = reinterpret_cast<float4*>(aligned) + 2 * ((size_t)out_offset + j + (size_t)it * n_indices);
        o[0] = make_float4(a.x, a.y, a.z, a.w * multiplier);
        o[1] = make_float4(b.x * multiplier, b.y * multiplier, b.z, b.w);
    }
}

// ---- bin

// cell key (template: cl/hashlightsample.cl:55-64); sentinels get key == cells so that the
// sort needs only bits(cells) key bits and they still land behind every real cell
__global__ __launch_bounds__(256) void bin_keys_kernel(const float* __restrict__ photons, int n, GridDev G,
                                                       uint32_t cells, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = reinterpret_cast<const float4*>(photons)[2 * (size_t)i];
    uint32_t key = cells;
    if (!(a.x == kFltMax || a.y == kFltMax || a.z == kFltMax)) {
        float fx = min_(max_(__builtin_floorf(a.x * (float)G.dx), 0.0f), (float)(G.dx - 1));
        float fy = min_(max_(__builtin_floorf(a.y * (float)G.dy), 0.0f), (float)(G.dy - 1));
        float fz = min_(max_(__builtin_floorf(a.z * (float)G.dz), 0.0f), (float)(G.dz - 1));
        key = (uint32_t)(int)fx + (uint32_t)G.dx * ((uint32_t)(int)fy + (uint32_t)G.dy * (uint32_t)(int)fz);
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

// order[j] = sorted photon index; compact (pos, power) records in cell order
__global__ __launch_bounds__(256) void bin_finalize_kernel(const float* __restrict__ photons,
                                                           const uint32_t* __restrict__ sorted_vals, int n,
                                                           int channels, uint32_t* __restrict__ order,
                                                           float* __restrict__ sorted) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t id = sorted_vals[j];
    order[j] = id;
    const float4* q = reinterpret_cast<const float4*>(photons) + 2 * (size_t)id;
    float4 a = q[0];
    if (channels == 1) {
        reinterpret_cast<float4*>(sorted)[j] = a;
    } else {
        float4 b = q[1];
        float4* o = reinterpret_cast<float4*>(sorted) + 2 * (size_t)j;
        o[0] = a;
        o[1] = make_float4(b.x, b.y, 0.f, 0.f);
    }
}

// cell_start[c] = first j with key[j] >= c, c = 0..cells (binary search in the sorted keys)
__global__ __launch_bounds__(256) void cell_start_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t cells,
                                                         uint32_t* __restrict__ cell_start) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > cells) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < c) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = lo;
}

// ---- gather

template <int CH>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ sorted,
                                                     const uint32_t* __restrict__ cell_start, GridDev G, float radius,
                                                     float k, int Rx, int Ry, int Rz, int accumulate,
                                                     float* __restrict__ out) {
    const uint32_t cells = (uint32_t)G.dx * G.dy * G.dz;
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= cells) return;
    int x = (int)(v % (uint32_t)G.dx);
    int y = (int)((v / (uint32_t)G.dx) % (uint32_t)G.dy);
    int z = (int)(v / ((uint32_t)G.dx * (uint32_t)G.dy));
    f3 vi = { (float)x, (float)y, (float)z };
    f3 c = transform_(G.i2t, vi);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    int xlo = max(x - Rx, 0), xhi = min(x + Rx, G.dx - 1);
    for (int cz = z - Rz; cz <= z + Rz; ++cz) {
        if (cz < 0 || cz >= G.dz) continue;
        for (int cy = y - Ry; cy <= y + Ry; ++cy) {
            if (cy < 0 || cy >= G.dy) continue;
            uint32_t row = (uint32_t)G.dx * ((uint32_t)cy + (uint32_t)G.dy * (uint32_t)cz);
            uint32_t jb = cell_start[row + xlo], je = cell_start[row + xhi + 1];
            for (uint32_t j = jb; j < je; ++j) {
                float4 a = reinterpret_cast<const float4*>(sorted)[(CH == 1 ? 1 : 2) * (size_t)j];
                f3 p = { a.x, a.y, a.z };
                Box3 bx = splat_box(G, p, radius);
                if (x < bx.sx || x >= bx.ex || y < bx.sy || y >= bx.ey || z < bx.sz || z >= bx.ez) continue;
                float w = splat_weight(c, p, radius);
                float vr = (a.w * k) * w;
                if (vr != 0.f) sr += vr;
                if (CH == 4) {
                    float4 b = reinterpret_cast<const float4*>(sorted)[2 * (size_t)j + 1];
                    float vg = (b.x * k) * w, vb = (b.y * k) * w;
                    if (vg != 0.f) sg += vg;
                    if (vb != 0.f) sb += vb;
                }
            }
        }
    }
    if (CH == 1) {
        out[v] = accumulate ? out[v] + sr : sr;
    } else {
        float4* o = reinterpret_cast<float4*>(out) + v;
        if (accumulate) { float4 t = *o; *o = make_float4(t.x + sr, t.y + sg, t.z + sb, t.w); }
        else *o = make_float4(sr, sg, sb, 0.f);
    }
}

int key_bits_for(uint32_t max_key) {  // bits needed to represent max_key
    int b = 1;
    while (b < 32 && (max_key >> b) != 0) ++b;
    return b;
}

}  // namespace

extern "C" {

int cpm_splat(cpm_ctx* ctx, const float* photons8, int total_photons, const cpm_grid_desc* grid, float radius,
              float scale, float* grid_out, cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, total_photons >= 0 && radius > 0.f, "cpm_splat: bad size or radius");
    if (total_photons == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && grid_out, "cpm_splat: null buffer");
    float k = kInv4Pi * scale;
    CPM_LAUNCH(ctx, splat_kernel, dim3(div_up(total_photons, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
                       total_photons, G, radius, k, grid_out);
    CPM_LAUNCH_CHECK(ctx, "splat_kernel");
    return CPM_OK;
}

int cpm_splat_selected(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                       const cpm_grid_desc* grid, float radius, float scale, float multiplier, int n_photons,
                       int n_interactions, float* grid_out, cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n_indices >= 0 && n_photons >= 0 && n_interactions >= 1 && radius > 0.f, "cpm_splat_selected: bad size");
    if (n_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && indices && grid_out, "cpm_splat_selected: null buffer");
    float k = kInv4Pi * scale;
    CPM_LAUNCH(ctx, splat_selected_kernel, dim3(div_up(n_indices, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
                       indices, n_indices, G, radius, k, multiplier, n_photons, n_interactions, grid_out);
    CPM_LAUNCH_CHECK(ctx, "splat_selected_kernel");
    return CPM_OK;
}

int cpm_copy_indexed_photons(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                             float multiplier, int n_photons, int n_interactions, float* aligned8, int out_offset,
                             cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    CPM_REQUIRE(ctx, n_indices >= 0 && n_photons >= 0 && n_interactions >= 1 && out_offset >= 0, "cpm_copy_indexed_photons: bad size");
    if (n_indices == 0) return CPM_OK;
    CPM_REQUIRE(ctx, photons8 && indices && aligned8, "cpm_copy_indexed_photons: null buffer");
    CPM_LAUNCH(ctx, copy_indexed_kernel, dim3(div_up(n_indices, 256)), dim3(256), 0, (hipStream_t)stream, photons8,
                       indices, n_indices, multiplier, n_photons, n_interactions, aligned8, out_offset);
    CPM_LAUNCH_CHECK(ctx, "copy_indexed_kernel");
    return CPM_OK;
}

int cpm_bin(cpm_ctx* ctx, const float* photons8, int n, const cpm_grid_desc* grid, uint32_t* order,
            uint32_t* cell_start, float* sorted_pos_power, cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n >= 0, "cpm_bin: n < 0");
    CPM_REQUIRE(ctx, cell_start, "cpm_bin: null cell_start");
    CPM_REQUIRE(ctx, n == 0 || (photons8 && order && sorted_pos_power), "cpm_bin: null buffer");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t cells = (uint32_t)G.dx * G.dy * G.dz;
    uint32_t* keys = (uint32_t*)scratch(ctx, CPM_SCR_BIN_KEYS, (size_t)(n > 0 ? n : 1) * 8);
    if (!keys) return CPM_ERR_OUT_OF_MEMORY;
    uint32_t* vals = keys + (n > 0 ? n : 1);
    if (n > 0) {
        CPM_LAUNCH(ctx, bin_keys_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, photons8, n, G, cells, keys, vals);
        CPM_LAUNCH_CHECK(ctx, "bin_keys_kernel");
        rc = cpm::radix_sort(ctx, keys, vals, (size_t)n, key_bits_for(cells), s);
        if (rc) return rc;
        CPM_LAUNCH(ctx, bin_finalize_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, photons8, vals, n, G.channels,
                           order, sorted_pos_power);
        CPM_LAUNCH_CHECK(ctx, "bin_finalize_kernel");
    }
    CPM_LAUNCH(ctx, cell_start_kernel, dim3(div_up((long long)cells + 1, 256)), dim3(256), 0, s, keys, (uint32_t)n,
                       cells, cell_start);
    CPM_LAUNCH_CHECK(ctx, "cell_start_kernel");
    return CPM_OK;
}

int cpm_gather(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
               const cpm_grid_desc* grid, float radius, float scale, int accumulate, float* grid_out,
               cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    GridDev G;
    int rc = make_grid_dev(ctx, grid, G);
    if (rc) return rc;
    CPM_REQUIRE(ctx, n >= 0 && radius > 0.f, "cpm_gather: bad size or radius");
    CPM_REQUIRE(ctx, cell_start && grid_out && (sorted_pos_power || n == 0), "cpm_gather: null buffer");
    const uint32_t cells = (uint32_t)G.dx * G.dy * G.dz;
    // cells whose photons can reach a voxel: |cell - voxel| <= floor(r * dim + 0.5) per axis,
    // with 1e-3 of slack for the fp32 rounding of the box / distance tests
    int Rx = (int)floorf(fmaf(radius, (float)G.dx, 0.501f));
    int Ry = (int)floorf(fmaf(radius, (float)G.dy, 0.501f));
    int Rz = (int)floorf(fmaf(radius, (float)G.dz, 0.501f));
    float k = kInv4Pi * scale;
    dim3 gridDim(div_up(cells, 256)), block(256);
    if (G.channels == 1)
        CPM_LAUNCH(ctx, gather_kernel<1>, gridDim, block, 0, (hipStream_t)stream, sorted_pos_power, cell_start, G,
                           radius, k, Rx, Ry, Rz, accumulate, grid_out);
    else
        CPM_LAUNCH(ctx, gather_kernel<4>, gridDim, block, 0, (hipStream_t)stream, sorted_pos_power, cell_start, G,
                           radius, k, Rx, Ry, Rz, accumulate, grid_out);
    CPM_LAUNCH_CHECK(ctx, "gather_kernel");
    return CPM_OK;
}

}  // extern "C"
