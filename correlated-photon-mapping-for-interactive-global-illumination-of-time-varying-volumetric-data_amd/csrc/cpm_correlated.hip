// cpm_correlated.hip -- helpers of the correlated re-trace (C1-C6, S2-S4): min/max bricks,
// temporal difference bricks, TF-difference importance per brick, per-photon importance by
// DDA through the importance grid, and the fused threshold + count + iota + sort selection.
#include "cpm_trace_body.hip.h"

#include <chrono>
#include <new>

using namespace cpm;

// cpm_selection (include/cpm/cpm.h): the state of one changed-photon selection
struct cpm_selection {
    size_t max_photons = 0;
    uint32_t per_tile = 0, max_tiles = 0;
    uint2* tile = nullptr;            // device, max_tiles
    uint32_t* local = nullptr;        // device, max_photons
    int32_t* count_dev = nullptr;     // device
    uint32_t* mask = nullptr;         // device, grow-only: occupancy bits of the importance grid of the last select call
    size_t mask_words = 0;
    const uint32_t* given_mask = nullptr;  // cpm_selection_set_occupancy: the caller's bits (cpm_importance_tf_occupancy) ...
    const float* given_mask_grid = nullptr;  // ... of this importance grid
    unsigned long long* mailbox = nullptr;      // pinned host memory, written by selection_compact_kernel
    unsigned long long* mailbox_dev = nullptr;  // its device address
    uint32_t n_tiles = 0;             // tiles appended since cpm_selection_begin
    // cpm_photon_importance_retrace: per launch since cpm_selection_begin (one per light) the order in which its workgroups
    // take the tiles -- costliest first, from the wall-clock the tiles took in the last MEASURED launch (see kRetraceTile)
    struct LaunchOrder { uint32_t n_tiles = 0; uint32_t* order = nullptr; uint32_t* cost = nullptr; uint32_t* keys = nullptr; bool fresh = true; /* no order yet */ };
    std::vector<LaunchOrder> orders;
    std::vector<uint32_t> pending_orders;  // launches measured in this selection: their orders are re-sorted by cpm_selection_finish
    uint32_t n_launches = 0;          // retrace launches since cpm_selection_begin
    uint32_t selections = 0;          // cpm_selection_begin calls
    bool measuring = false;           // this selection's retrace launches record what their tiles cost
    uint32_t epoch = 0;               // of the last cpm_selection_finish enqueued
    bool finished = false;            // a finish has been enqueued since begin
    bool failed = false;              // a select / retrace call since begin failed after its tiles were appended: the finish publishes 0
    hipStream_t last_stream = nullptr;
};

namespace {

struct BrickVol {
    const void* voxels;
    int dx, dy, dz, dtype;
    int ox, oy, oz, region;
    float norm, offset, one_minus_scaling;
};

CPM_DEV float raw_voxel(const void* v, int dtype, size_t idx) {
    if (dtype == CPM_U8) return (float)static_cast<const uint8_t*>(v)[idx];
    if (dtype == CPM_U16) return (float)static_cast<const uint16_t*>(v)[idx];
    return static_cast<const float*>(v)[idx];
}

// volumeMinMaxKernel (ref uniformgridcl/cl/uniformgrid/volumeminmax.cl:33-61).  One wave per
// brick: lane = one (y, z) row of the brick, contiguous in x; min/max are order-free, so the
// wave reduction gives the reference's values exactly.
__global__ __launch_bounds__(64) void minmax_kernel(BrickVol V, uint16_t* __restrict__ out) {
    const int brick = blockIdx.x;
    const int gx = brick % V.ox, gy = (brick / V.ox) % V.oy, gz = brick / (V.ox * V.oy);
    const int x0 = gx * V.region, y0 = gy * V.region, z0 = gz * V.region;
    const int ex = min(x0 + V.region, V.dx), ey = min(y0 + V.region, V.dy), ez = min(z0 + V.region, V.dz);
    float mn = kFltMax, mx = 0.f;
    const int rows = V.region * V.region;
    for (int r = threadIdx.x; r < rows; r += 64) {
        int y = y0 + r % V.region, z = z0 + r / V.region;
        if (y >= ey || z >= ez) continue;
        size_t base = (size_t)V.dx * ((size_t)y + (size_t)V.dy * (size_t)z);
        for (int x = x0; x < ex; ++x) {
            float s = raw_voxel(V.voxels, V.dtype, base + x) * V.norm;
            float val = (s + V.offset) * V.one_minus_scaling;
            mn = min_(mn, val);
            mx = max_(mx, val);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = min_(mn, __shfl_down(mn, off, 64));
        mx = max_(mx, __shfl_down(mx, off, 64));
    }
    if (threadIdx.x == 0) {
        out[2 * (size_t)brick] = (uint16_t)__builtin_rintf(min_(max_(mn, 0.f), 1.f) * 65535.f);
        out[2 * (size_t)brick + 1] = (uint16_t)__builtin_rintf(min_(max_(mx, 0.f), 1.f) * 65535.f);
    }
}

// VolumeRAMDifferenceAnalysisDispatcher (ref uniformgridcl/processors/dynamicvolumedifferenceanalysis.h:96-151):
// mean |b - a| per brick over the format's range; integer formats sum exactly (u64),
// float volumes are summed by one lane in the reference's x-y-z order (double).
__global__ __launch_bounds__(64) void difference_kernel(BrickVol A, const void* __restrict__ bvox, double range,
                                                        float* __restrict__ out) {
    const int brick = blockIdx.x;
    const int gx = brick % A.ox, gy = (brick / A.ox) % A.oy, gz = brick / (A.ox * A.oy);
    const int x0 = gx * A.region, y0 = gy * A.region, z0 = gz * A.region;
    const int ex = min(x0 + A.region, A.dx), ey = min(y0 + A.region, A.dy), ez = min(z0 + A.region, A.dz);
    const double cnt = (double)A.region * A.region * A.region;
    if (A.dtype == CPM_F32) {
        if (threadIdx.x != 0) return;
        double sum = 0;
        for (int z = z0; z < ez; ++z)
            for (int y = y0; y < ey; ++y)
                for (int x = x0; x < ex; ++x) {
                    size_t i = (size_t)x + (size_t)A.dx * ((size_t)y + (size_t)A.dy * (size_t)z);
                    sum += fabs((double)static_cast<const float*>(bvox)[i] - (double)static_cast<const float*>(A.voxels)[i]);
                }
        out[brick] = (float)((sum / cnt) / range);
        return;
    }
    unsigned long long sum = 0;
    const int rows = A.region * A.region;
    for (int r = threadIdx.x; r < rows; r += 64) {
        int y = y0 + r % A.region, z = z0 + r / A.region;
        if (y >= ey || z >= ez) continue;
        size_t base = (size_t)A.dx * ((size_t)y + (size_t)A.dy * (size_t)z);
        for (int x = x0; x < ex; ++x) {
            int a, b;
            if (A.dtype == CPM_U8) { a = static_cast<const uint8_t*>(A.voxels)[base + x]; b = static_cast<const uint8_t*>(bvox)[base + x]; }
            else { a = static_cast<const uint16_t*>(A.voxels)[base + x]; b = static_cast<const uint16_t*>(bvox)[base + x]; }
            sum += (unsigned long long)(a > b ? a - b : b - a);
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if (threadIdx.x == 0) out[brick] = (float)(((double)sum / cnt) / range);
}

// ---- streaming variants ------------------------------------------------------------------------------
// The per-brick kernels above give each lane one (y, z) row of a brick: 8 byte loads from 64 different
// cache lines per wave instruction -- 70 / 155 us for a 256^3 u8 volume (0.2 TB/s).  min / max and the
// integer |b - a| sums do not depend on the order of their operands, so the work can follow memory instead:
// one workgroup owns a ROW OF BRICKS (all bricks with one (gy, gz): region^2 full-length x rows), lanes read
// 16 bytes each along x (coalesced), reduce the voxels of a chunk that fall into one brick in registers and
// combine across rows with LDS atomics on per-brick slots (raw integer min / max / u64 sum; floats through the
// order-preserving integer key).  The value mapping (v * norm + offset) * (1 - scaling) is monotone, so it is
// applied once per brick to the raw extremes: the same values as mapping every voxel.
CPM_DEV uint32_t float_key(float f) {  // order-preserving float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
CPM_DEV float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// MODE 0: min / max bricks of A; 1: mean |B - A| bricks; 2: both in one pass -- the difference against A and the min / max
// of B (a time step: the new volume's bricks and what changed, with each volume read once)
template <int DT, int MODE>
__global__ __launch_bounds__(256) void brick_row_kernel(BrickVol A, const void* __restrict__ bvox, double range,
                                                        uint16_t* __restrict__ mm_out, float* __restrict__ diff_out) {
    constexpr bool DIFF = MODE != 0, MINMAX = MODE != 1;
    extern __shared__ unsigned long long s_slots[];  // DIFF: ox sums; MINMAX: ox minima and ox maxima (u32) behind them
    uint32_t* s_min = reinterpret_cast<uint32_t*>(s_slots + (DIFF ? A.ox : 0));
    uint32_t* s_max = s_min + A.ox;
    constexpr int ES = DT == CPM_U8 ? 1 : (DT == CPM_U16 ? 2 : 4);
    constexpr int EPC = 16 / ES;  // elements per 16-byte chunk
    const int gy = blockIdx.x % A.oy, gz = blockIdx.x / A.oy;
    const int R = A.region;
    const int y0 = gy * R, z0 = gz * R;
    for (int g = threadIdx.x; g < A.ox; g += blockDim.x) {
        if (DIFF) s_slots[g] = 0ull;
        if (MINMAX) { s_min[g] = 0xffffffffu; s_max[g] = 0u; }
    }
    __syncthreads();
    const int cpr = (A.dx + EPC - 1) / EPC;  // chunks per row
    const int total = R * R * cpr;
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int r = c / cpr, xc = c - r * cpr;
        const int ry = r % R, rz = r / R;
        const int y = y0 + ry, z = z0 + rz;
        if (y >= A.dy || z >= A.dz) continue;
        const size_t base = (size_t)A.dx * ((size_t)y + (size_t)A.dy * (size_t)z) + (size_t)xc * EPC;  // element index, 16-byte aligned
        const uint4 qa = *reinterpret_cast<const uint4*>(static_cast<const char*>(A.voxels) + base * ES);
        uint4 qb = make_uint4(0, 0, 0, 0);
        if (DIFF) qb = *reinterpret_cast<const uint4*>(static_cast<const char*>(bvox) + base * ES);
        const uint32_t wa[4] = { qa.x, qa.y, qa.z, qa.w }, wb[4] = { qb.x, qb.y, qb.z, qb.w };
        const int xb = xc * EPC;
        int g = xb / R, left = R - (xb - g * R);  // brick of the first element, elements left in it
        uint32_t mn = 0xffffffffu, mx = 0u;
        unsigned long long sum = 0;
        bool any = false;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            if (xb + e < A.dx) {
                uint32_t va, vb = 0;
                if (DT == CPM_U8) { va = (wa[e >> 2] >> (8 * (e & 3))) & 0xffu; vb = (wb[e >> 2] >> (8 * (e & 3))) & 0xffu; }
                else if (DT == CPM_U16) { va = (wa[e >> 1] >> (16 * (e & 1))) & 0xffffu; vb = (wb[e >> 1] >> (16 * (e & 1))) & 0xffffu; }
                else va = float_key(__uint_as_float(wa[e]));
                if (DIFF) sum += va > vb ? va - vb : vb - va;
                if (MINMAX) { const uint32_t vm = MODE == 2 ? vb : va; mn = vm < mn ? vm : mn; mx = vm > mx ? vm : mx; }
                any = true;
                if (--left == 0) {  // the brick ends inside the chunk
                    if (DIFF) atomicAdd(&s_slots[g], sum);
                    if (MINMAX) { atomicMin(&s_min[g], mn); atomicMax(&s_max[g], mx); }
                    ++g; left = R; mn = 0xffffffffu; mx = 0u; sum = 0; any = false;
                }
            }
        }
        if (any) {
            if (DIFF) atomicAdd(&s_slots[g], sum);
            if (MINMAX) { atomicMin(&s_min[g], mn); atomicMax(&s_max[g], mx); }
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < A.ox; g += blockDim.x) {
        const size_t brick = (size_t)g + (size_t)A.ox * ((size_t)gy + (size_t)A.oy * (size_t)gz);
        if (DIFF) {
            const double cnt = (double)R * R * R;
            diff_out[brick] = (float)(((double)s_slots[g] / cnt) / range);
        }
        if (MINMAX) {
            const float lo = DT == CPM_F32 ? key_float(s_min[g]) : (float)s_min[g];
            const float hi = DT == CPM_F32 ? key_float(s_max[g]) : (float)s_max[g];
            const float a = (lo * A.norm + A.offset) * A.one_minus_scaling;
            const float b = (hi * A.norm + A.offset) * A.one_minus_scaling;
            const float mnv = min_(kFltMax, min_(a, b)), mxv = max_(0.f, max_(a, b));  // the reference's initial values
            mm_out[2 * brick] = (uint16_t)__builtin_rintf(min_(max_(mnv, 0.f), 1.f) * 65535.f);
            mm_out[2 * brick + 1] = (uint16_t)__builtin_rintf(min_(max_(mxv, 0.f), 1.f) * 65535.f);
        }
    }
}

// rows must start on 16-byte boundaries for the vector loads (hipMalloc aligns the block itself)
bool rows_are_16_byte_aligned(const BrickVol& V) {
    const int es = V.dtype == CPM_U8 ? 1 : (V.dtype == CPM_U16 ? 2 : 4);
    return ((size_t)V.dx * es) % 16 == 0;
}

CPM_DEV float4 mix4(float4 a, float4 b, float t) {
    return make_float4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
CPM_DEV float4 min4(float4 a, float4 b) { return make_float4(min_(a.x, b.x), min_(a.y, b.y), min_(a.z, b.z), min_(a.w, b.w)); }
CPM_DEV float4 max4(float4 a, float4 b) { return make_float4(max_(a.x, b.x), max_(a.y, b.y), max_(a.z, b.z), max_(a.w, b.w)); }

// importanceForRangeTF + tfPointsImportance with -D INCREMENTAL_TF_IMPORTANCE
// (ref importancesamplingcl/cl/minmaxuniformgrid3dimportance.cl:163-169,186-227)
CPM_DEV float importance_for_range_tf(float rx, float ry, const float* __restrict__ pos, const float4* __restrict__ col, int nPoints) {
    int i = 0;
    while (i < nPoints - 1 && rx > pos[i + 1]) ++i;
    float4 color = mix4(col[i], col[i + 1], (rx - pos[i]) / (pos[i + 1] - pos[i]));
    float4 mn = color, mx = color;
    if (ry <= pos[i + 1]) {
        float4 nc = mix4(col[i], col[i + 1], (ry - pos[i]) / (pos[i + 1] - pos[i]));
        mx = max4(mx, nc);
        return mx.x + mx.y + mx.z + mx.w;
    } else {
        float4 nc = col[i + 1];
        mn = min4(mn, nc); mx = max4(mx, nc);
        ++i;
    }
    while (i < nPoints - 1 && ry > pos[i + 1]) {
        float4 nc = col[i + 1];
        mn = min4(mn, nc); mx = max4(mx, nc);
        ++i;
    }
    if (i < nPoints - 1) {
        color = mix4(col[i], col[i + 1], (ry - pos[i]) / (pos[i + 1] - pos[i]));
        mn = min4(mn, color); mx = max4(mx, color);
    }
    (void)mn;
    return mx.x + mx.y + mx.z + mx.w;
}

// One bit per importance-grid cell, set where the cell's importance is anything but +0.0f (what the selection's grid walk
// tests before it loads a cell): every wave writes its 64 cells' two words.
CPM_DEV void write_occupancy(uint32_t* __restrict__ bits, uint32_t i, uint32_t n_cells, bool set) {
    const unsigned long long m = __ballot(set);
    if ((threadIdx.x & 63u) == 0u && i < ((n_cells + 63u) & ~63u)) {
        bits[(i >> 5)] = (uint32_t)m;
        bits[(i >> 5) + 1] = (uint32_t)(m >> 32);
    }
}

// The same kernel with the break points handed over as kernel arguments (<= kTfArgPoints of them: a TF has tens) and staged
// in LDS: no host-to-device copy ahead of the launch, nothing for the caller's arrays to outlive.
constexpr int kTfArgPoints = 48;
struct TfPointArgs { float4 col[kTfArgPoints]; float pos[kTfArgPoints]; };
CPM_DEV float importance_for_range_tf(float rx, float ry, const float* __restrict__ pos, const float4* __restrict__ col, int nPoints);
__global__ __launch_bounds__(256) void importance_tf_args_kernel(const uint16_t* __restrict__ mm, const uint16_t* __restrict__ prev,
                                                                 const float* __restrict__ diff, int n_cells, const TfPointArgs P,
                                                                 int n_points, float* __restrict__ out, uint32_t* __restrict__ occupancy) {
    __shared__ float4 s_col[kTfArgPoints];
    __shared__ float s_pos[kTfArgPoints];
    if ((int)threadIdx.x < n_points) { s_col[threadIdx.x] = P.col[threadIdx.x]; s_pos[threadIdx.x] = P.pos[threadIdx.x]; }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (i < n_cells) {
        uint16_t lo = mm[2 * i], hi = mm[2 * i + 1];
        if (prev) {
            uint16_t pl = prev[2 * i], ph = prev[2 * i + 1];
            lo = pl < lo ? pl : lo;
            hi = ph > hi ? ph : hi;
        }
        float rx = (1.f / 65535.f) * (float)lo, ry = (1.f / 65535.f) * (float)hi;
        float imp = importance_for_range_tf(rx, ry, s_pos, s_col, n_points);
        v = prev ? diff[i] * imp : imp;
        out[i] = v;
    }
    if (occupancy) write_occupancy(occupancy, (uint32_t)i, (uint32_t)n_cells, i < n_cells && __float_as_uint(v) != 0u);
}

// classifyMinMaxUniformGrid3DImportanceKernel / classifyTimeVarying... (ref ...importance.cl:269-330)
__global__ __launch_bounds__(256) void importance_tf_kernel(const uint16_t* __restrict__ mm, const uint16_t* __restrict__ prev,
                                                            const float* __restrict__ diff, int n_cells,
                                                            const float* __restrict__ pos, const float4* __restrict__ col,
                                                            int n_points, float* __restrict__ out, uint32_t* __restrict__ occupancy) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (i < n_cells) {
        uint16_t lo = mm[2 * i], hi = mm[2 * i + 1];
        if (prev) {
            uint16_t pl = prev[2 * i], ph = prev[2 * i + 1];
            lo = pl < lo ? pl : lo;
            hi = ph > hi ? ph : hi;
        }
        float rx = (1.f / 65535.f) * (float)lo, ry = (1.f / 65535.f) * (float)hi;
        float imp = importance_for_range_tf(rx, ry, pos, col, n_points);
        v = prev ? diff[i] * imp : imp;
        out[i] = v;
    }
    if (occupancy) write_occupancy(occupancy, (uint32_t)i, (uint32_t)n_cells, i < n_cells && __float_as_uint(v) != 0u);
}

struct ImpGrid {
    const float* grid;
    int dims[3];
    float cell[3];
    float inv_cell[3];  // exact reciprocal of a power-of-two cell size
    int cell_pow2[3];
    Affine t2i;
    RecLayout rec;      // where the halves of the photon records lie (the context's layout: cpm_set_photon_layout)
};

void set_cell(ImpGrid& G, int a, float cell) {
    G.cell[a] = cell;
    int e = 0;
    const float m = frexpf(cell, &e);  // cell = m * 2^e, m in [0.5, 1)
    G.cell_pow2[a] = (m == 0.5f && e > -120 && e < 120) ? 1 : 0;

    G.inv_cell[a] = G.cell_pow2[a] ? 1.0f / cell : 0.f;
}

// setupUniformGridTraversal + stepToNextCellNextHit (OPTIMIZE_STEP_FOR_SIMD) driven by
// uniformGridImportance (ref uniformgridcl/cl/uniformgrid/uniformgrid.cl:38-69,147-167;
// progressivephotonmapping/cl/photonrecomputationdetector.cl:55-90)
// MASK: `mask` holds one bit per cell (set = the cell's importance is not +0.0f); a clear bit stands for the value +0.0f
// without the load -- the same operand, so the same sum (the multiply and the add are still performed).
// cell_pow2[a]: the cell size is a power of two and inv_cell[a] its exact reciprocal; x * inv_cell is then the correctly
// rounded quotient x / cell (both are exact scalings) without the division sequence.
// (Walking several cells ahead with their loads in flight together, and finishing long walks by dense waves from an LDS
// queue, were both built, bit-identical, measured slower and removed: docs/EXPERIMENTS.md.)
template <bool MASK>
CPM_DEV float uniform_grid_importance(const ImpGrid& G, const uint32_t* mask, const float x1[3], const float x2[3]) {
    int cell[3], cellEnd[3], di[3];
    float dt[3], deltatx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float maxc = (float)(G.dims[a] - 1);
        const float q1 = G.cell_pow2[a] ? x1[a] * G.inv_cell[a] : x1[a] / G.cell[a];
        float cf = min_(max_(__builtin_floorf(q1), 0.f), maxc);
        cell[a] = (int)cf;
        float ef = G.cell_pow2[a] ? x2[a] * G.inv_cell[a] : x2[a] / G.cell[a];
        ef = min_(max_(ef, -1.f), (float)G.dims[a]);
        int ei = (int)ef;
        cellEnd[a] = ei < 0 ? 0 : (ei > G.dims[a] - 1 ? G.dims[a] - 1 : ei);
        di[a] = (x1[a] < x2[a]) ? 1 : ((x1[a] > x2[a]) ? -1 : 0);
        float invAbsDir = 1.f / __builtin_fabsf(x2[a] - x1[a]);
        float minx = G.cell[a] * cf;
        float maxx = minx + G.cell[a];
        dt[a] = ((x1[a] > x2[a]) ? (x1[a] - minx) : (maxx - x1[a])) * invAbsDir;
        deltatx[a] = G.cell[a] * invAbsDir;
    }
    float importance = 0.f, dt1 = 0.f;
    bool cont = true;
    int cap = G.dims[0] + G.dims[1] + G.dims[2] + 4;  // every wave reaches its exit, NaN input included
    const uint32_t sy = (uint32_t)G.dims[0], sz = (uint32_t)G.dims[0] * (uint32_t)G.dims[1];
    while (cont && cap-- > 0) {
        const uint32_t ci = (uint32_t)cell[0] + (uint32_t)cell[1] * sy + (uint32_t)cell[2] * sz;
        float val = 0.f;
        if (!MASK || ((mask[ci >> 5] >> (ci & 31u)) & 1u)) val = G.grid[ci];
        float dt0 = dt1;
        bool ax0 = dt[0] <= dt[1] && dt[0] <= dt[2];
        bool ax1 = !ax0 && (dt[0] > dt[1] && dt[1] <= dt[2]);
        // branch-free axis select (the reference's SIMD variant)
        float dsel = ax0 ? dt[0] : (ax1 ? dt[1] : dt[2]);
        int csel = ax0 ? cell[0] : (ax1 ? cell[1] : cell[2]);
        int esel = ax0 ? cellEnd[0] : (ax1 ? cellEnd[1] : cellEnd[2]);
        dt1 = dsel;
        if (csel == esel) {
            cont = false;
        } else {
            if (ax0) { dt[0] += deltatx[0]; cell[0] += di[0]; }
            else if (ax1) { dt[1] += deltatx[1]; cell[1] += di[1]; }
            else { dt[2] += deltatx[2]; cell[2] += di[2]; }
        }
        importance += val * (min_(1.f, dt1) - dt0);
    }
    float lx = x2[0] - x1[0], ly = x2[1] - x1[1], lz = x2[2] - x1[2];
    float len = __builtin_sqrtf(fma_(lz, lz, fma_(ly, ly, lx * lx)));
    return importance * len;
}

// convert_uint_sat_rtp(100 * imp) clamped to 0x7fffffff
CPM_DEV uint32_t importance_to_uint(float imp100) {
    if (!(imp100 > 0.f)) return 0u;
    float c = __builtin_ceilf(imp100);
    if (c >= 2147483648.f) return 2147483647u;
    uint32_t u = (uint32_t)c;
    return u > 2147483647u ? 2147483647u : u;
}

// photonRecomputationDetectorKernel's body for one light sample (ref progressivephotonmapping/cl/photonrecomputationdetector.cl:92-157):
// the importance, times 100, saturated to uint
template <bool MASK>
CPM_DEV uint32_t photon_importance_value(const ImpGrid& G, const uint32_t* mask, const float* __restrict__ photons, int photon_offset,
                                         const float* __restrict__ ls, const float* __restrict__ isect, int max_interactions,
                                         int total_photons, int fix_exit_point, int threadId) {
    const float bmin[3] = { 0.f, 0.f, 0.f }, bmax[3] = { 1.f, 1.f, 1.f };
    float recomputationImportance = 0.f;
    const float4* lsp = reinterpret_cast<const float4*>(ls) + 2 * (size_t)threadId;
    float4 l0 = lsp[0], l1 = lsp[1];
    f3 origin = { l0.x, l0.y, l0.z };
    f3 direction = decode_direction_(l1.z, l1.w);
    float2 ip = reinterpret_cast<const float2*>(isect)[threadId];
    float tStart = ip.x, tEnd = ip.y;
    if (tStart < tEnd) {
        f3 entry = { fma_(tStart, direction.x, origin.x), fma_(tStart, direction.y, origin.y), fma_(tStart, direction.z, origin.z) };
        for (int interaction = 0; interaction < max_interactions; ++interaction) {
            size_t photonId = (size_t)photon_offset + (size_t)interaction * total_photons + threadId;
            const float4* q = rec_at(photons, G.rec, photonId);
            const float4 a = q[0];   // (half B -- the direction -- only where a later interaction left the volume: in two planes it is another line)
            f3 exitp = { a.x, a.y, a.z };
            if (a.x == kFltMax || a.y == kFltMax || a.z == kFltMax) {
                if (interaction == 0) {
                    if (fix_exit_point) {
                        exitp.x = fma_(tEnd, direction.x, origin.x);
                        exitp.y = fma_(tEnd, direction.y, origin.y);
                        exitp.z = fma_(tEnd, direction.z, origin.z);
                    } else {  // origin omitted
                        exitp.x = tEnd * direction.x; exitp.y = tEnd * direction.y; exitp.z = tEnd * direction.z;
                    }
                } else if (entry.x == kFltMax || entry.y == kFltMax || entry.z == kFltMax) {
                    break;
                } else {
                    float t0 = 0.f, t1 = kFltMax;
                    const float4 b = q[G.rec.b];
                    f3 pd = decode_direction_(b.z, b.w);
                    if (a.w != kFltMax && ray_box_(bmin, bmax, entry, pd, t0, t1)) {
                        exitp.x = fma_(t1, pd.x, entry.x);
                        exitp.y = fma_(t1, pd.y, entry.y);
                        exitp.z = fma_(t1, pd.z, entry.z);
                    } else {
                        break;
                    }
                }
            }
            f3 ia = transform_(G.t2i, entry), ib = transform_(G.t2i, exitp);
            float x1[3] = { ia.x + 0.5f, ia.y + 0.5f, ia.z + 0.5f };
            float x2[3] = { ib.x + 0.5f, ib.y + 0.5f, ib.z + 0.5f };
            recomputationImportance += uniform_grid_importance<MASK>(G, mask, x1, x2);
            entry.x = a.x; entry.y = a.y; entry.z = a.z;
        }
    }
    return importance_to_uint(100.f * recomputationImportance);
}

__global__ __launch_bounds__(256) void photon_importance_kernel(ImpGrid G, const float* __restrict__ photons,
                                                                int photon_offset, const float* __restrict__ ls,
                                                                const float* __restrict__ isect, int n_light_samples,
                                                                int max_interactions, int total_photons,
                                                                int fix_exit_point, uint32_t* __restrict__ importances) {
    int threadId = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadId >= n_light_samples) return;
    importances[photon_offset + threadId] -= photon_importance_value<false>(G, nullptr, photons, photon_offset, ls, isect, max_interactions,
                                                                             total_photons, fix_exit_point, threadId);
}

// ---- the fused selection (cpm_photon_importance_select, cpm_selection_finish) -------------------------------------

// one bit per importance-grid cell: set where the cell's importance is anything but +0.0f
__global__ __launch_bounds__(256) void importance_mask_kernel(const float* __restrict__ grid, uint32_t n_cells, uint32_t* __restrict__ mask) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    write_occupancy(mask, i, n_cells, i < n_cells && __float_as_uint(grid[i]) != 0u);
}

struct SelTiles {
    uint2* tile;          // per tile: (count, start of its list in `local`)
    uint32_t* local;      // tile-local lists: the tile of photons [photon_offset + b0, ...) lists its selected photons from there
    uint32_t tile_first;  // this launch's first tile
    uint32_t per_tile;    // photons per tile = 256 * K
};

// MODE 0: importance by DDA through the grid (mask staged in LDS when MASK); MODE 1: the equal-importance rule.
// One workgroup = one tile of K * 256 consecutive light samples; thread t takes samples b0 + k * 256 + t.  Importances are
// updated as by photon_importance_kernel; the photons whose key is then < 0x7fffffff are ranked with ballots (ascending
// sample index) and listed at the tile's own place -- no atomics, no global prefix: cpm_selection_finish lines the tiles up.
template <int MODE, bool MASK, int K>
__global__ __launch_bounds__(256) void importance_select_kernel(ImpGrid G, const uint32_t* __restrict__ mask, uint32_t mask_words,
                                                                const float* __restrict__ photons, int photon_offset,
                                                                const float* __restrict__ ls, const float* __restrict__ isect,
                                                                int n_light_samples, int max_interactions, int total_photons,
                                                                int fix_exit_point, int eq_percentage, int eq_iteration,
                                                                uint32_t* __restrict__ importances, SelTiles S) {
    extern __shared__ uint32_t s_mask[];
    __shared__ uint32_t s_wcnt[K][4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const int b0 = (int)(blockIdx.x * S.per_tile);
    const int b1 = min(b0 + (int)S.per_tile, n_light_samples);
    if (MODE == 0 && MASK) {
        for (uint32_t i = t; i < mask_words; i += 256u) s_mask[i] = mask[i];
        __syncthreads();
    }
    unsigned long long ballots[K];
    uint32_t flags = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int threadId = b0 + k * 256 + (int)t;
        bool changed = false;
        if (threadId < b1) {
            uint32_t u;
            if (MODE == 0) {
                u = photon_importance_value<MASK>(G, s_mask, photons, photon_offset, ls, isect, max_interactions, total_photons, fix_exit_point, threadId);
            } else {
                const int photonId = photon_offset + threadId;
                u = ((photonId + eq_iteration) % (100 / eq_percentage) == 0) ? importance_to_uint(100.f * 1.f) : importance_to_uint(100.f * 0.f);
            }
            uint32_t key = importances[photon_offset + threadId];
            if (u != 0u) { key -= u; importances[photon_offset + threadId] = key; }   // (key -= 0 leaves the word as it is)
            changed = key < 2147483647u;
        }
        ballots[k] = __ballot(changed);
        flags |= changed ? (1u << k) : 0u;
        if (lane == 0) s_wcnt[k][wave] = (uint32_t)__popcll(ballots[k]);
    }
    __syncthreads();
    uint32_t before = 0;  // selected photons of the tile ahead of (chunk k, this wave)
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* list = S.local + (size_t)photon_offset + (size_t)b0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint32_t mine = before;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = s_wcnt[k][w];
            mine += w < (int)wave ? c : 0u;
            before += c;
        }
        if (flags & (1u << k)) list[mine + (uint32_t)__popcll(ballots[k] & lt)] = (uint32_t)(photon_offset + b0 + k * 256 + (int)t);
    }
    if (t == 0) S.tile[S.tile_first + blockIdx.x] = make_uint2(before, (uint32_t)(photon_offset + b0));
}

// Tile lists -> one ascending list.  Workgroup b = tile b: sums the counts of the tiles before it (<= a few thousand
// loads, no scan launch), copies its list behind them; workgroup 0 also publishes the total -- the device word the
// following launches read and the pinned host mailbox (epoch << 32 | count) the host polls instead of synchronising.
// (A workgroup takes kCompactGroup tiles: with one tile each, 4096 tiles of 256 photons meant 16 M count loads -- 13 us.)
constexpr uint32_t kCompactGroup = 16;  // tiles per workgroup of selection_compact_kernel
__global__ __launch_bounds__(256) void selection_compact_kernel(const uint2* __restrict__ tile, uint32_t n_tiles,
                                                                const uint32_t* __restrict__ local, uint32_t* __restrict__ indices,
                                                                int32_t* __restrict__ count_dev, unsigned long long* mailbox, uint32_t epoch) {
    __shared__ uint32_t red[2][4];
    __shared__ uint2 s_t[kCompactGroup];
    __shared__ uint32_t s_off[kCompactGroup + 1];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t g0 = blockIdx.x * kCompactGroup;
    uint32_t before = 0, total = 0;
    const uint32_t upto = blockIdx.x == 0 ? n_tiles : g0;  // (only workgroup 0 needs the total)
    for (uint32_t i = t; i < upto; i += 256u) {
        const uint32_t c = tile[i].x;
        total += c;
        before += i < g0 ? c : 0u;
    }
    if (t < kCompactGroup) s_t[t] = g0 + t < n_tiles ? tile[g0 + t] : make_uint2(0u, 0u);
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_down(before, off, 64); total += __shfl_down(total, off, 64); }
    if (lane == 0) { red[0][wave] = before; red[1][wave] = total; }
    __syncthreads();
    before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (blockIdx.x == 0 && t == 0) {
        *count_dev = (int32_t)total;
        __hip_atomic_store(mailbox, ((unsigned long long)epoch << 32) | (unsigned long long)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t q = 0; q < kCompactGroup; ++q) { s_off[q] = run; run += s_t[q].x; }
        s_off[kCompactGroup] = run;
    }
    __syncthreads();
    // the group's entries, flattened: entry e belongs to the tile q with s_off[q] <= e < s_off[q + 1]
    const uint32_t entries = s_off[kCompactGroup];
    for (uint32_t e = t; e < entries; e += 256u) {
        uint32_t q = 0;
#pragma unroll
        for (uint32_t k = 1; k < kCompactGroup; ++k) q += s_off[k] <= e ? 1u : 0u;  // (empty tiles repeat an offset: counted past)
        indices[before + e] = local[(size_t)s_t[q].y + (e - s_off[q])];
    }
}

// what a retrace launch's tiles cost -> sort keys (costliest first under an ascending sort), identity values; costs cleared
constexpr uint32_t kCostBits = 20;
__global__ __launch_bounds__(256) void retrace_order_keys_kernel(uint32_t* __restrict__ cost, uint32_t* __restrict__ keys, uint32_t* __restrict__ order, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t top = (1u << kCostBits) - 1u, c = cost[i];
    keys[i] = top - (c < top ? c : top);
    order[i] = i;
    cost[i] = 0u;
}

// photonRecomputationDetectorEqualImportanceKernel (ref ...detector.cl:160-194)
__global__ __launch_bounds__(256) void photon_importance_equal_kernel(int photon_offset, int n_light_samples, int percentage,
                                                                      int iteration, uint32_t* __restrict__ importances) {
    int threadId = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadId >= n_light_samples) return;
    float imp = 0.f;
    int photonId = photon_offset + threadId;
    if ((photonId + iteration) % (100 / percentage) == 0) imp = 1.f;
    importances[photon_offset + threadId] -= importance_to_uint(100.f * imp);
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, size_t n, uint32_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// thresholdKernel + clogs::Reduce + indexToBufferKernel in one pass
// (ref cl/threshold.cl:33-40, radixsortcl/ext/clogs/kernels/reduce.cl:96-154, cl/indextobuffer.cl:33-40)
__global__ __launch_bounds__(256) void threshold_count_iota_kernel(const uint32_t* __restrict__ imp, size_t n,
                                                                   uint32_t* __restrict__ idx, int* __restrict__ count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool changed = false;
    if (i < n) { changed = imp[i] < 2147483647u; idx[i] = (uint32_t)i; }
    unsigned long long m = __ballot(changed);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (int)__popcll(m));
}

// cpm_select_changed as a two-launch stable partition (default).  A tile = `per_tile` consecutive photons (a multiple of
// 256; at most kPartMaxTiles tiles).  Launch 1 counts the changed photons of every tile.  In launch 2 every workgroup
// sums the counts of the tiles before it itself (<= 1024 loads, no scan launch), then walks its tile 256 photons at
// a time: a ballot gives each photon its rank among the changed / unchanged ones of its wave, the four wave totals go
// through LDS, and the index is written behind everything that precedes it in its part.  Ascending in both parts,
// no atomics, importances read twice (8 MB at 1 M photons), indices written once.
constexpr int kPartMaxTiles = 1024;

__global__ __launch_bounds__(256) void partition_count_kernel(const uint32_t* __restrict__ imp, uint32_t n, uint32_t per_tile,
                                                              uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t wsum[4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b0 = blockIdx.x * per_tile, b1 = min(b0 + per_tile, n);
    uint32_t c = 0;
    for (uint32_t i = b0 + t; i < b1; i += 256) c += imp[i] < 2147483647u ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (t == 0) tile_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void partition_write_kernel(const uint32_t* __restrict__ imp, uint32_t n, uint32_t per_tile,
                                                              const uint32_t* __restrict__ tile_count, uint32_t num_tiles,
                                                              uint32_t* __restrict__ idx, int32_t* __restrict__ n_changed) {
    __shared__ uint32_t red[2][4];
    __shared__ uint32_t wcnt[2][4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // changed photons in the tiles before this one, and in all tiles
    uint32_t before = 0, total = 0;
    for (uint32_t i = t; i < num_tiles; i += 256) {
        const uint32_t c = tile_count[i];
        total += c;
        before += i < blockIdx.x ? c : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_down(before, off, 64); total += __shfl_down(total, off, 64); }
    if (lane == 0) { red[0][wave] = before; red[1][wave] = total; }
    __syncthreads();
    before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (blockIdx.x == 0 && t == 0) *n_changed = (int32_t)total;
    const uint32_t b0 = blockIdx.x * per_tile, b1 = min(b0 + per_tile, n);
    uint32_t out_c = before;                 // next free slot of the changed part
    uint32_t out_u = total + (b0 - before);  // ... of the unchanged part: unchanged photons before this tile
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t base = b0; base < b1; base += 256) {
        const uint32_t i = base + t;
        const bool valid = i < b1;
        const bool ch = valid && imp[i] < 2147483647u;
        const uint64_t mc = __ballot(ch), mv = __ballot(valid);
        const uint32_t wc = (uint32_t)__popcll(mc), wu = (uint32_t)__popcll(mv) - wc;
        if (lane == 0) { wcnt[0][wave] = wc; wcnt[1][wave] = wu; }
        __syncthreads();
        uint32_t pc = 0, pu = 0, tc = 0, tu = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = wcnt[0][w], u = wcnt[1][w];
            pc += w < (int)wave ? c : 0u; pu += w < (int)wave ? u : 0u;
            tc += c; tu += u;
        }
        if (valid) {
            const uint32_t rc = (uint32_t)__popcll(mc & lt_mask);
            if (ch) idx[out_c + pc + rc] = i;
            else idx[out_u + pu + (lane - rc - (uint32_t)__popcll(~mv & lt_mask))] = i;
        }
        out_c += tc; out_u += tu;
        __syncthreads();  // wcnt is rewritten by the next chunk
    }
}

// Detector + threshold + tracer in ONE launch (cpm_photon_importance_retrace): the importance branch of
// ProgressivePhotonTracerCL::process (ref processor/progressivephotontracercl.cpp:298-374,467-529) for the case that every
// changed photon is traced in this evaluation.  Pass 1 per tile as importance_select_kernel: importance of every sample,
// threshold, ballot-ranked list + count (for the index list the outport carries and the light-volume update walks).  Pass 2:
// the lanes whose photon changed re-trace it on the spot -- its records are first copied to old_sparse at the photon's own
// index (what the reference's prevPhotons_ snapshot holds for it), then overwritten by tracer::trace_photon, the same device
// function trace_kernel runs: same RNG stream, same operations, same bits; its importance key goes back to 0x7fffffff
// (resetPhotonImportance).  What this saves against cpm_photon_importance_select + cpm_selection_finish + cpm_trace_selected:
// the re-trace no longer waits for the compaction, and its launch -- a chain of dependent loads (count, index, sample, old
// record) for a few thousand photons, 17 us at config 3 -- is gone; the divergent walks (3 steps on average) hide among the
// importance pass's own waves.
// Tiles of kRetraceTile = 256 photons -- one per lane -- taken in the order of `order` (costliest first; nullable = tile b
// to workgroup b).  A tile's cost is its slowest wave's wall-clock: the rays through a transparent pocket walk 30 - 70 cells
// and are the re-traced photons too, they are lattice neighbours, and the launch used to end with a few such tiles started
// last (config 3: tiles of 512 in index order 46.2 us; costliest first 39.4 -- the costliest tile alone; tiles of 256 in index
// order 51 - 52, costliest first 34.1).  The cost is recorded (one atomic per wave) only in the launches the selection measures.
constexpr uint32_t kRetraceTile = 256;
#ifndef CPM_RETRACE_AHEAD
#define CPM_RETRACE_AHEAD 2
#endif
constexpr int kRetraceAhead = CPM_RETRACE_AHEAD;
// LINEAR: the volume's footprint copy is stale (a mixed time step): the re-traces fetch from the linear block (cpm::trace_volume_source)
// MULTI: the tiles of several lights in one launch (cpm_photon_importance_retrace_lights): a tile's place among the spans' tile ranges
// (TraceArgs::span, chunk_base = first tile) says whose samples it holds.
template <int DT, bool MASK, bool SINGLE, bool LINEAR = false, bool MULTI = false>
__global__ __launch_bounds__(256) void importance_retrace_kernel(ImpGrid G, const uint32_t* __restrict__ mask, uint32_t mask_words,
                                                                 int fix_exit_point, uint32_t* __restrict__ importances, SelTiles S,
                                                                 const tracer::TraceArgs A0, float* __restrict__ old_sparse,
                                                                 const uint32_t* __restrict__ order, uint32_t* __restrict__ cost) {
    tracer::TraceArgs A = A0;
    extern __shared__ uint32_t s_dyn[];  // [occupancy bits of the importance grid][TF alpha column(s)]
    uint32_t* s_mask = s_dyn;
    float* lut = reinterpret_cast<float*>(s_dyn + mask_words);
    float* luts = lut;
    __shared__ uint32_t s_wcnt[4];
    const unsigned long long w0 = cost ? wall_clock64() : 0ull;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t tile = order ? order[blockIdx.x] : blockIdx.x;
    uint32_t local_tile = tile;  // ... within its light
    if (MULTI) {  // (uniform)
        int sp = 0;
        while (sp + 1 < A0.n_spans && (int)tile >= A0.span[sp + 1].chunk_base) ++sp;
        local_tile = tile - (uint32_t)A0.span[sp].chunk_base;
        A.light_samples = A0.span[sp].light_samples;
        A.isect = A0.span[sp].isect;
        A.n_threads = A0.span[sp].n;
        A.p.n_light_samples = A0.span[sp].n;
        A.p.photon_offset = A0.span[sp].photon_offset;
    }
    const int n_light_samples = A.p.n_light_samples, photon_offset = A.p.photon_offset;
    const int b0 = (int)(local_tile * kRetraceTile);
    if (MASK) for (uint32_t i = t; i < mask_words; i += 256u) s_mask[i] = mask[i];
    for (int i = (int)t; i < A.tf_width; i += 256) lut[i] = A.tf_alpha[i];
    if (A.tfs_alpha != A.tf_alpha) {
        luts = lut + A.tf_width;
        for (int i = (int)t; i < A.tf_width; i += 256) luts[i] = A.tfs_alpha[i];
    }
    __syncthreads();
    // pass 1: importance, threshold, the tile's ascending list
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int threadId = b0 + (int)t;
    bool changed = false;
    if (threadId < n_light_samples) {
        const uint32_t u = photon_importance_value<MASK>(G, s_mask, A.photons, photon_offset, A.light_samples, A.isect, A.p.max_interactions,
                                                         A.p.total_photons, fix_exit_point, threadId);
        const uint32_t key = importances[photon_offset + threadId] - u;
        changed = key < 2147483647u;
    }
    const unsigned long long mc = __ballot(changed);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(mc);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t c = s_wcnt[w]; before += w < (int)wave ? c : 0u; total += c; }
    if (t == 0) S.tile[S.tile_first + tile] = make_uint2(total, (uint32_t)(photon_offset + b0));
    // pass 2: a changed photon is re-traced by the lane that found it
    unsigned steps = 0;
    if (changed) {
        S.local[(size_t)photon_offset + (size_t)b0 + before + (uint32_t)__popcll(mc & lt)] = (uint32_t)(photon_offset + threadId);
        const size_t totalPhotons = (size_t)A.p.total_photons;
        const int nInter = SINGLE ? 1 : A.p.max_interactions;
        for (int it = 0; it < nInter; ++it) {  // the records about to be replaced
            const size_t id = (size_t)photon_offset + (size_t)it * totalPhotons + (size_t)threadId;
            const float4* q = rec_at(A.photons, G.rec, id);   // (old_photons8 is laid out like photons8)
            float4* o = rec_at(old_sparse, G.rec, id);
            const float4 a = q[0], b = q[G.rec.b];
            o[0] = a; o[G.rec.b] = b;
        }
        const float4* lsp = reinterpret_cast<const float4*>(A.light_samples) + 2 * (size_t)threadId;
        const float4 l0 = lsp[0], l1 = lsp[1];
        const float2 ip = reinterpret_cast<const float2*>(A.isect)[threadId];
        const uint2 rs = reinterpret_cast<const uint2*>(A.rng)[photon_offset + threadId];
        const f3 direction = decode_direction_(l1.z, l1.w);
        float th, ph;
        encode_direction_(direction, th, ph);
        // (kRetraceAhead fetches of the walk in flight, as in trace_kernel; this launch is the importance pass, not its few walks:
        // 1 / 2 / 4 in flight 34.3 / 32.4 / 32.8 us, 8 -- registers -- 50.6)
        tracer::trace_photon<DT, SINGLE, kRetraceAhead, LINEAR>(A, lut, luts, threadId, l0, l1, ip, rs, direction, th, ph, steps);
        importances[photon_offset + threadId] = 2147483647u;  // resetPhotonImportance (tracercl.cpp:529)
    }
    // (lanes reconverge here; one lane of the wave reports for it)
    if (cost && lane == 0) atomicMax(&cost[tile], (uint32_t)(wall_clock64() - w0));
    if (A.step_counter) {
        unsigned sN = steps;
        for (int off = 32; off > 0; off >>= 1) sN += __shfl_down(sN, off, 64);
        if (lane == 0 && sN) atomicAdd(A.step_counter, (unsigned long long)sN);
    }
}

// The radix-pass form of cpm_select_changed (cpm_debug_set_select_partition(0)): flag = 0 for a photon whose importance
// says "re-trace" (key < 0x7fffffff), 1 otherwise; iota.
// One stable radix pass over the flag then partitions the indices (changed first, both parts ascending) and its
// digit total IS the count -- no atomics, no 31-bit sort.
__global__ __launch_bounds__(256) void changed_flag_iota_kernel(const uint32_t* __restrict__ imp, size_t n,
                                                                uint32_t* __restrict__ flag, uint32_t* __restrict__ idx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { flag[i] = imp[i] < 2147483647u ? 0u : 1u; idx[i] = (uint32_t)i; }
}
__global__ void select_single_kernel(const uint32_t* __restrict__ imp, uint32_t* __restrict__ idx, int32_t* __restrict__ count) {
    idx[0] = 0u;
    *count = imp[0] < 2147483647u ? 1 : 0;
}

int make_brick_vol(cpm_ctx* ctx, const cpm_volume* vol, int region, BrickVol& V) {
    if (!vol) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "volume", "null");
    if (region < 1 || region > 64) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "region", "must be in [1, 64]");
    const cpm_volume_desc& d = vol->desc;
    V.voxels = vol->voxels;
    V.dx = d.dims[0]; V.dy = d.dims[1]; V.dz = d.dims[2]; V.dtype = d.dtype;
    V.region = region;
    V.ox = (V.dx + region - 1) / region; V.oy = (V.dy + region - 1) / region; V.oz = (V.dz + region - 1) / region;
    V.norm = d.dtype == CPM_U8 ? (1.0f / 255.0f) : (d.dtype == CPM_U16 ? (1.0f / 65535.0f) : 1.0f);
    V.offset = d.format_offset;
    V.one_minus_scaling = 1.0f - d.format_scaling;
    return CPM_OK;
}

}  // namespace

extern "C" {

// test hook (include/cpm/cpm_profile.h): 1 (default) = streaming brick-row kernels where rows are 16-byte aligned
void cpm_debug_set_brick_streaming(cpm_ctx* ctx, int on) { if (ctx) ctx->dbg.brick_streaming = on; }  // 0 = always the per-brick kernels
void cpm_debug_set_select_partition(cpm_ctx* ctx, int on) { if (ctx) ctx->dbg.select_partition = on; }  // 0 = the radix-pass form
void cpm_debug_fail_next_select(cpm_ctx* ctx, int on) { if (ctx) ctx->dbg.fail_next_select = on; }

int cpm_volume_minmax(cpm_ctx* ctx, const cpm_volume* vol, int region, uint16_t* minmax2, cpm_stream stream) {
    CPM_ENTER(ctx);
    BrickVol V;
    int rc = make_brick_vol(ctx, vol, region, V);
    if (rc) return rc;
    CPM_REQUIRE(ctx, minmax2, "cpm_volume_minmax: null output");
    hipStream_t s = (hipStream_t)stream;
    if (ctx->dbg.brick_streaming && rows_are_16_byte_aligned(V) && (size_t)V.ox * 8 <= 48 * 1024) {
        const dim3 grid((unsigned)(V.oy * V.oz)), block(256);
        const size_t lds = (size_t)V.ox * 8;
        switch (V.dtype) {
            case CPM_U8: CPM_LAUNCH(ctx, (brick_row_kernel<CPM_U8, 0>), grid, block, lds, s, V, nullptr, 1.0, minmax2, nullptr); break;
            case CPM_U16: CPM_LAUNCH(ctx, (brick_row_kernel<CPM_U16, 0>), grid, block, lds, s, V, nullptr, 1.0, minmax2, nullptr); break;
            default: CPM_LAUNCH(ctx, (brick_row_kernel<CPM_F32, 0>), grid, block, lds, s, V, nullptr, 1.0, minmax2, nullptr); break;
        }
        CPM_LAUNCH_CHECK(ctx, "brick_row_kernel");
        return CPM_OK;
    }
    CPM_LAUNCH(ctx, minmax_kernel, dim3(V.ox * V.oy * V.oz), dim3(64), 0, s, V, minmax2);
    CPM_LAUNCH_CHECK(ctx, "minmax_kernel");
    return CPM_OK;
}

int cpm_volume_difference(cpm_ctx* ctx, const cpm_volume* cur, const cpm_volume* next, int region, float* out,
                          cpm_stream stream) {
    CPM_ENTER(ctx);
    BrickVol V;
    int rc = make_brick_vol(ctx, cur, region, V);
    if (rc) return rc;
    CPM_REQUIRE(ctx, next && out, "cpm_volume_difference: null argument");
    CPM_REQUIRE(ctx, memcmp(cur->desc.dims, next->desc.dims, sizeof(cur->desc.dims)) == 0 && cur->desc.dtype == next->desc.dtype,
                "cpm_volume_difference: volumes differ in shape or type");
    double range = V.dtype == CPM_U8 ? 255.0 : (V.dtype == CPM_U16 ? 65535.0 : 1.0);
    if (ctx->dbg.brick_streaming && V.dtype != CPM_F32 && rows_are_16_byte_aligned(V) && (size_t)V.ox * 8 <= 48 * 1024) {
        // (float volumes keep the per-brick kernel: their sum is defined in the reference's x-y-z order in double)
        const dim3 grid((unsigned)(V.oy * V.oz)), block(256);
        const size_t lds = (size_t)V.ox * 8;
        hipStream_t s = (hipStream_t)stream;
        if (V.dtype == CPM_U8) CPM_LAUNCH(ctx, (brick_row_kernel<CPM_U8, 1>), grid, block, lds, s, V, next->voxels, range, nullptr, out);
        else CPM_LAUNCH(ctx, (brick_row_kernel<CPM_U16, 1>), grid, block, lds, s, V, next->voxels, range, nullptr, out);
        CPM_LAUNCH_CHECK(ctx, "brick_row_kernel");
        return CPM_OK;
    }
    CPM_LAUNCH(ctx, difference_kernel, dim3(V.ox * V.oy * V.oz), dim3(64), 0, (hipStream_t)stream, V, next->voxels, range, out);
    CPM_LAUNCH_CHECK(ctx, "difference_kernel");
    return CPM_OK;
}

int cpm_volume_step(cpm_ctx* ctx, const cpm_volume* cur, const cpm_volume* next, int region, float* mean_abs_diff, uint16_t* next_minmax2,
                    cpm_stream stream) {
    CPM_ENTER(ctx);
    BrickVol V;
    int rc = make_brick_vol(ctx, cur, region, V);
    if (rc) return rc;
    CPM_REQUIRE(ctx, next && mean_abs_diff && next_minmax2, "cpm_volume_step: null argument");
    CPM_REQUIRE(ctx, memcmp(cur->desc.dims, next->desc.dims, sizeof(cur->desc.dims)) == 0 && cur->desc.dtype == next->desc.dtype,
                "cpm_volume_step: volumes differ in shape or type");
    if (ctx->dbg.brick_streaming && V.dtype != CPM_F32 && rows_are_16_byte_aligned(V) && (size_t)V.ox * 16 <= 48 * 1024 &&
        cur->desc.format_offset == next->desc.format_offset && cur->desc.format_scaling == next->desc.format_scaling) {
        const double range = V.dtype == CPM_U8 ? 255.0 : 65535.0;
        const dim3 grid((unsigned)(V.oy * V.oz)), block(256);
        const size_t lds = (size_t)V.ox * 16;
        hipStream_t s = (hipStream_t)stream;
        if (V.dtype == CPM_U8) CPM_LAUNCH(ctx, (brick_row_kernel<CPM_U8, 2>), grid, block, lds, s, V, next->voxels, range, next_minmax2, mean_abs_diff);
        else CPM_LAUNCH(ctx, (brick_row_kernel<CPM_U16, 2>), grid, block, lds, s, V, next->voxels, range, next_minmax2, mean_abs_diff);
        CPM_LAUNCH_CHECK(ctx, "brick_row_kernel");
        return CPM_OK;
    }
    rc = cpm_volume_difference(ctx, cur, next, region, mean_abs_diff, stream);
    if (rc) return rc;
    return cpm_volume_minmax(ctx, next, region, next_minmax2, stream);
}

int cpm_importance_tf(cpm_ctx* ctx, const uint16_t* minmax2, const uint16_t* prev_minmax2, const float* volume_diff,
                      int n_cells, const float* positions_host, const float* colors4_host, int n_points,
                      float* importance, cpm_stream stream) {
    return cpm_importance_tf_occupancy(ctx, minmax2, prev_minmax2, volume_diff, n_cells, positions_host, colors4_host, n_points, importance,
                                       nullptr, stream);
}

int cpm_importance_tf_occupancy(cpm_ctx* ctx, const uint16_t* minmax2, const uint16_t* prev_minmax2, const float* volume_diff,
                                int n_cells, const float* positions_host, const float* colors4_host, int n_points,
                                float* importance, uint32_t* occupancy, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_cells >= 0, "cpm_importance_tf: n_cells < 0");
    CPM_REQUIRE(ctx, n_points >= 2 && n_points <= 4096, "cpm_importance_tf: n_points must be in [2, 4096]");
    CPM_REQUIRE(ctx, positions_host && colors4_host, "cpm_importance_tf: null TF points");
    CPM_REQUIRE(ctx, (prev_minmax2 == nullptr) == (volume_diff == nullptr), "cpm_importance_tf: prev_minmax2 and volume_diff go together");
    if (n_cells == 0) return CPM_OK;
    CPM_REQUIRE(ctx, minmax2 && importance, "cpm_importance_tf: null buffer");
    hipStream_t s = (hipStream_t)stream;
    if (n_points <= kTfArgPoints) {  // the usual case: the points ride in the kernel arguments
        TfPointArgs P;
        memset(&P, 0, sizeof(P));
        memcpy(P.col, colors4_host, (size_t)n_points * 4 * sizeof(float));
        memcpy(P.pos, positions_host, (size_t)n_points * sizeof(float));
        CPM_LAUNCH(ctx, importance_tf_args_kernel, dim3(div_up(n_cells, 256)), dim3(256), 0, s, minmax2, prev_minmax2, volume_diff, n_cells, P,
                   n_points, importance, occupancy);
        CPM_LAUNCH_CHECK(ctx, "importance_tf_args_kernel");
        return CPM_OK;
    }
    // colours first (16-byte aligned), then positions
    float* dev = (float*)scratch(ctx, CPM_SCR_SMALL, (size_t)n_points * 5 * sizeof(float));
    if (!dev) return CPM_ERR_OUT_OF_MEMORY;
    CPM_HIP_CHECK(ctx, hipMemcpyAsync(dev, colors4_host, (size_t)n_points * 4 * sizeof(float), hipMemcpyHostToDevice, s));
    CPM_HIP_CHECK(ctx, hipMemcpyAsync(dev + (size_t)n_points * 4, positions_host, (size_t)n_points * sizeof(float), hipMemcpyHostToDevice, s));
    CPM_LAUNCH(ctx, importance_tf_kernel, dim3(div_up(n_cells, 256)), dim3(256), 0, s, minmax2, prev_minmax2, volume_diff,
                       n_cells, dev + (size_t)n_points * 4, reinterpret_cast<const float4*>(dev), n_points, importance, occupancy);
    CPM_LAUNCH_CHECK(ctx, "importance_tf_kernel");
    CPM_HIP_CHECK(ctx, hipStreamSynchronize(s));  // the caller's host arrays are consumed when this returns
    return CPM_OK;
}

int cpm_photon_importance(cpm_ctx* ctx, const float* importance_grid, const int32_t grid_dims[3], const float cell_size[3],
                          const float texture_to_index[16], const float* photons8, int photon_offset,
                          const float* light_samples8, const float* isect2, int n_light_samples, int max_interactions,
                          int total_photons, int fix_exit_point, uint32_t* importances, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, grid_dims && cell_size && texture_to_index, "cpm_photon_importance: null argument");
    CPM_REQUIRE(ctx, n_light_samples >= 0 && photon_offset >= 0 && max_interactions >= 1 && total_photons >= 0,
                "cpm_photon_importance: bad size");
    if (n_light_samples == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importance_grid && photons8 && light_samples8 && isect2 && importances, "cpm_photon_importance: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_photon_importance");
    CPM_REQUIRE_ALIGNED16(ctx, light_samples8, "cpm_photon_importance");
    ImpGrid G;
    G.grid = importance_grid;
    G.rec = rec_layout(ctx, photons8, (size_t)total_photons * (size_t)max_interactions);
    for (int a = 0; a < 3; ++a) {
        CPM_REQUIRE(ctx, grid_dims[a] >= 1 && cell_size[a] > 0.f, "cpm_photon_importance: grid dims / cell size");
        G.dims[a] = grid_dims[a];
        set_cell(G, a, cell_size[a]);
    }
    if (!affine_from_matrix(texture_to_index, G.t2i))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_photon_importance", "textureToIndex must be scale + translate");
    CPM_LAUNCH(ctx, photon_importance_kernel, dim3(div_up(n_light_samples, 256)), dim3(256), 0, (hipStream_t)stream, G,
                       photons8, photon_offset, light_samples8, isect2, n_light_samples, max_interactions, total_photons,
                       fix_exit_point, importances);
    CPM_LAUNCH_CHECK(ctx, "photon_importance_kernel");
    return CPM_OK;
}

int cpm_photon_importance_equal(cpm_ctx* ctx, int photon_offset, int n_light_samples, int percentage, int iteration,
                                uint32_t* importances, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_light_samples >= 0 && photon_offset >= 0, "cpm_photon_importance_equal: bad size");
    CPM_REQUIRE(ctx, percentage >= 1 && percentage <= 100, "cpm_photon_importance_equal: percentage must be in [1, 100]");
    if (n_light_samples == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importances, "cpm_photon_importance_equal: null buffer");
    CPM_LAUNCH(ctx, photon_importance_equal_kernel, dim3(div_up(n_light_samples, 256)), dim3(256), 0, (hipStream_t)stream,
                       photon_offset, n_light_samples, percentage, iteration, importances);
    CPM_LAUNCH_CHECK(ctx, "photon_importance_equal_kernel");
    return CPM_OK;
}

int cpm_reset_importance(cpm_ctx* ctx, uint32_t* importances, size_t offset, size_t n, cpm_stream stream) {
    CPM_ENTER(ctx);
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importances, "cpm_reset_importance: null buffer");
    CPM_LAUNCH(ctx, fill_u32_kernel, dim3(div_up((long long)n, 256)), dim3(256), 0, (hipStream_t)stream,
                       importances + offset, n, 2147483647u);
    CPM_LAUNCH_CHECK(ctx, "fill_u32_kernel");
    return CPM_OK;
}

int cpm_select_recompute(cpm_ctx* ctx, uint32_t* importances, size_t n, uint32_t* indices_out, int32_t* n_changed_dev,
                         cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_changed_dev, "cpm_select_recompute: null counter");
    hipStream_t s = (hipStream_t)stream;
    CPM_HIP_CHECK(ctx, hipMemsetAsync(n_changed_dev, 0, sizeof(int32_t), s));
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importances && indices_out, "cpm_select_recompute: null buffer");
    CPM_LAUNCH(ctx, threshold_count_iota_kernel, dim3(div_up((long long)n, 256)), dim3(256), 0, s, importances, n,
                       indices_out, n_changed_dev);
    CPM_LAUNCH_CHECK(ctx, "threshold_count_iota_kernel");
    // keys are <= 0x7fffffff: 31 significant bits
    return cpm::radix_sort(ctx, importances, indices_out, n, 31, s, nullptr, nullptr, nullptr, nullptr, false);
}

int cpm_select_changed(cpm_ctx* ctx, const uint32_t* importances, size_t n, uint32_t* indices_out, int32_t* n_changed_dev,
                       cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n_changed_dev, "cpm_select_changed: null counter");
    CPM_REQUIRE(ctx, n < (1ull << 31), "cpm_select_changed: n must be < 2^31");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { CPM_HIP_CHECK(ctx, hipMemsetAsync(n_changed_dev, 0, sizeof(int32_t), s)); return CPM_OK; }
    CPM_REQUIRE(ctx, importances && indices_out, "cpm_select_changed: null buffer");
    if (n == 1) {
        CPM_LAUNCH(ctx, select_single_kernel, dim3(1), dim3(1), 0, s, importances, indices_out, n_changed_dev);
        CPM_LAUNCH_CHECK(ctx, "select_single_kernel");
        return CPM_OK;
    }
    if (ctx->dbg.select_partition) {
        uint32_t num_tiles = (uint32_t)div_up((long long)n, 2048);
        if (num_tiles > (uint32_t)kPartMaxTiles) num_tiles = kPartMaxTiles;
        const uint32_t per_tile = (uint32_t)div_up(div_up((long long)n, num_tiles), 256) * 256u;
        num_tiles = (uint32_t)div_up((long long)n, per_tile);
        uint32_t* counts = (uint32_t*)scratch(ctx, CPM_SCR_MISC, kPartMaxTiles * sizeof(uint32_t));
        if (!counts) return CPM_ERR_OUT_OF_MEMORY;
        CPM_LAUNCH(ctx, partition_count_kernel, dim3(num_tiles), dim3(256), 0, s, importances, (uint32_t)n, per_tile, counts);
        CPM_LAUNCH_CHECK(ctx, "partition_count_kernel");
        CPM_LAUNCH(ctx, partition_write_kernel, dim3(num_tiles), dim3(256), 0, s, importances, (uint32_t)n, per_tile, counts, num_tiles,
                   indices_out, n_changed_dev);
        CPM_LAUNCH_CHECK(ctx, "partition_write_kernel");
        return CPM_OK;
    }
    uint32_t* flags = (uint32_t*)scratch(ctx, CPM_SCR_MISC, n * sizeof(uint32_t));
    if (!flags) return CPM_ERR_OUT_OF_MEMORY;
    CPM_LAUNCH(ctx, changed_flag_iota_kernel, dim3(div_up((long long)n, 256)), dim3(256), 0, s, importances, n, flags, indices_out);
    CPM_LAUNCH_CHECK(ctx, "changed_flag_iota_kernel");
    uint32_t *rk = nullptr, *rv = nullptr;
    int rc = cpm::radix_sort(ctx, flags, indices_out, n, 1, s, &rk, &rv, nullptr, nullptr, false);
    if (rc) return rc;
    if (rv != indices_out) CPM_HIP_CHECK(ctx, hipMemcpyAsync(indices_out, rv, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    const uint32_t* totals = cpm::sort_last_digit_totals(ctx, n);
    CPM_REQUIRE(ctx, totals, "cpm_select_changed: not available in the onesweep sort test mode");
    CPM_HIP_CHECK(ctx, hipMemcpyAsync(n_changed_dev, totals, sizeof(int32_t), hipMemcpyDeviceToDevice, s));  // keys with flag 0
    return CPM_OK;
}


// ---- the fused selection ---------------------------------------------------------------------------------------

int cpm_selection_create(cpm_ctx* ctx, size_t max_photons, cpm_selection** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, out, "cpm_selection_create: null out");
    *out = nullptr;
    CPM_REQUIRE(ctx, max_photons >= 1 && max_photons < (1ull << 31), "cpm_selection_create: 1 <= max_photons < 2^31");
    cpm_selection* s = new (std::nothrow) cpm_selection();
    if (!s) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_selection_create", "host allocation");
    s->max_photons = max_photons;
    // tiles of K * 256 photons, K in {1, 2, 4, 8}: about 2048 tiles at the sizes of the path (1 M photons: 512 per tile)
    uint32_t k = 1;
#ifndef CPM_SEL_TILES
#define CPM_SEL_TILES 2048   // (tuning: tools/build_variant.sh)
#endif
    while (k < 8 && max_photons / (256ull * k) > CPM_SEL_TILES) k *= 2;
    s->per_tile = 256u * k;
    // (sized for the smallest tiles in use, cpm_photon_importance_retrace's) + one partial tile per further light
    s->max_tiles = (uint32_t)((max_photons + kRetraceTile - 1) / kRetraceTile) + 64u;
    bool ok = hipMalloc(&s->tile, (size_t)s->max_tiles * sizeof(uint2)) == hipSuccess &&
              hipMalloc(&s->local, max_photons * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc(&s->count_dev, sizeof(int32_t)) == hipSuccess &&
              hipHostMalloc(&s->mailbox, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void**)&s->mailbox_dev, s->mailbox, 0) == hipSuccess &&
              hipMemset(s->count_dev, 0, sizeof(int32_t)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        cpm_selection_destroy(ctx, s);
        return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_selection_create", "device / pinned allocation");
    }
    *s->mailbox = 0ull;
    *out = s;
    return CPM_OK;
}

void cpm_selection_destroy(cpm_ctx* ctx, cpm_selection* s) {
    if (!s) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (s->last_stream || s->finished) (void)hipStreamSynchronize(s->last_stream);  // the mailbox must outlive its writer
    if (s->tile) (void)hipFree(s->tile);
    if (s->local) (void)hipFree(s->local);
    if (s->count_dev) (void)hipFree(s->count_dev);
    for (auto& o : s->orders) {
        if (o.order) (void)hipFree(o.order);
        if (o.cost) (void)hipFree(o.cost);
        if (o.keys) (void)hipFree(o.keys);
    }
    if (s->mask) (void)hipFree(s->mask);
    if (s->mailbox) (void)hipHostFree(s->mailbox);
    delete s;
}

int cpm_selection_set_occupancy(cpm_ctx* ctx, cpm_selection* s, const float* importance_grid, const uint32_t* occupancy) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s, "cpm_selection_set_occupancy: null selection");
    CPM_REQUIRE(ctx, (importance_grid != nullptr) == (occupancy != nullptr), "cpm_selection_set_occupancy: grid and bits go together");
    s->given_mask = occupancy;
    s->given_mask_grid = importance_grid;
    return CPM_OK;
}

int cpm_selection_begin(cpm_ctx* ctx, cpm_selection* s) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s, "cpm_selection_begin: null selection");
    s->n_tiles = 0;
    s->n_launches = 0;
    s->pending_orders.clear();
    s->finished = false;
    s->failed = false;
    // every kMeasureEvery-th selection records what its retrace tiles cost (the photon paths the cost depends on change slowly)
    constexpr uint32_t kMeasureEvery = 32;
    s->measuring = s->selections % kMeasureEvery == 0;
    ++s->selections;
    return CPM_OK;
}

}  // extern "C"

namespace {
// appends the tiles of one light's launch; *first = its first tile
int selection_append(cpm_ctx* ctx, cpm_selection* s, int photon_offset, int n_light_samples, uint32_t* first, uint32_t* tiles,
                     uint32_t per_tile = 0) {
    if (per_tile == 0) per_tile = s->per_tile;
    CPM_REQUIRE(ctx, !s->finished, "cpm_photon_importance_select: call cpm_selection_begin first");
    CPM_REQUIRE(ctx, (size_t)photon_offset + (size_t)n_light_samples <= s->max_photons, "cpm_photon_importance_select: photons exceed the selection's max_photons");
    *tiles = (uint32_t)div_up(n_light_samples, per_tile);
    CPM_REQUIRE(ctx, s->n_tiles + *tiles <= s->max_tiles, "cpm_photon_importance_select: too many lights for this selection");
    *first = s->n_tiles;
    s->n_tiles += *tiles;
    return CPM_OK;
}

// What a select / retrace call appended is taken back unless the call reaches its end: tiles no launch wrote must never reach
// selection_compact_kernel (it would sum uninitialised counts and copy garbage indices).  A selection with a failed call is
// marked: cpm_selection_finish then publishes a count of 0 and reports the failure.
struct AppendGuard {
    cpm_selection* s; uint32_t n_tiles0, n_launches0; bool ok = false;
    AppendGuard(cpm_selection* sel, uint32_t first) : s(sel), n_tiles0(first), n_launches0(sel->n_launches) {}
    ~AppendGuard() { if (!ok) { s->n_tiles = n_tiles0; s->n_launches = n_launches0; s->failed = true; } }
};

#define CPM_INJECTED_SELECT_FAILURE(ctx, who)                                                                          \
    do {                                                                                                             \
        if ((ctx)->dbg.fail_next_select) {                                                                           \
            (ctx)->dbg.fail_next_select = 0;                                                                         \
            return set_error((ctx), CPM_ERR_DEVICE, who, "injected failure (cpm_debug_fail_next_select)");           \
        }                                                                                                            \
    } while (0)

// LDS a select / retrace workgroup may ask for without an opt-in (occupancy bits + the tracer's LUTs + the static arrays)
constexpr size_t kSelectLdsBudget = 64 * 1024 - 2048;

// the occupancy bits a select / retrace launch stages in LDS: the caller's (cpm_selection_set_occupancy: made by the launch
// that made the grid) or, failing that, built here by one small launch
int selection_mask(cpm_ctx* ctx, cpm_selection* s, const float* importance_grid, long long cells, size_t words, hipStream_t st, const uint32_t** mask_out) {
    if (s->given_mask && s->given_mask_grid == importance_grid) { *mask_out = s->given_mask; return CPM_OK; }
    if (s->mask_words < words) {
        if (s->mask) { CPM_HIP_CHECK(ctx, hipStreamSynchronize(st)); (void)hipFree(s->mask); s->mask = nullptr; s->mask_words = 0; }
        CPM_HIP_CHECK(ctx, hipMalloc(&s->mask, words * 4));
        s->mask_words = words;
    }
    CPM_LAUNCH(ctx, importance_mask_kernel, dim3(div_up(cells, 256)), dim3(256), 0, st, importance_grid, (uint32_t)cells, s->mask);
    CPM_LAUNCH_CHECK(ctx, "importance_mask_kernel");
    *mask_out = s->mask;
    return CPM_OK;
}

template <int MODE, bool MASK>
void launch_select(cpm_ctx* ctx, cpm_selection* s, hipStream_t st, uint32_t tiles, size_t lds, const ImpGrid& G, const uint32_t* mask_bits, uint32_t mask_words,
                   const float* photons8, int photon_offset, const float* ls, const float* isect, int n_light_samples, int max_interactions,
                   int total_photons, int fix_exit_point, int pct, int iter, uint32_t* importances, const SelTiles& S) {
    const dim3 grid(tiles), block(256);
#define CPM_SEL_LAUNCH(KK)                                                                                                             \
    CPM_LAUNCH(ctx, (importance_select_kernel<MODE, MASK, KK>), grid, block, lds, st, G, mask_bits, mask_words, photons8, photon_offset, ls, isect, \
               n_light_samples, max_interactions, total_photons, fix_exit_point, pct, iter, importances, S)
    switch (s->per_tile / 256u) {
        case 1: CPM_SEL_LAUNCH(1); break;
        case 2: CPM_SEL_LAUNCH(2); break;
        case 4: CPM_SEL_LAUNCH(4); break;
        default: CPM_SEL_LAUNCH(8); break;
    }
#undef CPM_SEL_LAUNCH
}
}  // namespace

extern "C" {

int cpm_photon_importance_select(cpm_ctx* ctx, cpm_selection* s, const float* importance_grid, const int32_t grid_dims[3],
                                 const float cell_size[3], const float texture_to_index[16], const float* photons8, int photon_offset,
                                 const float* light_samples8, const float* isect2, int n_light_samples, int max_interactions,
                                 int total_photons, int fix_exit_point, uint32_t* importances, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s, "cpm_photon_importance_select: null selection");
    CPM_REQUIRE(ctx, grid_dims && cell_size && texture_to_index, "cpm_photon_importance_select: null argument");
    CPM_REQUIRE(ctx, n_light_samples >= 0 && photon_offset >= 0 && max_interactions >= 1 && total_photons >= 0,
                "cpm_photon_importance_select: bad size");
    if (n_light_samples == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importance_grid && photons8 && light_samples8 && isect2 && importances, "cpm_photon_importance_select: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_photon_importance_select");
    CPM_REQUIRE_ALIGNED16(ctx, light_samples8, "cpm_photon_importance_select");
    ImpGrid G;
    G.grid = importance_grid;
    G.rec = rec_layout(ctx, photons8, (size_t)total_photons * (size_t)max_interactions);
    unsigned long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        CPM_REQUIRE(ctx, grid_dims[a] >= 1 && cell_size[a] > 0.f, "cpm_photon_importance_select: grid dims / cell size");
        G.dims[a] = grid_dims[a];
        set_cell(G, a, cell_size[a]);
        cells *= (unsigned long long)grid_dims[a];
    }
    CPM_REQUIRE(ctx, cells < (1ull << 31), "cpm_photon_importance_select: grid too large");
    if (!affine_from_matrix(texture_to_index, G.t2i))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_photon_importance_select", "textureToIndex must be scale + translate");
    uint32_t first = 0, tiles = 0;
    int rc = selection_append(ctx, s, photon_offset, n_light_samples, &first, &tiles);
    if (rc) return rc;
    AppendGuard guard(s, first);
    CPM_INJECTED_SELECT_FAILURE(ctx, "cpm_photon_importance_select");
    hipStream_t st = (hipStream_t)stream;
    s->last_stream = st;
    SelTiles S{ s->tile, s->local, first, s->per_tile };
    // occupancy mask of the grid: 1 bit per cell, staged in LDS by every workgroup (4 KiB for the 32^3 grid of a 256^3 volume);
    // grids whose bits do not fit the LDS budget are walked without it
    const size_t words = (size_t)((cells + 63) / 64) * 2;
    const bool use_mask = words * 4 <= kSelectLdsBudget;
    if (use_mask) {
        const uint32_t* mask_bits = nullptr;
        rc = selection_mask(ctx, s, importance_grid, (long long)cells, words, st, &mask_bits);
        if (rc) return rc;
        launch_select<0, true>(ctx, s, st, tiles, words * 4, G, mask_bits, (uint32_t)words, photons8, photon_offset, light_samples8, isect2, n_light_samples,
                               max_interactions, total_photons, fix_exit_point, 1, 0, importances, S);
    } else {
        launch_select<0, false>(ctx, s, st, tiles, 0, G, nullptr, 0u, photons8, photon_offset, light_samples8, isect2, n_light_samples, max_interactions,
                                total_photons, fix_exit_point, 1, 0, importances, S);
    }
    CPM_LAUNCH_CHECK(ctx, "importance_select_kernel");
    guard.ok = true;
    return CPM_OK;
}

}  // extern "C"

namespace {
// cpm_photon_importance_retrace (one light: the span made of its arguments) and cpm_photon_importance_retrace_lights (several: one launch
// over all their tiles, the MULTI kernels)
int retrace_impl(cpm_ctx* ctx, cpm_selection* s, const float* importance_grid, const int32_t grid_dims[3], const float cell_size[3],
                 const float texture_to_index[16], const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                 const cpm_trace_params* params, const cpm_light_span* lights, int n_lights, int fix_exit_point, uint32_t* importances,
                 uint32_t* rng_state, float* photons8, float* old_photons8, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s, "cpm_photon_importance_retrace: null selection");
    CPM_REQUIRE(ctx, grid_dims && cell_size && texture_to_index && params, "cpm_photon_importance_retrace: null argument");
    CPM_REQUIRE(ctx, lights && n_lights >= 1 && n_lights <= CPM_MAX_TRACE_LIGHTS, "cpm_photon_importance_retrace_lights: 1 .. CPM_MAX_TRACE_LIGHTS lights");
    cpm_trace_params p = *params;
    long long n_all = 0;
    for (int l = 0; l < n_lights; ++l) {
        CPM_REQUIRE(ctx, lights[l].n_light_samples >= 0 && lights[l].photon_offset >= 0, "cpm_photon_importance_retrace: negative size");
        n_all += lights[l].n_light_samples;
    }
    if (n_lights > 1) { p.photon_offset = 0; p.n_light_samples = 0; }  // (the per-light fields come from the spans)
    tracer::TraceArgs A;
    size_t lut_bytes = 0;
    int rc = cpm::make_trace_args(ctx, vol, tf, tf_scattering, aabb, &p, A, lut_bytes);
    if (rc) return rc;
    CPM_REQUIRE(ctx, !(p.flags & CPM_TRACE_PROGRESSIVE), "cpm_photon_importance_retrace: a correlated re-trace does not write the RNG state back");
    CPM_REQUIRE(ctx, !(p.flags & CPM_TRACE_PHOTONS_PLANAR) || rec_layout(ctx, photons8, 1).stride == 1u,
                "cpm_photon_importance_retrace: the importance pass reads the records as the buffer was described (cpm_records_describe) or in the context's layout, not in a call's");
    if (n_all == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importance_grid && photons8 && importances && rng_state && old_photons8, "cpm_photon_importance_retrace: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_photon_importance_retrace");
    CPM_REQUIRE_ALIGNED16(ctx, old_photons8, "cpm_photon_importance_retrace");
    CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(rng_state) & 7u) == 0, "cpm_photon_importance_retrace: rng_state must be 8-byte aligned");
    for (int l = 0; l < n_lights; ++l) {
        const cpm_light_span& L = lights[l];
        if (L.n_light_samples == 0) continue;
        CPM_REQUIRE(ctx, L.light_samples8 && L.isect2, "cpm_photon_importance_retrace: null buffer");
        CPM_REQUIRE(ctx, (long long)L.photon_offset + L.n_light_samples <= (long long)p.total_photons,
                    "cpm_photon_importance_retrace: photon_offset + n_light_samples exceeds total_photons");
        CPM_REQUIRE_ALIGNED16(ctx, L.light_samples8, "cpm_photon_importance_retrace");
        CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(L.isect2) & 7u) == 0, "cpm_photon_importance_retrace: isect2 must be 8-byte aligned");
    }
    // a stale footprint copy (a mixed time step) has linear-block kernels for one light only: several lights go one by one
    if (n_lights > 1 && vol->quads_stale) {
        for (int l = 0; l < n_lights; ++l) {
            cpm_trace_params pl = *params;
            pl.photon_offset = lights[l].photon_offset;
            pl.n_light_samples = lights[l].n_light_samples;
            rc = retrace_impl(ctx, s, importance_grid, grid_dims, cell_size, texture_to_index, vol, tf, tf_scattering, aabb, &pl, lights + l, 1, fix_exit_point,
                              importances, rng_state, photons8, old_photons8, stream);
            if (rc) return rc;
        }
        return CPM_OK;
    }
    ImpGrid G;
    G.grid = importance_grid;
    G.rec = rec_layout(ctx, photons8, (size_t)p.total_photons * (size_t)p.max_interactions);
    unsigned long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        CPM_REQUIRE(ctx, grid_dims[a] >= 1 && cell_size[a] > 0.f, "cpm_photon_importance_retrace: grid dims / cell size");
        G.dims[a] = grid_dims[a];
        set_cell(G, a, cell_size[a]);
        cells *= (unsigned long long)grid_dims[a];
    }
    CPM_REQUIRE(ctx, cells < (1ull << 31), "cpm_photon_importance_retrace: grid too large");
    if (!affine_from_matrix(texture_to_index, G.t2i))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_photon_importance_retrace", "textureToIndex must be scale + translate");
    // the lights' tiles, one light after the other (what a call per light appends)
    const uint32_t tiles_before = s->n_tiles;
    uint32_t first = 0, tiles = 0;
    for (int l = 0; l < n_lights; ++l) {
        uint32_t f = 0, tl = 0;
        rc = selection_append(ctx, s, lights[l].photon_offset, lights[l].n_light_samples, &f, &tl, kRetraceTile);
        if (rc) { s->n_tiles = tiles_before; return rc; }
        if (l == 0) first = f;
        A.span[l] = { lights[l].light_samples8, lights[l].isect2, lights[l].n_light_samples, lights[l].photon_offset, (int)tiles };
        tiles += tl;
    }
    A.n_spans = n_lights;
    AppendGuard guard(s, first);
    CPM_INJECTED_SELECT_FAILURE(ctx, "cpm_photon_importance_retrace");
    hipStream_t st = (hipStream_t)stream;
    s->last_stream = st;
    SelTiles S{ s->tile, s->local, first, kRetraceTile };
    // this launch's tile order (one per light: the launches of a selection come in the same order every time)
    const uint32_t ordinal = s->n_launches++;
    if (s->orders.size() <= ordinal) s->orders.resize(ordinal + 1);
    cpm_selection::LaunchOrder& lo = s->orders[ordinal];
    if (lo.n_tiles != tiles) {
        if (lo.order) (void)hipFree(lo.order);
        if (lo.cost) (void)hipFree(lo.cost);
        if (lo.keys) (void)hipFree(lo.keys);
        lo = cpm_selection::LaunchOrder();
        const bool ok = hipMalloc(&lo.order, (size_t)tiles * 4) == hipSuccess && hipMalloc(&lo.cost, (size_t)tiles * 4) == hipSuccess &&
                        hipMalloc(&lo.keys, (size_t)tiles * 4) == hipSuccess && hipMemsetAsync(lo.cost, 0, (size_t)tiles * 4, st) == hipSuccess;
        if (!ok) {
            if (lo.order) (void)hipFree(lo.order);
            if (lo.cost) (void)hipFree(lo.cost);
            if (lo.keys) (void)hipFree(lo.keys);
            lo = cpm_selection::LaunchOrder();
            return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_photon_importance_retrace", "tile order buffers");
        }
        lo.n_tiles = tiles;
        lo.fresh = true;  // no order yet: index order, and this launch is measured whatever the selection's turn
    }
    const uint32_t* tile_order = lo.fresh ? nullptr : lo.order;
    uint32_t* tile_cost = (s->measuring || lo.fresh) ? lo.cost : nullptr;
    A.light_samples = lights[0].light_samples8;
    A.isect = lights[0].isect2;
    A.rng = rng_state;
    A.photons = photons8;
    A.rec_stride = G.rec.stride; A.rec_b = G.rec.b;  // (the context's layout: what the importance pass reads, the re-trace writes)
    A.n_threads = lights[0].n_light_samples;
    A.p.n_light_samples = lights[0].n_light_samples;
    A.p.photon_offset = lights[0].photon_offset;
    const size_t words = (size_t)((cells + 63) / 64) * 2;
    const bool use_mask = words * 4 + lut_bytes <= kSelectLdsBudget;  // (the tracer's LUTs share the dynamic LDS)
    const uint32_t* mask_bits = nullptr;
    if (use_mask) {
        rc = selection_mask(ctx, s, importance_grid, (long long)cells, words, st, &mask_bits);
        if (rc) return rc;
    }
    const uint32_t mw = use_mask ? (uint32_t)words : 0u;
    const size_t lds = (size_t)mw * 4 + lut_bytes;
    const bool single = p.max_interactions == 1 && !(p.flags & CPM_TRACE_NO_SINGLE_SCATTERING);
    const dim3 grid(tiles), block(256);
    bool linear = false;  // a stale footprint copy (mixed time step): the re-traces read the volume's linear block
    rc = cpm::trace_volume_source(ctx, vol, true, st, A, &linear);
    if (rc) return rc;
#define CPM_RETRACE_LAUNCH_L(DT, L)                                                                                                              \
    do {                                                                                                                                         \
        if (use_mask && single) CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, true, true, L>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);   \
        else if (use_mask) CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, true, false, L>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);       \
        else if (single) CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, false, true, L>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);        \
        else CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, false, false, L>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);                   \
    } while (0)
#define CPM_RETRACE_LAUNCH_M(DT)                                                                                                                 \
    do {                                                                                                                                         \
        if (use_mask && single) CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, true, true, false, true>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);   \
        else if (use_mask) CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, true, false, false, true>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);       \
        else if (single) CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, false, true, false, true>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);        \
        else CPM_LAUNCH(ctx, (importance_retrace_kernel<DT, false, false, false, true>), grid, block, lds, st, G, mask_bits, mw, fix_exit_point, importances, S, A, old_photons8, tile_order, tile_cost);                   \
    } while (0)
#define CPM_RETRACE_LAUNCH(DT)                                  \
    do {                                                        \
        if (n_lights > 1) CPM_RETRACE_LAUNCH_M(DT);             \
        else if (linear) CPM_RETRACE_LAUNCH_L(DT, true);        \
        else CPM_RETRACE_LAUNCH_L(DT, false);                   \
    } while (0)
    switch (vol->desc.dtype) {
        case CPM_U8: CPM_RETRACE_LAUNCH(CPM_U8); break;
        case CPM_U16: CPM_RETRACE_LAUNCH(CPM_U16); break;
        default: CPM_RETRACE_LAUNCH(CPM_F32); break;
    }
#undef CPM_RETRACE_LAUNCH_L
#undef CPM_RETRACE_LAUNCH_M
#undef CPM_RETRACE_LAUNCH
    CPM_LAUNCH_CHECK(ctx, "importance_retrace_kernel");
    if (tile_cost) s->pending_orders.push_back(ordinal);  // re-sorted behind the compaction (cpm_selection_finish)
    guard.ok = true;
    return CPM_OK;
}
}  // namespace

extern "C" {

int cpm_photon_importance_retrace(cpm_ctx* ctx, cpm_selection* s, const float* importance_grid, const int32_t grid_dims[3],
                                  const float cell_size[3], const float texture_to_index[16], const cpm_volume* vol, const cpm_tf* tf,
                                  const cpm_tf* tf_scattering, const float aabb[8], const cpm_trace_params* params,
                                  const float* light_samples8, const float* isect2, int fix_exit_point, uint32_t* importances,
                                  uint32_t* rng_state, float* photons8, float* old_photons8, cpm_stream stream) {
    if (ctx && !params) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_photon_importance_retrace", "null params");
    const cpm_light_span one = { light_samples8, isect2, params ? params->n_light_samples : 0, params ? params->photon_offset : 0 };
    return retrace_impl(ctx, s, importance_grid, grid_dims, cell_size, texture_to_index, vol, tf, tf_scattering, aabb, params, &one, 1, fix_exit_point,
                        importances, rng_state, photons8, old_photons8, stream);
}

int cpm_photon_importance_retrace_lights(cpm_ctx* ctx, cpm_selection* s, const float* importance_grid, const int32_t grid_dims[3],
                                         const float cell_size[3], const float texture_to_index[16], const cpm_volume* vol, const cpm_tf* tf,
                                         const cpm_tf* tf_scattering, const float aabb[8], const cpm_trace_params* params,
                                         const cpm_light_span* lights, int n_lights, int fix_exit_point, uint32_t* importances,
                                         uint32_t* rng_state, float* photons8, float* old_photons8, cpm_stream stream) {
    return retrace_impl(ctx, s, importance_grid, grid_dims, cell_size, texture_to_index, vol, tf, tf_scattering, aabb, params, lights, n_lights, fix_exit_point,
                        importances, rng_state, photons8, old_photons8, stream);
}

int cpm_photon_importance_equal_select(cpm_ctx* ctx, cpm_selection* s, int photon_offset, int n_light_samples, int percentage,
                                       int iteration, uint32_t* importances, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s, "cpm_photon_importance_equal_select: null selection");
    CPM_REQUIRE(ctx, n_light_samples >= 0 && photon_offset >= 0, "cpm_photon_importance_equal_select: bad size");
    CPM_REQUIRE(ctx, percentage >= 1 && percentage <= 100, "cpm_photon_importance_equal_select: percentage must be in [1, 100]");
    if (n_light_samples == 0) return CPM_OK;
    CPM_REQUIRE(ctx, importances, "cpm_photon_importance_equal_select: null buffer");
    uint32_t first = 0, tiles = 0;
    int rc = selection_append(ctx, s, photon_offset, n_light_samples, &first, &tiles);
    if (rc) return rc;
    AppendGuard guard(s, first);
    CPM_INJECTED_SELECT_FAILURE(ctx, "cpm_photon_importance_equal_select");
    hipStream_t st = (hipStream_t)stream;
    s->last_stream = st;
    SelTiles S{ s->tile, s->local, first, s->per_tile };
    ImpGrid G = {};
    launch_select<1, false>(ctx, s, st, tiles, 0, G, nullptr, 0u, nullptr, photon_offset, nullptr, nullptr, n_light_samples, 1, 0, 0, percentage, iteration,
                            importances, S);
    CPM_LAUNCH_CHECK(ctx, "importance_select_kernel");
    guard.ok = true;
    return CPM_OK;
}

int cpm_selection_finish(cpm_ctx* ctx, cpm_selection* s, uint32_t* indices_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s, "cpm_selection_finish: null selection");
    CPM_REQUIRE(ctx, !s->finished, "cpm_selection_finish: already finished (cpm_selection_begin starts the next one)");
    CPM_REQUIRE(ctx, indices_out || s->n_tiles == 0, "cpm_selection_finish: null indices_out");
    hipStream_t st = (hipStream_t)stream;
    s->last_stream = st;
    s->finished = true;
    ++s->epoch;
    // a selection one of whose calls failed selects nothing: what follows on the stream (re-trace, delta splat) then does nothing
    // either, and the caller learns of it here -- and falls back to a full frame
    const uint32_t n_tiles = s->failed ? 0u : s->n_tiles;
    // (with no tiles the one workgroup only publishes a count of 0)
    CPM_LAUNCH(ctx, selection_compact_kernel, dim3(n_tiles ? (unsigned)div_up(n_tiles, kCompactGroup) : 1u), dim3(256), 0, st, s->tile, n_tiles,
               s->local, indices_out, s->count_dev, s->mailbox_dev, s->epoch);
    CPM_LAUNCH_CHECK(ctx, "selection_compact_kernel");
    if (s->failed) {
        s->pending_orders.clear();
        return set_error(ctx, CPM_ERR_DEVICE, "cpm_selection_finish", "a select / retrace call of this selection failed: nothing is selected");
    }
    // the measured launches' tile orders, costliest first -- behind everything the host waits for (a key launch + the radix sort)
    for (uint32_t ordinal : s->pending_orders) {
        cpm_selection::LaunchOrder& lo = s->orders[ordinal];
        CPM_LAUNCH(ctx, retrace_order_keys_kernel, dim3((unsigned)div_up(lo.n_tiles, 256)), dim3(256), 0, st, lo.cost, lo.keys, lo.order, lo.n_tiles);
        CPM_LAUNCH_CHECK(ctx, "retrace_order_keys_kernel");
        int rc = cpm_sort_pairs(ctx, lo.keys, lo.order, lo.n_tiles, (int)kCostBits, stream);
        if (rc) return rc;
        lo.fresh = false;  // from the next launch on `order` is a permutation of the tiles
    }
    s->pending_orders.clear();
    return CPM_OK;
}

const int32_t* cpm_selection_count_device(const cpm_selection* s) { return s ? s->count_dev : nullptr; }

int cpm_selection_count(cpm_ctx* ctx, cpm_selection* s, int32_t* n_out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, s && n_out, "cpm_selection_count: null argument");
    *n_out = 0;
    if (s->epoch == 0) return CPM_OK;  // nothing was ever selected
    // the compaction kernel of the last finish writes (epoch << 32 | count) into pinned host memory; everything enqueued
    // behind it keeps running while the host reads it
    const volatile unsigned long long* mb = s->mailbox;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long v = __atomic_load_n(mb, __ATOMIC_ACQUIRE);
        if ((uint32_t)(v >> 32) == s->epoch) { *n_out = (int32_t)(uint32_t)v; return CPM_OK; }
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        __builtin_ia32_pause();
    }
    // the mailbox did not arrive (a stream that is not running?): fall back to the stream itself
    CPM_HIP_CHECK(ctx, hipStreamSynchronize(s->last_stream));
    const unsigned long long v = __atomic_load_n(mb, __ATOMIC_ACQUIRE);
    if ((uint32_t)(v >> 32) == s->epoch) { *n_out = (int32_t)(uint32_t)v; return CPM_OK; }
    CPM_HIP_CHECK(ctx, hipMemcpy(n_out, s->count_dev, sizeof(int32_t), hipMemcpyDeviceToHost));
    return CPM_OK;
}

}  // extern "C"
