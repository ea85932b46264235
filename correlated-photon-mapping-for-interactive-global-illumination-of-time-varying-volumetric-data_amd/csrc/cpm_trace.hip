// cpm_trace.hip -- Woodcock (delta-tracking) photon tracer (R3, R4, R5).
//
// Replaces photonTracerKernel (ref progressivephotonmapping/cl/photontracer.cl:69-216) and
// woodcockTracking (ref cl/transmittance.cl:126-144).
//
// MI355X mapping
//   * one work-item per light sample, 256-thread workgroups (4 waves, one per SIMD);
//     consecutive lanes trace consecutive samples of the emission lattice, i.e. parallel,
//     ~quarter-voxel-apart rays: the 8 trilinear corners of neighbouring lanes share cache
//     lines, which is the property the reference's index sort preserves
//     (processor/progressivephotontracercl.cpp:467-473).
//   * CDNA4 has no sampler hardware.  The image3d_t fetch becomes ONE load of the sample's 2 x 2 x 2
//     footprint from the footprint-ordered copy of the volume (cpm_volume::quads: two neighbouring
//     elements = 8 / 16 / 32 bytes for u8 / u16 / f32) plus 7 two-fma lerps; the clamp-to-edge rule
//     is applied to the coordinate and built into the copy's last row / slice, so the footprint is
//     always in range (DESIGN.md "Arithmetic contract").
//   * the transfer function is read only through its alpha channel (color.w, scattering.w):
//     the alpha column is staged once per workgroup into LDS (width*4 bytes = 4 KiB for
//     Inviwo's 1024-texel LUT) and sampled with two ds_read + one lerp.
//   * MWC64X state lives in two VGPRs; photons and light samples move as 2 x float4
//     (coalesced 2 KiB per wave-instruction).
#include "cpm_trace_body.hip.h"

using namespace cpm;

using namespace cpm::tracer;

namespace {

// SINGLE: max_interactions == 1 known at compile time -- the scatter branch (phase-function sample, re-encoded direction, slab
// test) and the sentinel loop leave the instruction stream of the headline configuration; the NO_SINGLE_SCATTERING variant and
// I > 1 take the general kernel.
// (two steps of the walk in flight, tracer::woodcock_ahead: measured 32.1 -> 30.2 us at config 2, 124.6 -> 118.0 at 4 M photons /
// 512^3, 105.7 -> 101.4 at 8 steps per photon; 3 and 4 in flight cost registers and lose)
#ifndef CPM_TRACE_AHEAD
#define CPM_TRACE_AHEAD 2
#endif
CPM_DEV int default_chunk(int b, unsigned n_chunks) {
    if (b < (int)(n_chunks & ~127u)) {
        const int x = b & 7, j = b >> 3;
        b = ((((j >> 4) << 3) + x) << 4) + (j & 15);
    }
    return b;
}

// LINEAR: the volume's footprint copy is stale (cpm_volume_mix) and this launch re-traces selected photons: four x-pair fetches of
// the linear block per sample instead of one (cpm::trace_volume_source); same voxels, same lerps, same bits.
// MULTI: the samples of several lights (cpm_trace_lights): the workgroup's chunk says whose -- that light's buffers, count and photon
// offset stand in for the launch's.
template <int DT, int EMIT, bool SINGLE = false, bool LINEAR = false, bool MULTI = false>
__global__ __launch_bounds__(256, 8) void trace_kernel(const TraceArgs A0) {
    TraceArgs A = A0;
    extern __shared__ float lds[];
    // One decoded / re-encoded direction per workgroup: a directional light gives every sample the same (theta, phi),
    // and decodeDirection + encodeDirection (two sincos, acos, atan2, a division: ~200 instructions) is a pure function of
    // those two words.  Thread 0 provides it for ITS sample (from the emitter's hint, else by evaluating it); a wave whose
    // lanes all carry the same bit patterns takes the shared result (the same operations on the same inputs: the same
    // bits), any other wave evaluates per lane.
    __shared__ float s_dir[8];
    float* lut = lds;
    float* luts = lds;

    // Which 256 samples: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b does not take
    // chunk b but a chunk of a 4096-sample tile t = b (mod 8): an XCD then works on whole tiles -- 4 rows of the emission
    // lattice, whose rays share volume lines in that XCD's L2 -- instead of every 8th quarter row, and the tile it wrote is
    // the tile the bin's workgroup t (same XCD) reads back.  Measured: 38.4 -> 37.0 us at config 2, 177 -> 159 us at 4 M
    // photons / 512^3, 4-9 % on every transfer function tried (tools/trace_exp.py).  Interleaving tiles, not eighths of the
    // lattice, keeps the XCDs balanced (round 1: contiguous eighths cost 701 -> 971 us on a sparse TF).
    // With a cpm_trace_order the same assignment of tiles to XCDs comes from its table, in which every XCD's heaviest chunks
    // of the last launches stand first (cpm_trace_order_update).
    int chunk = blockIdx.x;
    if (A.chunk_order) chunk = (int)A.chunk_order[blockIdx.x];
    else chunk = default_chunk(chunk, gridDim.x);
    int local_chunk = chunk;  // ... within its light
    if (MULTI) {  // (uniform)
        int sp = 0;
        while (sp + 1 < A0.n_spans && chunk >= A0.span[sp + 1].chunk_base) ++sp;
        local_chunk = chunk - A0.span[sp].chunk_base;
        A.light_samples = A0.span[sp].light_samples;
        A.isect = A0.span[sp].isect;
        A.n_threads = A0.span[sp].n;
        A.p.n_light_samples = A0.span[sp].n;
        A.p.photon_offset = A0.span[sp].photon_offset;
    }
    const int gid = local_chunk * (int)blockDim.x + (int)threadIdx.x;
    int threadId = gid;
    int nThreads = A.n_threads;
    if (A.n_threads_dev) {  // cpm_trace_selected: the launch covers the budget, the count says how much of it is work
        nThreads = min(nThreads, *A.n_threads_dev);
        if (chunk * (int)blockDim.x >= nThreads) return;  // the whole workgroup, before anything is staged
    }
    bool live = gid < nThreads;
    if (live && A.recompute_indices) {  // -D PHOTON_RECOMPUTATION (photontracer.cl:97-106)
        threadId = (int)A.recompute_indices[gid] - A.p.photon_offset;
        live = threadId >= 0 && threadId < A.p.n_light_samples;
    }
    const int photonOffset = A.p.photon_offset;
    const uint32_t maxInteractions = SINGLE ? 1u : (uint32_t)A.p.max_interactions;
    const size_t totalPhotons = (size_t)A.p.total_photons;
    uint2* rng = reinterpret_cast<uint2*>(A.rng);
    if (live && A.old_photons) {  // what the light-volume update subtracts: the records this thread is about to replace
        for (uint32_t k = 0; k < maxInteractions; ++k) {
            const float4* q = reinterpret_cast<const float4*>(A.photons) + (size_t)A.rec_stride * ((size_t)photonOffset + k * totalPhotons + (size_t)threadId);
            float4* o = reinterpret_cast<float4*>(A.old_photons) + 2 * ((size_t)k * A.old_stride + (size_t)gid);  // (kept as float8 records)
            const float4 a = q[0], b = q[A.rec_b];
            o[0] = a; o[1] = b;
        }
    }
    // this lane's inputs, requested before the LUT is staged so that the loads overlap it
    float4 l0 = make_float4(0.f, 0.f, 0.f, 0.f), l1 = l0;
    float2 ip = make_float2(0.f, -1.f);
    uint2 rs = make_uint2(0u, 0u);
    if (live) {
        if (EMIT == EMIT_NONE) {
            const float4* lsp = reinterpret_cast<const float4*>(A.light_samples) + 2 * (size_t)threadId;
            l0 = lsp[0]; l1 = lsp[1];
            ip = reinterpret_cast<const float2*>(A.isect)[threadId];
        }
        rs = rng[photonOffset + threadId];
    }
    for (int i = threadIdx.x; i < A.tf_width; i += blockDim.x) lut[i] = A.tf_alpha[i];
    if (A.tfs_alpha != A.tf_alpha) {
        luts = lds + A.tf_width;
        for (int i = threadIdx.x; i < A.tf_width; i += blockDim.x) luts[i] = A.tfs_alpha[i];
    }
    if (EMIT == EMIT_DIRECTIONAL) {  // the hint was made for exactly this light (cpm_trace_emitted)
        if (threadIdx.x < 7) s_dir[threadIdx.x] = A.dir_hint[threadIdx.x];
    } else if (EMIT == EMIT_NONE && threadIdx.x == 0) {
        // the emitter's hint (cpm_directional_light_samples left decode / encode of ITS (theta, phi) in the context): taken
        // when it is for this thread's (theta, phi); otherwise thread 0 evaluates them itself
        const float4 h0 = reinterpret_cast<const float4*>(A.dir_hint)[0], h1 = reinterpret_cast<const float4*>(A.dir_hint)[1];
        if (h1.w == 1.0f && __float_as_uint(h0.x) == __float_as_uint(l1.z) && __float_as_uint(h0.y) == __float_as_uint(l1.w)) {
            s_dir[0] = h0.x; s_dir[1] = h0.y; s_dir[2] = h0.z; s_dir[3] = h0.w; s_dir[4] = h1.x; s_dir[5] = h1.y; s_dir[6] = h1.z;
        } else {
            const f3 d0 = decode_direction_(l1.z, l1.w);
            float t0, p0;
            encode_direction_(d0, t0, p0);
            s_dir[0] = l1.z; s_dir[1] = l1.w; s_dir[2] = d0.x; s_dir[3] = d0.y; s_dir[4] = d0.z; s_dir[5] = t0; s_dir[6] = p0;
        }
    }
    __syncthreads();
    if (!live) return;

    if (EMIT != EMIT_NONE) {
        // what cpm_uniform_samples_2d -> cpm_*_light_samples -> cpm_light_sample_box_intersection leave in their buffers for this
        // sample, from the same device functions (cpm_emit.hip.h): 40 bytes per photon not read
        const float4 ls = lattice_sample_(A.first_sample + threadId, A.lattice_x, A.lattice_y);
        f3 o, pw, d;
        float t0, p0;
        if (EMIT == EMIT_DIRECTIONAL) {
            directional_sample_(A.light, ls, o, pw);
            t0 = s_dir[0]; p0 = s_dir[1];
            d = { s_dir[2], s_dir[3], s_dir[4] };
        } else {
            point_sample_(A.light, ls, o, pw, t0, p0);
            d = decode_direction_(t0, p0);
        }
        Box b;
        for (int a = 0; a < 3; ++a) { b.mn[a] = A.bmin[a]; b.mx[a] = A.bmax[a]; }
        ip = box_entry_exit_(b, o, d);
        l0 = make_float4(o.x, o.y, o.z, pw.x);
        l1 = make_float4(pw.y, pw.z, t0, p0);
    }

    unsigned steps = 0;
    f3 direction;
    float th, ph;  // encodeDirection(direction)
    // (a point light's samples each have their own direction: s_dir is not even written in that mode)
    if (EMIT != EMIT_POINT && __all(__float_as_uint(l1.z) == __float_as_uint(s_dir[0]) && __float_as_uint(l1.w) == __float_as_uint(s_dir[1]))) {
        direction.x = s_dir[2]; direction.y = s_dir[3]; direction.z = s_dir[4];
        th = s_dir[5]; ph = s_dir[6];
    } else {
        direction = decode_direction_(l1.z, l1.w);
        encode_direction_(direction, th, ph);
    }
    // (SINGLE: two steps of the walk in flight; the general kernel's registers are full without them: I = 4 97.3 us with two, 93.6 with one)
    trace_photon<DT, SINGLE, (SINGLE ? CPM_TRACE_AHEAD : 1), LINEAR>(A, lut, luts, threadId, l0, l1, ip, rs, direction, th, ph, steps);
    if (A.chunk_cost) {  // what this chunk cost: the wave's longest walk (cpm_trace_order); lanes past the end have left
        unsigned m = steps;
        const unsigned long long alive = __ballot(true);
        const unsigned lane = threadIdx.x & 63u;
        for (unsigned off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)m, (int)off, 64);
            if ((alive >> (lane ^ off)) & 1ull) m = max(m, o);
        }
        if (lane == 0) {
            A.chunk_cost[4u * (uint32_t)chunk + (threadIdx.x >> 6)] = m;  // (a plain store per wave: per-wave atomics cost a launch 2 us)
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&A.chunk_cost[4u * gridDim.x], 1u);  // launches measured
        }
    }
    if (A.step_counter) {
        // statistics only: wave-level sum, one atomic per wave
        unsigned s = steps;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(A.step_counter, (unsigned long long)s);
    }
}

}  // namespace

namespace cpm {

int make_trace_args(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                    const cpm_trace_params* params, tracer::TraceArgs& A, size_t& lds) {
    CPM_REQUIRE(ctx, vol && tf && aabb && params, "cpm_trace: null argument");
    const cpm_trace_params& p = *params;
    CPM_REQUIRE(ctx, p.n_light_samples >= 0 && p.photon_offset >= 0 && p.total_photons >= 0, "cpm_trace: negative size");
    CPM_REQUIRE(ctx, p.max_interactions >= 1 && p.max_interactions <= 64, "cpm_trace: max_interactions in [1, 64]");
    CPM_REQUIRE(ctx, (long long)p.photon_offset + p.n_light_samples <= (long long)p.total_photons,
                "cpm_trace: photon_offset + n_light_samples exceeds total_photons");
    if (tf_scattering) CPM_REQUIRE(ctx, tf_scattering->width == tf->width, "cpm_trace: tf widths differ");
    const cpm_volume_desc& d = vol->desc;
    CPM_REQUIRE(ctx, (unsigned long long)d.dims[0] * d.dims[1] * d.dims[2] < (1ull << 32), "cpm_trace: volume too large");
    A = tracer::TraceArgs{};
    A.vol.voxels = vol->quads;
    A.vol.fx = (float)d.dims[0]; A.vol.fy = (float)d.dims[1]; A.vol.fz = (float)d.dims[2];
    A.vol.mx1 = (float)(d.dims[0] - 1); A.vol.my1 = (float)(d.dims[1] - 1); A.vol.mz1 = (float)(d.dims[2] - 1);
    A.vol.mx2 = (float)(d.dims[0] - 2);
    A.vol.my2 = (float)(d.dims[1] > 2 ? d.dims[1] - 2 : 0);
    A.vol.mz2 = (float)(d.dims[2] > 2 ? d.dims[2] - 2 : 0);
    A.vol.sy = (uint32_t)d.dims[0];
    A.vol.sz = (uint32_t)d.dims[0] * (uint32_t)d.dims[1];
    A.vol.mul24 = A.vol.sz < (1u << 24) && d.dims[0] < (1 << 24) && d.dims[1] < (1 << 24) && d.dims[2] < (1 << 24);
    A.vol.norm = d.dtype == CPM_U8 ? (1.0f / 255.0f) : (d.dtype == CPM_U16 ? (1.0f / 65535.0f) : 1.0f);
    A.vol.offset = d.format_offset;
    A.vol.one_minus_scaling = 1.0f - d.format_scaling;
    A.tf_alpha = tf->alpha;
    A.tfs_alpha = tf_scattering ? tf_scattering->alpha : tf->alpha;
    A.tf_width = tf->width;
    A.tf_wf = (float)tf->width; A.tf_m1 = (float)(tf->width - 1); A.tf_m2 = (float)(tf->width - 2);
    for (int a = 0; a < 3; ++a) { A.bmin[a] = aabb[a]; A.bmax[a] = aabb[4 + a]; }
    A.p = p;
    A.step_counter = ctx->dbg.step_counter;
    A.dir_hint = ctx->dir_hint;
    lds = (size_t)tf->width * sizeof(float) * (A.tfs_alpha != A.tf_alpha ? 2 : 1);
    return CPM_OK;
}

int trace_volume_source(cpm_ctx* ctx, const cpm_volume* vol, bool sparse_launch, hipStream_t s, tracer::TraceArgs& A, bool* linear) {
    *linear = false;
    if (!vol->quads_stale) return CPM_OK;  // (make_trace_args left A.vol.voxels = vol->quads)
    if (sparse_launch) {
        A.vol.voxels = vol->voxels;
        *linear = true;
        return CPM_OK;
    }
    // (the copy is derived data of the volume the caller handed over as const: re-deriving it does not change what the volume holds)
    return build_quads(ctx, const_cast<cpm_volume*>(vol), vol->voxels, false, s);
}

}

extern "C" {

// statistics hook used by bench.py: when non-null, trace launches add their Woodcock
// iteration counts to this device counter (not part of cpm.h's stable surface)
void cpm_debug_set_step_counter(cpm_ctx* ctx, unsigned long long* dev_counter) { if (ctx) ctx->dbg.step_counter = dev_counter; }

}  // extern "C"

namespace {
__global__ void emit_hint_kernel(Light L, float* __restrict__ hint) { directional_hint_(L, hint); }

struct SelectedArgs {  // cpm_trace_selected's extras
    const int32_t* n_dev = nullptr;
    float* old_photons = nullptr;
    uint32_t* reset_importances = nullptr;
    // cpm_trace_lights' extras: the lights of the launch (light_samples8 / isect2 / params' offset and count are then unused)
    const cpm_light_span* lights = nullptr;
    int n_lights = 0;
};

int trace_impl(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
               const cpm_trace_params* params, const float* light_samples8, const float* isect2, const cpm_emitter_desc* emitter,
               const uint32_t* recompute_indices, int n_recompute, uint32_t* rng_state, float* photons8,
               cpm_stream stream, const SelectedArgs& sel = SelectedArgs()) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, vol && tf && aabb && params, "cpm_trace: null argument");
    const cpm_trace_params& p = *params;
    CPM_REQUIRE(ctx, p.n_light_samples >= 0 && p.photon_offset >= 0 && p.total_photons >= 0, "cpm_trace: negative size");
    CPM_REQUIRE(ctx, p.max_interactions >= 1 && p.max_interactions <= 64, "cpm_trace: max_interactions in [1, 64]");
    CPM_REQUIRE(ctx, (long long)p.photon_offset + p.n_light_samples <= (long long)p.total_photons,
                "cpm_trace: photon_offset + n_light_samples exceeds total_photons");
    CPM_REQUIRE(ctx, n_recompute >= 0, "cpm_trace: n_recompute < 0");
    if (tf_scattering) CPM_REQUIRE(ctx, tf_scattering->width == tf->width, "cpm_trace: tf widths differ");
    int n_threads = recompute_indices ? n_recompute : p.n_light_samples;
    if (sel.lights) {  // the launch's chunks: every light's samples rounded up to whole chunks (sizes checked first: a negative count
                       // must not cancel another light's chunks and pass for an empty launch)
        long long chunks = 0;
        for (int l = 0; l < sel.n_lights; ++l) {
            CPM_REQUIRE(ctx, sel.lights[l].n_light_samples >= 0 && sel.lights[l].photon_offset >= 0, "cpm_trace_lights: negative size");
            chunks += div_up(sel.lights[l].n_light_samples, 256);
        }
        CPM_REQUIRE(ctx, chunks * 256 < (1ll << 31), "cpm_trace_lights: too many samples");
        n_threads = (int)(chunks * 256);
    }
    if (n_threads == 0) return CPM_OK;
    CPM_REQUIRE(ctx, rng_state && photons8, "cpm_trace: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, photons8, "cpm_trace");
    CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(rng_state) & 7u) == 0, "cpm_trace: rng_state must be 8-byte aligned");
    if (sel.lights) {
        for (int l = 0; l < sel.n_lights; ++l) {
            const cpm_light_span& L = sel.lights[l];
            CPM_REQUIRE(ctx, L.n_light_samples >= 0 && L.photon_offset >= 0 && (long long)L.photon_offset + L.n_light_samples <= (long long)p.total_photons,
                        "cpm_trace_lights: a light's photon_offset + n_light_samples exceeds total_photons");
            CPM_REQUIRE(ctx, L.n_light_samples == 0 || (L.light_samples8 && L.isect2), "cpm_trace_lights: null buffer");
            CPM_REQUIRE_ALIGNED16(ctx, L.light_samples8, "cpm_trace_lights");
            CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(L.isect2) & 7u) == 0, "cpm_trace_lights: isect2 must be 8-byte aligned");
        }
    } else if (!emitter) {
        CPM_REQUIRE(ctx, light_samples8 && isect2, "cpm_trace: null buffer");
        CPM_REQUIRE_ALIGNED16(ctx, light_samples8, "cpm_trace");
        CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(isect2) & 7u) == 0, "cpm_trace: isect2 must be 8-byte aligned");
    } else {
        CPM_REQUIRE(ctx, emitter->kind == CPM_EMIT_DIRECTIONAL || emitter->kind == CPM_EMIT_POINT, "cpm_trace_emitted: emitter kind");
        CPM_REQUIRE(ctx, emitter->nx > 0 && emitter->ny > 0 && (long long)emitter->nx * emitter->ny < (1ll << 24),
                    "cpm_trace_emitted: 0 < nx*ny < 2^24 (cpm_uniform_samples_2d's range)");
        CPM_REQUIRE(ctx, emitter->first_sample >= 0 && (long long)emitter->first_sample + p.n_light_samples <= (long long)emitter->nx * emitter->ny,
                    "cpm_trace_emitted: first_sample + n_light_samples exceeds the lattice");
    }
    const cpm_volume_desc& d = vol->desc;
    TraceArgs A;
    size_t lds = 0;
    int rc_args = cpm::make_trace_args(ctx, vol, tf, tf_scattering, aabb, params, A, lds);
    if (rc_args) return rc_args;
    A.light_samples = light_samples8;
    A.isect = isect2;
    A.recompute_indices = recompute_indices;
    A.n_threads = n_threads;
    A.n_threads_dev = sel.n_dev;
    A.old_photons = sel.old_photons;
    A.old_stride = (uint32_t)n_threads;
    A.reset_importances = sel.reset_importances;
    if (sel.old_photons) CPM_REQUIRE_ALIGNED16(ctx, sel.old_photons, "cpm_trace_selected");
    A.rng = rng_state;
    A.photons = photons8;
    // the call's flag, else how the buffer was described (cpm_records_describe: its own N * I is the distance between the planes), else the context's
    size_t described_n = 0;
    const int described = described_layout(ctx, photons8, &described_n);
    const bool planar = (p.flags & CPM_TRACE_PHOTONS_PLANAR) != 0 || (described >= 0 ? described == CPM_PHOTONS_PLANAR : ctx->photon_layout == CPM_PHOTONS_PLANAR);
    A.rec_stride = planar ? 1u : 2u;
    A.rec_b = planar ? (described == CPM_PHOTONS_PLANAR ? (uint32_t)described_n : (uint32_t)((long long)p.total_photons * p.max_interactions)) : 1u;
    if (sel.lights) {
        A.n_spans = sel.n_lights;
        int base = 0;
        for (int l = 0; l < sel.n_lights; ++l) {
            A.span[l] = { sel.lights[l].light_samples8, sel.lights[l].isect2, sel.lights[l].n_light_samples, sel.lights[l].photon_offset, base };
            base += div_up(sel.lights[l].n_light_samples, 256);
        }
    }

    hipStream_t s = (hipStream_t)stream;
    // a stale footprint copy (the volume is cpm_volume_mix's output): a launch over selected photons whose number is on the device,
    // or at most a quarter of the samples, reads the linear block; any other launch re-derives the copy first
    bool linear = false;
    const bool sparse = recompute_indices && !emitter && (sel.n_dev || (long long)n_recompute * 4 <= (long long)p.n_light_samples);
    int rc_src = cpm::trace_volume_source(ctx, vol, sparse, s, A, &linear);
    if (rc_src) return rc_src;
    dim3 grid(div_up(n_threads, 256)), block(256);
    if (ctx->trace_order && !recompute_indices && !sel.n_dev) {  // a plain launch over all the samples: in the order's order
        const cpm_trace_order* o = ctx->trace_order;
        CPM_REQUIRE(ctx, o->n_light_samples == n_threads, "cpm_trace: the trace order set on this context was created for another number of samples");
        A.chunk_order = o->order;
        A.chunk_cost = ctx->trace_order_measure ? o->cost : nullptr;
    }
    int emit = EMIT_NONE;
    if (emitter) {
        emit = emitter->kind == CPM_EMIT_DIRECTIONAL ? EMIT_DIRECTIONAL : EMIT_POINT;
        for (int a = 0; a < 3; ++a) {
            A.light.radiance[a] = emitter->radiance[a]; A.light.a[a] = emitter->direction_or_position[a];
            A.light.origin[a] = emitter->plane_origin[a]; A.light.u[a] = emitter->tangent_u[a]; A.light.v[a] = emitter->tangent_v[a];
        }
        A.light.area = emitter->plane_area;
        A.lattice_x = (float)emitter->nx; A.lattice_y = (float)emitter->ny;
        A.first_sample = emitter->first_sample;
        if (emit == EMIT_DIRECTIONAL) {
            A.dir_hint = ctx->dir_hint + 8;
            if (!ctx->emit_hint_valid || memcmp(ctx->emit_hint_for, A.light.a, sizeof(ctx->emit_hint_for)) != 0) {
                ctx->emit_hint_valid = false;
                CPM_LAUNCH(ctx, emit_hint_kernel, dim3(1), dim3(1), 0, s, A.light, ctx->dir_hint + 8);
                CPM_LAUNCH_CHECK(ctx, "emit_hint_kernel");
                memcpy(ctx->emit_hint_for, A.light.a, sizeof(ctx->emit_hint_for));
                ctx->emit_hint_valid = true;
            }
        }
    }
    const bool single = p.max_interactions == 1 && !(p.flags & CPM_TRACE_NO_SINGLE_SCATTERING);
#define CPM_TRACE_LAUNCH_E(DT, E)                                                          \
    do {                                                                                   \
        if (single) CPM_LAUNCH(ctx, (trace_kernel<DT, E, true>), grid, block, lds, s, A);  \
        else CPM_LAUNCH(ctx, (trace_kernel<DT, E, false>), grid, block, lds, s, A);        \
    } while (0)
#define CPM_TRACE_LAUNCH(DT)                                                  \
    do {                                                                      \
        if (linear && single) CPM_LAUNCH(ctx, (trace_kernel<DT, EMIT_NONE, true, true>), grid, block, lds, s, A);   \
        else if (linear) CPM_LAUNCH(ctx, (trace_kernel<DT, EMIT_NONE, false, true>), grid, block, lds, s, A);       \
        else if (emit == EMIT_NONE) CPM_TRACE_LAUNCH_E(DT, EMIT_NONE);        \
        else if (emit == EMIT_DIRECTIONAL) CPM_TRACE_LAUNCH_E(DT, EMIT_DIRECTIONAL); \
        else CPM_TRACE_LAUNCH_E(DT, EMIT_POINT);                              \
    } while (0)
#define CPM_TRACE_LAUNCH_MULTI(DT)                                                                                         \
    do {                                                                                                                   \
        if (single) CPM_LAUNCH(ctx, (trace_kernel<DT, EMIT_NONE, true, false, true>), grid, block, lds, s, A);            \
        else CPM_LAUNCH(ctx, (trace_kernel<DT, EMIT_NONE, false, false, true>), grid, block, lds, s, A);                  \
    } while (0)
    if (sel.lights) {
        switch (d.dtype) {
            case CPM_U8: CPM_TRACE_LAUNCH_MULTI(CPM_U8); break;
            case CPM_U16: CPM_TRACE_LAUNCH_MULTI(CPM_U16); break;
            default: CPM_TRACE_LAUNCH_MULTI(CPM_F32); break;
        }
    } else
    switch (d.dtype) {
        case CPM_U8: CPM_TRACE_LAUNCH(CPM_U8); break;
        case CPM_U16: CPM_TRACE_LAUNCH(CPM_U16); break;
        default: CPM_TRACE_LAUNCH(CPM_F32); break;
    }
#undef CPM_TRACE_LAUNCH_MULTI
#undef CPM_TRACE_LAUNCH
#undef CPM_TRACE_LAUNCH_E
    CPM_LAUNCH_CHECK(ctx, "trace_kernel");
    return CPM_OK;
}

// cpm_trace_order_update: one workgroup per XCD.  The chunks XCD x works on are, in lattice order, default_chunk(8 j + x);
// those whose cost is among the XCD's heaviest eighth come first, the others follow, both in lattice order (a stable
// partition by ballots); chunks past the last full block of 128 keep their place.  Then the costs are cleared.
constexpr int kOrderThreads = 1024, kOrderBins = 2048;  // (two bins per thread)
// a chunk's cost: the sum of its four waves' longest walks in the last measured launch
CPM_DEV uint32_t chunk_cost_of(const uint32_t* __restrict__ cost, uint32_t c) {
    const uint4 w = reinterpret_cast<const uint4*>(cost)[c];
    return w.x + w.y + w.z + w.w;
}
__global__ __launch_bounds__(kOrderThreads) void trace_order_kernel(uint32_t* __restrict__ order, uint32_t* __restrict__ cost, uint32_t n_chunks) {
    __shared__ uint32_t s_hist[kOrderBins];
    __shared__ uint32_t s_wave[kOrderThreads / 64];
    __shared__ uint32_t s_thr, s_shift, s_heavy;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, x = blockIdx.x;
    const uint32_t full = n_chunks & ~127u, mine = full / 8u;  // this XCD's chunks
    if (cost[4u * n_chunks] == 0u) return;  // nothing measured since the last update (uniform)
    if (x == 0) for (uint32_t b = full + t; b < n_chunks; b += kOrderThreads) { order[b] = b; }
    if (mine == 0u) return;
    // the costs' range -> a shift that brings them under kOrderBins
    uint32_t mx = 0;
    for (uint32_t j = t; j < mine; j += kOrderThreads) mx = max(mx, chunk_cost_of(cost, (uint32_t)default_chunk((int)(8u * j + x), n_chunks)));
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    if (lane == 0) s_wave[wave] = mx;
    for (uint32_t i = t; i < kOrderBins; i += kOrderThreads) s_hist[i] = 0u;
    __syncthreads();
    if (t == 0) {
        uint32_t m = 0;
        for (int w = 0; w < kOrderThreads / 64; ++w) m = max(m, s_wave[w]);
        uint32_t sh = 0;
        while ((m >> sh) >= (uint32_t)kOrderBins) ++sh;
        s_shift = sh;
    }
    __syncthreads();
    const uint32_t sh = s_shift;
    for (uint32_t j = t; j < mine; j += kOrderThreads) atomicAdd(&s_hist[chunk_cost_of(cost, (uint32_t)default_chunk((int)(8u * j + x), n_chunks)) >> sh], 1u);
    __syncthreads();
    // bins above `thr` hold at most an eighth of the chunks: those are the heavy ones.  From the top: r = kOrderBins - 1 - bin,
    // cum(r) = chunks in the bins r' <= r; thread t owns r = 2 t and 2 t + 1; R = how many r have cum(r) <= want (cum is
    // monotone, so these are the first R), thr = kOrderBins - 1 - R, the heavy ones number cum(R - 1).
    {
        const uint32_t want = mine / 8u;
        if (t == 0) { s_thr = 0u; s_heavy = 0u; }
        const uint32_t h0 = s_hist[kOrderBins - 1 - 2 * t], h1 = s_hist[kOrderBins - 2 - 2 * t];
        uint32_t inc = h0 + h1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
        const uint32_t c1 = before + inc, c0 = c1 - h1;  // cum(2 t), cum(2 t + 1)
        uint32_t r_ok = (c0 <= want ? 1u : 0u) + (c1 <= want ? 1u : 0u);
        if (r_ok) { atomicAdd(&s_thr, r_ok); atomicMax(&s_heavy, r_ok == 2u ? c1 : c0); }
        __syncthreads();
        if (t == 0) s_thr = (uint32_t)(kOrderBins - 1) - s_thr;
        __syncthreads();
    }
    const uint32_t thr = s_thr, n_heavy = s_heavy;
    // stable partition of the XCD's list: heavy chunks to positions [0, n_heavy), the others behind them
    uint32_t base_h = 0, base_l = n_heavy;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t j0 = 0; j0 < mine; j0 += kOrderThreads) {  // (uniform)
        const uint32_t j = j0 + t;
        const bool in = j < mine;
        const uint32_t c = in ? (uint32_t)default_chunk((int)(8u * j + x), n_chunks) : 0u;
        const bool heavy = in && (chunk_cost_of(cost, c) >> sh) > thr;
        const unsigned long long mh = __ballot(heavy), ml = __ballot(in && !heavy);
        __syncthreads();
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mh) | ((uint32_t)__popcll(ml) << 16);
        __syncthreads();
        uint32_t bh = 0, bl = 0, th = 0, tl = 0;
        for (uint32_t w = 0; w < kOrderThreads / 64; ++w) {
            const uint32_t v = s_wave[w];
            if (w < wave) { bh += v & 0xffffu; bl += v >> 16; }
            th += v & 0xffffu; tl += v >> 16;
        }
        if (in) {
            const uint32_t pos = heavy ? base_h + bh + (uint32_t)__popcll(mh & lt) : base_l + bl + (uint32_t)__popcll(ml & lt);
            order[8u * pos + x] = c;  // the XCD's pos-th workgroup
        }
        base_h += th; base_l += tl;
    }
}

__global__ __launch_bounds__(256) void trace_order_clear_kernel(uint32_t* __restrict__ cost, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) cost[i] = 0u;
}
}  // namespace

extern "C" {

int cpm_trace_order_create(cpm_ctx* ctx, int n_light_samples, cpm_trace_order** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, out, "cpm_trace_order_create: null out");
    *out = nullptr;
    CPM_REQUIRE(ctx, n_light_samples > 0, "cpm_trace_order_create: n_light_samples < 1");
    cpm_trace_order* o = new (std::nothrow) cpm_trace_order();
    if (!o) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_trace_order_create", "host allocation");
    o->n_light_samples = n_light_samples;
    o->n_chunks = (uint32_t)div_up(n_light_samples, 256);
    std::vector<uint32_t> init(o->n_chunks);
    const uint32_t full = o->n_chunks & ~127u;
    for (uint32_t b = 0; b < o->n_chunks; ++b) {
        uint32_t c = b;
        if (b < full) { const uint32_t x = b & 7u, j = b >> 3; c = ((((j >> 4) << 3) + x) << 4) + (j & 15u); }
        init[b] = c;
    }
    bool ok = hipMalloc(&o->order, (size_t)o->n_chunks * 4) == hipSuccess && hipMalloc(&o->cost, (4 * (size_t)o->n_chunks + 1) * 4) == hipSuccess &&
              hipMemcpy(o->order, init.data(), (size_t)o->n_chunks * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(o->cost, 0, (4 * (size_t)o->n_chunks + 1) * 4) == hipSuccess;
    if (!ok) {
        if (o->order) (void)hipFree(o->order);
        if (o->cost) (void)hipFree(o->cost);
        delete o;
        return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_trace_order_create", "device allocation");
    }
    *out = o;
    return CPM_OK;
}

void cpm_trace_order_destroy(cpm_ctx* ctx, cpm_trace_order* order) {
    if (!order) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        if (ctx->trace_order == order) { ctx->trace_order = nullptr; ctx->trace_order_measure = false; }
    }
    if (order->order) (void)hipFree(order->order);
    if (order->cost) (void)hipFree(order->cost);
    delete order;
}

int cpm_trace_set_order(cpm_ctx* ctx, cpm_trace_order* order, int measure) {
    CPM_ENTER(ctx);
    ctx->trace_order = order;
    ctx->trace_order_measure = order != nullptr && measure != 0;
    return CPM_OK;
}

// test hook (include/cpm/cpm_profile.h): the order table and the costs gathered so far, copied to the host (synchronises)
int cpm_debug_trace_order_read(cpm_ctx* ctx, const cpm_trace_order* order, uint32_t* order_out, uint32_t* cost_out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, order, "cpm_debug_trace_order_read: null order");
    CPM_HIP_CHECK(ctx, hipDeviceSynchronize());
    if (order_out) CPM_HIP_CHECK(ctx, hipMemcpy(order_out, order->order, (size_t)order->n_chunks * 4, hipMemcpyDeviceToHost));
    if (cost_out) {  // per chunk the sum of its waves' slots, then the launches measured
        std::vector<uint32_t> raw(4 * (size_t)order->n_chunks + 1);
        CPM_HIP_CHECK(ctx, hipMemcpy(raw.data(), order->cost, raw.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < order->n_chunks; ++c) cost_out[c] = raw[4 * c] + raw[4 * c + 1] + raw[4 * c + 2] + raw[4 * c + 3];
        cost_out[order->n_chunks] = raw[4 * (size_t)order->n_chunks];
    }
    return CPM_OK;
}

// measurement hook (include/cpm/cpm_profile.h): another table for the order object -- any permutation of its chunks (checked)
int cpm_debug_trace_order_write(cpm_ctx* ctx, cpm_trace_order* order, const uint32_t* table) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, order && table, "cpm_debug_trace_order_write: null argument");
    std::vector<uint8_t> seen(order->n_chunks, 0);
    for (uint32_t b = 0; b < order->n_chunks; ++b) {
        CPM_REQUIRE(ctx, table[b] < order->n_chunks && !seen[table[b]], "cpm_debug_trace_order_write: not a permutation of the chunks");
        seen[table[b]] = 1;
    }
    CPM_HIP_CHECK(ctx, hipDeviceSynchronize());
    CPM_HIP_CHECK(ctx, hipMemcpy(order->order, table, (size_t)order->n_chunks * 4, hipMemcpyHostToDevice));
    return CPM_OK;
}

int cpm_trace_order_update(cpm_ctx* ctx, cpm_trace_order* order, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, order && order->order && order->cost, "cpm_trace_order_update: null order");
    hipStream_t s = (hipStream_t)stream;
    CPM_LAUNCH(ctx, trace_order_kernel, dim3(8), dim3(kOrderThreads), 0, s, order->order, order->cost, order->n_chunks);
    CPM_LAUNCH_CHECK(ctx, "trace_order_kernel");
    CPM_LAUNCH(ctx, trace_order_clear_kernel, dim3(div_up(4ll * order->n_chunks + 1, 256)), dim3(256), 0, s, order->cost, 4u * order->n_chunks + 1u);
    CPM_LAUNCH_CHECK(ctx, "trace_order_clear_kernel");
    return CPM_OK;
}

}  // extern "C"

extern "C" {

int cpm_trace(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
              const cpm_trace_params* params, const float* light_samples8, const float* isect2,
              const uint32_t* recompute_indices, int n_recompute, uint32_t* rng_state, float* photons8,
              cpm_stream stream) {
    return trace_impl(ctx, vol, tf, tf_scattering, aabb, params, light_samples8, isect2, nullptr, recompute_indices, n_recompute,
                      rng_state, photons8, stream);
}

int cpm_trace_lights_order_samples(const cpm_light_span* lights, int n_lights) {
    long long chunks = 0;
    for (int l = 0; lights && l < n_lights; ++l) chunks += lights[l].n_light_samples > 0 ? div_up(lights[l].n_light_samples, 256) : 0;
    return chunks * 256 < (1ll << 31) ? (int)(chunks * 256) : 0;
}

int cpm_trace_lights(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                     const cpm_trace_params* params, const cpm_light_span* lights, int n_lights, uint32_t* rng_state, float* photons8,
                     cpm_stream stream) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;   // (as every entry point: no context, nothing to report through)
    if (!(lights && n_lights >= 1 && n_lights <= CPM_MAX_TRACE_LIGHTS))
        return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_trace_lights", "1 .. CPM_MAX_TRACE_LIGHTS lights");
    if (!params) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_trace_lights", "null params");
    cpm_trace_params p = *params;   // (the per-light fields come from the spans)
    p.photon_offset = 0;
    p.n_light_samples = 0;
    SelectedArgs sel;
    sel.lights = lights;
    sel.n_lights = n_lights;
    return trace_impl(ctx, vol, tf, tf_scattering, aabb, &p, nullptr, nullptr, nullptr, nullptr, 0, rng_state, photons8, stream, sel);
}

int cpm_trace_selected(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                       const cpm_trace_params* params, const float* light_samples8, const float* isect2, const uint32_t* indices,
                       const int32_t* n_indices_dev, int max_indices, float* old_photons8, uint32_t* reset_importances,
                       uint32_t* rng_state, float* photons8, cpm_stream stream) {
    if (ctx && !(indices && n_indices_dev)) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_trace_selected", "null indices / count");
    SelectedArgs sel;
    sel.n_dev = n_indices_dev;
    sel.old_photons = old_photons8;
    sel.reset_importances = reset_importances;
    return trace_impl(ctx, vol, tf, tf_scattering, aabb, params, light_samples8, isect2, nullptr, indices, max_indices, rng_state, photons8,
                      stream, sel);
}

int cpm_trace_emitted(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                      const cpm_trace_params* params, const cpm_emitter_desc* emitter, const uint32_t* recompute_indices,
                      int n_recompute, uint32_t* rng_state, float* photons8, cpm_stream stream) {
    if (ctx && !emitter) return set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_trace_emitted", "null emitter");
    return trace_impl(ctx, vol, tf, tf_scattering, aabb, params, nullptr, nullptr, emitter, recompute_indices, n_recompute,
                      rng_state, photons8, stream);
}

}  // extern "C"
