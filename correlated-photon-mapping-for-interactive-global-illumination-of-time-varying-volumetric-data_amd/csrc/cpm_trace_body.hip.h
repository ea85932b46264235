// cpm_trace_body.hip.h -- the photon tracer's device code (R3, R4), shared by the translation units that trace:
// cpm_trace.hip (trace_kernel: cpm_trace, cpm_trace_emitted, cpm_trace_selected) and cpm_correlated.hip
// (importance_retrace_kernel: detector + threshold + tracer in one launch).  Replaces photonTracerKernel
// (ref progressivephotonmapping/cl/photontracer.cl:69-216) and woodcockTracking (ref cl/transmittance.cl:126-144).
#pragma once
#include <type_traits>
#include "cpm_ctx.h"
#include "cpm_emit.hip.h"

namespace cpm {
namespace tracer {

struct VolDev {
    const void* voxels;      // cpm_volume::quads -- or cpm_volume::voxels for the LINEAR instantiations (the copy is stale)
    float fx, fy, fz;        // (float)dim
    float mx1, my1, mz1;     // dim - 1
    float mx2, my2, mz2;     // max(dim - 2, 0)
    uint32_t sy, sz;         // row / slice stride in elements
    int mul24;               // strides and indices fit 24 bits: v_mul_u32_u24 (full rate) instead of v_mul_lo_u32 (quarter)
    float norm, offset, one_minus_scaling;
};

struct TraceArgs {
    VolDev vol;
    const float* tf_alpha;
    const float* tfs_alpha;   // == tf_alpha when the reference's "tf passed twice" quirk applies
    int tf_width;
    float tf_wf, tf_m1, tf_m2;
    float bmin[3], bmax[3];
    cpm_trace_params p;
    const float* light_samples;
    const float* isect;
    const uint32_t* recompute_indices;
    int n_threads;
    // cpm_trace_selected: the number of indices lives on the device (nullable: n_threads is it); the records about to be
    // overwritten are kept in old_photons[k * old_stride + j]; the traced photons' importance keys are reset
    const int32_t* n_threads_dev;
    float* old_photons;
    uint32_t old_stride;
    uint32_t* reset_importances;
    uint32_t* rng;
    float* photons;
    // where record j's two 16-byte halves lie, in float4 units: half A = (position, power.r) at j * rec_stride, half B =
    // (power.g, power.b, theta, phi) rec_b behind it.  (2, 1): the reference's float8 record (cl/photon.cl:49-63); (1, N * I): the
    // two-plane layout of CPM_TRACE_PHOTONS_PLANAR -- what the bin reads lies in one plane.
    uint32_t rec_stride, rec_b;
    unsigned long long* step_counter;  // nullable (statistics build of the launch)
    const uint32_t* chunk_order;       // cpm_trace_order (nullable): workgroup b takes chunk chunk_order[b] ...
    uint32_t* chunk_cost;              // ... and every wave adds its longest walk to chunk_cost[chunk]
    const float* dir_hint;             // cpm_ctx::dir_hint (emitted mode: the hint of ITS light, cpm_ctx::dir_hint + 8)
    // emitted mode (cpm_trace_emitted): light sample and entry / exit of lattice sample first_sample + thread, in registers
    Light light;
    float lattice_x, lattice_y;
    int first_sample;
    // cpm_trace_lights: several lights' samples in one launch -- chunk c belongs to the span with chunk_base <= c < the next one's
    struct Span { const float* light_samples; const float* isect; int n, photon_offset, chunk_base; };
    int n_spans;
    Span span[CPM_MAX_TRACE_LIGHTS];
};

enum { EMIT_NONE = 0, EMIT_DIRECTIONAL = 1, EMIT_POINT = 2 };

// One fetch = the whole 2 x 2 x 2 footprint of a trilinear sample: two neighbouring elements of cpm_volume::quads, each
// { v(x, y, z), v(x, y + 1, z), v(x, y, z + 1), v(x, y + 1, z + 1) } (clamped at the last row / slice).  Four x-pair fetches of
// the linear layout before: the texture-address path is the loop's second bottleneck; two fetches per sample took 10 % off the
// config-2 trace and 24 % off a 33-step-per-photon one, one fetch a further 5 % and 13 %.
template <int DT> struct FootprintLoad;
template <> struct FootprintLoad<CPM_U8> {
    static CPM_DEV void load(const void* base, uint32_t idx, float (&v)[8]) {
        uint32_t w[2];
        __builtin_memcpy(w, static_cast<const uint8_t*>(base) + 4 * (size_t)idx, 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // v_cvt_f32_ubyte0..3
            v[4 * i + 0] = (float)(w[i] & 0xffu);
            v[4 * i + 1] = (float)((w[i] >> 8) & 0xffu);
            v[4 * i + 2] = (float)((w[i] >> 16) & 0xffu);
            v[4 * i + 3] = (float)(w[i] >> 24);
        }
    }
};
template <> struct FootprintLoad<CPM_U16> {
    static CPM_DEV void load(const void* base, uint32_t idx, float (&v)[8]) {
        uint32_t w[4];
        __builtin_memcpy(w, static_cast<const uint16_t*>(base) + 4 * (size_t)idx, 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i + 0] = (float)(w[i] & 0xffffu);
            v[2 * i + 1] = (float)(w[i] >> 16);
        }
    }
};
template <> struct FootprintLoad<CPM_F32> {
    static CPM_DEV void load(const void* base, uint32_t idx, float (&v)[8]) {
        __builtin_memcpy(v, static_cast<const float*>(base) + 4 * (size_t)idx, 32);
    }
};

CPM_DEV void coord(float s, float dimf, float m1, float m2, float& fl, float& a) {
    // u' = clamp(s*w - 0.5, 0, w-1), i0 = min(floor(u'), w-2) (0 <= floor(u'), 0 <= m2), a = u' - i0.
    // v_med3_f32 is the clamp in one instruction (the operands are never NaN here); min/max pairs cost two
    // plus a canonicalising v_max of the uniform bound each -- 16 instructions per Woodcock step.
    const float u = __builtin_amdgcn_fmed3f(fma_(s, dimf, -0.5f), 0.0f, m1);
    fl = __builtin_amdgcn_fmed3f(__builtin_floorf(u), 0.0f, m2);
    a = u - fl;
}

// The same eight voxels from the volume's linear block (x fastest): four fetches of an x-pair, rows (y, z), (y', z), (y, z'), (y', z')
// with y' = min(y + 1, dim.y - 1), z' likewise -- exactly what a footprint element holds -- into the same order.
template <int DT> struct LinearLoad {
    typedef typename std::conditional<DT == CPM_U8, uint8_t, typename std::conditional<DT == CPM_U16, uint16_t, float>::type>::type T;
    static CPM_DEV void load(const VolDev& V, uint32_t b00, int iy, int iz, float (&v)[8]) {
        const T* base = static_cast<const T*>(V.voxels) + b00;
        const uint32_t up = (float)iy < V.my1 ? V.sy : 0u, back = (float)iz < V.mz1 ? V.sz : 0u;
        const uint32_t off[4] = { 0u, up, back, up + back };   // [z][y] = 00 01(y') 10(z') 11
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            T pr[2];
            __builtin_memcpy(pr, base + off[k], sizeof(pr));
            v[k] = (float)pr[0]; v[4 + k] = (float)pr[1];
        }
    }
};

template <int DT, bool LINEAR = false>
CPM_DEV float sample_volume(const VolDev& V, float px, float py, float pz) {
    float flx, fly, flz, ax, ay, az;
    coord(px, V.fx, V.mx1, V.mx2, flx, ax);
    coord(py, V.fy, V.my1, V.my2, fly, ay);
    coord(pz, V.fz, V.mz1, V.mz2, flz, az);
    int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    uint32_t b00 = V.mul24 ? (uint32_t)ix + __umul24(V.sy, (uint32_t)iy) + __umul24(V.sz, (uint32_t)iz)
                           : (uint32_t)ix + V.sy * (uint32_t)iy + V.sz * (uint32_t)iz;
    float v[8];  // [x][z][y]: 000 010 001 011 | 100 110 101 111
    if (LINEAR) LinearLoad<DT>::load(V, b00, iy, iz, v);
    else FootprintLoad<DT>::load(V.voxels, b00, v);
    float c00 = lerp_(v[0], v[4], ax);
    float c10 = lerp_(v[1], v[5], ax);
    float c01 = lerp_(v[2], v[6], ax);
    float c11 = lerp_(v[3], v[7], ax);
    float c0 = lerp_(c00, c10, ay);
    float c1 = lerp_(c01, c11, ay);
    float c = lerp_(c0, c1, az);
    float s = c * V.norm;
    return (s + V.offset) * V.one_minus_scaling;
}

// read_imagef(tf, smpNormClampEdgeLinear, (float2)(v, 0.5f)).w from the LDS alpha column
CPM_DEV float sample_alpha(const float* lut, float wf, float m1, float m2, float v) {
    float fl, a;
    coord(v, wf, m1, m2, fl, a);
    int i = (int)fl;
    return lerp_(lut[i], lut[i + 1], a);
}

// build-defined phase-function sampling (Inviwo sampleShadingFunction is not in the
// reference tree): Henyey-Greenstein (g = material.x) or isotropic about `w`.
CPM_DEV float phase_cos(int type, float g, float u1) {
    if (type == CPM_PHASE_ISOTROPIC || __builtin_fabsf(g) < 1e-3f) return fma_(-2.0f, u1, 1.0f);
    float g2 = g * g;
    float sq = (1.0f - g2) / fma_(2.0f * g, u1, 1.0f - g);
    return (1.0f + g2 - sq * sq) / (2.0f * g);
}
CPM_DEV float phase_pdf(int type, float g, float cosT) {
    if (type == CPM_PHASE_ISOTROPIC || __builtin_fabsf(g) < 1e-3f) return kInv4Pi;
    float g2 = g * g;
    float den = fma_(-2.0f * g, cosT, 1.0f + g2);
    return kInv4Pi * (1.0f - g2) / (den * __builtin_sqrtf(den));
}
CPM_DEV f3 phase_sample(int type, float g, f3 w, float u1, float u2, float* pdf) {
    float cosT = phase_cos(type, g, u1);
    cosT = min_(max_(cosT, -1.0f), 1.0f);
    float sinT = __builtin_sqrtf(max_(0.0f, fma_(-cosT, cosT, 1.0f)));
    float sp, cp;
    sincos_(kTwoPi * u2, sp, cp);
    f3 a;
    if (__builtin_fabsf(w.z) < 0.999f) { a.x = 0; a.y = 0; a.z = 1; } else { a.x = 1; a.y = 0; a.z = 0; }
    f3 u = cross3_(a, w);
    float il = 1.0f / __builtin_sqrtf(dot3_(u, u));
    u.x *= il; u.y *= il; u.z *= il;
    f3 v = cross3_(w, u);
    float ku = sinT * cp, kv = sinT * sp;
    f3 d;
    d.x = fma_(cosT, w.x, fma_(kv, v.x, ku * u.x));
    d.y = fma_(cosT, w.y, fma_(kv, v.y, ku * u.y));
    d.z = fma_(cosT, w.z, fma_(kv, v.z, ku * u.z));
    if (pdf) *pdf = phase_pdf(type, g, cosT);
    return d;
}

CPM_DEV void write_photon(const TraceArgs& A, size_t id, f3 p, f3 pw, float th, float ph) {
    // (plain stores: with the tile-wise chunk order below a streaming hint no longer helps this launch, and the bin's count
    // launch reads the records 0.8 us sooner without it -- 5 us sooner at 4 M photons)
    float4* q = reinterpret_cast<float4*>(A.photons) + (size_t)A.rec_stride * id;
    q[0] = make_float4(p.x, p.y, p.z, pw.x);
    q[A.rec_b] = make_float4(pw.y, pw.z, th, ph);
}

template <int DT, bool LINEAR = false>
CPM_DEV float woodcock(const VolDev& V, const float* lut, float wf, float m1, float m2, f3 o, f3 d, float tStart,
                       float tEnd, uint32_t& rx, uint32_t& rc, unsigned& steps, float& last_sample, float& last_opacity) {
    constexpr float invTauMaxSampleBaseInterval = 1.f / (1.f * 150.f);  // tauMax = 1 (photontracer.cl:160)
    float t = tStart;
    float opacity, u2;
    do {
        float u1 = rand01_(rx, rc);
        t = fma_(-log_(u1), invTauMaxSampleBaseInterval, t);
        // the fetched value cannot influence the result once t > tEnd (the loop ends
        // whatever it is), so the fetch is skipped there; the RNG draw is not.
        opacity = 0.f;
        if (t <= tEnd) {
            float vs = sample_volume<DT, LINEAR>(V, fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z));
            opacity = sample_alpha(lut, wf, m1, m2, vs);
            last_sample = vs;
        }
        u2 = rand01_(rx, rc);
        ++steps;
    } while (u2 >= opacity && t <= tEnd);
    last_opacity = opacity;
    return t;
}

// The same walk with the next AHEAD steps' volume fetches in flight together.  A Woodcock step's POSITION depends on the
// random numbers only -- t += -log(u1) / 150 -- never on what was fetched; only whether the walk ends there does.  So the
// positions of steps j + 1 .. j + AHEAD are known before step j's sample arrives: their fetches are issued back to back, then
// the steps are decided in order; at the accepted step the RNG state is put back to what it was right behind that step's
// second draw.  Same draws in the same order, same operations on the same operands: the same t, sample, opacity, state and
// step count as woodcock() -- at the price of up to AHEAD - 1 fetches (and logs) past the end.  Two in flight pay everywhere
// (the second fetch hides behind the first's latency: -6 % on the config-2 trace); more cost registers and bandwidth.
template <int DT, int AHEAD, bool LINEAR = false>
CPM_DEV float woodcock_ahead(const VolDev& V, const float* lut, float wf, float m1, float m2, f3 o, f3 d, float tStart,
                             float tEnd, uint32_t& rx, uint32_t& rc, unsigned& steps, float& last_sample, float& last_opacity) {
    constexpr float invTauMaxSampleBaseInterval = 1.f / (1.f * 150.f);
    float t = tStart;
    for (;;) {
        float tj[AHEAD], u2j[AHEAD], vs[AHEAD];
        uint32_t sx[AHEAD], sc[AHEAD];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
            const float u1 = rand01_(rx, rc);
            t = fma_(-log_(u1), invTauMaxSampleBaseInterval, t);
            tj[j] = t;
            u2j[j] = rand01_(rx, rc);
            sx[j] = rx; sc[j] = rc;
            // beyond tEnd (or t = inf / NaN) nothing is read from the fetch: a safe address instead of the walk's
            const bool in = t <= tEnd;
            vs[j] = sample_volume<DT, LINEAR>(V, in ? fma_(t, d.x, o.x) : 0.f, in ? fma_(t, d.y, o.y) : 0.f, in ? fma_(t, d.z, o.z) : 0.f);
        }
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
            const bool in = tj[j] <= tEnd;
            float opacity = 0.f;
            if (in) { opacity = sample_alpha(lut, wf, m1, m2, vs[j]); last_sample = vs[j]; }
            ++steps;
            if (!(u2j[j] >= opacity && in)) {
                rx = sx[j]; rc = sc[j];
                last_opacity = opacity;
                return tj[j];
            }
        }
    }
}

// One light sample's walk: photontracer.cl:129-215 from the loaded sample on, in three pieces: walk_init -- everything before the
// loop; walk_segment -- ONE turn of the loop `while (scatterEvent)` (a Woodcock walk, the record of the interaction it ends in,
// the decision to scatter on); walk_finish -- the sentinel records, the RNG write-back, the importance reset.  trace_photon runs
// them back to back.  (The pieces exist because a tracer with one launch per interaction was built on them -- walks parked in a
// queue between launches -- and measured slower: docs/EXPERIMENTS.md, round 4.)
struct WalkState {
    f3 origin, direction, power;
    float tStart, tEnd, th, ph;   // (th, ph) = encodeDirection(direction)
    uint32_t rx, rc, nInteractions;
    bool scatterEvent;
};

// `direction` = decodeDirection(l1.z, l1.w) and (th, ph) = encodeDirection(direction), evaluated by the caller (once per workgroup
// for a directional light, per lane otherwise: the same operations on the same inputs either way).
template <int DT, bool SINGLE, int AHEAD, bool LINEAR = false>
CPM_DEV void walk_init(const TraceArgs& A, const float* lut, float4 l0, float4 l1, float2 ip, uint2 rs, f3 direction, float th, float ph,
                       unsigned& steps, WalkState& S) {
    const uint32_t maxInteractions = SINGLE ? 1u : (uint32_t)A.p.max_interactions;
    S.rx = rs.x; S.rc = rs.y;
    S.nInteractions = 0;
    S.origin = { l0.x, l0.y, l0.z };
    S.direction = direction;
    S.th = th; S.ph = ph;
    float mi = (float)maxInteractions;
    S.power = { l0.w, l1.x, l1.y };
    if (maxInteractions != 1) { S.power.x = S.power.x / mi; S.power.y = S.power.y / mi; S.power.z = S.power.z / mi; }  // x / 1.0f == x
    S.tStart = ip.x; S.tEnd = ip.y;
    S.scatterEvent = S.tStart < S.tEnd;
    const float wf = A.tf_wf, m1 = A.tf_m1, m2 = A.tf_m2;
    if (!SINGLE && (A.p.flags & CPM_TRACE_NO_SINGLE_SCATTERING)) {  // photontracer.cl:143-157
        float vs_unused, op_unused;
        float t = AHEAD > 1 ? woodcock_ahead<DT, (AHEAD > 1 ? AHEAD : 2), LINEAR>(A.vol, lut, wf, m1, m2, S.origin, S.direction, S.tStart, S.tEnd, S.rx, S.rc, steps, vs_unused, op_unused)
                            : woodcock<DT, LINEAR>(A.vol, lut, wf, m1, m2, S.origin, S.direction, S.tStart, S.tEnd, S.rx, S.rc, steps, vs_unused, op_unused);
        if (S.scatterEvent) {
            S.origin.x = fma_(t, S.direction.x, S.origin.x);
            S.origin.y = fma_(t, S.direction.y, S.origin.y);
            S.origin.z = fma_(t, S.direction.z, S.origin.z);
            S.tStart = 0.f; S.tEnd = kFltMax;
            float u1 = rand01_(S.rx, S.rc), u2 = rand01_(S.rx, S.rc);
            float pdf;
            S.direction = phase_sample(A.p.shading_type, A.p.material[0], S.direction, u1, u2, &pdf);
            encode_direction_(S.direction, S.th, S.ph);
            S.scatterEvent = ray_box_(A.bmin, A.bmax, S.origin, S.direction, S.tStart, S.tEnd);
            S.power.x = S.power.x / pdf; S.power.y = S.power.y / pdf; S.power.z = S.power.z / pdf;
            S.tStart = S.tStart + 0.5f * A.p.step_size;
        }
    }
}

// What follows a Woodcock walk that ended at t with the accepted step's sample and alpha (photontracer.cl:161-196): the collision
// point, the power division, the decision to scatter on, the record, and -- when scattering -- the new direction, its encoding and
// the slab test.  S.scatterEvent says whether another walk follows.
template <bool SINGLE>
CPM_DEV void walk_interact(const TraceArgs& A, const float* lut, const float* luts, int threadId, float t, float volumeSample, float colorW, WalkState& S) {
    const int photonOffset = A.p.photon_offset;
    const uint32_t maxInteractions = SINGLE ? 1u : (uint32_t)A.p.max_interactions;
    const size_t totalPhotons = (size_t)A.p.total_photons;
    const float wf = A.tf_wf, m1 = A.tf_m1, m2 = A.tf_m2;
    S.scatterEvent = t <= S.tEnd;
    if (S.scatterEvent) {
        S.origin.x = fma_(t, S.direction.x, S.origin.x);
        S.origin.y = fma_(t, S.direction.y, S.origin.y);
        S.origin.z = fma_(t, S.direction.z, S.origin.z);
        size_t photonId = (size_t)photonOffset + S.nInteractions * totalPhotons + (size_t)threadId;
        // (th, ph) = encodeDirection(direction) (photontracer.cl:167): current, see above.
        // The reference samples the volume and the TF again at the collision point
        // (photontracer.cl:170-173).  The accepted Woodcock iteration sampled exactly that point --
        // fma(t, d, o) with the same t, d, o -- so its volume sample and alpha ARE those values.
        float dv = max_(colorW, 0.01f);
        S.power.x = S.power.x / dv; S.power.y = S.power.y / dv; S.power.z = S.power.z / dv;
        ++S.nInteractions;
        bool scatter = false;
        float scatteringAlbedo = 0.f;
        if (S.nInteractions < maxInteractions) {  // the albedo is only read behind this test (photontracer.cl:179)
            float scatW = (luts == lut) ? colorW : sample_alpha(luts, wf, m1, m2, volumeSample);
            scatteringAlbedo = scatW / (scatW + colorW);
            scatter = rand01_(S.rx, S.rc) < scatteringAlbedo;
        }
        if (scatter) {
            S.power.x *= scatteringAlbedo; S.power.y *= scatteringAlbedo; S.power.z *= scatteringAlbedo;
            write_photon(A, photonId, S.origin, S.power, S.th, S.ph);
            S.tStart = 0.f; S.tEnd = kFltMax;
            float u1 = rand01_(S.rx, S.rc), u2 = rand01_(S.rx, S.rc);
            S.direction = phase_sample(A.p.shading_type, A.p.material[0], S.direction, u1, u2, nullptr);
            encode_direction_(S.direction, S.th, S.ph);
            S.scatterEvent = ray_box_(A.bmin, A.bmax, S.origin, S.direction, S.tStart, S.tEnd);
            S.tStart = S.tStart + 0.5f * A.p.step_size;
        } else {
            write_photon(A, photonId, S.origin, S.power, S.th, S.ph);
            S.power.x = S.power.y = S.power.z = kFltMax;  // read by the recomputation detector
            S.scatterEvent = false;
        }
    }
}

// one turn of photontracer.cl:158-197; call while S.scatterEvent
template <int DT, bool SINGLE, int AHEAD, bool LINEAR = false>
CPM_DEV void walk_segment(const TraceArgs& A, const float* lut, const float* luts, int threadId, unsigned& steps, WalkState& S) {
    const float wf = A.tf_wf, m1 = A.tf_m1, m2 = A.tf_m2;
    float volumeSample = 0.f, colorW = 0.f;
    float t = AHEAD > 1 ? woodcock_ahead<DT, (AHEAD > 1 ? AHEAD : 2), LINEAR>(A.vol, lut, wf, m1, m2, S.origin, S.direction, S.tStart, S.tEnd, S.rx, S.rc, steps, volumeSample, colorW)
                        : woodcock<DT, LINEAR>(A.vol, lut, wf, m1, m2, S.origin, S.direction, S.tStart, S.tEnd, S.rx, S.rc, steps, volumeSample, colorW);
    walk_interact<SINGLE>(A, lut, luts, threadId, t, volumeSample, colorW, S);
}

template <bool SINGLE>
CPM_DEV void walk_finish(const TraceArgs& A, int threadId, const WalkState& S) {
    const int photonOffset = A.p.photon_offset;
    const uint32_t maxInteractions = SINGLE ? 1u : (uint32_t)A.p.max_interactions;
    const size_t totalPhotons = (size_t)A.p.total_photons;
    for (uint32_t i = S.nInteractions; i < maxInteractions; ++i) {  // photontracer.cl:199-209 (th, ph: the current direction)
        size_t photonId = (size_t)photonOffset + i * totalPhotons + (size_t)threadId;
        f3 p = { kFltMax, kFltMax, kFltMax };
        f3 pw = { S.power.x, kFltMax, kFltMax };
        write_photon(A, photonId, p, pw, S.th, S.ph);
    }
    if (A.p.flags & CPM_TRACE_PROGRESSIVE) reinterpret_cast<uint2*>(A.rng)[photonOffset + threadId] = make_uint2(S.rx, S.rc);  // :211-215
    if (A.reset_importances) A.reset_importances[photonOffset + threadId] = 2147483647u;  // resetPhotonImportance (tracercl.cpp:529)
}

// Writes the photon records (and sentinels) of sample `threadId`, its RNG state when progressive, its importance key when
// A.reset_importances is set.
template <int DT, bool SINGLE, int AHEAD = 1, bool LINEAR = false>
CPM_DEV void trace_photon(const TraceArgs& A, const float* lut, const float* luts, int threadId, float4 l0, float4 l1, float2 ip, uint2 rs,
                          f3 direction, float th, float ph, unsigned& steps) {
    WalkState S;
    walk_init<DT, SINGLE, AHEAD, LINEAR>(A, lut, l0, l1, ip, rs, direction, th, ph, steps, S);
    while (S.scatterEvent) walk_segment<DT, SINGLE, AHEAD, LINEAR>(A, lut, luts, threadId, steps, S);
    walk_finish<SINGLE>(A, threadId, S);
}

}  // namespace tracer

// kernel arguments of a trace from the ABI's arguments (validated); lds = dynamic LDS bytes of the LUT(s)
int make_trace_args(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                    const cpm_trace_params* params, tracer::TraceArgs& A, size_t& lds);
// A.vol.voxels of a launch: the footprint copy -- rebuilt first when it is stale and the launch covers all the samples -- or, for a
// re-trace of a few photons through a stale copy, the linear block (*linear = true: the LINEAR kernels)
int trace_volume_source(cpm_ctx* ctx, const cpm_volume* vol, bool sparse_launch, hipStream_t s, tracer::TraceArgs& A, bool* linear);

}  // namespace cpm
