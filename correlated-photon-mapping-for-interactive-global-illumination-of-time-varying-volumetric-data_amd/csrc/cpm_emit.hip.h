// cpm_emit.hip.h -- the emission chain as device functions: lattice sample (E1) -> light sample (E3 / E5) -> entry / exit
// of the volume's box (E4).  Used by the stand-alone emitter kernels (cpm_rng_emission.hip: buffers for every consumer) and
// by the tracer's emitted mode (cpm_trace.hip: the same values in registers, not read from those buffers) -- one
// definition, the same operations on the same inputs, the same bits.
#pragma once
#include "cpm_ctx.h"

namespace cpm {

struct Light {
    float radiance[3];
    float a[3];  // direction (directional) or position (point)
    float origin[3], u[3], v[3];
    float area;
};

struct Box { float mn[3], mx[3]; };

// importancesamplingcl/cl/uniformsamplegenerator2d.cl:35-52
CPM_DEV float4 lattice_sample_(int i, float dimx, float dimy) {
    float fi = (float)i;
    float cx = __builtin_fmodf(fi, dimx);
    float cy = fi / dimx;  // row coordinate not floored
    return make_float4((0.5f + cx) / dimx, (0.5f + cy) / dimy, 0.f, 1.f);
}

// lightcl/cl/directionallightsampler.cl:38-63 without the direction (the same for every sample: directional_hint_)
CPM_DEV void directional_sample_(const Light& L, float4 s, f3& o, f3& pw) {
    o = { fma_(L.v[0], s.y, fma_(L.u[0], s.x, L.origin[0])),
          fma_(L.v[1], s.y, fma_(L.u[1], s.x, L.origin[1])),
          fma_(L.v[2], s.y, fma_(L.u[2], s.x, L.origin[2])) };
    float pdf = s.w / L.area;
    pw = { L.radiance[0] / pdf, L.radiance[1] / pdf, L.radiance[2] / pdf };
}

// What the tracer derives from a directional light's (theta, phi), once per light instead of once per sample:
// hint = { theta0, phi0 = encode(direction); d1 = decode(theta0, phi0); theta1, phi1 = encode(d1); 1 }
CPM_DEV void directional_hint_(const Light& L, float* hint) {
    const f3 d0 = { L.a[0], L.a[1], L.a[2] };
    float t0, p0, t1, p1;
    encode_direction_(d0, t0, p0);
    const f3 d1 = decode_direction_(t0, p0);
    encode_direction_(d1, t1, p1);
    hint[0] = t0; hint[1] = p0; hint[2] = d1.x; hint[3] = d1.y; hint[4] = d1.z; hint[5] = t1; hint[6] = p1;
    hint[7] = 1.0f;
}

// build-defined point-light emitter
CPM_DEV void point_sample_(const Light& L, float4 s, f3& o, f3& pw, float& th, float& ph) {
    float z = fma_(-2.0f, s.x, 1.0f);
    float r = __builtin_sqrtf(max_(0.0f, fma_(-z, z, 1.0f)));
    float sp, cp;
    sincos_(kTwoPi * s.y, sp, cp);
    f3 d = { r * cp, r * sp, z };
    float pdf = s.w * kInv4Pi;
    pw = { L.radiance[0] / pdf, L.radiance[1] / pdf, L.radiance[2] / pdf };
    o = { L.a[0], L.a[1], L.a[2] };
    encode_direction_(d, th, ph);
}

// lightcl/cl/intersection/lightsamplemeshintersection.cl:37-58 for the cube proxy; d = decodeDirection of the sample's angles
CPM_DEV float2 box_entry_exit_(const Box& b, f3 o, f3 d) {
    float t0 = 0.f, t1 = kFltMax;
    bool hit = ray_box_(b.mn, b.mx, o, d, t0, t1);
    if (!hit) { t0 = 0.f; t1 = -1.f; }
    return make_float2(t0, t1);
}

}  // namespace cpm
