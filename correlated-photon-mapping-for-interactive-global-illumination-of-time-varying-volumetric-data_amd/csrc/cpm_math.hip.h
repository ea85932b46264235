// cpm_math.hip.h -- device side of the arithmetic contract (DESIGN.md "Arithmetic
// contract"): every elementary function the path needs, as an explicit sequence of
// IEEE-754 binary32 operations (+ - * / sqrt, fma only where written).  The reference
// leaves these to the OpenCL implementation (native_log:
// progressivephotonmapping/cl/transmittance.cl:135; sin/cos/acos/atan2 in
// encodeDirection/decodeDirection, host twin progressivephotonmapping/photondata.cpp:100-117).
// Compiled with -ffp-contract=off so that nothing fuses or re-associates behind our back;
// division and sqrt are hipcc's correctly rounded forms (default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpm {

#define CPM_DEV __device__ __forceinline__

constexpr float kFltMax = 3.402823466e+38f;
constexpr float kPi = 3.14159265358979324f;
constexpr float kHalfPi = 1.57079632679489662f;
constexpr float kQuarterPi = 0.785398163397448310f;
constexpr float kTwoPi = 6.28318530717958648f;
constexpr float kInv4Pi = 0.0795774715459476679f;  // isotropicPhaseFunction()

CPM_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
CPM_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
CPM_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
CPM_DEV float lerp_(float x, float y, float a) { return fma_(a, y, fma_(-a, x, x)); }

// natural log, x in [0, inf) normal; log(0) = -inf
CPM_DEV float log_(float x) {
    if (x == 0.0f) return -__builtin_inff();
    uint32_t ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292E-2f;
    p = fma_(p, f, -1.1514610310E-1f);
    p = fma_(p, f, 1.1676998740E-1f);
    p = fma_(p, f, -1.2420140846E-1f);
    p = fma_(p, f, 1.4249322787E-1f);
    p = fma_(p, f, -1.6668057665E-1f);
    p = fma_(p, f, 2.0000714765E-1f);
    p = fma_(p, f, -2.4999993993E-1f);
    p = fma_(p, f, 3.3333331174E-1f);
    float y = f * z;
    y = y * p;
    float fe = (float)e;
    y = fma_(fe, -2.12194440e-4f, y);
    y = fma_(-0.5f, z, y);
    float r = f + y;
    r = fma_(fe, 0.693359375f, r);
    return r;
}

// sin and cos, |x| <= 2*pi
CPM_DEV void sincos_(float x, float& s, float& c) {
    float kf = __builtin_rintf(x * 0.636619772f);
    int k = (int)kf;
    float r = fma_(kf, -1.5703125f, x);
    r = fma_(kf, -4.837512969970703125e-4f, r);
    r = fma_(kf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = fma_(-1.9515295891E-4f, z, 8.3321608736E-3f);
    sp = fma_(sp, z, -1.6666654611E-1f);
    float sr = fma_(sp * z, r, r);
    float cp = fma_(2.443315711809948E-5f, z, -1.388731625493765E-3f);
    cp = fma_(cp, z, 4.166664568298827E-2f);
    float cr = fma_(cp * z, z, fma_(-0.5f, z, 1.0f));
    int q = k & 3;
    float ss = (q & 1) ? cr : sr;
    float cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

CPM_DEV float asin_poly_(float a, float z) {
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    return fma_(p * z, a, a);
}

CPM_DEV float acos_(float x) {
    x = min_(max_(x, -1.0f), 1.0f);
    float ax = __builtin_fabsf(x);
    if (ax <= 0.5f) {
        float r = asin_poly_(x, x * x);
        return kHalfPi - r;
    }
    float z = fma_(-0.5f, ax, 0.5f);
    float a = __builtin_sqrtf(z);
    float r = asin_poly_(a, z);
    r = r + r;
    return x > 0.0f ? r : kPi - r;
}

CPM_DEV float atan_pos_(float t) {
    float y0;
    if (t > 2.414213562373095f) { y0 = kHalfPi; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y0 = kQuarterPi; t = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = fma_(p, z, -1.38776856032E-1f);
    p = fma_(p, z, 1.99777106478E-1f);
    p = fma_(p, z, -3.33329491539E-1f);
    float r = fma_(p * z, t, t);
    return y0 + r;
}

CPM_DEV float atan2_(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return kHalfPi;
        if (y < 0.0f) return -kHalfPi;
        return 0.0f;
    }
    float q = y / x;
    float r = atan_pos_(__builtin_fabsf(q));
    if (q < 0.0f) r = -r;
    if (x < 0.0f) r = (y >= 0.0f) ? r + kPi : r - kPi;
    return r;
}

struct f3 { float x, y, z; };

CPM_DEV float dot3_(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
CPM_DEV f3 cross3_(f3 a, f3 b) {
    f3 r;
    r.x = fma_(a.y, b.z, -(a.z * b.y));
    r.y = fma_(a.z, b.x, -(a.x * b.z));
    r.z = fma_(a.x, b.y, -(a.y * b.x));
    return r;
}

// Inviwo encodeDirection / decodeDirection (host twin photondata.cpp:100-117)
CPM_DEV void encode_direction_(f3 d, float& theta, float& phi) {
    theta = acos_(d.z);
    phi = atan2_(d.y, d.x);
}
CPM_DEV f3 decode_direction_(float theta, float phi) {
    float st, ct, sp, cp;
    sincos_(theta, st, ct);
    sincos_(phi, sp, cp);
    f3 d = { st * cp, st * sp, ct };
    return d;
}

// Inviwo transformPoint(float16 m, float3 p) for an axis-aligned scale + translate
// matrix: only the diagonal and the translation column are read; the (zero)
// off-diagonal terms a full column-major product would add are exact no-ops
// (fma(0, y, t) == t for finite y), so this equals the oracle's full product.
struct Affine { float sx, sy, sz, tx, ty, tz; };
CPM_DEV f3 transform_(const Affine& m, f3 p) {
    f3 r = { fma_(m.sx, p.x, m.tx), fma_(m.sy, p.y, m.ty), fma_(m.sz, p.z, m.tz) };
    return r;
}

// Inviwo rayBoxIntersection: slab test clipping [t0, t1]
CPM_DEV bool ray_box_(const float* pmin, const float* pmax, f3 o, f3 d, float& t0, float& t1) {
    float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    float nx = (pmin[0] - o.x) * ix, fx = (pmax[0] - o.x) * ix;
    float ny = (pmin[1] - o.y) * iy, fy = (pmax[1] - o.y) * iy;
    float nz = (pmin[2] - o.z) * iz, fz = (pmax[2] - o.z) * iz;
    float tnx = min_(nx, fx), tfx = max_(nx, fx);
    float tny = min_(ny, fy), tfy = max_(ny, fy);
    float tnz = min_(nz, fz), tfz = max_(nz, fz);
    t0 = max_(t0, max_(tnx, max_(tny, tnz)));
    t1 = min_(t1, min_(tfx, min_(tfy, tfz)));
    return t0 <= t1;
}

// Epanechnikov density kernel (progressivephotonmapping/cl/densityestimationkernel.cl:43-60)
CPM_DEV float density_kernel_(float x) { return x <= 1.f ? 0.75f * (1.f - x * x) : 0.f; }

// MWC64X (rndgenmwc64x/cl/random.cl:58-68,85-95)
constexpr uint32_t kMwcA = 4294883355u;
CPM_DEV uint32_t mwc_next_(uint32_t& x, uint32_t& c) {
    uint32_t res = x ^ c;
    // x' = lo(A x + c), c' = hi(A x + c): one v_mad_u64_u32 instead of mul_lo + mul_hi + add + addc
    // (32-bit integer multiplies issue at quarter rate: this is 8 of them per Woodcock step otherwise)
    const uint64_t t = (uint64_t)kMwcA * (uint64_t)x + (uint64_t)c;
    x = (uint32_t)t;
    c = (uint32_t)(t >> 32);
    return res;
}
// random_01: uint -> float (RNE) / 4294967295.0f; the divisor rounds to 2^32, so the
// quotient is the exact product with 2^-32.
CPM_DEV float rand01_(uint32_t& x, uint32_t& c) { return (float)mwc_next_(x, c) * 2.3283064365386963e-10f; }

}  // namespace cpm
