/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 *
 * cpm_oracle_math.h -- the arithmetic contract, CPU side.
 *
 * The reference leaves its elementary functions to the OpenCL implementation
 * (native_log: progressivephotonmapping/cl/transmittance.cl:135; sin/cos/acos/
 * atan2 inside Inviwo's encodeDirection/decodeDirection, host twin at
 * progressivephotonmapping/photondata.cpp:100-117; image sampling weights).
 * A one-ulp difference in one free-flight distance flips an accept/reject and
 * changes a photon's whole path, so parity with "the OpenCL output" is only
 * defined once these are pinned.  The contract (DESIGN.md "Arithmetic
 * contract") pins them as explicit sequences of IEEE-754 binary32 operations:
 * + - * / sqrt, fma ONLY where written, round-to-nearest-even, no contraction.
 * The HIP kernels implement the same sequences (csrc/cpm_math.hip.h, written
 * independently); tests compare the two bit for bit.
 *
 * Compile with -ffp-contract=off and without -ffast-math.
 */
#ifndef CPM_ORACLE_MATH_H
#define CPM_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float om_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline uint32_t om_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float om_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* min/max that return the non-NaN operand (IEEE minNum/maxNum = v_min_f32/v_max_f32) */
static inline float om_min(float a, float b) { return fminf(a, b); }
static inline float om_max(float a, float b) { return fmaxf(a, b); }

/* lerp exact at both ends: a == 0 -> x, a == 1 -> y */
static inline float om_lerp(float x, float y, float a) { return om_fma(a, y, om_fma(-a, x, x)); }

/* natural log for x in [0, +inf) normal; log(0) = -inf.  Cephes logf scheme. */
static inline float om_log(float x) {
    if (x == 0.0f) return -INFINITY;
    uint32_t ix = om_bits(x);
    int e = (int)(ix >> 23) - 127;
    float m = om_float((ix & 0x007fffffu) | 0x3f800000u); /* [1, 2) */
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292E-2f;
    p = om_fma(p, f, -1.1514610310E-1f);
    p = om_fma(p, f, 1.1676998740E-1f);
    p = om_fma(p, f, -1.2420140846E-1f);
    p = om_fma(p, f, 1.4249322787E-1f);
    p = om_fma(p, f, -1.6668057665E-1f);
    p = om_fma(p, f, 2.0000714765E-1f);
    p = om_fma(p, f, -2.4999993993E-1f);
    p = om_fma(p, f, 3.3333331174E-1f);
    float y = f * z;
    y = y * p;
    float fe = (float)e;
    y = om_fma(fe, -2.12194440e-4f, y);
    y = om_fma(-0.5f, z, y);
    float r = f + y;
    r = om_fma(fe, 0.693359375f, r);
    return r;
}

/* sin and cos for |x| <= 2*pi (angles of encoded directions). */
static inline void om_sincos(float x, float* s, float* c) {
    float kf = rintf(x * 0.636619772f); /* round-to-nearest-even of x * 2/pi */
    int k = (int)kf;
    float r = om_fma(kf, -1.5703125f, x);
    r = om_fma(kf, -4.837512969970703125e-4f, r);
    r = om_fma(kf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = om_fma(-1.9515295891E-4f, z, 8.3321608736E-3f);
    sp = om_fma(sp, z, -1.6666654611E-1f);
    float sr = om_fma(sp * z, r, r);
    float cp = om_fma(2.443315711809948E-5f, z, -1.388731625493765E-3f);
    cp = om_fma(cp, z, 4.166664568298827E-2f);
    float cr = om_fma(cp * z, z, om_fma(-0.5f, z, 1.0f));
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

static inline float om_asin_poly(float a, float z) {
    float p = 4.2163199048E-2f;
    p = om_fma(p, z, 2.4181311049E-2f);
    p = om_fma(p, z, 4.5470025998E-2f);
    p = om_fma(p, z, 7.4953002686E-2f);
    p = om_fma(p, z, 1.6666752422E-1f);
    return om_fma(p * z, a, a);
}

/* acos with the argument clamped to [-1, 1] (the host twin clamps:
 * progressivephotonmapping/photondata.cpp:107). */
static inline float om_acos(float x) {
    x = om_min(om_max(x, -1.0f), 1.0f);
    float ax = fabsf(x);
    if (ax <= 0.5f) {
        float r = om_asin_poly(x, x * x);
        return 1.57079632679489662f - r;
    }
    float z = om_fma(-0.5f, ax, 0.5f);
    float a = sqrtf(z);
    float r = om_asin_poly(a, z);
    r = r + r;
    return x > 0.0f ? r : 3.14159265358979324f - r;
}

static inline float om_atan_pos(float t) { /* t >= 0 */
    float y0;
    if (t > 2.414213562373095f) { y0 = 1.57079632679489662f; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y0 = 0.785398163397448310f; t = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = om_fma(p, z, -1.38776856032E-1f);
    p = om_fma(p, z, 1.99777106478E-1f);
    p = om_fma(p, z, -3.33329491539E-1f);
    float r = om_fma(p * z, t, t);
    return y0 + r;
}

static inline float om_atan2(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return 1.57079632679489662f;
        if (y < 0.0f) return -1.57079632679489662f;
        return 0.0f;
    }
    float q = y / x;
    float r = om_atan_pos(fabsf(q));
    if (q < 0.0f) r = -r;
    if (x < 0.0f) r = (y >= 0.0f) ? r + 3.14159265358979324f : r - 3.14159265358979324f;
    return r;
}

#endif
