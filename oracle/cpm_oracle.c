/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 *
 * cpm_oracle.c -- CPU restatement of the reference path; see cpm_oracle.h for
 * what is pinned against the reference and what is not ("[INVIWO]" marks
 * semantics restated from the OpenCL 1.2 specification / Inviwo call sites).
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 */
#include "cpm_oracle.h"
#include "cpm_oracle_math.h"

#include <float.h>
#include <stdlib.h>

#if defined(__GNUC__) && defined(__x86_64__)
#define CPMO_CLONES __attribute__((target_clones("default", "avx2,fma")))
#else
#define CPMO_CLONES
#endif

static int g_threads = 1;
void cpmo_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int cpmo_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------ small vector helpers */

typedef struct { float x, y, z; } v3;

static inline float dot3(v3 a, v3 b) { return om_fma(a.z, b.z, om_fma(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    v3 r;
    r.x = om_fma(a.y, b.z, -(a.z * b.y));
    r.y = om_fma(a.z, b.x, -(a.x * b.z));
    r.z = om_fma(a.x, b.y, -(a.y * b.x));
    return r;
}
/* [INVIWO] transformPoint(float16 m, float3 p): column-major 4x4 times (p, 1), xyz. */
static inline v3 transform_point(const float* m, v3 p) {
    v3 r;
    r.x = om_fma(m[0], p.x, om_fma(m[4], p.y, om_fma(m[8], p.z, m[12])));
    r.y = om_fma(m[1], p.x, om_fma(m[5], p.y, om_fma(m[9], p.z, m[13])));
    r.z = om_fma(m[2], p.x, om_fma(m[6], p.y, om_fma(m[10], p.z, m[14])));
    return r;
}

/* ------------------------------------------------------------------ math, exposed */

float cpmo_log(float x) { return om_log(x); }
void cpmo_sincos(float x, float* s, float* c) { om_sincos(x, s, c); }
float cpmo_acos(float x) { return om_acos(x); }
float cpmo_atan2(float y, float x) { return om_atan2(y, x); }

/* [INVIWO] encodeDirection / decodeDirection; host twin:
 * progressivephotonmapping/photondata.cpp:100-117. */
static inline void encode_direction(v3 d, float* theta, float* phi) {
    *theta = om_acos(d.z);
    *phi = om_atan2(d.y, d.x);
}
static inline v3 decode_direction(float theta, float phi) {
    float st, ct, sp, cp;
    om_sincos(theta, &st, &ct);
    om_sincos(phi, &sp, &cp);
    v3 d = { st * cp, st * sp, ct };
    return d;
}
void cpmo_encode_direction(const float d[3], float angles[2]) {
    v3 v = { d[0], d[1], d[2] };
    encode_direction(v, &angles[0], &angles[1]);
}
void cpmo_decode_direction(const float angles[2], float d[3]) {
    v3 v = decode_direction(angles[0], angles[1]);
    d[0] = v.x; d[1] = v.y; d[2] = v.z;
}

/* progressivephotonmapping/cl/densityestimationkernel.cl:43-60 (Epanechnikov) */
static inline float density_kernel(float x) {
    if (x <= 1.f) return (0.75f) * (1.f - x * x);
    return 0.f;
}
float cpmo_density_kernel(float x) { return density_kernel(x); }

/* ------------------------------------------------------------------ RNG */

/* rndgenmwc64x/cl/random.cl:46-47 */
#define MWC64X_A 4294883355u
#define MWC64X_M 18446383549859758079ull

/* rndgenmwc64x/cl/random.cl:58-68 (MWC64X_Step), :85-90 (MWC64X_NextUint) */
static inline uint32_t mwc_next(uint32_t* x, uint32_t* c) {
    uint32_t X = *x, C = *c;
    uint32_t res = X ^ C;
    uint32_t Xn = MWC64X_A * X + C;
    uint32_t carry = (uint32_t)(Xn < C);
    uint32_t Cn = (uint32_t)(((uint64_t)MWC64X_A * X) >> 32) + carry; /* mad_hi(A, X, carry) */
    *x = Xn;
    *c = Cn;
    return res;
}
/* rndgenmwc64x/cl/random.cl:92-95: uint -> float (RNE), divided by 4294967295.0f (== 2^32) */
static inline float rand01(uint32_t* x, uint32_t* c) { return (float)mwc_next(x, c) / 4294967295.0f; }

void cpmo_mwc64x_next(uint32_t* x, uint32_t* c, uint32_t* out_uint, float* out_01) {
    uint32_t u = mwc_next(x, c);
    if (out_uint) *out_uint = u;
    if (out_01) *out_01 = (float)u / 4294967295.0f;
}

/* rndgenmwc64x/cl/skip_mwc.cl:40-46 */
static uint64_t add_mod64(uint64_t a, uint64_t b, uint64_t M) {
    uint64_t v = a + b;
    if ((v >= M) || (v < a)) v = v - M;
    return v;
}
/* rndgenmwc64x/cl/skip_mwc.cl:54-64 */
static uint64_t mul_mod64(uint64_t a, uint64_t b, uint64_t M) {
    uint64_t r = 0;
    while (a != 0) {
        if (a & 1) r = add_mod64(r, b, M);
        b = add_mod64(b, b, M);
        a = a >> 1;
    }
    return r;
}
/* rndgenmwc64x/cl/skip_mwc.cl:71-81 */
static uint64_t pow_mod64(uint64_t a, uint64_t e, uint64_t M) {
    uint64_t sqr = a, acc = 1;
    while (e != 0) {
        if (e & 1) acc = mul_mod64(acc, sqr, M);
        sqr = mul_mod64(sqr, sqr, M);
        e = e >> 1;
    }
    return acc;
}

/* rndgenmwc64x/cl/skip_mwc.cl:91-105 (MWC_SeedImpl_Mod64, vecSize 1, vecOffset 0),
 * rndgenmwc64x/cl/random.cl:77-82, kernel rndgenmwc64x/cl/randstategen.cl:39-47 */
void cpmo_seed_streams(uint32_t* state, size_t n, uint64_t gap) {
    const uint64_t BASEID = 4077358422479273989ull;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        uint64_t dist = (uint64_t)state[2 * i] + (uint64_t)i * gap;
        uint64_t m = pow_mod64(MWC64X_A, dist, MWC64X_M);
        uint64_t x = mul_mod64(BASEID, m, MWC64X_M);
        state[2 * i] = (uint32_t)(x / MWC64X_A);
        state[2 * i + 1] = (uint32_t)(x % MWC64X_A);
    }
}

/* rndgenmwc64x/cl/randomnumbergenerator.cl:34-50 (known-answer harness) */
void cpmo_random_fill(uint32_t* state, size_t n, int draws, float* out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = state[2 * i], c = state[2 * i + 1];
        for (int k = 0; k < draws; ++k) out[i + (size_t)k * n] = rand01(&x, &c);
        state[2 * i] = x;
        state[2 * i + 1] = c;
    }
}

/* glibc random_r TYPE_3 (degree 31, separation 3) as used by srand()/rand():
 * what rndgenmwc64x/mwc64xseedgenerator.cpp:56-64 draws on a glibc host. */
void cpmo_glibc_rand_sequence(uint32_t seed, uint32_t* out, size_t n) {
    size_t total = 344 + n;
    int32_t* r = (int32_t*)malloc(total * sizeof(int32_t));
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
        int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (int32_t)w;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (size_t i = 34; i < total; ++i) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    for (size_t k = 0; k < n; ++k) out[k] = ((uint32_t)r[344 + k]) >> 1;
    free(r);
}

/* ------------------------------------------------------------------ sampling [INVIWO] */

static inline float fetch_voxel_raw(const cpmo_volume* v, int x, int y, int z) {
    size_t idx = (size_t)x + (size_t)v->dims[0] * ((size_t)y + (size_t)v->dims[1] * (size_t)z);
    switch (v->dtype) {
        case CPMO_U8: return (float)((const uint8_t*)v->voxels)[idx];
        case CPMO_U16: return (float)((const uint16_t*)v->voxels)[idx];
        default: return ((const float*)v->voxels)[idx];
    }
}
static inline float norm_scale(int dtype) {
    return dtype == CPMO_U8 ? (1.0f / 255.0f) : (dtype == CPMO_U16 ? (1.0f / 65535.0f) : 1.0f);
}

/* OpenCL 1.2 s8.2 linear filter, normalised coords, CLK_ADDRESS_CLAMP_TO_EDGE:
 * u = s*w, i0 = floor(u - 0.5), i1 = i0 + 1 (both clamped to [0, w-1]), a = frac(u - 0.5).
 * Restated with the clamp applied to the coordinate instead of the indices:
 *   u' = clamp(u - 0.5, 0, w - 1), i0 = min(floor(u'), w - 2), i1 = i0 + 1, a = u' - i0,
 * which selects the same texels with the same weights inside the image and returns the
 * edge texel exactly outside it (om_lerp is exact at a = 0 and a = 1); NaN maps to 0. */
static inline void linear_coord(float s, int dim, int* i0, int* i1, float* a) {
    float u = om_fma(s, (float)dim, -0.5f);
    u = om_max(u, 0.0f);
    u = om_min(u, (float)(dim - 1));
    float fl = floorf(u);
    fl = om_max(om_min(fl, (float)(dim - 2)), 0.0f);
    *a = u - fl;
    int j = (int)fl;
    *i0 = j;
    *i1 = j + 1 > dim - 1 ? dim - 1 : j + 1;
}

/* getNormalizedVoxel(volume, params, pos).x =
 *   (read_imagef(volume, smpNormClampEdgeLinear, pos).x + formatOffset) * (1 - formatScaling)
 * (use sites progressivephotonmapping/cl/transmittance.cl:137, cl/photontracer.cl:170). */
float cpmo_sample_volume(const cpmo_volume* v, float px, float py, float pz) {
    int x0, x1, y0, y1, z0, z1;
    float ax, ay, az;
    linear_coord(px, v->dims[0], &x0, &x1, &ax);
    linear_coord(py, v->dims[1], &y0, &y1, &ay);
    linear_coord(pz, v->dims[2], &z0, &z1, &az);
    float c00 = om_lerp(fetch_voxel_raw(v, x0, y0, z0), fetch_voxel_raw(v, x1, y0, z0), ax);
    float c10 = om_lerp(fetch_voxel_raw(v, x0, y1, z0), fetch_voxel_raw(v, x1, y1, z0), ax);
    float c01 = om_lerp(fetch_voxel_raw(v, x0, y0, z1), fetch_voxel_raw(v, x1, y0, z1), ax);
    float c11 = om_lerp(fetch_voxel_raw(v, x0, y1, z1), fetch_voxel_raw(v, x1, y1, z1), ax);
    float c0 = om_lerp(c00, c10, ay);
    float c1 = om_lerp(c01, c11, ay);
    float c = om_lerp(c0, c1, az);
    float s = c * norm_scale(v->dtype);
    return (s + v->format_offset) * (1.0f - v->format_scaling);
}

/* read_imagef(tfData, smpNormClampEdgeLinear, (float2)(v, 0.5f)).w for a width x 1 image */
float cpmo_sample_tf_alpha(const float* tf_rgba, int width, float v) {
    int i0, i1;
    float a;
    linear_coord(v, width, &i0, &i1, &a);
    return om_lerp(tf_rgba[4 * i0 + 3], tf_rgba[4 * i1 + 3], a);
}

/* [INVIWO] rayBoxIntersection(bbox, o, d, &t0, &t1): slab test clipping [t0, t1]. */
static inline int ray_box(const float* pmin, const float* pmax, v3 o, v3 d, float* t0, float* t1) {
    float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    float nx = (pmin[0] - o.x) * ix, fx = (pmax[0] - o.x) * ix;
    float ny = (pmin[1] - o.y) * iy, fy = (pmax[1] - o.y) * iy;
    float nz = (pmin[2] - o.z) * iz, fz = (pmax[2] - o.z) * iz;
    float tnx = om_min(nx, fx), tfx = om_max(nx, fx);
    float tny = om_min(ny, fy), tfy = om_max(ny, fy);
    float tnz = om_min(nz, fz), tfz = om_max(nz, fz);
    *t0 = om_max(*t0, om_max(tnx, om_max(tny, tnz)));
    *t1 = om_min(*t1, om_min(tfx, om_min(tfy, tfz)));
    return *t0 <= *t1;
}

/* ------------------------------------------------------------------ emission */

/* importancesamplingcl/cl/uniformsamplegenerator2d.cl:35-52 (row coordinate not floored, Q14) */
void cpmo_uniform_samples_2d(int nx, int ny, float* s) {
    int n = nx * ny;
    float dx = (float)nx, dy = (float)ny;
    for (int i = 0; i < n; ++i) {
        float fi = (float)i;
        float cx = fmodf(fi, dx);
        float cy = fi / dx;
        s[4 * i + 0] = (0.5f + cx) / dx;
        s[4 * i + 1] = (0.5f + cy) / dy;
        s[4 * i + 2] = 0.f;
        s[4 * i + 3] = 1.f;
    }
}

/* lightcl/cl/directionallightsampler.cl:38-63 + writeLightSample
 * (lightcl/cl/datastructures/lightsample.cl:79-88) */
void cpmo_directional_light_samples(const float* s, int n, const float radiance[4],
                                    const float direction[4], const float o[4], const float u[4],
                                    const float v[4], float area, float* ls) {
    v3 dir = { direction[0], direction[1], direction[2] };
    float theta, phi;
    encode_direction(dir, &theta, &phi);
    for (int i = 0; i < n; ++i) {
        float sx = s[4 * i], sy = s[4 * i + 1], w = s[4 * i + 3];
        float pdf = w / area;
        ls[8 * i + 0] = om_fma(v[0], sy, om_fma(u[0], sx, o[0]));
        ls[8 * i + 1] = om_fma(v[1], sy, om_fma(u[1], sx, o[1]));
        ls[8 * i + 2] = om_fma(v[2], sy, om_fma(u[2], sx, o[2]));
        ls[8 * i + 3] = radiance[0] / pdf;
        ls[8 * i + 4] = radiance[1] / pdf;
        ls[8 * i + 5] = radiance[2] / pdf;
        ls[8 * i + 6] = theta;
        ls[8 * i + 7] = phi;
    }
}

/* Build-defined point-light emitter (shape after
 * importancesamplingcl/cl/light/light.cl:82-125: pdf = 1/(4 pi)). */
void cpmo_point_light_samples(const float* s, int n, const float radiance[4],
                              const float position[4], float* ls) {
    for (int i = 0; i < n; ++i) {
        float su = s[4 * i], sv = s[4 * i + 1], w = s[4 * i + 3];
        float z = om_fma(-2.0f, su, 1.0f);
        float r = sqrtf(om_max(0.0f, om_fma(-z, z, 1.0f)));
        float ph = 6.28318530717958648f * sv;
        float sp, cp;
        om_sincos(ph, &sp, &cp);
        v3 d = { r * cp, r * sp, z };
        float pdf = w * 0.0795774715459476679f;
        float theta, phi;
        encode_direction(d, &theta, &phi);
        ls[8 * i + 0] = position[0];
        ls[8 * i + 1] = position[1];
        ls[8 * i + 2] = position[2];
        ls[8 * i + 3] = radiance[0] / pdf;
        ls[8 * i + 4] = radiance[1] / pdf;
        ls[8 * i + 5] = radiance[2] / pdf;
        ls[8 * i + 6] = theta;
        ls[8 * i + 7] = phi;
    }
}

/* lightcl/cl/intersection/lightsamplemeshintersection.cl:37-58 for the cube proxy */
void cpmo_light_sample_box_intersection(const float* ls, int n, const float aabb[8], float* isect) {
    for (int i = 0; i < n; ++i) {
        v3 o = { ls[8 * i], ls[8 * i + 1], ls[8 * i + 2] };
        v3 d = decode_direction(ls[8 * i + 6], ls[8 * i + 7]);
        float t0 = 0.f, t1 = FLT_MAX;
        int hit = ray_box(aabb, aabb + 4, o, d, &t0, &t1);
        if (!hit) { t0 = 0.f; t1 = -1.f; }
        isect[2 * i] = t0;
        isect[2 * i + 1] = t1;
    }
}

/* Same kernel against a triangle list.  [INVIWO] rayMeshIntersection is restated as
 * Moeller-Trumbore per triangle; t0 = nearest hit with t >= 0 (0 when only one
 * hit: origin inside), t1 = farthest hit. */
void cpmo_light_sample_mesh_intersection(const float* vtx, const int32_t* idx, int n_indices,
                                         const float* ls, int n, float* isect) {
    for (int i = 0; i < n; ++i) {
        v3 o = { ls[8 * i], ls[8 * i + 1], ls[8 * i + 2] };
        v3 d = decode_direction(ls[8 * i + 6], ls[8 * i + 7]);
        float tmin = FLT_MAX, tmax = -1.f;
        int hits = 0;
        for (int t = 0; t + 2 < n_indices; t += 3) {
            const float* a = vtx + 3 * idx[t];
            const float* b = vtx + 3 * idx[t + 1];
            const float* c = vtx + 3 * idx[t + 2];
            v3 e1 = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
            v3 e2 = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
            v3 p = cross3(d, e2);
            float det = dot3(e1, p);
            if (fabsf(det) < 1e-12f) continue;
            float inv = 1.0f / det;
            v3 s = { o.x - a[0], o.y - a[1], o.z - a[2] };
            float uu = dot3(s, p) * inv;
            if (uu < 0.f || uu > 1.f) continue;
            v3 q = cross3(s, e1);
            float vv = dot3(d, q) * inv;
            if (vv < 0.f || uu + vv > 1.f) continue;
            float tt = dot3(e2, q) * inv;
            if (tt < 0.f) continue;
            ++hits;
            tmin = om_min(tmin, tt);
            tmax = om_max(tmax, tt);
        }
        float t0, t1;
        if (hits == 0) { t0 = 0.f; t1 = -1.f; }
        else if (hits == 1) { t0 = 0.f; t1 = tmax; }
        else { t0 = tmin; t1 = tmax; }
        isect[2 * i] = t0;
        isect[2 * i + 1] = t1;
    }
}

/* ------------------------------------------------------------------ trace */

typedef struct {
    const cpmo_volume* vol;
    const float* tf;
    const float* tfs;
    int tfw;
    uint64_t steps;
} trace_env;

/* progressivephotonmapping/cl/transmittance.cl:126-144 (SAMPLING_BASE_INTERVAL_RCP = 150, :40) */
static inline float woodcock(trace_env* e, v3 o, v3 d, float tStart, float tEnd, float tauMax,
                             uint32_t* rx, uint32_t* rc) {
    float invTauMaxSampleBaseInterval = 1.f / (tauMax * 150.f);
    float invTauMax = 1.f / (tauMax);
    float t = tStart;
    float opacity;
    float u2;
    do {
        float u1 = rand01(rx, rc);
        t = om_fma(-om_log(u1), invTauMaxSampleBaseInterval, t);
        float px = om_fma(t, d.x, o.x), py = om_fma(t, d.y, o.y), pz = om_fma(t, d.z, o.z);
        float volumeSample = cpmo_sample_volume(e->vol, px, py, pz);
        opacity = cpmo_sample_tf_alpha(e->tf, e->tfw, volumeSample);
        u2 = rand01(rx, rc);
        e->steps++;
    } while (u2 >= opacity * invTauMax && t <= tEnd);
    return t;
}

/* [INVIWO] sampleShadingFunction: build-defined phase-function sampling.
 * Henyey-Greenstein (g = material.x) or isotropic about the incoming direction. */
static inline float phase_cos(int type, float g, float u1) {
    if (type == CPMO_PHASE_ISOTROPIC || fabsf(g) < 1e-3f) return om_fma(-2.0f, u1, 1.0f);
    float g2 = g * g;
    float sq = (1.0f - g2) / om_fma(2.0f * g, u1, 1.0f - g);
    return (1.0f + g2 - sq * sq) / (2.0f * g);
}
static inline float phase_pdf(int type, float g, float cosT) {
    if (type == CPMO_PHASE_ISOTROPIC || fabsf(g) < 1e-3f) return 0.0795774715459476679f;
    float g2 = g * g;
    float den = om_fma(-2.0f * g, cosT, 1.0f + g2);
    return 0.0795774715459476679f * (1.0f - g2) / (den * sqrtf(den));
}
static inline v3 phase_sample(int type, float g, v3 w, float u1, float u2, float* pdf) {
    float cosT = phase_cos(type, g, u1);
    cosT = om_min(om_max(cosT, -1.0f), 1.0f);
    float sinT = sqrtf(om_max(0.0f, om_fma(-cosT, cosT, 1.0f)));
    float sp, cp;
    om_sincos(6.28318530717958648f * u2, &sp, &cp);
    v3 a;
    if (fabsf(w.z) < 0.999f) { a.x = 0; a.y = 0; a.z = 1; } else { a.x = 1; a.y = 0; a.z = 0; }
    v3 u = cross3(a, w);
    float il = 1.0f / sqrtf(dot3(u, u));
    u.x *= il; u.y *= il; u.z *= il;
    v3 v = cross3(w, u);
    float ku = sinT * cp, kv = sinT * sp;
    v3 d;
    d.x = om_fma(cosT, w.x, om_fma(kv, v.x, ku * u.x));
    d.y = om_fma(cosT, w.y, om_fma(kv, v.y, ku * u.y));
    d.z = om_fma(cosT, w.z, om_fma(kv, v.z, ku * u.z));
    if (pdf) *pdf = phase_pdf(type, g, cosT);
    return d;
}

static inline void write_photon(float* photons, size_t id, v3 p, v3 pw, float th, float ph) {
    float* q = photons + 8 * id;
    q[0] = p.x; q[1] = p.y; q[2] = p.z;
    q[3] = pw.x; q[4] = pw.y; q[5] = pw.z;
    q[6] = th; q[7] = ph;
}

/* progressivephotonmapping/cl/photontracer.cl:69-216 for one work-item */
static void trace_one(trace_env* e, const float aabb[8], const cpmo_trace_params* P,
                      const float* ls, const float* isect, int threadId, uint32_t* rng,
                      float* photons) {
    const int photonOffset = P->photon_offset;
    const uint32_t maxInteractions = (uint32_t)P->max_interactions;
    const size_t totalPhotons = (size_t)P->total_photons;
    uint32_t rx = rng[2 * (photonOffset + threadId)], rc = rng[2 * (photonOffset + threadId) + 1];
    uint32_t nInteractions = 0;
    const float* s = ls + 8 * (size_t)threadId;
    v3 origin = { s[0], s[1], s[2] };
    float mi = (float)maxInteractions;
    v3 power = { s[3] / mi, s[4] / mi, s[5] / mi };
    v3 direction = decode_direction(s[6], s[7]);
    float tStart = isect[2 * threadId], tEnd = isect[2 * threadId + 1];
    int scatterEvent = tStart < tEnd;

    if (P->flags & CPMO_TRACE_NO_SINGLE_SCATTERING) { /* :143-157 */
        float t = woodcock(e, origin, direction, tStart, tEnd, 1.f, &rx, &rc);
        if (scatterEvent) {
            origin.x = om_fma(t, direction.x, origin.x);
            origin.y = om_fma(t, direction.y, origin.y);
            origin.z = om_fma(t, direction.z, origin.z);
            tStart = 0.f; tEnd = FLT_MAX;
            float u1 = rand01(&rx, &rc), u2 = rand01(&rx, &rc);
            float pdf;
            direction = phase_sample(P->shading_type, P->material[0], direction, u1, u2, &pdf);
            scatterEvent = ray_box(aabb, aabb + 4, origin, direction, &tStart, &tEnd);
            power.x = power.x / pdf; power.y = power.y / pdf; power.z = power.z / pdf;
            tStart = tStart + 0.5f * P->step_size;
        }
    }
    while (scatterEvent) { /* :158-197 */
        float t = woodcock(e, origin, direction, tStart, tEnd, 1.f, &rx, &rc);
        scatterEvent = t <= tEnd;
        if (scatterEvent) {
            origin.x = om_fma(t, direction.x, origin.x);
            origin.y = om_fma(t, direction.y, origin.y);
            origin.z = om_fma(t, direction.z, origin.z);
            size_t photonId = (size_t)photonOffset + nInteractions * totalPhotons + (size_t)threadId;
            float th, ph;
            encode_direction(direction, &th, &ph);
            float volumeSample = cpmo_sample_volume(e->vol, origin.x, origin.y, origin.z);
            float colorW = cpmo_sample_tf_alpha(e->tf, e->tfw, volumeSample);
            float scatW = cpmo_sample_tf_alpha(e->tfs, e->tfw, volumeSample);
            float scatteringAlbedo = scatW / (scatW + colorW);
            float dv = om_max(colorW, 0.01f);
            power.x = power.x / dv; power.y = power.y / dv; power.z = power.z / dv;
            ++nInteractions;
            if (nInteractions < maxInteractions && rand01(&rx, &rc) < scatteringAlbedo) {
                power.x *= scatteringAlbedo; power.y *= scatteringAlbedo; power.z *= scatteringAlbedo;
                write_photon(photons, photonId, origin, power, th, ph);
                tStart = 0.f; tEnd = FLT_MAX;
                float u1 = rand01(&rx, &rc), u2 = rand01(&rx, &rc);
                direction = phase_sample(P->shading_type, P->material[0], direction, u1, u2, 0);
                scatterEvent = ray_box(aabb, aabb + 4, origin, direction, &tStart, &tEnd);
                tStart = tStart + 0.5f * P->step_size;
            } else {
                write_photon(photons, photonId, origin, power, th, ph);
                power.x = power.y = power.z = FLT_MAX; /* used by the recomputation detector */
                scatterEvent = 0;
            }
        }
    }
    float th, ph;
    encode_direction(direction, &th, &ph);
    for (uint32_t i = nInteractions; i < maxInteractions; ++i) { /* :199-209 sentinel */
        size_t photonId = (size_t)photonOffset + i * totalPhotons + (size_t)threadId;
        v3 p = { FLT_MAX, FLT_MAX, FLT_MAX };
        v3 pw = { power.x, FLT_MAX, FLT_MAX };
        write_photon(photons, photonId, p, pw, th, ph);
    }
    if (P->flags & CPMO_TRACE_PROGRESSIVE) { /* :211-215 */
        rng[2 * (photonOffset + threadId)] = rx;
        rng[2 * (photonOffset + threadId) + 1] = rc;
    }
}

/* statistics for tools/ (divergence models): when set, cpmo_trace stores every photon's Woodcock iteration count */
static uint32_t* g_step_array = 0;
void cpmo_debug_set_step_array(uint32_t* per_photon_steps) { g_step_array = per_photon_steps; }

CPMO_CLONES
void cpmo_trace(const cpmo_volume* vol, const float* tf_rgba, int tf_width,
                const float* tf_scattering_rgba, const float aabb[8],
                const cpmo_trace_params* P, const float* ls, const float* isect,
                const uint32_t* recompute_indices, int n_recompute, uint32_t* rng,
                float* photons, uint64_t* steps_out) {
    uint64_t total_steps = 0;
    int n = recompute_indices ? n_recompute : P->n_light_samples;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1024) reduction(+ : total_steps)
    for (int gid = 0; gid < n; ++gid) {
        int threadId = gid;
        if (recompute_indices) { /* photontracer.cl:97-106 */
            threadId = (int)recompute_indices[gid] - P->photon_offset;
            if (threadId < 0 || threadId >= P->n_light_samples) continue;
        }
        trace_env e = { vol, tf_rgba, tf_scattering_rgba ? tf_scattering_rgba : tf_rgba, tf_width, 0 };
        trace_one(&e, aabb, P, ls, isect, threadId, rng, photons);
        total_steps += e.steps;
        if (g_step_array) g_step_array[threadId] = (uint32_t)e.steps;
    }
    if (steps_out) *steps_out = total_steps;
}

/* ------------------------------------------------------------------ splat (reference gather) */

/* processor/photontolightvolumeprocessorcl.cpp:388-390 with photondata.cpp:38,79-81 */
float cpmo_relative_irradiance_scale(double radius, double n_photons) {
    const double pi = 3.14159265358979323846;
    double vol = radius * radius * radius * (pi * 4. / 3.);
    return (float)((1. / pi) / (vol * n_photons));
}

/* progressivephotonmapping/cl/photonstolightvolume.cl:31-79 for one photon whose power is
 * already scaled; add = 1: accumulate into grid, returns nothing. */
static inline void splat_photon(float* out, const cpmo_grid_desc* g, const float* ph, v3 pw,
                                float radius) {
    v3 p = { ph[0], ph[1], ph[2] };
    if (p.x == FLT_MAX || p.y == FLT_MAX || p.z == FLT_MAX) return;
    v3 lo = { p.x - radius, p.y - radius, p.z - radius };
    v3 hi = { p.x + radius, p.y + radius, p.z + radius };
    v3 a = transform_point(g->texture_to_index, lo);
    v3 b = transform_point(g->texture_to_index, hi);
    int sx = (int)a.x, sy = (int)a.y, sz = (int)a.z; /* convert_int3: truncate */
    if (sx < 0) sx = 0; if (sy < 0) sy = 0; if (sz < 0) sz = 0;
    int ex = (int)(b.x + 1.f), ey = (int)(b.y + 1.f), ez = (int)(b.z + 1.f);
    if (ex > g->dims[0]) ex = g->dims[0];
    if (ey > g->dims[1]) ey = g->dims[1];
    if (ez > g->dims[2]) ez = g->dims[2];
    for (int z = sz; z < ez; ++z)
        for (int y = sy; y < ey; ++y)
            for (int x = sx; x < ex; ++x) {
                size_t voxelIndex = (size_t)x + (size_t)y * g->dims[0] + (size_t)z * g->dims[0] * g->dims[1];
                v3 vi = { (float)x, (float)y, (float)z };
                v3 c = transform_point(g->index_to_texture, vi);
                float dx = c.x - p.x, dy = c.y - p.y, dz = c.z - p.z;
                float dist = sqrtf(om_fma(dz, dz, om_fma(dy, dy, dx * dx)));
                float weight = density_kernel(dist / radius);
                if (g->channels == 1) {
                    float v = pw.x * weight;
                    if (v != 0.f) out[voxelIndex] += v;
                } else {
                    float vx = pw.x * weight, vy = pw.y * weight, vz = pw.z * weight;
                    if (vx != 0.f) out[voxelIndex * 4] += vx;
                    if (vy != 0.f) out[voxelIndex * 4 + 1] += vy;
                    if (vz != 0.f) out[voxelIndex * 4 + 2] += vz;
                }
            }
}

/* cl/photonstolightvolume.cl:139-166: power *= isotropicPhaseFunction() * scale, then splat.
 * [INVIWO] isotropicPhaseFunction() = 1/(4 pi). */
void cpmo_splat(const float* photons, int total_photons, const cpmo_grid_desc* g, float radius,
                float scale, float* out) {
    float k = 0.0795774715459476679f * scale;
    for (int i = 0; i < total_photons; ++i) {
        const float* ph = photons + 8 * (size_t)i;
        v3 pw = { ph[3] * k, ph[4] * k, ph[5] * k };
        splat_photon(out, g, ph, pw, radius);
    }
}

/* cl/photonstolightvolume.cl:168-202 */
void cpmo_splat_selected(const float* photons, const uint32_t* indices, int n_indices,
                         const cpmo_grid_desc* g, float radius, float scale, float multiplier,
                         int n_photons, int n_interactions, float* out) {
    float k = 0.0795774715459476679f * scale;
    for (int j = 0; j < n_indices; ++j) {
        size_t id = indices[j];
        for (int it = 0; it < n_interactions; ++it) {
            const float* ph = photons + 8 * ((size_t)it * n_photons + id);
            v3 pw = { ph[3] * k, ph[4] * k, ph[5] * k };
            pw.x *= multiplier; pw.y *= multiplier; pw.z *= multiplier;
            splat_photon(out, g, ph, pw, radius);
        }
    }
}

/* cl/photonstolightvolume.cl:225-248 */
void cpmo_copy_indexed_photons(const float* photons, const uint32_t* indices, int n_indices,
                               float multiplier, int n_photons, int n_interactions, float* aligned,
                               int out_offset) {
    for (int j = 0; j < n_indices; ++j) {
        size_t id = indices[j];
        for (int it = 0; it < n_interactions; ++it) {
            const float* ph = photons + 8 * ((size_t)it * n_photons + id);
            float* q = aligned + 8 * ((size_t)out_offset + j + (size_t)it * n_indices);
            q[0] = ph[0]; q[1] = ph[1]; q[2] = ph[2];
            q[3] = ph[3] * multiplier; q[4] = ph[4] * multiplier; q[5] = ph[5] * multiplier;
            q[6] = ph[6]; q[7] = ph[7];
        }
    }
}

/* ------------------------------------------------------------------ sort / bin / gather */

/* Team size of the short parallel regions of the sort / bin: their barriers do not scale to hundreds of threads
 * (measured on a 2 x 64-core host: 9 ms at 64 threads, 800 ms at 256 for one bin of 1 M photons). */
static int bin_threads(void) { return g_threads < 1 ? 1 : (g_threads > 64 ? 64 : g_threads); }

/* Semantics of clogs::Radixsort::enqueue (radixsortcl/ext/clogs/src/radixsort.cpp:169-259):
 * stable, ascending on the low key_bits bits, result in place. */
void cpmo_sort_pairs(uint32_t* keys, uint32_t* values, size_t n, int key_bits) {
    if (key_bits <= 0 || key_bits > 32) key_bits = 32;
    uint32_t* k2 = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* v2 = values ? (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t)) : NULL;
    /* LSD counting sort on 16-bit digits.  With OpenMP threads the input is cut into contiguous chunks, one per
     * thread: per-(digit, chunk) counts, offsets taken in (digit, chunk) order, every chunk scattered in its own
     * order -- the same stable result as the one-thread loop, whatever the thread count. */
    int T = g_threads < 1 ? 1 : (g_threads > 16 ? 16 : g_threads);
    if (n < 65536) T = 1;
    size_t* cnt = (size_t*)malloc((size_t)T * 65536 * sizeof(size_t));
    uint32_t *ks = keys, *kd = k2, *vs = values, *vd = v2;
    for (int shift = 0; shift < key_bits; shift += 16) {
        int bits = key_bits - shift < 16 ? key_bits - shift : 16;
        uint32_t mask = (1u << bits) - 1u;
        const size_t chunk = (n + (size_t)T - 1) / (size_t)T;
#pragma omp parallel for num_threads(bin_threads()) schedule(static, 1)
        for (int t = 0; t < T; ++t) {
            size_t* c = cnt + (size_t)t * 65536;
            memset(c, 0, 65536 * sizeof(size_t));
            size_t lo = (size_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
            for (size_t i = lo; i < hi; ++i) c[(ks[i] >> shift) & mask]++;
        }
        size_t run = 0;
        for (uint32_t d = 0; d <= mask; ++d)
            for (int t = 0; t < T; ++t) { size_t c = cnt[(size_t)t * 65536 + d]; cnt[(size_t)t * 65536 + d] = run; run += c; }
#pragma omp parallel for num_threads(bin_threads()) schedule(static, 1)
        for (int t = 0; t < T; ++t) {
            size_t* c = cnt + (size_t)t * 65536;
            size_t lo = (size_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
            for (size_t i = lo; i < hi; ++i) {
                size_t d = c[(ks[i] >> shift) & mask]++;
                kd[d] = ks[i];
                if (vs) vd[d] = vs[i];
            }
        }
        uint32_t* tp = ks; ks = kd; kd = tp;
        tp = vs; vs = vd; vd = tp;
    }
    if (ks != keys) {
        memcpy(keys, ks, n * sizeof(uint32_t));
        if (values) memcpy(values, vs, n * sizeof(uint32_t));
    }
    free(k2); free(v2); free(cnt);
}
void cpmo_sort_keys(uint32_t* keys, size_t n, int key_bits) { cpmo_sort_pairs(keys, NULL, n, key_bits); }

static inline int key_bits_for(uint32_t cells) {
    int b = 1;
    while (b < 32 && (1u << b) < cells) ++b;
    return b;
}

/* Cell key of a photon: template progressivephotonmapping/cl/hashlightsample.cl:55-64;
 * contract SURVEY S6 (no reference counterpart). */
static inline uint32_t cell_key(const float* ph, const cpmo_grid_desc* g) {
    if (ph[0] == FLT_MAX || ph[1] == FLT_MAX || ph[2] == FLT_MAX) return 0xffffffffu;
    int c[3];
    for (int a = 0; a < 3; ++a) {
        float f = floorf(ph[a] * (float)g->dims[a]);
        f = om_min(om_max(f, 0.0f), (float)(g->dims[a] - 1));
        c[a] = (int)f;
    }
    return (uint32_t)c[0] + (uint32_t)g->dims[0] * ((uint32_t)c[1] + (uint32_t)g->dims[1] * (uint32_t)c[2]);
}

void cpmo_bin(const float* photons, int n, const cpmo_grid_desc* g, uint32_t* order,
              uint32_t* cell_start, float* sorted) {
    uint32_t cells = (uint32_t)g->dims[0] * g->dims[1] * g->dims[2];
    uint32_t* keys = (uint32_t*)malloc(((size_t)n ? (size_t)n : 1) * sizeof(uint32_t));
#pragma omp parallel for num_threads(bin_threads()) schedule(static)
    for (int i = 0; i < n; ++i) { keys[i] = cell_key(photons + 8 * (size_t)i, g); order[i] = (uint32_t)i; }
    /* all 32 bits: the sentinel key 0xffffffff must sort last */
    cpmo_sort_pairs(keys, order, (size_t)n, 32);
    (void)key_bits_for;
    /* cell_start[c] = first sorted position with key >= c: one lower bound per entry */
#pragma omp parallel for num_threads(bin_threads()) schedule(static)
    for (long long c = 0; c <= (long long)cells; ++c) {
        size_t lo = 0, hi = (size_t)n;
        while (lo < hi) {
            size_t mid = (lo + hi) >> 1;
            if (keys[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
        }
        cell_start[c] = (uint32_t)lo;
    }
    int stride = g->channels == 1 ? 4 : 8;
#pragma omp parallel for num_threads(bin_threads()) schedule(static)
    for (int i = 0; i < n; ++i) {
        const float* ph = photons + 8 * (size_t)order[i];
        float* q = sorted + (size_t)stride * i;
        q[0] = ph[0]; q[1] = ph[1]; q[2] = ph[2]; q[3] = ph[3];
        if (stride == 8) { q[4] = ph[4]; q[5] = ph[5]; q[6] = 0.f; q[7] = 0.f; }
    }
    free(keys);
}

/* Per-voxel restatement of cl/photonstolightvolume.cl:42-75: the terms photon p adds to
 * voxel (x, y, z), summed in (dz, dy, cell, sorted index) order. */
CPMO_CLONES
void cpmo_gather(const float* sorted, const uint32_t* cell_start, int n, const cpmo_grid_desc* g,
                 float radius, float scale, int accumulate, float* out) {
    (void)n;
    const int dx_ = g->dims[0], dy_ = g->dims[1], dz_ = g->dims[2];
    const float k = 0.0795774715459476679f * scale;
    const int Rx = (int)floorf(om_fma(radius, (float)dx_, 0.501f));
    const int Ry = (int)floorf(om_fma(radius, (float)dy_, 0.501f));
    const int Rz = (int)floorf(om_fma(radius, (float)dz_, 0.501f));
    const int stride = g->channels == 1 ? 4 : 8;
    /* (z, y) rows are the tasks: photons pile up on a few faces, whole z-slabs would leave most threads idle */
#pragma omp parallel for collapse(2) num_threads(g_threads) schedule(dynamic, 4)
    for (int z = 0; z < dz_; ++z)
        for (int y = 0; y < dy_; ++y)
            for (int x = 0; x < dx_; ++x) {
                size_t voxelIndex = (size_t)x + (size_t)y * dx_ + (size_t)z * dx_ * dy_;
                v3 vi = { (float)x, (float)y, (float)z };
                v3 c = transform_point(g->index_to_texture, vi);
                float sr = 0.f, sg = 0.f, sb = 0.f;
                int xlo = x - Rx < 0 ? 0 : x - Rx;
                int xhi = x + Rx > dx_ - 1 ? dx_ - 1 : x + Rx;
                for (int cz = z - Rz; cz <= z + Rz; ++cz) {
                    if (cz < 0 || cz >= dz_) continue;
                    for (int cy = y - Ry; cy <= y + Ry; ++cy) {
                        if (cy < 0 || cy >= dy_) continue;
                        size_t row = (size_t)dx_ * ((size_t)cy + (size_t)dy_ * (size_t)cz);
                        uint32_t jb = cell_start[row + xlo], je = cell_start[row + xhi + 1];
                        for (uint32_t j = jb; j < je; ++j) {
                            const float* ph = sorted + (size_t)stride * j;
                            v3 p = { ph[0], ph[1], ph[2] };
                            v3 lo = { p.x - radius, p.y - radius, p.z - radius };
                            v3 hi = { p.x + radius, p.y + radius, p.z + radius };
                            v3 a = transform_point(g->texture_to_index, lo);
                            v3 b = transform_point(g->texture_to_index, hi);
                            int sx = (int)a.x, sy = (int)a.y, sz = (int)a.z;
                            if (sx < 0) sx = 0; if (sy < 0) sy = 0; if (sz < 0) sz = 0;
                            int ex = (int)(b.x + 1.f), ey = (int)(b.y + 1.f), ez = (int)(b.z + 1.f);
                            if (ex > dx_) ex = dx_; if (ey > dy_) ey = dy_; if (ez > dz_) ez = dz_;
                            if (x < sx || x >= ex || y < sy || y >= ey || z < sz || z >= ez) continue;
                            float ddx = c.x - p.x, ddy = c.y - p.y, ddz = c.z - p.z;
                            float dist = sqrtf(om_fma(ddz, ddz, om_fma(ddy, ddy, ddx * ddx)));
                            float weight = density_kernel(dist / radius);
                            float vr = (ph[3] * k) * weight;
                            if (vr != 0.f) sr += vr;
                            if (stride == 8) {
                                float vg = (ph[4] * k) * weight, vb = (ph[5] * k) * weight;
                                if (vg != 0.f) sg += vg;
                                if (vb != 0.f) sb += vb;
                            }
                        }
                    }
                }
                if (g->channels == 1) {
                    out[voxelIndex] = accumulate ? out[voxelIndex] + sr : sr;
                } else {
                    float* o = out + 4 * voxelIndex;
                    if (accumulate) { o[0] += sr; o[1] += sg; o[2] += sb; }
                    else { o[0] = sr; o[1] = sg; o[2] = sb; o[3] = 0.f; }
                }
            }
}

/* ---- tolerance-mode formulation (cpm_bin_fast + cpm_gather_fast): NO reference counterpart -- the reference adds
 * the same terms with CAS float atomics in arrival order (cl/photonstolightvolume.cl:15-29,62-75).  Restated here
 * so that the HIP path can be checked bit for bit: the weight is 0.75 * (1 - d^2 / r^2) for d^2 <= r^2
 * (cl/densityestimationkernel.cl:43-60 with x^2 = d^2 / r^2), every contribution is truncated to 64-bit fixed
 * point with the scale below, the per-voxel sums are exact integers (order-free), one rounding to float.
 * The candidate voxels of a photon are the integers within r * textureToIndex + 1e-3 of its index-space coordinate per
 * axis, clipped to the grid -- nothing of the kernels' brick bookkeeping enters the result. */

static float fast_fixed_scale(float maxpow, float k) {
    float m = maxpow * fabsf(k) * 0.75f;
    if (!(m > 0.f) || m > FLT_MAX) return 1.0f;
    union { float f; uint32_t u; } b; b.f = m;
    int e = (int)((b.u >> 23) & 0xffu) - 126; /* m < 2^e */
    int sh = 30 - e;                           /* every contribution fits int32; < 2^31 of them fit int64 */
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    b.u = (uint32_t)(sh + 127) << 23;
    return b.f;
}

void cpmo_gather_fast(const float* photons, int n, const cpmo_grid_desc* g, float radius, float scale,
                      int accumulate, float* out) {
    const int dims[3] = { g->dims[0], g->dims[1], g->dims[2] };
    const size_t cells = (size_t)dims[0] * dims[1] * dims[2];
    const int ch3 = g->channels == 1 ? 1 : 3;
    const float k = 0.0795774715459476679f * scale;
    float maxpow = 0.f;
    for (int i = 0; i < n; ++i) {
        const float* ph = photons + 8 * (size_t)i;
        if (ph[0] == FLT_MAX || ph[1] == FLT_MAX || ph[2] == FLT_MAX) continue;
        for (int c = 0; c < ch3; ++c) { float a = fabsf(ph[3 + c]); if (a <= FLT_MAX && a > maxpow) maxpow = a; }
    }
    const float S = fast_fixed_scale(maxpow, k);
    const float invS = 1.0f / S;
    long long* acc = (long long*)calloc(cells * (size_t)ch3, sizeof(long long));
    const float* T = g->texture_to_index; const float* I = g->index_to_texture;
    const float sT[3] = { T[0], T[5], T[10] }, tT[3] = { T[12], T[13], T[14] };
    const float sI[3] = { I[0], I[5], I[10] }, tI[3] = { I[12], I[13], I[14] };
    const float r2 = radius * radius, inv_r2 = 1.0f / r2;
    /* candidates per axis: floor(2 r') + 1 with r' = r * textureToIndex + 1e-3 along the widest axis (what the record capacity
     * and the device loops are sized from); a box is cut to that many (the extra integer rounding can admit lies at r' > r) */
    const float rmax = fmaxf(radius * sT[0], fmaxf(radius * sT[1], radius * sT[2])) + 1e-3f;
    const int maxc = (int)floorf(2.f * rmax) + 1;
    for (int i = 0; i < n; ++i) {
        const float* ph = photons + 8 * (size_t)i;
        if (ph[0] == FLT_MAX || ph[1] == FLT_MAX || ph[2] == FLT_MAX) continue;
        int s[3], e[3];
        for (int a = 0; a < 3; ++a) {
            const float u = om_fma(sT[a], ph[a], tT[a]);
            const float rg = radius * sT[a] + 1e-3f;
            /* the candidate voxels: the integers within rg of u, clipped to the grid (clamped as floats: a far-away photon
             * must not reach an out-of-range float -> int conversion) */
            int lo = (int)om_min(om_max(ceilf(u - rg), 0.0f), (float)dims[a]);
            int hi = (int)om_max(om_min(floorf(u + rg), (float)(dims[a] - 1)), -1.0f);
            if (hi > lo + maxc - 1) hi = lo + maxc - 1;
            s[a] = lo; e[a] = hi;
        }
        /* a non-finite power has no fixed-point image: that channel of that photon contributes nothing */
        const float pk[3] = { fabsf(ph[3]) <= FLT_MAX ? ph[3] * k : 0.f, fabsf(ph[4]) <= FLT_MAX ? ph[4] * k : 0.f,
                              fabsf(ph[5]) <= FLT_MAX ? ph[5] * k : 0.f };
        for (int z = s[2]; z <= e[2]; ++z) {
            const float dz = om_fma(sI[2], (float)z, tI[2]) - ph[2];
            for (int y = s[1]; y <= e[1]; ++y) {
                const float dy = om_fma(sI[1], (float)y, tI[1]) - ph[1];
                for (int x = s[0]; x <= e[0]; ++x) {
                    const float dx = om_fma(sI[0], (float)x, tI[0]) - ph[0];
                    const float d2 = om_fma(dz, dz, om_fma(dy, dy, dx * dx));
                    if (!(d2 <= r2)) continue;
                    const float w = 0.75f * (1.0f - d2 * inv_r2);
                    const size_t v = (size_t)x + (size_t)dims[0] * ((size_t)y + (size_t)dims[1] * (size_t)z);
                    for (int c = 0; c < ch3; ++c) acc[(size_t)c * cells + v] += (long long)(int)((pk[c] * w) * S);
                }
            }
        }
    }
    for (size_t v = 0; v < cells; ++v) {
        if (ch3 == 1) {
            const float f = (float)acc[v] * invS;
            out[v] = accumulate ? out[v] + f : f;
        } else {
            float* o = out + 4 * v;
            const float fr = (float)acc[v] * invS, fg = (float)acc[cells + v] * invS, fb = (float)acc[2 * cells + v] * invS;
            if (accumulate) { o[0] += fr; o[1] += fg; o[2] += fb; }
            else { o[0] = fr; o[1] = fg; o[2] = fb; o[3] = 0.f; }
        }
    }
    free(acc);
}

/* ------------------------------------------------------------------ correlated re-trace */

static inline float normalized_voxel(const cpmo_volume* v, int x, int y, int z) {
    float s = fetch_voxel_raw(v, x, y, z) * norm_scale(v->dtype);
    return (s + v->format_offset) * (1.0f - v->format_scaling);
}

/* uniformgridcl/cl/uniformgrid/volumeminmax.cl:33-61; [INVIWO] writeImageVec2UInt16f =
 * round-to-nearest of clamp(v, 0, 1) * 65535 */
void cpmo_volume_minmax(const cpmo_volume* v, int region, uint16_t* out) {
    int ox = (v->dims[0] + region - 1) / region, oy = (v->dims[1] + region - 1) / region,
        oz = (v->dims[2] + region - 1) / region;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int gz = 0; gz < oz; ++gz)
        for (int gy = 0; gy < oy; ++gy)
            for (int gx = 0; gx < ox; ++gx) {
                float mn = FLT_MAX, mx = 0.f;
                int ex = gx * region + region, ey = gy * region + region, ez = gz * region + region;
                if (ex > v->dims[0]) ex = v->dims[0];
                if (ey > v->dims[1]) ey = v->dims[1];
                if (ez > v->dims[2]) ez = v->dims[2];
                for (int z = gz * region; z < ez; ++z)
                    for (int y = gy * region; y < ey; ++y)
                        for (int x = gx * region; x < ex; ++x) {
                            float val = normalized_voxel(v, x, y, z);
                            mn = om_min(mn, val);
                            mx = om_max(mx, val);
                        }
                size_t o = (size_t)gx + (size_t)ox * ((size_t)gy + (size_t)oy * gz);
                out[2 * o] = (uint16_t)rintf(om_min(om_max(mn, 0.f), 1.f) * 65535.f);
                out[2 * o + 1] = (uint16_t)rintf(om_min(om_max(mx, 0.f), 1.f) * 65535.f);
            }
}

/* uniformgridcl/processors/dynamicvolumedifferenceanalysis.h:96-151 with the default data
 * range of the format (dataScaling 1, dataRange = [0, max]): mean |b - a| / range per brick,
 * divisor region^3 even for clipped border bricks (as the reference). */
void cpmo_volume_difference(const cpmo_volume* a, const cpmo_volume* b, int region, float* out) {
    int ox = (a->dims[0] + region - 1) / region, oy = (a->dims[1] + region - 1) / region,
        oz = (a->dims[2] + region - 1) / region;
    double range = a->dtype == CPMO_U8 ? 255.0 : (a->dtype == CPMO_U16 ? 65535.0 : 1.0);
    double cnt = (double)region * region * region;
    for (int gz = 0; gz < oz; ++gz)
        for (int gy = 0; gy < oy; ++gy)
            for (int gx = 0; gx < ox; ++gx) {
                double sum = 0;
                int ex = gx * region + region, ey = gy * region + region, ez = gz * region + region;
                if (ex > a->dims[0]) ex = a->dims[0];
                if (ey > a->dims[1]) ey = a->dims[1];
                if (ez > a->dims[2]) ez = a->dims[2];
                for (int z = gz * region; z < ez; ++z)
                    for (int y = gy * region; y < ey; ++y)
                        for (int x = gx * region; x < ex; ++x)
                            sum += fabs((double)fetch_voxel_raw(b, x, y, z) - (double)fetch_voxel_raw(a, x, y, z));
                out[(size_t)gx + (size_t)ox * ((size_t)gy + (size_t)oy * gz)] = (float)((sum / cnt) / range);
            }
}

typedef struct { float x, y, z, w; } v4;
static inline v4 mix4(v4 a, v4 b, float t) { /* OpenCL mix: x + (y - x) * a */
    v4 r = { a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t };
    return r;
}
static inline v4 min4(v4 a, v4 b) { v4 r = { om_min(a.x, b.x), om_min(a.y, b.y), om_min(a.z, b.z), om_min(a.w, b.w) }; return r; }
static inline v4 max4(v4 a, v4 b) { v4 r = { om_max(a.x, b.x), om_max(a.y, b.y), om_max(a.z, b.z), om_max(a.w, b.w) }; return r; }
static inline v4 ld4(const float* c, int i) { v4 r = { c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3] }; return r; }

/* importancesamplingcl/cl/minmaxuniformgrid3dimportance.cl:163-169 (-D INCREMENTAL_TF_IMPORTANCE) */
static inline float tf_points_importance(v4 color, v4 nextColor) {
    (void)color;
    return nextColor.x + nextColor.y + nextColor.z + nextColor.w;
}
/* importancesamplingcl/cl/minmaxuniformgrid3dimportance.cl:186-227 */
static float importance_for_range_tf(float rx, float ry, const float* positions, const float* colors, int nPoints) {
    int i = 0;
    while (i < nPoints - 1 && rx > positions[i + 1]) ++i;
    v4 color = mix4(ld4(colors, i), ld4(colors, i + 1), (rx - positions[i]) / (positions[i + 1] - positions[i]));
    v4 minColor = color, maxColor = color;
    if (ry <= positions[i + 1]) {
        v4 nextColor = mix4(ld4(colors, i), ld4(colors, i + 1), (ry - positions[i]) / (positions[i + 1] - positions[i]));
        minColor = min4(minColor, nextColor);
        maxColor = max4(maxColor, nextColor);
        return tf_points_importance(minColor, maxColor);
    } else {
        v4 nextColor = ld4(colors, i + 1);
        minColor = min4(minColor, nextColor);
        maxColor = max4(maxColor, nextColor);
        ++i;
    }
    while (i < nPoints - 1 && ry > positions[i + 1]) {
        v4 nextColor = ld4(colors, i + 1);
        minColor = min4(minColor, nextColor);
        maxColor = max4(maxColor, nextColor);
        ++i;
    }
    if (i < nPoints - 1) {
        color = mix4(ld4(colors, i), ld4(colors, i + 1), (ry - positions[i]) / (positions[i + 1] - positions[i]));
        minColor = min4(minColor, color);
        maxColor = max4(maxColor, color);
    }
    return tf_points_importance(minColor, maxColor);
}

/* kernels classifyMinMaxUniformGrid3DImportanceKernel (:269-289) and
 * classifyTimeVaryingMinMaxUniformGrid3DImportanceKernel (:291-330).
 * Literal quirk kept: tfPointsImportance(min, max) sums the MAX colour's channels. */
void cpmo_importance_tf(const uint16_t* mm, const uint16_t* prev, const float* diff, int n_cells,
                        const float* positions, const float* colors, int n_points, float* out) {
    for (int i = 0; i < n_cells; ++i) {
        uint16_t lo = mm[2 * i], hi = mm[2 * i + 1];
        if (prev) {
            if (prev[2 * i] < lo) lo = prev[2 * i];
            if (prev[2 * i + 1] > hi) hi = prev[2 * i + 1];
        }
        float rx = (1.f / 65535.f) * (float)lo, ry = (1.f / 65535.f) * (float)hi;
        float imp = importance_for_range_tf(rx, ry, positions, colors, n_points);
        out[i] = prev ? diff[i] * imp : imp;
    }
}

/* uniformgridcl/cl/uniformgrid/uniformgrid.cl:38-69 + :147-167 (OPTIMIZE_STEP_FOR_SIMD) driven by
 * progressivephotonmapping/cl/photonrecomputationdetector.cl:55-90.
 * Added: float clamps before int conversion and an iteration cap of nx+ny+nz+4 (a GPU
 * kernel must terminate on NaN input); neither changes a finite-input result. */
static float uniform_grid_importance(const float x1[3], const float x2[3], const float cellDim[3],
                                     const float* grid, const int32_t dims[3]) {
    int cell[3], cellEnd[3], di[3];
    float dt[3], deltatx[3];
    for (int a = 0; a < 3; ++a) {
        float maxc = (float)(dims[a] - 1);
        float cf = om_min(om_max(floorf(x1[a] / cellDim[a]), 0.f), maxc);
        cell[a] = (int)cf;
        float ef = x2[a] / cellDim[a];
        ef = om_min(om_max(ef, -1.f), (float)dims[a]); /* then truncate + clamp */
        int ei = (int)ef;
        cellEnd[a] = ei < 0 ? 0 : (ei > dims[a] - 1 ? dims[a] - 1 : ei);
        di[a] = (x1[a] < x2[a]) ? 1 : ((x1[a] > x2[a]) ? -1 : 0);
        float invAbsDir = 1.f / fabsf(x2[a] - x1[a]);
        float minx = cellDim[a] * cf;
        float maxx = minx + cellDim[a];
        dt[a] = ((x1[a] > x2[a]) ? (x1[a] - minx) : (maxx - x1[a])) * invAbsDir;
        deltatx[a] = cellDim[a] * invAbsDir;
    }
    float importance = 0.f, dt1 = 0.f;
    int cont = 1;
    int cap = dims[0] + dims[1] + dims[2] + 4;
    while (cont && cap-- > 0) {
        float val = grid[(size_t)cell[0] + (size_t)cell[1] * dims[0] + (size_t)cell[2] * dims[0] * dims[1]];
        float dt0 = dt1;
        int ax = (dt[0] <= dt[1] && dt[0] <= dt[2]) ? 0 : ((dt[0] > dt[1] && dt[1] <= dt[2]) ? 1 : 2);
        dt1 = dt[ax];
        if (cell[ax] == cellEnd[ax]) cont = 0;
        else { dt[ax] += deltatx[ax]; cell[ax] += di[ax]; }
        importance += val * (om_min(1.f, dt1) - dt0);
    }
    float lx = x2[0] - x1[0], ly = x2[1] - x1[1], lz = x2[2] - x1[2];
    float len = sqrtf(om_fma(lz, lz, om_fma(ly, ly, lx * lx)));
    return importance * len;
}

/* convert_uint_sat_rtp(100 * imp), then min(., 0x7fffffff) (Q9) */
static inline uint32_t importance_to_uint(float imp100) {
    if (!(imp100 > 0.f)) return 0u; /* also NaN */
    float c = ceilf(imp100);
    if (c >= 2147483648.f) return 2147483647u;
    uint32_t u = (uint32_t)c;
    return u > 2147483647u ? 2147483647u : u;
}

/* progressivephotonmapping/cl/photonrecomputationdetector.cl:92-157 */
void cpmo_photon_importance(const float* grid, const int32_t dims[3], const float cell_size[3],
                            const float t2i[16], const float* photons, int photon_offset,
                            const float* ls, const float* isect, int n_light_samples,
                            int max_interactions, int total_photons, int fix_exit_point,
                            uint32_t* importances) {
    const float bmin[3] = { 0.f, 0.f, 0.f }, bmax[3] = { 1.f, 1.f, 1.f };
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1024)
    for (int threadId = 0; threadId < n_light_samples; ++threadId) {
        float recomputationImportance = 0.f;
        const float* s = ls + 8 * (size_t)threadId;
        v3 origin = { s[0], s[1], s[2] };
        v3 direction = decode_direction(s[6], s[7]);
        float tStart = isect[2 * threadId], tEnd = isect[2 * threadId + 1];
        if (tStart < tEnd) {
            v3 entry = { om_fma(tStart, direction.x, origin.x), om_fma(tStart, direction.y, origin.y),
                         om_fma(tStart, direction.z, origin.z) };
            for (int interaction = 0; interaction < max_interactions; ++interaction) {
                size_t photonId = (size_t)photon_offset + (size_t)interaction * total_photons + threadId;
                const float* ph = photons + 8 * photonId;
                v3 exitp = { ph[0], ph[1], ph[2] };
                if (ph[0] == FLT_MAX || ph[1] == FLT_MAX || ph[2] == FLT_MAX) {
                    if (interaction == 0) {
                        if (fix_exit_point) {
                            exitp.x = om_fma(tEnd, direction.x, origin.x);
                            exitp.y = om_fma(tEnd, direction.y, origin.y);
                            exitp.z = om_fma(tEnd, direction.z, origin.z);
                        } else { /* Q8: origin omitted */
                            exitp.x = tEnd * direction.x; exitp.y = tEnd * direction.y; exitp.z = tEnd * direction.z;
                        }
                    } else if (entry.x == FLT_MAX || entry.y == FLT_MAX || entry.z == FLT_MAX) {
                        break;
                    } else {
                        float t0 = 0.f, t1 = FLT_MAX;
                        v3 pd = decode_direction(ph[6], ph[7]);
                        if (ph[3] != FLT_MAX && ray_box(bmin, bmax, entry, pd, &t0, &t1)) {
                            /* reference adds to the FLT_MAX sentinel (:138); restated from entry */
                            exitp.x = om_fma(t1, pd.x, entry.x);
                            exitp.y = om_fma(t1, pd.y, entry.y);
                            exitp.z = om_fma(t1, pd.z, entry.z);
                        } else {
                            break;
                        }
                    }
                }
                v3 a = transform_point(t2i, entry), b = transform_point(t2i, exitp);
                float x1[3] = { a.x + 0.5f, a.y + 0.5f, a.z + 0.5f };
                float x2[3] = { b.x + 0.5f, b.y + 0.5f, b.z + 0.5f };
                recomputationImportance += uniform_grid_importance(x1, x2, cell_size, grid, dims);
                entry.x = ph[0]; entry.y = ph[1]; entry.z = ph[2];
            }
        }
        importances[photon_offset + threadId] -= importance_to_uint(100.f * recomputationImportance);
    }
}

/* progressivephotonmapping/cl/photonrecomputationdetector.cl:160-194 */
void cpmo_photon_importance_equal(int photon_offset, int n_light_samples, int percentage,
                                  int iteration, uint32_t* importances) {
    for (int threadId = 0; threadId < n_light_samples; ++threadId) {
        float imp = 0.f;
        int photonId = photon_offset + threadId;
        if ((photonId + iteration) % (100 / percentage) == 0) imp = 1.f;
        importances[photon_offset + threadId] -= importance_to_uint(100.f * imp);
    }
}

/* cl/threshold.cl:33-40 + clogs reduce (count) + cl/indextobuffer.cl:33-40 +
 * sortIndicesByImportance (processor/progressivephotontracercl.cpp:325-363,689-706) */
void cpmo_select_recompute(uint32_t* importances, size_t n, uint32_t* indices_out, int32_t* n_changed) {
    int32_t cnt = 0;
    for (size_t i = 0; i < n; ++i) { cnt += (int32_t)(importances[i] < 2147483647u); indices_out[i] = (uint32_t)i; }
    cpmo_sort_pairs(importances, indices_out, n, 32);
    *n_changed = cnt;
}

/* the selection without the ranking: changed photons first, both parts in ascending index order
 * (cl/threshold.cl:33-40 + count + cl/indextobuffer.cl:33-40, followed by a stable partition) */
void cpmo_select_changed(const uint32_t* importances, size_t n, uint32_t* indices_out, int32_t* n_changed) {
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) if (importances[i] < 2147483647u) indices_out[k++] = (uint32_t)i;
    *n_changed = (int32_t)k;
    for (size_t i = 0; i < n; ++i) if (!(importances[i] < 2147483647u)) indices_out[k++] = (uint32_t)i;
}

/* ---------------------------------------------------------------------------------------------
 * temporal interpolation */

/* uniformgridcl/cl/buffermixer.cl:37-48 with MIX_T float: out = mix(x, y, a) = x + (y - x) * a */
void cpmo_mix_f32(const float* x, const float* y, float a, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = x[i] + (y[i] - x[i]) * a;
}

/* the same kernel as BufferMixerCL::compileKernel builds it for Vec2UINT16 (buffermixercl.cpp:235-239):
 * convert_ushort2(mix(convert_float2(x), convert_float2(y), a)); float -> integer conversion rounds
 * toward zero (OpenCL 1.2 section 6.2.3.3) */
void cpmo_mix_u16x2(const uint16_t* x, const uint16_t* y, float a, size_t n_pairs, uint16_t* out) {
    for (size_t i = 0; i < 2 * n_pairs; ++i) {
        float fx = (float)x[i], fy = (float)y[i];
        float r = fx + (fy - fx) * a;
        out[i] = (uint16_t)(int)r;
    }
}

/* uniformgridcl/glsl/volume_mix.frag:42-52 driven by volumesequenceplayer.cpp:87-140: texture() of a
 * normalised integer format yields v / (2^b - 1); GLSL mix(x, y, a) = x * (1 - a) + y * a; the colour
 * attachment of the same format stores round(clamp(f, 0, 1) * (2^b - 1)) (nearest, ties to even). */
void cpmo_volume_mix(const cpmo_volume* v0, const cpmo_volume* v1, float weight, void* out_voxels) {
    size_t n = (size_t)v0->dims[0] * v0->dims[1] * v0->dims[2];
    float oma = 1.0f - weight;
    for (size_t i = 0; i < n; ++i) {
        if (v0->dtype == CPMO_F32) {
            float x = ((const float*)v0->voxels)[i], y = ((const float*)v1->voxels)[i];
            ((float*)out_voxels)[i] = x * oma + y * weight;
        } else {
            float maxv = v0->dtype == CPMO_U8 ? 255.0f : 65535.0f;
            float x = v0->dtype == CPMO_U8 ? (float)((const uint8_t*)v0->voxels)[i] : (float)((const uint16_t*)v0->voxels)[i];
            float y = v0->dtype == CPMO_U8 ? (float)((const uint8_t*)v1->voxels)[i] : (float)((const uint16_t*)v1->voxels)[i];
            float r = (x / maxv) * oma + (y / maxv) * weight;
            r = om_min(om_max(r, 0.0f), 1.0f);
            int q = (int)rintf(r * maxv);
            if (v0->dtype == CPMO_U8) ((uint8_t*)out_voxels)[i] = (uint8_t)q;
            else ((uint16_t*)out_voxels)[i] = (uint16_t)q;
        }
    }
}
