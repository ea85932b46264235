// cpm_rng_emission.hip -- MWC64X stream seeding (R2), the RNG known-answer harness and the
// photon-emission kernels (E1, E3, E4, E5).  All of them are one-off or tiny elementwise
// passes; they are written for clarity, one work-item per element, coalesced 16/32-byte
// accesses.
#include "cpm_ctx.h"
#include "cpm_emit.hip.h"

using namespace cpm;

namespace {

constexpr uint64_t kM = 18446383549859758079ull;        // MWC64X_M (rndgenmwc64x/cl/random.cl:47)
constexpr uint64_t kD = 360523849793537ull;             // 2^64 mod M
constexpr uint64_t kBaseId = 4077358422479273989ull;    // MWC_BASEID (rndgenmwc64x/cl/skip_mwc.cl:98)

// (a * b) mod M for a, b < M.  The reference does this with up to 64 modular additions
// (rndgenmwc64x/cl/skip_mwc.cl:54-64); here: one 64x64->128 multiply and a fold of the
// high word with 2^64 = D (mod M).  Exact integer arithmetic, so the results are the
// same values.
__device__ __forceinline__ uint64_t mul_mod(uint64_t a, uint64_t b) {
    uint64_t lo = a * b;
    uint64_t hi = __umul64hi(a, b);
    while (hi != 0) {
        uint64_t tl = hi * kD;
        uint64_t th = __umul64hi(hi, kD);
        uint64_t s = lo + tl;
        th += (s < lo) ? 1ull : 0ull;
        lo = s;
        hi = th;
    }
    if (lo >= kM) lo -= kM;
    return lo;
}

__device__ __forceinline__ uint64_t pow_mod(uint64_t a, uint64_t e) {
    uint64_t sqr = a, acc = 1;
    while (e != 0) {
        if (e & 1) acc = mul_mod(acc, sqr);
        sqr = mul_mod(sqr, sqr);
        e >>= 1;
    }
    return acc;
}

// rndgenmwc64x/cl/randstategen.cl:39-47 (+ :52-60 with gap = maxSamplesPerStream)
__global__ void seed_streams_kernel(uint32_t* __restrict__ state, size_t n, uint64_t gap) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2* st = reinterpret_cast<uint2*>(state);
    uint64_t dist = (uint64_t)st[i].x + (uint64_t)i * gap;
    uint64_t m = pow_mod((uint64_t)kMwcA, dist);
    uint64_t x = mul_mod(kBaseId, m);
    st[i] = make_uint2((uint32_t)(x / kMwcA), (uint32_t)(x % kMwcA));
}

// rndgenmwc64x/cl/randomnumbergenerator.cl:34-50
__global__ void random_fill_kernel(uint32_t* __restrict__ state, size_t n, int draws, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2* st = reinterpret_cast<uint2*>(state);
    uint2 s = st[i];
    for (int k = 0; k < draws; ++k) out[i + (size_t)k * n] = rand01_(s.x, s.y);
    st[i] = s;
}

// importancesamplingcl/cl/uniformsamplegenerator2d.cl:35-52
__global__ void uniform_samples_2d_kernel(float dimx, float dimy, int n, float4* __restrict__ samples) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    samples[i] = lattice_sample_(i, dimx, dimy);
}

__device__ __forceinline__ void store_sample(float* ls, int i, f3 o, f3 pw, float th, float ph) {
    float4* q = reinterpret_cast<float4*>(ls) + 2 * (size_t)i;
    q[0] = make_float4(o.x, o.y, o.z, pw.x);
    q[1] = make_float4(pw.y, pw.z, th, ph);
}

// lightcl/cl/directionallightsampler.cl:38-63
__global__ void directional_light_kernel(const float4* __restrict__ samples, int n, Light L, float* __restrict__ ls,
                                         float* __restrict__ dir_hint) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) directional_hint_(L, dir_hint);  // every sample of this light carries the same (theta, phi): what the tracer derives from it, once
    f3 o, pw;
    directional_sample_(L, samples[i], o, pw);
    f3 d = { L.a[0], L.a[1], L.a[2] };
    float th, ph;
    encode_direction_(d, th, ph);
    store_sample(ls, i, o, pw, th, ph);
}

// build-defined point-light emitter
__global__ void point_light_kernel(const float4* __restrict__ samples, int n, Light L, float* __restrict__ ls) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o, pw;
    float th, ph;
    point_sample_(L, samples[i], o, pw, th, ph);
    store_sample(ls, i, o, pw, th, ph);
}

// lightcl/cl/intersection/lightsamplemeshintersection.cl:37-58 for the cube proxy
__global__ void box_intersection_kernel(const float* __restrict__ ls, int n, Box b, float2* __restrict__ isect) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* q = reinterpret_cast<const float4*>(ls) + 2 * (size_t)i;
    float4 a = q[0], c = q[1];
    f3 o = { a.x, a.y, a.z };
    isect[i] = box_entry_exit_(b, o, decode_direction_(c.z, c.w));
}

// same kernel against a triangle list (Moeller-Trumbore per triangle)
__global__ void mesh_intersection_kernel(const float* __restrict__ vtx, const int* __restrict__ idx, int n_indices,
                                         const float* __restrict__ ls, int n, float2* __restrict__ isect) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* q = reinterpret_cast<const float4*>(ls) + 2 * (size_t)i;
    float4 a4 = q[0], c4 = q[1];
    f3 o = { a4.x, a4.y, a4.z };
    f3 d = decode_direction_(c4.z, c4.w);
    float tmin = kFltMax, tmax = -1.f;
    int hits = 0;
    for (int t = 0; t + 2 < n_indices; t += 3) {
        const float* a = vtx + 3 * idx[t];
        const float* b = vtx + 3 * idx[t + 1];
        const float* c = vtx + 3 * idx[t + 2];
        f3 e1 = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
        f3 e2 = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
        f3 p = cross3_(d, e2);
        float det = dot3_(e1, p);
        if (__builtin_fabsf(det) < 1e-12f) continue;
        float inv = 1.0f / det;
        f3 s = { o.x - a[0], o.y - a[1], o.z - a[2] };
        float uu = dot3_(s, p) * inv;
        if (uu < 0.f || uu > 1.f) continue;
        f3 qv = cross3_(s, e1);
        float vv = dot3_(d, qv) * inv;
        if (vv < 0.f || uu + vv > 1.f) continue;
        float tt = dot3_(e2, qv) * inv;
        if (tt < 0.f) continue;
        ++hits;
        tmin = min_(tmin, tt);
        tmax = max_(tmax, tt);
    }
    float t0, t1;
    if (hits == 0) { t0 = 0.f; t1 = -1.f; }
    else if (hits == 1) { t0 = 0.f; t1 = tmax; }
    else { t0 = tmin; t1 = tmax; }
    isect[i] = make_float2(t0, t1);
}

}  // namespace

extern "C" {

int cpm_seed_streams(cpm_ctx* ctx, uint32_t* state, size_t n, uint64_t gap, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, state || n == 0, "cpm_seed_streams: null state");
    if (n == 0) return CPM_OK;
    CPM_LAUNCH(ctx, seed_streams_kernel, dim3(div_up((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, state, n, gap);
    CPM_LAUNCH_CHECK(ctx, "seed_streams_kernel");
    return CPM_OK;
}

int cpm_random_fill(cpm_ctx* ctx, uint32_t* state, size_t n, int draws, float* out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, (state && out) || n == 0, "cpm_random_fill: null argument");
    CPM_REQUIRE(ctx, draws >= 0, "cpm_random_fill: draws < 0");
    if (n == 0) return CPM_OK;
    CPM_LAUNCH(ctx, random_fill_kernel, dim3(div_up((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, state, n, draws, out);
    CPM_LAUNCH_CHECK(ctx, "random_fill_kernel");
    return CPM_OK;
}

int cpm_uniform_samples_2d(cpm_ctx* ctx, int nx, int ny, float* samples4, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, nx >= 0 && ny >= 0 && (long long)nx * ny < (1ll << 24), "cpm_uniform_samples_2d: 0 <= nx*ny < 2^24");
    int n = nx * ny;
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, samples4, "cpm_uniform_samples_2d: null output");
    CPM_LAUNCH(ctx, uniform_samples_2d_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (float)nx, (float)ny, n, reinterpret_cast<float4*>(samples4));
    CPM_LAUNCH_CHECK(ctx, "uniform_samples_2d_kernel");
    return CPM_OK;
}

int cpm_directional_light_samples(cpm_ctx* ctx, const float* samples4, int n, const float radiance[4],
                                  const float direction[4], const float plane_origin[4], const float tangent_u[4],
                                  const float tangent_v[4], float plane_area, float* ls, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n >= 0, "cpm_directional_light_samples: n < 0");
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, samples4 && radiance && direction && plane_origin && tangent_u && tangent_v && ls,
                "cpm_directional_light_samples: null argument");
    Light L;
    for (int a = 0; a < 3; ++a) {
        L.radiance[a] = radiance[a]; L.a[a] = direction[a]; L.origin[a] = plane_origin[a];
        L.u[a] = tangent_u[a]; L.v[a] = tangent_v[a];
    }
    L.area = plane_area;
    CPM_LAUNCH(ctx, directional_light_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(samples4), n, L, ls, ctx->dir_hint);
    CPM_LAUNCH_CHECK(ctx, "directional_light_kernel");
    return CPM_OK;
}

int cpm_point_light_samples(cpm_ctx* ctx, const float* samples4, int n, const float radiance[4],
                            const float position[4], float* ls, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n >= 0, "cpm_point_light_samples: n < 0");
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, samples4 && radiance && position && ls, "cpm_point_light_samples: null argument");
    Light L = {};
    for (int a = 0; a < 3; ++a) { L.radiance[a] = radiance[a]; L.a[a] = position[a]; }
    CPM_LAUNCH(ctx, point_light_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(samples4), n, L, ls);
    CPM_LAUNCH_CHECK(ctx, "point_light_kernel");
    return CPM_OK;
}

int cpm_light_sample_box_intersection(cpm_ctx* ctx, const float* ls, int n, const float aabb[8], float* isect2,
                                      cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n >= 0, "cpm_light_sample_box_intersection: n < 0");
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, ls && aabb && isect2, "cpm_light_sample_box_intersection: null argument");
    Box b;
    for (int a = 0; a < 3; ++a) { b.mn[a] = aabb[a]; b.mx[a] = aabb[4 + a]; }
    CPM_LAUNCH(ctx, box_intersection_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, ls, n, b,
                       reinterpret_cast<float2*>(isect2));
    CPM_LAUNCH_CHECK(ctx, "box_intersection_kernel");
    return CPM_OK;
}

int cpm_light_sample_mesh_intersection(cpm_ctx* ctx, const float* vertices3, const int32_t* indices, int n_indices,
                                       const float* ls, int n, float* isect2, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n >= 0 && n_indices >= 0, "cpm_light_sample_mesh_intersection: negative size");
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, (vertices3 && indices) || n_indices == 0, "cpm_light_sample_mesh_intersection: null mesh");
    CPM_REQUIRE(ctx, ls && isect2, "cpm_light_sample_mesh_intersection: null argument");
    CPM_LAUNCH(ctx, mesh_intersection_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, vertices3,
                       indices, n_indices, ls, n, reinterpret_cast<float2*>(isect2));
    CPM_LAUNCH_CHECK(ctx, "mesh_intersection_kernel");
    return CPM_OK;
}

}  // extern "C"
