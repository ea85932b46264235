// cpm_temporal.hip -- temporal interpolation between time steps.
//
// Replaces mixKernel (ref uniformgridcl/cl/buffermixer.cl:37-48, built by BufferMixerCL::compileKernel,
// uniformgridcl/buffermixercl.cpp:230-243) used by UniformGrid3DPlayerProcessor::process
// (uniformgridcl/processors/uniformgrid3dplayerprocessor.cpp:87-115), and the fragment shader
// glsl/volume_mix.frag driven by VolumeSequencePlayer::process (processors/volumesequenceplayer.cpp:87-140).
//
// MI355X mapping: pure streaming kernels, the only ones on the path whose roofline is HBM bandwidth
// in the literal sense.  Each lane moves 16 bytes per operand per iteration (global_load_dwordx4 /
// global_store_dwordx4, 1 KiB per wave-instruction), two iterations in flight, grid sized to ~8
// workgroups per CU and grid-strided beyond that; the unaligned head/tail of raw buffers is
// finished by scalar lanes.
#include "cpm_ctx.h"

using namespace cpm;

namespace {

// OpenCL mix(x, y, a) = x + (y - x) * a, uncontracted (-ffp-contract=off)
CPM_DEV float mix_cl(float x, float y, float a) { return x + (y - x) * a; }
// GLSL mix(x, y, a) = x * (1 - a) + y * a
CPM_DEV float mix_glsl(float x, float y, float a, float one_minus_a) { return x * one_minus_a + y * a; }

// convert_ushort2(float2): default rounding of float -> integer conversions is toward zero
CPM_DEV uint32_t mix_u16_pair(uint32_t x, uint32_t y, float a) {
    float lo = mix_cl((float)(x & 0xffffu), (float)(y & 0xffffu), a);
    float hi = mix_cl((float)(x >> 16), (float)(y >> 16), a);
    return ((uint32_t)(int)lo & 0xffffu) | ((uint32_t)(int)hi << 16);
}

__global__ __launch_bounds__(256) void mix_f32_kernel(const float* __restrict__ x, const float* __restrict__ y, float a,
                                                      size_t n, float* __restrict__ out) {
    const size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    float4* o4 = reinterpret_cast<float4*>(out);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = x4[i], q = y4[i];
        o4[i] = make_float4(mix_cl(p.x, q.x, a), mix_cl(p.y, q.y, a), mix_cl(p.z, q.z, a), mix_cl(p.w, q.w, a));
    }
    size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = mix_cl(x[t], y[t], a);
}

__global__ __launch_bounds__(256) void mix_u16x2_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ y,
                                                        float a, size_t n_pairs, uint32_t* __restrict__ out) {
    const size_t n4 = n_pairs >> 2;
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    const uint4* y4 = reinterpret_cast<const uint4*>(y);
    uint4* o4 = reinterpret_cast<uint4*>(out);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        uint4 p = x4[i], q = y4[i];
        o4[i] = make_uint4(mix_u16_pair(p.x, q.x, a), mix_u16_pair(p.y, q.y, a), mix_u16_pair(p.z, q.z, a),
                           mix_u16_pair(p.w, q.w, a));
    }
    size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_pairs) out[t] = mix_u16_pair(x[t], y[t], a);
}

// texture() of a normalised integer format returns v / (2^b - 1); the colour attachment of the same
// format stores round-to-nearest(clamp(f, 0, 1) * (2^b - 1)).
template <int BITS>
CPM_DEV uint32_t mix_unorm(uint32_t x, uint32_t y, float a, float oma) {
    constexpr float maxv = (float)((1u << BITS) - 1u);
    float r = mix_glsl((float)x / maxv, (float)y / maxv, a, oma);
    r = min_(max_(r, 0.0f), 1.0f);
    return (uint32_t)(int)__builtin_rintf(r * maxv);
}
// u8: the 256 possible v / 255 quotients come from an LDS table filled with the same division, so a
// voxel costs two ds_read instead of two correctly-rounded divisions (the kernel stays a byte mover)
CPM_DEV uint32_t mix_u8_word(const float* lut, uint32_t x, uint32_t y, float a, float oma) {
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 32; b += 8) {
        float v = mix_glsl(lut[(x >> b) & 0xffu], lut[(y >> b) & 0xffu], a, oma);
        v = min_(max_(v, 0.0f), 1.0f);
        r |= (uint32_t)(int)__builtin_rintf(v * 255.0f) << b;
    }
    return r;
}
CPM_DEV uint32_t mix_u16_word(uint32_t x, uint32_t y, float a, float oma) {
    return mix_unorm<16>(x & 0xffffu, y & 0xffffu, a, oma) | (mix_unorm<16>(x >> 16, y >> 16, a, oma) << 16);
}

// volumes are allocated with a 16-byte tail pad (cpm_volume_create), so whole uint4 words cover them
template <int DT>
__global__ __launch_bounds__(256) void volume_mix_kernel(const uint4* __restrict__ x, const uint4* __restrict__ y, float a,
                                                         size_t n16, uint4* __restrict__ out) {
    __shared__ float lut[256];
    if (DT == CPM_U8) {
        lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
        __syncthreads();
    }
    const float oma = 1.0f - a;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 p = x[i], q = y[i], r;
        if (DT == CPM_U8) {
            r = make_uint4(mix_u8_word(lut, p.x, q.x, a, oma), mix_u8_word(lut, p.y, q.y, a, oma),
                           mix_u8_word(lut, p.z, q.z, a, oma), mix_u8_word(lut, p.w, q.w, a, oma));
        } else if (DT == CPM_U16) {
            r = make_uint4(mix_u16_word(p.x, q.x, a, oma), mix_u16_word(p.y, q.y, a, oma), mix_u16_word(p.z, q.z, a, oma),
                           mix_u16_word(p.w, q.w, a, oma));
        } else {
            r = make_uint4(__float_as_uint(mix_glsl(__uint_as_float(p.x), __uint_as_float(q.x), a, oma)),
                           __float_as_uint(mix_glsl(__uint_as_float(p.y), __uint_as_float(q.y), a, oma)),
                           __float_as_uint(mix_glsl(__uint_as_float(p.z), __uint_as_float(q.z), a, oma)),
                           __float_as_uint(mix_glsl(__uint_as_float(p.w), __uint_as_float(q.w), a, oma)));
        }
        out[i] = r;
    }
}

// Measured on MI355X (tools/mix_bw.py): past the Infinity Cache one 16-byte vector per lane and no grid
// stride streams at the copy ceiling (1 GiB operands: 6.0 TB/s, the same as torch.lerp; 8 workgroups
// per CU grid-striding: 5.2 TB/s); launches of a few thousand workgroups (a 256^3 u8 volume) finish
// sooner with 8 workgroups per CU looping twice (8.2 vs 9.6 us).
int stream_grid(const cpm_ctx* ctx, size_t vectors) {
    size_t blocks = (vectors + 255) / 256;
    const int forced = ctx->dbg.stream_wg_per_cu;  // -1 = by size; >= 0 forces (tuning hook)
    int per_cu = forced >= 0 ? forced : (blocks <= 16384 ? 8 : 0);
    const size_t cap = per_cu > 0 ? (size_t)256 * per_cu : (size_t)0x7fffffff;
    if (blocks > cap) blocks = cap;
    return blocks ? (int)blocks : 1;
}

}  // namespace

extern "C" {

// tuning hook: workgroups per CU for the streaming kernels (0 = one vector per lane, no grid stride)
void cpm_debug_set_stream_wg_per_cu(cpm_ctx* ctx, int n) { if (ctx) ctx->dbg.stream_wg_per_cu = n; }

int cpm_mix_buffers(cpm_ctx* ctx, const void* x, const void* y, float a, size_t n_elements, int type, void* out,
                    cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, type == CPM_MIX_F32 || type == CPM_MIX_U16X2, "cpm_mix_buffers: type");
    if (n_elements == 0) return CPM_OK;
    CPM_REQUIRE(ctx, x && y && out, "cpm_mix_buffers: null buffer");
    CPM_REQUIRE(ctx, (((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15u) == 0, "cpm_mix_buffers: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int grid = stream_grid(ctx, (n_elements >> 2) + 256);
    if (type == CPM_MIX_F32) {
        CPM_LAUNCH(ctx, mix_f32_kernel, dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)y, a, n_elements, (float*)out);
        CPM_LAUNCH_CHECK(ctx, "mix_f32_kernel");
    } else {
        CPM_LAUNCH(ctx, mix_u16x2_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t*)x, (const uint32_t*)y, a, n_elements,
                   (uint32_t*)out);
        CPM_LAUNCH_CHECK(ctx, "mix_u16x2_kernel");
    }
    return CPM_OK;
}

int cpm_volume_mix(cpm_ctx* ctx, const cpm_volume* v0, const cpm_volume* v1, float weight, cpm_volume* out,
                   cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, v0 && v1 && out, "cpm_volume_mix: null argument");
    CPM_REQUIRE(ctx, memcmp(v0->desc.dims, v1->desc.dims, sizeof(v0->desc.dims)) == 0 && v0->desc.dtype == v1->desc.dtype &&
                         memcmp(v0->desc.dims, out->desc.dims, sizeof(v0->desc.dims)) == 0 && v0->desc.dtype == out->desc.dtype,
                "cpm_volume_mix: volumes differ in shape or type");
    size_t n16 = (v0->bytes + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    const uint4 *x = (const uint4*)v0->voxels, *y = (const uint4*)v1->voxels;
    uint4* o = (uint4*)out->voxels;
    int grid = stream_grid(ctx, n16);
    switch (v0->desc.dtype) {
        case CPM_U8: CPM_LAUNCH(ctx, volume_mix_kernel<CPM_U8>, dim3(grid), dim3(256), 0, s, x, y, weight, n16, o); break;
        case CPM_U16: CPM_LAUNCH(ctx, volume_mix_kernel<CPM_U16>, dim3(grid), dim3(256), 0, s, x, y, weight, n16, o); break;
        default: CPM_LAUNCH(ctx, volume_mix_kernel<CPM_F32>, dim3(grid), dim3(256), 0, s, x, y, weight, n16, o); break;
    }
    CPM_LAUNCH_CHECK(ctx, "volume_mix_kernel");
    // The tracer's footprint copy of the mixed volume is left to whoever needs it (cpm::trace_volume_source): a trace over all the
    // samples re-derives it first; the correlated update's re-traces of a few per cent of the photons read the linear block instead,
    // so a time step that is served by an update never pays the re-layout (17 us at 256^3 u8, as much as half of the update's kernels).
    out->quads_stale = true;
    return CPM_OK;
}

}  // extern "C"
