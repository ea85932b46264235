// cpm_gl.hip -- the consumer side of the light volume: HIP <-> OpenGL sharing (include/cpm/cpm.h, "OpenGL sharing").
//
// Replaces Inviwo's CL-GL sharing on this path: `SyncCLGL` + `BufferCLGL` for the photon buffer and `VolumeCLGL` for the
// light volume the raycaster samples (ref processor/photontolightvolumeprocessorcl.cpp:184-194 acquire, :404-406 the
// buffer -> 3-D image copy; processor/progressivephotontracercl.cpp:93 `glsharing`).
// CDNA has no image hardware: hipMalloc3DArray answers "operation not supported" on gfx950 (measured), so a GL 3-D texture
// cannot be mapped as an array the way VolumeCLGL maps it.  What can be shared is a GL BUFFER object: it is mapped to a
// device pointer the kernels take as any other buffer.  The light volume therefore reaches the raycaster through a pixel
// unpack buffer: texels written into the mapped buffer on the device (float32 as they are -- the gather may even write
// straight into it -- or converted to float16 in one launch), then the HOST's glTexSubImage3D from that buffer, a copy
// inside the GL driver.  No host round trip either way.
//
// Nothing here links OpenGL: the interop entry points live in the HIP runtime, and whether the calling thread has a current
// context is asked of the GL library the HOST application loaded (glXGetCurrentContext / eglGetCurrentContext found in the
// process).  Without one every registration returns CPM_ERR_UNSUPPORTED -- the runtime is not even asked.
#include "cpm_ctx.h"

#include <dlfcn.h>
#include <hip/hip_fp16.h>
#include <hip/hip_gl_interop.h>

using namespace cpm;

struct cpm_gl_resource {
    hipGraphicsResource* res = nullptr;
    bool mapped = false;
};

namespace {

// does the calling thread have a current OpenGL context?  Asked of the GL / EGL library already in the process.
bool gl_context_current() {
    typedef void* (*get_ctx_fn)();
    const char* names[] = { "glXGetCurrentContext", "eglGetCurrentContext" };
    const char* libs[] = { "libGL.so.1", "libEGL.so.1" };
    for (int i = 0; i < 2; ++i) {
        get_ctx_fn fn = reinterpret_cast<get_ctx_fn>(dlsym(RTLD_DEFAULT, names[i]));
        void* h = nullptr;
        if (!fn) {  // loaded with local visibility (a plug-in host): look inside it without loading anything new
            h = dlopen(libs[i], RTLD_LAZY | RTLD_NOLOAD);
            if (h) fn = reinterpret_cast<get_ctx_fn>(dlsym(h, names[i]));
        }
        const bool current = fn && fn() != nullptr;
        if (h) dlclose(h);
        if (current) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void to_half_kernel(const float* __restrict__ src, __half* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
        *reinterpret_cast<__half2*>(dst + i) = lo;
        *reinterpret_cast<__half2*>(dst + i + 2) = hi;
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = __float2half_rn(src[k]);
    }
}

int require_context(cpm_ctx* ctx, const char* what) {
    if (!gl_context_current())
        return set_error(ctx, CPM_ERR_UNSUPPORTED, what, "no current OpenGL context on the calling thread (make the host's context current first)");
    return CPM_OK;
}

}  // namespace

extern "C" {

int cpm_gl_available(cpm_ctx* ctx) {
    if (!ctx) return 0;
    return gl_context_current() ? 1 : 0;
}

int cpm_gl_register_buffer(cpm_ctx* ctx, unsigned gl_buffer, int read_only, cpm_gl_resource** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, out, "cpm_gl_register_buffer: null out");
    *out = nullptr;
    CPM_REQUIRE(ctx, gl_buffer != 0, "cpm_gl_register_buffer: buffer name 0");
    int rc = require_context(ctx, "cpm_gl_register_buffer");
    if (rc) return rc;
    cpm_gl_resource* r = new (std::nothrow) cpm_gl_resource();
    if (!r) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_gl_register_buffer", "host allocation");
    hipError_t e = hipGraphicsGLRegisterBuffer(&r->res, gl_buffer, read_only ? hipGraphicsRegisterFlagsReadOnly : hipGraphicsRegisterFlagsNone);
    if (e != hipSuccess) { delete r; return set_error(ctx, CPM_ERR_DEVICE, "hipGraphicsGLRegisterBuffer", hipGetErrorString(e)); }
    *out = r;
    return CPM_OK;
}

int cpm_gl_acquire(cpm_ctx* ctx, cpm_gl_resource* const* resources, int n, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n >= 0 && n <= 16 && (resources || n == 0), "cpm_gl_acquire: 0 <= n <= 16 resources");
    hipGraphicsResource* list[16];
    for (int i = 0; i < n; ++i) {
        CPM_REQUIRE(ctx, resources[i] && resources[i]->res, "cpm_gl_acquire: null resource");
        CPM_REQUIRE(ctx, !resources[i]->mapped, "cpm_gl_acquire: resource already acquired");
        list[i] = resources[i]->res;
    }
    if (n == 0) return CPM_OK;
    CPM_HIP_CHECK(ctx, hipGraphicsMapResources(n, list, (hipStream_t)stream));
    for (int i = 0; i < n; ++i) resources[i]->mapped = true;
    return CPM_OK;
}

int cpm_gl_release(cpm_ctx* ctx, cpm_gl_resource* const* resources, int n, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, n >= 0 && n <= 16 && (resources || n == 0), "cpm_gl_release: 0 <= n <= 16 resources");
    hipGraphicsResource* list[16];
    for (int i = 0; i < n; ++i) {
        CPM_REQUIRE(ctx, resources[i] && resources[i]->res, "cpm_gl_release: null resource");
        CPM_REQUIRE(ctx, resources[i]->mapped, "cpm_gl_release: resource not acquired");
        list[i] = resources[i]->res;
    }
    if (n == 0) return CPM_OK;
    CPM_HIP_CHECK(ctx, hipGraphicsUnmapResources(n, list, (hipStream_t)stream));
    for (int i = 0; i < n; ++i) resources[i]->mapped = false;
    return CPM_OK;
}

int cpm_gl_buffer_pointer(cpm_ctx* ctx, cpm_gl_resource* buffer, void** dev_ptr, size_t* bytes) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, buffer && buffer->res && dev_ptr, "cpm_gl_buffer_pointer: null argument");
    CPM_REQUIRE(ctx, buffer->mapped, "cpm_gl_buffer_pointer: acquire the resource first");
    size_t sz = 0;
    CPM_HIP_CHECK(ctx, hipGraphicsResourceGetMappedPointer(dev_ptr, &sz, buffer->res));
    if (bytes) *bytes = sz;
    return CPM_OK;
}

int cpm_light_volume_texels(cpm_ctx* ctx, const float* light_volume, size_t n, int texel, void* texels_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, texel == CPM_GL_TEXEL_F32 || texel == CPM_GL_TEXEL_F16, "cpm_light_volume_texels: texel format");
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, light_volume && texels_out, "cpm_light_volume_texels: null buffer");
    CPM_REQUIRE_ALIGNED16(ctx, light_volume, "cpm_light_volume_texels");
    CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(texels_out) & 3u) == 0, "cpm_light_volume_texels: texels_out must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (texel == CPM_GL_TEXEL_F32) {
        if (texels_out != light_volume) CPM_HIP_CHECK(ctx, hipMemcpyAsync(texels_out, light_volume, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return CPM_OK;
    }
    CPM_LAUNCH(ctx, to_half_kernel, dim3((unsigned)div_up((long long)((n + 3) / 4), 256)), dim3(256), 0, s, light_volume, static_cast<__half*>(texels_out), n);
    CPM_LAUNCH_CHECK(ctx, "to_half_kernel");
    return CPM_OK;
}

int cpm_gl_copy_to_buffer(cpm_ctx* ctx, const float* light_volume, size_t n, int texel, cpm_gl_resource* buffer, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, buffer && buffer->res, "cpm_gl_copy_to_buffer: null resource");
    CPM_REQUIRE(ctx, buffer->mapped, "cpm_gl_copy_to_buffer: acquire the resource first");
    void* dst = nullptr;
    size_t bytes = 0;
    CPM_HIP_CHECK(ctx, hipGraphicsResourceGetMappedPointer(&dst, &bytes, buffer->res));
    CPM_REQUIRE(ctx, bytes >= n * (texel == CPM_GL_TEXEL_F16 ? 2u : 4u), "cpm_gl_copy_to_buffer: the GL buffer is smaller than the light volume's texels");
    return cpm_light_volume_texels(ctx, light_volume, n, texel, dst, stream);
}

void cpm_gl_unregister(cpm_ctx* ctx, cpm_gl_resource* resource) {
    if (!resource) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (resource->res) {
        if (resource->mapped) (void)hipGraphicsUnmapResources(1, &resource->res, nullptr);
        (void)hipGraphicsUnregisterResource(resource->res);
    }
    delete resource;
}

}  // extern "C"
