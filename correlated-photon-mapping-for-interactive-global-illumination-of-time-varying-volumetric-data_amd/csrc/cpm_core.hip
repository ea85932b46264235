// cpm_core.hip -- context, volume / transfer-function residency, host helpers.
#include <math.h>
#include <stdlib.h>

#include <new>

#include "cpm_ctx.h"
#include "cpm/cpm_profile.h"

static thread_local std::string g_create_error;

namespace cpm {

int set_error(cpm_ctx* ctx, int status, const char* what, const char* detail) {
    std::string msg = std::string(what ? what : "") + ": " + (detail ? detail : "");
    if (ctx) ctx->last_error = msg; else g_create_error = msg;
    return status;
}

void* scratch(cpm_ctx* ctx, int slot, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (ctx->scratch_bytes[slot] >= bytes) return ctx->scratch[slot];
    if (ctx->scratch[slot]) { (void)hipFree(ctx->scratch[slot]); ctx->scratch[slot] = nullptr; ctx->scratch_bytes[slot] = 0; }
    size_t want = bytes + bytes / 4 + 256;  // grow with slack
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "hipMalloc(scratch)", hipGetErrorString(e)); return nullptr; }
    ctx->scratch[slot] = p;
    ctx->scratch_bytes[slot] = want;
    return p;
}

bool affine_from_matrix(const float m[16], Affine& out) {
    // column-major: m[4*col + row]; require diag + translation only
    const int offdiag[] = { 1, 2, 3, 4, 6, 7, 8, 9, 11 };
    for (int i : offdiag) if (m[i] != 0.0f) return false;
    if (m[15] != 1.0f) return false;
    out.sx = m[0]; out.sy = m[5]; out.sz = m[10];
    out.tx = m[12]; out.ty = m[13]; out.tz = m[14];
    return true;
}

ProfScope::ProfScope(cpm_ctx* c, const char* name, hipStream_t stream) : ctx(c), s(stream) {
    if (!ctx || !ctx->profiling) { ctx = nullptr; return; }
    hipEvent_t a = nullptr;
    if (ctx->prof_pool.size() >= 2) {
        a = ctx->prof_pool.back(); ctx->prof_pool.pop_back();
        b = ctx->prof_pool.back(); ctx->prof_pool.pop_back();
    } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        ctx = nullptr;
        return;
    }
    (void)hipEventRecord(a, s);
    ctx->prof_pending.push_back({ name, a, b });
}
ProfScope::~ProfScope() {
    if (ctx) (void)hipEventRecord(b, s);
}

}  // namespace cpm

using namespace cpm;

// voxels -> quads (cpm_volume::quads): a lane takes VEC elements of rows (y, z), (y + 1, z), (y, z + 1), (y + 1, z + 1) (4 bytes each
// when the row length allows) and writes them interleaved (16 bytes); COPY also writes the first row through to the volume's
// own linear block (a device->device cpm_volume_update is this one launch instead of a copy plus a launch).
template <typename T, int VEC, bool COPY>
__global__ void __launch_bounds__(256) quads_kernel(const T* __restrict__ src, T* __restrict__ linear, T* __restrict__ quads,
                                                     uint32_t dx, uint32_t dy, uint32_t dz, uint32_t chunks_per_row, uint32_t n_chunks) {
    const size_t slice = (size_t)dx * dy;
    for (uint32_t c = blockIdx.x * 256u + threadIdx.x; c < n_chunks; c += gridDim.x * 256u) {
        const uint32_t row = c / chunks_per_row, cx = (c - row * chunks_per_row) * VEC;
        const uint32_t z = row / dy, y = row - z * dy;
        const size_t at = (size_t)row * dx + cx, up = (y + 1 < dy) ? dx : 0, back = (z + 1 < dz) ? slice : 0;
        T r[4][VEC], o[4 * VEC];
        __builtin_memcpy(r[0], __builtin_assume_aligned(src + at, VEC * sizeof(T)), sizeof(r[0]));
        __builtin_memcpy(r[1], __builtin_assume_aligned(src + at + up, VEC * sizeof(T)), sizeof(r[0]));
        __builtin_memcpy(r[2], __builtin_assume_aligned(src + at + back, VEC * sizeof(T)), sizeof(r[0]));
        __builtin_memcpy(r[3], __builtin_assume_aligned(src + at + up + back, VEC * sizeof(T)), sizeof(r[0]));
#pragma unroll
        for (int i = 0; i < VEC; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[4 * i + k] = r[k][i];
        __builtin_memcpy(__builtin_assume_aligned(quads + 4 * at, VEC * sizeof(T)), o, sizeof(o));
        if (COPY) __builtin_memcpy(__builtin_assume_aligned(linear + at, VEC * sizeof(T)), r[0], sizeof(r[0]));
    }
}

template <typename T, bool COPY>
static int launch_quads(cpm_ctx* ctx, cpm_volume* vol, const void* src, hipStream_t s) {
    const uint32_t dx = (uint32_t)vol->desc.dims[0], dy = (uint32_t)vol->desc.dims[1], dz = (uint32_t)vol->desc.dims[2];
    // 4 bytes of each of the four rows in, 16 bytes out per lane: consecutive lanes write consecutive 16-byte pieces (a lane
    // that took 16 bytes per row wrote 64 of its own: four store instructions each touching a quarter of 64 lines -- 31 us at 256^3)
    constexpr int kVec = 4 / (int)sizeof(T);
    const bool wide = dx % kVec == 0 && (reinterpret_cast<uintptr_t>(src) & 3u) == 0;
    const unsigned long long chunks = (unsigned long long)(wide ? dx / kVec : dx) * dy * dz;
    CPM_REQUIRE(ctx, chunks < (1ull << 32), "cpm_volume: too large");
    const uint32_t grid = (uint32_t)std::min<unsigned long long>((chunks + 255) / 256, 256ull * 64);
    if (wide)
        CPM_LAUNCH(ctx, (quads_kernel<T, kVec, COPY>), dim3(grid), dim3(256), 0, s, static_cast<const T*>(src), static_cast<T*>(vol->voxels),
                   static_cast<T*>(vol->quads), dx, dy, dz, dx / kVec, (uint32_t)chunks);
    else
        CPM_LAUNCH(ctx, (quads_kernel<T, 1, COPY>), dim3(grid), dim3(256), 0, s, static_cast<const T*>(src), static_cast<T*>(vol->voxels),
                   static_cast<T*>(vol->quads), dx, dy, dz, dx, (uint32_t)chunks);
    CPM_LAUNCH_CHECK(ctx, "quads_kernel");
    return CPM_OK;
}

namespace cpm {
int build_quads(cpm_ctx* ctx, cpm_volume* vol, const void* src, bool copy_linear, hipStream_t s) {
    vol->quads_stale = false;
    switch (vol->desc.dtype) {
        case CPM_U8: return copy_linear ? launch_quads<uint8_t, true>(ctx, vol, src, s) : launch_quads<uint8_t, false>(ctx, vol, src, s);
        case CPM_U16: return copy_linear ? launch_quads<uint16_t, true>(ctx, vol, src, s) : launch_quads<uint16_t, false>(ctx, vol, src, s);
        default: return copy_linear ? launch_quads<uint32_t, true>(ctx, vol, src, s) : launch_quads<uint32_t, false>(ctx, vol, src, s);
    }
}
}  // namespace cpm

extern "C" {

int cpm_abi_version(void) { return CPM_ABI_VERSION; }

// ---- measurement hooks (include/cpm/cpm_profile.h)
void cpm_profile_enable(cpm_ctx* ctx, int on) { if (ctx) ctx->profiling = on != 0; }

int cpm_set_photon_layout(cpm_ctx* ctx, int layout) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    if (layout != CPM_PHOTONS_INTERLEAVED && layout != CPM_PHOTONS_PLANAR)
        return cpm::set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_set_photon_layout", "layout must be CPM_PHOTONS_INTERLEAVED or CPM_PHOTONS_PLANAR");
    ctx->photon_layout = layout;
    return CPM_OK;
}
int cpm_get_photon_layout(const cpm_ctx* ctx) { return ctx ? ctx->photon_layout : CPM_PHOTONS_INTERLEAVED; }
int cpm_records_describe(cpm_ctx* ctx, const float* base, int layout, size_t n_records) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    if (!base || (layout != CPM_PHOTONS_INTERLEAVED && layout != CPM_PHOTONS_PLANAR) || n_records >= (1ull << 31))
        return cpm::set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_records_describe", "null buffer, unknown layout or more than 2^31 records");
    for (auto& b : ctx->record_buffers)
        if (b.base == base) { b.layout = layout; b.n_records = n_records; return CPM_OK; }
    if (ctx->record_buffers.size() >= 16) return cpm::set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_records_describe", "16 buffers described already (cpm_records_forget)");
    ctx->record_buffers.push_back({ base, layout, n_records });
    return CPM_OK;
}
int cpm_records_forget(cpm_ctx* ctx, const float* base) {
    if (!ctx) return CPM_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < ctx->record_buffers.size(); ++i)
        if (ctx->record_buffers[i].base == base) { ctx->record_buffers.erase(ctx->record_buffers.begin() + (long)i); return CPM_OK; }
    return CPM_OK;
}
void cpm_profile_reset(cpm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);  // one host thread may drive several contexts / GPUs
    (void)hipDeviceSynchronize();
    for (auto& r : ctx->prof_pending) { ctx->prof_pool.push_back(r.a); ctx->prof_pool.push_back(r.b); }
    ctx->prof_pending.clear();
    ctx->prof_entries.clear();
}
int cpm_profile_collect(cpm_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& r : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            std::string name(r.name);
            while (!name.empty() && name.front() == '(') name.erase(name.begin());
            while (!name.empty() && name.back() == ')') name.pop_back();
            cpm_prof_entry* e = nullptr;
            for (auto& x : ctx->prof_entries) if (x.name == name) { e = &x; break; }
            if (!e) { ctx->prof_entries.push_back({ name, 0.0, 0 }); e = &ctx->prof_entries.back(); }
            e->total_ms += ms;
            e->calls += 1;
        }
        ctx->prof_pool.push_back(r.a);
        ctx->prof_pool.push_back(r.b);
    }
    ctx->prof_pending.clear();
    return (int)ctx->prof_entries.size();
}
const char* cpm_profile_name(const cpm_ctx* ctx, int i) { return (ctx && i >= 0 && i < (int)ctx->prof_entries.size()) ? ctx->prof_entries[i].name.c_str() : ""; }
double cpm_profile_total_ms(const cpm_ctx* ctx, int i) { return (ctx && i >= 0 && i < (int)ctx->prof_entries.size()) ? ctx->prof_entries[i].total_ms : 0.0; }
long cpm_profile_calls(const cpm_ctx* ctx, int i) { return (ctx && i >= 0 && i < (int)ctx->prof_entries.size()) ? ctx->prof_entries[i].calls : 0; }

int cpm_create(int device, cpm_ctx** out) {
    if (!out) return set_error(nullptr, CPM_ERR_INVALID_ARGUMENT, "cpm_create", "out == NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(nullptr, CPM_ERR_NO_DEVICE, "cpm_create",
                         "no HIP device visible; libcpm_hip has no CPU fallback");
    if (device < 0 || device >= count)
        return set_error(nullptr, CPM_ERR_INVALID_ARGUMENT, "cpm_create", "device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return set_error(nullptr, CPM_ERR_DEVICE, "hipSetDevice", hipGetErrorString(e));
    cpm_ctx* ctx = new (std::nothrow) cpm_ctx();
    if (!ctx) return set_error(nullptr, CPM_ERR_OUT_OF_MEMORY, "cpm_create", "host allocation failed");
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        // the LDS a workgroup may take once hipFuncSetAttribute has allowed it (gfx950: 160 KiB; the attributes' largest -- ROCm releases
        // differ in which of them carries the opt-in limit)
        size_t lds = prop.sharedMemPerBlock;
        if (prop.maxSharedMemoryPerMultiProcessor > lds) lds = prop.maxSharedMemoryPerMultiProcessor;
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess && (size_t)optin > lds) lds = (size_t)optin;
        (void)hipGetLastError();
        if (lds >= 16 * 1024) ctx->lds_per_block = lds;
    }
    if (hipMalloc((void**)&ctx->dir_hint, 16 * sizeof(float)) != hipSuccess || hipMemset(ctx->dir_hint, 0, 16 * sizeof(float)) != hipSuccess) {
        delete ctx;
        return set_error(nullptr, CPM_ERR_OUT_OF_MEMORY, "cpm_create", "device allocation failed");
    }
    *out = ctx;
    return CPM_OK;
}

void cpm_destroy(cpm_ctx* ctx) {
    if (!ctx) return;
    for (int i = 0; i < 8; ++i) if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->dir_hint) (void)hipFree(ctx->dir_hint);
    for (int i = 0; i < cpm_ctx::kTfStageSlots; ++i)
        if (ctx->tf_stage_done[i]) { (void)hipEventSynchronize(ctx->tf_stage_done[i]); (void)hipEventDestroy(ctx->tf_stage_done[i]); }
    if (ctx->tf_stage) (void)hipHostFree(ctx->tf_stage);
    for (auto& r : ctx->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    delete ctx;
}

const char* cpm_last_error_string(const cpm_ctx* ctx) {
    return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

// glibc random_r TYPE_3 (what srand()/rand() run), see cpm.h
void cpm_glibc_rand_sequence(uint32_t seed, uint32_t* out, size_t n) {
    const size_t total = 344 + n;
    int32_t* r = (int32_t*)malloc(total * sizeof(int32_t));
    if (!r) return;
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
        long long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        long long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (int32_t)w;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (size_t i = 34; i < total; ++i) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    for (size_t k = 0; k < n; ++k) out[k] = ((uint32_t)r[344 + k]) >> 1;
    free(r);
}

static void default_matrices(const int32_t dims[3], float t2i[16], float i2t[16]) {
    memset(t2i, 0, 16 * sizeof(float));
    memset(i2t, 0, 16 * sizeof(float));
    for (int a = 0; a < 3; ++a) {
        // textureToIndex = scale(dim) then translate(-0.5); indexToTexture its inverse
        t2i[5 * a] = (float)dims[a];
        t2i[12 + a] = -0.5f;
        i2t[5 * a] = 1.0f / (float)dims[a];
        i2t[12 + a] = 0.5f / (float)dims[a];
    }
    t2i[15] = 1.0f;
    i2t[15] = 1.0f;
}

void cpm_volume_desc_default(cpm_volume_desc* d, const int32_t dims[3], int32_t dtype) {
    memset(d, 0, sizeof(*d));
    for (int a = 0; a < 3; ++a) d->dims[a] = dims[a];
    d->dtype = dtype;
    d->format_scaling = 0.0f;
    d->format_offset = 0.0f;
    default_matrices(dims, d->texture_to_index, d->index_to_texture);
}

void cpm_grid_desc_default(cpm_grid_desc* d, const int32_t dims[3], int32_t channels) {
    memset(d, 0, sizeof(*d));
    for (int a = 0; a < 3; ++a) d->dims[a] = dims[a];
    d->channels = channels;
    default_matrices(dims, d->texture_to_index, d->index_to_texture);
}

float cpm_relative_irradiance_scale(double radius, double n_photons) {
    const double pi = 3.14159265358979323846;
    double vol = radius * radius * radius * (pi * 4. / 3.);
    return (float)((1. / pi) / (vol * n_photons));
}

static size_t dtype_size(int dtype) { return dtype == CPM_U8 ? 1 : (dtype == CPM_U16 ? 2 : 4); }


int cpm_volume_create(cpm_ctx* ctx, const cpm_volume_desc* desc, const void* voxels, int is_device,
                      cpm_stream stream, cpm_volume** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, desc && out, "cpm_volume_create: null argument");
    CPM_REQUIRE(ctx, desc->dtype >= CPM_U8 && desc->dtype <= CPM_F32, "cpm_volume_create: dtype");
    CPM_REQUIRE(ctx, desc->dims[0] >= 2 && desc->dims[1] >= 1 && desc->dims[2] >= 1, "cpm_volume_create: dims (x >= 2)");
    Affine a;
    if (!affine_from_matrix(desc->texture_to_index, a) || !affine_from_matrix(desc->index_to_texture, a))
        return set_error(ctx, CPM_ERR_UNSUPPORTED, "cpm_volume_create", "texture/index matrices must be scale + translate");
    cpm_volume* v = new (std::nothrow) cpm_volume();
    if (!v) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_volume_create", "host allocation failed");
    v->desc = *desc;
    v->bytes = (size_t)desc->dims[0] * desc->dims[1] * desc->dims[2] * dtype_size(desc->dtype);
    hipError_t e = hipMalloc(&v->voxels, v->bytes + 16);  // tail pad: paired x loads never leave the allocation
    if (e != hipSuccess) { delete v; return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "hipMalloc(volume)", hipGetErrorString(e)); }
    e = hipMalloc(&v->quads, 4 * v->bytes + 64);
    if (e != hipSuccess) { (void)hipFree(v->voxels); delete v; return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "hipMalloc(volume quads)", hipGetErrorString(e)); }
    *out = v;
    e = hipMemsetAsync((char*)v->quads + 4 * v->bytes, 0, 64, (hipStream_t)stream);
    if (e != hipSuccess) { cpm_volume_destroy(ctx, v); *out = nullptr; return set_error(ctx, CPM_ERR_DEVICE, "hipMemsetAsync(volume)", hipGetErrorString(e)); }
    if (!voxels) {  // storage for a volume produced on the device (cpm_volume_mix): zero-filled
        e = hipMemsetAsync(v->voxels, 0, v->bytes + 16, (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemsetAsync(v->quads, 0, 4 * v->bytes, (hipStream_t)stream);
        if (e != hipSuccess) { cpm_volume_destroy(ctx, v); *out = nullptr; return set_error(ctx, CPM_ERR_DEVICE, "hipMemsetAsync(volume)", hipGetErrorString(e)); }
        return CPM_OK;
    }
    int rc = cpm_volume_update(ctx, v, voxels, is_device, stream);
    if (rc != CPM_OK) { cpm_volume_destroy(ctx, v); *out = nullptr; }
    return rc;
}

int cpm_volume_update(cpm_ctx* ctx, cpm_volume* vol, const void* voxels, int is_device, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, vol && voxels, "cpm_volume_update: null argument");
    hipStream_t s = (hipStream_t)stream;
    CPM_HIP_CHECK(ctx, hipMemsetAsync((char*)vol->voxels + vol->bytes, 0, 16, s));
    if (is_device && voxels != vol->voxels && (reinterpret_cast<uintptr_t>(voxels) & 3u) == 0)
        return build_quads(ctx, vol, voxels, true, s);  // copy and quads in one launch
    if (voxels != vol->voxels)
        CPM_HIP_CHECK(ctx, hipMemcpyAsync(vol->voxels, voxels, vol->bytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (!is_device) CPM_HIP_CHECK(ctx, hipStreamSynchronize(s));  // caller may free the host buffer
    return build_quads(ctx, vol, vol->voxels, false, s);
}

void* cpm_volume_device_data(const cpm_volume* vol, size_t* bytes) {
    if (!vol) return nullptr;
    if (bytes) *bytes = vol->bytes;
    return vol->voxels;
}

int cpm_volume_download(cpm_ctx* ctx, const cpm_volume* vol, void* voxels_host, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, vol && voxels_host, "cpm_volume_download: null argument");
    hipStream_t s = (hipStream_t)stream;
    CPM_HIP_CHECK(ctx, hipMemcpyAsync(voxels_host, vol->voxels, vol->bytes, hipMemcpyDeviceToHost, s));
    CPM_HIP_CHECK(ctx, hipStreamSynchronize(s));
    return CPM_OK;
}

void cpm_volume_destroy(cpm_ctx* ctx, cpm_volume* vol) {
    (void)ctx;
    if (!vol) return;
    if (vol->voxels) (void)hipFree(vol->voxels);
    if (vol->quads) (void)hipFree(vol->quads);
    delete vol;
}

__global__ void tf_alpha_kernel(const float* __restrict__ rgba, int width, float* __restrict__ alpha) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) alpha[i] = rgba[4 * i + 3];
}

int cpm_tf_create(cpm_ctx* ctx, const float* rgba, int width, int is_device, cpm_stream stream, cpm_tf** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, rgba && out, "cpm_tf_create: null argument");
    CPM_REQUIRE(ctx, width >= 2 && width <= 16384, "cpm_tf_create: width must be in [2, 16384]");
    cpm_tf* tf = new (std::nothrow) cpm_tf();
    if (!tf) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_tf_create", "host allocation failed");
    tf->width = width;
    hipError_t e = hipMalloc((void**)&tf->rgba, (size_t)width * 5 * sizeof(float));
    if (e != hipSuccess) { delete tf; return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "hipMalloc(tf)", hipGetErrorString(e)); }
    tf->alpha = tf->rgba + (size_t)width * 4;
    *out = tf;
    int rc = cpm_tf_update(ctx, tf, rgba, is_device, stream);
    if (rc != CPM_OK) { cpm_tf_destroy(ctx, tf); *out = nullptr; }
    return rc;
}

// LUT from a pinned host slot (read over the host link by the kernel itself) -> rgba + alpha column, one launch
__global__ void tf_upload_kernel(const float4* __restrict__ src, int width, float4* __restrict__ rgba, float* __restrict__ alpha) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) {
        const float4 c = src[i];
        rgba[i] = c;
        alpha[i] = c.w;
    }
}

int cpm_tf_update(cpm_ctx* ctx, cpm_tf* tf, const float* rgba, int is_device, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, tf && rgba, "cpm_tf_update: null argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t floats = (size_t)tf->width * 4;
    if (is_device) {
        CPM_HIP_CHECK(ctx, hipMemcpyAsync(tf->rgba, rgba, floats * sizeof(float), hipMemcpyDeviceToDevice, s));
        CPM_LAUNCH(ctx, tf_alpha_kernel, dim3(div_up(tf->width, 256)), dim3(256), 0, s, tf->rgba, tf->width, tf->alpha);
        CPM_LAUNCH_CHECK(ctx, "tf_alpha_kernel");
        return CPM_OK;
    }
    // host data: consumed here (copied into a pinned slot), read by the launch -- the caller's array may go away on return
    // and the host does not wait for the stream
    if (ctx->tf_stage_floats < floats) {
        for (int i = 0; i < cpm_ctx::kTfStageSlots; ++i)
            if (ctx->tf_stage_done[i]) CPM_HIP_CHECK(ctx, hipEventSynchronize(ctx->tf_stage_done[i]));
        if (ctx->tf_stage) (void)hipHostFree(ctx->tf_stage);
        ctx->tf_stage = nullptr;
        ctx->tf_stage_floats = 0;
        CPM_HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->tf_stage, cpm_ctx::kTfStageSlots * floats * sizeof(float),
                                         hipHostMallocMapped | hipHostMallocCoherent));  // the upload kernel reads a slot the CPU rewrote: never from a stale L2 line
        CPM_HIP_CHECK(ctx, hipHostGetDevicePointer((void**)&ctx->tf_stage_dev, ctx->tf_stage, 0));
        ctx->tf_stage_floats = floats;
        for (int i = 0; i < cpm_ctx::kTfStageSlots; ++i)
            if (!ctx->tf_stage_done[i]) CPM_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->tf_stage_done[i], hipEventDisableTiming));
    }
    const int slot = ctx->tf_stage_next;
    ctx->tf_stage_next = (slot + 1) % cpm_ctx::kTfStageSlots;
    CPM_HIP_CHECK(ctx, hipEventSynchronize(ctx->tf_stage_done[slot]));  // (an event never recorded is complete)
    float* host = ctx->tf_stage + (size_t)slot * ctx->tf_stage_floats;
    memcpy(host, rgba, floats * sizeof(float));
    CPM_LAUNCH(ctx, tf_upload_kernel, dim3(div_up(tf->width, 256)), dim3(256), 0, s,
               reinterpret_cast<const float4*>(ctx->tf_stage_dev + (size_t)slot * ctx->tf_stage_floats), tf->width,
               reinterpret_cast<float4*>(tf->rgba), tf->alpha);
    CPM_LAUNCH_CHECK(ctx, "tf_upload_kernel");
    CPM_HIP_CHECK(ctx, hipEventRecord(ctx->tf_stage_done[slot], s));
    return CPM_OK;
}

void cpm_tf_destroy(cpm_ctx* ctx, cpm_tf* tf) {
    (void)ctx;
    if (!tf) return;
    if (tf->rgba) (void)hipFree(tf->rgba);
    delete tf;
}

}  // extern "C"
