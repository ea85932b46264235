// cpm_comm.hip -- the path's one exchange step: the sum of the per-GPU irradiance grids.
//
// Photons shard by index range (photon i = light sample i + RNG stream i, whatever the shard:
// ref progressivephotonmapping/cl/photontracer.cl:102,123,166 address photons by photonOffset + thread), every GPU
// bins and gathers its shard into its own full-size light volume, and the volumes are summed with ONE collective per
// frame -- RCCL over xGMI, enqueued on the caller's stream.  Nothing else on the data path communicates.
// In the reference this is where PhotonToLightVolumeProcessorCL::process hands the light volume on
// (ref processor/photontolightvolumeprocessorcl.cpp:356-412).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first cpm_comm_* call), so a single-GPU host needs no RCCL
// and libcpm_hip.so carries no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>

#include "cpm_ctx.h"

using namespace cpm;

struct cpm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    // touched-brick reduce: union mask, compact payload, its size read back once per call
    uint32_t* d_count = nullptr;
    uint32_t* h_count = nullptr;
};

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    Rccl& R = g_rccl;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) { R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (R.lib) break; }
    if (!R.lib) { R.error = "librccl.so.1 not found (dlopen)"; return; }
    auto sym = [&](const char* s) -> void* {
        void* p = dlsym(R.lib, s);
        if (!p && R.error.empty()) R.error = std::string("librccl: missing symbol ") + s;
        return p;
    };
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
    R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
    R.Reduce = reinterpret_cast<decltype(R.Reduce)>(sym("ncclReduce"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
}

const Rccl* rccl(cpm_ctx* ctx) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty()) { set_error(ctx, CPM_ERR_UNSUPPORTED, "RCCL", g_rccl.error.c_str()); return nullptr; }
    return &g_rccl;
}

#define CPM_NCCL_CHECK(ctx, R, expr)                                                                            \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return set_error((ctx), CPM_ERR_DEVICE, #expr, (R)->GetErrorString(r_));         \
    } while (0)

int alloc_side(cpm_ctx* ctx, cpm_comm* c) {
    CPM_HIP_CHECK(ctx, hipMalloc((void**)&c->d_count, 16));
    CPM_HIP_CHECK(ctx, hipHostMalloc((void**)&c->h_count, 16, hipHostMallocDefault));
    return CPM_OK;
}

// ---- touched bricks: (mask) -> compact list of brick ids; pack / unpack of their 4 x 4 x 4 voxels

// mask -> ascending list of the marked brick ids (every rank derives the same list from the same union mask) and its
// length: one 1024-thread workgroup, each thread counts a contiguous run of bricks, block prefix, ordered write
__global__ __launch_bounds__(1024) void brick_list_kernel(const uint8_t* __restrict__ mask, uint32_t nb, uint32_t* __restrict__ list,
                                                          uint32_t* __restrict__ count) {
    __shared__ uint32_t s_w[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = min((uint32_t)t * per, nb), b1 = min(b0 + per, nb);
    uint32_t c = 0;
    for (uint32_t b = b0; b < b1; ++b) c += mask[b] != 0;
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) before += s_w[w]; total += s_w[w]; }
    uint32_t pos = before + incl - c;
    for (uint32_t b = b0; b < b1; ++b) if (mask[b] != 0) list[pos++] = b;
    if (t == 0) *count = total;
}

template <bool PACK>
__global__ __launch_bounds__(64) void brick_copy_kernel(const uint32_t* __restrict__ list, int dx, int dy, int dz, int channels,
                                                        int bxn, int byn, const float* __restrict__ grid_in, float* __restrict__ grid_out,
                                                        float* __restrict__ packed) {
    const uint32_t b = list[blockIdx.x];
    const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
    const int l = threadIdx.x;
    const int x = bx * 4 + (l & 3), y = by * 4 + ((l >> 2) & 3), z = bz * 4 + (l >> 4);
    const bool in = x < dx && y < dy && z < dz;
    const size_t v = (size_t)x + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z);
    for (int c = 0; c < channels; ++c) {
        const size_t p = ((size_t)blockIdx.x * 64 + l) * channels + c;
        if (PACK) packed[p] = in ? grid_in[v * channels + c] : 0.f;
        else if (in) grid_out[v * channels + c] = packed[p];
    }
}

}  // namespace

extern "C" {

int cpm_comm_get_unique_id(cpm_ctx* ctx, uint8_t* id_out) {
    if (!id_out) return ctx ? set_error(ctx, CPM_ERR_INVALID_ARGUMENT, "cpm_comm_get_unique_id", "null id") : CPM_ERR_INVALID_ARGUMENT;
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    static_assert(NCCL_UNIQUE_ID_BYTES == CPM_COMM_ID_BYTES, "cpm.h: CPM_COMM_ID_BYTES");
    ncclUniqueId id;
    CPM_NCCL_CHECK(ctx, R, R->GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return CPM_OK;
}

int cpm_comm_create(cpm_ctx* ctx, const uint8_t* id_bytes, int rank, int n_ranks, cpm_comm** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, id_bytes && out, "cpm_comm_create: null argument");
    CPM_REQUIRE(ctx, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "cpm_comm_create: rank / size");
    *out = nullptr;
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    cpm_comm* c = new (std::nothrow) cpm_comm();
    if (!c) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_comm_create", "host allocation failed");
    c->rank = rank; c->size = n_ranks; c->device = ctx->device;
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = R->CommInitRank(&c->comm, n_ranks, id, rank);
    if (r != ncclSuccess) { delete c; return set_error(ctx, CPM_ERR_DEVICE, "ncclCommInitRank", R->GetErrorString(r)); }
    int rc = alloc_side(ctx, c);
    if (rc) { cpm_comm_destroy(c); return rc; }
    *out = c;
    return CPM_OK;
}

int cpm_comm_create_all(cpm_ctx* const* ctxs, int n, cpm_comm** comms_out) {
    if (!ctxs || n < 1 || !comms_out || !ctxs[0]) return CPM_ERR_INVALID_ARGUMENT;
    cpm_ctx* ctx0 = ctxs[0];
    const Rccl* R = rccl(ctx0);
    if (!R) return CPM_ERR_UNSUPPORTED;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return set_error(ctx0, CPM_ERR_INVALID_ARGUMENT, "cpm_comm_create_all", "null context");
        devs[i] = ctxs[i]->device;
        comms_out[i] = nullptr;
    }
    CPM_NCCL_CHECK(ctx0, R, R->CommInitAll(comms.data(), n, devs.data()));
    // on any failure below: every end created so far is destroyed (with its RCCL communicator), the RCCL communicators not
    // yet wrapped are destroyed directly, and comms_out is all NULL again
    auto fail = [&](int from, int rc) {
        for (int j = 0; j < n; ++j) {
            if (comms_out[j]) { cpm_comm_destroy(comms_out[j]); comms_out[j] = nullptr; }
            else if (j >= from && comms[j] && R->CommDestroy) { (void)hipSetDevice(devs[j]); (void)R->CommDestroy(comms[j]); }
        }
        return rc;
    };
    for (int i = 0; i < n; ++i) {
        cpm_comm* c = new (std::nothrow) cpm_comm();
        if (!c) return fail(i, set_error(ctx0, CPM_ERR_OUT_OF_MEMORY, "cpm_comm_create_all", "host allocation failed"));
        c->comm = comms[i]; c->rank = i; c->size = n; c->device = devs[i];
        comms_out[i] = c;
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) return fail(i + 1, set_error(ctx0, CPM_ERR_DEVICE, "cpm_comm_create_all", "hipSetDevice"));
        int rc = alloc_side(ctxs[i], c);
        if (rc) return fail(i + 1, rc);
    }
    return CPM_OK;
}

void cpm_comm_destroy(cpm_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_count) (void)hipFree(c->d_count);
    if (c->h_count) (void)hipHostFree(c->h_count);
    delete c;
}

int cpm_comm_rank(const cpm_comm* c) { return c ? c->rank : -1; }
int cpm_comm_size(const cpm_comm* c) { return c ? c->size : 0; }

int cpm_allreduce_grid(cpm_ctx* ctx, cpm_comm* comm, const float* send, float* recv, size_t count, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && ((send && recv) || count == 0), "cpm_allreduce_grid: null argument");
    if (count == 0) return CPM_OK;
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    ProfScope ps(ctx, "rccl_allreduce_grid", (hipStream_t)stream);
    CPM_NCCL_CHECK(ctx, R, R->AllReduce(send, recv, count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream));
    return CPM_OK;
}

int cpm_reduce_grid(cpm_ctx* ctx, cpm_comm* comm, const float* send, float* recv, size_t count, int root, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && (send || count == 0), "cpm_reduce_grid: null argument");
    CPM_REQUIRE(ctx, root >= 0 && root < comm->size, "cpm_reduce_grid: root");
    CPM_REQUIRE(ctx, comm->rank != root || recv || count == 0, "cpm_reduce_grid: the root needs a receive buffer");
    if (count == 0) return CPM_OK;
    if (comm->size == 1) {
        if (recv != send) CPM_HIP_CHECK(ctx, hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return CPM_OK;
    }
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    ProfScope ps(ctx, "rccl_reduce_grid", (hipStream_t)stream);
    CPM_NCCL_CHECK(ctx, R, R->Reduce(send, recv, count, ncclFloat32, ncclSum, root, comm->comm, (hipStream_t)stream));
    return CPM_OK;
}

int cpm_allreduce_grids(cpm_ctx* const* ctxs, cpm_comm* const* comms, float* const* grids, size_t count, const cpm_stream* streams, int n) {
    if (!ctxs || !comms || !grids || n < 1 || !ctxs[0]) return CPM_ERR_INVALID_ARGUMENT;
    cpm_ctx* ctx0 = ctxs[0];
    if (count == 0 || n == 1) return CPM_OK;
    for (int i = 0; i < n; ++i)
        if (!comms[i] || !grids[i]) return set_error(ctx0, CPM_ERR_INVALID_ARGUMENT, "cpm_allreduce_grids", "null communicator or grid");
    const Rccl* R = rccl(ctx0);
    if (!R) return CPM_ERR_UNSUPPORTED;
    CPM_NCCL_CHECK(ctx0, R, R->GroupStart());
    for (int i = 0; i < n; ++i) {
        ncclResult_t r = R->AllReduce(grids[i], grids[i], count, ncclFloat32, ncclSum, comms[i]->comm,
                                      (hipStream_t)(streams ? streams[i] : nullptr));
        if (r != ncclSuccess) { (void)R->GroupEnd(); return set_error(ctx0, CPM_ERR_DEVICE, "ncclAllReduce", R->GetErrorString(r)); }
    }
    CPM_NCCL_CHECK(ctx0, R, R->GroupEnd());
    return CPM_OK;
}

// The delta path touches few bricks (a TF edit re-traces ~0.5 % of the photons): summing 8 MiB over xGMI for them costs
// of the order of the whole update.  Here the ranks agree on the UNION of their touched 4x4x4-voxel bricks (one small
// max-reduce of the byte mask: brick b = bx + ceil(dx/4) * (by + ceil(dy/4) * bz), the mask of
// cpm_mark_touched_bricks), pack the voxels of those bricks of their PARTIAL light volumes, sum only that, and write the
// sums into `total`: total[brick] = sum over ranks of partial[brick] for every brick of the union; all other voxels of
// `total` are left as they are (no rank changed them, so the previous sum still holds).  total may alias partial.
// Falls back to the dense reduce when the union is more than a quarter of the bricks.  One 4-byte read-back per call:
// the collective's element count must be known on the host (ref: the reference's own read-back of the changed-photon
// count, processor/progressivephotontracercl.cpp:343-345,374).
int cpm_allreduce_grid_bricks(cpm_ctx* ctx, cpm_comm* comm, const float* partial, float* total, const cpm_grid_desc* gd,
                              uint8_t* brick_mask, uint32_t* n_union_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && partial && total && gd && brick_mask, "cpm_allreduce_grid_bricks: null argument");
    CPM_REQUIRE(ctx, gd->channels == 1 || gd->channels == 4, "cpm_allreduce_grid_bricks: channels");
    CPM_REQUIRE(ctx, gd->dims[0] >= 1 && gd->dims[1] >= 1 && gd->dims[2] >= 1, "cpm_allreduce_grid_bricks: dims");
    hipStream_t s = (hipStream_t)stream;
    const int bxn = div_up(gd->dims[0], 4), byn = div_up(gd->dims[1], 4), bzn = div_up(gd->dims[2], 4);
    const size_t nb = (size_t)bxn * byn * bzn;
    const size_t cells = (size_t)gd->dims[0] * gd->dims[1] * gd->dims[2];
    const Rccl* R = comm->size > 1 ? rccl(ctx) : nullptr;
    if (comm->size > 1 && !R) return CPM_ERR_UNSUPPORTED;
    if (comm->size > 1)
        CPM_NCCL_CHECK(ctx, R, R->AllReduce(brick_mask, brick_mask, nb, ncclUint8, ncclMax, comm->comm, s));
    uint32_t* list = (uint32_t*)scratch(ctx, CPM_SCR_MISC, nb * sizeof(uint32_t));
    if (!list) return CPM_ERR_OUT_OF_MEMORY;
    CPM_LAUNCH(ctx, brick_list_kernel, dim3(1), dim3(1024), 0, s, brick_mask, (uint32_t)nb, list, comm->d_count);
    CPM_LAUNCH_CHECK(ctx, "brick_list_kernel");
    CPM_HIP_CHECK(ctx, hipMemcpyAsync(comm->h_count, comm->d_count, 4, hipMemcpyDeviceToHost, s));
    CPM_HIP_CHECK(ctx, hipStreamSynchronize(s));
    const uint32_t n_union = comm->h_count[0];
    if (n_union_out) *n_union_out = n_union;
    if (n_union == 0) return CPM_OK;
    if (comm->size > 1 && (size_t)n_union * 4 > nb) {  // dense is cheaper than pack + sum + unpack
        CPM_NCCL_CHECK(ctx, R, R->AllReduce(partial, total, cells * gd->channels, ncclFloat32, ncclSum, comm->comm, s));
        return CPM_OK;
    }
    const size_t packed_count = (size_t)n_union * 64 * gd->channels;
    float* packed = (float*)scratch(ctx, CPM_SCR_SMALL, packed_count * sizeof(float));
    if (!packed) return CPM_ERR_OUT_OF_MEMORY;
    CPM_LAUNCH(ctx, brick_copy_kernel<true>, dim3(n_union), dim3(64), 0, s, list, gd->dims[0], gd->dims[1], gd->dims[2], gd->channels, bxn, byn, partial, total, packed);
    if (comm->size > 1)
        CPM_NCCL_CHECK(ctx, R, R->AllReduce(packed, packed, packed_count, ncclFloat32, ncclSum, comm->comm, s));
    CPM_LAUNCH(ctx, brick_copy_kernel<false>, dim3(n_union), dim3(64), 0, s, list, gd->dims[0], gd->dims[1], gd->dims[2], gd->channels, bxn, byn, partial, total, packed);
    CPM_LAUNCH_CHECK(ctx, "brick_copy_kernel");
    return CPM_OK;
}

}  // extern "C"
