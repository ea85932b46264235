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
#include <stdlib.h>
#include <rccl/rccl.h>

#include <chrono>
#include <mutex>
#include <new>

#include "cpm_ctx.h"
#include "cpm/cpm_profile.h"

using namespace cpm;

struct cpm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    // touched-brick reduce: union mask, compact payload, its size read back once per call
    uint32_t* d_count = nullptr;
    uint32_t* h_count = nullptr;
};

// cpm_allreduce_grid_sparse: everything one (communicator, grid shape) needs, allocated once
struct cpm_sparse_reduce {
    static constexpr int kSlots = 8;  // tickets in flight (issued, not yet completed)
    cpm_comm* comm = nullptr;
    int device = 0;
    int dims[3] = { 0, 0, 0 }, channels = 1, bxn = 0, byn = 0, bzn = 0;
    uint32_t nb = 0;
    size_t cells = 0;
    uint8_t* mask = nullptr;        // nb bytes (+ 16): this rank's bricks, then the union
    uint32_t* list = nullptr;       // nb: union bricks, ascending
    uint32_t* slot = nullptr;       // nb: brick -> position in the list
    uint32_t* count = nullptr;      // device word: size of the union
    float* payload = nullptr;       // nb * 64 * channels floats (the dense size: any capacity fits)
    unsigned long long* mailbox = nullptr;      // pinned host, kSlots words: ticket << 32 | union count
    unsigned long long* mailbox_dev = nullptr;
    uint64_t next_ticket = 1;
    struct Slot {
        uint64_t ticket = 0;
        const float* partial = nullptr; float* total = nullptr;
        int root = -1; uint32_t capacity = 0; bool dense = false, completed = true;
        uint32_t n_union = 0; bool known = false;
        hipStream_t stream = nullptr;
    } slots[kSlots];
    uint32_t last_union[2] = { 0, 0 };  // union counts of the two most recent tickets whose mailbox has been read, [0] the older
    uint64_t last_union_ticket = 0;
};

// cpm_reduce_grid_bricklists: everything one (communicator, grid shape, root) needs
constexpr int kBricklistMaxRanks = 16;
struct cpm_bricklist_reduce {
    static constexpr int kSlots = 4;  // tickets in flight (opened, not yet completed)
    cpm_comm* comm = nullptr;
    int device = 0, root = 0;
    int dims[3] = { 0, 0, 0 }, channels = 1, bxn = 0, byn = 0, bzn = 0;
    uint32_t nb = 0, room = 0;   // 4x4x4 bricks of the grid; slots a sender's buffer holds (nb rounded up to 64)
    size_t cells = 0;
    uint32_t* ctl = nullptr;      // sender, device: per slot 2 words (slots handed out, workgroups done), zero between launches
    uint32_t* slot_of = nullptr;  // root, device: per sender (rank order, the root left out) nb words: brick -> slot in that sender's
                                  // segment; never cleared -- an entry counts only if the slot it names carries the brick's id
    uint32_t* who = nullptr;      // root, device: nb words, bit i = sender i lists the brick in the pass under way; zero between passes
    // pinned host words: [kSlots][kBricklistMaxRanks] ticket << 32 | the brick count of rank r's segment -- written by the last workgroup of
    // the sender's fill launch (its own word) and by the root's add launch (every sender's word, from the segment's header); behind them
    // the same again for the exchanges repeated at exact size (the root's acknowledgement of the header it then found)
    unsigned long long* mailbox = nullptr;
    unsigned long long* mailbox_dev = nullptr;
    // per slot: a sender's segment (room for every brick) / the root's received segments; the root's buffer of a repeated exchange
    void* seg[kSlots] = { nullptr, nullptr, nullptr, nullptr };
    size_t seg_bytes[kSlots] = { 0, 0, 0, 0 };
    void* again = nullptr;
    size_t again_bytes = 0;
    uint64_t next_ticket = 1;
    bool poisoned = false;  // a header that was not its ticket's: the ranks' capacities may have diverged -- every later call fails
    struct Slot {
        uint64_t ticket = 0;
        float* grid = nullptr;                      // the root's
        uint32_t cap[kBricklistMaxRanks] = {};      // per sender (a sender fills its own entry only)
        uint32_t counts[kBricklistMaxRanks] = {};
        bool known[kBricklistMaxRanks] = {};
        bool completed = true, exchanged = false;
        int resent = 0;
        hipStream_t stream = nullptr;               // the exchange's
    } slots[kSlots];
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
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    Rccl& R = g_rccl;
    // CPM_RCCL_LIBRARY: another build of RCCL -- or the tests' double (tests/fake_rccl: shared memory between ranks that share one GPU)
    const char* override_path = getenv("CPM_RCCL_LIBRARY");
    if (override_path && override_path[0]) {
        R.lib = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) { R.error = std::string("CPM_RCCL_LIBRARY: cannot load ") + override_path; return; }
    }
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) { if (R.lib) break; R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
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
    R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
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

// ---- sparse, synchronisation-free reduce (cpm_allreduce_grid_sparse) ---------------------------------------------------
// Bricks are 4 x 4 x 4 voxels, brick b = bx + ceil(dx/4) * (by + ceil(dy/4) * bz) (the numbering of
// cpm_mark_touched_bricks).  Four short launches around two collectives, none of which the host waits for:
//   brick_nonzero_kernel   this rank's bricks that hold a non-zero voxel -> byte mask            (or the caller's mask)
//   [ncclAllReduce max]    the UNION of the ranks' masks: 1 byte per brick
//   brick_slots_kernel     union mask -> ascending brick list + brick -> slot table + count; the count also goes to a pinned
//                          host mailbox (ticket << 32 | count) -- the host reads it LATER, to size the following payloads
//   brick_pack_kernel      the first min(count, capacity) bricks -> payload (zeros behind them); nothing when count > capacity
//   [ncclAllReduce sum]    capacity * 64 * channels floats -- a size the host fixed BEFORE the launch
//   brick_unpack_kernel    payload -> total (zeros or nothing elsewhere); nothing when count > capacity (the caller's
//                          cpm_sparse_reduce_complete then enqueues the dense sum: `partial` is untouched)

constexpr uint32_t kNoSlot = 0xffffffffu;

// 16 lanes per brick, one row of 4 voxels each: a row = one 16-byte load (4 for 4 channels) when dx is a multiple of 4;
// the 16 row flags of a brick are OR-ed across its lanes (4 shuffles), lane 0 of the brick writes the byte
template <int CH, bool VEC>
__global__ __launch_bounds__(256) void brick_nonzero_kernel(const float* __restrict__ grid, int dx, int dy, int dz, int bxn, int byn,
                                                            uint32_t nb, uint8_t* __restrict__ mask) {
    const uint32_t b = blockIdx.x * 16u + (threadIdx.x >> 4);
    const int r = threadIdx.x & 15;
    int any = 0;
    if (b < nb) {
        const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
        const int y = by * 4 + (r & 3), z = bz * 4 + (r >> 2);
        if (y < dy && z < dz) {
            const size_t v = (size_t)(bx * 4) + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z);
            if (VEC) {
                const float4* q = reinterpret_cast<const float4*>(grid + v * CH);
#pragma unroll
                for (int c = 0; c < CH; ++c) { const float4 f = q[c]; any |= f.x != 0.f || f.y != 0.f || f.z != 0.f || f.w != 0.f; }
            } else {
                for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
                    for (int c = 0; c < CH; ++c) any |= grid[(v + x) * CH + c] != 0.f;
            }
        }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) any |= __shfl_xor(any, off, 64);
    if (r == 0 && b < nb) mask[b] = any ? 1 : 0;
}

// the non-zero bytes of a word as 0 / 1 bytes (a caller's mask may hold any non-zero value)
CPM_DEV uint32_t nonzero_bytes(uint32_t w) { return ((w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7; }

// union mask -> ascending list of its bricks, the inverse table slot[brick] (kNoSlot outside the union) and the count.
// Workgroup g takes bricks [4096 g, 4096 g + 4096), 4 per lane (one word of the mask); what lies before its chunk it counts
// itself (16-byte loads of at most nb bytes out of L2: no scan launch, no order between workgroups -- every rank derives the
// same list from the same mask).  The last workgroup knows the total: it goes to the device word the pack / unpack launches
// read and to the host's mailbox.  `mask` is readable (and ignored) up to the next multiple of 16 bytes.
__global__ __launch_bounds__(1024) void brick_slots_kernel(const uint8_t* __restrict__ mask, uint32_t nb, uint32_t* __restrict__ list,
                                                           uint32_t* __restrict__ slot, uint32_t* __restrict__ count,
                                                           unsigned long long* mailbox, uint32_t ticket) {
    __shared__ uint32_t s_w[16], s_p[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t chunk = blockIdx.x * 4096u;
    // my four bricks (requested first), then the bricks before the chunk
    const uint32_t b0 = chunk + 4u * (uint32_t)t;
    uint32_t w = b0 < nb ? *reinterpret_cast<const uint32_t*>(mask + b0) : 0u;
    if (b0 + 4 > nb && b0 < nb) w &= 0xffffffffu >> (8u * (b0 + 4u - nb));  // bytes at and beyond nb do not exist
    uint32_t before = 0;
    for (uint32_t i = (uint32_t)t * 16u; i < chunk; i += 1024u * 16u) {
        const uint4 m = *reinterpret_cast<const uint4*>(mask + i);
        before += __popc(nonzero_bytes(m.x)) + __popc(nonzero_bytes(m.y)) + __popc(nonzero_bytes(m.z)) + __popc(nonzero_bytes(m.w));
    }
    const uint32_t nz = nonzero_bytes(w);
    const uint32_t c = __popc(nz);
    uint32_t incl = c, pre = before;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    if (lane == 63) s_w[wave] = incl;
    if (lane == 0) s_p[wave] = pre;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { base += s_p[q]; if (q < wave) base += s_w[q]; total += s_p[q] + s_w[q]; }
    uint32_t pos = base + incl - c;
    if (b0 < nb) {
        uint32_t sl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = (nz >> (8 * q)) & 1u;
            sl[q] = in ? pos : kNoSlot;
            if (in) list[pos++] = b0 + (uint32_t)q;
        }
        if (b0 + 4 <= nb) *reinterpret_cast<uint4*>(slot + b0) = make_uint4(sl[0], sl[1], sl[2], sl[3]);
        else for (uint32_t q = 0; b0 + q < nb; ++q) slot[b0 + q] = sl[q];
    }
    if (blockIdx.x == gridDim.x - 1 && t == 0) {
        *count = total;
        if (mailbox) __hip_atomic_store(mailbox, ((unsigned long long)ticket << 32) | (unsigned long long)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// 16 lanes per brick, one row of 4 voxels each (a 16-byte piece of a grid row <-> a 16-byte piece of the payload, whose
// bricks are 64 * CH consecutive floats in (z, y, x, channel) order)
template <int CH, bool VEC>
__global__ __launch_bounds__(256) void brick_pack_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t capacity,
                                                         int dx, int dy, int dz, int bxn, int byn, const float* __restrict__ grid,
                                                         float* __restrict__ payload) {
    const uint32_t n = *count;
    if (n > capacity) return;  // overflow: the dense sum follows (cpm_sparse_reduce_complete)
    const uint32_t s = blockIdx.x * 16u + (threadIdx.x >> 4);
    if (s >= capacity) return;
    const int r = threadIdx.x & 15;
    float4 f[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) f[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < n) {
        const uint32_t b = list[s];
        const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
        const int y = by * 4 + (r & 3), z = bz * 4 + (r >> 2);
        if (y < dy && z < dz) {
            const size_t v = (size_t)(bx * 4) + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z);
            if (VEC) {
#pragma unroll
                for (int c = 0; c < CH; ++c) f[c] = reinterpret_cast<const float4*>(grid + v * CH)[c];
            } else {
                float* ff = reinterpret_cast<float*>(f);
                for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
                    for (int c = 0; c < CH; ++c) ff[x * CH + c] = grid[(v + x) * CH + c];
            }
        }
    }
    float4* o = reinterpret_cast<float4*>(payload) + ((size_t)s * 16 + r) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) o[c] = f[c];
}

// ZERO: 16 lanes per brick of the GRID (all of them) -- `total` is a separate buffer that must become the whole sum: a union
// brick takes its sums from the payload, any other brick is zeroed.  Otherwise 16 lanes per brick of the UNION (the list):
// everything else of `total` is left alone
template <int CH, bool VEC, bool ZERO>
__global__ __launch_bounds__(256) void brick_unpack_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ slot,
                                                           const uint32_t* __restrict__ count, uint32_t capacity,
                                                           uint32_t nb, int dx, int dy, int dz, int bxn, int byn,
                                                           const float* __restrict__ payload, float* __restrict__ total) {
    const uint32_t n = *count;
    if (n > capacity) return;
    const uint32_t i = blockIdx.x * 16u + (threadIdx.x >> 4);
    uint32_t b, s;
    if (ZERO) { if (i >= nb) return; b = i; s = slot[b]; }
    else { if (i >= n) return; s = i; b = list[s]; }
    const int r = threadIdx.x & 15;
    const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
    const int y = by * 4 + (r & 3), z = bz * 4 + (r >> 2);
    if (y >= dy || z >= dz) return;
    float4 f[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) f[c] = s == kNoSlot ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(payload)[((size_t)s * 16 + r) * CH + c];
    const size_t v = (size_t)(bx * 4) + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z);
    if (VEC) {
#pragma unroll
        for (int c = 0; c < CH; ++c) reinterpret_cast<float4*>(total + v * CH)[c] = f[c];
    } else {
        const float* ff = reinterpret_cast<const float*>(f);
        for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
            for (int c = 0; c < CH; ++c) total[(v + x) * CH + c] = ff[x * CH + c];
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

// ---- cpm_allreduce_grid_sparse ------------------------------------------------------------------------------------------

uint32_t cpm_sparse_reduce_capacity_for(uint32_t n_bricks, long long previous_union) {
    if (n_bricks == 0) return 0;
    unsigned long long c;
    if (previous_union < 0) c = ((unsigned long long)n_bricks / 4 + 63ull) & ~63ull;                     // nothing known yet: a quarter
    else c = ((unsigned long long)previous_union + (unsigned long long)previous_union / 4 + 64ull + 63ull) & ~63ull;  // + 25 % + 64
    if (c * 2 > n_bricks) return n_bricks;  // beyond half of the bricks pack + sum + unpack move more than the dense sum
    return (uint32_t)c;
}

int cpm_sparse_reduce_create(cpm_ctx* ctx, cpm_comm* comm, const cpm_grid_desc* gd, cpm_sparse_reduce** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && gd && out, "cpm_sparse_reduce_create: null argument");
    CPM_REQUIRE(ctx, gd->channels == 1 || gd->channels == 4, "cpm_sparse_reduce_create: channels");
    CPM_REQUIRE(ctx, gd->dims[0] >= 1 && gd->dims[1] >= 1 && gd->dims[2] >= 1, "cpm_sparse_reduce_create: dims");
    CPM_REQUIRE(ctx, (unsigned long long)gd->dims[0] * gd->dims[1] * gd->dims[2] < (1ull << 31), "cpm_sparse_reduce_create: more than 2^31 cells");
    *out = nullptr;
    cpm_sparse_reduce* sr = new (std::nothrow) cpm_sparse_reduce();
    if (!sr) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_sparse_reduce_create", "host allocation failed");
    sr->comm = comm; sr->device = ctx->device;
    for (int a = 0; a < 3; ++a) sr->dims[a] = gd->dims[a];
    sr->channels = gd->channels;
    sr->bxn = div_up(gd->dims[0], 4); sr->byn = div_up(gd->dims[1], 4); sr->bzn = div_up(gd->dims[2], 4);
    sr->nb = (uint32_t)((size_t)sr->bxn * sr->byn * sr->bzn);
    sr->cells = (size_t)gd->dims[0] * gd->dims[1] * gd->dims[2];
    const size_t nb = sr->nb;
    const bool ok = hipMalloc((void**)&sr->mask, nb + 16) == hipSuccess && hipMalloc((void**)&sr->list, nb * 4) == hipSuccess &&
                    hipMalloc((void**)&sr->slot, nb * 4) == hipSuccess && hipMalloc((void**)&sr->count, 16) == hipSuccess &&
                    hipMalloc((void**)&sr->payload, nb * 64 * sizeof(float) * (size_t)sr->channels) == hipSuccess &&
                    hipHostMalloc((void**)&sr->mailbox, cpm_sparse_reduce::kSlots * 8, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                    hipHostGetDevicePointer((void**)&sr->mailbox_dev, sr->mailbox, 0) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        cpm_sparse_reduce_destroy(sr);
        return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_sparse_reduce_create", "device / pinned allocation failed");
    }
    for (int i = 0; i < cpm_sparse_reduce::kSlots; ++i) sr->mailbox[i] = 0ull;
    *out = sr;
    return CPM_OK;
}

void cpm_sparse_reduce_destroy(cpm_sparse_reduce* sr) {
    if (!sr) return;
    (void)hipSetDevice(sr->device);
    // the mailbox must outlive the launches that write it
    for (auto& sl : sr->slots) if (sl.ticket && !sl.known) { (void)hipStreamSynchronize(sl.stream); break; }
    if (sr->mask) (void)hipFree(sr->mask);
    if (sr->list) (void)hipFree(sr->list);
    if (sr->slot) (void)hipFree(sr->slot);
    if (sr->count) (void)hipFree(sr->count);
    if (sr->payload) (void)hipFree(sr->payload);
    if (sr->mailbox) (void)hipHostFree(sr->mailbox);
    delete sr;
}

uint32_t cpm_sparse_reduce_bricks(const cpm_sparse_reduce* sr) { return sr ? sr->nb : 0; }

}  // extern "C"

namespace {

// the union count of an issued ticket: a poll of the pinned mailbox its brick_slots_kernel writes (work enqueued behind that
// launch keeps running); after 2 s without it, the stream itself
int sparse_union_count(cpm_ctx* ctx, cpm_sparse_reduce* sr, cpm_sparse_reduce::Slot& sl) {
    if (sl.known) return CPM_OK;
    const volatile unsigned long long* mb = sr->mailbox + (sl.ticket % cpm_sparse_reduce::kSlots);
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long v = __atomic_load_n(mb, __ATOMIC_ACQUIRE);
        if ((uint32_t)(v >> 32) == (uint32_t)sl.ticket) { sl.n_union = (uint32_t)v; sl.known = true; break; }
        if (synced) return set_error(ctx, CPM_ERR_DEVICE, "cpm_sparse_reduce", "the union count of a ticket never arrived");
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            CPM_HIP_CHECK(ctx, hipStreamSynchronize(sl.stream));
            synced = true;
        }
        __builtin_ia32_pause();
    }
    if (sl.ticket > sr->last_union_ticket) {
        sr->last_union[0] = sr->last_union[1]; sr->last_union[1] = sl.n_union; sr->last_union_ticket = sl.ticket;
    }
    return CPM_OK;
}

int sparse_dense_sum(cpm_ctx* ctx, cpm_sparse_reduce* sr, const float* partial, float* total, int root, hipStream_t s) {
    const size_t count = sr->cells * (size_t)sr->channels;
    if (sr->comm->size == 1) {
        if (total != partial) CPM_HIP_CHECK(ctx, hipMemcpyAsync(total, partial, count * sizeof(float), hipMemcpyDeviceToDevice, s));
        return CPM_OK;
    }
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    ProfScope ps(ctx, "rccl_sparse_dense_fallback", s);
    if (root < 0) CPM_NCCL_CHECK(ctx, R, R->AllReduce(partial, total, count, ncclFloat32, ncclSum, sr->comm->comm, s));
    else CPM_NCCL_CHECK(ctx, R, R->Reduce(partial, total, count, ncclFloat32, ncclSum, root, sr->comm->comm, s));
    return CPM_OK;
}

}  // namespace

extern "C" {

int cpm_allreduce_grid_sparse(cpm_ctx* ctx, cpm_sparse_reduce* sr, const float* partial, float* total, const uint8_t* brick_mask,
                              int mask_kind, int root, uint32_t capacity_bricks, uint64_t* ticket_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, sr && partial && total, "cpm_allreduce_grid_sparse: null argument");
    CPM_REQUIRE(ctx, mask_kind == CPM_SPARSE_MASK_TOUCHED || mask_kind == CPM_SPARSE_MASK_NONZERO, "cpm_allreduce_grid_sparse: mask_kind");
    CPM_REQUIRE(ctx, root < sr->comm->size, "cpm_allreduce_grid_sparse: root");
    CPM_REQUIRE_ALIGNED16(ctx, partial, "cpm_allreduce_grid_sparse");
    CPM_REQUIRE_ALIGNED16(ctx, total, "cpm_allreduce_grid_sparse");
    CPM_REQUIRE(ctx, capacity_bricks <= sr->nb, "cpm_allreduce_grid_sparse: capacity beyond the number of bricks");
    hipStream_t s = (hipStream_t)stream;
    const uint64_t ticket = sr->next_ticket;
    cpm_sparse_reduce::Slot& sl = sr->slots[ticket % cpm_sparse_reduce::kSlots];
    CPM_REQUIRE(ctx, sl.completed, "cpm_allreduce_grid_sparse: 8 tickets issued and not completed (cpm_sparse_reduce_complete)");
    // the payload's size is fixed HERE, on the host, from a union count every rank has read from its own mailbox and that is
    // the same number on every rank (the union is): the count of ticket - 2, which has long been written when ticket is
    // issued (its reduce had to finish before the caller could reuse that ticket's grid buffer)
    uint32_t capacity = capacity_bricks;
    if (capacity == 0) {
        long long prev = -1;
        if (ticket >= 3) {
            cpm_sparse_reduce::Slot& old = sr->slots[(ticket - 2) % cpm_sparse_reduce::kSlots];
            if (old.ticket == ticket - 2) { int rc = sparse_union_count(ctx, sr, old); if (rc) return rc; prev = (long long)old.n_union; }
        }
        capacity = cpm_sparse_reduce_capacity_for(sr->nb, prev);
    }
    const bool dense = capacity >= sr->nb;
    const Rccl* R = sr->comm->size > 1 ? rccl(ctx) : nullptr;
    if (sr->comm->size > 1 && !R) return CPM_ERR_UNSUPPORTED;
    const bool vec = (sr->dims[0] & 3) == 0;
    const int dx = sr->dims[0], dy = sr->dims[1], dz = sr->dims[2];
    uint8_t* mask = sr->mask;
    // 1. this rank's bricks
    if (brick_mask) {
        CPM_HIP_CHECK(ctx, hipMemcpyAsync(mask, brick_mask, sr->nb, hipMemcpyDeviceToDevice, s));
    } else {
        const dim3 g((unsigned)div_up(sr->nb, 16));
        if (sr->channels == 1) {
            if (vec) CPM_LAUNCH(ctx, (brick_nonzero_kernel<1, true>), g, dim3(256), 0, s, partial, dx, dy, dz, sr->bxn, sr->byn, sr->nb, mask);
            else CPM_LAUNCH(ctx, (brick_nonzero_kernel<1, false>), g, dim3(256), 0, s, partial, dx, dy, dz, sr->bxn, sr->byn, sr->nb, mask);
        } else {
            if (vec) CPM_LAUNCH(ctx, (brick_nonzero_kernel<4, true>), g, dim3(256), 0, s, partial, dx, dy, dz, sr->bxn, sr->byn, sr->nb, mask);
            else CPM_LAUNCH(ctx, (brick_nonzero_kernel<4, false>), g, dim3(256), 0, s, partial, dx, dy, dz, sr->bxn, sr->byn, sr->nb, mask);
        }
        CPM_LAUNCH_CHECK(ctx, "brick_nonzero_kernel");
    }
    // 2. the union
    if (sr->comm->size > 1) {
        ProfScope ps(ctx, "rccl_sparse_mask", s);
        CPM_NCCL_CHECK(ctx, R, R->AllReduce(mask, mask, sr->nb, ncclUint8, ncclMax, sr->comm->comm, s));
    }
    // 3. list, slots, count (-> mailbox)
    CPM_LAUNCH(ctx, brick_slots_kernel, dim3((unsigned)div_up(sr->nb, 4096)), dim3(1024), 0, s, mask, sr->nb, sr->list, sr->slot, sr->count,
               sr->mailbox_dev + (ticket % cpm_sparse_reduce::kSlots), (uint32_t)ticket);
    CPM_LAUNCH_CHECK(ctx, "brick_slots_kernel");
    if (dense) {
        int rc = sparse_dense_sum(ctx, sr, partial, total, root, s);
        if (rc) return rc;
    } else {
        // 4. pack -> sum -> unpack, all sized by `capacity`
        const dim3 pg((unsigned)div_up(capacity, 16));
#define CPM_SPARSE_PACK(CH, VEC) CPM_LAUNCH(ctx, (brick_pack_kernel<CH, VEC>), pg, dim3(256), 0, s, sr->list, sr->count, capacity, dx, dy, dz, sr->bxn, sr->byn, partial, sr->payload)
        if (sr->channels == 1) { if (vec) CPM_SPARSE_PACK(1, true); else CPM_SPARSE_PACK(1, false); }
        else { if (vec) CPM_SPARSE_PACK(4, true); else CPM_SPARSE_PACK(4, false); }
#undef CPM_SPARSE_PACK
        CPM_LAUNCH_CHECK(ctx, "brick_pack_kernel");
        const size_t n = (size_t)capacity * 64 * (size_t)sr->channels;
        if (sr->comm->size > 1) {
            ProfScope ps(ctx, "rccl_sparse_payload", s);
            if (root < 0) CPM_NCCL_CHECK(ctx, R, R->AllReduce(sr->payload, sr->payload, n, ncclFloat32, ncclSum, sr->comm->comm, s));
            else CPM_NCCL_CHECK(ctx, R, R->Reduce(sr->payload, sr->payload, n, ncclFloat32, ncclSum, root, sr->comm->comm, s));
        }
        if (root < 0 || root == sr->comm->rank) {
            // a separate `total` becomes the whole sum (zeros outside the union) unless the caller's mask says which bricks
            // changed: then everything else of `total` still holds (the delta path)
            const bool zero = total != partial && (!brick_mask || mask_kind == CPM_SPARSE_MASK_NONZERO);
            const dim3 ug((unsigned)div_up(zero ? sr->nb : capacity, 16));
#define CPM_SPARSE_UNPACK(CH, VEC, ZERO) CPM_LAUNCH(ctx, (brick_unpack_kernel<CH, VEC, ZERO>), ug, dim3(256), 0, s, sr->list, sr->slot, sr->count, capacity, sr->nb, dx, dy, dz, sr->bxn, sr->byn, sr->payload, total)
            if (sr->channels == 1) {
                if (vec) { if (zero) CPM_SPARSE_UNPACK(1, true, true); else CPM_SPARSE_UNPACK(1, true, false); }
                else { if (zero) CPM_SPARSE_UNPACK(1, false, true); else CPM_SPARSE_UNPACK(1, false, false); }
            } else {
                if (vec) { if (zero) CPM_SPARSE_UNPACK(4, true, true); else CPM_SPARSE_UNPACK(4, true, false); }
                else { if (zero) CPM_SPARSE_UNPACK(4, false, true); else CPM_SPARSE_UNPACK(4, false, false); }
            }
#undef CPM_SPARSE_UNPACK
            CPM_LAUNCH_CHECK(ctx, "brick_unpack_kernel");
        }
    }
    sl = cpm_sparse_reduce::Slot();
    sl.ticket = ticket; sl.partial = partial; sl.total = total; sl.root = root; sl.capacity = capacity; sl.dense = dense;
    sl.completed = false; sl.stream = s;
    sr->next_ticket = ticket + 1;
    if (ticket_out) *ticket_out = ticket;
    return CPM_OK;
}

int cpm_sparse_reduce_complete(cpm_ctx* ctx, cpm_sparse_reduce* sr, uint64_t ticket, cpm_stream stream, cpm_sparse_reduce_info* info) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, sr && ticket >= 1 && ticket < sr->next_ticket, "cpm_sparse_reduce_complete: no such ticket");
    cpm_sparse_reduce::Slot& sl = sr->slots[ticket % cpm_sparse_reduce::kSlots];
    CPM_REQUIRE(ctx, sl.ticket == ticket, "cpm_sparse_reduce_complete: the ticket is more than 8 calls old");
    int rc = sparse_union_count(ctx, sr, sl);
    if (rc) return rc;
    int mode = sl.dense ? 1 : 0;
    if (!sl.completed && !sl.dense && sl.n_union > sl.capacity) {
        // more bricks than the payload was sized for: pack and unpack did nothing (every rank took the same decision from the
        // same count), `partial` is as it was -- the dense sum, behind the sparse chain on the caller's stream
        rc = sparse_dense_sum(ctx, sr, sl.partial, sl.total, sl.root, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (!sl.dense && sl.n_union > sl.capacity) mode = 2;
    sl.completed = true;
    if (info) {
        const uint64_t brick_bytes = 64ull * sizeof(float) * (uint64_t)sr->channels;
        info->ticket = ticket; info->n_bricks = sr->nb; info->n_union = sl.n_union; info->capacity = sl.capacity; info->mode = mode;
        info->dense_bytes = (uint64_t)sr->cells * sizeof(float) * (uint64_t)sr->channels;
        info->reduce_bytes = (uint64_t)sr->nb + (mode != 1 ? (uint64_t)sl.capacity * brick_bytes : 0ull) + (mode != 0 ? info->dense_bytes : 0ull);
    }
    return CPM_OK;
}

}  // extern "C"

namespace {
__global__ __launch_bounds__(256) void brick_mask_or_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (dst[i] | src[i]) ? 1 : 0;
}
}  // namespace

extern "C" int cpm_brick_mask_or(cpm_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t n, cpm_stream stream) {
    CPM_ENTER(ctx);
    if (n == 0) return CPM_OK;
    CPM_REQUIRE(ctx, dst && src, "cpm_brick_mask_or: null mask");
    CPM_LAUNCH(ctx, brick_mask_or_kernel, dim3((unsigned)div_up((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    CPM_LAUNCH_CHECK(ctx, "brick_mask_or_kernel");
    return CPM_OK;
}

// ---- cpm_reduce_grid_bricklists ----------------------------------------------------------------------------------------------
// Segment layout, slot counter and header: cpm_ctx.h (shared with cpm_gather_fast_segment, which fills a segment straight from the
// brick gather).  A sender's buffer has room for every brick of the grid; the first `capacity` slots travel.

namespace {

__host__ __device__ inline unsigned char* seg_slot(unsigned char* seg, uint32_t s, int ch) { return seg + sizeof(SegHeader) + (size_t)s * seg_slot_bytes(ch); }
__host__ __device__ inline const unsigned char* seg_slot(const unsigned char* seg, uint32_t s, int ch) { return seg + sizeof(SegHeader) + (size_t)s * seg_slot_bytes(ch); }
inline uint32_t round_up_64(uint64_t v) { return (uint32_t)((v + 63ull) & ~63ull); }

// sender, from a dense grid: every listed 4x4x4 brick -> a slot of the segment.  Listed = marked, when the gather left marks (exact there; a
// marked brick of zeros is listed as zeros: harmless), else holding a non-zero value.  A fixed grid of workgroups; every WAVE takes 4-brick quads
// at a stride of all the waves (lit bricks come in clusters: neighbouring quads go to different waves), 16 lanes per brick, a 16-byte piece of a
// grid row each.  Lane k of a wave holds the marks of the wave's k-th quad (one 4-byte load each, all in flight together); the wave counts its
// listed bricks, the workgroup takes its slots with ONE atomic on the segment's counter, and the waves write their bricks at wave-local ranks
// (shuffles; no barrier and no dependent load in the loop: the rows of several quads are in flight together).
// (The first form took a slot per workgroup of 16 bricks and ended every workgroup with the done atomic: 2 x 16 Ki returning atomics on two
// addresses of one cache line at config 4's size, served one after the other -- 390 us for a launch that moves 12 MB; the second walked
// contiguous runs of quads with a marks load per iteration: 32 dependent round trips per wave, and the lit clusters on a few waves: 34 - 52 us.)
constexpr uint32_t kPackQuadsPerWave = 64;   // at most: lane k holds quad k's marks
constexpr int kPackThreads = 1024;            // 16 waves per workgroup: few workgroups (one slot atomic and one done atomic each), many waves
template <int CH, bool VEC>
__global__ __launch_bounds__(kPackThreads) void bricklist_pack_grid_kernel(const float* __restrict__ grid, const uint8_t* __restrict__ marks, uint32_t nb, int dx, int dy,
                                                                  int dz, int bxn, int byn, uint32_t quads_per_wave, SegTarget st) {
    constexpr uint32_t kWaves = kPackThreads / 64;
    __shared__ uint32_t s_cnt[kWaves], s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = t & 15;
    const uint32_t quads = (nb + 3u) / 4u, waves = gridDim.x * kWaves, first = blockIdx.x * kWaves + (uint32_t)wv;
    auto load_row = [&](uint32_t b, float4* f) {
        const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
        const int y = by * 4 + (r & 3), z = bz * 4 + (r >> 2);
#pragma unroll
        for (int c = 0; c < CH; ++c) f[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b >= nb || y >= dy || z >= dz) return;
        const size_t v = (size_t)(bx * 4) + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z);
        if (VEC) {
#pragma unroll
            for (int c = 0; c < CH; ++c) f[c] = reinterpret_cast<const float4*>(grid + v * CH)[c];
        } else {
            float* ff = reinterpret_cast<float*>(f);
            for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
                for (int c = 0; c < CH; ++c) ff[x * CH + c] = grid[(v + x) * CH + c];
        }
    };
    // lane k: which bricks of the wave's k-th quad (first + k * waves) are listed, one bit each
    uint32_t l4_mine = 0u;
    if (marks) {
        const uint32_t q = first + (uint32_t)lane * waves;
        if ((uint32_t)lane < quads_per_wave && q < quads) {
#pragma unroll
            for (uint32_t g = 0; g < 4u; ++g) { const uint32_t b = q * 4u + g; l4_mine |= (b < nb && marks[b] != 0) ? (1u << g) : 0u; }
        }
    } else {
        for (uint32_t k = 0; k < quads_per_wave; ++k) {   // (uniform per wave; quads beyond the grid load nothing)
            const uint32_t b = (first + k * waves) * 4u + (uint32_t)(lane >> 4);
            float4 f[CH];
            load_row(first + k * waves < quads ? b : nb, f);
            bool nz = false;
#pragma unroll
            for (int c = 0; c < CH; ++c) nz = nz || f[c].x != 0.f || f[c].y != 0.f || f[c].z != 0.f || f[c].w != 0.f;
            const unsigned long long m = __ballot(nz);
            uint32_t bits = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) bits |= ((m >> (16 * g)) & 0xffffull) != 0ull ? (1u << g) : 0u;
            if ((uint32_t)lane == k) l4_mine = bits;
        }
    }
    uint32_t cnt = (uint32_t)__popc(l4_mine);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) s_cnt[wv] = cnt;
    __syncthreads();
    if (t == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < kWaves; ++w) total += s_cnt[w];
        s_base = total ? atomicAdd(&st.ctl[0], total) : 0u;
    }
    __syncthreads();
    uint32_t running = s_base;
    for (int w = 0; w < wv; ++w) running += s_cnt[w];
    if (cnt) {   // (uniform per wave)
        constexpr int kAhead = CH == 1 ? 4 : 2;   // quads whose rows are in flight together
        const uint32_t g = (uint32_t)(lane >> 4);
        for (uint32_t k0 = 0; k0 < quads_per_wave; k0 += kAhead) {
            uint32_t slot[kAhead], brick[kAhead];
            bool on[kAhead];
            float4 f[kAhead][CH];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const uint32_t k = k0 + (uint32_t)u;
                const uint32_t l4 = k < quads_per_wave ? (uint32_t)__shfl(l4_mine, (int)(k & 63u), 64) : 0u;
                brick[u] = (first + k * waves) * 4u + g;
                slot[u] = running + (uint32_t)__popc(l4 & ((1u << g) - 1u));
                on[u] = ((l4 >> g) & 1u) != 0u && slot[u] < st.room;
                running += (uint32_t)__popc(l4);
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) load_row(on[u] ? brick[u] : nb, f[u]);   // (a brick index of nb loads nothing)
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                if (!on[u]) continue;
                unsigned char* p = seg_slot(st.seg, slot[u], CH);
                if (r == 0) *reinterpret_cast<uint4*>(p) = make_uint4(brick[u], 0u, 0u, 0u);
                float4* o = reinterpret_cast<float4*>(p + 16) + (size_t)r * CH;
#pragma unroll
                for (int c = 0; c < CH; ++c) o[c] = f[u][c];
            }
        }
    }
    __syncthreads();
    if (t == 0) seg_finish(st, gridDim.x);
}

// root: the received segments of one ticket (or one segment of an exchange repeated at exact size)
struct RootSegs {
    int n;                                         // senders in this pass
    uint32_t slots[kBricklistMaxRanks];            // slots received from sender i
    uint32_t hdr_cap[kBricklistMaxRanks];          // the capacity its header must carry
    uint32_t table[kBricklistMaxRanks];            // which brick -> slot table is sender i's (its place among the senders)
    int rank[kBricklistMaxRanks];                  // its rank: the mailbox word its count goes to
    unsigned long long off[kBricklistMaxRanks];    // where its segment starts in the receive buffer
};
// bricks of a received segment that this pass adds: none from a header that is not the ticket's, none from a list that outgrew its
// segment (that sender goes again at exact size and is added last)
__device__ inline uint32_t seg_usable(const unsigned char* seg, uint32_t slots, uint32_t hdr_cap, uint32_t ticket, uint32_t* raw) {
    const SegHeader h = *reinterpret_cast<const SegHeader*>(seg);
    const bool sane = h.magic == kSegMagic && h.ticket == ticket && h.capacity == hdr_cap;
    if (raw) *raw = sane ? h.count : 0xffffffffu;
    return sane && h.count <= slots ? h.count : 0u;
}
__device__ inline int sender_of(const RootSegs& S, uint32_t wg, uint32_t per, uint32_t& local) {
    uint32_t first = 0;
    for (int i = 0; i < S.n; ++i) {
        const uint32_t w = (S.slots[i] + per - 1u) / per;
        if (wg < first + w) { local = wg - first; return i; }
        first += w;
    }
    local = 0;
    return -1;
}

// at[q] for a run-time q (the array lives in registers: a select chain instead of scratch memory)
template <int N>
__device__ inline uint32_t at_of(const uint32_t (&at)[N], int q) {
    uint32_t v = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < N; ++k) v = k == q ? at[k] : v;
    return v;
}

// root, launch 1 of 2: every received brick's slot -> its sender's brick -> slot table, and the sender's bit -> the brick's word of `who`
// (which senders list it; all zero between passes: the group that sums a brick clears its word)
__global__ __launch_bounds__(256) void bricklist_index_kernel(const unsigned char* __restrict__ base, RootSegs S, uint32_t ticket, uint32_t nb, int ch,
                                                              uint32_t* __restrict__ slot_of, uint32_t* __restrict__ who) {
    uint32_t local;
    const int i = sender_of(S, blockIdx.x, 256u, local);
    if (i < 0) return;
    const unsigned char* seg = base + S.off[i];
    const uint32_t n = seg_usable(seg, S.slots[i], S.hdr_cap[i], ticket, nullptr);
    const uint32_t s = local * 256u + threadIdx.x;
    if (s >= n) return;
    const uint32_t b = *reinterpret_cast<const uint32_t*>(seg_slot(seg, s, ch));
    if (b >= nb) return;
    slot_of[(size_t)S.table[i] * nb + b] = s;
    atomicOr(&who[b], 1u << i);
}

// root, launch 2 of 2: the sum.  16 lanes per received brick.  who[brick] says which senders list it: the LOWEST-ranked one's group sums the
// brick -- the grid's piece, its own values, then every higher-ranked lister's (their slots from the tables; a slot counts only when it
// carries the brick's id) in rank order -- one read-modify-write of the grid per listed brick, no two groups on the same brick; every other
// group leaves after one look-up.  That group also clears the brick's word for the next pass.  The first workgroup of a sender hands its
// header's count to the host (mailbox word; 0xffffffff = not this ticket's header).
template <int CH, bool VEC>
__global__ __launch_bounds__(256) void bricklist_add_kernel(const unsigned char* __restrict__ base, RootSegs S, uint32_t ticket, uint32_t nb,
                                                            const uint32_t* __restrict__ slot_of, uint32_t* __restrict__ who, int dx, int dy, int dz, int bxn,
                                                            int byn, float* __restrict__ grid, unsigned long long* mailbox, uint32_t n_groups) {
    constexpr int kMax = kBricklistMaxRanks - 1;   // senders
    __shared__ uint32_t s_n[kBricklistMaxRanks];
    if ((int)threadIdx.x < S.n) {
        uint32_t raw;
        s_n[threadIdx.x] = seg_usable(base + S.off[threadIdx.x], S.slots[threadIdx.x], S.hdr_cap[threadIdx.x], ticket, &raw);
        if (blockIdx.x == 0 && mailbox)   // (every sender's count -> the host)
            __hip_atomic_store(mailbox + S.rank[threadIdx.x], ((unsigned long long)ticket << 32) | (unsigned long long)raw, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    // a fixed grid of workgroups strides over the groups of 16 received bricks (the launch runs beside the display GPU's own frame: it should
    // not take the whole device for a latency chain)
    for (uint32_t vb = blockIdx.x; vb < n_groups; vb += gridDim.x) {
    uint32_t local;
    const int i = sender_of(S, vb, 16u, local);
    if (i < 0) continue;
    // round trip 1: this group's brick id and its values
    const uint32_t s = local * 16u + (threadIdx.x >> 4);
    const int r = threadIdx.x & 15;
    if (s >= s_n[i]) continue;
    const unsigned char* mine = seg_slot(base + S.off[i], s, CH);
    const uint32_t b = *reinterpret_cast<const uint32_t*>(mine);
    float4 own[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) own[c] = (reinterpret_cast<const float4*>(mine + 16) + (size_t)r * CH)[c];
    if (b >= nb) continue;
    // round trip 2: who lists the brick; is this the slot its sender's table kept (a brick listed twice by one sender: one slot counts)
    const uint32_t lists = who[b] & ((1u << S.n) - 1u);
    const uint32_t at_own = slot_of[(size_t)S.table[i] * nb + b];
    if (!((lists >> i) & 1u) || at_own != s) continue;
    if (lists & ((1u << i) - 1u)) continue;           // a lower rank lists it: that group sums the brick
    // round trip 3: the grid's piece and the other listers' slots
    const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
    const int y = by * 4 + (r & 3), z = bz * 4 + (r >> 2);
    const bool inside = y < dy && z < dz;
    const size_t v = (size_t)(bx * 4) + (size_t)dx * ((size_t)(inside ? y : 0) + (size_t)dy * (size_t)(inside ? z : 0));
    uint32_t at[kMax];
#pragma unroll
    for (int q = 0; q < kMax; ++q) at[q] = (q > i && ((lists >> q) & 1u)) ? slot_of[(size_t)S.table[q] * nb + b] : 0xffffffffu;
    float4 acc[CH];
    if (VEC) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = reinterpret_cast<const float4*>(grid + v * CH)[c];
    } else {
        float* ff = reinterpret_cast<float*>(acc);
        for (int k = 0; k < 4 * CH; ++k) ff[k] = 0.f;
        if (inside)
            for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
                for (int c = 0; c < CH; ++c) ff[x * CH + c] = grid[(v + x) * CH + c];
    }
    if (r == 0) who[b] = 0u;   // (every other group of this brick leaves whatever it reads here: none of them has the lowest bit)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = make_float4(acc[c].x + own[c].x, acc[c].y + own[c].y, acc[c].z + own[c].z, acc[c].w + own[c].w);
    // round trip 4, only for bricks other senders list too: their values (and their slots' heads: the brick's id, or the slot does not count), in rank order
    constexpr int kAhead = CH == 1 ? 4 : 2;   // senders' values in flight
    for (int q0 = i + 1; q0 < S.n && (lists >> q0) != 0u; q0 += kAhead) {   // (uniform per group of 16 lanes)
        float4 d[kAhead][CH];
        uint32_t head[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int q = q0 + u;
            const uint32_t a = at_of(at, q);
            const bool on = q < S.n && ((lists >> q) & 1u) && a < s_n[q < S.n ? q : 0];
            head[u] = 0xffffffffu;
#pragma unroll
            for (int c = 0; c < CH; ++c) d[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on) {
                const unsigned char* sl = seg_slot(base + S.off[q], a, CH);
                head[u] = *reinterpret_cast<const uint32_t*>(sl);
                const float4* p = reinterpret_cast<const float4*>(sl + 16) + (size_t)r * CH;
#pragma unroll
                for (int c = 0; c < CH; ++c) d[u][c] = p[c];
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (head[u] == b) {   // (a sender that does not list the brick adds nothing)
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = make_float4(acc[c].x + d[u][c].x, acc[c].y + d[u][c].y, acc[c].z + d[u][c].z, acc[c].w + d[u][c].w);
            }
        }
    }
    if (!inside) continue;
    if (VEC) {
#pragma unroll
        for (int c = 0; c < CH; ++c) reinterpret_cast<float4*>(grid + v * CH)[c] = acc[c];
    } else {
        const float* ff = reinterpret_cast<const float*>(acc);
        for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
            for (int c = 0; c < CH; ++c) grid[(v + x) * CH + c] = ff[x * CH + c];
    }
    }
}

// one segment on this device added into a dense grid (cpm_bricklist_segment_to_grid): its first min(count, room) slots
template <int CH, bool VEC>
__global__ __launch_bounds__(256) void segment_to_grid_kernel(const unsigned char* __restrict__ seg, uint32_t room, uint32_t nb, int dx, int dy, int dz, int bxn,
                                                              int byn, float* __restrict__ grid) {
    const SegHeader h = *reinterpret_cast<const SegHeader*>(seg);
    const uint32_t n = h.magic == kSegMagic ? (h.count < room ? h.count : room) : 0u;
    for (uint32_t s = blockIdx.x * 16u + (threadIdx.x >> 4); s < n; s += gridDim.x * 16u) {
        const int r = threadIdx.x & 15;
        const unsigned char* p = seg_slot(seg, s, CH);
        const uint32_t b = *reinterpret_cast<const uint32_t*>(p);
        if (b >= nb) continue;
        const int bx = (int)(b % (uint32_t)bxn), by = (int)((b / (uint32_t)bxn) % (uint32_t)byn), bz = (int)(b / (uint32_t)(bxn * byn));
        const int y = by * 4 + (r & 3), z = bz * 4 + (r >> 2);
        if (y >= dy || z >= dz) continue;
        const size_t v = (size_t)(bx * 4) + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z);
        const float4* q = reinterpret_cast<const float4*>(p + 16) + (size_t)r * CH;
        if (VEC) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float4* g = reinterpret_cast<float4*>(grid + v * CH) + c;
                const float4 a = *g, d = q[c];
                *g = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
            }
        } else {
            const float* ff = reinterpret_cast<const float*>(q);
            for (int x = 0; x < 4 && bx * 4 + x < dx; ++x)
                for (int c = 0; c < CH; ++c) grid[(v + x) * CH + c] += ff[x * CH + c];
        }
    }
}

int bricklist_grow(cpm_ctx* ctx, void** buf, size_t* have, size_t need, hipStream_t busy) {
    if (*have >= need) return CPM_OK;
    if (*buf) {  // (launches that read the old block may still be queued)
        CPM_HIP_CHECK(ctx, hipStreamSynchronize(busy));
        (void)hipFree(*buf);
        *buf = nullptr; *have = 0;
    }
    const size_t bytes = need + need / 4 + 4096;
    if (hipMalloc(buf, bytes) != hipSuccess) { (void)hipGetLastError(); *buf = nullptr; return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_reduce_grid_bricklists", "segment buffer"); }
    *have = bytes;
    return CPM_OK;
}

// rank r's brick count of an opened ticket: a poll of the pinned word its fill launch (a sender's own) or the root's add launch wrote
int bricklist_count(cpm_ctx* ctx, cpm_bricklist_reduce* br, cpm_bricklist_reduce::Slot& sl, int r) {
    if (sl.known[r]) return CPM_OK;
    const volatile unsigned long long* mb = br->mailbox + (sl.ticket % cpm_bricklist_reduce::kSlots) * kBricklistMaxRanks + r;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long v = __atomic_load_n(mb, __ATOMIC_ACQUIRE);
        if ((uint32_t)(v >> 32) == (uint32_t)sl.ticket) { sl.counts[r] = (uint32_t)v; sl.known[r] = true; return CPM_OK; }
        if (synced) return set_error(ctx, CPM_ERR_DEVICE, "cpm_bricklist_reduce", "the brick count of a ticket never arrived (was its segment filled and exchanged?)");
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            CPM_HIP_CHECK(ctx, hipDeviceSynchronize());   // (fill and exchange may have gone to different streams)
            synced = true;
        }
        __builtin_ia32_pause();
    }
}

int bricklist_poisoned(cpm_ctx* ctx, const cpm_bricklist_reduce* br) {
    return br->poisoned ? set_error(ctx, CPM_ERR_DEVICE, "cpm_bricklist_reduce", "an earlier ticket's segment did not carry its header: the ranks' capacities may differ; make a new one") : CPM_OK;
}

SegTarget sender_target(cpm_bricklist_reduce* br, const cpm_bricklist_reduce::Slot& sl) {
    const int slot_i = (int)(sl.ticket % cpm_bricklist_reduce::kSlots);
    SegTarget st;
    st.seg = static_cast<unsigned char*>(br->seg[slot_i]);
    st.capacity = sl.cap[br->comm->rank]; st.room = br->room; st.ticket = (uint32_t)sl.ticket;
    st.ctl = br->ctl + 2 * slot_i;
    st.mailbox = br->mailbox_dev + (size_t)slot_i * kBricklistMaxRanks + br->comm->rank;
    return st;
}

// the root's two launches over the segments `S` names in `base`
struct RootGrid { int dims[3]; int channels, bxn, byn; uint32_t nb; uint32_t* slot_of; uint32_t* who; };
RootGrid root_grid_of(const cpm_bricklist_reduce* br) {
    RootGrid g;
    for (int a = 0; a < 3; ++a) g.dims[a] = br->dims[a];
    g.channels = br->channels; g.bxn = br->bxn; g.byn = br->byn; g.nb = br->nb; g.slot_of = br->slot_of; g.who = br->who;
    return g;
}
int bricklist_root_add(cpm_ctx* ctx, const RootGrid& rg, const RootSegs& S, const void* base, uint32_t ticket, float* grid, unsigned long long* mailbox,
                       hipStream_t s) {
    uint32_t w_index = 0, w_add = 0;
    for (int i = 0; i < S.n; ++i) { w_index += (S.slots[i] + 255u) / 256u; w_add += (S.slots[i] + 15u) / 16u; }
    if (w_add == 0) return CPM_OK;
    const unsigned char* b = static_cast<const unsigned char*>(base);
    CPM_LAUNCH(ctx, bricklist_index_kernel, dim3(w_index), dim3(256), 0, s, b, S, ticket, rg.nb, rg.channels, rg.slot_of, rg.who);
    CPM_LAUNCH_CHECK(ctx, "bricklist_index_kernel");
    const bool vec = (rg.dims[0] & 3) == 0;
    // (CPM_ROOT_ADD_WGS: measurement hook for the launch's footprint beside the root's own frame)
    static const uint32_t add_cap = []() { const char* e = getenv("CPM_ROOT_ADD_WGS"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 0u; }();
    const uint32_t add_wgs = add_cap && add_cap < w_add ? add_cap : w_add;
#define CPM_BL_ADD(CH, VEC) CPM_LAUNCH(ctx, (bricklist_add_kernel<CH, VEC>), dim3(add_wgs), dim3(256), 0, s, b, S, ticket, rg.nb, rg.slot_of, rg.who, rg.dims[0], rg.dims[1], rg.dims[2], rg.bxn, rg.byn, grid, mailbox, w_add)
    if (rg.channels == 1) { if (vec) CPM_BL_ADD(1, true); else CPM_BL_ADD(1, false); }
    else { if (vec) CPM_BL_ADD(4, true); else CPM_BL_ADD(4, false); }
#undef CPM_BL_ADD
    CPM_LAUNCH_CHECK(ctx, "bricklist_add_kernel");
    return CPM_OK;
}

inline uint32_t table_of(const cpm_bricklist_reduce* br, int r) { return (uint32_t)(r < br->root ? r : r - 1); }

int pack_grid_launch(cpm_ctx* ctx, const SegTarget& st, const int dims[3], int channels, const float* grid, const uint8_t* marks, hipStream_t s) {
    const int bxn = div_up(dims[0], 4), byn = div_up(dims[1], 4), bzn = div_up(dims[2], 4);
    const uint32_t nb = (uint32_t)((size_t)bxn * byn * bzn);
    const bool vec = (dims[0] & 3) == 0;
    // a fixed grid: two workgroups of 16 waves per CU (fewer for small grids), every wave 4-brick quads at a stride of all the waves
    const uint32_t quads = (nb + 3u) / 4u, wpw = (uint32_t)kPackThreads / 64u;
    uint32_t wgs = (uint32_t)(2 * ctx->num_cus);
    if ((quads + wpw - 1u) / wpw < wgs) wgs = (quads + wpw - 1u) / wpw;
    if (wgs == 0) wgs = 1;
    if ((quads + wgs * wpw - 1u) / (wgs * wpw) > kPackQuadsPerWave) wgs = (quads + wpw * kPackQuadsPerWave - 1u) / (wpw * kPackQuadsPerWave);   // (lane k holds quad k's marks)
    const uint32_t per_wave = (quads + wgs * wpw - 1u) / (wgs * wpw);
    const dim3 g(wgs);
#define CPM_BL_PACK(CH, VEC) CPM_LAUNCH(ctx, (bricklist_pack_grid_kernel<CH, VEC>), g, dim3(kPackThreads), 0, s, grid, marks, nb, dims[0], dims[1], dims[2], bxn, byn, per_wave, st)
    if (channels == 1) { if (vec) CPM_BL_PACK(1, true); else CPM_BL_PACK(1, false); }
    else { if (vec) CPM_BL_PACK(4, true); else CPM_BL_PACK(4, false); }
#undef CPM_BL_PACK
    CPM_LAUNCH_CHECK(ctx, "bricklist_pack_grid_kernel");
    return CPM_OK;
}

}  // namespace

extern "C" {

uint32_t cpm_bricklist_capacity_for(uint32_t n_bricks, long long previous_count) {
    if (n_bricks == 0) return 0;
    const uint32_t all = round_up_64(n_bricks);
    const uint64_t c = previous_count < 0 ? (uint64_t)n_bricks / 4 : (uint64_t)previous_count + (uint64_t)previous_count / 4 + 64ull;
    const uint32_t cap = round_up_64(c);
    return cap == 0 ? 64u : (cap > all ? all : cap);
}

uint64_t cpm_bricklist_segment_bytes(uint32_t capacity, int channels) { return (uint64_t)seg_size(capacity, channels); }

int cpm_bricklist_reduce_create(cpm_ctx* ctx, cpm_comm* comm, const cpm_grid_desc* gd, int root, cpm_bricklist_reduce** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && gd && out, "cpm_bricklist_reduce_create: null argument");
    CPM_REQUIRE(ctx, gd->channels == 1 || gd->channels == 4, "cpm_bricklist_reduce_create: channels");
    CPM_REQUIRE(ctx, gd->dims[0] >= 1 && gd->dims[1] >= 1 && gd->dims[2] >= 1, "cpm_bricklist_reduce_create: dims");
    CPM_REQUIRE(ctx, (unsigned long long)gd->dims[0] * gd->dims[1] * gd->dims[2] < (1ull << 31), "cpm_bricklist_reduce_create: more than 2^31 cells");
    CPM_REQUIRE(ctx, root >= 0 && root < comm->size, "cpm_bricklist_reduce_create: root");
    CPM_REQUIRE(ctx, comm->size <= kBricklistMaxRanks, "cpm_bricklist_reduce_create: more than 16 ranks");
    *out = nullptr;
    cpm_bricklist_reduce* br = new (std::nothrow) cpm_bricklist_reduce();
    if (!br) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_bricklist_reduce_create", "host allocation failed");
    br->comm = comm; br->device = ctx->device; br->root = root;
    for (int a = 0; a < 3; ++a) br->dims[a] = gd->dims[a];
    br->channels = gd->channels;
    br->bxn = div_up(gd->dims[0], 4); br->byn = div_up(gd->dims[1], 4); br->bzn = div_up(gd->dims[2], 4);
    br->nb = (uint32_t)((size_t)br->bxn * br->byn * br->bzn);
    br->room = round_up_64(br->nb);
    br->cells = (size_t)gd->dims[0] * gd->dims[1] * gd->dims[2];
    const size_t nb = br->nb, words = 2 * (size_t)cpm_bricklist_reduce::kSlots * kBricklistMaxRanks;
    bool ok = hipHostMalloc((void**)&br->mailbox, words * 8, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void**)&br->mailbox_dev, br->mailbox, 0) == hipSuccess;
    if (ok && comm->size > 1) {
        if (comm->rank != root) {
            ok = hipMalloc((void**)&br->ctl, 2 * cpm_bricklist_reduce::kSlots * 4) == hipSuccess &&
                 hipMemset(br->ctl, 0, 2 * cpm_bricklist_reduce::kSlots * 4) == hipSuccess;
        } else {
            const size_t bytes = (size_t)(comm->size - 1) * nb * 4;
            ok = hipMalloc((void**)&br->slot_of, bytes) == hipSuccess && hipMemset(br->slot_of, 0xff, bytes) == hipSuccess &&
                 hipMalloc((void**)&br->who, nb * 4) == hipSuccess && hipMemset(br->who, 0, nb * 4) == hipSuccess;
        }
    }
    if (!ok) {
        (void)hipGetLastError();
        cpm_bricklist_reduce_destroy(br);
        return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_bricklist_reduce_create", "device / pinned allocation failed");
    }
    for (size_t i = 0; i < words; ++i) br->mailbox[i] = 0ull;
    *out = br;
    return CPM_OK;
}

void cpm_bricklist_reduce_destroy(cpm_bricklist_reduce* br) {
    if (!br) return;
    (void)hipSetDevice(br->device);
    for (auto& sl : br->slots) if (sl.ticket && !sl.completed) { (void)hipDeviceSynchronize(); break; }  // the mailbox outlives its writers
    for (void* p : { (void*)br->ctl, (void*)br->slot_of, (void*)br->who, br->again }) if (p) (void)hipFree(p);
    for (void* p : br->seg) if (p) (void)hipFree(p);
    if (br->mailbox) (void)hipHostFree(br->mailbox);
    delete br;
}

uint32_t cpm_bricklist_reduce_bricks(const cpm_bricklist_reduce* br) { return br ? br->nb : 0; }

int cpm_bricklist_reduce_open(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t* ticket_out, cpm_bricklist_segment* seg_out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, br, "cpm_bricklist_reduce_open: null argument");
    int rc = bricklist_poisoned(ctx, br);
    if (rc) return rc;
    constexpr int kSlots = cpm_bricklist_reduce::kSlots;
    const uint64_t ticket = br->next_ticket;
    const int slot_i = (int)(ticket % kSlots);
    cpm_bricklist_reduce::Slot& sl = br->slots[slot_i];
    CPM_REQUIRE(ctx, sl.completed, "cpm_bricklist_reduce_open: 4 tickets opened and not completed (cpm_bricklist_reduce_complete)");
    const int rank = br->comm->rank, size = br->comm->size, root = br->root;
    cpm_bricklist_reduce::Slot fresh;
    fresh.ticket = ticket; fresh.completed = false; fresh.stream = sl.stream;
    if (size > 1) {
        // capacities: from the counts of ticket - 2 (every rank's at the root, its own at a sender), long written when this ticket is opened
        cpm_bricklist_reduce::Slot* old = nullptr;
        if (ticket >= 3 && br->slots[(ticket - 2) % kSlots].ticket == ticket - 2) old = &br->slots[(ticket - 2) % kSlots];
        for (int r = 0; r < size; ++r) {
            if (r == root || (rank != root && r != rank)) continue;
            long long prev = -1;
            if (old) {
                rc = bricklist_count(ctx, br, *old, r);
                if (rc) return rc;
                if (old->counts[r] == 0xffffffffu) { br->poisoned = true; return bricklist_poisoned(ctx, br); }
                prev = (long long)old->counts[r];
            }
            fresh.cap[r] = cpm_bricklist_capacity_for(br->nb, prev);
        }
        size_t need = 0;
        if (rank != root) need = seg_size(br->room, br->channels);
        else for (int r = 0; r < size; ++r) if (r != root) need += (seg_size(fresh.cap[r], br->channels) + 255) & ~(size_t)255;
        rc = bricklist_grow(ctx, &br->seg[slot_i], &br->seg_bytes[slot_i], need, sl.stream);
        if (rc) return rc;
    }
    sl = fresh;
    br->next_ticket = ticket + 1;
    if (ticket_out) *ticket_out = ticket;
    if (seg_out) {
        memset(seg_out, 0, sizeof *seg_out);
        seg_out->ticket = (uint32_t)ticket; seg_out->channels = (uint32_t)br->channels; seg_out->room = br->room;
        if (size > 1 && rank != root) {
            const SegTarget st = sender_target(br, sl);
            seg_out->segment = st.seg; seg_out->capacity = st.capacity; seg_out->control = st.ctl; seg_out->mailbox = st.mailbox;
        }
    }
    return CPM_OK;
}

int cpm_bricklist_pack_grid(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t ticket, const float* grid, const uint8_t* nonzero_bricks, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, br && ticket >= 1 && ticket < br->next_ticket, "cpm_bricklist_pack_grid: no such ticket");
    cpm_bricklist_reduce::Slot& sl = br->slots[ticket % cpm_bricklist_reduce::kSlots];
    CPM_REQUIRE(ctx, sl.ticket == ticket && !sl.completed && !sl.exchanged, "cpm_bricklist_pack_grid: the ticket is not open");
    if (br->comm->size == 1 || br->comm->rank == br->root) return CPM_OK;
    CPM_REQUIRE(ctx, grid, "cpm_bricklist_pack_grid: null grid");
    CPM_REQUIRE_ALIGNED16(ctx, grid, "cpm_bricklist_pack_grid");
    return pack_grid_launch(ctx, sender_target(br, sl), br->dims, br->channels, grid, nonzero_bricks, (hipStream_t)stream);
}

int cpm_debug_root_add_segments(cpm_ctx* ctx, const cpm_bricklist_segment* segs, int n, const cpm_grid_desc* gd, float* grid, uint32_t* slot_of,
                                cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, segs && n >= 1 && n < kBricklistMaxRanks && gd && grid && slot_of, "cpm_debug_root_add_segments: bad argument");
    CPM_REQUIRE(ctx, gd->channels == 1 || gd->channels == 4, "cpm_debug_root_add_segments: channels");
    CPM_REQUIRE_ALIGNED16(ctx, grid, "cpm_debug_root_add_segments");
    RootGrid rg;
    for (int a = 0; a < 3; ++a) rg.dims[a] = gd->dims[a];
    rg.channels = gd->channels; rg.bxn = div_up(gd->dims[0], 4); rg.byn = div_up(gd->dims[1], 4);
    rg.nb = (uint32_t)((size_t)rg.bxn * rg.byn * div_up(gd->dims[2], 4));
    rg.slot_of = slot_of; rg.who = slot_of + (size_t)n * rg.nb;
    RootSegs S;
    memset(&S, 0, sizeof S);
    S.n = n;
    for (int i = 0; i < n; ++i) {   // (segments anywhere on the device: offsets from address 0)
        CPM_REQUIRE(ctx, segs[i].segment && segs[i].ticket == segs[0].ticket && (int)segs[i].channels == gd->channels, "cpm_debug_root_add_segments: segment");
        S.slots[i] = segs[i].capacity; S.hdr_cap[i] = segs[i].capacity; S.table[i] = (uint32_t)i; S.rank[i] = i;
        S.off[i] = (unsigned long long)reinterpret_cast<uintptr_t>(segs[i].segment);
    }
    return bricklist_root_add(ctx, rg, S, nullptr, segs[0].ticket, grid, nullptr, (hipStream_t)stream);
}

int cpm_debug_pack_grid_segment(cpm_ctx* ctx, const cpm_bricklist_segment* seg, const cpm_grid_desc* gd, const float* grid, const uint8_t* nonzero_bricks,
                                cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, seg && seg->segment && seg->control && gd && grid, "cpm_debug_pack_grid_segment: null argument");
    CPM_REQUIRE(ctx, (gd->channels == 1 || gd->channels == 4) && (uint32_t)gd->channels == seg->channels, "cpm_debug_pack_grid_segment: channels");
    CPM_REQUIRE_ALIGNED16(ctx, grid, "cpm_debug_pack_grid_segment");
    SegTarget st;
    st.seg = static_cast<unsigned char*>(seg->segment);
    st.capacity = seg->capacity; st.room = seg->room; st.ticket = seg->ticket; st.ctl = seg->control; st.mailbox = seg->mailbox;
    return pack_grid_launch(ctx, st, gd->dims, gd->channels, grid, nonzero_bricks, (hipStream_t)stream);
}

int cpm_bricklist_reduce_exchange(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t ticket, float* root_grid, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, br && ticket >= 1 && ticket < br->next_ticket, "cpm_bricklist_reduce_exchange: no such ticket");
    int rc = bricklist_poisoned(ctx, br);
    if (rc) return rc;
    constexpr int kSlots = cpm_bricklist_reduce::kSlots;
    const int slot_i = (int)(ticket % kSlots);
    cpm_bricklist_reduce::Slot& sl = br->slots[slot_i];
    CPM_REQUIRE(ctx, sl.ticket == ticket && !sl.completed && !sl.exchanged, "cpm_bricklist_reduce_exchange: the ticket is not open (or was exchanged already)");
    hipStream_t s = (hipStream_t)stream;
    const int rank = br->comm->rank, size = br->comm->size, root = br->root;
    sl.stream = s;
    if (size > 1) {
        const Rccl* R = rccl(ctx);
        if (!R) return CPM_ERR_UNSUPPORTED;
        if (rank != root) {
            const size_t bytes = seg_size(sl.cap[rank], br->channels);
            ProfScope ps(ctx, "rccl_bricklist_send", s);
            CPM_NCCL_CHECK(ctx, R, R->Send(br->seg[slot_i], bytes, ncclInt8, root, br->comm->comm, s));
        } else {
            CPM_REQUIRE(ctx, root_grid, "cpm_bricklist_reduce_exchange: the root needs its grid");
            CPM_REQUIRE_ALIGNED16(ctx, root_grid, "cpm_bricklist_reduce_exchange");
            sl.grid = root_grid;
            unsigned char* base = static_cast<unsigned char*>(br->seg[slot_i]);
            RootSegs S;
            memset(&S, 0, sizeof S);
            {
                ProfScope ps(ctx, "rccl_bricklist_recv", s);
                CPM_NCCL_CHECK(ctx, R, R->GroupStart());
                size_t off = 0;
                for (int r = 0; r < size; ++r) {  // in rank order: a brick several ranks list is summed in that order
                    if (r == root) continue;
                    const size_t bytes = seg_size(sl.cap[r], br->channels);
                    ncclResult_t e = R->Recv(base + off, bytes, ncclInt8, r, br->comm->comm, s);
                    if (e != ncclSuccess) { (void)R->GroupEnd(); return set_error(ctx, CPM_ERR_DEVICE, "ncclRecv", R->GetErrorString(e)); }
                    S.slots[S.n] = sl.cap[r]; S.hdr_cap[S.n] = sl.cap[r]; S.table[S.n] = table_of(br, r); S.rank[S.n] = r; S.off[S.n] = off;
                    ++S.n;
                    off += (bytes + 255) & ~(size_t)255;
                }
                CPM_NCCL_CHECK(ctx, R, R->GroupEnd());
            }
            rc = bricklist_root_add(ctx, root_grid_of(br), S, base, (uint32_t)ticket, root_grid, br->mailbox_dev + (size_t)slot_i * kBricklistMaxRanks, s);
            if (rc) { br->poisoned = true; return rc; }   // (the senders' sends are under way and `who` may hold bits nobody clears: no later ticket)
        }
    }
    sl.exchanged = true;
    return CPM_OK;
}

int cpm_reduce_grid_bricklists(cpm_ctx* ctx, cpm_bricklist_reduce* br, float* grid, const uint8_t* nonzero_bricks, uint64_t* ticket_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, br && grid, "cpm_reduce_grid_bricklists: null argument");
    CPM_REQUIRE_ALIGNED16(ctx, grid, "cpm_reduce_grid_bricklists");
    uint64_t ticket = 0;
    int rc = cpm_bricklist_reduce_open(ctx, br, &ticket, nullptr);
    if (rc) return rc;
    rc = cpm_bricklist_pack_grid(ctx, br, ticket, grid, nonzero_bricks, stream);
    if (!rc) rc = cpm_bricklist_reduce_exchange(ctx, br, ticket, grid, stream);
    if (rc) {  // (nothing of this ticket reached the wire, or the transport failed: the slot is given back either way)
        br->slots[ticket % cpm_bricklist_reduce::kSlots].completed = true;
        return rc;
    }
    if (ticket_out) *ticket_out = ticket;
    return CPM_OK;
}

int cpm_bricklist_reduce_complete(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t ticket, cpm_stream stream, cpm_bricklist_info* info) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, br && ticket >= 1 && ticket < br->next_ticket, "cpm_bricklist_reduce_complete: no such ticket");
    constexpr int kSlots = cpm_bricklist_reduce::kSlots;
    const int slot_i = (int)(ticket % kSlots);
    cpm_bricklist_reduce::Slot& sl = br->slots[slot_i];
    CPM_REQUIRE(ctx, sl.ticket == ticket, "cpm_bricklist_reduce_complete: the ticket is more than 4 calls old");
    int rc = bricklist_poisoned(ctx, br);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int rank = br->comm->rank, size = br->comm->size, root = br->root;
    uint64_t sent = 0, received = 0;
    uint32_t listed = 0;
    CPM_REQUIRE(ctx, sl.completed || sl.exchanged, "cpm_bricklist_reduce_complete: the ticket was opened but never exchanged");
    if (size > 1) {
        const Rccl* R = rccl(ctx);
        if (!R) return CPM_ERR_UNSUPPORTED;
        for (int r = 0; r < size; ++r) {
            if (r == root || (rank != root && r != rank)) continue;
            rc = bricklist_count(ctx, br, sl, r);
            if (rc) return rc;
            const uint32_t n = sl.counts[r];
            if (n == 0xffffffffu) {
                br->poisoned = true;
                return set_error(ctx, CPM_ERR_DEVICE, "cpm_bricklist_reduce_complete", "a received segment did not carry this ticket's header");
            }
            const size_t first_bytes = seg_size(sl.cap[r], br->channels);
            if (rank == root) { received += first_bytes; listed += n; } else sent += first_bytes;
            if (sl.completed || n <= sl.cap[r]) continue;
            // the list had outgrown its segment: this pair exchanges again, at the exact size both now know -- from the sender's same
            // buffer (it has room for every brick: nothing is rebuilt), added at the root after everything else
            const uint32_t exact = round_up_64(n);
            const size_t bytes = seg_size(exact, br->channels);
            if (rank != root) {
                CPM_NCCL_CHECK(ctx, R, R->Send(br->seg[slot_i], bytes, ncclInt8, root, br->comm->comm, s));
                sent += bytes;
            } else {
                rc = bricklist_grow(ctx, &br->again, &br->again_bytes, bytes, s);
                if (rc) return rc;
                CPM_NCCL_CHECK(ctx, R, R->Recv(br->again, bytes, ncclInt8, r, br->comm->comm, s));
                RootSegs S;
                memset(&S, 0, sizeof S);
                S.n = 1; S.slots[0] = exact; S.hdr_cap[0] = sl.cap[r]; S.table[0] = table_of(br, r); S.rank[0] = r; S.off[0] = 0;
                unsigned long long* ack_dev = br->mailbox_dev + (size_t)(kSlots + slot_i) * kBricklistMaxRanks;
                volatile unsigned long long* ack = br->mailbox + (size_t)(kSlots + slot_i) * kBricklistMaxRanks + r;
                rc = bricklist_root_add(ctx, root_grid_of(br), S, br->again, (uint32_t)ticket, sl.grid, ack_dev, s);
                if (rc) { br->poisoned = true; return rc; }
                // (a rare path: wait for it and see that the segment that came was this ticket's and carried the count both sides sized it for)
                CPM_HIP_CHECK(ctx, hipStreamSynchronize(s));
                const unsigned long long v = __atomic_load_n(ack, __ATOMIC_ACQUIRE);
                if ((uint32_t)(v >> 32) != (uint32_t)ticket || (uint32_t)v != n) {
                    br->poisoned = true;
                    return set_error(ctx, CPM_ERR_DEVICE, "cpm_bricklist_reduce_complete", "the segment of a repeated exchange did not carry the expected header: bricks were not added");
                }
                received += bytes;
            }
            ++sl.resent;
        }
    }
    sl.completed = true;
    if (info) {
        info->ticket = ticket; info->n_bricks = br->nb;
        info->n_own = (size > 1 && rank != root) ? sl.counts[rank] : 0u;
        info->capacity = (size > 1 && rank != root) ? sl.cap[rank] : 0u;
        info->resent = sl.resent;
        info->sent_bytes = sent; info->received_bytes = received;
        info->dense_bytes = (uint64_t)br->cells * sizeof(float) * (uint64_t)br->channels;
        info->listed_bricks = listed;
    }
    return CPM_OK;
}

int cpm_bricklist_segment_to_grid(cpm_ctx* ctx, const cpm_bricklist_segment* seg, const cpm_grid_desc* gd, float* grid_out, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, seg && seg->segment && gd && grid_out, "cpm_bricklist_segment_to_grid: null argument");
    CPM_REQUIRE(ctx, (gd->channels == 1 || gd->channels == 4) && (uint32_t)gd->channels == seg->channels, "cpm_bricklist_segment_to_grid: channels");
    CPM_REQUIRE(ctx, gd->dims[0] >= 1 && gd->dims[1] >= 1 && gd->dims[2] >= 1, "cpm_bricklist_segment_to_grid: dims");
    CPM_REQUIRE_ALIGNED16(ctx, grid_out, "cpm_bricklist_segment_to_grid");
    const int bxn = div_up(gd->dims[0], 4), byn = div_up(gd->dims[1], 4), bzn = div_up(gd->dims[2], 4);
    const uint32_t nb = (uint32_t)((size_t)bxn * byn * bzn);
    const bool vec = (gd->dims[0] & 3) == 0;
    const unsigned char* sg = static_cast<const unsigned char*>(seg->segment);
    const dim3 g((unsigned)(seg->room / 16u < 4096u ? (seg->room + 15u) / 16u : 4096u));
    hipStream_t s = (hipStream_t)stream;
#define CPM_SEG2GRID(CH, VEC) CPM_LAUNCH(ctx, (segment_to_grid_kernel<CH, VEC>), g, dim3(256), 0, s, sg, seg->room, nb, gd->dims[0], gd->dims[1], gd->dims[2], bxn, byn, grid_out)
    if (gd->channels == 1) { if (vec) CPM_SEG2GRID(1, true); else CPM_SEG2GRID(1, false); }
    else { if (vec) CPM_SEG2GRID(4, true); else CPM_SEG2GRID(4, false); }
#undef CPM_SEG2GRID
    CPM_LAUNCH_CHECK(ctx, "segment_to_grid_kernel");
    return CPM_OK;
}

int cpm_comm_send(cpm_ctx* ctx, cpm_comm* comm, const void* buf, size_t bytes, int peer, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && (buf || bytes == 0), "cpm_comm_send: null argument");
    CPM_REQUIRE(ctx, peer >= 0 && peer < comm->size && peer != comm->rank, "cpm_comm_send: peer");
    if (bytes == 0) return CPM_OK;
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    ProfScope ps(ctx, "rccl_send", (hipStream_t)stream);
    CPM_NCCL_CHECK(ctx, R, R->Send(buf, bytes, ncclInt8, peer, comm->comm, (hipStream_t)stream));
    return CPM_OK;
}

int cpm_comm_recv(cpm_ctx* ctx, cpm_comm* comm, void* buf, size_t bytes, int peer, cpm_stream stream) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, comm && (buf || bytes == 0), "cpm_comm_recv: null argument");
    CPM_REQUIRE(ctx, peer >= 0 && peer < comm->size && peer != comm->rank, "cpm_comm_recv: peer");
    if (bytes == 0) return CPM_OK;
    const Rccl* R = rccl(ctx);
    if (!R) return CPM_ERR_UNSUPPORTED;
    ProfScope ps(ctx, "rccl_recv", (hipStream_t)stream);
    CPM_NCCL_CHECK(ctx, R, R->Recv(buf, bytes, ncclInt8, peer, comm->comm, (hipStream_t)stream));
    return CPM_OK;
}

}  // extern "C"
