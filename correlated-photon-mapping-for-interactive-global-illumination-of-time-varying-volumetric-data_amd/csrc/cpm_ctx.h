// cpm_ctx.h -- context, error plumbing and device-side descriptors shared by the
// translation units of libcpm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "cpm/cpm_ext.h"
#include "cpm_math.hip.h"

struct cpm_prof_record { const char* name; hipEvent_t a, b; };
struct cpm_prof_entry { std::string name; double total_ms = 0; long calls = 0; };

struct cpm_ctx {
    // per-kernel HIP-event profiling (include/cpm/cpm_profile.h); off by default
    bool profiling = false;
    std::vector<cpm_prof_record> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    std::vector<cpm_prof_entry> prof_entries;
    int device = 0;
    int num_cus = 256;
    size_t lds_per_block = 160 * 1024;  // LDS a workgroup may use on this device (cpm_create asks the device)
    std::string last_error;
    // grow-only scratch arenas (no allocation in steady state)
    void* scratch[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t scratch_bytes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    // device, 8 floats: (theta, phi) of the last directional-light emission, decodeDirection of it, encodeDirection of that,
    // valid flag -- a pure function of (theta, phi), evaluated once by the emitter so that the tracer's workgroups need not
    // (cpm_trace.hip; used only for samples whose (theta, phi) bit patterns match)
    float* dir_hint = nullptr;
    // dir_hint + 8: the same 8 floats for the directional light of cpm_trace_emitted (which has no emitter launch to
    // leave them); written by a one-lane launch whenever the direction differs from the one they were made for
    float emit_hint_for[3] = { 0.f, 0.f, 0.f };
    bool emit_hint_valid = false;
    // test / measurement hooks (include/cpm/cpm_profile.h: cpm_debug_*): per context, so that one context's experiment never
    // changes what another context (another GPU, a concurrent frame) runs
    struct {
        unsigned long long* step_counter = nullptr;   // cpm_trace adds its Woodcock iteration counts
        unsigned long long* gather_stamps = nullptr;  // cpm_gather writes per-brick wave stamps
        int bin_fused = 1, gather_coop = 1, gather_force_voxel = 0;
        int sort_mode = 0, sort_items = 0;
        int brick_streaming = 1, select_partition = 1;
        int stream_wg_per_cu = -1;
        int fail_next_select = 0;
    } dbg;
    struct cpm_trace_order* trace_order = nullptr;  // cpm_trace_set_order (not owned)
    bool trace_order_measure = false;               // ... and whether the launches add their costs to it
    size_t fast_hist_words = 0;  // cpm_bin_fast: size of the histograms whose all-zero state is established (0 = none)
    int fast_hist_parity = 0;    // which of the two (histogram, cursors) pairs the next cpm_bin_fast uses
    const void* fast_last_table = nullptr;  // the last cpm_bin_fast's table and radius: cpm_gather_fast refuses another radius for it
    float fast_last_radius = 0.f;
    // cpm_set_photon_layout: how the photon-record buffers of N * I records handed to this context are laid out (CPM_PHOTONS_*) ...
    int photon_layout = 0;
    // ... unless the buffer has been described itself (cpm_records_describe): base -> layout and its N * I
    struct RecordBuffer { const void* base; int layout; size_t n_records; };
    std::vector<RecordBuffer> record_buffers;
    // cpm_tf_update from host memory: the LUT goes through a ring of pinned host slots the upload kernel reads directly --
    // no staged copy ahead of the kernel, no wait for it behind (a slot is reused only after the launch that read it)
    static constexpr int kTfStageSlots = 4;
    float* tf_stage = nullptr;          // pinned host, kTfStageSlots x tf_stage_floats
    float* tf_stage_dev = nullptr;      // its device address
    size_t tf_stage_floats = 0;
    hipEvent_t tf_stage_done[kTfStageSlots] = { nullptr, nullptr, nullptr, nullptr };
    int tf_stage_next = 0;
};

// scratch slots
enum {
    CPM_SCR_SORT_KEYS = 0,   // ping-pong keys
    CPM_SCR_SORT_VALS = 1,   // ping-pong values
    CPM_SCR_SORT_HIST = 2,   // per-tile digit histograms
    CPM_SCR_BIN_KEYS = 3,    // cell keys of cpm_bin
    CPM_SCR_SMALL = 4,       // TF points etc.
    CPM_SCR_MISC = 5,
    CPM_SCR_FAST_BIN = 6,    // cpm_bin_fast: two (brick histogram, cursors) pairs, used in turn, zeroed by the call before
    CPM_SCR_FAST_STAGE = 7   // cpm_gather_fast, wide boxes: one 64-bit tile with its halo per brick
};

struct cpm_volume {
    cpm_volume_desc desc;
    void* voxels = nullptr;  // device, x fastest, padded by 16 bytes
    // the tracer's copy: element (x, y, z) = { v(x, y, z), v(x, y', z), v(x, y, z'), v(x, y', z') } with y' = min(y + 1, dim.y - 1),
    // z' likewise, x fastest -- the 2 x 2 x 2 footprint of a trilinear fetch is ONE load of two neighbouring elements
    // (cpm_trace.hip); 4 * bytes + 64, rebuilt by everything that writes `voxels` (cpm_volume_update, cpm_volume_mix)
    void* quads = nullptr;
    // cpm_volume_mix leaves the copy to whoever needs it: a trace over all the samples rebuilds it first (cpm::trace_volume_source),
    // re-traces of selected photons read the linear block instead (four fetches per sample instead of one, for a few thousand
    // photons) -- a time step of a played sequence that is served by a correlated update never pays the 17 us re-layout
    mutable bool quads_stale = false;
    size_t bytes = 0;
};

struct cpm_tf {
    int width = 0;
    float* rgba = nullptr;   // device, width * 4
    float* alpha = nullptr;  // device, width (what the tracer stages into LDS)
};

// cpm_trace_order_*: the order in which a trace launch's workgroups take the 256-sample chunks, and what the chunks cost
struct cpm_trace_order {
    int n_light_samples = 0;
    uint32_t n_chunks = 0;
    uint32_t* order = nullptr;  // device, n_chunks: workgroup b takes chunk order[b]
    uint32_t* cost = nullptr;   // device, 4 n_chunks + 1: per chunk and wave the wave's longest walk in the last measured launch; [4 n_chunks] = launches measured
};

namespace cpm {

int set_error(cpm_ctx* ctx, int status, const char* what, const char* detail);
void* scratch(cpm_ctx* ctx, int slot, size_t bytes);  // nullptr on failure (error set)
bool affine_from_matrix(const float m[16], Affine& out);
// rebuild vol->quads from `src` (a device block laid out like vol->voxels; vol->voxels itself, or the source of a
// device->device update, which is then also copied into vol->voxels by the same launch)
int build_quads(cpm_ctx* ctx, cpm_volume* vol, const void* src, bool copy_linear, hipStream_t stream);

#define CPM_HIP_CHECK(ctx, expr)                                                        \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return cpm::set_error((ctx), CPM_ERR_DEVICE, #expr, hipGetErrorString(e_)); \
    } while (0)

#define CPM_LAUNCH_CHECK(ctx, name)                                                     \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return cpm::set_error((ctx), CPM_ERR_DEVICE, name, hipGetErrorString(e_)); \
    } while (0)

// Every entry point: a null context is an argument error; the context's device becomes the calling thread's current
// device (a host thread that drives several contexts -- one per GPU -- may call them in any order).
#define CPM_ENTER(ctx)                                                                  \
    do {                                                                                \
        if (!(ctx)) return CPM_ERR_INVALID_ARGUMENT;                                    \
        int cur_ = -1;                                                                  \
        if (hipGetDevice(&cur_) != hipSuccess || cur_ != (ctx)->device) {               \
            hipError_t e_ = hipSetDevice((ctx)->device);                                \
            if (e_ != hipSuccess) return cpm::set_error((ctx), CPM_ERR_DEVICE, "hipSetDevice", hipGetErrorString(e_)); \
        }                                                                               \
    } while (0)

#define CPM_REQUIRE(ctx, cond, msg)                                                     \
    do {                                                                                \
        if (!(cond)) return cpm::set_error((ctx), CPM_ERR_INVALID_ARGUMENT, msg, #cond); \
    } while (0)

// float8 / float4 buffers are read and written with 16-byte accesses (include/cpm/cpm.h, conventions)
#define CPM_REQUIRE_ALIGNED16(ctx, ptr, msg) \
    CPM_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0, msg ": buffer is not 16-byte aligned")

// Records a start/stop hipEvent pair around one kernel launch when ctx->profiling is set.
struct ProfScope {
    cpm_ctx* ctx; hipStream_t s; hipEvent_t b = nullptr;
    ProfScope(cpm_ctx* c, const char* name, hipStream_t stream);
    ~ProfScope();
};

#define CPM_LAUNCH(ctx, kernel, grid, block, lds, stream, ...)                          \
    do {                                                                                \
        cpm::ProfScope cpm_ps_((ctx), #kernel, (stream));                               \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);              \
    } while (0)

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Where the halves of photon record j lie in a record buffer (include/cpm/cpm.h, CPM_PHOTONS_*): half A = (x, y, z, powerR) at float4
// index stride * j, half B = (powerG, powerB, theta, phi) `b` float4s behind it.  float8 records: {2, 1}; two planes of n records: {1, n}.
struct RecLayout { uint32_t stride, b; };
inline RecLayout rec_interleaved() { return RecLayout{ 2u, 1u }; }
// ... of a buffer of n_records records in the context's layout (cpm_set_photon_layout)
inline RecLayout rec_layout(const cpm_ctx* ctx, size_t n_records) {
    return ctx->photon_layout == 1 /* CPM_PHOTONS_PLANAR */ ? RecLayout{ 1u, (uint32_t)n_records } : rec_interleaved();
}
// ... of the buffer starting at `base`: as it was described (cpm_records_describe: its own N * I is the distance between the planes), else the
// context's layout with the call's count
inline RecLayout rec_layout(const cpm_ctx* ctx, const void* base, size_t n_records) {
    for (const auto& b : ctx->record_buffers)
        if (b.base == base) return b.layout == 1 ? RecLayout{ 1u, (uint32_t)b.n_records } : rec_interleaved();
    return rec_layout(ctx, n_records);
}
// the described layout of `base` (CPM_PHOTONS_*), or -1 when the buffer was not described
inline int described_layout(const cpm_ctx* ctx, const void* base, size_t* n_records) {
    for (const auto& b : ctx->record_buffers)
        if (b.base == base) { if (n_records) *n_records = b.n_records; return b.layout; }
    return -1;
}
__host__ __device__ inline const float4* rec_at(const float* base, RecLayout R, size_t j) { return reinterpret_cast<const float4*>(base) + (size_t)R.stride * j; }
__host__ __device__ inline float4* rec_at(float* base, RecLayout R, size_t j) { return reinterpret_cast<float4*>(base) + (size_t)R.stride * j; }

// what the last scatter pass of cpm_bin's sort also produces (cpm_sort.hip, radix_scatter_kernel<..., BINSINK>)
struct BinSink {
    const float* photons = nullptr;   // photon records ...
    RecLayout rec = { 2u, 1u };       // ... and where their halves lie
    int channels = 1;
    uint32_t* order = nullptr;        // sorted photon indices
    float* sorted = nullptr;          // compact (pos, power) records in cell order
    uint32_t* cell_start = nullptr;   // run-start table, preset to 0xffffffff
};

// keys/vals sorted in place (or left in *res_keys / *res_vals without the copy-back when those are given);
// sink (nullable): see BinSink; *sink_done tells whether the last pass consumed it (then *res_vals is not written)
// first_hist_done: pass 0's per-tile digit histogram is already in sort_first_hist()'s table
int radix_sort(cpm_ctx* ctx, uint32_t* keys, uint32_t* vals, size_t n, int key_bits, hipStream_t s,
               uint32_t** res_keys, uint32_t** res_vals, const BinSink* sink, bool* sink_done, bool first_hist_done);
int sort_items_for(const cpm_ctx* ctx, size_t n);
uint32_t* sort_first_hist(cpm_ctx* ctx, size_t n, int key_bits, uint32_t* num_tiles_out);
const uint32_t* sort_last_digit_totals(cpm_ctx* ctx, size_t n);

// device-side grid description
struct GridDev {
    int dx, dy, dz, channels;
    Affine t2i, i2t;
};


// ---- brick-list segments (cpm_reduce_grid_bricklists; written by cpm_comm.hip's pack launch and by cpm_gather_fast_segment) -------------
// A segment = [ SegHeader, 16 bytes ][ slot 0 ][ slot 1 ] ... ; a slot = [ brick id, 0, 0, 0 ][ 64 * channels floats ]: one non-zero
// 4 x 4 x 4-voxel brick of a rank's light volume, values in (z, y, x, channel) order.  Slots are handed out by ONE device counter per
// launch (any order: the root adds by brick id, a sender lists a brick once); the launch's last workgroup writes the header and the
// sender's pinned mailbox word and leaves the counter words zero for the next launch.  The first `capacity` slots travel.
constexpr uint32_t kSegMagic = 0x62726b32u;  // "brk2"
struct SegHeader { uint32_t count, capacity, ticket, magic; };
__host__ __device__ inline size_t seg_slot_bytes(int channels) { return 16u + 256u * (size_t)channels; }
__host__ __device__ inline size_t seg_size(uint32_t capacity, int channels) { return sizeof(SegHeader) + (size_t)capacity * seg_slot_bytes(channels); }
struct SegTarget {
    unsigned char* seg = nullptr;          // device: header + `room` slots
    uint32_t capacity = 0, room = 0, ticket = 0;
    uint32_t* ctl = nullptr;               // device, 2 words: slots handed out, workgroups done (zero between launches)
    unsigned long long* mailbox = nullptr; // device address of the pinned word: ticket << 32 | count
};
// the end of a launch that fills a segment: every workgroup's thread 0 calls this after its last slot; the last one publishes the count
// No fence: a workgroup's slot atomics are RETURNING atomics whose results it has used -- they are performed before its done atomic is issued --,
// and the slots' bytes need not be visible before the launch ends.  (A device-scope fence here is an L2 write-back per workgroup: it cost the pack
// launch 20 of its 32 us.)
__device__ inline void seg_finish(const SegTarget& st, uint32_t n_workgroups) {
    const uint32_t d = atomicAdd(&st.ctl[1], 1u);
    if (d + 1u != n_workgroups) return;
    const uint32_t n = atomicExch(&st.ctl[0], 0u);
    atomicExch(&st.ctl[1], 0u);
    *reinterpret_cast<SegHeader*>(st.seg) = SegHeader{ n, st.capacity, st.ticket, kSegMagic };
    if (st.mailbox)
        __hip_atomic_store(st.mailbox, ((unsigned long long)st.ticket << 32) | (unsigned long long)n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace cpm
