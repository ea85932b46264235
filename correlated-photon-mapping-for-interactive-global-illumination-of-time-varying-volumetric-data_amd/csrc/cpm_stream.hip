// cpm_stream.hip -- a time-varying sequence whose steps live in HOST memory: the upload inside the step ("per step: upload
// volume, min/max, mean-abs-diff, ..."), hidden behind the step before it.
//
// The reference steps a host-side sequence: VolumeSequencePlayer::process picks two elements of a std::vector<std::shared_ptr<Volume>> and
// OpenGL uploads whichever is not resident yet (ref uniformgridcl/processors/volumesequenceplayer.cpp:94-124); the difference analysis walks
// the same elements on the CPU (ref uniformgridcl/processors/dynamicvolumedifferenceanalysis.h:96-151).  A sequence that does not stay on
// the device -- or has not been through its first loop -- pays 16 MiB of PCIe per 256^3 step: ~0.33 ms against a 0.07 ms correlated update.
//
// Here: a small ring of device volumes (linear block + the tracer's footprint copy each) over the host sequence, filled by a copy stream
// the library owns.  cpm_volume_stream_prefetch(tag, host voxels) enqueues, on that stream: wait until the consumer's work enqueued so far
// has drained (the slot it overwrites was last used there), H2D copy, footprint re-layout, an event; it returns at once.
// cpm_volume_stream_acquire(tag) makes the caller's stream wait for that event and hands the volume out.  A caller that prefetches step
// t + 1 before it acquires step t runs step t's min/max -> difference -> importance -> re-trace -> delta splat while step t + 1 crosses
// PCIe: the step costs max(upload, update), not their sum.  The host buffers should be pinned (cpm_pinned_alloc; hipMemcpyAsync from
// pageable memory is staged and blocks the host).
#include <new>

#include "cpm_ctx.h"

using namespace cpm;

struct cpm_volume_stream {
    static constexpr int kMaxSlots = 8;
    cpm_volume_desc desc;
    int device = 0, n_slots = 0;
    hipStream_t copy = nullptr;
    uint64_t clock = 0;
    struct Slot {
        cpm_volume* vol = nullptr;
        uint64_t tag = 0, last_use = 0;
        bool valid = false, timed = false;
        hipEvent_t ready = nullptr, fence = nullptr, t0 = nullptr, t1 = nullptr;
    } slots[kMaxSlots];
    // figures (cpm_volume_stream_stats)
    uint64_t uploads = 0, hits = 0, inline_uploads = 0, bytes = 0;
    double upload_ms = 0.0;   // H2D alone, from the events of the uploads that have finished
    uint64_t uploads_timed = 0;
};

namespace {

void harvest(cpm_volume_stream* vs, cpm_volume_stream::Slot& sl) {
    if (!sl.timed || hipEventQuery(sl.t1) != hipSuccess) { (void)hipGetLastError(); return; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sl.t0, sl.t1) == hipSuccess) { vs->upload_ms += ms; ++vs->uploads_timed; }
    (void)hipGetLastError();
    sl.timed = false;
}

int upload(cpm_ctx* ctx, cpm_volume_stream* vs, uint64_t tag, const void* host, hipStream_t consumer, cpm_volume_stream::Slot** out) {
    // the slot handed out longest ago (an empty one first)
    cpm_volume_stream::Slot* victim = nullptr;
    for (int i = 0; i < vs->n_slots; ++i) {
        cpm_volume_stream::Slot& s = vs->slots[i];
        if (!s.valid) { victim = &s; break; }
        if (!victim || s.last_use < victim->last_use) victim = &s;
    }
    harvest(vs, *victim);
    if (victim->timed) {   // (its last upload's events are still pending: they are recorded again below, so read them now)
        CPM_HIP_CHECK(ctx, hipEventSynchronize(victim->t1));
        harvest(vs, *victim);
    }
    cpm_volume* v = victim->vol;
    // everything the consumer has enqueued so far may read the slot's old contents
    CPM_HIP_CHECK(ctx, hipEventRecord(victim->fence, consumer));
    CPM_HIP_CHECK(ctx, hipStreamWaitEvent(vs->copy, victim->fence, 0));
    CPM_HIP_CHECK(ctx, hipEventRecord(victim->t0, vs->copy));
    CPM_HIP_CHECK(ctx, hipMemcpyAsync(v->voxels, host, v->bytes, hipMemcpyHostToDevice, vs->copy));
    CPM_HIP_CHECK(ctx, hipEventRecord(victim->t1, vs->copy));
    int rc = build_quads(ctx, v, v->voxels, false, vs->copy);
    if (rc) { victim->valid = false; return rc; }
    CPM_HIP_CHECK(ctx, hipEventRecord(victim->ready, vs->copy));
    victim->tag = tag; victim->valid = true; victim->timed = true;
    victim->last_use = ++vs->clock;
    ++vs->uploads;
    vs->bytes += v->bytes;
    *out = victim;
    return CPM_OK;
}

cpm_volume_stream::Slot* find(cpm_volume_stream* vs, uint64_t tag) {
    for (int i = 0; i < vs->n_slots; ++i) if (vs->slots[i].valid && vs->slots[i].tag == tag) return &vs->slots[i];
    return nullptr;
}

}  // namespace

extern "C" {

int cpm_pinned_alloc(cpm_ctx* ctx, size_t bytes, void** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, out && bytes > 0, "cpm_pinned_alloc: bad argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "hipHostMalloc", hipGetErrorString(e)); }
    return CPM_OK;
}

void cpm_pinned_free(cpm_ctx* ctx, void* p) {
    (void)ctx;
    if (p) (void)hipHostFree(p);
}

int cpm_volume_stream_create(cpm_ctx* ctx, const cpm_volume_desc* desc, int n_slots, cpm_volume_stream** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, desc && out, "cpm_volume_stream_create: null argument");
    CPM_REQUIRE(ctx, n_slots >= 2 && n_slots <= cpm_volume_stream::kMaxSlots, "cpm_volume_stream_create: 2 to 8 slots");
    *out = nullptr;
    cpm_volume_stream* vs = new (std::nothrow) cpm_volume_stream();
    if (!vs) return set_error(ctx, CPM_ERR_OUT_OF_MEMORY, "cpm_volume_stream_create", "host allocation failed");
    vs->desc = *desc; vs->device = ctx->device; vs->n_slots = n_slots;
    bool ok = hipStreamCreateWithFlags(&vs->copy, hipStreamNonBlocking) == hipSuccess;
    int rc = CPM_OK;
    for (int i = 0; ok && i < n_slots; ++i) {
        cpm_volume_stream::Slot& s = vs->slots[i];
        rc = cpm_volume_create(ctx, desc, nullptr, 0, (cpm_stream)vs->copy, &s.vol);
        if (rc) break;
        ok = hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.fence, hipEventDisableTiming) == hipSuccess &&
             hipEventCreate(&s.t0) == hipSuccess && hipEventCreate(&s.t1) == hipSuccess;
    }
    if (rc || !ok) {
        (void)hipGetLastError();
        cpm_volume_stream_destroy(ctx, vs);
        return rc ? rc : set_error(ctx, CPM_ERR_DEVICE, "cpm_volume_stream_create", "stream / event creation failed");
    }
    *out = vs;
    return CPM_OK;
}

void cpm_volume_stream_destroy(cpm_ctx* ctx, cpm_volume_stream* vs) {
    if (!vs) return;
    (void)hipSetDevice(vs->device);
    if (vs->copy) (void)hipStreamSynchronize(vs->copy);
    for (auto& s : vs->slots) {
        for (hipEvent_t e : { s.ready, s.fence, s.t0, s.t1 }) if (e) (void)hipEventDestroy(e);
        if (s.vol) cpm_volume_destroy(ctx, s.vol);
    }
    if (vs->copy) (void)hipStreamDestroy(vs->copy);
    delete vs;
}

int cpm_volume_stream_prefetch(cpm_ctx* ctx, cpm_volume_stream* vs, uint64_t tag, const void* host_voxels, cpm_stream consumer) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, vs && host_voxels, "cpm_volume_stream_prefetch: null argument");
    cpm_volume_stream::Slot* s = find(vs, tag);
    if (s) { s->last_use = ++vs->clock; return CPM_OK; }   // (about to be used: not the next upload's victim)
    return upload(ctx, vs, tag, host_voxels, (hipStream_t)consumer, &s);
}

int cpm_volume_stream_acquire(cpm_ctx* ctx, cpm_volume_stream* vs, uint64_t tag, const void* host_voxels, cpm_stream consumer, cpm_volume** out) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, vs && out, "cpm_volume_stream_acquire: null argument");
    *out = nullptr;
    cpm_volume_stream::Slot* s = find(vs, tag);
    if (s) ++vs->hits;
    else {
        CPM_REQUIRE(ctx, host_voxels, "cpm_volume_stream_acquire: the step is not resident and no host voxels were given");
        int rc = upload(ctx, vs, tag, host_voxels, (hipStream_t)consumer, &s);
        if (rc) return rc;
        ++vs->inline_uploads;
    }
    CPM_HIP_CHECK(ctx, hipStreamWaitEvent((hipStream_t)consumer, s->ready, 0));
    s->last_use = ++vs->clock;
    *out = s->vol;
    return CPM_OK;
}

int cpm_volume_stream_stats(cpm_ctx* ctx, cpm_volume_stream* vs, cpm_volume_stream_info* info) {
    CPM_ENTER(ctx);
    CPM_REQUIRE(ctx, vs && info, "cpm_volume_stream_stats: null argument");
    for (int i = 0; i < vs->n_slots; ++i) harvest(vs, vs->slots[i]);
    info->uploads = vs->uploads; info->hits = vs->hits; info->uploads_at_acquire = vs->inline_uploads; info->bytes_uploaded = vs->bytes;
    info->uploads_timed = vs->uploads_timed; info->upload_ms_total = vs->upload_ms;
    info->bytes_per_step = vs->n_slots ? vs->slots[0].vol->bytes : 0;
    return CPM_OK;
}

}  // extern "C"
