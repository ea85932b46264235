// cpm_host_c.cpp -- a plain-C facade over the host layer: builds the processor network of the
// CorrelatedPhotonMappingSingleVolume workspace (sample generator -> directional light sampler ->
// photon tracer -> photon-to-light-volume, plus min/max -> importance -> tracer) and evaluates it,
// so that tests can drive the C++ surface through ctypes.  Connections follow
// workspaces/CorrelatedPhotonMappingSingleVolume.inv:1178-1271.
#include <vector>
#include <chrono>
#include <cstring>
#include <sstream>

#include <cpm/cpm_profile.h>

#include "cpm_modules.h"

using namespace inviwo;

struct cpmh_network {
    std::shared_ptr<Volume> volume;
    DataOutport<Volume> volumeSource{ "data" };
    DataOutport<Mesh> proxyGeometry{ "proxyGeometry" };
    DataOutport<DirectionalLight> lightSource{ "LightSource" };
    UniformSampleGenerator2DProcessorCL sampleGenerator;
    DirectionalLightSamplerCLProcessor lightSampler;
    VolumeMinMaxCLProcessor minMax;
    MinMaxUniformGrid3DImportanceCLProcessor importance;
    ProgressivePhotonTracerCL tracer;
    PhotonToLightVolumeProcessorCL lightVolume;
    TransferFunction tf;
    bool correlated = false;
    cpm_comm* comm = nullptr;  // cpmh_enable_shard_reduce
    // further lights of the workspace ("Directional light source 2" -> "Directional light sampler 2" -> the tracer's LightSamples
    // multi-inport, workspaces/CorrelatedPhotonMappingSingleVolume.inv:1068-1170,1255-1270): they share the sample generator
    struct ExtraLight {
        DataOutport<DirectionalLight> source{ "LightSource" };
        DirectionalLightSamplerCLProcessor sampler;
    };
    std::vector<std::unique_ptr<ExtraLight>> extraLights;
    ~cpmh_network() { if (comm) cpm_comm_destroy(comm); }
};

static TransferFunction make_tf(const float* p5, int n) {
    TransferFunction tf;
    for (int i = 0; i < n; ++i) tf.add((double)p5[5 * i], vec4(p5[5 * i + 1], p5[5 * i + 2], p5[5 * i + 3], p5[5 * i + 4]));
    return tf;
}

extern "C" {

cpmh_network* cpmh_create(const void* voxels, int dtype, int dx, int dy, int dz, int nx, int ny, const float light_position[3],
                          const float light_direction[3], const float* tf_points5, int n_points, int volume_size_option,
                          int max_scattering, int correlated) {
    if (!CpmRuntime::get().valid()) return nullptr;
    auto* net = new cpmh_network();
    net->volume = std::make_shared<Volume>(size3_t{ (size_t)dx, (size_t)dy, (size_t)dz }, dtype);
    net->volume->ramBytes.assign((const uint8_t*)voxels, (const uint8_t*)voxels + (size_t)dx * dy * dz * net->volume->elementSize());
    net->volumeSource.setData(net->volume);
    net->proxyGeometry.setData(Mesh::unitCube());
    auto light = std::make_shared<DirectionalLight>();
    light->position = vec3(light_position[0], light_position[1], light_position[2]);
    light->direction = vec3(light_direction[0], light_direction[1], light_direction[2]);
    net->lightSource.setData(light);
    net->tf = make_tf(tf_points5, n_points);
    net->correlated = correlated != 0;
    // connections (workspace :1178-1271)
    net->sampleGenerator.nSamplesProp_.set(ivec2{ nx, ny });
    net->lightSampler.boundingVolumePort_.connectTo(&net->proxyGeometry);
    net->lightSampler.samplesPort_.connectTo(&net->sampleGenerator.samplesPort_);
    net->lightSampler.lightsPort_.connectTo(&net->lightSource);
    net->tracer.volumePort_.connectTo(&net->volumeSource);
    net->tracer.lightSamples_.connectTo(&net->lightSampler.lightSamplesPort_);
    net->tracer.maxScatteringEvents_.set(max_scattering);
    net->tracer.transferFunction_ = net->tf;
    net->tracer.clipX_.set(ivec2{ 0, dx }); net->tracer.clipY_.set(ivec2{ 0, dy }); net->tracer.clipZ_.set(ivec2{ 0, dz });
    if (net->correlated) {
        net->minMax.inport_.connectTo(&net->volumeSource);
        net->importance.minMaxUniformGrid3DInport_.connectTo(&net->minMax.outport_);
        net->importance.setTransferFunction(net->tf);
        net->tracer.recomputationImportanceGrid_.connectTo(&net->importance.importanceUniformGrid3DOutport_);
        net->lightVolume.recomputedPhotonIndicesPort_.connectTo(&net->tracer.recomputedIndicesPort_);
    }
    net->lightVolume.volumeInport_.connectTo(&net->volumeSource);
    net->lightVolume.photons_.connectTo(&net->tracer.outport_);
    net->lightVolume.volumeSizeOption_.set(volume_size_option);
    return net;
}

void cpmh_destroy(cpmh_network* net) { delete net; }

// One more directional light feeding the tracer's multi-inport (one tracer launch per light, photonOffset = the photons of the
// lights before it: progressivephotontracercl.cpp:481-527,543-549).  Before the first evaluation.  Returns the light's index.
int cpmh_add_light(cpmh_network* net, const float light_position[3], const float light_direction[3]) {
    if (!net) return -1;
    auto e = std::make_unique<cpmh_network::ExtraLight>();
    auto light = std::make_shared<DirectionalLight>();
    light->position = vec3(light_position[0], light_position[1], light_position[2]);
    light->direction = vec3(light_direction[0], light_direction[1], light_direction[2]);
    e->source.setData(light);
    e->sampler.boundingVolumePort_.connectTo(&net->proxyGeometry);
    e->sampler.samplesPort_.connectTo(&net->sampleGenerator.samplesPort_);
    e->sampler.lightsPort_.connectTo(&e->source);
    net->tracer.lightSamples_.connectTo(&e->sampler.lightSamplesPort_);
    net->extraLights.push_back(std::move(e));
    return (int)net->extraLights.size();
}
int cpmh_n_lights(cpmh_network* net) { return net ? 1 + (int)net->extraLights.size() : 0; }
// the clip ranges of CubeProxyGeometry, to which the tracer's clip properties are linked (workspace :393-420,740-757: clipX
// 73..512 etc.), in voxels: the proxy mesh the light samplers intersect AND the tracer's box
void cpmh_set_clip(cpmh_network* net, int x0, int x1, int y0, int y1, int z0, int z1) {
    const size3_t d = net->volume->getDimensions();
    net->proxyGeometry.setData(Mesh::box(vec3((float)x0 / (float)d.x, (float)y0 / (float)d.y, (float)z0 / (float)d.z),
                                         vec3((float)x1 / (float)d.x, (float)y1 / (float)d.y, (float)z1 / (float)d.z)));
    net->tracer.clipX_.set(ivec2{ x0, x1 }); net->tracer.clipY_.set(ivec2{ y0, y1 }); net->tracer.clipZ_.set(ivec2{ z0, z1 });
}
// test hook (cpm_profile.h): the next select / retrace launch of the tracer's importance branch fails after appending its tiles
void cpmh_debug_fail_next_select(cpmh_network*) { cpm_debug_fail_next_select(CpmRuntime::get().ctx(), 1); }

// Evaluate the network once, upstream first (what Inviwo's evaluator does on invalidation).
// first != 0: everything (light samples included); else only what a TF edit / timer tick invalidates.
int cpmh_evaluate(cpmh_network* net, int first) {
    if (!net) return -1;
    if (first) {
        net->sampleGenerator.process();
        net->lightSampler.process();
        for (auto& e : net->extraLights) e->sampler.process();
        if (net->correlated) net->minMax.process();
    }
    if (net->correlated) net->importance.process();
    net->tracer.process();
    net->lightVolume.process();
    return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
}

// A transfer-function edit: the linked TF properties of tracer, importance processor (and raycaster).
void cpmh_set_transfer_function(cpmh_network* net, const float* tf_points5, int n_points) {
    net->tf = make_tf(tf_points5, n_points);
    net->tracer.setTransferFunction(net->tf);
    if (net->correlated) net->importance.setTransferFunction(net->tf);
}

int cpmh_set_property_float(cpmh_network* net, const char* processor, const char* id, float value) {
    Processor* p = !strcmp(processor, "tracer") ? (Processor*)&net->tracer : !strcmp(processor, "lightvolume") ? (Processor*)&net->lightVolume : nullptr;
    if (!p) return -1;
    if (auto* f = dynamic_cast<FloatProperty*>(p->getPropertyByIdentifier(id))) { f->set(value); return 0; }
    if (auto* i = dynamic_cast<IntProperty*>(p->getPropertyByIdentifier(id))) { i->set((int)value); return 0; }
    if (auto* b = dynamic_cast<BoolProperty*>(p->getPropertyByIdentifier(id))) { b->set(value != 0.f); return 0; }
    return -2;
}
int cpmh_set_property_string(cpmh_network* net, const char* processor, const char* id, const char* value) {
    Processor* p = !strcmp(processor, "lightvolume") ? (Processor*)&net->lightVolume : !strcmp(processor, "tracer") ? (Processor*)&net->tracer : nullptr;
    if (!p) return -1;
    if (auto* s = dynamic_cast<StringOptionProperty*>(p->getPropertyByIdentifier(id))) { s->set(value); return 0; }
    return -2;
}

void cpmh_light_volume_dims(cpmh_network* net, int dims[3], int* channels) {
    auto lv = net->lightVolume.outport_.getData();
    const size3_t d = lv->getDimensions();
    dims[0] = (int)d.x; dims[1] = (int)d.y; dims[2] = (int)d.z;
    *channels = lv->channels;
}
int cpmh_download_light_volume(cpmh_network* net, float* out) {
    auto lv = net->lightVolume.outport_.getData();
    return hipMemcpy(out, lv->data.device(), lv->data.getSizeInBytes(), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
int cpmh_n_photons(cpmh_network* net) { auto p = net->tracer.outport_.getData(); return p ? (int)p->getNumberOfPhotons() : 0; }
// out: float8 records (the reference's), whatever layout the library keeps them in on the device
int cpmh_download_photons(cpmh_network* net, float* out) {
    auto p = net->tracer.outport_.getData();
    if (hipMemcpy(out, p->photons_.device(), p->photons_.getSizeInBytes(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (cpm_get_photon_layout(CpmRuntime::get().ctx()) == CPM_PHOTONS_PLANAR) {
        const size_t n = p->photons_.getSizeInBytes() / 32;  // records
        std::vector<float> planes(out, out + 8 * n);
        for (size_t j = 0; j < n; ++j)
            for (int c = 0; c < 4; ++c) { out[8 * j + c] = planes[4 * j + c]; out[8 * j + 4 + c] = planes[4 * (n + j) + c]; }
    }
    return 0;
}
int cpmh_n_recomputed(cpmh_network* net) { auto r = net->tracer.recomputedIndicesPort_.getData(); return r ? r->resolveCount() : -1; }
int cpmh_remaining(cpmh_network* net) { return net->tracer.remainingPhotonsToUpdate(); }
const char* cpmh_last_light_volume_path(cpmh_network* net) { return net->lightVolume.lastPath(); }
const char* cpmh_last_tracer_decision(cpmh_network* net) { return net->tracer.lastDecision(); }
// measured GPU-timeline cost of the two ways to serve a change: { full trace, full light volume, branch trace, branch light volume } ms (-1 = not measured yet)
void cpmh_path_costs(cpmh_network* net, float out[4]) {
    net->tracer.pollCosts(); net->lightVolume.pollCosts();  // (what the next evaluation would pick up first)
    const auto& c = net->tracer.costs();
    out[0] = c.fullTraceMs; out[1] = c.fullLightVolumeMs; out[2] = c.branchTraceMs(); out[3] = c.branchLightVolumeMs();
}
// Multi-GPU call site, driven with a communicator of size 1 on this process's device (a real RCCL communicator: the
// network's photons are then "the one shard", the outport carries the reduced volume).
int cpmh_enable_shard_reduce(cpmh_network* net) {
    if (net->comm) return 0;
    uint8_t id[CPM_COMM_ID_BYTES];
    auto& rt = CpmRuntime::get();
    if (!rt.check(cpm_comm_get_unique_id(rt.ctx(), id), "cpm_comm_get_unique_id")) return -1;
    if (!rt.check(cpm_comm_create(rt.ctx(), id, 0, 1, &net->comm), "cpm_comm_create")) return -2;
    net->lightVolume.setCommunicator(net->comm);
    return 0;
}
const char* cpmh_last_reduce(cpmh_network* net) { return net->lightVolume.lastReduce(); }
// OpenGL sharing call site: the host's pixel-unpack buffer the light volume's texels are written into (needs its GL context)
void cpmh_share_light_volume_gl(cpmh_network* net, unsigned gl_buffer, int texel) { net->lightVolume.shareLightVolumeWithGL(gl_buffer, texel); }
const char* cpmh_last_gl_copy(cpmh_network* net) { return net->lightVolume.lastGLCopy(); }
// progressive refinement: switch it on (the tracer then writes its RNG state back), one timer tick + evaluation per call
int cpmh_enable_refinement(cpmh_network* net, int on) {
    net->tracer.enableProgressiveRefinement_.set(on != 0);
    net->tracer.progressiveRefinementChanged();
    return 0;
}
int cpmh_refine(cpmh_network* net) {
    net->tracer.onTimerEvent();
    net->tracer.process();
    net->lightVolume.process();
    return hipDeviceSynchronize() == hipSuccess ? net->tracer.outport_.getData()->iteration() : -2;
}
double cpmh_radius(cpmh_network* net) { return net->tracer.outport_.getData()->getRadiusRelativeToSceneSize(); }
// light plane the directional sampler fitted: origin, u, v (3 floats each), area
void cpmh_light_plane(cpmh_network* net, float out[10]) {
    vec3 o, u, v; float area;
    std::tie(o, u, v, area) = net->lightSampler.lightSampler_.lastPlane();
    const float t[10] = { o.x, o.y, o.z, u.x, u.y, u.z, v.x, v.y, v.z, area };
    memcpy(out, t, sizeof(t));
}
// ... of light `light` (0 = the network's first), and the direction its sampler used
int cpmh_light_plane_of(cpmh_network* net, int light, float out[10], float dir[3]) {
    if (!net || light < 0 || light > (int)net->extraLights.size()) return -1;
    DirectionalLightSamplerCLProcessor& s = light == 0 ? net->lightSampler : net->extraLights[light - 1]->sampler;
    auto src = light == 0 ? net->lightSource.getData() : net->extraLights[light - 1]->source.getData();
    vec3 o, u, v; float area;
    std::tie(o, u, v, area) = s.lightSampler_.lastPlane();
    const float t[10] = { o.x, o.y, o.z, u.x, u.y, u.z, v.x, v.y, v.z, area };
    memcpy(out, t, sizeof(t));
    vec3 d = normalize(src->direction);
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
    return 0;
}

// direction the sampler used (normalised as the reference does) and the tracer's 1024-texel TF LUT
void cpmh_light_direction(cpmh_network* net, float out[3]) {
    vec3 d = normalize(net->lightSource.getData()->direction);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void cpmh_tf_lut(cpmh_network* net, float* out4096) {
    std::vector<float> lut = net->tracer.transferFunction_.lut(1024);
    memcpy(out4096, lut.data(), lut.size() * sizeof(float));
}

// The drop-in surface as text: "classId|in:a,b|out:c|prop:x,y" per line.
const char* cpmh_describe_surface(cpmh_network* net) {
    static std::string s;
    std::ostringstream os;
    Processor* ps[] = { &net->sampleGenerator, &net->lightSampler, &net->minMax, &net->importance, &net->tracer, &net->lightVolume };
    for (Processor* p : ps) {
        os << p->getProcessorInfo().classIdentifier << "|in:";
        for (auto& i : p->getInportIds()) os << i << ",";
        os << "|out:";
        for (auto& o : p->getOutportIds()) os << o << ",";
        os << "|prop:";
        for (auto& q : p->getPropertyIds()) os << q << ",";
        os << "\n";
    }
    s = os.str();
    return s.c_str();
}

// ---- timing through the C++ layer (bench.py's correlated figures; VERDICT r02: not Python wall time) ------------------------
// Every repetition is timed from before the host-side edit until the device is idle again (wall clock around
// process() calls + one hipDeviceSynchronize): what an Inviwo evaluation of the same network costs end to end.

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// `reps` transfer-function edits alternating between two point lists (edit, revert, edit, ...): importance grid, tracer
// (importance branch), light volume.  out_ms[reps]; out_n[reps] = photons re-traced (-1: everything).
int cpmh_bench_tf_edits(cpmh_network* net, const float* tf_a5, int na, const float* tf_b5, int nb, int reps, double* out_ms, int* out_n) {
    if (!net || !net->correlated) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now_ms();
        if (r % 2 == 0) cpmh_set_transfer_function(net, tf_a5, na);
        else cpmh_set_transfer_function(net, tf_b5, nb);
        net->importance.process();
        net->tracer.process();
        net->lightVolume.process();
        if (hipDeviceSynchronize() != hipSuccess) return -2;
        out_ms[r] = now_ms() - t0;
        if (out_n) out_n[r] = cpmh_n_recomputed(net);
    }
    return 0;
}

// The same edits with the host's clock read between the stages: stage_ms[5 r + k] = time since the edit when (0) the property is set,
// (1) the importance processor, (2) the tracer, (3) the light-volume processor have returned, (4) the device is idle.
int cpmh_bench_tf_edits_timeline(cpmh_network* net, const float* tf_a5, int na, const float* tf_b5, int nb, int reps, double* stage_ms) {
    if (!net || !net->correlated || !stage_ms) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int r = 0; r < reps; ++r) {
        double* o = stage_ms + 5 * (size_t)r;
        const double t0 = now_ms();
        if (r % 2 == 0) cpmh_set_transfer_function(net, tf_a5, na);
        else cpmh_set_transfer_function(net, tf_b5, nb);
        o[0] = now_ms() - t0;
        net->importance.process();
        o[1] = now_ms() - t0;
        net->tracer.process();
        o[2] = now_ms() - t0;
        net->lightVolume.process();
        o[3] = now_ms() - t0;
        if (hipDeviceSynchronize() != hipSuccess) return -2;
        o[4] = now_ms() - t0;
    }
    return 0;
}

// `reps` full frames of the same network: everything invalidated (tracer: all photons; light volume: bin + gather).
int cpmh_bench_full_frames(cpmh_network* net, int reps, double* out_ms) {
    if (!net) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now_ms();
        net->tracer.invalidateProgressiveRendering(PhotonData::InvalidationReason::All);
        net->tracer.process();
        net->lightVolume.process();
        if (hipDeviceSynchronize() != hipSuccess) return -2;
        out_ms[r] = now_ms() - t0;
    }
    return 0;
}

// `reps` full frames enqueued back to back, ONE synchronisation at the end: the network's THROUGHPUT (what bench.py's headline
// measures through the Python driver); cpmh_bench_full_frames above is its LATENCY from an idle device.  host_ms (nullable):
// the host time the evaluations themselves took (enqueueing).
int cpmh_bench_frames_back_to_back(cpmh_network* net, int reps, double* total_ms, double* host_ms) {
    if (!net) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const double t0 = now_ms();
    for (int r = 0; r < reps; ++r) {
        net->tracer.invalidateProgressiveRendering(PhotonData::InvalidationReason::All);
        net->tracer.process();
        net->lightVolume.process();
    }
    const double t1 = now_ms();
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (total_ms) *total_ms = now_ms() - t0;
    if (host_ms) *host_ms = t1 - t0;
    return 0;
}

// Per-kernel GPU time of `reps` full frames of the network (the library's HIP-event hook on the runtime's context):
// "kernel name=ms per frame;..." -- what bench.py reports for the workspace point.
const char* cpmh_profile_full_frames(cpmh_network* net, int reps) {
    static std::string out;
    out.clear();
    if (!net || reps < 1) return out.c_str();
    cpm_ctx* ctx = CpmRuntime::get().ctx();
    if (hipDeviceSynchronize() != hipSuccess) return out.c_str();
    cpm_profile_reset(ctx);
    cpm_profile_enable(ctx, 1);
    for (int r = 0; r < reps; ++r) {
        net->tracer.invalidateProgressiveRendering(PhotonData::InvalidationReason::All);
        net->tracer.process();
        net->lightVolume.process();
    }
    const int n = cpm_profile_collect(ctx);
    std::ostringstream os;
    for (int i = 0; i < n; ++i) os << cpm_profile_name(ctx, i) << "=" << cpm_profile_total_ms(ctx, i) / reps << ";";
    if (!CpmRuntime::get().profiling()) cpm_profile_enable(ctx, 0);
    cpm_profile_reset(ctx);
    out = os.str();
    return out.c_str();
}

// ---- time-varying data: .u3d files and the sequence processors ------------------------------------------

struct cpmh_sequence;
int cpmh_sequence_evaluate(cpmh_sequence* s);
int cpmh_sequence_set_time(cpmh_sequence* s, float time);

static std::string g_u3d_error;
const char* cpmh_last_error() { return g_u3d_error.c_str(); }

static std::shared_ptr<UniformGrid3DBase> make_grid(int format) {
    if (format == 0) return std::make_shared<MinMaxUniformGrid3D>();
    return std::make_shared<ImportanceUniformGrid3D>();
}

// format 0 = Vec2UINT16 (min/max grid), 1 = FLOAT32; `data` holds `count` elements back to back.  Host only.
int cpmh_u3d_write(const char* path, int format, const int dims[3], const int cell[3], const float model[16], const float world[16],
                   const void* data, int count, int overwrite) {
    try {
        UniformGrid3DVector v;
        const char* src = (const char*)data;
        for (int t = 0; t < count; ++t) {
            auto g = make_grid(format);
            g->setCellDimension(size3_t{ (size_t)cell[0], (size_t)cell[1], (size_t)cell[2] });
            mat4 m, w;
            for (int i = 0; i < 16; ++i) { m[i] = model[i]; w[i] = world[i]; }
            g->setModelMatrix(m); g->setWorldMatrix(w);
            g->setDimensions(size3_t{ (size_t)dims[0], (size_t)dims[1], (size_t)dims[2] });
            std::memcpy(g->hostData(), src + (size_t)t * g->getSizeInBytes(), g->getSizeInBytes());
            v.push_back(g);
        }
        UniformGrid3DWriter w;
        w.setOverwrite(overwrite != 0);
        w.writeData(&v, path);
        return 0;
    } catch (const std::exception& e) {
        g_u3d_error = e.what();
        return -1;
    }
}

static std::shared_ptr<UniformGrid3DVector> g_u3d_last;
int cpmh_u3d_read(const char* path, int* format, int dims[3], int cell[3], float model[16], float world[16], int* count,
                  unsigned long long* element_bytes) {
    try {
        g_u3d_last = UniformGrid3DReader().readData(path);
        auto& g = *g_u3d_last->front();
        *format = std::string(g.getDataFormatString()) == "Vec2UINT16" ? 0 : 1;
        dims[0] = (int)g.getDimensions().x; dims[1] = (int)g.getDimensions().y; dims[2] = (int)g.getDimensions().z;
        cell[0] = (int)g.getCellDimension().x; cell[1] = (int)g.getCellDimension().y; cell[2] = (int)g.getCellDimension().z;
        for (int i = 0; i < 16; ++i) { model[i] = g.getModelMatrix()[i]; world[i] = g.getWorldMatrix()[i]; }
        *count = (int)g_u3d_last->size();
        *element_bytes = g.getSizeInBytes();
        return 0;
    } catch (const std::exception& e) {
        g_u3d_error = e.what();
        g_u3d_last.reset();
        return -1;
    }
}
int cpmh_u3d_read_data(void* out) {  // the elements of the last successful cpmh_u3d_read
    if (!g_u3d_last) return -1;
    char* dst = (char*)out;
    for (auto& g : *g_u3d_last) { std::memcpy(dst, g->hostData(), g->getSizeInBytes()); dst += g->getSizeInBytes(); }
    return 0;
}

// volume sequence -> { VolumeSequencePlayer, VolumeMinMaxCLProcessor (sequence ports) -> UniformGrid3DPlayerProcessor,
//                      DynamicVolumeDifferenceAnalysis -> UniformGrid3DPlayerProcessor }
struct cpmh_sequence {
    std::shared_ptr<VolumeSequence> volumes = std::make_shared<VolumeSequence>();
    DataOutport<VolumeSequence> source{ "data" };
    VolumeSequencePlayer volumePlayer;
    VolumeMinMaxCLProcessor minMax;
    DynamicVolumeDifferenceAnalysis difference;
    UniformGrid3DPlayerProcessor minMaxPlayer, differencePlayer;
#ifdef CPM_HOST_EXTRAS
    UniformGrid3DVectorSource gridSource;
    UniformGrid3DPlayerProcessor gridSourcePlayer;
#endif
    bool analysed = false;
};

cpmh_sequence* cpmh_sequence_create(const void* voxels, int dtype, int dx, int dy, int dz, int count, int region) {
    if (!CpmRuntime::get().valid()) return nullptr;
    auto* s = new cpmh_sequence();
    for (int t = 0; t < count; ++t) {
        auto v = std::make_shared<Volume>(size3_t{ (size_t)dx, (size_t)dy, (size_t)dz }, dtype);
        const size_t bytes = (size_t)dx * dy * dz * v->elementSize();
        v->ramBytes.assign((const uint8_t*)voxels + (size_t)t * bytes, (const uint8_t*)voxels + (size_t)(t + 1) * bytes);
        s->volumes->push_back(v);
    }
    s->source.setData(s->volumes);
    s->minMax.volumeRegionSize_.set(region);
    s->difference.volumeRegionSize_.set(region);
    s->volumePlayer.inport_.connectTo(&s->source);
    s->minMax.vectorInport_.connectTo(&s->source);
    s->difference.inport_.connectTo(&s->source);
    s->minMaxPlayer.inport_.connectTo(&s->minMax.vectorOutport_);
    s->differencePlayer.inport_.connectTo(&s->difference.outport_);
#ifdef CPM_HOST_EXTRAS
    s->gridSourcePlayer.inport_.connectTo(&s->gridSource.port_);
#endif
    return s;
}
void cpmh_sequence_destroy(cpmh_sequence* s) { delete s; }

// The time-varying form of the workspace: the network's volume comes from the sequence's VolumeSequencePlayer, the
// importance processor's min/max grid and volume-difference grid from the two UniformGrid3D players
// (minMaxUniformGrid3D <- InterpolatedData of the min/max player, volumeDifferenceInfo <- InterpolatedData of the difference player).
int cpmh_attach_sequence(cpmh_network* net, cpmh_sequence* s) {
    if (!net || !s || !net->correlated) return -1;
    if (cpmh_sequence_evaluate(s) != 0) return -2;
    net->tracer.volumePort_.disconnectAll();
    net->tracer.volumePort_.connectTo(&s->volumePlayer.outport_);
    net->lightVolume.volumeInport_.disconnectAll();
    net->lightVolume.volumeInport_.connectTo(&s->volumePlayer.outport_);
    net->importance.minMaxUniformGrid3DInport_.disconnectAll();
    net->importance.minMaxUniformGrid3DInport_.connectTo(&s->minMaxPlayer.outport_);
    net->importance.volumeDifferenceInfoInport_.disconnectAll();
    net->importance.volumeDifferenceInfoInport_.connectTo(&s->differencePlayer.outport_);
    return 0;
}
// One displayed time: players (volume mix, grid mixes), importance (time-varying), tracer, light volume.
// times_ms[0] = the players, times_ms[1] = importance + tracer + light volume (each until the device is idle).
int cpmh_sequence_step(cpmh_network* net, cpmh_sequence* s, float time, double* times_ms) {
    if (!net || !s) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const double t0 = now_ms();
    cpmh_sequence_set_time(s, time);
    if (cpmh_sequence_evaluate(s) != 0) return -3;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const double t1 = now_ms();
    net->importance.process();
    net->tracer.process();
    net->lightVolume.process();
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (times_ms) { times_ms[0] = t1 - t0; times_ms[1] = now_ms() - t1; }
    return cpmh_n_recomputed(net);
}

// The same displayed time as ONE measurement: players + importance + tracer + light volume enqueued back to back, one
// synchronisation at the end (cpmh_sequence_step waits in the middle to report its two parts, which costs the second part an
// idle device).  Returns the photons re-traced (-1: everything), *total_ms the wall clock.
int cpmh_sequence_step_total(cpmh_network* net, cpmh_sequence* s, float time, double* total_ms) {
    if (!net || !s) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const double t0 = now_ms();
    cpmh_sequence_set_time(s, time);
    if (cpmh_sequence_evaluate(s) != 0) return -3;
    net->importance.process();
    net->tracer.process();
    net->lightVolume.process();
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (total_ms) *total_ms = now_ms() - t0;
    return cpmh_n_recomputed(net);
}

static void for_each_clock(cpmh_sequence* s, const std::function<void(SequenceClock&)>& f) {
    f(s->volumePlayer.clock_); f(s->minMaxPlayer.clock_); f(s->differencePlayer.clock_);
#ifdef CPM_HOST_EXTRAS
    f(s->gridSourcePlayer.clock_);
#endif
}
int cpmh_sequence_evaluate(cpmh_sequence* s) {
    if (!s->analysed) {  // per-sequence analysis runs once (the sequence does not change)
        s->minMax.process();
        s->difference.process();
        s->analysed = true;
    }
    if (!s->volumePlayer.keepSequenceOnDevice_.get())   // a sequence that stays in host memory: the analyses' device copies of its elements go again
        for (const auto& v : *s->volumes) v->invalidateDeviceRepresentation();
    s->volumePlayer.process();
    s->minMaxPlayer.process();
    s->differencePlayer.process();
#ifdef CPM_HOST_EXTRAS
    s->gridSource.process();
    s->gridSourcePlayer.process();
#endif
    return (s->volumePlayer.outport_.getData() && s->minMaxPlayer.outport_.getData() && s->differencePlayer.outport_.getData()) ? 0 : -1;
}
// keep != 0 (default): the elements become resident on first use and stay (Inviwo's VolumeCL representations); 0: they stay in host memory and
// reach the device through the player's ring of three volumes, the next element prefetched on the library's copy stream (cpm_volume_stream)
void cpmh_sequence_keep_on_device(cpmh_sequence* s, int keep) {
    if (s) s->volumePlayer.keepSequenceOnDevice_.set(keep != 0);
}
// out[0] = finished uploads whose time has been read, out[1] = their H2D time in ms, out[2] = bytes per element, out[3] = elements a frame had to
// wait for (uploaded at the acquire, not ahead of it); 0 when the player streams, -1 when it does not
int cpmh_sequence_stream_stats(cpmh_sequence* s, double* out) {
    unsigned long long n = 0, late = 0, bytes = 0;
    double ms = 0.0;
    if (!s || !out || !s->volumePlayer.streamStats(&n, &late, &ms, &bytes)) return -1;
    out[0] = (double)n; out[1] = ms; out[2] = (double)bytes; out[3] = (double)late;
    return 0;
}
void cpmh_sequence_set_time_per_element(cpmh_sequence* s, float seconds) {
    for_each_clock(s, [&](SequenceClock& c) { c.timePerElement_.set(seconds); });
}
int cpmh_sequence_set_time(cpmh_sequence* s, float time) {
    for_each_clock(s, [&](SequenceClock& c) { c.time_.set(time); });
    return s->volumePlayer.clock_.index_.get();
}
int cpmh_sequence_tick(cpmh_sequence* s) {  // one firing of the play timers
    for_each_clock(s, [&](SequenceClock& c) { c.onSequenceTimerEvent(); });
    return s->volumePlayer.clock_.index_.get();
}
float cpmh_sequence_time(cpmh_sequence* s) { return s->volumePlayer.clock_.time_.get(); }
float cpmh_sequence_max_time(cpmh_sequence* s) { return s->volumePlayer.clock_.time_.getMaxValue(); }
float cpmh_sequence_weight(cpmh_sequence* s) { return s->volumePlayer.clock_.weight(); }
// what: 0 = interpolated volume, 1 = interpolated min/max grid, 2 = interpolated difference grid, 3 = grid from the .u3d source
int cpmh_sequence_download(cpmh_sequence* s, int what, void* out) {
    if (what == 0) {
        auto v = s->volumePlayer.outport_.getData();
        if (!v || !v->downloadToRAM()) return -1;
        std::memcpy(out, v->ramBytes.data(), v->ramBytes.size());
        return 0;
    }
#ifdef CPM_HOST_EXTRAS
    auto g = what == 1 ? s->minMaxPlayer.outport_.getData() : (what == 2 ? s->differencePlayer.outport_.getData() : s->gridSourcePlayer.outport_.getData());
#else
    if (what != 1 && what != 2) return -1;
    auto g = what == 1 ? s->minMaxPlayer.outport_.getData() : s->differencePlayer.outport_.getData();
#endif
    if (!g) return -1;
    std::memcpy(out, g->hostData(), g->getSizeInBytes());
    return 0;
}
#ifdef CPM_HOST_EXTRAS
// what: 1 = the min/max grids of all time steps, 2 = the difference grids
int cpmh_sequence_export(cpmh_sequence* s, int what, const char* path) {
    UniformGrid3DExport exp;
    DataOutport<UniformGrid3DVector>* src = what == 1 ? &s->minMax.vectorOutport_ : &s->difference.outport_;
    if (!src->getData()) return -1;
    exp.port_.connectTo(src);
    exp.file_.set(path);
    exp.overwrite_.set(true);
    exp.exportData();
    return 0;
}
void cpmh_sequence_load_grids(cpmh_sequence* s, const char* path) { s->gridSource.filePath.set(path); }
#endif
const char* cpmh_sequence_describe_surface(cpmh_sequence* s) {
    static std::string str;
    std::ostringstream os;
#ifdef CPM_HOST_EXTRAS
    UniformGrid3DExport exp;
    UniformGrid3DSequenceSelector sel;
    Processor* ps[] = { &s->volumePlayer, &s->minMax, &s->difference, &s->minMaxPlayer, &s->gridSource, &exp, &sel };
#else
    Processor* ps[] = { &s->volumePlayer, &s->minMax, &s->difference, &s->minMaxPlayer };
#endif
    for (Processor* p : ps) {
        os << p->getProcessorInfo().classIdentifier << "|in:";
        for (auto& i : p->getInportIds()) os << i << ",";
        os << "|out:";
        for (auto& o : p->getOutportIds()) os << o << ",";
        os << "|prop:";
        for (auto& q : p->getPropertyIds()) os << q << ",";
        os << "\n";
    }
    str = os.str();
    return str.c_str();
}

// ---- modules and the processor factory ---------------------------------------------------------------------------

// "module|version|processor ids,|port class ids,|data formats," per line; registers the modules on first use
const char* cpmh_modules_describe() {
    static std::vector<std::unique_ptr<InviwoModule>> modules = registerCorrelatedPhotonMappingModules();
    static std::string str;
    std::ostringstream os;
    for (auto& m : modules) {
        os << m->getIdentifier() << "|" << m->getVersion() << "|";
        for (auto& p : m->processors()) os << p << ",";
        os << "|";
        for (auto& p : m->ports()) os << p << ",";
        os << "|";
        for (auto& f : m->dataFormats()) os << f << ",";
        os << "\n";
    }
    str = os.str();
    return str.c_str();
}
// instantiate a processor by class identifier (what deserialising a workspace does); returns its surface line or ""
const char* cpmh_factory_create(const char* class_identifier) {
    static std::string str;
    cpmh_modules_describe();
    auto p = ProcessorFactory::get().create(class_identifier);
    if (!p) { str.clear(); return str.c_str(); }
    std::ostringstream os;
    os << p->getProcessorInfo().classIdentifier << "|in:";
    for (auto& i : p->getInportIds()) os << i << ",";
    os << "|out:";
    for (auto& o : p->getOutportIds()) os << o << ",";
    os << "|prop:";
    for (auto& q : p->getPropertyIds()) os << q << ",";
    str = os.str();
    return str.c_str();
}
#ifdef CPM_HOST_EXTRAS
// RadixSortCL processor: sorts n (key, data) u32 pairs on the device through the processor's ports
int cpmh_radixsort_processor(uint32_t* keys, uint32_t* data, int n) {
    if (!CpmRuntime::get().valid()) return -1;
    cpmh_modules_describe();
    auto p = ProcessorFactory::get().create("org.inviwo.RadixSortCL");
    auto* rs = dynamic_cast<RadixSortCL*>(p.get());
    if (!rs) return -2;
    auto kb = std::make_shared<Buffer<uint32_t>>((size_t)n);
    auto db = std::make_shared<Buffer<uint32_t>>((size_t)n);
    std::memcpy(kb->ram().data(), keys, (size_t)n * 4);
    std::memcpy(db->ram().data(), data, (size_t)n * 4);
    DataOutport<Buffer<uint32_t>> ko{ "keys" }, dout{ "data" };
    ko.setData(kb); dout.setData(db);
    rs->keysPort_.connectTo(&ko);
    rs->inputPort_.connectTo(&dout);
    rs->process();
    auto out = rs->outputPort_.getData();
    if (!out) return -3;
    std::memcpy(keys, kb->hostData(), (size_t)n * 4);
    std::memcpy(data, out->hostData(), (size_t)n * 4);
    return 0;
}
#endif

}  // extern "C"
