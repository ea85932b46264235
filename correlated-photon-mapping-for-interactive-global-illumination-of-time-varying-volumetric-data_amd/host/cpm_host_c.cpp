// cpm_host_c.cpp -- a plain-C facade over the host layer: builds the processor network of the
// CorrelatedPhotonMappingSingleVolume workspace (sample generator -> directional light sampler ->
// photon tracer -> photon-to-light-volume, plus min/max -> importance -> tracer) and evaluates it,
// so that tests can drive the C++ surface through ctypes.  Connections follow
// workspaces/CorrelatedPhotonMappingSingleVolume.inv:1178-1271.
#include <cstring>
#include <sstream>

#include "cpm_processors.h"

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

// Evaluate the network once, upstream first (what Inviwo's evaluator does on invalidation).
// first != 0: everything (light samples included); else only what a TF edit / timer tick invalidates.
int cpmh_evaluate(cpmh_network* net, int first) {
    if (!net) return -1;
    if (first) {
        net->sampleGenerator.process();
        net->lightSampler.process();
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
    Processor* p = !strcmp(processor, "lightvolume") ? (Processor*)&net->lightVolume : nullptr;
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
int cpmh_download_photons(cpmh_network* net, float* out) {
    auto p = net->tracer.outport_.getData();
    return hipMemcpy(out, p->photons_.device(), p->photons_.getSizeInBytes(), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
int cpmh_n_recomputed(cpmh_network* net) { auto r = net->tracer.recomputedIndicesPort_.getData(); return r ? r->nRecomputedPhotons : -1; }
int cpmh_remaining(cpmh_network* net) { return net->tracer.remainingPhotonsToUpdate(); }
const char* cpmh_last_light_volume_path(cpmh_network* net) { return net->lightVolume.lastPath(); }
double cpmh_radius(cpmh_network* net) { return net->tracer.outport_.getData()->getRadiusRelativeToSceneSize(); }
// light plane the directional sampler fitted: origin, u, v (3 floats each), area
void cpmh_light_plane(cpmh_network* net, float out[10]) {
    vec3 o, u, v; float area;
    std::tie(o, u, v, area) = net->lightSampler.lightSampler_.lastPlane();
    const float t[10] = { o.x, o.y, o.z, u.x, u.y, u.z, v.x, v.y, v.z, area };
    memcpy(out, t, sizeof(t));
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

}  // extern "C"
