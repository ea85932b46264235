// cpm_processors.cpp -- see cpm_processors.h.  Host logic only; all device work goes through
// include/cpm/cpm.h.  Error behaviour follows the reference: log, skip the step, carry on.
#include "cpm_processors.h"

#include <cpm/cpm_profile.h>

#include "cpm_hostmath.h"

#include <cstdlib>

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace inviwo {

void LogErrorImpl(const std::string& source, const std::string& msg) { fprintf(stderr, "[error] %s: %s\n", source.c_str(), msg.c_str()); }
void LogInfoImpl(const std::string& source, const std::string& msg) { fprintf(stderr, "[info] %s: %s\n", source.c_str(), msg.c_str()); }

// ---- runtime ----------------------------------------------------------------------------------------

CpmRuntime& CpmRuntime::get() {
    static CpmRuntime rt;
    return rt;
}
CpmRuntime::CpmRuntime() {
    int rc = cpm_create(0, &ctx_);
    if (rc != CPM_OK) {
        LogError(std::string("cpm_create failed: ") + cpm_last_error_string(nullptr));
        ctx_ = nullptr;
    }
    const char* e = std::getenv("CPM_PROFILING");
    profiling_ = ctx_ && e && e[0] && e[0] != '0';
    if (profiling_) cpm_profile_enable(ctx_, 1);
    // The records travel from processor to processor inside this library only, so they lie the way the device reads them fastest: two
    // planes (the brick bin of a one-channel light volume then streams 16 of a record's 32 bytes).  CPM_HOST_PHOTON_LAYOUT=interleaved keeps
    // the reference's float8 record in the buffers (what a processor outside this library reading the `photons` port would expect).
    const char* lay = std::getenv("CPM_HOST_PHOTON_LAYOUT");
    if (ctx_ && !(lay && std::string(lay) == "interleaved")) cpm_set_photon_layout(ctx_, CPM_PHOTONS_PLANAR);
}
void CpmRuntime::beginProfile() const {
    if (profiling_) cpm_profile_reset(ctx_);
}
void CpmRuntime::logProfile(const char* label) const {
    if (!profiling_) return;
    const int n = cpm_profile_collect(ctx_);  // synchronises, like the reference's wait on the last event
    std::string line = std::string(label) + ": ";
    double total = 0;
    char buf[160];
    for (int i = 0; i < n; ++i) {
        const double ms = cpm_profile_total_ms(ctx_, i);
        total += ms;
        std::snprintf(buf, sizeof buf, "%s%s x%ld %.3f ms", i ? " + " : "", cpm_profile_name(ctx_, i), cpm_profile_calls(ctx_, i), ms);
        line += buf;
    }
    std::snprintf(buf, sizeof buf, " = %.3f ms", total);
    LogInfo(line + buf);
}
CpmRuntime::~CpmRuntime() { cpm_destroy(ctx_); }
bool CpmRuntime::check(int status, const char* what) const {
    if (status == CPM_OK) return true;
    LogError(std::string(what) + ": " + cpm_last_error_string(ctx_));
    return false;
}

StreamSpan::~StreamSpan() {
    if (a_) (void)hipEventDestroy(a_);
    if (b_) (void)hipEventDestroy(b_);
}
void StreamSpan::begin(hipStream_t s, float* target, bool force) {
    poll();
    if (pending_) { open_ = false; return; }  // the previous span has not completed yet: skip this measurement
    if (target && *target >= 0.f && !force && ++skipped_[target] < kEvery) { open_ = false; return; }  // known: sampled every kEvery-th time
    if (target) skipped_[target] = 0;
    if (!a_ && (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess)) { a_ = b_ = nullptr; open_ = false; return; }
    target_ = target;
    open_ = hipEventRecord(a_, s) == hipSuccess;
}
void StreamSpan::end(hipStream_t s) {
    if (!open_) return;
    open_ = false;
    pending_ = hipEventRecord(b_, s) == hipSuccess;
}
void StreamSpan::poll() {
    if (!pending_ || hipEventQuery(b_) != hipSuccess) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a_, b_) == hipSuccess && target_)
        *target_ = *target_ < -1.5f ? -1.f : (*target_ < 0.f ? ms : 0.5f * (*target_ + ms));  // first sample dropped, then a running mean
    pending_ = false;
}

// ---- data ---------------------------------------------------------------------------------------------

void TransferFunction::sort() {
    std::stable_sort(points_.begin(), points_.end(), [](const TFPrimitive& a, const TFPrimitive& b) { return a.pos < b.pos; });
}

std::vector<float> TransferFunction::lut(int width) const {
    std::vector<float> out((size_t)width * 4, 0.f);
    if (points_.empty()) return out;
    for (int i = 0; i < width; ++i) {
        const double x = ((double)i + 0.5) / (double)width;
        double c[4];
        if (x <= points_.front().pos) {
            const vec4& k = points_.front().color; c[0] = k.x; c[1] = k.y; c[2] = k.z; c[3] = k.w;
        } else if (x >= points_.back().pos) {
            const vec4& k = points_.back().color; c[0] = k.x; c[1] = k.y; c[2] = k.z; c[3] = k.w;
        } else {
            size_t j = 0;
            while (j + 1 < points_.size() && points_[j + 1].pos <= x) ++j;
            const TFPrimitive &a = points_[j], &b = points_[j + 1];
            const double fa[4] = { a.color.x, a.color.y, a.color.z, a.color.w }, fb[4] = { b.color.x, b.color.y, b.color.z, b.color.w };
            for (int k = 0; k < 4; ++k) c[k] = (fb[k] - fa[k]) / (b.pos - a.pos) * (x - a.pos) + fa[k];
        }
        for (int k = 0; k < 4; ++k) out[4 * (size_t)i + k] = (float)c[k];
    }
    return out;
}

// A photon's direction is stored as (polar angle from +z, azimuth from +x): photondata.cpp:100-117, the host twin of the
// kernels' encodeDirection / decodeDirection.
void Photon::setDirection(vec3 d) {
    const float cosPolar = d.z < -1.f ? -1.f : (d.z > 1.f ? 1.f : d.z);
    encodedDirection = vec2{ std::acos(cosPolar), std::atan2(d.y, d.x) };
}
vec3 Photon::getDirection() const {
    const float polar = encodedDirection.x, azimuth = encodedDirection.y, ring = std::sin(polar);
    return vec3{ ring * std::cos(azimuth), ring * std::sin(azimuth), std::cos(polar) };
}

// PhotonData (photondata.cpp:36-98): record storage is two vec4 per photon and interaction; the radius is kept in world units
// and handed out relative to the scene; the progressive schedule is Knaus & Zwicker's r' = r ((i + alpha) / (i + 1))^(1/3).
const double PhotonData::scaleToMakeLightPowerOfOneVisibleForDirectionalLightSource = 1. / M_PI;
void PhotonData::setSize(size_t count, int interactions) {
    maxPhotonInteractions_ = interactions;
    if (count != 0) photons_.setSize(count * 2 * interactions);
}
void PhotonData::setRadius(double relativeToScene, double sceneExtent) {
    sceneRadius_ = sceneExtent;
    worldSpaceRadius_ = relativeToScene * sceneExtent;
}
void PhotonData::advanceToNextIteration(double alpha) {
    const double shrunk = progressiveSphereRadius(getRadius(), iteration_, alpha);
    setRadius(shrunk);
    ++iteration_;
}
double PhotonData::progressiveSphereRadius(double r, int i, double alpha) {
    const double ratio = ((double)i + alpha) / (1.0 + (double)i);
    return r * std::pow(ratio, 1. / 3.);
}
double PhotonData::sphereVolume(double r) {
    const double unitBall = M_PI * 4. / 3.;
    return std::pow(r, 3) * unitBall;
}

std::shared_ptr<Mesh> Mesh::unitCube() { return box(vec3(0.f, 0.f, 0.f), vec3(1.f, 1.f, 1.f)); }
// the proxy geometry CubeProxyGeometry emits for clip ranges lo..hi (data space): 8 corners, 12 triangles
std::shared_ptr<Mesh> Mesh::box(vec3 lo, vec3 hi) {
    auto m = std::make_shared<Mesh>();
    for (int z = 0; z < 2; ++z) for (int y = 0; y < 2; ++y) for (int x = 0; x < 2; ++x) m->vertices.push_back(vec3(x ? hi.x : lo.x, y ? hi.y : lo.y, z ? hi.z : lo.z));
    const int quads[6][4] = { { 0, 1, 3, 2 }, { 4, 6, 7, 5 }, { 0, 4, 5, 1 }, { 2, 3, 7, 6 }, { 0, 2, 6, 4 }, { 1, 5, 7, 3 } };
    for (auto& q : quads) { const int t[6] = { q[0], q[1], q[2], q[0], q[2], q[3] }; m->indices.insert(m->indices.end(), t, t + 6); }
    return m;
}

// ---- host geometry ----------------------------------------------------------------------------------------
// The reference's entry points (lightcl/*.h) over this build's own host arithmetic: host/cpm_hostmath.cpp.

namespace geometry {

namespace {
std::vector<cpm_host::Pt2> toHost(const std::vector<vec2>& in) {
    std::vector<cpm_host::Pt2> out(in.size());
    for (size_t i = 0; i < in.size(); ++i) out[i] = { in[i].x, in[i].y };
    return out;
}
std::vector<cpm_host::Pt3> toHost(const std::vector<vec3>& in) {
    std::vector<cpm_host::Pt3> out(in.size());
    for (size_t i = 0; i < in.size(); ++i) out[i] = { in[i].x, in[i].y, in[i].z };
    return out;
}
inline cpm_host::Pt3 toHost(vec3 p) { return { p.x, p.y, p.z }; }
inline vec3 fromHost(cpm_host::Pt3 p) { return vec3(p.x, p.y, p.z); }
inline vec2 fromHost(cpm_host::Pt2 p) { return vec2{ p.x, p.y }; }
}  // namespace

void projectPointsOnPlane(const std::vector<vec3>& pts, const Plane& in, vec3 axis0, vec3 axis1, std::vector<vec2>& coords) {
    for (cpm_host::Pt2 c : cpm_host::planeCoordinates(toHost(pts), toHost(in.point), toHost(in.normal), toHost(axis0), toHost(axis1))) coords.push_back(fromHost(c));
}
std::vector<vec2> convexHull2D(std::vector<vec2> pts) {
    std::vector<vec2> out;
    for (cpm_host::Pt2 c : cpm_host::hullCycle(toHost(pts))) out.push_back(fromHost(c));
    return out;
}
std::tuple<vec2, vec2, vec2> mimumBoundingRectangle(const std::vector<vec2>& hull) {
    const cpm_host::Rectangle2 r = cpm_host::smallestRectangle(toHost(hull));
    return std::make_tuple(fromHost(r.corner), fromHost(r.side0), fromHost(r.side1));
}
std::tuple<vec3, vec3, vec3> fitPlaneAlignedOrientedBoundingBox2D(const std::vector<vec3>& pts, const Plane& in) {
    const cpm_host::LightRectangle r = cpm_host::fitLightRectangle(toHost(pts), toHost(in.point), toHost(in.normal));
    return std::make_tuple(fromHost(r.origin), fromHost(r.u), fromHost(r.v));
}

}  // namespace geometry

// ---- algorithm classes ------------------------------------------------------------------------------------

void MWC64XSeedGenerator::generateRandomSeeds(Buffer<uvec2>* buffer, unsigned int seed) {  // mwc64xseedgenerator.cpp:51-90
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    const size_t n = buffer->getSize();
    std::vector<uint32_t> bases(n);
    cpm_glibc_rand_sequence(seed, bases.data(), n);  // srand(seed); rand() per stream (Q13: platform independent)
    auto& ram = buffer->ram();
    for (size_t i = 0; i < n; ++i) ram[i] = uvec2{ bases[i], 0u };
    buffer->upload(rt.stream());
    rt.check(cpm_seed_streams(rt.ctx(), reinterpret_cast<uint32_t*>(buffer->device()), n, 1099511627776ull, rt.stream()), "cpm_seed_streams");
}

void UniformSampleGenerator2DCL::generateNextSamples(SampleBuffer& out, ivec2 n) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    out.setSize((size_t)n.x * n.y);
    rt.check(cpm_uniform_samples_2d(rt.ctx(), n.x, n.y, reinterpret_cast<float*>(out.device()), rt.stream()), "cpm_uniform_samples_2d");
}

void DirectionalLightSamplerCL::sampleLightSource(const Mesh* mesh, const SampleBuffer* samples, const DirectionalLight* light, LightSamples& out) {
    auto& rt = CpmRuntime::get();  // directionallightsamplercl.cpp:57-112
    if (!rt.valid() || mesh->vertices.empty()) return;
    if (samples->getSize() != out.getSize()) out.setSize(samples->getSize());
    vec3 lightDirection = normalize(light->direction);
    std::tie(origin_, u_, v_) = geometry::fitPlaneAlignedOrientedBoundingBox2D(mesh->vertices, geometry::Plane{ light->position, lightDirection });
    area_ = length(u_) * length(v_);
    const float rad[4] = { light->radiance.x, light->radiance.y, light->radiance.z, 1.f }, dir[4] = { lightDirection.x, lightDirection.y, lightDirection.z, 0.f };
    const float o[4] = { origin_.x, origin_.y, origin_.z, 1.f }, u[4] = { u_.x, u_.y, u_.z, 0.f }, v[4] = { v_.x, v_.y, v_.z, 0.f };
    rt.check(cpm_directional_light_samples(rt.ctx(), reinterpret_cast<const float*>(samples->device()), (int)samples->getSize(), rad, dir, o, u, v,
                                           area_, reinterpret_cast<float*>(out.getLightSamples()->device()), rt.stream()),
             "cpm_directional_light_samples");
    out.advanceIteration();
    ++out.changeStamp;
}

void LightSampleMeshIntersectionCL::meshSampleIntersection(const Mesh* mesh, LightSamples* samples) {  // lightsamplemeshintersectioncl.cpp:51-99
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || samples->getSize() == 0) return;
    Buffer<vec3> vtx(mesh->vertices.size());
    vtx.ram() = mesh->vertices;
    vtx.upload(rt.stream());
    Buffer<int> idx(mesh->indices.size());
    idx.ram() = mesh->indices;
    idx.upload(rt.stream());
    rt.check(cpm_light_sample_mesh_intersection(rt.ctx(), reinterpret_cast<const float*>(vtx.device()), idx.device(), (int)idx.getSize(),
                                                reinterpret_cast<const float*>(samples->getLightSamples()->device()), (int)samples->getSize(),
                                                reinterpret_cast<float*>(samples->getIntersectionPoints()->device()), rt.stream()),
             "cpm_light_sample_mesh_intersection");
    (void)hipStreamSynchronize(rt.stream());  // vtx / idx go out of scope
}

PhotonTracerCL::~PhotonTracerCL() {
    auto& rt = CpmRuntime::get();
    if (tf_) cpm_tf_destroy(rt.ctx(), tf_);
    for (auto& lo : launchOrders_) if (lo.second.order) cpm_trace_order_destroy(rt.ctx(), lo.second.order);
    if (allLightsOrder_.order) cpm_trace_order_destroy(rt.ctx(), allLightsOrder_.order);
}
namespace {
// a per-photon side buffer (importance keys, ...) follows the photon count
template <typename T>
void onePerPhoton(Buffer<T>& buffer, const PhotonData& photons) {
    const size_t n = photons.getNumberOfPhotons();
    if (buffer.getSize() != n) buffer.setSize(n);
}
}  // namespace
void PhotonTracerCL::seedStreamsFor(const PhotonData& photons) {  // one RNG stream per photon, seeded when the count changes (photontracercl.cpp:71-73)
    const size_t n = photons.getNumberOfPhotons();
    if (randomState_.getSize() != n) setRandomSeedSize(n);
}
void PhotonTracerCL::setRandomSeedSize(size_t nPhotons) {
    if (nPhotons > 0) {
        randomState_.setSize(nPhotons);
        MWC64XSeedGenerator().generateRandomSeeds(&randomState_, 0);
    }
}
void PhotonTracerCL::syncTF(const TransferFunction& tf) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    // (the points first: evaluating the 1024-texel LUT costs the host as much as the launch it decides about)
    const auto& pts = tf.points();
    bool same = tf_ && pts.size() == tfPoints_.size();
    for (size_t i = 0; same && i < pts.size(); ++i)
        same = pts[i].pos == tfPoints_[i].pos && std::memcmp(&pts[i].color, &tfPoints_[i].color, sizeof(vec4)) == 0;
    if (same) return;
    tfPoints_ = pts;
    std::vector<float> lut = tf.lut(1024);
    if (tf_ && lut == tfLut_) return;
    // (cpm_tf_update consumes the host LUT on return and does not wait for the stream.  Putting the upload on a stream of its
    // own, beside the importance pass, was measured: no gain on the branch, +10 us of cross-stream latency on a full frame.)
    if (!tf_) rt.check(cpm_tf_create(rt.ctx(), lut.data(), 1024, 0, rt.stream(), &tf_), "cpm_tf_create");
    else rt.check(cpm_tf_update(rt.ctx(), tf_, lut.data(), 0, rt.stream()), "cpm_tf_update");
    tfLut_ = std::move(lut);
    for (auto& lo : launchOrders_) lo.second.stale = true;  // what the launches cost may have changed
    allLightsOrder_.stale = true;
}
void PhotonTracerCL::tracePhotons(const Volume* volume, const TransferFunction& transferFunction, const float aabb[8],
                                  const AdvancedMaterialProperty& material, float stepSize, const LightSamples* lightSamples,
                                  const Buffer<unsigned int>* photonsToRecomputeIndices, int nInvalidPhotons, int photonOffset, int batch,
                                  int maxInteractions, PhotonData* photonOutData) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    seedStreamsFor(*photonOutData);
    cpm_volume* vol_ = volume->getDeviceRepresentation();  // volume->getRepresentation<VolumeCL>() (:111)
    syncTF(transferFunction);
    if (!vol_ || !tf_) return;
    cpm_trace_params p = {};
    const vec4 m = material.getCombinedMaterialParameters();
    p.material[0] = m.x; p.material[1] = m.y; p.material[2] = m.z; p.material[3] = m.w;
    p.step_size = stepSize;
    p.photon_offset = photonOffset;
    p.n_light_samples = (int)lightSamples->getSize();
    p.max_interactions = maxInteractions;
    p.total_photons = (int)photonOutData->getNumberOfPhotons();
    p.shading_type = material.getPhaseFunctionEnum();
    // photontracercl.cpp:198-210, with the += the reference meant (Q5)
    p.flags = (onlyMultipleScattering_ ? CPM_TRACE_NO_SINGLE_SCATTERING : 0) | (progressive_ ? CPM_TRACE_PROGRESSIVE : 0);
    p.iteration = photonOutData->iteration();
    p.batch = batch;
    // a full launch: in the order of this light's measured chunk costs
    LaunchOrder* lo = nullptr;
    bool measure = false;
    if (adaptiveLaunchOrder_ && !photonsToRecomputeIndices && p.n_light_samples > 0) {
        for (auto& e : launchOrders_) if (e.first == lightSamples) lo = &e.second;
        if (!lo) {
            if (launchOrders_.size() >= 16) {  // lights that came and went: start over rather than grow
                for (auto& e : launchOrders_) if (e.second.order) cpm_trace_order_destroy(rt.ctx(), e.second.order);
                launchOrders_.clear();
            }
            launchOrders_.push_back({ lightSamples, LaunchOrder() });
            lo = &launchOrders_.back().second;
        }
        if (lo->order && lo->n != p.n_light_samples) { cpm_trace_order_destroy(rt.ctx(), lo->order); lo->order = nullptr; }
        if (!lo->order) {
            if (!rt.check(cpm_trace_order_create(rt.ctx(), p.n_light_samples, &lo->order), "cpm_trace_order_create")) lo = nullptr;
            else { lo->n = p.n_light_samples; lo->sinceMeasured = 0; }
        }
        if (lo) {
            if (lo->volume != (const void*)vol_) { lo->volume = vol_; lo->stale = true; }
            // (a stale order is still a valid order; while the transfer function is being dragged every launch is "stale", and a
            // measured launch + re-sort costs what ten launches gain)
            measure = lo->sinceMeasured == 0 || lo->sinceMeasured >= kMeasureEvery || (lo->stale && lo->sinceMeasured >= kMeasureAtLeastApart);
            cpm_trace_set_order(rt.ctx(), lo->order, measure ? 1 : 0);
        }
    }
    rt.check(cpm_trace(rt.ctx(), vol_, tf_, nullptr, aabb, &p, reinterpret_cast<const float*>(lightSamples->getLightSamples()->device()),
                       reinterpret_cast<const float*>(lightSamples->getIntersectionPoints()->device()),
                       photonsToRecomputeIndices ? photonsToRecomputeIndices->device() : nullptr, nInvalidPhotons,
                       reinterpret_cast<uint32_t*>(randomState_.device()), reinterpret_cast<float*>(photonOutData->photons_.device()), rt.stream()),
             "cpm_trace");
    if (lo) {
        cpm_trace_set_order(rt.ctx(), nullptr, 0);
        if (measure) {
            rt.check(cpm_trace_order_update(rt.ctx(), lo->order, rt.stream()), "cpm_trace_order_update");
            lo->sinceMeasured = 0;
            lo->stale = false;
        }
        ++lo->sinceMeasured;
    }
}

bool PhotonTracerCL::tracePhotonsAllLights(const Volume* volume, const TransferFunction& transferFunction, const float aabb[8],
                                           const AdvancedMaterialProperty& material, float stepSize, const std::vector<const LightSamples*>& lights,
                                           int maxInteractions, PhotonData* photonOutData) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || lights.size() < 2 || lights.size() > (size_t)CPM_MAX_TRACE_LIGHTS) return false;
    seedStreamsFor(*photonOutData);
    cpm_volume* vol_ = volume->getDeviceRepresentation();
    syncTF(transferFunction);
    if (!vol_ || !tf_) return false;
    cpm_trace_params p = {};
    const vec4 m = material.getCombinedMaterialParameters();
    p.material[0] = m.x; p.material[1] = m.y; p.material[2] = m.z; p.material[3] = m.w;
    p.step_size = stepSize;
    p.max_interactions = maxInteractions;
    p.total_photons = (int)photonOutData->getNumberOfPhotons();
    p.shading_type = material.getPhaseFunctionEnum();
    p.flags = (onlyMultipleScattering_ ? CPM_TRACE_NO_SINGLE_SCATTERING : 0) | (progressive_ ? CPM_TRACE_PROGRESSIVE : 0);
    p.iteration = photonOutData->iteration();
    cpm_light_span spans[CPM_MAX_TRACE_LIGHTS];
    int offset = 0;
    for (size_t l = 0; l < lights.size(); ++l) {
        spans[l].light_samples8 = reinterpret_cast<const float*>(lights[l]->getLightSamples()->device());
        spans[l].isect2 = reinterpret_cast<const float*>(lights[l]->getIntersectionPoints()->device());
        spans[l].n_light_samples = (int)lights[l]->getSize();
        spans[l].photon_offset = offset;
        offset += (int)lights[l]->getSize();
    }
    // the launch's chunks in the order of their measured costs, as for a light's own launch (tracePhotons)
    LaunchOrder* lo = nullptr;
    bool measure = false;
    const int orderSamples = cpm_trace_lights_order_samples(spans, (int)lights.size());
    if (adaptiveLaunchOrder_ && orderSamples > 0) {
        lo = &allLightsOrder_;
        if (lo->order && (lo->n != orderSamples || allLightsOrderFor_ != lights)) { cpm_trace_order_destroy(rt.ctx(), lo->order); *lo = LaunchOrder(); }
        if (!lo->order) {
            if (!rt.check(cpm_trace_order_create(rt.ctx(), orderSamples, &lo->order), "cpm_trace_order_create")) lo = nullptr;
            else { lo->n = orderSamples; lo->sinceMeasured = 0; allLightsOrderFor_ = lights; }
        }
        if (lo) {
            if (lo->volume != (const void*)vol_) { lo->volume = vol_; lo->stale = true; }
            measure = lo->sinceMeasured == 0 || lo->sinceMeasured >= kMeasureEvery || (lo->stale && lo->sinceMeasured >= kMeasureAtLeastApart);
            cpm_trace_set_order(rt.ctx(), lo->order, measure ? 1 : 0);
        }
    }
    const bool ok = rt.check(cpm_trace_lights(rt.ctx(), vol_, tf_, nullptr, aabb, &p, spans, (int)lights.size(), reinterpret_cast<uint32_t*>(randomState_.device()),
                                              reinterpret_cast<float*>(photonOutData->photons_.device()), rt.stream()), "cpm_trace_lights");
    if (lo) {
        cpm_trace_set_order(rt.ctx(), nullptr, 0);
        if (measure && ok) {
            rt.check(cpm_trace_order_update(rt.ctx(), lo->order, rt.stream()), "cpm_trace_order_update");
            lo->sinceMeasured = 0;
            lo->stale = false;
        }
        ++lo->sinceMeasured;
    }
    return ok;
}

int RecomputedPhotonIndices::resolveCount() {
    if (countPending && selection) {
        auto& rt = CpmRuntime::get();
        int32_t n = 0;
        if (rt.check(cpm_selection_count(rt.ctx(), selection, &n), "cpm_selection_count")) nRecomputedPhotons = n;
        countPending = false;
        const size_t total = indicesToRecomputedPhotons.getSize();
        if (costs && total > 0 && n >= 0) costs->sawFraction((float)n / (float)total);
    }
    return nRecomputedPhotons;
}

void PhotonTracerCL::tracePhotonsSelected(const Volume* volume, const TransferFunction& transferFunction, const float aabb[8],
                                          const AdvancedMaterialProperty& material, float stepSize, const LightSamples* lightSamples,
                                          const Buffer<unsigned int>* indices, const int32_t* nIndicesDevice, int maxIndices, vec4* replacedPhotons,
                                          unsigned int* resetImportances, int photonOffset, int maxInteractions, PhotonData* photonOutData) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    seedStreamsFor(*photonOutData);
    cpm_volume* vol_ = volume->getDeviceRepresentation();
    syncTF(transferFunction);
    if (!vol_ || !tf_) return;
    cpm_trace_params p = {};
    const vec4 m = material.getCombinedMaterialParameters();
    p.material[0] = m.x; p.material[1] = m.y; p.material[2] = m.z; p.material[3] = m.w;
    p.step_size = stepSize;
    p.photon_offset = photonOffset;
    p.n_light_samples = (int)lightSamples->getSize();
    p.max_interactions = maxInteractions;
    p.total_photons = (int)photonOutData->getNumberOfPhotons();
    p.shading_type = material.getPhaseFunctionEnum();
    p.flags = (onlyMultipleScattering_ ? CPM_TRACE_NO_SINGLE_SCATTERING : 0) | (progressive_ ? CPM_TRACE_PROGRESSIVE : 0);
    p.iteration = photonOutData->iteration();
    rt.check(cpm_trace_selected(rt.ctx(), vol_, tf_, nullptr, aabb, &p, reinterpret_cast<const float*>(lightSamples->getLightSamples()->device()),
                                reinterpret_cast<const float*>(lightSamples->getIntersectionPoints()->device()), indices->device(), nIndicesDevice,
                                maxIndices, reinterpret_cast<float*>(replacedPhotons), resetImportances,
                                reinterpret_cast<uint32_t*>(randomState_.device()), reinterpret_cast<float*>(photonOutData->photons_.device()), rt.stream()),
             "cpm_trace_selected");
}

bool PhotonTracerCL::importanceRetrace(cpm_selection* selection, const Volume* volume, const ImportanceUniformGrid3D* grid,
                                       const TransferFunction& transferFunction, const float aabb[8], const AdvancedMaterialProperty& material,
                                       float stepSize, const LightSamples* lightSamples, Buffer<unsigned int>& importances, vec4* replacedPhotons,
                                       int photonOffset, int maxInteractions, bool fixExitPoint, PhotonData* photonOutData) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return false;
    seedStreamsFor(*photonOutData);
    onePerPhoton(importances, *photonOutData);
    cpm_volume* vol_ = volume->getDeviceRepresentation();
    syncTF(transferFunction);
    if (!vol_ || !tf_) return false;
    cpm_trace_params p = {};
    const vec4 m = material.getCombinedMaterialParameters();
    p.material[0] = m.x; p.material[1] = m.y; p.material[2] = m.z; p.material[3] = m.w;
    p.step_size = stepSize;
    p.photon_offset = photonOffset;
    p.n_light_samples = (int)lightSamples->getSize();
    p.max_interactions = maxInteractions;
    p.total_photons = (int)photonOutData->getNumberOfPhotons();
    p.shading_type = material.getPhaseFunctionEnum();
    p.flags = onlyMultipleScattering_ ? CPM_TRACE_NO_SINGLE_SCATTERING : 0;  // (with an importance grid connected the tracer is never progressive)
    p.iteration = photonOutData->iteration();
    const size3_t gd = grid->getDimensions(), cd = grid->getCellDimension(), vd = volume->getDimensions();
    const int32_t dims[3] = { (int32_t)gd.x, (int32_t)gd.y, (int32_t)gd.z };
    const float cell[3] = { (float)cd.x, (float)cd.y, (float)cd.z };
    cpm_volume_desc d;
    const int32_t vdims[3] = { (int32_t)vd.x, (int32_t)vd.y, (int32_t)vd.z };
    cpm_volume_desc_default(&d, vdims, volume->dtype());
    // (the grid's occupancy bits came with the grid: no launch of the selection's own for them)
    cpm_selection_set_occupancy(rt.ctx(), selection, grid->occupancyValid ? grid->data.device() : nullptr,
                                grid->occupancyValid ? grid->occupancy.device() : nullptr);
    return rt.check(cpm_photon_importance_retrace(rt.ctx(), selection, grid->data.device(), dims, cell, d.texture_to_index, vol_, tf_, nullptr, aabb, &p,
                                           reinterpret_cast<const float*>(lightSamples->getLightSamples()->device()),
                                           reinterpret_cast<const float*>(lightSamples->getIntersectionPoints()->device()), fixExitPoint ? 1 : 0,
                                           importances.device(), reinterpret_cast<uint32_t*>(randomState_.device()),
                                           reinterpret_cast<float*>(photonOutData->photons_.device()), reinterpret_cast<float*>(replacedPhotons),
                                           rt.stream()),
             "cpm_photon_importance_retrace");
}

int PhotonTracerCL::importanceRetraceAllLights(cpm_selection* selection, const Volume* volume, const ImportanceUniformGrid3D* grid,
                                               const TransferFunction& transferFunction, const float aabb[8], const AdvancedMaterialProperty& material,
                                               float stepSize, const std::vector<const LightSamples*>& lights, Buffer<unsigned int>& importances,
                                               vec4* replacedPhotons, int maxInteractions, bool fixExitPoint, PhotonData* photonOutData) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || lights.size() < 2 || lights.size() > (size_t)CPM_MAX_TRACE_LIGHTS) return -1;
    seedStreamsFor(*photonOutData);
    onePerPhoton(importances, *photonOutData);
    cpm_volume* vol_ = volume->getDeviceRepresentation();
    syncTF(transferFunction);
    if (!vol_ || !tf_) return 0;
    cpm_trace_params p = {};
    const vec4 m = material.getCombinedMaterialParameters();
    p.material[0] = m.x; p.material[1] = m.y; p.material[2] = m.z; p.material[3] = m.w;
    p.step_size = stepSize;
    p.max_interactions = maxInteractions;
    p.total_photons = (int)photonOutData->getNumberOfPhotons();
    p.shading_type = material.getPhaseFunctionEnum();
    p.flags = onlyMultipleScattering_ ? CPM_TRACE_NO_SINGLE_SCATTERING : 0;  // (with an importance grid connected the tracer is never progressive)
    p.iteration = photonOutData->iteration();
    cpm_light_span spans[CPM_MAX_TRACE_LIGHTS];
    int offset = 0;
    for (size_t l = 0; l < lights.size(); ++l) {
        spans[l].light_samples8 = reinterpret_cast<const float*>(lights[l]->getLightSamples()->device());
        spans[l].isect2 = reinterpret_cast<const float*>(lights[l]->getIntersectionPoints()->device());
        spans[l].n_light_samples = (int)lights[l]->getSize();
        spans[l].photon_offset = offset;
        offset += (int)lights[l]->getSize();
    }
    const size3_t gd = grid->getDimensions(), cd = grid->getCellDimension(), vd = volume->getDimensions();
    const int32_t dims[3] = { (int32_t)gd.x, (int32_t)gd.y, (int32_t)gd.z };
    const float cell[3] = { (float)cd.x, (float)cd.y, (float)cd.z };
    cpm_volume_desc d;
    const int32_t vdims[3] = { (int32_t)vd.x, (int32_t)vd.y, (int32_t)vd.z };
    cpm_volume_desc_default(&d, vdims, volume->dtype());
    cpm_selection_set_occupancy(rt.ctx(), selection, grid->occupancyValid ? grid->data.device() : nullptr,
                                grid->occupancyValid ? grid->occupancy.device() : nullptr);
    return rt.check(cpm_photon_importance_retrace_lights(rt.ctx(), selection, grid->data.device(), dims, cell, d.texture_to_index, vol_, tf_, nullptr, aabb, &p,
                                                         spans, (int)lights.size(), fixExitPoint ? 1 : 0, importances.device(),
                                                         reinterpret_cast<uint32_t*>(randomState_.device()),
                                                         reinterpret_cast<float*>(photonOutData->photons_.device()), reinterpret_cast<float*>(replacedPhotons),
                                                         rt.stream()),
                    "cpm_photon_importance_retrace_lights") ? 1 : 0;
}

bool PhotonRecomputationDetector::photonRecomputationImportanceSelect(cpm_selection* selection, const PhotonData* photonData, int photonOffset,
                                                                      const Volume* origVolume, const ImportanceUniformGrid3D* grid,
                                                                      const LightSamples& lightSamples, Buffer<unsigned int>& imp, bool fixExitPoint) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return false;
    onePerPhoton(imp, *photonData);
    if (getEqualImportance()) {
        return rt.check(cpm_photon_importance_equal_select(rt.ctx(), selection, photonOffset, (int)lightSamples.getSize(), getPercentage(), getIteration(),
                                                           imp.device(), rt.stream()), "cpm_photon_importance_equal_select");
    }
    const size3_t gd = grid->getDimensions(), cd = grid->getCellDimension(), vd = origVolume->getDimensions();
    const int32_t dims[3] = { (int32_t)gd.x, (int32_t)gd.y, (int32_t)gd.z };
    const float cell[3] = { (float)cd.x, (float)cd.y, (float)cd.z };
    cpm_volume_desc d;
    const int32_t vdims[3] = { (int32_t)vd.x, (int32_t)vd.y, (int32_t)vd.z };
    cpm_volume_desc_default(&d, vdims, origVolume->dtype());
    cpm_selection_set_occupancy(rt.ctx(), selection, grid->occupancyValid ? grid->data.device() : nullptr,
                                grid->occupancyValid ? grid->occupancy.device() : nullptr);
    return rt.check(cpm_photon_importance_select(rt.ctx(), selection, grid->data.device(), dims, cell, d.texture_to_index,
                                          reinterpret_cast<const float*>(photonData->photons_.device()), photonOffset,
                                          reinterpret_cast<const float*>(lightSamples.getLightSamples()->device()),
                                          reinterpret_cast<const float*>(lightSamples.getIntersectionPoints()->device()), (int)lightSamples.getSize(),
                                          photonData->getMaxPhotonInteractions(), (int)photonData->getNumberOfPhotons(), fixExitPoint ? 1 : 0, imp.device(),
                                          rt.stream()),
             "cpm_photon_importance_select");
}

void PhotonRecomputationDetector::photonRecomputationImportance(const PhotonData* photonData, int photonOffset, const Volume* origVolume,
                                                                const ImportanceUniformGrid3D* grid, const LightSamples& lightSamples,
                                                                Buffer<unsigned int>& imp) {
    auto& rt = CpmRuntime::get();  // photonrecomputationdetector.cpp:49-121 (size check fixed: Q11)
    if (!rt.valid()) return;
    onePerPhoton(imp, *photonData);
    if (getEqualImportance()) {
        rt.check(cpm_photon_importance_equal(rt.ctx(), photonOffset, (int)lightSamples.getSize(), getPercentage(), getIteration(), imp.device(), rt.stream()),
                 "cpm_photon_importance_equal");
        return;
    }
    const size3_t gd = grid->getDimensions(), cd = grid->getCellDimension(), vd = origVolume->getDimensions();
    const int32_t dims[3] = { (int32_t)gd.x, (int32_t)gd.y, (int32_t)gd.z };
    const float cell[3] = { (float)cd.x, (float)cd.y, (float)cd.z };
    cpm_volume_desc d;
    const int32_t vdims[3] = { (int32_t)vd.x, (int32_t)vd.y, (int32_t)vd.z };
    cpm_volume_desc_default(&d, vdims, origVolume->dtype());
    rt.check(cpm_photon_importance(rt.ctx(), grid->data.device(), dims, cell, d.texture_to_index,
                                   reinterpret_cast<const float*>(photonData->photons_.device()), photonOffset,
                                   reinterpret_cast<const float*>(lightSamples.getLightSamples()->device()),
                                   reinterpret_cast<const float*>(lightSamples.getIntersectionPoints()->device()), (int)lightSamples.getSize(),
                                   photonData->getMaxPhotonInteractions(), (int)photonData->getNumberOfPhotons(), 0, imp.device(), rt.stream()),
             "cpm_photon_importance");
}

// ---- processors ----------------------------------------------------------------------------------------------

UniformSampleGenerator2DProcessorCL::UniformSampleGenerator2DProcessorCL() {
    addPortId("samples", false); addPortId("DirectionalSamples", false);
    addProperty(nSamplesProp_); addProperty(workGroupSize_); addProperty(useGLSharing_);
}
void UniformSampleGenerator2DProcessorCL::process() {  // uniformsamplegenerator2dprocessorcl.cpp:77-96
    generator_.generateNextSamples(*samples_, nSamplesProp_.get());
    samplesPort_.setData(samples_);
    if (directionalSamplesPort_.isConnected()) {  // uniformsamplegenerator2dcl.cpp:79-101: a device copy of the position samples
        directionalSamples_->setSize(samples_->getSize());
        (void)hipMemcpyAsync(directionalSamples_->device(), samples_->device(), samples_->getSizeInBytes(), hipMemcpyDeviceToDevice,
                             CpmRuntime::get().stream());
    } else if (directionalSamples_->getSize() != 0) {
        directionalSamples_->setSize(0);
    }
    directionalSamplesPort_.setData(directionalSamples_);
}

DirectionalLightSamplerCLProcessor::DirectionalLightSamplerCLProcessor() {
    addPortId("SceneGeometry", true); addPortId("samples", true); addPortId("light", true); addPortId("LightSamples", false);
    addProperty(workGroupSize_);
}
void DirectionalLightSamplerCLProcessor::process() {  // directionallightsamplerclprocessor.cpp:79-89
    if (!boundingVolumePort_.isReady() || !samplesPort_.isReady() || !lightsPort_.isReady()) return;
    lightSamples_->resetIteration();  // the light / samples changed
    lightSampler_.sampleLightSource(boundingVolumePort_.getData().get(), samplesPort_.getData().get(), lightsPort_.getData().get(), *lightSamples_);
    intersector_.meshSampleIntersection(boundingVolumePort_.getData().get(), lightSamples_.get());
    lightSamplesPort_.setData(lightSamples_);
}

VolumeMinMaxCLProcessor::VolumeMinMaxCLProcessor() {
    addPortId("volume", true); addPortId("output", false);
    addPortId("VolumeSequenceInput", true); addPortId("UniformGrid3DVectorOut", false);
    inport_.setOptional(true); vectorInport_.setOptional(true);
    addProperty(volumeRegionSize_); addProperty(workGroupSize_); addProperty(useGLSharing_);
}
void VolumeMinMaxCLProcessor::process() {  // volumeminmaxclprocessor.cpp:88-122
    if (!CpmRuntime::get().valid()) return;
    if (vectorInport_.isReady()) {
        outport_.setData(nullptr);
        auto output = std::make_shared<UniformGrid3DVector>();
        for (auto& elem : *vectorInport_.getData())
            if (auto result = compute(elem.get())) output->emplace_back(std::move(result));
        vectorOutport_.setData(output);
    }
    if (inport_.isReady())
        if (auto result = compute(inport_.getData().get())) outport_.setData(result);
}
std::shared_ptr<MinMaxUniformGrid3D> VolumeMinMaxCLProcessor::compute(const Volume* volume) {  // :148-184
    auto& rt = CpmRuntime::get();
    cpm_volume* vol = volume->getDeviceRepresentation();
    if (!vol) return nullptr;
    const size_t r = (size_t)volumeRegionSize_.get();
    const size3_t s = volume->getDimensions();
    auto grid = std::make_shared<MinMaxUniformGrid3D>();
    grid->setCellDimension(size3_t{ r, r, r });
    grid->setModelMatrix(volume->getModelMatrix());
    grid->setWorldMatrix(volume->getWorldMatrix());
    grid->setDimensions(size3_t{ (s.x + r - 1) / r, (s.y + r - 1) / r, (s.z + r - 1) / r });
    if (!rt.check(cpm_volume_minmax(rt.ctx(), vol, (int)r, grid->data.device(), rt.stream()), "cpm_volume_minmax")) return nullptr;
    return grid;
}

// ---- Volume device representation -------------------------------------------------------------------
Volume::~Volume() { invalidateDeviceRepresentation(); }
void Volume::invalidateDeviceRepresentation() {
    if (dev_) cpm_volume_destroy(CpmRuntime::get().ctx(), dev_);
    dev_ = nullptr;
}
cpm_volume* Volume::getDeviceRepresentation() const {
    if (dev_) return dev_;
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return nullptr;
    cpm_volume_desc d;
    const int32_t dims[3] = { (int32_t)dims_.x, (int32_t)dims_.y, (int32_t)dims_.z };
    cpm_volume_desc_default(&d, dims, dtype_);
    const size_t bytes = dims_.x * dims_.y * dims_.z * elementSize();
    const void* src = ramBytes.size() == bytes ? ramBytes.data() : nullptr;
    if (!rt.check(cpm_volume_create(rt.ctx(), &d, src, 0, rt.stream(), &dev_), "cpm_volume_create")) dev_ = nullptr;
    return dev_;
}
bool Volume::downloadToRAM() {
    if (!dev_) return !ramBytes.empty();
    auto& rt = CpmRuntime::get();
    ramBytes.resize(dims_.x * dims_.y * dims_.z * elementSize());
    return rt.check(cpm_volume_download(rt.ctx(), dev_, ramBytes.data(), rt.stream()), "cpm_volume_download");
}

MinMaxUniformGrid3DImportanceCLProcessor::MinMaxUniformGrid3DImportanceCLProcessor() {
    addPortId("minMaxUniformGrid3D", true); addPortId("volumeDifferenceInfo", true); addPortId("importanceUniformGrid3D", false);
    addProperty(incrementalImportance); addProperty(useAssociatedColor_); addProperty(TFPointEpsilon_);
    for (PropertyBase* q : std::initializer_list<PropertyBase*>{ &opacityWeight_, &opacityDiffWeight_, &colorWeight_, &colorDiffWeight_,
                                                                &transferFunctionProperty_, &workGroupSize_, &useGLSharing_ })
        addProperty(*q);
    transferFunctionProperty_.onChange([this]() { setTransferFunction(transferFunctionProperty_.get()); });  // .cpp:94-96
}
void MinMaxUniformGrid3DImportanceCLProcessor::setTransferFunction(const TransferFunction& tf) {
    transferFunction_ = tf;
    tfChanged_ = true;
}
void MinMaxUniformGrid3DImportanceCLProcessor::updateTransferFunctionData() {
    positions_.clear(); colors_.clear();
    const auto& tf = transferFunction_;
    auto col = [&](const TFPrimitive& p) { return useAssociatedColor_ ? vec4(p.color.x * p.color.w, p.color.y * p.color.w, p.color.z * p.color.w, p.color.w * p.color.w) : p.color; };
    if (tf.size() == 0) { positions_ = { 0.f, 1.f }; colors_ = { vec4(), vec4() }; return; }
    if (tf.get(0).pos > 0.) { positions_.push_back(0.f); colors_.push_back(col(tf.get(0))); }
    for (size_t i = 0; i < tf.size(); ++i) { positions_.push_back((float)tf.get(i).pos); colors_.push_back(col(tf.get(i))); }
    if (tf.get(tf.size() - 1).pos < 1.) { positions_.push_back(1.f); colors_.push_back(col(tf.get(tf.size() - 1))); }
}
void MinMaxUniformGrid3DImportanceCLProcessor::updateTransferFunctionDifferenceData() {
    // minmaxuniformgrid3dimportanceclprocessor.cpp:364-501, through this build's own statement of it (cpm_hostmath.cpp, rule D)
    auto nodes = [](const TransferFunction& f) {
        std::vector<cpm_host::TfNode> v(f.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const TFPrimitive& q = f.get(i);
            v[i].pos = q.pos;
            const float c[4] = { q.color.x, q.color.y, q.color.z, q.color.w };
            std::memcpy(v[i].rgba, c, sizeof c);
        }
        return v;
    };
    cpm_host::TfBreakpoints list;
    if (!cpm_host::tfDifference(nodes(transferFunction_), nodes(prevTransferFunction_), TFPointEpsilon_.get(), useAssociatedColor_, list)) {
        updateTransferFunctionData();  // one of the two is empty: no difference function, classify with the function itself
        return;
    }
    positions_ = list.pos;
    colors_.resize(list.size());
    for (size_t i = 0; i < list.size(); ++i) colors_[i] = vec4(list.rgba[4 * i], list.rgba[4 * i + 1], list.rgba[4 * i + 2], list.rgba[4 * i + 3]);
}
void MinMaxUniformGrid3DImportanceCLProcessor::process() {  // minmaxuniformgrid3dimportanceclprocessor.cpp:110-216
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || !minMaxUniformGrid3DInport_.isReady()) return;
    auto minMax = std::dynamic_pointer_cast<MinMaxUniformGrid3D>(minMaxUniformGrid3DInport_.getData());
    if (!minMax) { LogError("minMaxUniformGrid3DInport_ expects MinMaxUniformGrid3D as input"); return; }
    const size3_t d = minMax->getDimensions();
    const size3_t cur = importance_->getDimensions();
    if (cur.x != d.x || cur.y != d.y || cur.z != d.z) { importance_->setDimensions(d); importance_->setCellDimension(minMax->getCellDimension()); }
    const bool volumeChanged = minMaxUniformGrid3DInport_.changedSinceLastCheck();  // minMaxUniformGrid3DInport_.onChange (:73-74)
    if (tfChanged_) {
        if (prevTransferFunction_.size() == 0 || !incrementalImportance) updateTransferFunctionData();
        else updateTransferFunctionDifferenceData();
        prevTransferFunction_ = transferFunction_;
        tfChanged_ = false;
    } else if (volumeChanged || positions_.empty()) {  // InvalidationReason::Volume (:134-136): importance of a range = the TF itself
        updateTransferFunctionData();
    }
    const int nElements = (int)(d.x * d.y * d.z);
    importance_->occupancy.setSize(2 * (((size_t)nElements + 63) / 64));
    importance_->occupancyValid = false;
    if (volumeDifferenceInfoInport_.isReady() && prevMinMaxUniformGrid3D_ != nullptr && prevMinMaxUniformGrid3D_.get() != minMax.get()) {
        // time-varying data changed (:149-190): importance x mean |v_(t+1) - v_t| over the union of the old and new brick ranges
        auto diff = std::dynamic_pointer_cast<DynamicVolumeInfoUniformGrid3D>(volumeDifferenceInfoInport_.getData());
        if (!diff) { LogError("volumeDifferenceInfoInport_ expects DynamicVolumeInfoUniformGrid3D as input"); return; }
        if (!diff->hasDeviceData()) diff->uploadHostData();
        auto* prev = const_cast<MinMaxUniformGrid3D*>(prevMinMaxUniformGrid3D_.get());
        importance_->occupancyValid =
            rt.check(cpm_importance_tf_occupancy(rt.ctx(), minMax->data.device(), prev->data.device(), diff->data.device(), nElements, positions_.data(),
                                                 reinterpret_cast<const float*>(colors_.data()), (int)positions_.size(), importance_->data.device(),
                                                 importance_->occupancy.device(), rt.stream()),
                     "cpm_importance_tf(time-varying)");
    } else {
        importance_->occupancyValid =
            rt.check(cpm_importance_tf_occupancy(rt.ctx(), minMax->data.device(), nullptr, nullptr, nElements, positions_.data(),
                                                 reinterpret_cast<const float*>(colors_.data()), (int)positions_.size(), importance_->data.device(),
                                                 importance_->occupancy.device(), rt.stream()),
                     "cpm_importance_tf");
    }
    prevMinMaxUniformGrid3D_ = minMax;  // :212-213
    importanceUniformGrid3DOutport_.setData(importance_);
}

ProgressivePhotonTracerCL::ProgressivePhotonTracerCL() {
    for (const char* id : { "volume", "recomputationImportance", "LightSamples" }) addPortId(id, true);
    for (const char* id : { "photons", "recomputedIndices" }) addPortId(id, false);
    recomputationImportanceGrid_.setOptional(true);
    recomputationImportanceGrid_.onConnect([this]() { invalidateProgressiveRendering(PhotonData::InvalidationReason::All); });
    for (PropertyBase* p : std::initializer_list<PropertyBase*>{ &samplingRate_, &radius_, &sceneRadianceScaling_, &maxIncrementalPhotonsToUpdate_,
                                                                &equalIncrementalImportance_, &spatialSorting_, &maxScatteringEvents_, &noSingleScattering_,
                                                                &alphaProp_, &workGroupSize_, &useGLSharing_, &enableProgressiveRefinement_,
                                                                &enableProgressivePhotonRecomputation_, &clipX_, &clipY_, &clipZ_, &fusedImportanceBranch_,
                                                                &equalImportancePercentage_, &importanceBranchPolicy_, &retraceInImportancePass_, &traceLightsInOneLaunch_ })
        addProperty(*p);
    addProperty(advancedMaterial_); addProperty(camera_); addProperty(invalidateRendering_); addProperty(transferFunctionProperty_);
    // what an edit of each property invalidates (progressivephotontracercl.cpp:137-188): the camera only the view-dependent part,
    // material / scattering depth / the "invalidate" button and the clip ranges everything
    using Why = PhotonData::InvalidationReason;
    auto invalidates = [this](PropertyBase& edited, Why why) { edited.onChange([this, why]() { invalidateProgressiveRendering(why); }); };
    invalidates(camera_, Why::Camera);
    for (PropertyBase* edited : std::initializer_list<PropertyBase*>{ &invalidateRendering_, &advancedMaterial_, &maxScatteringEvents_ }) invalidates(*edited, Why::All);
    for (PropertyBase* range : std::initializer_list<PropertyBase*>{ &clipX_, &clipY_, &clipZ_ }) range->onChange(std::bind(&ProgressivePhotonTracerCL::onClipChange, this));
    transferFunctionProperty_.onChange([this]() { setTransferFunction(transferFunctionProperty_.get()); });
    equalIncrementalImportance_.onChange([this]() { photonRecomputationDetector_.setEqualImportance(equalIncrementalImportance_.get()); });
    noSingleScattering_.onChange([this]() { photonTracer_.setNoSingleScattering(noSingleScattering_.get()); });
}
ProgressivePhotonTracerCL::~ProgressivePhotonTracerCL() {
    if (selection_) cpm_selection_destroy(CpmRuntime::get().ctx(), selection_);
}
// the end of every evaluation: both outports carry their data, the reasons collected since the last one travel with the photons
void ProgressivePhotonTracerCL::publishPhotons() {
    recomputedIndicesPort_.setData(recomputedPhotonIndices_);
    photonData_->setInvalidationReason(invalidationFlag_);
    invalidationFlag_ = PhotonData::InvalidationReason(0);
    outport_.setData(photonData_);
}
void ProgressivePhotonTracerCL::onClipChange() {  // progressivephotontracercl.cpp:672-686
    if (!volumePort_.isReady()) return;
    const size3_t d = volumePort_.getData()->getDimensions();
    aabb_[0] = (float)clipX_.get().x / (float)d.x; aabb_[1] = (float)clipY_.get().x / (float)d.y; aabb_[2] = (float)clipZ_.get().x / (float)d.z; aabb_[3] = 1.f;
    aabb_[4] = (float)clipX_.get().y / (float)d.x; aabb_[5] = (float)clipY_.get().y / (float)d.y; aabb_[6] = (float)clipZ_.get().y / (float)d.z; aabb_[7] = 1.f;
    invalidateProgressiveRendering(PhotonData::InvalidationReason::All);
}
void ProgressivePhotonTracerCL::resetPhotonImportance(size_t offset, size_t n) {  // :607-611
    auto& rt = CpmRuntime::get();
    const bool whole = offset == 0 && n == photonRecomputationImportance_.getSize();
    // (a full frame resets every key: nothing to do when no importance pass has touched them since the last such reset --
    // the case of every frame served without the branch; the reference fills the buffer each time)
    if (whole && importancesAreReset_) return;
    rt.check(cpm_reset_importance(rt.ctx(), photonRecomputationImportance_.device(), offset, n, rt.stream()), "cpm_reset_importance");
    if (whole) importancesAreReset_ = true;
}
void ProgressivePhotonTracerCL::process() {  // progressivephotontracercl.cpp:219-605
    auto& rt = CpmRuntime::get();
    if (!photonTracer_.isValid() || !volumePort_.isReady()) return;
    rt.beginProfile();
    span_.poll();
    photonTracer_.syncTF(transferFunction_);  // the LUT upload goes first: nothing waits for it later
    if (volumePort_.changedSinceLastCheck()) invalidateProgressiveRendering(PhotonData::InvalidationReason::Volume);  // volumePort_.onChange (:107-108)
    recomputedPhotonIndices_->costs = &costs_;
    recomputedPhotonIndices_->keepsReplaced = fusedImportanceBranch_.get() && maxIncrementalPhotonsToUpdate_.get() >= 100.f &&
                                              recomputationImportanceGrid_.isConnected();
    const auto lights = lightSamples_.getVectorData();
    size_t nPhotons = 0;
    // lightSamples_.onChange (tracercl.cpp:119-126): a light whose samples were rewritten since the last
    // evaluation and that reports isReset() invalidates with reason Light
    std::vector<std::pair<const LightSamples*, size_t>> now;
    for (auto& l : lights) {
        nPhotons += l->getSize();
        now.emplace_back(l.get(), l->changeStamp);
        bool seen = false;
        for (auto& s : seenLights_) seen |= (s.first == l.get() && s.second == l->changeStamp);
        if (!seen && l->isReset()) invalidateProgressiveRendering(PhotonData::InvalidationReason::Light);
    }
    seenLights_ = std::move(now);
    if (nPhotons != photonData_->getNumberOfPhotons() || maxScatteringEvents_.get() != photonData_->getMaxPhotonInteractions()) {
        photonData_->setSize(nPhotons, maxScatteringEvents_.get());
        costs_ = PathCosts();  // another problem size: measure again
        invalidateProgressiveRendering(PhotonData::InvalidationReason::All);
    }
    const Volume* volume = volumePort_.getData().get();
    const size3_t vd = volume->getDimensions();
    const float sceneRadius = getSceneRadius();
    const float spacing = std::min(1.f / (float)vd.x, std::min(1.f / (float)vd.y, 1.f / (float)vd.z));
    const float stepSize = samplingRate_.get() * spacing;
    const int maxInteractions = maxScatteringEvents_.get();
    const int flag = static_cast<int>(invalidationFlag_);
    const int lightFlag = static_cast<int>(PhotonData::InvalidationReason::Light), tfFlag = static_cast<int>(PhotonData::InvalidationReason::TransferFunction),
              volFlag = static_cast<int>(PhotonData::InvalidationReason::Volume), camFlag = static_cast<int>(PhotonData::InvalidationReason::Camera);
    if (flag == 0 || (flag & (lightFlag | camFlag | tfFlag | volFlag))) photonData_->resetIteration();
    if (photonData_->iteration() == 0) {
        const float r = radius_.get();
        const float rx = r / (float)vd.x, ry = r / (float)vd.y, rz = r / (float)vd.z;  // indexToTexture * (r, r, r, 0)
        photonData_->setRadius(std::sqrt(rx * rx + ry * ry + rz * rz), sceneRadius);
        photonData_->setIteration(1);
    } else {
        photonData_->advanceToNextIteration(alphaProp_.get());
    }
    // a full frame: every light's samples traced -- in one launch where there are several lights, else (or should that launch be
    // refused) light by light as the reference does (:543-549)
    auto traceAllLights = [&]() {
        if (lights.size() > 1 && traceLightsInOneLaunch_.get()) {
            std::vector<const LightSamples*> all;
            for (auto& l : lights) all.push_back(l.get());
            if (photonTracer_.tracePhotonsAllLights(volume, transferFunction_, aabb_, advancedMaterial_, stepSize, all, maxInteractions, photonData_.get())) return;
        }
        int offset = 0;
        for (auto& l : lights) {
            photonTracer_.tracePhotons(volume, transferFunction_, aabb_, advancedMaterial_, stepSize, l.get(), nullptr, 0, offset, 0, maxInteractions,
                                       photonData_.get());
            offset += (int)l->getSize();
        }
    };
    size_t nPhotonsToCompute = photonData_->getNumberOfPhotons();
    if (!(flag & lightFlag) && recomputationImportanceGrid_.isReady() && photonRecomputationDetector_.isValid()) {
        if (photonRecomputationImportance_.getSize() != photonData_->getNumberOfPhotons()) {
            photonRecomputationImportance_.setSize(photonData_->getNumberOfPhotons());
            importancesAreReset_ = false;
            resetPhotonImportance(0, photonRecomputationImportance_.getSize());
        }
        if (recomputedPhotonIndices_->indicesToRecomputedPhotons.getSize() != photonData_->getNumberOfPhotons())
            recomputedPhotonIndices_->indicesToRecomputedPhotons.setSize(photonData_->getNumberOfPhotons());
        recomputedPhotonIndices_->replacedValid = false;
        recomputedPhotonIndices_->countPending = false;
        // The whole branch with the count kept on the device (cpm.h, "the correlated update without a host round trip"):
        // possible when every changed photon is traced in this evaluation -- a budget below 100 % needs the ranking by
        // importance, a host decision on the count.
        bool fused = fusedImportanceBranch_.get() && (flag & (tfFlag | volFlag)) && maxIncrementalPhotonsToUpdate_.get() >= 100.f;
        // The branch exists to avoid re-tracing everything; where its measured cost (importance pass over ALL photons, re-trace,
        // add-remove) exceeds the full frame's, take the frame: the correlated RNG streams make the photons the same either way.
        bool takeFullFrame = false;
        if (fused && importanceBranchPolicy_.get() == "never") {
            takeFullFrame = true;
        } else if (fused && importanceBranchPolicy_.get() == "adaptive") {
            auto& c = costs_;
            const bool fullFrameCheaper = c.known() && c.branchTraceMs() + c.branchLightVolumeMs() > c.fullTraceMs + c.fullLightVolumeMs;
            c.probing = false;
            if (fullFrameCheaper && c.evaluationsSinceProbe < 32) {
                takeFullFrame = true;
                ++c.evaluationsSinceProbe;
            } else {
                c.probing = fullFrameCheaper;   // the 33rd such evaluation: the branch is taken to see what it costs now
                c.evaluationsSinceProbe = 0;
            }
        }
        // the TF / volume change served by a full frame in place of the importance branch (the policy's choice, or the branch failed)
        auto fullFrameInPlaceOfBranch = [&](const char* decision) {
            lastDecision_ = decision;
            span_.begin(rt.stream(), &costs_.fullTraceMs);
            traceAllLights();
            resetPhotonImportance(0, photonRecomputationImportance_.getSize());
            span_.end(rt.stream());
            recomputedPhotonIndices_->nRecomputedPhotons = -1;
            recomputedPhotonIndices_->takenInPlaceOfBranch = true;
            recomputedPhotonIndices_->countPending = false;
            recomputedPhotonIndices_->replacedValid = false;
            remainingPhotonsToUpdate_ = 0;
            remainingPhotonsOffset_ = 0;
            enableProgressiveRefinement_.set(false);
            publishPhotons();
            if (rt.profiling()) rt.logProfile("Photon tracing");
        };
        if (takeFullFrame) {
            fused = false;
            fullFrameInPlaceOfBranch(importanceBranchPolicy_.get() == "never" ? "full frame (policy)" : "full frame (measured cheaper than the importance branch)");
            return;
        }
        recomputedPhotonIndices_->takenInPlaceOfBranch = false;
        if (fused) {
            lastDecision_ = "importance branch";
            span_.begin(rt.stream(), &costs_.branchTraceMs(), costs_.probing);
            auto grid = std::dynamic_pointer_cast<ImportanceUniformGrid3D>(recomputationImportanceGrid_.getData());
            if (!grid) { LogError("UniformGrid3DInport require ImportanceUniformGrid3D as input"); return; }
            const size_t N = photonData_->getNumberOfPhotons();
            if (!selection_ || selectionPhotons_ != N) {
                if (selection_) cpm_selection_destroy(rt.ctx(), selection_);
                selection_ = nullptr;
                if (!rt.check(cpm_selection_create(rt.ctx(), N, &selection_), "cpm_selection_create")) return;
                selectionPhotons_ = N;
            }
            auto& rec = *recomputedPhotonIndices_;
            if (rec.replacedPhotons.getSize() != photonData_->photons_.getSize()) rec.replacedPhotons.setSize(photonData_->photons_.getSize());
            photonRecomputationDetector_.setPercentage(equalImportancePercentage_.get() > 0 ? equalImportancePercentage_.get() : (int)maxIncrementalPhotonsToUpdate_.get());
            photonRecomputationDetector_.setIteration(photonRecomputationDetector_.getIteration() + 1);
            rt.check(cpm_selection_begin(rt.ctx(), selection_), "cpm_selection_begin");
            importancesAreReset_ = false;  // the importance pass below lowers the keys
            const bool oneLaunch = retraceInImportancePass_.get() && !photonRecomputationDetector_.getEqualImportance() &&
                                   !photonTracer_.isProgressive();
            int offset = 0;
            // A selection one of whose launches failed publishes a count of 0 (cpm_selection_finish reports it): nothing behind
            // it on the stream re-traces or splats with indices no kernel wrote, and the change is served by a full frame.
            bool selected = true;
            int allLights = -1;  // every light's importance pass + re-trace in ONE launch where there are several lights
            if (oneLaunch && lights.size() > 1 && traceLightsInOneLaunch_.get()) {
                std::vector<const LightSamples*> all;
                for (auto& l : lights) all.push_back(l.get());
                allLights = photonTracer_.importanceRetraceAllLights(selection_, volume, grid.get(), transferFunction_, aabb_, advancedMaterial_, stepSize, all,
                                                                     photonRecomputationImportance_, rec.replacedPhotons.device(), maxInteractions, fixExitPoint,
                                                                     photonData_.get());
                if (allLights == 0) selected = false;
            }
            if (oneLaunch) {
                for (auto& l : lights) {  // detector + threshold + tracer + importance reset of a light in one launch (:298-356,467-529)
                    if (allLights >= 0) break;
                    selected &= photonTracer_.importanceRetrace(selection_, volume, grid.get(), transferFunction_, aabb_, advancedMaterial_, stepSize, l.get(),
                                                                photonRecomputationImportance_, rec.replacedPhotons.device(), offset, maxInteractions, fixExitPoint,
                                                                photonData_.get());
                    offset += (int)l->getSize();
                }
                selected &= rt.check(cpm_selection_finish(rt.ctx(), selection_, rec.indicesToRecomputedPhotons.device(), rt.stream()), "cpm_selection_finish");
            } else {
                for (auto& l : lights) {  // detector + threshold + count + index lists, per light (:298-356)
                    selected &= photonRecomputationDetector_.photonRecomputationImportanceSelect(selection_, photonData_.get(), offset, volume, grid.get(), *l,
                                                                                                 photonRecomputationImportance_, fixExitPoint);
                    offset += (int)l->getSize();
                }
                selected &= rt.check(cpm_selection_finish(rt.ctx(), selection_, rec.indicesToRecomputedPhotons.device(), rt.stream()), "cpm_selection_finish");
            }
            if (!selected) {
                span_.end(rt.stream());
                costs_ = PathCosts();  // (that span measured a failure)
                fullFrameInPlaceOfBranch("full frame (the importance branch failed)");
                return;
            }
            if (!oneLaunch) {
                offset = 0;
                for (auto& l : lights) {  // ascending indices = emission-lattice order (:467-473); importance reset in the same launch (:529)
                    photonTracer_.tracePhotonsSelected(volume, transferFunction_, aabb_, advancedMaterial_, stepSize, l.get(), &rec.indicesToRecomputedPhotons,
                                                       cpm_selection_count_device(selection_), (int)N, rec.replacedPhotons.device(),
                                                       photonRecomputationImportance_.device(), offset, maxInteractions, photonData_.get());
                    offset += (int)l->getSize();
                }
            }
            span_.end(rt.stream());
            rec.selection = selection_;
            rec.countPending = true;
            rec.nRecomputedPhotons = 0;  // resolved on first use (resolveCount)
            rec.replacedStride = oneLaunch ? 0 : (int)N;  // 0: the replaced records sit at the photons' own indices
            rec.replacedValid = true;
            rankedByImportance_ = false;
            remainingPhotonsOffset_ = 0;
            remainingPhotonsToUpdate_ = 0;  // everything changed was traced
            enableProgressiveRefinement_.set(false);
            publishPhotons();
            if (rt.profiling()) {
                rt.logProfile("Photon tracing");
                const int nr = rec.resolveCount();
                char buf[96];
                std::snprintf(buf, sizeof buf, "Computed photons: %d = %.2f %%", nr, N ? 100.0 * (double)nr / (double)N : 0.0);
                LogInfo(buf);
            }
            return;
        }
        if (flag & (tfFlag | volFlag)) {
            auto grid = std::dynamic_pointer_cast<ImportanceUniformGrid3D>(recomputationImportanceGrid_.getData());
            if (!grid) { LogError("UniformGrid3DInport require ImportanceUniformGrid3D as input"); return; }
            photonRecomputationDetector_.setPercentage(equalImportancePercentage_.get() > 0 ? equalImportancePercentage_.get() : (int)maxIncrementalPhotonsToUpdate_.get());
            photonRecomputationDetector_.setIteration(photonRecomputationDetector_.getIteration() + 1);
            int offset = 0;
            importancesAreReset_ = false;
            for (auto& l : lights) {
                photonRecomputationDetector_.photonRecomputationImportance(photonData_.get(), offset, volume, grid.get(), *l, photonRecomputationImportance_);
                offset += (int)l->getSize();
            }
            // threshold + count + iota as one stable partition (changed photons first, ascending); the count is read
            // once (Q10).  The 31-bit sort by importance (sortIndicesByImportance, :358-363) runs only when the changed
            // photons exceed this evaluation's budget -- otherwise all of them are traced now, in index order.
            rt.check(cpm_select_changed(rt.ctx(), photonRecomputationImportance_.device(), photonRecomputationImportance_.getSize(),
                                        recomputedPhotonIndices_->indicesToRecomputedPhotons.device(), nChanged_.device(), rt.stream()),
                     "cpm_select_changed");
            nChanged_.download(rt.stream());
            const int nPhotonsToRecompute = nChanged_.ram()[0];
            const int budget = (int)((maxIncrementalPhotonsToUpdate_.get() / 100.f) * (float)photonData_->getNumberOfPhotons());
            rankedByImportance_ = nPhotonsToRecompute > budget;
            if (rankedByImportance_)
                rt.check(cpm_select_recompute(rt.ctx(), photonRecomputationImportance_.device(), photonRecomputationImportance_.getSize(),
                                              recomputedPhotonIndices_->indicesToRecomputedPhotons.device(), nChanged_.device(), rt.stream()),
                         "cpm_select_recompute");
            remainingPhotonsOffset_ = 0;
            if (remainingPhotonsToUpdate_ < 0 || nPhotonsToRecompute > 0) remainingPhotonsToUpdate_ = nPhotonsToRecompute;
        }
        const int maxPhotonsToUpdate = (int)((maxIncrementalPhotonsToUpdate_.get() / 100.f) * (float)photonData_->getNumberOfPhotons());
        nPhotonsToCompute = (size_t)std::max(0, std::min(remainingPhotonsToUpdate_, maxPhotonsToUpdate));
        unsigned int* idx = recomputedPhotonIndices_->indicesToRecomputedPhotons.device();
        if (remainingPhotonsOffset_ > 0 && nPhotonsToCompute > 0)  // the reference's overlap-safe block move (:389-419)
            (void)hipMemcpyAsync(idx, idx + remainingPhotonsOffset_, nPhotonsToCompute * sizeof(unsigned int), hipMemcpyDeviceToDevice, rt.stream());
        recomputedPhotonIndices_->nRecomputedPhotons = (int)nPhotonsToCompute;
        if (recomputedPhotonIndices_->nRecomputedPhotons > 0) {
            if (spatialSorting_.get() && rankedByImportance_)  // ascending index = emission-lattice order (:467-473)
                rt.check(cpm_sort_keys(rt.ctx(), idx, nPhotonsToCompute, 0, rt.stream()), "cpm_sort_keys");
            int offset = 0;
            for (auto& l : lights) {
                photonTracer_.tracePhotons(volume, transferFunction_, aabb_, advancedMaterial_, stepSize, l.get(),
                                           &recomputedPhotonIndices_->indicesToRecomputedPhotons, recomputedPhotonIndices_->nRecomputedPhotons, offset, 0,
                                           maxInteractions, photonData_.get());
                offset += (int)l->getSize();
            }
            if (rankedByImportance_) resetPhotonImportance((size_t)remainingPhotonsOffset_, nPhotonsToCompute);
            else resetPhotonImportance(0, photonRecomputationImportance_.getSize());  // every changed photon was traced
        }
        remainingPhotonsOffset_ += (int)nPhotonsToCompute;
        remainingPhotonsToUpdate_ -= (int)nPhotonsToCompute;
        enableProgressiveRefinement_.set(remainingPhotonsToUpdate_ > 0 && enableProgressivePhotonRecomputation_.get());
    } else {
        lastDecision_ = "full frame";
        span_.begin(rt.stream(), &costs_.fullTraceMs);
        traceAllLights();
        span_.end(rt.stream());
        recomputedPhotonIndices_->takenInPlaceOfBranch = false;
        recomputedPhotonIndices_->nRecomputedPhotons = -1;
        recomputedPhotonIndices_->countPending = false;
        recomputedPhotonIndices_->replacedValid = false;
        remainingPhotonsToUpdate_ = 0;
        remainingPhotonsOffset_ = 0;
        if (photonRecomputationImportance_.getSize() > 0) resetPhotonImportance(0, photonRecomputationImportance_.getSize());
    }
    (void)nPhotonsToCompute;
    publishPhotons();
    if (rt.profiling()) {  // "Photon tracing: ... = X ms", "Computed photons: n = p %" (tracercl.cpp:562-598)
        rt.logProfile("Photon tracing");
        const int nr = recomputedPhotonIndices_->nRecomputedPhotons;
        const size_t np = photonData_->getNumberOfPhotons();
        char buf[96];
        std::snprintf(buf, sizeof buf, "Computed photons: %zu = %.2f %%", nr < 0 ? np : (size_t)nr, np ? 100.0 * (double)(nr < 0 ? np : (size_t)nr) / (double)np : 0.0);
        LogInfo(buf);
    }
}

PhotonToLightVolumeProcessorCL::PhotonToLightVolumeProcessorCL() {
    for (const char* id : { "volume", "photons", "recomputedPhotonIndices" }) addPortId(id, true);
    addPortId("lightvolume", false);
    recomputedPhotonIndicesPort_.setOptional(true);
    for (PropertyBase* p : std::initializer_list<PropertyBase*>{ &incrementalRecomputationThreshold_, &volumeSizeOption_, &volumeDataTypeOption_,
                                                                &alignChangedPhotons_, &workGroupSize_, &useGLSharing_, &formulation_, &progressiveAccumulation_,
                                                                &exactIncrementalUpdate_, &information_ })
        addProperty(*p);
    volumeSizeOption_.onChange([this]() { volumeSizeOptionChanged(); });
    volumeDataTypeOption_.onChange([this]() {
        lightVolume_->channels = volumeDataTypeOption_.get() == "4xfloat32" ? 4 : 1;
        lightVolume_->data.setSize(0);
        prevPhotons_.setSize(0);
    });
    outport_.setData(lightVolume_);
}
void PhotonToLightVolumeProcessorCL::volumeSizeOptionChanged() {  // photontolightvolumeprocessorcl.cpp:474-488
    if (volumeInport_.hasData() && volumeSizeOption_.get() != 0) {
        const size3_t in = volumeInport_.getData()->getDimensions();
        const size_t k = (size_t)volumeSizeOption_.get();
        const size3_t ns{ in.x / k, in.y / k, in.z / k }, cur = lightVolume_->getDimensions();
        if (ns.x != cur.x || ns.y != cur.y || ns.z != cur.z) { lightVolume_->setDimensions(ns); prevPhotons_.setSize(0); }
    }
}
void PhotonToLightVolumeProcessorCL::process() {  // photontolightvolumeprocessorcl.cpp:137-354
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || !photons_.isReady() || !volumeInport_.isReady()) return;
    rt.beginProfile();
    span_.poll();
    struct LogAtExit { const CpmRuntime& r; ~LogAtExit() { r.logProfile("Photons to light volume"); } } logAtExit{ rt };
    struct SpanAtExit { StreamSpan& s; hipStream_t st; ~SpanAtExit() { s.end(st); } } spanAtExit{ span_, rt.stream() };
    auto photonData = photons_.getData();
    if (volumeSizeOption_.get() == 0) {  // size from the photon radius (:144-163, Q15)
        const size_t n = (size_t)std::ceil(1.0 / photonData->getRadiusRelativeToSceneSize());
        const size3_t cur = lightVolume_->getDimensions();
        if (cur.x != n || cur.y != n || cur.z != n) { lightVolume_->setDimensions(size3_t{ n, n, n }); prevPhotons_.setSize(0); }
    } else {
        volumeSizeOptionChanged();
    }
    const size3_t outDim = lightVolume_->getDimensions();
    const size_t cells = outDim.x * outDim.y * outDim.z;
    information_.dimensions.set(std::to_string(outDim.x) + " x " + std::to_string(outDim.y) + " x " + std::to_string(outDim.z));
    information_.format.set(volumeDataTypeOption_.get() == "float32" ? "FLOAT32" : "Vec4FLOAT32");
    const int channels = lightVolume_->channels;
    bool fresh = false;
    if (lightVolume_->data.getSize() != cells * channels) { lightVolume_->data.setSize(cells * channels); fresh = true; }
    cpm_grid_desc g;
    const int32_t gd[3] = { (int32_t)outDim.x, (int32_t)outDim.y, (int32_t)outDim.z };
    cpm_grid_desc_default(&g, gd, channels);
    const int nPhotons = (int)photonData->getNumberOfPhotons(), nInter = photonData->getMaxPhotonInteractions();
    const float radius = (float)photonData->getRadiusRelativeToSceneSize();
    const float scale = cpm_relative_irradiance_scale(photonData->getRadiusRelativeToSceneSize(), (double)nPhotons);
    const float* photons = reinterpret_cast<const float*>(photonData->photons_.device());
    float* out = lightVolume_->data.device();
    const int maxRecomputationPhotons = (int)((float)nPhotons * (incrementalRecomputationThreshold_.get() / 100.f));
    const bool haveIdx = recomputedPhotonIndicesPort_.isReady();
    RecomputedPhotonIndices* rec = haveIdx ? recomputedPhotonIndicesPort_.getData().get() : nullptr;
    // the records the re-traced photons had before: kept by the tracer for exactly those photons (fused branch), else this
    // processor's whole-buffer snapshot of the previous evaluation
    const bool exactAddRemove = exactIncrementalUpdate_.get() && formulation_.get() == "gather";
    const bool useReplaced = haveIdx && rec->replacedValid && !exactAddRemove && rec->replacedPhotons.getSize() == photonData->photons_.getSize();
    // A fused tracer evaluation left the count on the device and it may still be on its way.  The add-remove launch does not need
    // it on the host -- it reads the device word and stands aside by itself when the count reaches the rebuild threshold
    // (apply_below) -- so it is enqueued FIRST, and only then does the host read the count (its pinned mailbox, not the
    // stream): the launch is already queued behind the tracer's when the count arrives, instead of being launched into an idle
    // GPU after it (8 - 10 us of every update).
    bool deltaEnqueued = false;
    if (haveIdx && rec->countPending && useReplaced && !fresh && maxRecomputationPhotons > 0) {
        if (rec->costs) span_.begin(rt.stream(), &rec->costs->branchLightVolumeMs(), rec->costs->probing);
        uint8_t* mask = nullptr;
        if (comm_) {  // multi-GPU: the bricks an old or new position touches, marked by the same launch
            const size_t nb = ((outDim.x + 3) / 4) * ((outDim.y + 3) / 4) * ((outDim.z + 3) / 4);
            brickMask_.setSize(nb);
            (void)hipMemsetAsync(brickMask_.device(), 0, nb, rt.stream());
            mask = brickMask_.device();
        }
        deltaEnqueued = rt.check(cpm_splat_delta(rt.ctx(), reinterpret_cast<const float*>(rec->replacedPhotons.device()), rec->replacedStride, photons,
                                                 rec->indicesToRecomputedPhotons.device(), rec->countDevice(), nPhotons, maxRecomputationPhotons, &g, radius,
                                                 scale, nPhotons, nInter, mask, out, rt.stream()), "cpm_splat_delta");
    }
    const int nRecomputed = haveIdx ? rec->resolveCount() : -1;
    const bool havePrev = useReplaced || (prevPhotonsValid_ && prevPhotons_.getSize() == photonData->photons_.getSize());
    const bool canAddRemove = !fresh && haveIdx && havePrev && nRecomputed > 0 && nRecomputed < maxRecomputationPhotons;
    // a progressive iteration (i > 1): this evaluation's estimate goes to a side buffer and is averaged in below.  Only when
    // the tracer really ran a progressive iteration over all photons -- a timer-driven continuation of a correlated update
    // (indices connected, a batch re-traced) is an add-remove on the light volume itself.
    const bool progressiveIteration = progressiveAccumulation_.get() && !fresh && photonData->iteration() > 1 && (!haveIdx || nRecomputed < 0) &&
                                      static_cast<int>(photonData->getInvalidationReason()) == static_cast<int>(PhotonData::InvalidationReason::Progressive);
    if (progressiveIteration) {
        estimate_.setSize(cells * channels);
        out = estimate_.device();
    }
    marksAreNonzero_ = false;
    bool partialUpdate = false;  // this evaluation only touched the re-traced photons
    bool marksDone = false;      // brickMask_ already holds the old AND new positions' bricks
    // what this evaluation costs on the GPU's timeline, filed under the way the tracer served the change (PathCosts)
    if (haveIdx && nRecomputed != 0 && rec->costs && !deltaEnqueued)  // (with the add-remove enqueued ahead, its span is already open)
        span_.begin(rt.stream(), (nRecomputed < 0) ? &rec->costs->fullLightVolumeMs : &rec->costs->branchLightVolumeMs(), nRecomputed >= 0 && rec->costs->probing);
    if (canAddRemove) {
        partialUpdate = true;
        const unsigned int* idx = rec->indicesToRecomputedPhotons.device();
        if (exactAddRemove) {
            // exact add-remove: mark the bricks an old or new position touches, re-bin, re-gather those bricks only
            const size_t nb = ((outDim.x + 3) / 4) * ((outDim.y + 3) / 4) * ((outDim.z + 3) / 4);
            brickMask_.setSize(nb);
            (void)hipMemsetAsync(brickMask_.device(), 0, nb, rt.stream());
            const size_t m = (size_t)nPhotons * nInter;
            order_.setSize(m); cellStart_.setSize(cells + 1); sorted_.setSize(m * (channels == 1 ? 4 : 8));
            bool ok = rt.check(cpm_mark_touched_bricks(rt.ctx(), reinterpret_cast<const float*>(prevPhotons_.device()), idx, nRecomputed, nPhotons,
                                                       nInter, &g, radius, brickMask_.device(), rt.stream()), "cpm_mark_touched_bricks(old)");
            ok = ok && rt.check(cpm_mark_touched_bricks(rt.ctx(), photons, idx, nRecomputed, nPhotons, nInter, &g, radius, brickMask_.device(),
                                                        rt.stream()), "cpm_mark_touched_bricks(new)");
            ok = ok && rt.check(cpm_bin(rt.ctx(), photons, (int)m, &g, order_.device(), cellStart_.device(), sorted_.device(), rt.stream()), "cpm_bin");
            if (ok) rt.check(cpm_gather_bricks(rt.ctx(), sorted_.device(), cellStart_.device(), (int)m, &g, radius, scale, brickMask_.device(), out,
                                               rt.stream()), "cpm_gather_bricks");
            marksDone = true;
            lastPath_ = "exact incremental";
        } else if (useReplaced && deltaEnqueued) {
            // add-remove (:196-298) in one launch over the device count: enqueued above, ahead of the count's arrival
            marksDone = comm_ != nullptr;
            lastPath_ = "incremental";
        } else if (useReplaced) {
            // add-remove (:196-298) in one launch over the device count: - the records the tracer replaced, + the new ones
            uint8_t* mask = nullptr;
            if (comm_) {  // multi-GPU: the bricks an old or new position touches, marked by the same launch
                const size_t nb = ((outDim.x + 3) / 4) * ((outDim.y + 3) / 4) * ((outDim.z + 3) / 4);
                brickMask_.setSize(nb);
                (void)hipMemsetAsync(brickMask_.device(), 0, nb, rt.stream());
                mask = brickMask_.device();
                marksDone = true;
            }
            rt.check(cpm_splat_delta(rt.ctx(), reinterpret_cast<const float*>(rec->replacedPhotons.device()), rec->replacedStride, photons, idx,
                                     rec->countDevice(), nPhotons, 0, &g, radius, scale, nPhotons, nInter, mask, out, rt.stream()), "cpm_splat_delta");
            lastPath_ = "incremental";
        } else {
        // add-remove (:196-298): -old, +new over the re-traced photons
        if (comm_) {  // multi-GPU: remember which bricks the OLD positions touch (the snapshot is refreshed below)
            const size_t nb = ((outDim.x + 3) / 4) * ((outDim.y + 3) / 4) * ((outDim.z + 3) / 4);
            brickMask_.setSize(nb);
            (void)hipMemsetAsync(brickMask_.device(), 0, nb, rt.stream());
            rt.check(cpm_mark_touched_bricks(rt.ctx(), reinterpret_cast<const float*>(prevPhotons_.device()), idx, nRecomputed, nPhotons, nInter, &g,
                                             radius, brickMask_.device(), rt.stream()), "cpm_mark_touched_bricks(old)");
        }
        rt.check(cpm_splat_selected(rt.ctx(), reinterpret_cast<const float*>(prevPhotons_.device()), idx, nRecomputed, &g, radius, scale, -1.f, nPhotons,
                                    nInter, out, rt.stream()), "cpm_splat_selected(-)");
        rt.check(cpm_splat_selected(rt.ctx(), photons, idx, nRecomputed, &g, radius, scale, 1.f, nPhotons, nInter, out, rt.stream()), "cpm_splat_selected(+)");
        lastPath_ = "incremental";
        }
    } else if (fresh || !havePrev || nRecomputed < 0 || nRecomputed >= maxRecomputationPhotons) {
        if (formulation_.get() == "splat") {  // the reference's formulation: clear + atomic splat (:299-339)
            (void)hipMemsetAsync(out, 0, cells * channels * sizeof(float), rt.stream());
            rt.check(cpm_splat_records(rt.ctx(), photons, nPhotons * nInter, nPhotons, &g, radius, scale, out, rt.stream()), "cpm_splat");
        } else if (formulation_.get() == "fast" && cpm_gather_fast_supported_on(rt.ctx(), &g, radius)) {
            // (asked of THIS device: a brick's LDS tile must fit what a workgroup may use here, else the bit-exact pair below serves the frame)
            // brick bin + LDS-tile gather, fixed-point sums: the reference's terms within the stated fp32 tolerance
            const size_t m = (size_t)nPhotons * nInter;
            // (a record per (photon, brick it reaches): cpm_fast_record_capacity)
            brickTable_.setSize(cpm_fast_table_entries(&g, (int)m));
            sorted_.setSize(cpm_fast_record_capacity(&g, (int)m, radius) * (channels == 1 ? 4 : 8));
            // (multi-GPU: the gather also marks the non-zero 4x4x4 bricks of the volume it writes -- what the sparse reduce sums over)
            uint8_t* marks = nullptr;
            if (comm_ && !progressiveIteration) {
                nonzeroMarks_.setSize(((outDim.x + 3) / 4) * ((outDim.y + 3) / 4) * ((outDim.z + 3) / 4) + 16);
                marks = nonzeroMarks_.device();
            }
            if (rt.check(cpm_bin_fast(rt.ctx(), photons, (int)m, &g, radius, brickTable_.device(), sorted_.device(), rt.stream()), "cpm_bin_fast") &&
                rt.check(cpm_gather_fast_marked(rt.ctx(), sorted_.device(), brickTable_.device(), (int)m, &g, radius, scale, 0, out, marks, rt.stream()),
                         "cpm_gather_fast"))
                marksAreNonzero_ = marks != nullptr;
        } else {  // sort/bin + deterministic per-cell gather
            const size_t m = (size_t)nPhotons * nInter;
            order_.setSize(m); cellStart_.setSize(cells + 1); sorted_.setSize(m * (channels == 1 ? 4 : 8));
            if (rt.check(cpm_bin(rt.ctx(), photons, (int)m, &g, order_.device(), cellStart_.device(), sorted_.device(), rt.stream()), "cpm_bin"))
                rt.check(cpm_gather(rt.ctx(), sorted_.device(), cellStart_.device(), (int)m, &g, radius, scale, 0, out, rt.stream()), "cpm_gather");
        }
        lastPath_ = "full";
    } else {
        lastPath_ = "unchanged";
    }
    // Snapshot for the next add-remove (:343-352).  The reference copies the whole photon buffer after every evaluation with
    // the index port connected (64 MiB of traffic per 1 M photon frame).  Here the tracer's fused branch hands over the replaced
    // records themselves, so the snapshot is only kept where something reads it: the exact add-remove, or a tracer that does
    // not keep the records (budget below 100 %, fusedImportanceBranch off).
    const bool tracerKeepsRecords = haveIdx && rec->keepsReplaced && !exactAddRemove && snapshotFree_;
    if (haveIdx && nRecomputed != 0 && !tracerKeepsRecords) {
        if (partialUpdate && prevPhotonsValid_) {  // only the re-traced photons differ from the snapshot: move those (the reference copies everything)
            const unsigned int* idx = rec->indicesToRecomputedPhotons.device();
            rt.check(cpm_snapshot_selected_photons(rt.ctx(), photons, idx, nRecomputed, nPhotons, nInter,
                                                   reinterpret_cast<float*>(prevPhotons_.device()), rt.stream()),
                     "cpm_snapshot_selected_photons");
        } else {
            if (prevPhotons_.getSize() != photonData->photons_.getSize()) prevPhotons_.setSize(photonData->photons_.getSize());
            (void)hipMemcpyAsync(prevPhotons_.device(), photonData->photons_.device(), photonData->photons_.getSizeInBytes(), hipMemcpyDeviceToDevice, rt.stream());
        }
        prevPhotonsValid_ = true;
    } else if (haveIdx && nRecomputed != 0) {
        prevPhotonsValid_ = false;  // the photons moved on without the snapshot
    }
    if (progressiveIteration && lastPath_[0] == 'f') {  // L_i = mix(L_(i-1), E_i, 1 / i)
        rt.check(cpm_mix_buffers(rt.ctx(), lightVolume_->data.device(), estimate_.device(), 1.0f / (float)photonData->iteration(), cells * channels,
                                 CPM_MIX_F32, lightVolume_->data.device(), rt.stream()), "cpm_mix_buffers(progressive average)");
        lastPath_ = "progressive";
    }
    if (comm_ && lastPath_[0] != 'u') {
        const unsigned int* idx = partialUpdate ? recomputedPhotonIndicesPort_.getData()->indicesToRecomputedPhotons.device() : nullptr;
        // NOTE: the snapshot above has already been refreshed; the touched bricks of the OLD positions were marked before it
        // (marksDone: the new positions' bricks as well -- nothing left to mark)
        reduceOverShards(g, cells * channels, partialUpdate, marksDone ? photons : nullptr, photons, idx, nRecomputed, nPhotons, nInter, radius);
        copyToGLBuffer(reducedVolume_->data.device(), cells * channels);
        outport_.setData(reducedVolume_);
        return;
    }
    copyToGLBuffer(lightVolume_->data.device(), cells * channels);
    outport_.setData(lightVolume_);
}

void PhotonToLightVolumeProcessorCL::dropGLBuffer() {
    if (glBuffer_ && CpmRuntime::get().valid()) cpm_gl_unregister(CpmRuntime::get().ctx(), glBuffer_);
    glBuffer_ = nullptr;
}

// VolumeCLGL + SyncCLGL + enqueueCopyBufferToImage (photontolightvolumeprocessorcl.cpp:184-194,404-406): acquire the host's
// pixel-unpack buffer, write the texels into it, release -- all on the frame's stream; the host's glTexSubImage3D follows.
void PhotonToLightVolumeProcessorCL::copyToGLBuffer(const float* volume, size_t n) {
    lastGLCopy_ = "none";
    if (!useGLSharing_.get() || glBufferName_ == 0) return;
    auto& rt = CpmRuntime::get();
    if (!cpm_gl_available(rt.ctx())) { lastGLCopy_ = "no context"; return; }
    lastGLCopy_ = "failed";
    if (!glBuffer_ && cpm_gl_register_buffer(rt.ctx(), glBufferName_, 0, &glBuffer_) != CPM_OK) return;
    if (cpm_gl_acquire(rt.ctx(), &glBuffer_, 1, rt.stream()) != CPM_OK) return;
    const int rc = cpm_gl_copy_to_buffer(rt.ctx(), volume, n, glTexel_, glBuffer_, rt.stream());
    if (cpm_gl_release(rt.ctx(), &glBuffer_, 1, rt.stream()) == CPM_OK && rc == CPM_OK) lastGLCopy_ = "copied";
}

void PhotonToLightVolumeProcessorCL::dropSparseReduce() {
    if (sparseReduce_) cpm_sparse_reduce_destroy(sparseReduce_);
    sparseReduce_ = nullptr;
}
void PhotonToLightVolumeProcessorCL::setCommunicator(cpm_comm* comm) {
    if (comm != comm_) dropSparseReduce();
    comm_ = comm;
}

// The one exchange step of the path: sum of the shards' partial light volumes -- over the union of the shards' non-zero
// (full evaluation) or touched (add-remove) 4x4x4 bricks only, enqueued without a stream synchronisation
// (cpm_allreduce_grid_sparse).  The ticket is completed here, before the outport hands the volume on: a consumer on this
// stream must find the sum, also after an overflow (the dense sum is then enqueued behind the sparse chain); that reads one
// word of pinned memory the frame's own launches write.
void PhotonToLightVolumeProcessorCL::reduceOverShards(const cpm_grid_desc& g, size_t count, bool partialUpdate, const float* marksComplete,
                                                      const float* photons, const unsigned int* idx, int nRecomputed, int nPhotons,
                                                      int nInter, float radius) {
    auto& rt = CpmRuntime::get();
    const size3_t outDim = lightVolume_->getDimensions();
    if (!reducedVolume_ || reducedVolume_->data.getSize() != count) {
        reducedVolume_ = std::make_shared<Volume>(outDim, CPM_F32);
        reducedVolume_->channels = lightVolume_->channels;
        reducedVolume_->data.setSize(count);
        partialUpdate = false;  // nothing to update incrementally yet
        firstSumIntoReduced_ = true;
    }
    if (sparseReduce_ && (sparseReduceDims_.x != outDim.x || sparseReduceDims_.y != outDim.y || sparseReduceDims_.z != outDim.z ||
                          sparseReduceChannels_ != lightVolume_->channels))
        dropSparseReduce();
    if (!sparseReduce_) {
        if (!rt.check(cpm_sparse_reduce_create(rt.ctx(), comm_, &g, &sparseReduce_), "cpm_sparse_reduce_create")) sparseReduce_ = nullptr;
        sparseReduceDims_ = outDim;
        sparseReduceChannels_ = lightVolume_->channels;
    }
    // Which bricks this shard hands in, and as what.  Shards decide between rebuild and add-remove on their own counts, so in one frame
    // one shard may rebuild while another updates: the kind of mask must not depend on that decision.  Rule: the first sum into a new
    // `reducedVolume_` (every shard alike: same evaluation) is NONZERO -- the whole volume is defined, zeros outside the union; every
    // later one is TOUCHED -- the bricks where THIS shard's contribution changed: an update's old and new positions, or, for a rebuild,
    // everything the shard had lit (its marks of the last rebuild and what its updates touched since: litBefore_) or lights now.
    const size_t nb4 = (size_t)((outDim.x + 3) / 4) * ((outDim.y + 3) / 4) * ((outDim.z + 3) / 4);
    const bool firstSum = firstSumIntoReduced_;
    firstSumIntoReduced_ = false;
    const uint8_t* mask = nullptr;
    int maskKind = CPM_SPARSE_MASK_TOUCHED;
    if (litBefore_.getSize() != nb4 + 16) { litBefore_.setSize(nb4 + 16); litBeforeValid_ = false; }
    if (partialUpdate && brickMask_.getSize() != 0 && idx) {
        // add-remove: only bricks touched by an old or a new position of a re-traced photon changed on this shard.  The old
        // positions were marked into brickMask_ before the snapshot moved on (see process()); add the new ones, then sum
        // the union of all shards' bricks only.
        if (marksComplete != nullptr ||
            rt.check(cpm_mark_touched_bricks(rt.ctx(), photons, idx, nRecomputed, nPhotons, nInter, &g, radius, brickMask_.device(), rt.stream()),
                     "cpm_mark_touched_bricks(new)")) {
            mask = brickMask_.device();
            if (litBeforeValid_) litBeforeValid_ = rt.check(cpm_brick_mask_or(rt.ctx(), litBefore_.device(), brickMask_.device(), nb4, rt.stream()), "cpm_brick_mask_or");
        }
    } else if (firstSum) {
        if (marksAreNonzero_) { mask = nonzeroMarks_.device(); maskKind = CPM_SPARSE_MASK_NONZERO; }   // (no marks: the reduce finds the non-zero bricks itself)
    } else {
        // a rebuild while a previous sum stands
        allBricks_.setSize(nb4 + 16);
        if (marksAreNonzero_ && litBeforeValid_ &&
            hipMemcpyAsync(allBricks_.device(), nonzeroMarks_.device(), nb4, hipMemcpyDeviceToDevice, rt.stream()) == hipSuccess &&
            rt.check(cpm_brick_mask_or(rt.ctx(), allBricks_.device(), litBefore_.device(), nb4, rt.stream()), "cpm_brick_mask_or")) {
            mask = allBricks_.device();
        } else {  // what it had lit is not known (a formulation without marks): every brick counts as touched -- the dense sum
            (void)hipMemsetAsync(allBricks_.device(), 1, nb4, rt.stream());
            mask = allBricks_.device();
        }
    }
    if (!partialUpdate) {  // what this shard lights from here on
        litBeforeValid_ = marksAreNonzero_ &&
                          hipMemcpyAsync(litBefore_.device(), nonzeroMarks_.device(), nb4, hipMemcpyDeviceToDevice, rt.stream()) == hipSuccess;
    }
    if (sparseReduce_) {
        uint64_t ticket = 0;
        cpm_sparse_reduce_info info = {};
        if (rt.check(cpm_allreduce_grid_sparse(rt.ctx(), sparseReduce_, lightVolume_->data.device(), reducedVolume_->data.device(), mask, maskKind, -1, 0,
                                               &ticket, rt.stream()), "cpm_allreduce_grid_sparse") &&
            rt.check(cpm_sparse_reduce_complete(rt.ctx(), sparseReduce_, ticket, rt.stream(), &info), "cpm_sparse_reduce_complete")) {
            lastReduce_ = info.mode == 0 ? (mask && maskKind == CPM_SPARSE_MASK_TOUCHED ? "touched bricks" : "non-zero bricks") : (info.mode == 1 ? "dense" : "dense (overflow)");
            return;
        }
    }
    rt.check(cpm_allreduce_grid(rt.ctx(), comm_, lightVolume_->data.device(), reducedVolume_->data.device(), count, rt.stream()), "cpm_allreduce_grid");
    lastReduce_ = "dense";
}

}  // namespace inviwo
