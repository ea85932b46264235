// cpm_processors.h -- host layer above the C-ABI, mirroring the reference's operator / plugin
// interface for the path: same class names, class identifiers, port ids, property ids,
// defaults and error behaviour (log and carry on), so that the modules stay a drop-in for the
// CorrelatedPhotonMappingSingleVolume workspace.  Each class cites what it mirrors (paths
// relative to /root/reference/modules).  Everything below calls only include/cpm/cpm.h.
#pragma once
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "cpm/cpm_ext.h"
#include "inviwo_lite.h"

namespace inviwo {

// The one libcpm_hip context of the process (stands where OpenCL::getPtr() stood).
class CpmRuntime {
public:
    static CpmRuntime& get();
    cpm_ctx* ctx() const { return ctx_; }
    hipStream_t stream() const { return nullptr; }
    bool valid() const { return ctx_ != nullptr; }
    // log-and-swallow, like the reference's catch (cl::Error&) { LogError(...) }
    bool check(int status, const char* what) const;
    // IVW_PROFILING's counterpart (the reference logs OpenCL event times per stage: tracercl.cpp:562-598,
    // ...processorcl.cpp:247-261,328-338): with CPM_PROFILING=1 in the environment every kernel launch is bracketed by
    // HIP events and a processor logs "<label>: kernel a ms + kernel b ms + ... = X ms" after it has evaluated.
    bool profiling() const { return profiling_; }
    void beginProfile() const;
    void logProfile(const char* label) const;
private:
    bool profiling_ = false;
    CpmRuntime();
    ~CpmRuntime();
    cpm_ctx* ctx_ = nullptr;
};

// Measured cost of the two ways to serve a transfer-function / volume change, in GPU-timeline milliseconds (HIP events
// recorded in the stream around a processor's launches, read back at a later evaluation -- never waited for):
// tracer + light volume of a full frame against importance branch + add-remove.  Owned by the tracer processor, which decides
// with it; the light-volume processor files its share through the indices port's payload.
struct PathCosts {
    // (-2: never run; -1: run once -- a kernel's first launch carries its one-off load, that sample is dropped; >= 0: running mean)
    float fullTraceMs = -2.f, fullLightVolumeMs = -2.f;       // everything re-traced; volume rebuilt
    // importance branch; add-remove (or rebuild above its threshold).  The branch's cost grows with the fraction re-traced
    // (1 % -> 100 %: 0.067 -> 0.12 ms at config 3), so it is kept per coarse bucket of that fraction -- < 1 %, < 5 %, < 25 %, the
    // rest -- and looked up / filed under the bucket of the LAST fraction seen: consecutive edits of one interaction are alike.
    static constexpr int kBuckets = 4;
    float branchTraceMsBy[kBuckets] = { -2.f, -2.f, -2.f, -2.f }, branchLightVolumeMsBy[kBuckets] = { -2.f, -2.f, -2.f, -2.f };
    int bucket = 0;
    int evaluationsSinceProbe = 0;                            // full frames taken in place of the branch since it was last measured
    bool probing = false;                                     // this evaluation takes the branch only to re-measure it: its spans are always sampled
    static int bucketOf(float fraction) { return fraction < 0.01f ? 0 : fraction < 0.05f ? 1 : fraction < 0.25f ? 2 : 3; }
    void sawFraction(float fraction) { const int b = bucketOf(fraction); if (b != bucket) { bucket = b; evaluationsSinceProbe = 0; } }
    float& branchTraceMs() { return branchTraceMsBy[bucket]; }
    float& branchLightVolumeMs() { return branchLightVolumeMsBy[bucket]; }
    float branchTraceMs() const { return branchTraceMsBy[bucket]; }
    float branchLightVolumeMs() const { return branchLightVolumeMsBy[bucket]; }
    bool known() const { return fullTraceMs >= 0.f && fullLightVolumeMs >= 0.f && branchTraceMs() >= 0.f && branchLightVolumeMs() >= 0.f; }
};

// A pair of events around the launches of one evaluation; the elapsed time is picked up when a later evaluation finds it ready.
class StreamSpan {
public:
    ~StreamSpan();
    // records the start; `target` receives the milliseconds once they are known.  A cost that is known is sampled again only
    // every kEvery-th time: an event pair costs the frame ~3 us of its 70 on the GPU's timeline
    // (the count of skipped samples is kept PER TARGET -- a span serves several costs, and the rarely taken path's probes, every 33rd
    // evaluation under the adaptive policy, must not be thinned by the common path's samples: ADVICE r04 -- and `force` takes the sample
    // whatever the count: a probe is always measured)
    void begin(hipStream_t s, float* target, bool force = false);
    static constexpr int kEvery = 16;
    void end(hipStream_t s);
    void poll();                               // non-blocking: stores the elapsed time if the end event has completed
private:
    hipEvent_t a_ = nullptr, b_ = nullptr;
    float* target_ = nullptr;
    bool pending_ = false, open_ = false;
    std::map<const float*, int> skipped_;
};

// ---- data types (L2) ---------------------------------------------------------------------------

// progressivephotonmapping/photondata.h:47-56
struct Photon {
    vec3 pos;
    vec3 power;
    vec2 encodedDirection;
    void setDirection(vec3 dir);
    vec3 getDirection() const;
};
static_assert(sizeof(Photon) == 32, "Photon must match the 32-byte device record");

// progressivephotonmapping/photondata.h:58-63
struct RecomputedPhotonIndices {
    Buffer<unsigned int> indicesToRecomputedPhotons;
    int nRecomputedPhotons = -1;
    bool isInitialized() const { return nRecomputedPhotons != -1; }
    void setUninitialized() { nRecomputedPhotons = -1; }
    // The fused importance branch (cpm_selection_*, cpm_trace_selected) leaves the count on the device: nRecomputedPhotons is
    // then resolved on first use from the selection's host mailbox (no stream synchronisation; ref tracercl.cpp:343-345,374
    // blocks on an event here).  `selection` is owned by the tracer processor.
    cpm_selection* selection = nullptr;
    bool countPending = false;
    int resolveCount();
    const int32_t* countDevice() const { return selection ? cpm_selection_count_device(selection) : nullptr; }
    // ... and the tracer keeps the records it overwrites: replacedPhotons[k * replacedStride + j] is what photon
    // indicesToRecomputedPhotons[j] was at interaction k before this evaluation -- the part of the light-volume processor's
    // prevPhotons_ snapshot (ref photontolightvolumeprocessorcl.cpp:343-352: a copy of the WHOLE buffer per evaluation) that
    // the add-remove update reads.  replacedValid: filled by the evaluation that produced the current indices.
    Buffer<vec4> replacedPhotons;
    int replacedStride = 0;
    bool replacedValid = false;
    bool keepsReplaced = false;  // the tracer is configured to hand over replaced records on its importance branch (no snapshot needed downstream)
    bool takenInPlaceOfBranch = false;  // nRecomputedPhotons == -1 because the tracer chose the full frame over its importance branch
    PathCosts* costs = nullptr;         // the tracer's measurements (the light-volume processor adds its share)
};

// progressivephotonmapping/photondata.h:65-156, photondata.cpp:36-98
class PhotonData {
public:
    enum class InvalidationReason { Camera = 1, TransferFunction = 2, Light = 4, Progressive = 8, Volume = 16, All = 31 };
    void setSize(size_t numberOfPhotons, int maxPhotonInteractions);
    size_t getNumberOfPhotons() const { return photons_.getSize() / (2 * maxPhotonInteractions_); }
    int getMaxPhotonInteractions() const { return maxPhotonInteractions_; }
    void setRadius(double radiusRelativeToSceneSize, double sceneRadius);
    void setRadius(double radius) { worldSpaceRadius_ = radius; }
    void advanceToNextIteration(double alpha = 0.5);
    double getRadiusRelativeToSceneSize() const { return getRadius() / sceneRadius_; }
    double getRadius() const { return worldSpaceRadius_; }
    static double progressiveSphereRadius(double radius, int iteration, double alpha);
    static double sphereVolume(double radius);
    double getSceneRadius() const { return sceneRadius_; }
    void resetIteration() { iteration_ = 0; }
    int iteration() const { return iteration_; }
    void setIteration(int v) { iteration_ = v; }
    InvalidationReason getInvalidationReason() const { return invalidationFlag_; }
    void setInvalidationReason(InvalidationReason v) { invalidationFlag_ = v; }
    Buffer<vec4> photons_;  // 2 vec4 per photon record, N * I records
    static const double scaleToMakeLightPowerOfOneVisibleForDirectionalLightSource;
private:
    int maxPhotonInteractions_ = 1;
    double sceneRadius_ = 1.0, worldSpaceRadius_ = 0.01;
    int iteration_ = 0;
    InvalidationReason invalidationFlag_ = InvalidationReason::All;
};
inline PhotonData::InvalidationReason operator|(PhotonData::InvalidationReason a, PhotonData::InvalidationReason b) {
    return static_cast<PhotonData::InvalidationReason>(static_cast<int>(a) | static_cast<int>(b));
}

// lightcl/lightsample.h:88-115 (32-byte POD samples: SURVEY Q12)
class LightSamples {
public:
    explicit LightSamples(size_t n = 0) { setSize(n); }
    Buffer<Photon>* getLightSamples() { return &lightSamples_; }
    const Buffer<Photon>* getLightSamples() const { return &lightSamples_; }
    Buffer<vec2>* getIntersectionPoints() { return &intersectionPoints_; }
    const Buffer<vec2>* getIntersectionPoints() const { return &intersectionPoints_; }
    void setSize(size_t n) { lightSamples_.setSize(n); intersectionPoints_.setSize(n); }
    size_t getSize() const { return lightSamples_.getSize(); }
    void resetIteration() { iteration_ = 0; }
    void advanceIteration() { ++iteration_; }
    bool isReset() const { return iteration_ <= 1; }
    size_t getIteration() const { return iteration_; }
    size_t changeStamp = 0;  // bumped by the sampler whenever the samples are rewritten (stands for the port's onChange)
private:
    Buffer<Photon> lightSamples_;
    Buffer<vec2> intersectionPoints_;
    size_t iteration_ = 0;
};
using SampleBuffer = Buffer<vec4>;  // lightcl/sample.h

// uniformgridcl/uniformgrid3d.h:63-136
class UniformGrid3DBase {
public:
    virtual ~UniformGrid3DBase() = default;
    size3_t getDimensions() const { return dims_; }
    void setDimensions(size3_t d) { dims_ = d; resize(d.x * d.y * d.z); }
    size3_t getCellDimension() const { return cellDim_; }
    void setCellDimension(size3_t c) { cellDim_ = c; }
    const mat4& getModelMatrix() const { return model_; }
    const mat4& getWorldMatrix() const { return world_; }
    void setModelMatrix(const mat4& m) { model_ = m; }
    void setWorldMatrix(const mat4& m) { world_ = m; }
    // format-generic access (getDataFormat()->getString(), getSizeInBytes(), getData(), clone())
    virtual const char* getDataFormatString() const = 0;
    virtual size_t getSizeInBytes() const = 0;
    virtual void* deviceData() = 0;
    virtual void* hostData() = 0;       // current RAM representation (downloads when the device copy is the master)
    virtual void uploadHostData() = 0;  // after writing through hostData()
    virtual bool hasDeviceData() const = 0;
    virtual int mixType() const = 0;    // cpm_mix_type for BufferMixerCL::mix
    virtual size_t mixElements() const = 0;
    virtual std::shared_ptr<UniformGrid3DBase> clone() const = 0;  // same shape, metadata and data
protected:
    virtual void resize(size_t n) = 0;
    void copyMetaTo(UniformGrid3DBase& o) const { o.cellDim_ = cellDim_; o.model_ = model_; o.world_ = world_; o.setDimensions(dims_); }
    size3_t dims_{ 0, 0, 0 }, cellDim_{ 8, 8, 8 };
    mat4 model_ = identityMatrix(), world_ = identityMatrix();
};
using UniformGrid3DVector = std::vector<std::shared_ptr<UniformGrid3DBase>>;

template <typename Derived, typename T, int Components>
struct UniformGrid3DTyped : UniformGrid3DBase {
    Buffer<T> data;
    size_t getSizeInBytes() const override { return data.getSizeInBytes(); }
    void* deviceData() override { return data.device(); }
    void* hostData() override { return data.hostData(); }
    void uploadHostData() override { data.upload(); }
    bool hasDeviceData() const override { return data.hasDevice(); }
    size_t mixElements() const override { return data.getSize() / (Components == 2 ? 2 : 1); }
    std::shared_ptr<UniformGrid3DBase> clone() const override {
        auto c = std::make_shared<Derived>();
        copyMetaTo(*c);
        auto& src = const_cast<Buffer<T>&>(data);
        if (src.hasDevice()) {
            if (data.getSize()) (void)hipMemcpy(c->data.device(), src.device(), data.getSizeInBytes(), hipMemcpyDeviceToDevice);
        } else {
            c->data.ram() = src.ram();
        }
        return c;
    }
protected:
    void resize(size_t n) override { data.setSize(Components * n); }
};
// uniformgridcl/minmaxuniformgrid3d.h:41-42: UniformGrid3D<DataVec2UInt16::type>
struct MinMaxUniformGrid3D : UniformGrid3DTyped<MinMaxUniformGrid3D, uint16_t, 2> {
    const char* getDataFormatString() const override { return "Vec2UINT16"; }
    int mixType() const override { return CPM_MIX_U16X2; }
};
// UniformGrid3D<float>: importance grids (importancesamplingcl) and per-brick volume differences
// (DynamicVolumeInfoUniformGrid3D, uniformgridcl/processors/dynamicvolumedifferenceanalysis.h)
struct ImportanceUniformGrid3D : UniformGrid3DTyped<ImportanceUniformGrid3D, float, 1> {
    const char* getDataFormatString() const override { return "FLOAT32"; }
    int mixType() const override { return CPM_MIX_F32; }
    // one bit per cell, set where the importance is not +0: written by the launch that wrote `data`
    // (cpm_importance_tf_occupancy) and handed to the tracer's selection, which otherwise makes the bits itself.  Only valid
    // while occupancyFor == data.device() contents' last writer -- the importance processor sets and clears it.
    Buffer<uint32_t> occupancy;
    bool occupancyValid = false;
};
struct DynamicVolumeInfoUniformGrid3D : UniformGrid3DTyped<DynamicVolumeInfoUniformGrid3D, float, 1> {
    const char* getDataFormatString() const override { return "FLOAT32"; }
    int mixType() const override { return CPM_MIX_F32; }
};

struct DirectionalLight {  // what baseLightToPackedLight yields in data space
    vec3 position{ 0.5f, 0.5f, -1.5f };
    vec3 direction{ 0.f, 0.f, 1.f };   // photon travel direction
    vec3 radiance{ 1.f, 1.f, 1.f };
};
struct Mesh {  // proxy geometry: vertex positions (data space) + triangle indices
    std::vector<vec3> vertices;
    std::vector<int> indices;
    static std::shared_ptr<Mesh> unitCube();
    static std::shared_ptr<Mesh> box(vec3 lo, vec3 hi);
};

// ---- host geometry (lightcl/*.cpp) ---------------------------------------------------------------

namespace geometry {
struct Plane { vec3 point, normal; };
void projectPointsOnPlane(const std::vector<vec3>& points, const Plane& plane, vec3 u, vec3 v, std::vector<vec2>& out);  // pointplaneprojection.cpp:39-54
std::vector<vec2> convexHull2D(std::vector<vec2> points);                                                                 // convexhull2d.cpp:38-130
std::tuple<vec2, vec2, vec2> mimumBoundingRectangle(const std::vector<vec2>& hull);                                      // orientedboundingbox2d.cpp:40-78
std::tuple<vec3, vec3, vec3> fitPlaneAlignedOrientedBoundingBox2D(const std::vector<vec3>& points, const Plane& plane);  // :80-100
}  // namespace geometry

// ---- algorithm classes (L3) ------------------------------------------------------------------------

// rndgenmwc64x/mwc64xseedgenerator.{h,cpp}
class MWC64XSeedGenerator {
public:
    void generateRandomSeeds(Buffer<uvec2>* buffer, unsigned int seed);
};

// importancesamplingcl/uniformsamplegenerator2dcl.{h,cpp}
class UniformSampleGenerator2DCL {
public:
    void generateNextSamples(SampleBuffer& positionSamplesOut, ivec2 nSamples);
};

// lightcl/directionallightsamplercl.{h,cpp} + lightcl/lightsamplemeshintersectioncl.{h,cpp}
class DirectionalLightSamplerCL {
public:
    void sampleLightSource(const Mesh* mesh, const SampleBuffer* samples, const DirectionalLight* light, LightSamples& lightSamplesOut);
    std::tuple<vec3, vec3, vec3, float> lastPlane() const { return std::make_tuple(origin_, u_, v_, area_); }
private:
    vec3 origin_, u_, v_;
    float area_ = 0;
};
class LightSampleMeshIntersectionCL {
public:
    void meshSampleIntersection(const Mesh* mesh, LightSamples* samples);
};

// Inviwo's AdvancedMaterialProperty (composite "material": phaseFunction, IOR, roughness, specularColor, anisotropy) as far
// as the tracer reads it: the phase function enum and the combined parameters, .x = anisotropy g
struct AdvancedMaterialProperty : CompositeProperty {
    AdvancedMaterialProperty() : CompositeProperty("material", "Material") {
        addProperty(phaseFunctionProp); addProperty(indexOfRefractionProp); addProperty(roughnessProp);
        addProperty(specularColorProp); addProperty(anisotropyProp);
        anisotropyProp.onChange([this]() { combined.x = anisotropyProp.get(); changed(); });
        phaseFunctionProp.onChange([this]() {
            phaseFunction = phaseFunctionProp.get() == "Isotropic" ? CPM_PHASE_ISOTROPIC : CPM_PHASE_HENYEY_GREENSTEIN;
            changed();
        });
    }
    StringOptionProperty phaseFunctionProp{ "phaseFunction", "Phase function", "HenyeyGreenstein" };
    FloatProperty indexOfRefractionProp{ "IOR", "Index of refraction", 1.f };   // unused by the volumetric tracer
    FloatProperty roughnessProp{ "roughness", "Roughness", 0.1f };              // unused by the volumetric tracer
    FloatVec4Property specularColorProp{ "specularColor", "Specular color", vec4(1.f, 1.f, 1.f, 1.f) };  // unused
    FloatProperty anisotropyProp{ "anisotropy", "Anisotropy (g)", 0.f };
    vec4 combined{ 0.f, 0.f, 0.f, 0.f };  // .x = anisotropy g
    int phaseFunction = CPM_PHASE_HENYEY_GREENSTEIN;
    vec4 getCombinedMaterialParameters() const { return combined; }
    int getPhaseFunctionEnum() const { return phaseFunction; }
};
// Inviwo's CameraProperty as the tracer uses it: a change invalidates with reason Camera (tracercl.cpp:161-165)
struct CameraProperty : CompositeProperty {
    CameraProperty() : CompositeProperty("camera", "Camera") {
        for (PropertyBase* q : std::initializer_list<PropertyBase*>{ &cameraType, &lookFrom, &lookTo, &lookUp, &aspectRatio, &nearPlane, &farPlane, &fov,
                                                                    &fitToBasis, &mouseChangeFocusPoint })
            addProperty(*q);
        for (PropertyBase* q : getProperties()) q->onChange([this]() { changed(); });
    }
    StringOptionProperty cameraType{ "cameraType", "Camera Type", "PerspectiveCamera" };
    FloatVec3Property lookFrom{ "lookFrom", "Look from", vec3(0.f, 0.f, -2.f) }, lookTo{ "lookTo", "Look to", vec3(0.f, 0.f, 0.f) },
        lookUp{ "lookUp", "Look up", vec3(0.f, 1.f, 0.f) };
    FloatProperty aspectRatio{ "aspectRatio", "Aspect Ratio", 1.f }, nearPlane{ "near", "Near Plane", 0.1f }, farPlane{ "far", "Far Plane", 100.f },
        fov{ "fov", "FOV", 38.f };
    BoolProperty fitToBasis{ "fitToBasis_", "Fit to basis", true }, mouseChangeFocusPoint{ "mouseChangeFocusPoint", "Change Focus Point", false };
};

// progressivephotonmapping/photontracercl.{h,cpp}
class PhotonTracerCL {
public:
    PhotonTracerCL() = default;
    ~PhotonTracerCL();
    bool isValid() const { return CpmRuntime::get().valid(); }
    // photontracercl.cpp:67-132 (Volume/TF overload) and :135-174 (device overload) folded into one
    void tracePhotons(const Volume* volume, const TransferFunction& transferFunction, const float aabb[8],
                      const AdvancedMaterialProperty& material, float stepSize, const LightSamples* lightSamples,
                      const Buffer<unsigned int>* photonsToRecomputeIndices, int nInvalidPhotons, int photonOffset,
                      int batch, int maxInteractions, PhotonData* photonOutData);
    // every light of a full frame in ONE launch (cpm_trace_lights: light l at photon offset sum of the sizes before it) instead of the
    // reference's launch per light (processor/progressivephotontracercl.cpp:543-549): same photons, the launch's fixed part paid
    // once.  false = does not apply (one light, more than CPM_MAX_TRACE_LIGHTS): the caller loops.
    bool tracePhotonsAllLights(const Volume* volume, const TransferFunction& transferFunction, const float aabb[8],
                               const AdvancedMaterialProperty& material, float stepSize, const std::vector<const LightSamples*>& lights,
                               int maxInteractions, PhotonData* photonOutData);
    void setNoSingleScattering(bool v) { onlyMultipleScattering_ = v; }
    void setProgressive(bool v) { progressive_ = v; }
    bool isProgressive() const { return progressive_; }
    // the same over a device-side count (cpm_trace_selected): thread j < min(*nIndicesDevice, maxIndices); the records about to
    // be overwritten go to replacedPhotons (stride maxIndices), the traced photons' importance keys are reset
    // detector + threshold + tracer of one light in one launch (cpm_photon_importance_retrace); the replaced records go to
    // replacedPhotons at the photons' own indices
    bool importanceRetrace(cpm_selection* selection, const Volume* volume, const ImportanceUniformGrid3D* grid, const TransferFunction& transferFunction,
                           const float aabb[8], const AdvancedMaterialProperty& material, float stepSize, const LightSamples* lightSamples,
                           Buffer<unsigned int>& importances, vec4* replacedPhotons, int photonOffset, int maxInteractions, bool fixExitPoint,
                           PhotonData* photonOutData);
    // ... for every light of the evaluation in ONE launch (cpm_photon_importance_retrace_lights; light l at photon offset sum of the sizes
    // before it): -1 = does not apply (one light, more than CPM_MAX_TRACE_LIGHTS: the caller loops), 0 = failed, 1 = enqueued
    int importanceRetraceAllLights(cpm_selection* selection, const Volume* volume, const ImportanceUniformGrid3D* grid,
                                   const TransferFunction& transferFunction, const float aabb[8], const AdvancedMaterialProperty& material,
                                   float stepSize, const std::vector<const LightSamples*>& lights, Buffer<unsigned int>& importances, vec4* replacedPhotons,
                                   int maxInteractions, bool fixExitPoint, PhotonData* photonOutData);
    void tracePhotonsSelected(const Volume* volume, const TransferFunction& transferFunction, const float aabb[8],
                              const AdvancedMaterialProperty& material, float stepSize, const LightSamples* lightSamples,
                              const Buffer<unsigned int>* indices, const int32_t* nIndicesDevice, int maxIndices, vec4* replacedPhotons,
                              unsigned int* resetImportances, int photonOffset, int maxInteractions, PhotonData* photonOutData);
    void setRandomSeedSize(size_t nPhotons);   // :176-182
    void seedStreamsFor(const PhotonData& photons);
    Buffer<uvec2>& randomState() { return randomState_; }
    // Bring the device LUT up to date with `tf` (one small launch, no host wait).  Called at the top of the tracer processor's
    // evaluation; tracePhotons* call it too (a no-op then).
    void syncTF(const TransferFunction& tf);
    // Full launches take their 256-sample chunks in the order of their measured costs (cpm_trace_order_*): per light one
    // order object; a launch is measured -- and the order re-sorted behind it -- when the light is new, when the transfer
    // function or the volume changed since the last measured launch (but no more often than every kMeasureAtLeastApart-th
    // launch), and every kMeasureEvery-th launch otherwise.
    void setAdaptiveLaunchOrder(bool v) { adaptiveLaunchOrder_ = v; }
    static constexpr int kMeasureEvery = 256, kMeasureAtLeastApart = 32;
private:
    struct LaunchOrder { cpm_trace_order* order = nullptr; int n = 0; int sinceMeasured = 0; bool stale = false; const void* volume = nullptr; };
    std::vector<std::pair<const LightSamples*, LaunchOrder>> launchOrders_;
    LaunchOrder allLightsOrder_;                         // ... of the launch over all lights
    std::vector<const LightSamples*> allLightsOrderFor_;
    bool adaptiveLaunchOrder_ = true;
    Buffer<uvec2> randomState_;
    bool onlyMultipleScattering_ = false, progressive_ = false;
    cpm_tf* tf_ = nullptr;
    std::vector<float> tfLut_;
    std::vector<TFPrimitive> tfPoints_;
};

// progressivephotonmapping/photonrecomputationdetector.{h,cpp}
class PhotonRecomputationDetector {
public:
    bool isValid() const { return CpmRuntime::get().valid(); }
    void photonRecomputationImportance(const PhotonData* photonData, int photonOffset, const Volume* origVolume,
                                       const ImportanceUniformGrid3D* uniformGridVolume, const LightSamples& lightSamples,
                                       Buffer<unsigned int>& recomputationImportance);
    // the same fused with threshold + count + index lists (cpm_photon_importance_select): appends this light to `selection`
    bool photonRecomputationImportanceSelect(cpm_selection* selection, const PhotonData* photonData, int photonOffset, const Volume* origVolume,
                                             const ImportanceUniformGrid3D* uniformGridVolume, const LightSamples& lightSamples,
                                             Buffer<unsigned int>& recomputationImportance, bool fixExitPoint);
    void setPercentage(int p) { percentage_ = p; }
    int getPercentage() const { return percentage_; }
    void setIteration(int i) { iteration_ = i; }
    int getIteration() const { return iteration_; }
    void setEqualImportance(bool e) { equalImportance_ = e; }
    bool getEqualImportance() const { return equalImportance_; }
private:
    int percentage_ = 100, iteration_ = 0;
    bool equalImportance_ = false;
};

// ---- processors (L4) ----------------------------------------------------------------------------------

// importancesamplingcl/processors/uniformsamplegenerator2dprocessorcl.{h,cpp}
class UniformSampleGenerator2DProcessorCL : public Processor {
public:
    UniformSampleGenerator2DProcessorCL();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.UniformSampleGenerator2DCL", "UniformSampleGenerator2DCL", "Sampling" }; }
    void process() override;
    DataOutport<SampleBuffer> samplesPort_{ "samples" };
    DataOutport<SampleBuffer> directionalSamplesPort_{ "DirectionalSamples" };  // a copy of the samples, filled when connected (:79-101)
    Property<ivec2> nSamplesProp_{ "nSamples", "N samples", ivec2{ 256, 256 } };
    IntVec2Property workGroupSize_{ "wgsize", "Work group size", ivec2{ 8, 8 } };  // inert
    BoolProperty useGLSharing_{ "glsharing", "Use OpenGL sharing", true };           // inert
private:
    UniformSampleGenerator2DCL generator_;
    std::shared_ptr<SampleBuffer> samples_ = std::make_shared<SampleBuffer>();
    std::shared_ptr<SampleBuffer> directionalSamples_ = std::make_shared<SampleBuffer>();
};

// lightcl/processors/directionallightsamplerclprocessor.{h,cpp}
class DirectionalLightSamplerCLProcessor : public Processor {
public:
    DirectionalLightSamplerCLProcessor();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.DirectionalLightSamplerCL", "DirectionalLightSamplerCL", "Light sampling" }; }
    void process() override;
    DataInport<Mesh> boundingVolumePort_{ "SceneGeometry" };
    DataInport<SampleBuffer> samplesPort_{ "samples" };
    DataInport<DirectionalLight> lightsPort_{ "light" };
    DataOutport<LightSamples> lightSamplesPort_{ "LightSamples" };
    IntProperty workGroupSize_{ "wgsize", "Work group size", 64 };  // inert: kept for the workspace
    DirectionalLightSamplerCL lightSampler_;
private:
    LightSampleMeshIntersectionCL intersector_;
    std::shared_ptr<LightSamples> lightSamples_ = std::make_shared<LightSamples>();
};

// uniformgridcl/processors/volumeminmaxclprocessor.{h,cpp}
class VolumeMinMaxCLProcessor : public Processor {
public:
    VolumeMinMaxCLProcessor();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.VolumeMinMaxCLProcessor", "VolumeMinMaxCLProcessor", "UniformGrid3D" }; }
    void process() override;
    DataInport<Volume> inport_{ "volume" };
    DataOutport<UniformGrid3DBase> outport_{ "output" };
    DataInport<VolumeSequence> vectorInport_{ "VolumeSequenceInput" };          // :62
    DataOutport<UniformGrid3DVector> vectorOutport_{ "UniformGrid3DVectorOut" };  // :63
    IntProperty volumeRegionSize_{ "region", "Region size", 8 };
    Property<ivec2> workGroupSize_{ "wgsize", "Work group size", ivec2{ 4, 4 } };   // inert (ivec3 in the reference)
    BoolProperty useGLSharing_{ "glsharing", "Use OpenGL sharing", true };          // inert
private:
    std::shared_ptr<MinMaxUniformGrid3D> compute(const Volume* volume);  // :148-184
};

// importancesamplingcl/processors/minmaxuniformgrid3dimportanceclprocessor.{h,cpp}
class MinMaxUniformGrid3DImportanceCLProcessor : public Processor {
public:
    MinMaxUniformGrid3DImportanceCLProcessor();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.MinMaxUniformGrid3DImportanceCLProcessor", "MinMaxUniformGrid3DImportance", "UniformGrid3D" }; }
    void process() override;
    // TF edit entry point (the TransferFunctionProperty's onChange in the reference, :86-90)
    void setTransferFunction(const TransferFunction& tf);
    DataInport<UniformGrid3DBase> minMaxUniformGrid3DInport_{ "minMaxUniformGrid3D" };
    DataInport<UniformGrid3DBase> volumeDifferenceInfoInport_{ "volumeDifferenceInfo" };
    DataOutport<UniformGrid3DBase> importanceUniformGrid3DOutport_{ "importanceUniformGrid3D" };
    BoolProperty incrementalImportance{ "incrementalImportance", "Incremental importance", true };
    BoolProperty useAssociatedColor_{ "useAssociatedColor", "Associated color", false };
    FloatProperty TFPointEpsilon_{ "TFPointEpsilon", "Minimum change threshold", 1e-4f };
    // weights of the non-incremental importance (tfPointsImportance without -D INCREMENTAL_TF_IMPORTANCE): the reference builds
    // the kernel WITH the define (.cpp:101), so they do not enter the result; kept for the workspace
    FloatProperty opacityWeight_{ "constantWeight", "Opacity weight", 1.f }, opacityDiffWeight_{ "opacityDiffWeight", "Opacity difference weight", 0.f },
        colorWeight_{ "colorWeight", "Color weight", 0.f }, colorDiffWeight_{ "colorDiffWeight", "Color difference weight", 0.f };
    TransferFunctionProperty transferFunctionProperty_{ "transferfunction", "Transfer function", TransferFunction() };  // .set() == setTransferFunction
    IntProperty workGroupSize_{ "wgsize", "Work group size", 128 };      // inert
    BoolProperty useGLSharing_{ "glsharing", "Use OpenGL sharing", true };  // inert
    const std::vector<float>& tfPointPositions() const { return positions_; }
    const std::vector<vec4>& tfPointColors() const { return colors_; }
private:
    void updateTransferFunctionData();            // :304-362
    void updateTransferFunctionDifferenceData();  // :364-501
    TransferFunction transferFunction_, prevTransferFunction_;
    bool tfChanged_ = false;
    std::shared_ptr<const MinMaxUniformGrid3D> prevMinMaxUniformGrid3D_;  // .h:126: the grid of the previous evaluation (time-varying data)
    std::vector<float> positions_;
    std::vector<vec4> colors_;
    std::shared_ptr<ImportanceUniformGrid3D> importance_ = std::make_shared<ImportanceUniformGrid3D>();
};

// progressivephotonmapping/processor/progressivephotontracercl.{h,cpp}
class ProgressivePhotonTracerCL : public Processor {
public:
    ProgressivePhotonTracerCL();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.ProgressivePhotonTracerCL", "ProgressivePhotonTracer", "Photons" }; }
    void process() override;
    void invalidateProgressiveRendering(PhotonData::InvalidationReason r) { invalidationFlag_ = invalidationFlag_ | r; }
    // One firing of the 100 ms refinement timer (the caller owns the timer; ref progressivephotontracercl.cpp:622-626,
    // 642-645): the next evaluation is a progressive iteration -- same light samples, continued RNG streams, next radius.
    void onTimerEvent() { invalidationFlag_ = invalidationFlag_ | PhotonData::InvalidationReason::Progressive; }
    // ref :647-651: the tracer writes its RNG state back only when refinement is on and no importance grid is connected
    void progressiveRefinementChanged() { photonTracer_.setProgressive(enableProgressiveRefinement_.get() && !recomputationImportanceGrid_.isConnected()); }
    void setTransferFunction(const TransferFunction& tf) { transferFunction_ = tf; invalidateProgressiveRendering(PhotonData::InvalidationReason::TransferFunction); }
    int remainingPhotonsToUpdate() const { return remainingPhotonsToUpdate_; }

    DataInport<Volume> volumePort_{ "volume" };
    DataInport<UniformGrid3DBase> recomputationImportanceGrid_{ "recomputationImportance" };
    DataInport<LightSamples> lightSamples_{ "LightSamples" };  // multi-inport
    DataOutport<PhotonData> outport_{ "photons" };
    DataOutport<RecomputedPhotonIndices> recomputedIndicesPort_{ "recomputedIndices" };

    FloatProperty samplingRate_{ "samplingRate", "Sampling rate", 1.0f };
    FloatProperty radius_{ "radius", "Photon radius (# voxels)", 1.f };
    FloatProperty sceneRadianceScaling_{ "radianceScale", "Scene radiance scale", 1.f };
    FloatProperty maxIncrementalPhotonsToUpdate_{ "maxIncrementalPhotonsToUpdate", "Max photons per update (%)", 100.f };
    BoolProperty equalIncrementalImportance_{ "equalImportance", "Equal importance", false };
    BoolProperty spatialSorting_{ "spatialSorting", "Spatial sorting", true };
    IntProperty maxScatteringEvents_{ "maxScatteringEvents", "Max scattering events", 1 };
    BoolProperty noSingleScattering_{ "noSingleScattering", "No single scattering", false };
    FloatProperty alphaProp_{ "alpha", "Progressive alpha", 0.5f };
    IntVec2Property workGroupSize_{ "wgsize", "Work group size", ivec2{ 8, 8 } };   // inert: kept for the workspace
    BoolProperty useGLSharing_{ "glsharing", "Use OpenGL sharing", true };          // inert
    BoolProperty enableProgressiveRefinement_{ "enableRefinement", "Progressive refinement", false };
    BoolProperty enableProgressivePhotonRecomputation_{ "enableProgressiveRecomputation", "Progressive recomputation", true };
    Property<ivec2> clipX_{ "clipX", "Clip X Slices", ivec2{ 0, 256 } }, clipY_{ "clipY", "Clip Y Slices", ivec2{ 0, 256 } },
        clipZ_{ "clipZ", "Clip Z Slices", ivec2{ 0, 256 } };
    AdvancedMaterialProperty advancedMaterial_;
    CameraProperty camera_;
    ButtonProperty invalidateRendering_{ "invalidate", "Invalidate rendering" };
    TransferFunctionProperty transferFunctionProperty_{ "transferFunction", "Transfer function", TransferFunction() };  // .set() == setTransferFunction
    TransferFunction transferFunction_;
    PhotonTracerCL photonTracer_;
    bool fixExitPoint = false;
    // true (default): detector, threshold and tracer in ONE launch (cpm_photon_importance_retrace); false: selection, compaction
    // and cpm_trace_selected as separate launches (still no host round trip).  The equal-importance detector takes the latter.
    BoolProperty retraceInImportancePass_{ "retraceInImportancePass", "Re-trace inside the importance pass", true };
    BoolProperty traceLightsInOneLaunch_{ "traceLightsInOneLaunch", "Full frames trace all lights in one launch", true };
    // false: the importance branch launch by launch with its host read of the count in the middle (always taken when the update
    // budget is below 100 %: ranking by importance is a host decision); true (default): the count stays on the device
    BoolProperty fusedImportanceBranch_{ "fusedImportanceBranch", "Importance branch without host round trip", true };
    // "adaptive": where the measured cost of importance branch + add-remove exceeds that of re-tracing and rebuilding
    // everything (CpmRuntime::PathCosts), a TF / volume change is served by the full frame -- the same photons (correlated RNG
    // streams), the same light volume within the add-remove tolerance; the branch is measured again every 32nd such evaluation.
    // Applies when every changed photon would be traced in one evaluation (budget 100 %).  Not a reference property.
    // "adaptive" (default) | "always" (the reference's behaviour: the branch on every TF / volume change) | "never" (the full frame)
    StringOptionProperty importanceBranchPolicy_{ "importanceBranchPolicy", "Importance branch", "adaptive" };
    const char* lastDecision() const { return lastDecision_; }
    PathCosts& costs() { return costs_; }
    void pollCosts() { span_.poll(); }  // picks up a finished evaluation's span now (process() does so by itself)
    // measurement aid (not a reference property): with equalImportance on, select every (100 / p)-th photon while the update
    // budget stays what maxIncrementalPhotonsToUpdate says (the reference uses that one property for both: 0 = as the reference)
    IntProperty equalImportancePercentage_{ "equalImportancePercentage", "Equal importance: percentage selected", 0 };
    ~ProgressivePhotonTracerCL();
private:
    cpm_selection* selection_ = nullptr;
    size_t selectionPhotons_ = 0;
    StreamSpan span_;
    PathCosts costs_;
    const char* lastDecision_ = "none";
    void onClipChange();
    void publishPhotons();
    float getSceneRadius() const { return 0.5f * std::sqrt(12.f); }  // unit-model volume spanning [-1, 1]^3
    void resetPhotonImportance(size_t offset, size_t n);
    std::shared_ptr<PhotonData> photonData_ = std::make_shared<PhotonData>();
    std::shared_ptr<RecomputedPhotonIndices> recomputedPhotonIndices_ = std::make_shared<RecomputedPhotonIndices>();
    PhotonRecomputationDetector photonRecomputationDetector_;
    Buffer<unsigned int> photonRecomputationImportance_;
    bool importancesAreReset_ = false;  // every key is 0x7fffffff (no importance pass since the last whole-buffer reset)
    Buffer<int> nChanged_{ 1 };
    PhotonData::InvalidationReason invalidationFlag_ = PhotonData::InvalidationReason::All;
    float aabb_[8] = { 0, 0, 0, 1, 1, 1, 1, 1 };
    int remainingPhotonsOffset_ = 0, remainingPhotonsToUpdate_ = -1;
    bool rankedByImportance_ = true;  // indices / importances are sorted by importance (cpm_select_recompute ran)
    std::vector<std::pair<const LightSamples*, size_t>> seenLights_;  // (samples, change stamp) at the last evaluation
};

// progressivephotonmapping/processor/photontolightvolumeprocessorcl.{h,cpp}
class PhotonToLightVolumeProcessorCL : public Processor {
public:
    PhotonToLightVolumeProcessorCL();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.PhotonToLightVolumeProcessorCL", "PhotonToLightVolumeProcessorCL", "Photons" }; }
    void process() override;
    DataInport<Volume> volumeInport_{ "volume" };
    DataInport<PhotonData> photons_{ "photons" };
    DataInport<RecomputedPhotonIndices> recomputedPhotonIndicesPort_{ "recomputedPhotonIndices" };
    DataOutport<Volume> outport_{ "lightvolume" };
    FloatProperty incrementalRecomputationThreshold_{ "incrementalRecomputationThreshold", "Max % invalid photons to use add-remove", 50.f };
    IntProperty volumeSizeOption_{ "volumeSizeOption", "Light Volume Size", 0 };  // 0 = radius, 1, 2, 4 = input dims / n
    StringOptionProperty volumeDataTypeOption_{ "volumeDataType", "Output data type", "float32" };
    BoolProperty alignChangedPhotons_{ "alignChangedPhotons", "Mem-align changed photons", false };
    IntProperty workGroupSize_{ "wgsize", "Work group size", 128 };      // inert
    BoolProperty useGLSharing_{ "glsharing", "Use OpenGL sharing", true };  // see shareLightVolumeWithGL
    // "gather" (default: sort/bin + deterministic gather) or "splat" (the reference's atomic formulation)
    // VolumeInformationProperty "Information" of the output volume (read-only in the reference's UI)
    struct InformationProperty : CompositeProperty {
        InformationProperty() : CompositeProperty("Information", "Light volume information") {
            addProperty(dimensions); addProperty(format); addProperty(dataRange); addProperty(valueRange); addProperty(valueUnit);
        }
        StringOptionProperty dimensions{ "dimensions", "Dimensions", "" }, format{ "format", "Format", "FLOAT32" };
        FloatVec2Property dataRange{ "dataRange", "Data range", vec2{ 0.f, 1.f } }, valueRange{ "valueRange", "Value range", vec2{ 0.f, 1.f } };
        StringOptionProperty valueUnit{ "valueUnit", "Value unit", "arb. unit." };
    } information_;
    // "fast" (default): brick bin + LDS-tile gather with fixed-point sums (tolerance mode, bitwise reproducible; falls back
    // to "gather" where cpm_gather_fast_supported_on says no for the context's device); "gather": cell sort + per-voxel sequential gather (bit-exact
    // contract, what exactIncrementalUpdate needs); "splat": the reference's atomic formulation
    StringOptionProperty formulation_{ "formulation", "Density estimation", "fast" };
    // add-remove of the re-traced photons: false (default) = the reference's -old / +new atomic splats; true = re-bin and
    // re-gather exactly the bricks they touch (bit-identical to a full gather) -- not a property of the reference
    BoolProperty exactIncrementalUpdate_{ "exactIncrementalUpdate", "Exact incremental update", false };
    // progressive refinement: an evaluation whose only invalidation reason is Progressive (iteration i > 1) contributes its
    // estimate E_i to the running average L_i = L_(i-1) + (E_i - L_(i-1)) / i instead of replacing the light volume.  Not a
    // property of the reference (it re-splats every iteration and leaves the averaging to the consumer).
    BoolProperty progressiveAccumulation_{ "progressiveAccumulation", "Average progressive iterations", true };
    const char* lastPath() const { return lastPath_; }
    void pollCosts() { span_.poll(); }
    // Multi-GPU: this processor's photons are ONE shard of the frame's photons and lightVolume_ is the shard's
    // partial light volume; with a communicator set, the outport carries the sum over the shards -- one cpm_allreduce_grid
    // per full evaluation, cpm_allreduce_grid_bricks (touched bricks only) after an add-remove update.  The call site is
    // where the reference hands the finished volume to the outport (photontolightvolumeprocessorcl.cpp:404-411).
    void setCommunicator(cpm_comm* comm);
    const char* lastReduce() const { return lastReduce_; }
    // OpenGL sharing (`glsharing`, ref photontolightvolumeprocessorcl.cpp:184-194,404-406).  CDNA cannot map a GL texture (no
    // image hardware); the finished light volume is written on the device into a GL pixel-unpack BUFFER of the host's context
    // (VolumeCLGL + enqueueCopyBufferToImage in the reference), from which the host issues glTexSubImage3D into the raycaster's
    // texture.  The host names its buffer once; registration happens at the next evaluation with the GL context current
    // (Inviwo's processors run on the GL thread).  texel: CPM_GL_TEXEL_F32 / _F16.  Without a current context (or with
    // `glsharing` off) the outport's device buffer is all there is -- as before.
    void shareLightVolumeWithGL(unsigned glPixelUnpackBuffer, int texel) { glBufferName_ = glPixelUnpackBuffer; glTexel_ = texel; dropGLBuffer(); }
    const char* lastGLCopy() const { return lastGLCopy_; }
    ~PhotonToLightVolumeProcessorCL() override { dropGLBuffer(); dropSparseReduce(); }
private:
    void dropGLBuffer();
    void copyToGLBuffer(const float* volume, size_t n);
    unsigned glBufferName_ = 0;
    int glTexel_ = CPM_GL_TEXEL_F32;
    cpm_gl_resource* glBuffer_ = nullptr;
    const char* lastGLCopy_ = "none";  // "none" | "copied" | "no context" | "failed"
    void volumeSizeOptionChanged();
    void reduceOverShards(const cpm_grid_desc& g, size_t count, bool partialUpdate, const float* prevPhotons, const float* photons,
                          const unsigned int* idx, int nRecomputed, int nPhotons, int nInter, float radius);
    cpm_comm* comm_ = nullptr;
    cpm_sparse_reduce* sparseReduce_ = nullptr;  // cpm_allreduce_grid_sparse state of (comm_, the light volume's shape)
    Buffer<uint8_t> nonzeroMarks_;               // cpm_gather_fast_marked: the non-zero 4x4x4 bricks of the volume just written
    bool marksAreNonzero_ = false;               // ... valid for this evaluation's lightVolume_
    Buffer<uint8_t> litBefore_, allBricks_;      // multi-GPU: the bricks this shard has lit since its last rebuild / the mask a rebuild hands in
    bool litBeforeValid_ = false, firstSumIntoReduced_ = false;
    size3_t sparseReduceDims_{ 0, 0, 0 };
    int sparseReduceChannels_ = 0;
    void dropSparseReduce();
    std::shared_ptr<Volume> reducedVolume_;
    Buffer<uint32_t> brickTable_;
    Buffer<float> estimate_;  // E_i of a progressive iteration
    const char* lastReduce_ = "none";
    std::shared_ptr<Volume> lightVolume_ = std::make_shared<Volume>(size3_t{ 1, 1, 1 }, CPM_F32);
    Buffer<vec4> prevPhotons_;
    StreamSpan span_;
    bool prevPhotonsValid_ = false;  // prevPhotons_ is the photon buffer as of the end of the previous evaluation
    bool snapshotFree_ = true;       // no whole-buffer snapshot while the tracer hands over the replaced records (set false to keep the reference's copy)
    Buffer<unsigned int> order_, cellStart_;
    Buffer<float> sorted_;
    Buffer<uint8_t> brickMask_;
    const char* lastPath_ = "none";
};

}  // namespace inviwo
