// inviwo_lite.h -- the sliver of Inviwo's public API the six modules' processors touch, so that
// the host layer of this build compiles and runs without Inviwo (which is not in this image).
// A maintainer integrating into a real Inviwo checkout deletes this header and includes Inviwo's
// own: names, ids and call shapes below are Inviwo's.
//
// What differs on purpose: Buffer<T>/Volume keep ONE device representation (HIP memory) next to
// the RAM one instead of Inviwo's RAM/CL/GL/CLGL zoo -- there is no OpenCL or OpenGL here.
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

struct cpm_volume;  // include/cpm/cpm.h

namespace inviwo {

using mat4 = std::array<float, 16>;  // column-major like glm::mat4
inline mat4 identityMatrix() { return { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 }; }

struct vec2 { float x = 0, y = 0; };
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
};
struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
};
struct ivec2 { int x = 0, y = 0; };
struct uvec2 { uint32_t x = 0, y = 0; };
struct size3_t { size_t x = 0, y = 0, z = 0; };
inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, vec3 a) { return a * s; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / length(a)); }

void LogErrorImpl(const std::string& source, const std::string& msg);
void LogInfoImpl(const std::string& source, const std::string& msg);
#define LogError(msg) ::inviwo::LogErrorImpl(__func__, std::string(msg))
#define LogInfo(msg) ::inviwo::LogInfoImpl(__func__, std::string(msg))

// ---- data --------------------------------------------------------------------------------------

template <typename T>
class Buffer {
public:
    explicit Buffer(size_t n = 0) { setSize(n); }
    ~Buffer() { release(); }
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    void setSize(size_t n) {
        if (n == size_) return;
        release();
        size_ = n;
        ram_.clear();
    }
    size_t getSize() const { return size_; }
    size_t getSizeInBytes() const { return size_ * sizeof(T); }
    // device representation (allocated on first use)
    T* device() {
        if (!dev_ && size_) {
            if (hipMalloc((void**)&dev_, size_ * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
        }
        return dev_;
    }
    const T* device() const { return const_cast<Buffer*>(this)->device(); }
    // RAM representation: download on request, upload explicitly
    std::vector<T>& ram() { ram_.resize(size_); return ram_; }
    void upload(hipStream_t s = nullptr) { if (size_) (void)hipMemcpyAsync(device(), ram_.data(), size_ * sizeof(T), hipMemcpyHostToDevice, s); }
    void download(hipStream_t s = nullptr) {
        ram_.resize(size_);
        if (size_) { (void)hipMemcpyAsync(ram_.data(), device(), size_ * sizeof(T), hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
    }
    // getRepresentation<BufferRAM>(): once a device block exists it is the master copy
    bool hasDevice() const { return dev_ != nullptr; }
    T* hostData() {
        if (dev_) download(); else ram_.resize(size_);
        return ram_.data();
    }

private:
    void release() { if (dev_) (void)hipFree(dev_); dev_ = nullptr; }
    size_t size_ = 0;
    T* dev_ = nullptr;
    std::vector<T> ram_;
};

struct TFPrimitive {
    double pos = 0;
    vec4 color;
};

class TransferFunction {
public:
    void clear() { points_.clear(); }
    void add(double pos, vec4 color) { points_.push_back({ pos, color }); sort(); }
    size_t size() const { return points_.size(); }
    const TFPrimitive& get(size_t i) const { return points_[i]; }
    std::vector<TFPrimitive>& points() { return points_; }
    const std::vector<TFPrimitive>& points() const { return points_; }
    // Inviwo: 1024 x 1 RGBA32F layer, texel i at (i + 0.5) / width, constant outside the end points
    std::vector<float> lut(int width = 1024) const;

private:
    void sort();
    std::vector<TFPrimitive> points_;
};

class Volume {
public:
    Volume(size3_t dims, int dtype) : dims_(dims), dtype_(dtype) {}
    ~Volume();
    Volume(const Volume&) = delete;
    Volume& operator=(const Volume&) = delete;
    size3_t getDimensions() const { return dims_; }
    void setDimensions(size3_t d) { dims_ = d; data.setSize(0); invalidateDeviceRepresentation(); }
    int dtype() const { return dtype_; }
    size_t elementSize() const { return dtype_ == 0 ? 1 : (dtype_ == 1 ? 2 : 4); }
    const mat4& getModelMatrix() const { return model_; }
    const mat4& getWorldMatrix() const { return world_; }
    void setModelMatrix(const mat4& m) { model_ = m; }
    void setWorldMatrix(const mat4& m) { world_ = m; }
    int channels = 1;               // light volumes: 1 (float32) or 4 (4xfloat32)
    std::vector<uint8_t> ramBytes;  // scalar source volumes live here until uploaded
    Buffer<float> data;             // light volumes: device float storage
    // getRepresentation<VolumeCL>() of a scalar volume: uploaded from ramBytes on first use and shared by every
    // processor that reads the volume; zero-filled device storage when there is no RAM data (a volume produced
    // on the device, e.g. VolumeSequencePlayer's output).  nullptr (error logged) without a device.
    ::cpm_volume* getDeviceRepresentation() const;
    void invalidateDeviceRepresentation();  // after editing ramBytes
    // getRepresentation<VolumeRAM>() of a device-produced volume: download into ramBytes
    bool downloadToRAM();
private:
    size3_t dims_;
    int dtype_;
    mat4 model_ = identityMatrix(), world_ = identityMatrix();
    mutable ::cpm_volume* dev_ = nullptr;
};
using VolumeSequence = std::vector<std::shared_ptr<Volume>>;

// ---- ports, properties, processors -----------------------------------------------------------------

template <typename T>
class DataOutport {
public:
    explicit DataOutport(std::string id) : id_(std::move(id)) {}
    const std::string& getIdentifier() const { return id_; }
    void setData(std::shared_ptr<T> d) { data_ = std::move(d); ++stamp_; }
    std::shared_ptr<T> getData() const { return data_; }
    bool isConnected() const { return nConnections_ > 0; }
    void connectionAdded() { ++nConnections_; }
    size_t stamp() const { return stamp_; }  // advances with every setData: what invalidates the connected inports
private:
    std::string id_;
    std::shared_ptr<T> data_;
    int nConnections_ = 0;
    size_t stamp_ = 0;
};

template <typename T>
class DataInport {
public:
    explicit DataInport(std::string id) : id_(std::move(id)) {}
    const std::string& getIdentifier() const { return id_; }
    void connectTo(DataOutport<T>* out) { sources_.push_back(out); out->connectionAdded(); if (onConnect_) onConnect_(); }
    void disconnectAll() { sources_.clear(); }
    bool isConnected() const { return !sources_.empty(); }
    bool isReady() const { return !sources_.empty() && sources_[0]->getData() != nullptr; }
    bool hasData() const { return isReady(); }
    std::shared_ptr<T> getData() const { return sources_.empty() ? nullptr : sources_[0]->getData(); }
    std::vector<std::shared_ptr<T>> getVectorData() const {  // multi-inport
        std::vector<std::shared_ptr<T>> v;
        for (auto* s : sources_) if (s->getData()) v.push_back(s->getData());
        return v;
    }
    void setOptional(bool o) { optional_ = o; }
    void onConnect(std::function<void()> f) { onConnect_ = std::move(f); }
    // Inport::onChange, polled: has the source port been given data since the last call?  (Inviwo fires the callback
    // when the upstream processor sets its outport; the processors here ask at the start of process().)
    bool changedSinceLastCheck() {
        const size_t now = sources_.empty() ? 0 : sources_[0]->stamp();
        const DataOutport<T>* src = sources_.empty() ? nullptr : sources_[0];
        const bool ch = now != seenStamp_ || src != seenSource_;
        seenStamp_ = now; seenSource_ = src;
        return ch;
    }
private:
    std::string id_;
    std::vector<DataOutport<T>*> sources_;
    bool optional_ = false;
    std::function<void()> onConnect_;
    size_t seenStamp_ = 0;
    const DataOutport<T>* seenSource_ = nullptr;
};

class PropertyBase {
public:
    PropertyBase(std::string id, std::string name) : id_(std::move(id)), name_(std::move(name)) {}
    virtual ~PropertyBase() = default;
    const std::string& getIdentifier() const { return id_; }
    void onChange(std::function<void()> f) { onChange_ = std::move(f); }
protected:
    void changed() { if (onChange_) onChange_(); }
    std::string id_, name_;
    std::function<void()> onChange_;
};

template <typename T>
class Property : public PropertyBase {
public:
    Property(std::string id, std::string name, T value) : PropertyBase(std::move(id), std::move(name)), value_(value) {}
    const T& get() const { return value_; }
    operator T() const { return value_; }
    void set(const T& v) { value_ = v; changed(); }
    // ordinal properties: range (no clamping here, as OrdinalProperty::set does not clamp either)
    const T& getMinValue() const { return min_; }
    const T& getMaxValue() const { return max_; }
    void setMinValue(const T& v) { min_ = v; }
    void setMaxValue(const T& v) { max_ = v; }
    void setReadOnly(bool r) { readOnly_ = r; }
    bool getReadOnly() const { return readOnly_; }
private:
    T value_;
    T min_{}, max_{};
    bool readOnly_ = false;
};
// ButtonProperty: pressButton() runs the onChange callback
class ButtonProperty : public PropertyBase {
public:
    using PropertyBase::PropertyBase;
    void pressButton() { changed(); }
};
// CompositeProperty: a named group; its members are also reachable by their own identifiers from the processor
class CompositeProperty : public PropertyBase {
public:
    using PropertyBase::PropertyBase;
    void addProperty(PropertyBase& p) { members_.push_back(&p); }
    const std::vector<PropertyBase*>& getProperties() const { return members_; }
private:
    std::vector<PropertyBase*> members_;
};
using FloatProperty = Property<float>;
using IntProperty = Property<int>;
using BoolProperty = Property<bool>;
using IntVec2Property = Property<ivec2>;
using StringOptionProperty = Property<std::string>;
using FloatVec2Property = Property<vec2>;
using FloatVec3Property = Property<vec3>;
using FloatVec4Property = Property<vec4>;
using TransferFunctionProperty = Property<TransferFunction>;

struct ProcessorInfo {
    std::string classIdentifier, displayName, category;
};

class Processor {
public:
    virtual ~Processor() = default;
    virtual const ProcessorInfo getProcessorInfo() const = 0;
    virtual void process() = 0;
    void addProperty(PropertyBase& p) {
        properties_[p.getIdentifier()] = &p;
        if (auto* c = dynamic_cast<CompositeProperty*>(&p))
            for (PropertyBase* m : c->getProperties()) addProperty(*m);
    }
    void addPortId(const std::string& id, bool inport) { (inport ? inports_ : outports_).push_back(id); }
    PropertyBase* getPropertyByIdentifier(const std::string& id) const {
        auto it = properties_.find(id);
        return it == properties_.end() ? nullptr : it->second;
    }
    const std::vector<std::string>& getInportIds() const { return inports_; }
    const std::vector<std::string>& getOutportIds() const { return outports_; }
    std::vector<std::string> getPropertyIds() const {
        std::vector<std::string> v;
        for (auto& kv : properties_) v.push_back(kv.first);
        return v;
    }
private:
    std::map<std::string, PropertyBase*> properties_;
    std::vector<std::string> inports_, outports_;
};

}  // namespace inviwo
