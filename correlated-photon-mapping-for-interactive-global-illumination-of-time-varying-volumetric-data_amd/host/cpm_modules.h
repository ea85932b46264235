// cpm_modules.h -- the six Inviwo modules of the reference as registration units over libcpm_hip, and the generic
// RadixSortCL processor.  In an Inviwo checkout each class below is the body of the module's existing
// <name>module.cpp (same module identifier, same registerProcessor / registerPort / registerDataReader calls, no
// OpenCL include directories); here they register into the small factory of inviwo_lite so that the surface can be
// instantiated by class identifier, the way a workspace (.inv) is deserialised.
#pragma once
#include "cpm_timevarying.h"

namespace inviwo {

class ProcessorFactory {
public:
    using Creator = std::function<std::unique_ptr<Processor>()>;
    static ProcessorFactory& get();
    void registerObject(const std::string& classIdentifier, Creator c) { creators_[classIdentifier] = std::move(c); }
    bool hasKey(const std::string& id) const { return creators_.count(id) != 0; }
    std::unique_ptr<Processor> create(const std::string& id) const;
    std::vector<std::string> getKeys() const;
private:
    std::map<std::string, Creator> creators_;
};

class InviwoModule {
public:
    explicit InviwoModule(std::string identifier) : identifier_(std::move(identifier)) {}
    virtual ~InviwoModule() = default;
    const std::string& getIdentifier() const { return identifier_; }
    virtual int getVersion() const { return 0; }
    const std::vector<std::string>& processors() const { return processors_; }
    const std::vector<std::string>& ports() const { return ports_; }
    const std::vector<std::string>& dataFormats() const { return dataFormats_; }
protected:
    template <typename T>
    void registerProcessor() {
        const std::string id = T().getProcessorInfo().classIdentifier;
        processors_.push_back(id);
        ProcessorFactory::get().registerObject(id, []() { return std::unique_ptr<Processor>(new T()); });
    }
    // port class identifiers as a workspace spells them: <DataTraits<T>::dataName()> + Inport / Outport / MultiInport
    void registerPort(const std::string& dataName, const char* kind) { ports_.push_back(dataName + kind); }
    void registerDataReaderWriter(const std::string& ext) { dataFormats_.push_back(ext); }
private:
    std::string identifier_;
    std::vector<std::string> processors_, ports_, dataFormats_;
};

// Processors of the six modules that are NOT on the workspace's path (SURVEY section 2 marks them out of scope): kept behind a
// build flag (-DCPM_HOST_EXTRAS: build.build_host_library(extras=True)); the default library neither compiles nor registers them.
#ifdef CPM_HOST_EXTRAS
// radixsortcl/processors/radixsortcl.{h,cpp}: sorts the key buffer and permutes the data buffer with it, in place, and
// passes the data buffer through (u32 keys, 4-byte data elements; clogs' other 40 type combinations are not built)
class RadixSortCL : public Processor {
public:
    RadixSortCL();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.RadixSortCL", "RadixSortCL", "Sorting" }; }
    void process() override;  // .cpp:208-259
    DataInport<Buffer<uint32_t>> keysPort_{ "unsortedKeys" };
    DataInport<Buffer<uint32_t>> inputPort_{ "unsortedData" };
    DataOutport<Buffer<uint32_t>> outputPort_{ "sortedData" };
};
#endif

// progressivephotonmappingmodule.cpp:43-51
struct ProgressivePhotonMappingModule : InviwoModule { ProgressivePhotonMappingModule(); };
// lightclmodule.cpp:39-84
struct LightCLModule : InviwoModule { LightCLModule(); int getVersion() const override { return 1; } };
// rndgenmwc64xmodule.cpp:40-44 (its two demo processors, RandomNumberGeneratorCL / 2DCL, are outside the path)
struct RndGenMWC64XModule : InviwoModule { RndGenMWC64XModule(); };
// uniformgridclmodule.cpp:49-74
struct UniformGridCLModule : InviwoModule { UniformGridCLModule(); int getVersion() const override { return 1; } };
// importancesamplingclmodule.cpp:42-83
struct ImportanceSamplingCLModule : InviwoModule { ImportanceSamplingCLModule(); int getVersion() const override { return 1; } };
// radixsortclmodule.cpp:53-70
struct RadixSortCLModule : InviwoModule { RadixSortCLModule(); };

// what InviwoApplication::registerModules does with the six factory objects
std::vector<std::unique_ptr<InviwoModule>> registerCorrelatedPhotonMappingModules();

}  // namespace inviwo
