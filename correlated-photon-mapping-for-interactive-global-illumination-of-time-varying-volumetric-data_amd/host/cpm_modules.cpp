// cpm_modules.cpp -- see cpm_modules.h
#include "cpm_modules.h"

namespace inviwo {

ProcessorFactory& ProcessorFactory::get() {
    static ProcessorFactory f;
    return f;
}
std::unique_ptr<Processor> ProcessorFactory::create(const std::string& id) const {
    auto it = creators_.find(id);
    if (it != creators_.end()) return it->second();
#ifndef CPM_HOST_EXTRAS
    // the four processors of the reference's modules that are not on the workspace's path exist only in the -DCPM_HOST_EXTRAS build
    // (libcpm_host_extras.so): say so instead of "unknown processor"
    for (const char* extra : { "org.inviwo.RadixSortCL", "org.inviwo.UniformGrid3DExport", "org.inviwo.UniformGrid3DSequenceSelector",
                               "org.inviwo.UniformGrid3DSourceProcessor", "org.inviwo.UniformGrid3DVectorSource" })
        if (id == extra) {
            LogError("processor " + id + " is not in this library: it is built with -DCPM_HOST_EXTRAS (libcpm_host_extras.so)");
            return nullptr;
        }
#endif
    LogError("unknown processor class identifier: " + id);
    return nullptr;
}
std::vector<std::string> ProcessorFactory::getKeys() const {
    std::vector<std::string> k;
    for (auto& kv : creators_) k.push_back(kv.first);
    return k;
}

#ifdef CPM_HOST_EXTRAS
RadixSortCL::RadixSortCL() { addPortId("unsortedKeys", true); addPortId("unsortedData", true); addPortId("sortedData", false); }
void RadixSortCL::process() {
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || !keysPort_.isReady() || !inputPort_.isReady()) return;
    auto keys = keysPort_.getData();
    auto data = inputPort_.getData();
    if (keys->getSize() != data->getSize()) { LogError("RadixSortCL: keys and data differ in size"); return; }
    if (!keys->hasDevice()) keys->upload(rt.stream());
    if (!data->hasDevice()) data->upload(rt.stream());
    // maxBits = 0: all key bits (radixsortcl.cpp:238)
    rt.check(cpm_sort_pairs(rt.ctx(), keys->device(), data->device(), keys->getSize(), 0, rt.stream()), "cpm_sort_pairs");
    outputPort_.setData(data);  // pass-through, as the reference does (:255-258)
}
#endif

ProgressivePhotonMappingModule::ProgressivePhotonMappingModule() : InviwoModule("ProgressivePhotonMapping") {
    registerProcessor<PhotonToLightVolumeProcessorCL>();
    registerProcessor<ProgressivePhotonTracerCL>();
    registerPort("PhotonData", "Inport");
    registerPort("PhotonData", "Outport");
    registerPort("RecomputedPhotonIndices", "Inport");   // photondata.h:188-189, used by the workspace (.inv:548)
    registerPort("RecomputedPhotonIndices", "Outport");
}
LightCLModule::LightCLModule() : InviwoModule("LightCL") {
    registerProcessor<DirectionalLightSamplerCLProcessor>();
    registerPort("LightSamples", "Inport");
    registerPort("LightSamples", "Outport");
    registerPort("LightSamples", "MultiInport");
}
RndGenMWC64XModule::RndGenMWC64XModule() : InviwoModule("RndGenMWC64X") {}
UniformGridCLModule::UniformGridCLModule() : InviwoModule("UniformGridCL") {
    registerProcessor<DynamicVolumeDifferenceAnalysis>();
    registerProcessor<UniformGrid3DPlayerProcessor>();
#ifdef CPM_HOST_EXTRAS
    registerProcessor<UniformGrid3DExport>();
    registerProcessor<UniformGrid3DSequenceSelector>();
    registerProcessor<UniformGrid3DVectorSource>();  // stands for UniformGrid3DSourceProcessor (reads a .u3d sequence)
#endif
    registerProcessor<VolumeMinMaxCLProcessor>();
    registerProcessor<VolumeSequencePlayer>();
    registerDataReaderWriter("u3d");
    registerPort("UniformGrid3DBase", "Inport");
    registerPort("UniformGrid3DBase", "Outport");
}
ImportanceSamplingCLModule::ImportanceSamplingCLModule() : InviwoModule("ImportanceSamplingCL") {
    registerProcessor<MinMaxUniformGrid3DImportanceCLProcessor>();
    registerProcessor<UniformSampleGenerator2DProcessorCL>();
}
RadixSortCLModule::RadixSortCLModule() : InviwoModule("RadixSortCL") {
#ifdef CPM_HOST_EXTRAS
    registerProcessor<RadixSortCL>();  // (the path sorts through cpm_sort_* / cpm_bin, not through this node)
#endif
}

std::vector<std::unique_ptr<InviwoModule>> registerCorrelatedPhotonMappingModules() {
    std::vector<std::unique_ptr<InviwoModule>> m;
    m.emplace_back(new ProgressivePhotonMappingModule());
    m.emplace_back(new LightCLModule());
    m.emplace_back(new RndGenMWC64XModule());
    m.emplace_back(new UniformGridCLModule());
    m.emplace_back(new ImportanceSamplingCLModule());
    m.emplace_back(new RadixSortCLModule());
    return m;
}

}  // namespace inviwo
