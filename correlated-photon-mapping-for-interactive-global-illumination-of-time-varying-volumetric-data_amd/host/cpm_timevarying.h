// cpm_timevarying.h -- host layer for time-varying data: the UniformGrid3D sequence file format (.u3d)
// and the sequence processors of the reference's uniformgridcl module, over libcpm_hip's C-ABI.
// Same class identifiers, port ids and property ids as the reference (file:line cited per item).
#pragma once
#include "cpm_processors.h"

namespace inviwo {

struct DataReaderException : std::runtime_error { using std::runtime_error::runtime_error; };
struct DataWriterException : std::runtime_error { using std::runtime_error::runtime_error; };

// uniformgridcl/uniformgrid3dwriter.{h,cpp}: <name>.u3d (text header) + <name>.raw (elements back to back)
class UniformGrid3DWriter {
public:
    void setOverwrite(bool o) { overwrite_ = o; }
    void writeData(const UniformGrid3DVector* vectorData, const std::string& filePath) const;  // :47-102
private:
    bool overwrite_ = true;
};
// uniformgridcl/uniformgrid3dreader.{h,cpp}
class UniformGrid3DReader {
public:
    std::shared_ptr<UniformGrid3DVector> readData(const std::string& filePath);  // :59-183
};

#ifdef CPM_HOST_EXTRAS  // (out of SURVEY section 8's scope: see cpm_modules.h)
// uniformgridcl/processors/uniformgrid3dvectorsource.{h,cpp} (DataSource<UniformGrid3DVector, ...>)
class UniformGrid3DVectorSource : public Processor {
public:
    UniformGrid3DVectorSource();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.UniformGrid3DVectorSource", "Uniform Grid 3D Vector Source", "Data Input" }; }
    void process() override;
    DataOutport<UniformGrid3DVector> port_{ "data" };
    StringOptionProperty filePath{ "filename", "UniformGrid3D file", "" };
private:
    std::string loaded_;
};
// uniformgridcl/processors/uniformgrid3dexport.{h,cpp} (DataExport<UniformGrid3DVector, ...>)
class UniformGrid3DExport : public Processor {
public:
    UniformGrid3DExport();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.UniformGrid3DExport", "Uniform Grid 3D Export", "Data Output" }; }
    void process() override {}
    void exportData();  // the export button
    DataInport<UniformGrid3DVector> port_{ "data" };
    StringOptionProperty file_{ "file", "File name", "newvolume.u3d" };
    BoolProperty overwrite_{ "overwrite", "Overwrite", false };
};
#endif

// uniformgridcl/processors/dynamicvolumedifferenceanalysis.{h,cpp}: per brick mean |next - cur| for every time step
// (a CPU loop in the reference, :96-151; cpm_volume_difference here)
class DynamicVolumeDifferenceAnalysis : public Processor {
public:
    DynamicVolumeDifferenceAnalysis();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.DynamicVolumeDifferenceAnalysis", "Dynamic Volume Difference Analysis", "UniformGrid3D" }; }
    void process() override;  // .cpp:59-104
    DataInport<VolumeSequence> inport_{ "data" };
    DataOutport<UniformGrid3DVector> outport_{ "DynamicDataInfo" };
    IntProperty volumeRegionSize_{ "region", "Region size", 8 };
};

// BufferMixerCL (uniformgridcl/buffermixercl.{h,cpp}): out = mix(x, y, a) on the device
class BufferMixerCL {
public:
    void mix(UniformGrid3DBase& x, UniformGrid3DBase& y, float a, UniformGrid3DBase& out);  // :47-92
};

// shared by the two players (uniformgrid3dplayerprocessor.cpp:117-150, volumesequenceplayer.cpp:142-180)
struct SequenceClock {
    FloatProperty time_{ "time", "Time", 0.f };
    IntProperty index_{ "selectedSequenceIndex", "Sequence index", 1 };
    FloatProperty timePerElement_;
    BoolProperty playSequence_{ "playSequence", "Play Sequence", false };
    IntProperty frameRate_;
    SequenceClock(const char* perElementId, const char* perElementName, const char* rateId);
    void onSequenceTimerEvent();            // one tick of the play timer (the caller owns the timer)
    void updateVolumeIndex();
    void onTimeStepChange(size_t nElements);
    // fractional position between element (index - 1) and the next one
    float weight() const { float ip; return std::modf(time_.get() / timePerElement_.get(), &ip); }
};

// uniformgridcl/processors/uniformgrid3dplayerprocessor.{h,cpp}
class UniformGrid3DPlayerProcessor : public Processor {
public:
    UniformGrid3DPlayerProcessor();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.UniformGrid3DPlayerProcessor", "Uniform Grid 3D Player", "UniformGrid3D" }; }
    void process() override;  // :87-115
    void onSequenceTimerEvent() { clock_.onSequenceTimerEvent(); }
    DataInport<UniformGrid3DVector> inport_{ "Sequence" };
    DataOutport<UniformGrid3DBase> outport_{ "InterpolatedData" };
    SequenceClock clock_{ "timePerElement", "Time Per element (s)", "frameRate" };
private:
    BufferMixerCL bufferMixer_;
    std::shared_ptr<UniformGrid3DBase> outData_, outDataPingPong_;
};

// uniformgridcl/processors/volumesequenceplayer.{h,cpp}
class VolumeSequencePlayer : public Processor {
public:
    VolumeSequencePlayer();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.VolumeSequencePlayer", "Volume Sequence Player", "Volume Operation" }; }
    void process() override;  // :87-140
    void onSequenceTimerEvent() { clock_.onSequenceTimerEvent(); }
    ~VolumeSequencePlayer();
    DataInport<VolumeSequence> inport_{ "volumeSequence" };
    DataOutport<Volume> outport_{ "InterpolatedVolume" };
    SequenceClock clock_{ "timePerVolume", "Time Per Volume (s)", "volumesPerSecond" };
    // (not in the reference, whose elements become resident on first use and stay: this build's default too) false: the sequence STAYS IN HOST
    // MEMORY -- the two elements a frame blends come through a ring of three device volumes filled by the library's copy stream
    // (cpm_volume_stream: the element after them is already crossing PCIe while this frame's blend, analyses and update run); for
    // sequences that do not fit on the device, and what SURVEY 8(d) counts as a time step: "upload volume, ..."
    BoolProperty keepSequenceOnDevice_{ "keepSequenceOnDevice", "Keep Sequence On Device", true };
    // uploads the copy stream has carried so far / elements a frame had to wait for (not prefetched): cpmh_sequence_stream_stats
    bool streamStats(unsigned long long* uploads, unsigned long long* uploadsAtAcquire, double* uploadMs, unsigned long long* bytesPerStep);
private:
    std::shared_ptr<Volume> outVolume_;
    ::cpm_volume_stream* stream_ = nullptr;       // keepSequenceOnDevice == false
    const VolumeSequence* streamedSequence_ = nullptr;
    std::vector<const void*> pinned_;             // elements' RAM blocks registered with the driver for the asynchronous copies
    size_t lastFirst_ = 0;                        // the element the last frame started from, and which way the walk has been going
    int direction_ = +1;
    void dropStream();
};

#ifdef CPM_HOST_EXTRAS
// uniformgridcl/processors/uniformgrid3dsequenceselector.{h,cpp}: VectorElementSelectorProcessor<UniformGrid3DBase>
// (port / property ids of Inviwo's VectorElementSelectorProcessor, assumed: "inport", "outport", "timeStep")
class UniformGrid3DSequenceSelector : public Processor {
public:
    UniformGrid3DSequenceSelector();
    const ProcessorInfo getProcessorInfo() const override { return { "org.inviwo.UniformGrid3DSequenceSelector", "Uniform Grid 3D Sequence Selector", "UniformGrid3D" }; }
    void process() override;
    DataInport<UniformGrid3DVector> inport_{ "inport" };
    DataOutport<UniformGrid3DBase> outport_{ "outport" };
    IntProperty index_{ "selectedSequenceIndex", "Sequence index", 1 };
};
#endif

}  // namespace inviwo
