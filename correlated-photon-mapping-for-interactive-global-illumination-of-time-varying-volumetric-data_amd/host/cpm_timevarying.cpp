// cpm_timevarying.cpp -- see cpm_timevarying.h
#include "cpm_timevarying.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>

namespace inviwo {

namespace {

// ---- the .u3d format, as this build states it -----------------------------------------------------------------------------
// A sequence of uniform grids is a text header <name>.u3d beside <name>.raw, the elements' data back to back.  Header: one
// "Key: value" per line, keys case-blind; blank lines and lines opening with '#' or '/' are skipped, text behind a '#' is
// dropped, and a line that does not split into exactly one key and one value at ':' is ignored.  Keys:
//   RawFile | ObjectFileName   the data file, relative to the header's directory
//   Resolution | Dimensions    nx ny nz count
//   Format                     element type by Inviwo's format name (Vec2UINT16 = min/max grid, FLOAT32 = float grid)
//   ModelMatrix, WorldMatrix   sixteen numbers, rows first (the matrices are held columns first)
//   CellDimensions             voxels per cell, per axis
// (uniformgridcl/uniformgrid3dwriter.cpp:47-102 writes it, uniformgrid3dreader.cpp:59-183 reads it.)

std::string stripped(const std::string& text) {
    const char* blanks = " \t\r\n";
    const size_t first = text.find_first_not_of(blanks);
    if (first == std::string::npos) return {};
    return text.substr(first, text.find_last_not_of(blanks) - first + 1);
}

struct PathParts {
    std::string folder, name;  // folder keeps its trailing '/', name has no extension
    explicit PathParts(const std::string& path) {
        const size_t cut = path.find_last_of('/');
        folder = cut == std::string::npos ? std::string() : path.substr(0, cut + 1);
        name = path.substr(folder.size());
        const size_t dot = name.find_last_of('.');
        if (dot != std::string::npos) name.erase(dot);
    }
    std::string sibling(const char* extension) const { return folder + name + "." + extension; }
};

bool readable(const std::string& path) { return std::ifstream(path).good(); }

struct U3dHeader {
    std::string dataFile, format;
    size_t extent[4] = { 0, 0, 0, 0 };  // nx ny nz count
    mat4 model = identityMatrix(), world = identityMatrix();
    size3_t cell{ 0, 0, 0 };
    bool formatSeen = false;

    static void rowsFirst(std::istream& in, mat4& m) {
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col) {
                float value = 0.f;
                in >> value;
                m[(size_t)col * 4 + row] = value;
            }
    }
    static void rowsFirst(std::ostream& os, const char* key, const mat4& m) {
        os << key << ":";
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col) os << " " << m[(size_t)col * 4 + row];
        os << "\n";
    }

    // one header line, its key already lower-cased
    void take(const std::string& key, const std::string& value, const std::string& folder) {
        enum Field { DataFile, Extent, Format, Model, World, Cell };
        static const std::map<std::string, Field> fields = { { "rawfile", DataFile }, { "objectfilename", DataFile }, { "resolution", Extent },
                                                              { "dimensions", Extent }, { "format", Format }, { "modelmatrix", Model },
                                                              { "worldmatrix", World }, { "celldimensions", Cell } };
        const auto known = fields.find(key);
        if (known == fields.end()) return;
        std::istringstream in(value);
        switch (known->second) {
            case DataFile: dataFile = folder + value; break;
            case Extent: for (size_t& e : extent) in >> e; break;
            case Format: in >> format; formatSeen = true; break;
            case Model: rowsFirst(in, model); break;
            case World: rowsFirst(in, world); break;
            case Cell: in >> cell.x >> cell.y >> cell.z; break;
        }
    }

    void parse(std::istream& text, const std::string& folder) {
        for (std::string line; std::getline(text, line);) {
            line = stripped(line);
            const char lead = line.empty() ? '#' : line.front();
            if (lead == '#' || lead == '/') continue;
            line = line.substr(0, line.find('#'));
            std::vector<std::string> fields;  // the pieces between colons; an empty piece at the very end does not count
            for (size_t from = 0; from <= line.size();) {
                const size_t colon = std::min(line.find(':', from), line.size());
                if (colon < line.size() || colon > from) fields.push_back(line.substr(from, colon - from));
                from = colon + 1;
            }
            if (fields.size() != 2) continue;
            std::string key = stripped(fields[0]);
            for (char& ch : key) ch = (char)std::tolower((unsigned char)ch);
            take(key, stripped(fields[1]), folder);
        }
    }
};

// Inviwo's data-format names: scalar or VecN of these element types
bool knownFormatName(std::string name) {
    if (name.compare(0, 3, "Vec") == 0 && name.size() > 3 && name[3] >= '2' && name[3] <= '4') name.erase(0, 4);
    for (const char* element : { "FLOAT16", "FLOAT32", "FLOAT64", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64" })
        if (name == element) return true;
    return false;
}

std::shared_ptr<UniformGrid3DBase> gridOfFormat(const std::string& format) {
    if (format == "Vec2UINT16") return std::make_shared<MinMaxUniformGrid3D>();
    if (format == "FLOAT32") return std::make_shared<ImportanceUniformGrid3D>();
    return nullptr;
}

}  // namespace

// ---- .u3d ---------------------------------------------------------------------------------------------

void UniformGrid3DWriter::writeData(const UniformGrid3DVector* vectorData, const std::string& filePath) const {
    if (!vectorData || vectorData->empty()) throw DataWriterException("UniformGrid3DWriter: nothing to write (empty sequence)");
    const PathParts where(filePath);
    const std::string dataPath = where.sibling("raw");
    if (!overwrite_ && (readable(filePath) || readable(dataPath)))
        throw DataWriterException("UniformGrid3DWriter: " + filePath + " (or its .raw) exists and overwriting is switched off");

    const UniformGrid3DBase& first = *vectorData->front();
    const size3_t n = first.getDimensions(), c = first.getCellDimension();
    std::ostringstream head;
    head.precision(9);
    head << "RawFile: " << where.name << ".raw\n"
         << "Resolution: " << n.x << " " << n.y << " " << n.z << " " << vectorData->size() << "\n"
         << "Format: " << first.getDataFormatString() << "\n";
    U3dHeader::rowsFirst(head, "ModelMatrix", first.getModelMatrix());
    U3dHeader::rowsFirst(head, "WorldMatrix", first.getWorldMatrix());
    head << "CellDimensions: " << c.x << " " << c.y << " " << c.z << "\n";

    std::ofstream text(filePath);
    if (!(text << head.str())) throw DataWriterException("UniformGrid3DWriter: cannot write " + filePath);
    text.close();
    std::ofstream raw(dataPath, std::ios::binary);
    for (const auto& grid : *vectorData) raw.write(static_cast<const char*>(grid->hostData()), (std::streamsize)grid->getSizeInBytes());
    if (!raw.good()) throw DataWriterException("UniformGrid3DWriter: cannot write " + dataPath);
}

std::shared_ptr<UniformGrid3DVector> UniformGrid3DReader::readData(const std::string& filePath) {
    std::ifstream text(filePath);
    if (!text.good()) throw DataReaderException("UniformGrid3DReader: cannot open " + filePath);
    U3dHeader head;
    head.parse(text, PathParts(filePath).folder);
    if (head.extent[0] + head.extent[1] + head.extent[2] + head.extent[3] == 0)
        throw DataReaderException("UniformGrid3DReader: no \"Resolution\" line in " + filePath);
    if (!head.formatSeen) throw DataReaderException("UniformGrid3DReader: no \"Format\" line in " + filePath);
    if (!knownFormatName(head.format)) throw DataReaderException("UniformGrid3DReader: \"" + head.format + "\" in " + filePath + " is not a data format name");
    std::shared_ptr<UniformGrid3DBase> grid = gridOfFormat(head.format);
    if (!grid) throw DataReaderException("UniformGrid3DReader: grids of format " + head.format + " are not supported (" + filePath + ")");
    grid->setCellDimension(head.cell);
    grid->setModelMatrix(head.model);
    grid->setWorldMatrix(head.world);
    grid->setDimensions(size3_t{ head.extent[0], head.extent[1], head.extent[2] });

    std::ifstream raw(head.dataFile, std::ios::binary);
    if (!raw.good()) throw DataReaderException("UniformGrid3DReader: cannot open the data file " + head.dataFile);
    auto sequence = std::make_shared<UniformGrid3DVector>();
    const std::streamsize bytesPerGrid = (std::streamsize)grid->getSizeInBytes();
    for (size_t element = 0; element < head.extent[3]; ++element) {
        sequence->push_back(element == 0 ? grid : grid->clone());
        if (!raw.read(static_cast<char*>(sequence->back()->hostData()), bytesPerGrid))
            throw DataReaderException("UniformGrid3DReader: " + head.dataFile + " ends before element " + std::to_string(element) + " is complete");
    }
    return sequence;
}

#ifdef CPM_HOST_EXTRAS
UniformGrid3DVectorSource::UniformGrid3DVectorSource() { addPortId("data", false); addProperty(filePath); }
void UniformGrid3DVectorSource::process() {
    if (filePath.get().empty() || filePath.get() == loaded_) return;
    try {
        port_.setData(UniformGrid3DReader().readData(filePath.get()));
        loaded_ = filePath.get();
    } catch (const DataReaderException& e) {
        LogError(e.what());
    }
}
UniformGrid3DExport::UniformGrid3DExport() { addPortId("data", true); addProperty(file_); addProperty(overwrite_); }
void UniformGrid3DExport::exportData() {
    if (!port_.isReady()) return;
    try {
        UniformGrid3DWriter w;
        w.setOverwrite(overwrite_.get());
        w.writeData(port_.getData().get(), file_.get());
    } catch (const DataWriterException& e) {
        LogError(e.what());
    }
}
#endif

// ---- difference analysis ----------------------------------------------------------------------------------
// Per time step t: the mean |v(t+1) - v(t)| of every brick of `region` voxels a side, the last step against the first
// (dynamicvolumedifferenceanalysis.cpp:60-104 loops over voxels on the CPU; here one launch per pair).

DynamicVolumeDifferenceAnalysis::DynamicVolumeDifferenceAnalysis() {
    addPortId("data", true); addPortId("DynamicDataInfo", false);
    addProperty(volumeRegionSize_);
}
void DynamicVolumeDifferenceAnalysis::process() {
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || !inport_.isReady()) return;
    const auto sequence = inport_.getData();
    const size_t steps = sequence->size(), side = (size_t)volumeRegionSize_.get();
    auto perStep = std::make_shared<UniformGrid3DVector>();
    auto bricksAlong = [side](size_t voxels) { return (voxels + side - 1) / side; };
    for (size_t t = 0; t < steps; ++t) {
        const auto& before = (*sequence)[t];
        const auto& after = (*sequence)[(t + 1) % steps];
        cpm_volume *dBefore = before->getDeviceRepresentation(), *dAfter = after->getDeviceRepresentation();
        if (!dBefore || !dAfter) return;
        auto info = std::make_shared<DynamicVolumeInfoUniformGrid3D>();
        info->setCellDimension(size3_t{ side, side, side });
        info->setModelMatrix(before->getModelMatrix());
        info->setWorldMatrix(before->getWorldMatrix());
        const size3_t voxels = before->getDimensions();
        info->setDimensions(size3_t{ bricksAlong(voxels.x), bricksAlong(voxels.y), bricksAlong(voxels.z) });
        if (!rt.check(cpm_volume_difference(rt.ctx(), dBefore, dAfter, (int)side, info->data.device(), rt.stream()), "cpm_volume_difference")) return;
        perStep->push_back(info);
    }
    outport_.setData(perStep);
}

// ---- players --------------------------------------------------------------------------------------------------
// Both players show element `index` blended with its successor (the last with the first) by the fraction of
// time / timePerElement; a sequence of one element is passed through.  The clock below is that rule (the reference keeps a copy of
// it in each player: uniformgrid3dplayerprocessor.cpp:117-152, volumesequenceplayer.cpp:142-180); the timer that ticks it belongs to
// the caller.

void BufferMixerCL::mix(UniformGrid3DBase& x, UniformGrid3DBase& y, float a, UniformGrid3DBase& out) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    for (UniformGrid3DBase* operand : { &x, &y })
        if (!operand->hasDeviceData()) operand->uploadHostData();  // RAM-only data (read from a .u3d) gets its device copy here
    rt.check(cpm_mix_buffers(rt.ctx(), x.deviceData(), y.deviceData(), a, x.mixElements(), x.mixType(), out.deviceData(), rt.stream()),
             "cpm_mix_buffers");
}

SequenceClock::SequenceClock(const char* perElementId, const char* perElementName, const char* rateId)
    : timePerElement_(perElementId, perElementName, 1.f), frameRate_(rateId, "Frame rate", 10) {
    time_.setMinValue(0.f); time_.setMaxValue(0.f);
    index_.setMinValue(1); index_.setMaxValue(1);
    index_.setReadOnly(true);
    time_.onChange(std::bind(&SequenceClock::updateVolumeIndex, this));
}
// a tick advances the time by one frame period (whole milliseconds, as the reference's timer interval) and wraps at the end
void SequenceClock::onSequenceTimerEvent() {
    const int wholeMilliseconds = 1000 / frameRate_.get();
    const float period = (float)wholeMilliseconds / 1000.f;
    float now = time_.get() + period;
    const float end = time_.getMaxValue();
    if (now > end) now -= end;
    time_.set(now);
    updateVolumeIndex();
}
// index = 1 + (whole elements elapsed) mod (elements)
void SequenceClock::updateVolumeIndex() {
    float whole = 0.f;
    std::modf(time_.get() / timePerElement_.get(), &whole);
    const int shown = 1 + (int)(static_cast<size_t>(whole) % (size_t)index_.getMaxValue());
    if (shown != index_.get()) index_.set(shown);
}
// a sequence of n elements spans (n - 1) element times; time and index are pulled back into range
void SequenceClock::onTimeStepChange(size_t nElements) {
    const float start = time_.getMinValue();
    time_.setMaxValue(start + static_cast<float>(nElements - 1) * timePerElement_.get());
    if (time_.get() > time_.getMaxValue()) time_.set(start);
    index_.setMaxValue(static_cast<int>(nElements));
    if (index_.get() > index_.getMaxValue()) index_.set(index_.getMinValue());
}

namespace {
// what a player's process() has to blend: elements `first` and `second` of a sequence of `count`, at `fraction` between them
struct BlendStep {
    size_t first, second;
    float fraction;
    BlendStep(SequenceClock& clock, size_t count) {
        if ((size_t)clock.index_.getMaxValue() != count) clock.onTimeStepChange(count);  // the sequence on the inport changed length
        fraction = clock.weight();
        first = (size_t)(clock.index_.get() - 1);
        second = (first + 1) % count;
    }
};
bool sameExtent(size3_t a, size3_t b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
}  // namespace

UniformGrid3DPlayerProcessor::UniformGrid3DPlayerProcessor() {
    addPortId("Sequence", true); addPortId("InterpolatedData", false);
    addProperty(clock_.time_); addProperty(clock_.index_); addProperty(clock_.timePerElement_);
    addProperty(clock_.frameRate_); addProperty(clock_.playSequence_);
    auto resize = [this]() { if (inport_.hasData()) clock_.onTimeStepChange(inport_.getData()->size()); };
    inport_.onConnect(resize);
    clock_.timePerElement_.onChange(resize);
}
void UniformGrid3DPlayerProcessor::process() {
    if (!inport_.isReady()) return;
    const auto sequence = inport_.getData();
    if (sequence->empty()) return;
    const BlendStep step(clock_, sequence->size());
    const auto& from = sequence->at(step.first);
    if (sequence->size() == 1) { outport_.setData(from); return; }
    const auto& towards = sequence->at(step.second);
    std::swap(outData_, outDataPingPong_);  // the consumer may still hold the grid handed out last time
    const bool fits = outData_ && sameExtent(outData_->getDimensions(), from->getDimensions()) &&
                      std::strcmp(outData_->getDataFormatString(), from->getDataFormatString()) == 0;
    if (!fits) outData_ = from->clone();  // shape, matrices and cell size of the sequence's elements
    bufferMixer_.mix(*from, *towards, step.fraction, *outData_);
    outport_.setData(outData_);
}

VolumeSequencePlayer::VolumeSequencePlayer() {
    addPortId("volumeSequence", true); addPortId("InterpolatedVolume", false);
    addProperty(clock_.time_); addProperty(clock_.index_); addProperty(clock_.timePerElement_);
    addProperty(clock_.frameRate_); addProperty(clock_.playSequence_);
    addProperty(keepSequenceOnDevice_);
    auto resize = [this]() { if (inport_.hasData()) clock_.onTimeStepChange(inport_.getData()->size()); };
    inport_.onConnect(resize);
    clock_.timePerElement_.onChange(resize);
}
void VolumeSequencePlayer::process() {
    auto& rt = CpmRuntime::get();
    if (!inport_.isReady()) return;
    const auto sequence = inport_.getData();
    if (sequence->empty()) return;
    const BlendStep step(clock_, sequence->size());
    const auto& from = sequence->at(step.first);
    if (sequence->size() == 1) { outport_.setData(from); return; }
    const auto& towards = sequence->at(step.second);
    if (!outVolume_ || !sameExtent(outVolume_->getDimensions(), from->getDimensions()) || outVolume_->dtype() != from->dtype()) {
        outVolume_ = std::make_shared<Volume>(from->getDimensions(), from->dtype());  // device storage only
        outVolume_->setModelMatrix(from->getModelMatrix());
        outVolume_->setWorldMatrix(from->getWorldMatrix());
    }
    cpm_volume *a = nullptr, *b = nullptr, *blended = outVolume_->getDeviceRepresentation();
    if (keepSequenceOnDevice_.get()) {
        if (stream_) dropStream();
        a = from->getDeviceRepresentation(); b = towards->getDeviceRepresentation();
    } else {
        // the elements stay in host memory (ref volumesequenceplayer.cpp:94-124 hands them to OpenGL, which uploads what is not resident): the two
        // this frame blends are acquired from the ring, the one after them is put on the copy stream now -- it crosses PCIe behind this frame's work
        const size_t bytes = from->getDimensions().x * from->getDimensions().y * from->getDimensions().z * from->elementSize();
        if (stream_ && streamedSequence_ != sequence.get()) dropStream();
        if (!stream_) {
            cpm_volume_desc d;
            const int32_t dims[3] = { (int32_t)from->getDimensions().x, (int32_t)from->getDimensions().y, (int32_t)from->getDimensions().z };
            cpm_volume_desc_default(&d, dims, from->dtype());
            if (!rt.check(cpm_volume_stream_create(rt.ctx(), &d, 3, &stream_), "cpm_volume_stream_create")) { stream_ = nullptr; return; }
            streamedSequence_ = sequence.get();
            for (const auto& v : *sequence)   // page-lock the elements' RAM once: a copy from pageable memory is staged and holds this thread
                if (v->ramBytes.size() == bytes && hipHostRegister(const_cast<uint8_t*>(v->ramBytes.data()), bytes, hipHostRegisterDefault) == hipSuccess)
                    pinned_.push_back(v->ramBytes.data());
            (void)hipGetLastError();
        }
        // (the element the walk reaches next: one further along when the clock moved on by one element since the last frame, one back when it
        // moved back by one, as before otherwise)
        const size_t count = sequence->size();
        if (step.first == (lastFirst_ + 1) % count) direction_ = +1;
        else if (step.first == (lastFirst_ + count - 1) % count) direction_ = -1;
        lastFirst_ = step.first;
        const size_t next = direction_ > 0 ? (step.second + 1) % count : (step.first + count - 1) % count;
        for (size_t e : { step.first, step.second, next }) {
            const auto& v = sequence->at(e);
            if (v->ramBytes.size() != bytes) { rt.check(CPM_ERR_INVALID_ARGUMENT, "VolumeSequencePlayer: an element without RAM data (or of another size) cannot be streamed"); return; }
        }
        if (!rt.check(cpm_volume_stream_acquire(rt.ctx(), stream_, step.first, from->ramBytes.data(), rt.stream(), &a), "cpm_volume_stream_acquire") ||
            !rt.check(cpm_volume_stream_acquire(rt.ctx(), stream_, step.second, towards->ramBytes.data(), rt.stream(), &b), "cpm_volume_stream_acquire"))
            return;
        if (next != step.first && next != step.second)
            rt.check(cpm_volume_stream_prefetch(rt.ctx(), stream_, next, sequence->at(next)->ramBytes.data(), rt.stream()), "cpm_volume_stream_prefetch");
    }
    if (!a || !b || !blended) return;
    if (rt.check(cpm_volume_mix(rt.ctx(), a, b, step.fraction, blended, rt.stream()), "cpm_volume_mix")) outport_.setData(outVolume_);
}
void VolumeSequencePlayer::dropStream() {
    auto& rt = CpmRuntime::get();
    if (stream_) cpm_volume_stream_destroy(rt.ctx(), stream_);   // (waits for its copy stream)
    stream_ = nullptr; streamedSequence_ = nullptr;
    for (const void* p : pinned_) (void)hipHostUnregister(const_cast<void*>(p));
    (void)hipGetLastError();
    pinned_.clear();
}
VolumeSequencePlayer::~VolumeSequencePlayer() { dropStream(); }
bool VolumeSequencePlayer::streamStats(unsigned long long* uploads, unsigned long long* uploadsAtAcquire, double* uploadMs, unsigned long long* bytesPerStep) {
    if (!stream_) return false;
    cpm_volume_stream_info i;
    if (cpm_volume_stream_stats(CpmRuntime::get().ctx(), stream_, &i) != CPM_OK) return false;
    if (uploads) *uploads = i.uploads_timed;
    if (uploadsAtAcquire) *uploadsAtAcquire = i.uploads_at_acquire;
    if (uploadMs) *uploadMs = i.upload_ms_total;
    if (bytesPerStep) *bytesPerStep = i.bytes_per_step;
    return true;
}

#ifdef CPM_HOST_EXTRAS
UniformGrid3DSequenceSelector::UniformGrid3DSequenceSelector() {
    addPortId("inport", true); addPortId("outport", false);
    addProperty(index_);
}
void UniformGrid3DSequenceSelector::process() {
    if (!inport_.isReady()) return;
    auto v = inport_.getData();
    if (v->empty()) return;
    size_t i = (size_t)std::max(1, index_.get()) - 1;
    outport_.setData(v->at(std::min(i, v->size() - 1)));
}
#endif

}  // namespace inviwo
