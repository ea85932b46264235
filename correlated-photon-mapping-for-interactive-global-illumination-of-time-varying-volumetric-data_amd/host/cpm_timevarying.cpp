// cpm_timevarying.cpp -- see cpm_timevarying.h
#include "cpm_timevarying.h"

#include <algorithm>
#include <fstream>
#include <sstream>

namespace inviwo {

namespace {

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::string toLower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}
std::vector<std::string> splitString(const std::string& s, char d) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, d)) out.push_back(item);
    if (out.empty()) out.push_back("");
    return out;
}
std::string parentPath(const std::string& p) {
    size_t i = p.find_last_of('/');
    return i == std::string::npos ? std::string() : p.substr(0, i + 1);
}
std::string replaceExtension(const std::string& p, const std::string& ext) {
    size_t slash = p.find_last_of('/'), dot = p.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return p + "." + ext;
    return p.substr(0, dot + 1) + ext;
}
std::string stem(const std::string& p) {
    size_t slash = p.find_last_of('/');
    std::string name = slash == std::string::npos ? p : p.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    return dot == std::string::npos ? name : name.substr(0, dot);
}
bool fileExists(const std::string& p) { return std::ifstream(p).good(); }

// glm::transpose(m) streamed row by row == the column-major matrix read row-wise
void writeMatrix(std::ostream& ss, const char* key, const mat4& m) {
    ss << key << ":";
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) ss << " " << m[c * 4 + r];
    ss << "\n";
}
mat4 readMatrix(std::stringstream& ss) {  // uniformgrid3dreader.cpp:100-115: mat[i][j] in stream order, then transpose
    mat4 m = identityMatrix();
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float v = 0;
            ss >> v;
            m[j * 4 + i] = v;
        }
    return m;
}

std::shared_ptr<UniformGrid3DBase> makeGrid(const std::string& format) {
    if (format == "Vec2UINT16") return std::make_shared<MinMaxUniformGrid3D>();
    if (format == "FLOAT32") return std::make_shared<ImportanceUniformGrid3D>();
    return nullptr;
}
const char* const kValidFormats =
    "FLOAT16, FLOAT32, FLOAT64, INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64, Vec2FLOAT16, Vec2FLOAT32, "
    "Vec2FLOAT64, Vec2INT8, Vec2INT16, Vec2INT32, Vec2INT64, Vec2UINT8, Vec2UINT16, Vec2UINT32, Vec2UINT64, Vec3FLOAT16, "
    "Vec3FLOAT32, Vec3FLOAT64, Vec3INT8, Vec3INT16, Vec3INT32, Vec3INT64, Vec3UINT8, Vec3UINT16, Vec3UINT32, Vec3UINT64, "
    "Vec4FLOAT16, Vec4FLOAT32, Vec4FLOAT64, Vec4INT8, Vec4INT16, Vec4INT32, Vec4INT64, Vec4UINT8, Vec4UINT16, Vec4UINT32, "
    "Vec4UINT64";
bool isInviwoFormat(const std::string& f) {
    std::stringstream ss(kValidFormats);
    std::string item;
    while (std::getline(ss, item, ',')) if (trim(item) == f) return true;
    return false;
}

}  // namespace

// ---- .u3d ---------------------------------------------------------------------------------------------

void UniformGrid3DWriter::writeData(const UniformGrid3DVector* vectorData, const std::string& filePath) const {
    if (!vectorData || vectorData->size() < 1) throw DataWriterException("Error: Cannot write empty vector");
    const std::string rawPath = replaceExtension(filePath, "raw");
    if (!overwrite_ && (fileExists(filePath) || fileExists(rawPath)))
        throw DataWriterException("Error: File already exists and overwrite is off: " + filePath);

    UniformGrid3DBase* data = vectorData->front().get();
    std::stringstream ss;
    ss.precision(9);
    const size3_t dim = data->getDimensions(), cell = data->getCellDimension();
    ss << "RawFile: " << stem(filePath) << ".raw\n";
    ss << "Resolution: " << dim.x << " " << dim.y << " " << dim.z << " " << vectorData->size() << "\n";
    ss << "Format: " << data->getDataFormatString() << "\n";
    writeMatrix(ss, "ModelMatrix", data->getModelMatrix());
    writeMatrix(ss, "WorldMatrix", data->getWorldMatrix());
    ss << "CellDimensions: " << cell.x << " " << cell.y << " " << cell.z << "\n";

    std::ofstream f(filePath.c_str());
    if (!f.good()) throw DataWriterException("Could not write to file: " + filePath);
    f << ss.str();
    f.close();

    std::ofstream fout(rawPath.c_str(), std::ios::out | std::ios::binary);
    if (!fout.good()) throw DataWriterException("Could not write to raw file: " + rawPath);
    for (auto& element : *vectorData) fout.write((const char*)element->hostData(), (std::streamsize)element->getSizeInBytes());
    fout.close();
}

std::shared_ptr<UniformGrid3DVector> UniformGrid3DReader::readData(const std::string& filePath) {
    const std::string fileDirectory = parentPath(filePath);
    std::ifstream f(filePath.c_str());
    if (!f.good()) throw DataReaderException("Error: Could not open file: " + filePath);
    std::string textLine, rawFile, formatFlag;
    mat4 modelMatrix = identityMatrix(), worldMatrix = identityMatrix();
    size3_t cellDimensions{ 0, 0, 0 };
    size_t resolution[4] = { 0, 0, 0, 0 };
    bool haveFormat = false;

    while (std::getline(f, textLine)) {
        textLine = trim(textLine);
        if (textLine == "" || textLine[0] == '#' || textLine[0] == '/') continue;
        auto parts = splitString(splitString(textLine, '#')[0], ':');
        if (parts.size() != 2) continue;
        const std::string key = toLower(trim(parts[0]));
        const std::string value = trim(parts[1]);
        std::stringstream ss(value);
        if (key == "objectfilename" || key == "rawfile") {
            rawFile = fileDirectory + value;
        } else if (key == "resolution" || key == "dimensions") {
            ss >> resolution[0] >> resolution[1] >> resolution[2] >> resolution[3];
        } else if (key == "format") {
            ss >> formatFlag;
            haveFormat = true;
        } else if (key == "modelmatrix") {
            modelMatrix = readMatrix(ss);
        } else if (key == "worldmatrix") {
            worldMatrix = readMatrix(ss);
        } else if (key == "celldimensions") {
            ss >> cellDimensions.x >> cellDimensions.y >> cellDimensions.z;
        }
    }
    if (resolution[0] == 0 && resolution[1] == 0 && resolution[2] == 0 && resolution[3] == 0)
        throw DataReaderException("Error: Unable to find \"Resolution\" tag in file: " + filePath);
    if (!haveFormat) throw DataReaderException("Error: Unable to find \"Format\" tag in file: " + filePath);
    if (!isInviwoFormat(formatFlag))
        throw DataReaderException("Error: Invalid format string found: " + formatFlag + " in " + filePath +
                                  " \nThe valid formats are:\n" + kValidFormats);
    std::shared_ptr<UniformGrid3DBase> data = makeGrid(formatFlag);
    if (!data) throw DataReaderException("Error: Unsupported data fromat \"Format\" tag in file: " + filePath);
    data->setCellDimension(cellDimensions);
    data->setModelMatrix(modelMatrix);
    data->setWorldMatrix(worldMatrix);
    data->setDimensions(size3_t{ resolution[0], resolution[1], resolution[2] });
    const size_t bytes = data->getSizeInBytes();

    auto dataVector = std::make_shared<UniformGrid3DVector>();
    std::ifstream fin(rawFile.c_str(), std::ios::in | std::ios::binary);
    if (!fin.good()) throw DataReaderException("Error: Unable to read from  file: " + rawFile);
    for (size_t t = 0; t < resolution[3]; ++t) {
        if (t == 0) dataVector->push_back(data);
        else dataVector->push_back(dataVector->front()->clone());
        fin.read((char*)dataVector->back()->hostData(), (std::streamsize)bytes);
        if ((size_t)fin.gcount() != bytes) throw DataReaderException("Error: raw file is too short: " + rawFile);
    }
    return dataVector;
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

DynamicVolumeDifferenceAnalysis::DynamicVolumeDifferenceAnalysis() {
    addPortId("data", true); addPortId("DynamicDataInfo", false);
    addProperty(volumeRegionSize_);
}
void DynamicVolumeDifferenceAnalysis::process() {
    auto& rt = CpmRuntime::get();
    if (!rt.valid() || !inport_.isReady()) return;
    auto data = inport_.getData();
    auto output = std::make_shared<UniformGrid3DVector>();
    const size_t r = (size_t)volumeRegionSize_.get();
    for (size_t timeStep = 0; timeStep < data->size(); ++timeStep) {
        const size_t nextTimeStep = (timeStep + 1) % data->size();
        auto& curVolume = (*data)[timeStep];
        auto& nextVolume = (*data)[nextTimeStep];
        cpm_volume *cur = curVolume->getDeviceRepresentation(), *next = nextVolume->getDeviceRepresentation();
        if (!cur || !next) return;
        const size3_t dim = curVolume->getDimensions();
        auto out = std::make_shared<DynamicVolumeInfoUniformGrid3D>();
        out->setCellDimension(size3_t{ r, r, r });
        out->setModelMatrix(curVolume->getModelMatrix());
        out->setWorldMatrix(curVolume->getWorldMatrix());
        out->setDimensions(size3_t{ (dim.x + r - 1) / r, (dim.y + r - 1) / r, (dim.z + r - 1) / r });
        if (!rt.check(cpm_volume_difference(rt.ctx(), cur, next, (int)r, out->data.device(), rt.stream()), "cpm_volume_difference")) return;
        output->emplace_back(out);
    }
    outport_.setData(output);
}

// ---- players --------------------------------------------------------------------------------------------------

void BufferMixerCL::mix(UniformGrid3DBase& x, UniformGrid3DBase& y, float a, UniformGrid3DBase& out) {
    auto& rt = CpmRuntime::get();
    if (!rt.valid()) return;
    if (!x.hasDeviceData()) x.uploadHostData();  // getRepresentation<BufferCL>() of RAM-only data (e.g. read from .u3d)
    if (!y.hasDeviceData()) y.uploadHostData();
    rt.check(cpm_mix_buffers(rt.ctx(), x.deviceData(), y.deviceData(), a, x.mixElements(), x.mixType(), out.deviceData(), rt.stream()),
             "cpm_mix_buffers");
}

SequenceClock::SequenceClock(const char* perElementId, const char* perElementName, const char* rateId)
    : timePerElement_(perElementId, perElementName, 1.f), frameRate_(rateId, "Frame rate", 10) {
    time_.setMinValue(0.f); time_.setMaxValue(0.f);
    index_.setMinValue(1); index_.setMaxValue(1);
    index_.setReadOnly(true);
    time_.onChange([this]() { updateVolumeIndex(); });
}
void SequenceClock::onSequenceTimerEvent() {  // uniformgrid3dplayerprocessor.cpp:117-127
    float time = time_.get();
    time = time + static_cast<float>(1000 / frameRate_.get()) / 1000.f;
    if (time > time_.getMaxValue()) time -= time_.getMaxValue();  // wrap around
    time_.set(time);
    updateVolumeIndex();
}
void SequenceClock::updateVolumeIndex() {  // :130-138
    float integerTime;
    std::modf(time_.get() / timePerElement_.get(), &integerTime);
    auto timeStep = static_cast<size_t>(integerTime) % (size_t)index_.getMaxValue();
    if ((int)timeStep != index_.get() - 1) index_.set(static_cast<int>(timeStep + 1));
}
void SequenceClock::onTimeStepChange(size_t nElements) {  // :140-152
    time_.setMaxValue(time_.getMinValue() + static_cast<float>(nElements - 1) * timePerElement_.get());
    if (time_.get() > time_.getMaxValue()) time_.set(time_.getMinValue());
    index_.setMaxValue(static_cast<int>(nElements));
    if (index_.get() > index_.getMaxValue()) index_.set(index_.getMinValue());
}

UniformGrid3DPlayerProcessor::UniformGrid3DPlayerProcessor() {
    addPortId("Sequence", true); addPortId("InterpolatedData", false);
    addProperty(clock_.time_); addProperty(clock_.index_); addProperty(clock_.timePerElement_);
    addProperty(clock_.frameRate_); addProperty(clock_.playSequence_);
    inport_.onConnect([this]() { if (inport_.hasData()) clock_.onTimeStepChange(inport_.getData()->size()); });
    clock_.timePerElement_.onChange([this]() { if (inport_.hasData()) clock_.onTimeStepChange(inport_.getData()->size()); });
}
void UniformGrid3DPlayerProcessor::process() {
    if (!inport_.isReady()) return;
    auto elements = inport_.getData();
    if (elements->empty()) return;
    if ((size_t)clock_.index_.getMaxValue() != elements->size()) clock_.onTimeStepChange(elements->size());  // inport_.onChange
    const float t = clock_.weight();
    const size_t timeStep = (size_t)(clock_.index_.get() - 1);
    const size_t nextTimeStep = (timeStep + 1) % elements->size();
    if (elements->size() > 1) {
        std::swap(outData_, outDataPingPong_);
        auto input0 = elements->at(timeStep);
        auto input1 = elements->at(nextTimeStep);
        const size3_t d0 = input0->getDimensions();
        if (!outData_ || outData_->getDimensions().x != d0.x || outData_->getDimensions().y != d0.y || outData_->getDimensions().z != d0.z ||
            std::string(outData_->getDataFormatString()) != input0->getDataFormatString()) {
            outData_ = input0->clone();
            outData_->setModelMatrix(input0->getModelMatrix());
            outData_->setWorldMatrix(input0->getWorldMatrix());
        }
        bufferMixer_.mix(*input0, *input1, t, *outData_);
        outport_.setData(outData_);
    } else {
        outport_.setData(elements->at(timeStep));
    }
}

VolumeSequencePlayer::VolumeSequencePlayer() {
    addPortId("volumeSequence", true); addPortId("InterpolatedVolume", false);
    addProperty(clock_.time_); addProperty(clock_.index_); addProperty(clock_.timePerElement_);
    addProperty(clock_.frameRate_); addProperty(clock_.playSequence_);
    inport_.onConnect([this]() { if (inport_.hasData()) clock_.onTimeStepChange(inport_.getData()->size()); });
    clock_.timePerElement_.onChange([this]() { if (inport_.hasData()) clock_.onTimeStepChange(inport_.getData()->size()); });
}
void VolumeSequencePlayer::process() {
    auto& rt = CpmRuntime::get();
    if (!inport_.isReady()) return;
    auto volumes = inport_.getData();
    if (volumes->empty()) return;
    if ((size_t)clock_.index_.getMaxValue() != volumes->size()) clock_.onTimeStepChange(volumes->size());
    const float t = clock_.weight();
    const size_t timeStep = (size_t)(clock_.index_.get() - 1);
    const size_t nextTimeStep = (timeStep + 1) % volumes->size();
    if (volumes->size() > 1) {
        auto inputVol0 = volumes->at(timeStep);
        auto inputVol1 = volumes->at(nextTimeStep);
        const size3_t d0 = inputVol0->getDimensions();
        if (!outVolume_ || outVolume_->getDimensions().x != d0.x || outVolume_->getDimensions().y != d0.y ||
            outVolume_->getDimensions().z != d0.z || outVolume_->dtype() != inputVol0->dtype()) {
            outVolume_ = std::make_shared<Volume>(d0, inputVol0->dtype());  // device storage only
            outVolume_->setModelMatrix(inputVol0->getModelMatrix());
            outVolume_->setWorldMatrix(inputVol0->getWorldMatrix());
        }
        cpm_volume *v0 = inputVol0->getDeviceRepresentation(), *v1 = inputVol1->getDeviceRepresentation();
        cpm_volume* out = outVolume_->getDeviceRepresentation();
        if (!v0 || !v1 || !out) return;
        if (!rt.check(cpm_volume_mix(rt.ctx(), v0, v1, t, out, rt.stream()), "cpm_volume_mix")) return;
        outport_.setData(outVolume_);
    } else {
        outport_.setData(volumes->at(timeStep));
    }
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
