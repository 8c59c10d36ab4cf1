/*
 * hrbf_io.h — header-only C++ readers for the inputs the reference's caller feeds HRBFFusion::processFrame with, so
 * that a C++ main can drive include/HRBFFusion.h without OpenCV / Pangolin (zlib is the only dependency, -lz):
 *
 *   ParameterFile / GlobalStateParam   Core/src/Utils/parameterFile.h:29-75,186-230, Utils/GlobalStateParams.h:12-63
 *                                      ("key = value;" lines, comment markers // # ; outside quotes, later lines win,
 *                                      std::stoi / std::stof / toBool conversions of Utils/stringUtilConvert.h)
 *   CameraFile                         the OpenCV FileStorage YAML read by GUI/src/HRBF_fusion.cpp:44-54 and
 *                                      Core/src/HRBFFusion.cpp:682-781 (Camera.fx/fy/cx/cy/width/height, DepthMapFactor
 *                                      with the "|factor| < 1e-5 -> 1, else 1/factor" rule, Camera.RGB)
 *   AssociationReader                  sensorType 3: "timestamp depthfile timestamp rgbfile" per line
 *                                      (Core/src/HRBFFusion.cpp:212-270, GUI/src/Tools/RawImageReader.cpp:3-95)
 *   decodePng                          8-bit RGB / RGBA / grey and 16-bit grey PNGs (what TUM / ICL-NUIM ship);
 *                                      non-interlaced; the five PNG filters; inflate by zlib
 *   KlgReader                          sensorType 2, the .klg raw log (GUI/src/Tools/RawLogReader.cpp:3-140): int32 frame
 *                                      count; per frame int64 timestamp, int32 depthSize, int32 imageSize, depth raw or
 *                                      zlib, colour raw or JPEG (hrbf_jpeg.h: libjpeg's default decode written out, bit for bit)
 *
 * The Python twins live in hrbffusion3d_amd/config.py and hrbffusion3d_amd/io.py; tests/test_cpp_io.py checks the two
 * against each other.  Nothing here touches the GPU.
 */
#ifndef HRBF_MI355_IO_H_
#define HRBF_MI355_IO_H_

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "hrbf_jpeg.h"

namespace hrbf_mi355 {

// ------------------------------------------------------------------------------------------------ ParameterFile
class ParameterFile {
public:
    explicit ParameterFile(const std::string &filename)
    {
        std::ifstream f(filename.c_str());
        if (!f.is_open()) throw std::runtime_error("cannot open parameter file " + filename);
        std::string line;
        while (std::getline(f, line)) {
            removeComments(line);
            strip(line);
            if (line.length() <= 1) continue;
            const size_t sep = line.find('=');
            if (sep == std::string::npos) continue;     // "No seperator found in line"
            std::string name = line.substr(0, sep), value = line.substr(sep + 1);
            strip(name); strip(value);
            if (name.empty()) continue;
            values_[name] = value;                       // later lines override earlier ones
        }
    }
    bool has(const std::string &name) const { return values_.count(name) != 0; }
    std::string getString(const std::string &name, const std::string &def = "") const
    {
        auto it = values_.find(name);
        return it == values_.end() ? def : it->second;
    }
    /* util::convertTo<bool>: everything but "false" / "False" / "0" is true */
    bool getBool(const std::string &name, bool def) const
    {
        auto it = values_.find(name);
        if (it == values_.end()) return def;
        return !(it->second == "false" || it->second == "False" || it->second == "0");
    }
    /* std::stoi / std::stof: the longest numeric prefix ("6.0" read as an int is 6) */
    int getInt(const std::string &name, int def) const
    {
        auto it = values_.find(name);
        if (it == values_.end()) return def;
        char *end = nullptr;
        const long v = std::strtol(it->second.c_str(), &end, 10);
        return end == it->second.c_str() ? def : (int)v;
    }
    float getFloat(const std::string &name, float def) const
    {
        auto it = values_.find(name);
        if (it == values_.end()) return def;
        char *end = nullptr;
        const float v = std::strtof(it->second.c_str(), &end);
        return end == it->second.c_str() ? def : v;
    }
    const std::map<std::string, std::string> &values() const { return values_; }

private:
    static void strip(std::string &s)
    {
        static const char *chars = " \t\";\r\n";      // ParameterFile::removeSpecialCharacters
        const size_t a = s.find_first_not_of(chars);
        if (a == std::string::npos) { s.clear(); return; }
        const size_t b = s.find_last_not_of(chars);
        s = s.substr(a, b - a + 1);
    }
    static void removeComments(std::string &s)
    {
        std::vector<size_t> q;
        for (size_t i = 0; i < s.size(); ++i) if (s[i] == '"') q.push_back(i);
        const char *markers[] = {"//", "#", ";"};
        for (const char *m : markers) {
            const size_t at = s.find(m);
            if (at == std::string::npos) continue;
            bool inside = false;
            for (size_t j = 0; j + 1 < q.size(); j += 2) if (at > q[j] && at < q[j + 1]) { inside = true; break; }
            if (!inside) s = s.substr(0, at);
        }
    }
    std::map<std::string, std::string> values_;
};

/* the GlobalStateParam fields the dense front-end reads, with the defaults of GUI/GlobalStateParam.txt */
struct GlobalState {
    std::string currentWorkingDirectory, klgFileName, AssociationFile, parameterFileCvFormat;
    int sensorType = 3;
    bool optimizationUseLocalBA = false, optimizationUseGlobalBA = false;
    bool preprocessingUsebilateralFilter = true;
    float preprocessingInitRadiusMultiplier = 4.0f, preprocessingCurvEstimationWindow = 3.0f, preprocessingCurvValidThreshold = 300.0f;
    float preprocessingNormalEstimationPCA = 1.0f;
    int preprocessingUseConfEval = 0;
    float preprocessingConfEvalEpsilon = 1000.0f;
    bool registrationPreAlignSO3 = true;
    float registrationJointICPWeight = 10.0f;
    bool registrationICPUseSparseICP = false, registrationICPUseCoorespondenceSearch = false;
    int registrationICPNeighborSearchRadius = 2;
    bool registrationICPUseWeightedICP = true;
    float registrationICPCurvWeightImpactControl = 10.0f;
    bool registrationColorUseRGBGrad = false;
    float preictionWindowMultiplier = 3.0f;
    int preictionMinNeighbors = 6, preictionMaxNeighbors = 10;
    float preictionConfThreshold = 3.0f;
    float fusionCleanWindowMultiplier = 2.0f;
    float globalConfidenceThreshold = 5.0f, globalDenseEnoughThresh = 0.75f, globalDepthCutoff = 3.5f;
    bool globalInputICLNUIMDataset = false, globalInputLoadTrajectory = false;
    std::string globalInputTrajectoryFormat = "TUM", globalInputTrajectoryFile;
    float globalOutputSavePointCloudConfThreshold = 0.0f;
    int globalStartFrame = 0, globalEndFrame = -1, globalFrameToSkip = 0;
    /* read by the reference's caller only (GUI/src/HRBF_fusion.cpp:87-93), carried so that such a caller compiles */
    float registrationICPErrorThreshold = 5e-05f, registrationICPCovarianceThreshold = 1e-05f, registrationColorPhotoThreshold = 115.0f;
    bool globalOutputSaveTrjectoryFile = false;
    std::string globalOutputSaveTrjectoryFileType = "TUM";

    static GlobalState fromFile(const std::string &filename) { return fromParameterFile(ParameterFile(filename)); }
    static GlobalState fromParameterFile(const ParameterFile &pf)
    {
        GlobalState g;
#define HRBF_S(n) g.n = pf.getString(#n, g.n)
#define HRBF_B(n) g.n = pf.getBool(#n, g.n)
#define HRBF_I(n) g.n = pf.getInt(#n, g.n)
#define HRBF_F(n) g.n = pf.getFloat(#n, g.n)
        HRBF_S(currentWorkingDirectory); HRBF_S(klgFileName); HRBF_S(AssociationFile); HRBF_S(parameterFileCvFormat);
        HRBF_I(sensorType); HRBF_B(optimizationUseLocalBA); HRBF_B(optimizationUseGlobalBA);
        HRBF_B(preprocessingUsebilateralFilter); HRBF_F(preprocessingInitRadiusMultiplier);
        HRBF_F(preprocessingCurvEstimationWindow); HRBF_F(preprocessingCurvValidThreshold);
        HRBF_F(preprocessingNormalEstimationPCA); HRBF_I(preprocessingUseConfEval); HRBF_F(preprocessingConfEvalEpsilon);
        HRBF_B(registrationPreAlignSO3); HRBF_F(registrationJointICPWeight); HRBF_B(registrationICPUseSparseICP);
        HRBF_B(registrationICPUseCoorespondenceSearch); HRBF_I(registrationICPNeighborSearchRadius);
        HRBF_B(registrationICPUseWeightedICP); HRBF_F(registrationICPCurvWeightImpactControl);
        HRBF_B(registrationColorUseRGBGrad); HRBF_F(preictionWindowMultiplier); HRBF_I(preictionMinNeighbors);
        HRBF_I(preictionMaxNeighbors); HRBF_F(preictionConfThreshold); HRBF_F(fusionCleanWindowMultiplier);
        HRBF_F(globalConfidenceThreshold); HRBF_F(globalDenseEnoughThresh); HRBF_F(globalDepthCutoff);
        HRBF_B(globalInputICLNUIMDataset); HRBF_B(globalInputLoadTrajectory);
        HRBF_F(globalOutputSavePointCloudConfThreshold); HRBF_I(globalStartFrame); HRBF_I(globalEndFrame);
        HRBF_I(globalFrameToSkip);
        HRBF_S(globalInputTrajectoryFormat); HRBF_S(globalInputTrajectoryFile);
        HRBF_F(registrationICPErrorThreshold); HRBF_F(registrationICPCovarianceThreshold); HRBF_F(registrationColorPhotoThreshold);
        HRBF_B(globalOutputSaveTrjectoryFile); HRBF_S(globalOutputSaveTrjectoryFileType);
#undef HRBF_S
#undef HRBF_B
#undef HRBF_I
#undef HRBF_F
        return g;
    }
};

// ------------------------------------------------------------------------------------------------ camera YAML
struct CameraFile {
    int width = 0, height = 0, rgb = 1;
    float fx = 0, fy = 0, cx = 0, cy = 0, depthScale = 1.0f;   // metres per raw unit = 1 / DepthMapFactor

    static CameraFile fromFile(const std::string &filename)
    {
        std::ifstream f(filename.c_str());
        if (!f.is_open()) throw std::runtime_error("cannot open camera file " + filename);
        std::map<std::string, double> v;
        std::string line;
        while (std::getline(f, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line = line.substr(0, hash);
            if (line.empty() || line[0] == '%' || line.compare(0, 3, "---") == 0) continue;
            const size_t c = line.find(':');
            if (c == std::string::npos) continue;
            std::string k = line.substr(0, c), val = line.substr(c + 1);
            auto trim = [](std::string &s) {
                const size_t a = s.find_first_not_of(" \t\r\n\""), b = s.find_last_not_of(" \t\r\n\"");
                s = a == std::string::npos ? "" : s.substr(a, b - a + 1);
            };
            trim(k); trim(val);
            if (k.empty() || val.empty()) continue;
            char *end = nullptr;
            const double d = std::strtod(val.c_str(), &end);
            if (end != val.c_str()) v[k] = d;
        }
        const char *need[] = {"Camera.fx", "Camera.fy", "Camera.cx", "Camera.cy", "Camera.width", "Camera.height"};
        for (const char *k : need) if (!v.count(k)) throw std::runtime_error(filename + " lacks " + k);
        CameraFile cam;
        cam.fx = (float)v["Camera.fx"]; cam.fy = (float)v["Camera.fy"]; cam.cx = (float)v["Camera.cx"]; cam.cy = (float)v["Camera.cy"];
        cam.width = (int)v["Camera.width"]; cam.height = (int)v["Camera.height"];
        const float factor = v.count("DepthMapFactor") ? (float)v["DepthMapFactor"] : 0.0f;
        cam.depthScale = std::fabs(factor) < 1e-5f ? 1.0f : 1.0f / factor;      // HRBFFusion.cpp:772-780
        cam.rgb = v.count("Camera.RGB") ? (int)v["Camera.RGB"] : 1;
        return cam;
    }
};

// ------------------------------------------------------------------------------------------------ trajectory files
/* The pose files TrajectoryManager::LoadFromFile reads for globalInputLoadTrajectory (Core/src/Utils/TrajectoryManager.cpp:
 * 61-282): "TUM" / "CoRBS" (stamp tx ty tz qx qy qz qw per line, '#' comments), "zhou" (three ints, then a 4x4 matrix in four
 * rows; every pose re-based on the first one, which becomes the identity), "ICL_NUIM_RT" (3x4 matrices, x mirrored on the left
 * and y on the right).  Poses are returned as column-major 4x4 (the layout of hrbf_set_pose / Eigen::Matrix4f::data()). */
struct PoseCM { float m[16]; };
inline void mul44cm(const float *a, const float *b, float *o)
{
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) {
        float v = 0.0f;
        for (int k = 0; k < 4; ++k) v += a[k * 4 + r] * b[c * 4 + k];
        o[c * 4 + r] = v;
    }
}
inline void rigidInverseCm(const float *a, float *o)
{
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) o[c * 4 + r] = a[r * 4 + c];
    for (int r = 0; r < 3; ++r) o[12 + r] = -(o[r] * a[12] + o[4 + r] * a[13] + o[8 + r] * a[14]);
    o[3] = o[7] = o[11] = 0.0f; o[15] = 1.0f;
}
inline std::vector<PoseCM> loadTrajectoryFile(const std::string &filename, const std::string &format, std::vector<int64_t> *stamps = nullptr)
{
    std::ifstream f(filename.c_str());
    if (!f.is_open()) throw std::runtime_error("cannot open trajectory file " + filename);
    std::vector<PoseCM> out;
    auto fromRows = [](const float r[16]) { PoseCM p; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) p.m[j * 4 + i] = r[i * 4 + j]; return p; };
    if (format == "TUM" || format == "CoRBS") {
        std::string line;
        while (!f.eof()) {
            std::getline(f, line);
            // a stream that went bad without reaching its end (a directory opened as a file, a read error) would never set eof:
            // the reference's loop spins forever there; this one reports it
            if (f.bad() || (f.fail() && !f.eof())) throw std::runtime_error(filename + ": read error in the trajectory file");
            if (line.empty() || line[0] == '#') continue;
            // TrajectoryManager.cpp:163-171,199-208: `if(file.eof()) break;` sits between the parse and the push_back, so a last
            // line WITHOUT a trailing newline is read and dropped — kept
            if (f.eof()) break;
            // the time stamp: std::remove(begin, begin + first_space, '.') closes the gaps inside the token but does not shorten
            // the string, so the token's tail keeps its old characters: "1305031102.175304" is parsed (%llu) as
            // 13050311021753044 — the digits without the dot and the last digit once more.  Reproduced, since the stamps are
            // written back by SaveTrajectoryToFile.
            const size_t sp = line.find_first_of(" ");
            std::string tok = line.substr(0, sp == std::string::npos ? line.size() : sp), kept;
            for (char ch : tok) if (ch != '.') kept.push_back(ch);
            for (size_t i = kept.size(); i < tok.size(); ++i) kept.push_back(tok[i]);
            unsigned long long utime = 0;
            float x, y, z, qx, qy, qz, qw;
            const std::string rest = kept + (sp == std::string::npos ? std::string() : line.substr(sp));
            if (sscanf(rest.c_str(), "%llu %f %f %f %f %f %f %f", &utime, &x, &y, &z, &qx, &qy, &qz, &qw) != 8) continue;   // the reference asserts n == 8
            // Eigen::Quaternionf(qw, qx, qy, qz) -> Isometry3f::rotate (TrajectoryManager.cpp:225-231): fp32, the quaternion is
            // used as read (not normalised), Eigen's toRotationMatrix operation order
            const float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
            const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
            const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
            const float R[16] = {1.0f - (tyy + tzz), txy - twz, txz + twy, x,
                                 txy + twz, 1.0f - (txx + tzz), tyz - twx, y,
                                 txz - twy, tyz + twx, 1.0f - (txx + tyy), z, 0, 0, 0, 1};
            out.push_back(fromRows(R));
            if (stamps) stamps->push_back((int64_t)utime);
        }
    } else if (format == "zhou") {
        int a, b, c;
        while (f >> a >> b >> c) {
            float r[16];
            for (float &v : r) if (!(f >> v)) throw std::runtime_error(filename + ": truncated zhou record");
            out.push_back(fromRows(r));
        }
        if (!out.empty()) {
            // poses[i] = poses[0].inverse() * poses[i] (TrajectoryManager.cpp:86-90): Eigen's general 4x4 inverse there, the rigid
            // inverse here — the same matrix up to fp32 rounding for the rigid poses such a file holds
            float inv0[16]; rigidInverseCm(out[0].m, inv0);
            for (size_t i = 1; i < out.size(); ++i) { PoseCM t; mul44cm(inv0, out[i].m, t.m); out[i] = t; }
            const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            memcpy(out[0].m, I, sizeof(I));
        }
    } else if (format == "ICL_NUIM_RT") {
        float r[16];
        for (;;) {
            bool ok = true;
            for (int i = 0; i < 12 && ok; ++i) ok = (bool)(f >> r[i]);
            if (!ok) break;
            r[12] = r[13] = r[14] = 0.0f; r[15] = 1.0f;
            // trans1 * pose * trans with trans1 = diag(-1, 1, 1, 1), trans = diag(1, -1, 1, 1): row 0 and column 1 change sign
            for (int j = 0; j < 4; ++j) r[j] = -r[j];
            for (int i = 0; i < 4; ++i) r[i * 4 + 1] = -r[i * 4 + 1];
            out.push_back(fromRows(r));
        }
    } else {
        throw std::runtime_error("trajectory format '" + format + "' is not supported (TUM, CoRBS, zhou, ICL_NUIM_RT)");
    }
    if (out.empty()) throw std::runtime_error(filename + ": no poses read");
    return out;
}

// ------------------------------------------------------------------------------------------------ PNG
struct Image {
    int width = 0, height = 0, channels = 0, bitDepth = 0;
    std::vector<uint8_t> data;    // 8-bit: channels bytes per pixel; 16-bit grey: native-endian uint16 per pixel
};

inline Image decodePng(const std::vector<uint8_t> &file)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file");
    auto be32 = [&](size_t at) { return ((uint32_t)file[at] << 24) | ((uint32_t)file[at + 1] << 16) | ((uint32_t)file[at + 2] << 8) | file[at + 3]; };
    Image im;
    int colorType = -1, interlace = 0;
    std::vector<uint8_t> idat;
    size_t at = 8;
    while (at + 12 <= file.size()) {
        const uint32_t len = be32(at);
        const char *type = (const char *)&file[at + 4];
        if (at + 12 + len > file.size()) throw std::runtime_error("truncated PNG chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            im.width = (int)be32(at + 8); im.height = (int)be32(at + 12);
            im.bitDepth = file[at + 16]; colorType = file[at + 17]; interlace = file[at + 20];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), file.begin() + (long)at + 8, file.begin() + (long)(at + 8 + len));
        } else if (!std::memcmp(type, "IEND", 4)) break;
        at += 12 + len;
    }
    if (interlace) throw std::runtime_error("interlaced PNG not supported");
    if (colorType == 0) im.channels = 1; else if (colorType == 2) im.channels = 3; else if (colorType == 6) im.channels = 4;
    else if (colorType == 4) im.channels = 2;
    else throw std::runtime_error("PNG colour type not supported (palette)");
    if (!(im.bitDepth == 8 || (im.bitDepth == 16 && im.channels == 1))) throw std::runtime_error("PNG bit depth not supported");
    const size_t bpp = (size_t)im.channels * (im.bitDepth / 8), stride = bpp * (size_t)im.width;
    std::vector<uint8_t> raw((stride + 1) * (size_t)im.height);
    uLongf outLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outLen, idat.data(), (uLong)idat.size()) != Z_OK || outLen != raw.size())
        throw std::runtime_error("PNG inflate failed");
    im.data.assign(stride * (size_t)im.height, 0);
    std::vector<uint8_t> prev(stride, 0);
    for (int y = 0; y < im.height; ++y) {
        const uint8_t *src = &raw[(stride + 1) * (size_t)y];
        const int filter = src[0];
        uint8_t *dst = &im.data[stride * (size_t)y];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? dst[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error("bad PNG filter");
            }
            dst[i] = (uint8_t)(src[1 + i] + pred);
        }
        std::memcpy(prev.data(), dst, stride);
    }
    if (im.bitDepth == 16) {   // big endian in the file -> native uint16
        uint16_t *p = (uint16_t *)im.data.data();
        for (size_t i = 0; i < (size_t)im.width * im.height; ++i) {
            const uint8_t *b = &im.data[2 * i];
            const uint16_t v = (uint16_t)((b[0] << 8) | b[1]);
            p[i] = v;
        }
    }
    return im;
}

inline std::vector<uint8_t> readFile(const std::string &path)
{
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

/* an RGB-D frame as HRBFFusion::processFrame borrows it: rgb W*H*3 (R,G,B), depth W*H uint16 */
struct Frame { int64_t timestamp = 0; std::vector<uint8_t> rgb; std::vector<uint16_t> depth; };

inline void loadRgbPng(const std::string &path, int W, int H, std::vector<uint8_t> &rgb)
{
    const Image im = decodePng(readFile(path));
    if (im.width != W || im.height != H || im.bitDepth != 8) throw std::runtime_error(path + ": unexpected size or bit depth");
    rgb.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const uint8_t *s = &im.data[i * im.channels];
        if (im.channels >= 3) { rgb[3 * i] = s[0]; rgb[3 * i + 1] = s[1]; rgb[3 * i + 2] = s[2]; }
        else rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = s[0];
    }
}
inline void loadDepthPng(const std::string &path, int W, int H, std::vector<uint16_t> &depth)
{
    const Image im = decodePng(readFile(path));
    if (im.width != W || im.height != H || im.bitDepth != 16 || im.channels != 1) throw std::runtime_error(path + ": expected a 16-bit grey PNG");
    depth.assign((const uint16_t *)im.data.data(), (const uint16_t *)im.data.data() + (size_t)W * H);
}

// ------------------------------------------------------------------------------------------------ frame sources
class AssociationReader {
public:
    AssociationReader(const std::string &assocFile, int W, int H) : W_(W), H_(H)
    {
        const size_t slash = assocFile.find_last_of('/');
        base_ = slash == std::string::npos ? "" : assocFile.substr(0, slash + 1);
        std::ifstream f(assocFile.c_str());
        if (!f.is_open()) throw std::runtime_error("cannot open " + assocFile);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ss(line);
            Entry e;
            if (ss >> e.td >> e.depth >> e.tr >> e.rgb) entries_.push_back(e);
        }
    }
    size_t size() const { return entries_.size(); }
    void read(size_t i, Frame &out) const
    {
        const Entry &e = entries_.at(i);
        out.timestamp = (int64_t)(e.td * 1000000.0);   // truncated, not rounded: GUI/src/Tools/RawImageReader.cpp:93
        loadDepthPng(base_ + e.depth, W_, H_, out.depth);
        loadRgbPng(base_ + e.rgb, W_, H_, out.rgb);
    }
private:
    struct Entry { double td, tr; std::string depth, rgb; };
    std::vector<Entry> entries_;
    std::string base_;
    int W_, H_;
};

class KlgReader {
public:
    KlgReader(const std::string &path, int W, int H, bool flipColors = false) : f_(path.c_str(), std::ios::binary), W_(W), H_(H), flip_(flipColors)
    {
        if (!f_.is_open()) throw std::runtime_error("cannot open " + path);
        int32_t n = 0;
        f_.read((char *)&n, 4);
        if (!f_ || n < 0) throw std::runtime_error(path + ": not a .klg log");
        num_ = (size_t)n;
    }
    size_t size() const { return num_; }
    /* sequential, like RawLogReader::getNext */
    void next(Frame &out)
    {
        int64_t ts; int32_t dsz, isz;
        f_.read((char *)&ts, 8); f_.read((char *)&dsz, 4); f_.read((char *)&isz, 4);
        if (!f_ || dsz < 0 || isz < 0) throw std::runtime_error("klg: truncated frame header");
        std::vector<uint8_t> d((size_t)dsz), im((size_t)isz);
        if (dsz) f_.read((char *)d.data(), dsz);
        if (isz) f_.read((char *)im.data(), isz);
        if (!f_) throw std::runtime_error("klg: truncated payload");
        const size_t n = (size_t)W_ * H_;
        out.timestamp = ts;
        out.depth.resize(n);
        if ((size_t)dsz == n * 2) std::memcpy(out.depth.data(), d.data(), n * 2);
        else {
            uLongf len = (uLongf)(n * 2);
            if (uncompress((Bytef *)out.depth.data(), &len, d.data(), (uLong)dsz) != Z_OK || len != n * 2)
                throw std::runtime_error("klg: depth does not inflate to W*H*2 bytes");
        }
        out.rgb.assign(n * 3, 0);
        if ((size_t)isz == n * 3) std::memcpy(out.rgb.data(), im.data(), n * 3);
        else if (isz > 0) {   /* JPEG colour (GUI/src/Tools/RawLogReader.cpp -> JPEGLoader.h:46-97): libjpeg's default decode, written out in hrbf_jpeg.h */
            int jw = 0, jh = 0;
            out.rgb = JpegDecoder::decodeRGB(im.data(), im.size(), jw, jh);
            if (jw != W_ || jh != H_) throw std::runtime_error("klg: the JPEG frame is not W x H");
        }
        if (flip_) for (size_t i = 0; i < n; ++i) std::swap(out.rgb[3 * i], out.rgb[3 * i + 2]);
    }
private:
    std::ifstream f_;
    int W_, H_;
    bool flip_;
    size_t num_ = 0;
};

}  // namespace hrbf_mi355
#endif
