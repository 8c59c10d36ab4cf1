// hrbf_run.cpp — the reference's caller loop in C++ (MainController::run, GUI/src/HRBF_fusion.cpp:190-497, without the
// GUI): GlobalStateParam.txt -> camera YAML -> frame source (association file or .klg) -> HRBFFusion::processFrame per
// frame -> trajectory file + PLY.  Readers: include/hrbf_io.h (zlib only); class: include/HRBFFusion.h over the C-ABI.
//
//   g++ -std=c++17 -O2 -I include tools/hrbf_run.cpp -o hrbf_run hrbffusion3d_amd/libhrbf_mi355.so -lz -Wl,-rpath,$PWD/hrbffusion3d_amd
//   hrbf_run --config GlobalStateParam.txt [--data-dir DIR] [--camera FILE.yaml] [--max-frames N] [--max-surfels N]
//            [--out trajectory.freiburg] [--ply map.ply]
//   hrbf_run --selftest --config F --data-dir D     (no GPU: parse the files, decode the first frame, print checksums)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include "HRBFFusion.h"
#include "hrbf_io.h"

using namespace hrbf_mi355;

static bool isDir(const std::string &p) { struct stat st; return !p.empty() && stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
static std::string dirOf(const std::string &p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? "." : p.substr(0, s); }
static std::string resolve(const GlobalState &g, const std::string &name, const std::string &base)
{
    if (name.empty() || name[0] == '/') return name;
    if (isDir(g.currentWorkingDirectory)) return g.currentWorkingDirectory + "/" + name;   // the reference chdir()s there
    return base + "/" + name;
}
static uint64_t fnv(const void *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= ((const uint8_t *)p)[i]; h *= 1099511628211ull; }
    return h;
}

/* GlobalStateParam -> hrbf_params, the reads of the path: HRBF_fusion.cpp:87-96 (constructor arguments),
   HRBFFusion.cpp:1263-1345, RGBDOdometry.cpp, IndexMap.cpp:413-518, GlobalModel.cpp:551-688 */
static hrbf_params paramsFrom(const GlobalState &g, const CameraFile &cam, int maxSurfels)
{
    hrbf_params p = paramsFromGlobalState(g, cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.depthScale, g.globalConfidenceThreshold,
                                          g.globalDepthCutoff, g.registrationJointICPWeight, false, g.registrationPreAlignSO3, false);
    p.max_surfels = maxSurfels;
    return p;
}

int main(int argc, char **argv)
{
    std::string config, dataDir, cameraFile, out, ply;
    int maxFrames = 0, maxSurfels = 4 * 1024 * 1024;
    bool selftest = false, dumpParams = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--config") config = next(); else if (a == "--data-dir") dataDir = next(); else if (a == "--camera") cameraFile = next();
        else if (a == "--out") out = next(); else if (a == "--ply") ply = next(); else if (a == "--max-frames") maxFrames = atoi(next().c_str());
        else if (a == "--max-surfels") maxSurfels = atoi(next().c_str()); else if (a == "--selftest") selftest = true;
        else if (a == "--dump-params") dumpParams = true;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (config.empty()) { fprintf(stderr, "usage: hrbf_run --config GlobalStateParam.txt [--data-dir D] [--camera Y] [--max-frames N] [--out T] [--ply P]\n"); return 2; }
    try {
        const GlobalState g = GlobalState::fromFile(config);
        if (dumpParams) {   // every member as read, one "name<TAB>[value]" per line (compared with the reference's own reader in tests/test_config.py)
#define D_S(n) printf("%s\t[%s]\n", #n, g.n.c_str());
#define D_I(n) printf("%s\t[%d]\n", #n, (int)g.n);
#define D_F(n) printf("%s\t[%.9g]\n", #n, (double)g.n);
            D_S(currentWorkingDirectory) D_I(sensorType) D_S(klgFileName) D_S(AssociationFile) D_S(parameterFileCvFormat)
            D_I(optimizationUseLocalBA) D_I(optimizationUseGlobalBA) D_I(preprocessingUsebilateralFilter)
            D_F(preprocessingInitRadiusMultiplier) D_F(preprocessingCurvEstimationWindow) D_F(preprocessingCurvValidThreshold)
            D_F(preprocessingNormalEstimationPCA) D_I(preprocessingUseConfEval) D_F(preprocessingConfEvalEpsilon)
            D_I(registrationPreAlignSO3) D_F(registrationJointICPWeight) D_I(registrationICPUseSparseICP)
            D_I(registrationICPUseCoorespondenceSearch) D_I(registrationICPNeighborSearchRadius) D_I(registrationICPUseWeightedICP)
            D_F(registrationICPCurvWeightImpactControl) D_I(registrationColorUseRGBGrad) D_F(preictionWindowMultiplier)
            D_I(preictionMinNeighbors) D_I(preictionMaxNeighbors) D_F(preictionConfThreshold) D_F(fusionCleanWindowMultiplier)
            D_F(globalConfidenceThreshold) D_F(globalDenseEnoughThresh) D_F(globalDepthCutoff) D_I(globalInputICLNUIMDataset)
            D_I(globalInputLoadTrajectory) D_S(globalInputTrajectoryFormat) D_S(globalInputTrajectoryFile)
            D_F(globalOutputSavePointCloudConfThreshold) D_I(globalStartFrame) D_I(globalEndFrame) D_I(globalFrameToSkip)
            D_F(registrationICPErrorThreshold) D_F(registrationICPCovarianceThreshold) D_F(registrationColorPhotoThreshold)
            D_I(globalOutputSaveTrjectoryFile) D_S(globalOutputSaveTrjectoryFileType)
#undef D_S
#undef D_I
#undef D_F
            return 0;
        }
        const std::string base = dataDir.empty() ? dirOf(config) : dataDir;
        if (cameraFile.empty()) cameraFile = resolve(g, g.parameterFileCvFormat, base);
        const CameraFile cam = CameraFile::fromFile(cameraFile);
        if (g.optimizationUseLocalBA || g.optimizationUseGlobalBA)
            fprintf(stderr, "note: optimizationUseLocalBA / GlobalBA are set; the sparse ORB back-end is out of scope (front-end only)\n");
        const bool flip = cam.rgb == 0;
        AssociationReader *assoc = nullptr; KlgReader *klg = nullptr;
        size_t total = 0, klgPos = 0;
        if (g.sensorType == 3) { assoc = new AssociationReader(resolve(g, g.AssociationFile, base), cam.width, cam.height); total = assoc->size(); }
        else if (g.sensorType == 2) { klg = new KlgReader(resolve(g, g.klgFileName, base), cam.width, cam.height, flip); total = klg->size(); }
        else throw std::runtime_error("sensorType 1 (live camera) has no counterpart here");
        if (maxFrames && (size_t)maxFrames < total) total = (size_t)maxFrames;
        Frame fr;
        auto fetch = [&](size_t i) {   // frame i of the source (the .klg log is sequential: read forward to it)
            if (assoc) { assoc->read(i, fr); if (flip) for (size_t k = 0; k < fr.depth.size(); ++k) std::swap(fr.rgb[3 * k], fr.rgb[3 * k + 2]); }
            else { while (klgPos <= i) { klg->next(fr); ++klgPos; } }
        };
        if (selftest) {
            if (total) fetch(0);
            printf("{\"sensorType\": %d, \"frames\": %zu, \"width\": %d, \"height\": %d, \"fx\": %.9g, \"fy\": %.9g, \"cx\": %.9g, \"cy\": %.9g, "
                   "\"depth_scale\": %.9g, \"rgb_order\": %d, \"confidence\": %.9g, \"depth_cutoff\": %.9g, \"icp_weight\": %.9g, \"so3\": %d, "
                   "\"bilateral\": %d, \"min_neighbors\": %d, \"search_radius\": %d, \"icl\": %d, \"timestamp0\": %lld, "
                   "\"rgb_fnv\": \"%016llx\", \"depth_fnv\": \"%016llx\"}\n",
                   g.sensorType, total, cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.depthScale, cam.rgb,
                   g.globalConfidenceThreshold, g.globalDepthCutoff, g.registrationJointICPWeight, (int)g.registrationPreAlignSO3,
                   (int)g.preprocessingUsebilateralFilter, g.preictionMinNeighbors, g.registrationICPNeighborSearchRadius,
                   (int)g.globalInputICLNUIMDataset, (long long)fr.timestamp,
                   (unsigned long long)fnv(fr.rgb.data(), fr.rgb.size()), (unsigned long long)fnv(fr.depth.data(), fr.depth.size() * 2));
            return 0;
        }
        HRBFFusion fusion(paramsFrom(g, cam, maxSurfels));
        /* globalInputLoadTrajectory: replay the poses of globalInputTrajectoryFile instead of registering
           (HRBFFusion.cpp:55-59,1105-1108) */
        if (g.globalInputLoadTrajectory) fusion.loadTrajectory(resolve(g, g.globalInputTrajectoryFile, base), g.globalInputTrajectoryFormat);
        const auto t0 = std::chrono::steady_clock::now();
        size_t n = 0;
        /* MainController::run (GUI/src/HRBF_fusion.cpp:190-239): globalStartFrame fast-forwards source and tick, globalFrameToSkip
           is applied once (tick jump + fusion weight), globalEndFrame bounds the tick */
        int framesToSkip = g.globalFrameToSkip;
        const int start = g.globalStartFrame;
        const int end = g.globalEndFrame > 0 ? g.globalEndFrame : 65535;
        size_t cur = 0;   // logReader->currentFrame
        while (cur < total && fusion.getTick() < end) {
            fetch(cur); ++cur;                                   // logReader->getNext()
            if (fusion.getTick() < start) {
                fusion.setTick(start);
                cur = (size_t)start;                             // fastForward(start); getNext()
                if (cur >= total) break;
                fetch(cur); ++cur;
            }
            const float weightMultiplier = (float)(framesToSkip + 1);
            if (framesToSkip > 0) {
                fusion.setTick(fusion.getTick() + framesToSkip);
                cur += (size_t)framesToSkip;                     // fastForward(currentFrame + framesToSkip)
                framesToSkip = 0;
            }
            fusion.processFrame(fr.rgb.data(), fr.depth.data(), fr.timestamp, weightMultiplier);   // enqueues; the readers overlap with the GPU
            ++n;
        }
        const size_t poses = fusion.getTrajectory().size();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!out.empty()) fusion.trajectory_manager->SaveTrajectoryToFile("TUM", out, g.globalInputICLNUIMDataset);
        if (!ply.empty()) fusion.savePly(ply, g.globalOutputSavePointCloudConfThreshold);
        printf("{\"frames\": %zu, \"poses\": %zu, \"seconds\": %.4f, \"fps_including_io\": %.2f, \"surfels\": %u}\n", n, poses, dt,
               dt > 0 ? n / dt : 0.0, fusion.getGlobalModel().lastCount());
        delete assoc; delete klg;
    } catch (const std::exception &e) {
        fprintf(stderr, "hrbf_run: %s\n", e.what());
        return 1;
    }
    return 0;
}
