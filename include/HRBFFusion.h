/*
 * HRBFFusion.h — header-only C++ shim with the reference's class surface over the C-ABI of
 * libhrbf_mi355.so, so that a caller written against Core/src/HRBFFusion.h (e.g.
 * GUI/src/HRBF_fusion.cpp:190-497) can switch libraries.
 *
 * Mirrors (reference file:line):
 *   HRBFFusion::HRBFFusion(countThresh, errThresh, confidence, depthCut, icpThresh, fastOdom, so3,
 *                          frameToFrameRGB)                         Core/src/HRBFFusion.h:87-94
 *   void processFrame(rgb, depth, timestamp, weightMultiplier)      Core/src/HRBFFusion.h:110-113
 *   getCurrPose / getTick / setTick / getGlobalModel / setters      Core/src/HRBFFusion.h:115-260
 *   GlobalModel::lastCount / downloadMap                            Core/src/GlobalModel.cpp:770-804
 *   savePly (binary PLY, 13 properties, normals negated)            Core/src/HRBFFusion.cpp:1737-1853
 *   TrajectoryManager::SaveTrajectoryToFile (TUM / zhou / lefloch)   Core/src/Utils/TrajectoryManager.cpp:284-373
 *
 * Differences a maintainer must know:
 *   - both construction styles exist: the reference's 8-argument constructor over the Resolution / Intrinsics /
 *     GlobalStateParam singletons (and the camera YAML for 1/DepthMapFactor), and one that takes them as arguments;
 *   - poses are exposed as 16 floats, column-major (bit-compatible with Eigen::Matrix4f::data());
 *     with Eigen available (#include <Eigen/Core> first) getCurrPose() returns an Eigen::Map;
 *   - GL handles (model(), textures) do not exist: use downloadMap() / getImage();
 *   - downloadMap() returns the CURRENT map; the reference copies from vbos[renderSource]
 *     (GlobalModel.cpp:791), i.e. the buffer of the previous pass with the new count;
 *   - errors throw std::runtime_error instead of exit(0) (Core/src/Cuda/convenience.cuh:64-71);
 *   - processFrame() only ENQUEUES the frame (no host synchronisation).  The reference pushes currPose into
 *     trajectory_manager->poses inside processFrame (HRBFFusion.cpp:1058,1129-1133); here the device appends every
 *     frame's pose to a pinned ring (hrbf_get_pose_log) and `poses` is brought up to date lazily — by
 *     syncTrajectory(), by SaveTrajectoryToFile() and by getTrajectory().  A caller that reads
 *     trajectory_manager->poses directly calls syncTrajectory() first.  getCurrPose() blocks, as it must.
 */
#ifndef HRBF_MI355_HRBFFUSION_H_
#define HRBF_MI355_HRBFFUSION_H_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "hrbf_mi355.h"
#include "hrbf_io.h"   // ParameterFile, GlobalState, CameraFile, trajectory files (header-only; zlib only if the PNG / klg readers are used)

namespace hrbf_mi355 {

struct Pose { float m[16]; };   // column-major 4x4, T_wc

/* ---- the three singletons the reference's HRBFFusion constructor reads instead of taking arguments -------------------------
 * Resolution / Intrinsics (Core/src/Utils/Resolution.h:26-68, Intrinsics.h:26-66): initialised by the first getInstance() call
 * that carries values (GUI/src/HRBF_fusion.cpp:51-52), read-only afterwards.  GlobalStateParam (Utils/GlobalStateParams.h:
 * 66-128): public members named like the keys of GlobalStateParam.txt, filled by readMembers(ParameterFile).  With these a
 * caller written against the reference — `GlobalStateParam::getInstance().readMembers(pf); Resolution::getInstance(w, h);
 * Intrinsics::getInstance(fx, fy, cx, cy); new HRBFFusion(icpCountThresh, icpErrThresh, confidence, depth, icp, fastOdom,
 * so3, frameToFrameRGB)` — compiles against this header unchanged (tests/cpp/reference_caller_test.cpp). */
class Resolution {
public:
    static const Resolution &getInstance(int width = 0, int height = 0)
    {
        static const Resolution instance(width, height);
        return instance;
    }
    const int &width() const { return w_; }
    const int &height() const { return h_; }
    const int &cols() const { return w_; }
    const int &rows() const { return h_; }
    const int &numPixels() const { return n_; }
private:
    Resolution(int w, int h) : w_(w), h_(h), n_(w * h)
    {
        if (!(w > 0 && h > 0)) throw std::runtime_error("You haven't initialised the Resolution class!");
    }
    const int w_, h_, n_;
};
class Intrinsics {
public:
    static const Intrinsics &getInstance(float fx = 0, float fy = 0, float cx = 0, float cy = 0)
    {
        static const Intrinsics instance(fx, fy, cx, cy);
        return instance;
    }
    const float &fx() const { return fx_; }
    const float &fy() const { return fy_; }
    const float &cx() const { return cx_; }
    const float &cy() const { return cy_; }
private:
    Intrinsics(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy)
    {
        if (!(fx != 0 && fy != 0)) throw std::runtime_error("You haven't initialised the Intrinsics class!");
    }
    const float fx_, fy_, cx_, cy_;
};
class GlobalStateParam : public GlobalState {
public:
    static GlobalStateParam &getInstance() { static GlobalStateParam s; return s; }
    static GlobalStateParam &get() { return getInstance(); }
    void readMembers(const ParameterFile &parameterFile) { static_cast<GlobalState &>(*this) = GlobalState::fromParameterFile(parameterFile); }
};
/* hrbf_params from the singletons + the reference constructor's arguments: exactly what HRBFFusion::HRBFFusion and the
   members it constructs read (HRBFFusion.cpp:26-75, the Uniform pushes of :1262-1345, GlobalModel.cpp, IndexMap.cpp, FillIn.cpp,
   RGBDOdometry.cpp); the camera YAML named by parameterFileCvFormat supplies 1 / DepthMapFactor (HRBFFusion.cpp:772-780) */
inline hrbf_params paramsFromGlobalState(const GlobalState &g, int width, int height, float fx, float fy, float cx, float cy,
                                         float depthScale, float confidence, float depthCut, float icpThresh, bool fastOdom,
                                         bool so3, bool frameToFrameRGB)
{
    hrbf_params p;
    hrbf_default_params(&p, width, height, fx, fy, cx, cy, depthScale);
    p.confidence_threshold = confidence; p.depth_cutoff = depthCut; p.icp_weight = icpThresh;
    p.fast_odom = fastOdom; p.so3 = so3; p.frame_to_frame_rgb = frameToFrameRGB;
    p.use_bilateral = g.preprocessingUsebilateralFilter; p.init_radius_multiplier = g.preprocessingInitRadiusMultiplier;
    p.curv_estimation_window = g.preprocessingCurvEstimationWindow; p.curv_valid_threshold = g.preprocessingCurvValidThreshold;
    p.normal_estimation_pca = g.preprocessingNormalEstimationPCA; p.use_conf_eval = g.preprocessingUseConfEval;
    p.conf_eval_epsilon = g.preprocessingConfEvalEpsilon;
    p.icp_use_corr_search = g.registrationICPUseCoorespondenceSearch; p.icp_search_radius = g.registrationICPNeighborSearchRadius;
    p.icp_use_weighted = g.registrationICPUseWeightedICP; p.icp_curv_weight_lambda = g.registrationICPCurvWeightImpactControl;
    p.rgb_use_grad_weight = g.registrationColorUseRGBGrad; p.use_sparse_icp = g.registrationICPUseSparseICP;
    p.predict_window_multiplier = g.preictionWindowMultiplier; p.predict_min_neighbors = g.preictionMinNeighbors;
    p.predict_max_neighbors = g.preictionMaxNeighbors; p.predict_conf_threshold = g.preictionConfThreshold;
    p.clean_window_multiplier = g.fusionCleanWindowMultiplier; p.dense_enough_thresh = g.globalDenseEnoughThresh;
    p.load_trajectory = g.globalInputLoadTrajectory;
    return p;
}

class GlobalModel {
public:
    explicit GlobalModel(hrbf_handle h) : h_(h) {}
    unsigned int lastCount() const { return hrbf_surfel_count(h_); }
    /* surfels held by THIS process: == lastCount() unless the map is sharded over ranks (hrbf_map_shard_init) */
    unsigned int localCount() const { return hrbf_local_surfel_count(h_); }
    /* caller-owned array of localCount()*20 floats (= Eigen::Vector4f[count*5]); delete[] it */
    float *downloadMap() const
    {
        const unsigned int n = localCount();
        float *out = new float[(size_t)(n ? n : 1) * 20];
        if (n && hrbf_download_map(h_, out, n) != HRBF_OK) { delete[] out; throw std::runtime_error(hrbf_last_error()); }
        return out;
    }
    /* GlobalModel::updateModel (GlobalModel.cpp:690-767): DeltaTransformKF as n column-major 4x4 matrices
       (Eigen::Matrix4f::data() of each element, back to back) */
    void updateModel(const float *deltaTransformKF16, int n)
    {
        if (hrbf_update_model(h_, deltaTransformKF16, n) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
    /* lActiveKFID as a byte mask indexed by submap id; n = 0: all active */
    void setActiveSubmaps(const unsigned char *active, int n)
    {
        if (hrbf_set_active_submaps(h_, active, n) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
    /* GlobalModel::{initialise,fuse,clean} (GlobalModel.h:50-107) with the scalar arguments of the reference; the
       GPUTexture arguments are the context's images.  pose: column-major 4x4 T_wc (Eigen::Matrix4f::data()). */
    void initialise(const float *init_pose16)
    {
        if (hrbf_initialise(h_, init_pose16) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
    void fuse(const float *pose16, const int &time, const float depthCutoff, const float indexsubmap = 0.f)
    {
        if (hrbf_fuse(h_, pose16, time, depthCutoff, (int)indexsubmap) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
    void clean(const float *pose16, const int &time, const float confThreshold, const float maxDepth)
    {
        if (hrbf_clean(h_, pose16, time, confThreshold, maxDepth) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
    /* sharded map only: re-cut the ranges evenly over the ranks (new surfels land on the last rank) */
    void rebalance()
    {
        if (hrbf_map_rebalance(h_) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
private:
    hrbf_handle h_;
};

/* IndexMap::{predictIndices,predictHRBF} (IndexMap.h:43-68); the images are fetched with HRBFFusion::getImage */
class IndexMap {
public:
    explicit IndexMap(hrbf_handle h) : h_(h) {}
    void predictIndices(const float *pose16, const int &time, const float depthCutoff, const int indexSubmap = 0)
    {
        if (hrbf_predict_indices(h_, pose16, time, depthCutoff, indexSubmap) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
    void predictHRBF()
    {
        if (hrbf_predict_hrbf(h_) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }
private:
    hrbf_handle h_;
};

class TrajectoryManager {
public:
    std::vector<Pose> poses;         // complete up to the last syncTrajectory() (see the header comment)
    std::vector<int64_t> timstamp;   // (sic) reference member name
    std::function<void()> sync;      // installed by HRBFFusion: pulls the poses the device has logged since

    /* format: "TUM" (ts tx ty tz qx qy qz qw; icl_nuim negates ty and prints an integer stamp),
       "zhou" (.log) or "lefloch" — TrajectoryManager.cpp:284-373 */
    bool SaveTrajectoryToFile(const std::string &type, const std::string &fname, bool icl_nuim = false) const
    {
        if (sync) sync();
        if (type == "zhou") {
            FILE *f = fopen(fname.c_str(), "w");
            if (!f) return false;
            for (size_t i = 0; i < poses.size(); ++i) {
                const float *m = poses[i].m;
                fprintf(f, "%d %d %d\n", (int)i, (int)i, (int)(i + 1));
                for (int r = 0; r < 4; ++r) fprintf(f, "%f %f %f %f\n", m[r], m[4 + r], m[8 + r], m[12 + r]);
            }
            fclose(f);
            return true;
        }
        if (type == "TUM") {
            FILE *f = fopen(fname.c_str(), "w");
            if (!f) return false;
            for (size_t i = 0; i < poses.size(); ++i) {
                const float *m = poses[i].m;
                float tx = m[12], ty = m[13], tz = m[14];
                const int64_t ts = i < timstamp.size() ? timstamp[i] : (int64_t)i;
                if (icl_nuim) { fprintf(f, "%d ", (int)ts); ty = -ty; }
                else fprintf(f, "%.6f ", (double)ts / 1000000.0);
                float q[4];
                quaternion(m, q);
                fprintf(f, "%g %g %g %g %g %g %g\n", tx, ty, tz, q[0], q[1], q[2], q[3]);
            }
            fclose(f);
            return true;
        }
        if (type == "lefloch") {
            FILE *f = fopen(fname.c_str(), "w");
            if (!f) return false;
            for (size_t i = 0; i < poses.size(); ++i) {
                fprintf(f, "%zu ", i);
                for (int k = 0; k < 16; ++k) fprintf(f, "%g ", poses[i].m[k]);   // column-major like poses[i](k, j)
                fprintf(f, "\n");
            }
            fclose(f);
            return true;
        }
        return false;
    }

    /* rotation (column-major 4x4) -> unit quaternion x y z w (Eigen::Quaternionf(Matrix3f) convention) */
    static void quaternion(const float *m, float q[4])
    {
        const float r00 = m[0], r11 = m[5], r22 = m[10];
        const float t = r00 + r11 + r22;
        if (t > 0.0f) {
            float s = std::sqrt(t + 1.0f);
            q[3] = 0.5f * s; s = 0.5f / s;
            q[0] = (m[6] - m[9]) * s; q[1] = (m[8] - m[2]) * s; q[2] = (m[1] - m[4]) * s;
        } else {
            int i = 0;
            if (r11 > r00) i = 1;
            if (r22 > (i == 0 ? r00 : r11)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            auto R = [&](int r, int c) { return m[c * 4 + r]; };
            float s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0f);
            q[i] = 0.5f * s; s = 0.5f / s;
            q[3] = (R(k, j) - R(j, k)) * s; q[j] = (R(j, i) + R(i, j)) * s; q[k] = (R(k, i) + R(i, k)) * s;
        }
    }
};

/* What callers read of the reference's frame-to-model odometry object (Core/src/Utils/RGBDOdometry.h:124-125): two public
   floats, `getFrameToModel().lastICPError` / `.lastICPCount` (GUI/src/HRBF_fusion.cpp:285-295).  Here they mirror
   hrbf_last_icp(): HRBFFusion::getFrameToModel() refreshes them (a blocking read of the last finished frame's residual
   record) and hands out the object, so the caller's lines compile and behave unchanged. */
class RGBDOdometry {
public:
    RGBDOdometry() : lastICPError(0.0f), lastICPCount(0.0f) {}
    float lastICPError;   // sqrt(residual[0]) / residual[1] of the last ICP iteration (RGBDOdometry.cpp:1135)
    float lastICPCount;   // residual[1]: number of inlier correspondences (RGBDOdometry.cpp:1136)
};

class HRBFFusion {
public:
    HRBFFusion(int width, int height, float fx, float fy, float cx, float cy, float depthScale,
               const int countThresh = 35000, const float errThresh = 5e-05f, const float confidence = 10.0f,
               const float depthCut = 3.0f, const float icpThresh = 10.0f, const bool fastOdom = false,
               const bool so3 = true, const bool frameToFrameRGB = false, int maxSurfels = 4596 * 4596, int device = 0)
        : h_(nullptr), model_(nullptr), load_trajectory_(false), synced_(0)
    {
        (void)countThresh; (void)errThresh;   // stored but never read on this path in the reference either
        hrbf_params p;
        hrbf_default_params(&p, width, height, fx, fy, cx, cy, depthScale);
        p.confidence_threshold = confidence; p.depth_cutoff = depthCut; p.icp_weight = icpThresh;
        p.fast_odom = fastOdom; p.so3 = so3; p.frame_to_frame_rgb = frameToFrameRGB; p.max_surfels = maxSurfels;
        if (hrbf_create(&p, device, &h_) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
        model_ = new GlobalModel(h_);
        index_ = new IndexMap(h_);
        trajectory_manager = new TrajectoryManager();
        trajectory_manager->sync = [this]() { this->syncTrajectory(); };
    }
    /* THE REFERENCE'S SIGNATURE (Core/src/HRBFFusion.h:87-94): resolution and intrinsics from the singletons, every tunable
       from GlobalStateParam::get(), 1 / DepthMapFactor from the camera YAML it names (HRBFFusion.cpp:682-781); with
       globalInputLoadTrajectory the poses of globalInputTrajectoryFile are replayed (HRBFFusion.cpp:55-59,1105-1108).
       The sparse back-end options (optimizationUseLocalBA / GlobalBA) are outside this library: refused, not ignored. */
    HRBFFusion(const int countThresh = 35000, const float errThresh = 5e-05f, const float confidence = 10.0f,
               const float depthCut = 3.0f, const float icpThresh = 10.0f, const bool fastOdom = false, const bool so3 = true,
               const bool frameToFrameRGB = false)
        : h_(nullptr), model_(nullptr), load_trajectory_(false), synced_(0)
    {
        (void)countThresh; (void)errThresh;
        const GlobalStateParam &g = GlobalStateParam::get();
        if (g.optimizationUseLocalBA || g.optimizationUseGlobalBA)
            throw std::runtime_error("optimizationUseLocalBA / optimizationUseGlobalBA: the ORB-SLAM2 back-end is not part of libhrbf_mi355");
        const Resolution &res = Resolution::getInstance();
        const Intrinsics &K = Intrinsics::getInstance();
        float depthScale = 1.0f / 5000.0f;
        if (!g.parameterFileCvFormat.empty()) depthScale = CameraFile::fromFile(g.parameterFileCvFormat).depthScale;
        const hrbf_params p = paramsFromGlobalState(g, res.width(), res.height(), K.fx(), K.fy(), K.cx(), K.cy(), depthScale, confidence,
                                                    depthCut, icpThresh, fastOdom, so3, frameToFrameRGB);
        load_trajectory_ = p.load_trajectory != 0;
        if (hrbf_create(&p, 0, &h_) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
        model_ = new GlobalModel(h_);
        index_ = new IndexMap(h_);
        trajectory_manager = new TrajectoryManager();
        trajectory_manager->sync = [this]() { this->syncTrajectory(); };
        if (load_trajectory_) {
            try { loadTrajectory(g.globalInputTrajectoryFile, g.globalInputTrajectoryFormat); }
            catch (...) {       // a constructor that throws runs no destructor: release what was acquired, then pass the error on
                delete trajectory_manager; delete index_; delete model_; hrbf_destroy(h_);
                trajectory_manager = nullptr; index_ = nullptr; model_ = nullptr; h_ = nullptr;
                throw;
            }
        }
    }
    /* globalInputLoadTrajectory: TrajectoryManager::LoadFromFile + `currPose = poses[0]` (HRBFFusion.cpp:55-59); from then on
       processFrame sets currPose = poses[tick - 1] instead of registering (HRBFFusion.cpp:1105-1108) */
    void loadTrajectory(const std::string &file, const std::string &format)
    {
        const std::vector<PoseCM> t = loadTrajectoryFile(file, format, &trajectory_manager->timstamp);
        replay_.clear(); trajectory_manager->poses.clear();
        for (const PoseCM &q : t) { Pose p; memcpy(p.m, q.m, sizeof(p.m)); replay_.push_back(p); trajectory_manager->poses.push_back(p); }
        setLoadTrajectory(true);
        setPose(replay_[0].m);
    }
    /* every tunable at once: `p` as filled from GlobalStateParam (hrbf_default_params + the fields of
       GUI/GlobalStateParam.txt:20-81; tools/hrbf_run.cpp shows the mapping) */
    explicit HRBFFusion(const hrbf_params &p, int device = 0) : h_(nullptr), model_(nullptr), load_trajectory_(p.load_trajectory != 0), synced_(0)
    {
        if (hrbf_create(&p, device, &h_) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
        model_ = new GlobalModel(h_);
        index_ = new IndexMap(h_);
        trajectory_manager = new TrajectoryManager();
        trajectory_manager->sync = [this]() { this->syncTrajectory(); };
    }
    ~HRBFFusion() { delete trajectory_manager; delete index_; delete model_; hrbf_destroy(h_); }
    HRBFFusion(const HRBFFusion &) = delete;
    HRBFFusion &operator=(const HRBFFusion &) = delete;

    void processFrame(const unsigned char *rgb, const unsigned short *depth, const int64_t &timestamp,
                      const float weightMultiplier = 1.f)
    {
        const int tick_before = hrbf_get_tick(h_);
        icp_cached_ = false;            /* whatever happens below, the values of the previous frame are not this frame's */
        replayPose(tick_before);
        if (hrbf_process_frame(h_, rgb, depth, timestamp, weightMultiplier) != HRBF_OK)
            throw std::runtime_error(hrbf_last_error());
        /* HRBFFusion.cpp:1058,1129-1133: the first frame pushes the initial pose, every later frame currPose and its
           time stamp — unless the trajectory is being replayed (globalInputLoadTrajectory).  No synchronisation here:
           which frames push is recorded, the poses are pulled from the device-written ring on demand. */
        const bool pushes = tick_before == 1 || !load_trajectory_;
        pushes_.push_back(pushes);
        if (tick_before > 1 && pushes) trajectory_manager->timstamp.push_back(timestamp);
        if (hrbf_frames_enqueued(h_) - synced_ > 16384u) pullPoses(false);
    }
    /* same, inputs already in device memory (HBM): nothing is copied and nothing blocks */
    void processFrameDevice(const void *d_rgb, const void *d_depth, const int64_t &timestamp, const float weightMultiplier = 1.f)
    {
        const int tick_before = hrbf_get_tick(h_);
        icp_cached_ = false;
        replayPose(tick_before);
        if (hrbf_process_frame_device(h_, d_rgb, d_depth, timestamp, weightMultiplier) != HRBF_OK)
            throw std::runtime_error(hrbf_last_error());
        const bool pushes = tick_before == 1 || !load_trajectory_;
        pushes_.push_back(pushes);
        if (tick_before > 1 && pushes) trajectory_manager->timstamp.push_back(timestamp);
        if (hrbf_frames_enqueued(h_) - synced_ > 16384u) pullPoses(false);
    }
    /* bring trajectory_manager->poses up to date (blocks until every enqueued frame is done) */
    void syncTrajectory() { pullPoses(true); }
    /* wait = false: only the frames whose pose has already landed (never blocks).  processFrame calls this once a quarter of
       the device's 65536-entry ring is pending, so that a caller who never looks at the trajectory cannot overrun it */
    void pullPoses(bool wait)
    {
        if (!replay_.empty()) return;   // replayed trajectory: `poses` is the loaded file, nothing is pushed
        const uint32_t n = wait ? hrbf_frames_enqueued(h_) : hrbf_frames_completed(h_);
        if (synced_ >= n) return;
        std::vector<float> buf((size_t)(n - synced_) * 16);
        const int got = hrbf_get_pose_log(h_, synced_, n - synced_, buf.data(), wait ? 1 : 0);
        if (got < 0) throw std::runtime_error(hrbf_last_error());
        for (int k = 0; k < got; ++k) {
            if (synced_ + (uint32_t)k < pushes_.size() && !pushes_[synced_ + (uint32_t)k]) continue;
            Pose p; memcpy(p.m, &buf[(size_t)k * 16], sizeof(p.m));
            trajectory_manager->poses.push_back(p);
        }
        synced_ += (uint32_t)got;
    }
    const std::vector<Pose> &getTrajectory() { syncTrajectory(); return trajectory_manager->poses; }
    /* frames whose pose has landed; never blocks (hrbf_frames_completed) */
    unsigned int framesCompleted() { return hrbf_frames_completed(h_); }
    void synchronize() { if (hrbf_synchronize(h_) != HRBF_OK) throw std::runtime_error(hrbf_last_error()); }
    /* globalInputLoadTrajectory: poses come from setPose() before each frame and nothing is pushed */
    void setLoadTrajectory(bool on) { load_trajectory_ = on; hrbf_set_load_trajectory(h_, on ? 1 : 0); }
    void setPose(const float *pose16) { if (hrbf_set_pose(h_, pose16) != HRBF_OK) throw std::runtime_error(hrbf_last_error()); }

    const float *getCurrPoseData() { hrbf_get_pose(h_, curr_.m); return curr_.m; }
#ifdef EIGEN_CORE_H
    Eigen::Map<const Eigen::Matrix4f> getCurrPose() { return Eigen::Map<const Eigen::Matrix4f>(getCurrPoseData()); }
#else
    const float *getCurrPose() { return getCurrPoseData(); }
#endif
    const int getTick() { return hrbf_get_tick(h_); }
    void setTick(const int &val) { hrbf_set_tick(h_, val); }
    GlobalModel &getGlobalModel() { return *model_; }
    IndexMap &getIndexMap() { return *index_; }
    /* Core/src/HRBFFusion.h:217 */
    /* the read-out synchronises the stream: done once per processed frame however often the caller asks (the reference's GUI
       reads lastICPError / lastICPCount six times per frame); after a failed read the values are NaN, not the previous frame's */
    RGBDOdometry &getFrameToModel()
    {
        const unsigned int now = hrbf_frames_enqueued(h_);
        if (!icp_cached_ || icp_cached_at_ != now) {
            /* a failed read is reported (NaN) but NOT cached: the next call asks again (round-5 advice) */
            const bool ok = hrbf_last_icp(h_, &frame_to_model_.lastICPError, &frame_to_model_.lastICPCount) == HRBF_OK;
            if (!ok) frame_to_model_.lastICPError = frame_to_model_.lastICPCount = std::numeric_limits<float>::quiet_NaN();
            icp_cached_ = ok; icp_cached_at_ = now;
        }
        return frame_to_model_;
    }
    /* a caller that runs a registration through the C-ABI's operator seams on handle() (hrbf_run_stage(ODOMETRY), hrbf_icp_step ...)
       — calls that do not count as frames — drops the cached values with this before reading getFrameToModel() again */
    void invalidateFrameToModel() { icp_cached_ = false; }
    float lastICPError() { return getFrameToModel().lastICPError; }
    float lastICPCount() { return getFrameToModel().lastICPCount; }
    /* setters applied every GUI frame (GUI/src/HRBF_fusion.cpp:448-456) */
    void setRgbOnly(const bool &val) { hrbf_set_rgb_only(h_, val); }
    void setIcpWeight(const float &val) { hrbf_set_icp_weight(h_, val); }
    void setPyramid(const bool &val) { hrbf_set_pyramid(h_, val); }
    void setFastOdom(const bool &val) { hrbf_set_fast_odom(h_, val); }
    void setSo3(const bool &val) { hrbf_set_so3(h_, val); }
    void setFrameToFrameRGB(const bool &val) { hrbf_set_frame_to_frame_rgb(h_, val); }
    void setConfidenceThreshold(const float &val) { hrbf_set_confidence_threshold(h_, val); }
    void setDepthCutoff(const float &val) { hrbf_set_depth_cutoff(h_, val); }
    bool getImage(int which, void *out, size_t bytes) { return hrbf_get_image(h_, which, out, bytes) == HRBF_OK; }
    hrbf_handle handle() { return h_; }
    /* more than one GPU, one process per GPU sharing ONE sequence: join the communicator whose id
       rank 0 obtained from hrbf_comm_unique_id; shardMap additionally cuts the (still empty) surfel map over the ranks */
    void joinRanks(int rank, int world, const unsigned char id128[128], bool shardMap, bool ownByHash = false)
    {
        if (hrbf_comm_init(h_, rank, world, id128) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
        /* ownByHash: ownership by spatial hash of the surfel's cell instead of contiguous ranges of the global order */
        if (shardMap && hrbf_map_shard_init(h_, ownByHash ? 2 : 1) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
    }

    /* binary little-endian PLY, 13 properties (HRBFFusion.cpp:1737-1853) */
    void savePly(const std::string &filename, float confThreshold = 0.0f)
    {
        const unsigned int n = model_->localCount();
        float *map = model_->downloadMap();
        int valid = 0;
        for (unsigned int i = 0; i < n; ++i) valid += map[(size_t)i * 20 + 3] > confThreshold;
        std::ofstream fs(filename.c_str(), std::ios::binary);
        fs << "ply\nformat binary_little_endian 1.0\nelement vertex " << valid
           << "\nproperty float x\nproperty float y\nproperty float z"
              "\nproperty uchar red\nproperty uchar green\nproperty uchar blue"
              "\nproperty float nx\nproperty float ny\nproperty float nz"
              "\nproperty float curvature_max\nproperty float curvature_min"
              "\nproperty float radius\nproperty float submapIndex\nend_header\n";
        for (unsigned int i = 0; i < n; ++i) {
            const float *s = &map[(size_t)i * 20];
            if (!(s[3] > confThreshold)) continue;
            fs.write((const char *)&s[0], 12);
            const int c = (int)s[4];
            const unsigned char rgb[3] = {(unsigned char)((c >> 16) & 0xFF), (unsigned char)((c >> 8) & 0xFF),
                                          (unsigned char)(c & 0xFF)};
            fs.write((const char *)rgb, 3);
            const float nrm[3] = {-s[8], -s[9], -s[10]};
            fs.write((const char *)nrm, 12);
            fs.write((const char *)&s[15], 4);   /* curvature_max = curv_map_max.w */
            fs.write((const char *)&s[19], 4);   /* curvature_min = curv_map_min.w */
            fs.write((const char *)&s[11], 4);   /* radius */
            fs.write((const char *)&s[5], 4);    /* submapIndex */
        }
        delete[] map;
    }

    TrajectoryManager *trajectory_manager;   /* public like HRBFFusion.h:383-384 */

private:
    void replayPose(int tick)
    {
        if (!load_trajectory_ || replay_.empty() || tick <= 1) return;
        if ((size_t)(tick - 1) >= replay_.size())
            throw std::runtime_error("globalInputLoadTrajectory: the trajectory file has fewer poses than frames");
        setPose(replay_[(size_t)(tick - 1)].m);
    }
    std::vector<Pose> replay_;   // poses of globalInputTrajectoryFile when the trajectory is replayed
    hrbf_handle h_;
    GlobalModel *model_;
    IndexMap *index_;
    Pose curr_;
    RGBDOdometry frame_to_model_;
    bool icp_cached_ = false; unsigned int icp_cached_at_ = 0;
    bool load_trajectory_;
    std::vector<bool> pushes_;   // per processed frame: does it contribute to trajectory_manager->poses
    uint32_t synced_;            // frames already folded into trajectory_manager->poses
};

}  // namespace hrbf_mi355

/* The reference's names at global scope, so that its caller compiles with only the #include changed.  Define
   HRBF_MI355_NO_GLOBAL_NAMES to keep them inside the namespace (e.g. when linking next to the reference itself). */
#ifndef HRBF_MI355_NO_GLOBAL_NAMES
using hrbf_mi355::HRBFFusion;
using hrbf_mi355::GlobalStateParam;
using hrbf_mi355::Intrinsics;
using hrbf_mi355::ParameterFile;
using hrbf_mi355::Resolution;
using hrbf_mi355::RGBDOdometry;
#endif
#endif
