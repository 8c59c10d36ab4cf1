// The reference's caller, against this library with only the #include changed.
//
// The statements between the BEGIN/END marks are the library-facing statements of MainController's constructor and run()
// (GUI/src/HRBF_fusion.cpp:35-54,87-100,174-181,190-239,283-297,470-497) in their order and spelling: parameter file ->
// GlobalStateParam, camera file -> Resolution / Intrinsics, the defaults read back from GlobalStateParam, `new HRBFFusion(...)`
// with the reference's eight arguments, the start / skip / processFrame sequence, the getters the GUI polls, the export
// calls.  What is NOT the reference's text: OpenCV's FileStorage (absent here) is replaced by hrbf_mi355::CameraFile, the log
// reader by a two-frame synthetic source, Eigen::Matrix4f by the 16 floats getCurrPose() points at, and the GUI is gone.
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "HRBFFusion.h"      // reference: #include <HRBFFusion.h> (Core/src/HRBFFusion.h)

struct SyntheticLogReader {   // stands in for RawImageLogReader: rgb / depth / timestamp of the current frame
    std::vector<unsigned char> rgbv; std::vector<unsigned short> depthv;
    unsigned char *rgb; unsigned short *depth; int64_t timestamp; int currentFrame, n, W, H;
    SyntheticLogReader(int w, int h, int frames) : rgbv((size_t)w * h * 3), depthv((size_t)w * h), timestamp(0), currentFrame(0), n(frames), W(w), H(h)
    { rgb = rgbv.data(); depth = depthv.data(); }
    bool hasMore() const { return currentFrame < n; }
    void getNext()
    {
        for (int i = 0; i < W * H; ++i) {
            const int x = i % W, y = i / W;
            rgbv[(size_t)i * 3] = (unsigned char)(40 + ((x / 8 + y / 8) * 37) % 150); rgbv[(size_t)i * 3 + 1] = 90; rgbv[(size_t)i * 3 + 2] = 120;
            depthv[i] = (unsigned short)(7000 + x * 2 + y);   // a slanted plane, 1.4 - 1.7 m at DepthMapFactor 5000
        }
        timestamp = 33333 * (int64_t)currentFrame;
        currentFrame++;
    }
    void fastForward(int frame) { currentFrame = frame; }
};

// stands in for the Pangolin widgets the polled lines write to (GUI/src/Tools/GUI.h): a text variable, a button, a data log
struct TextVar { std::string s; TextVar &Ref() { return *this; } void Set(const std::string &v) { s = v; } };
struct Button { bool on; bool Get() const { return on; } };
struct DataLog { std::vector<float> v, t; void Log(float a, float b) { v.push_back(a); t.push_back(b); } };
struct GuiStandIn {
    TextVar inliers_, res_, *trackInliers, *trackRes; Button start_, *start; DataLog resLog, inLog;
    GuiStandIn() : trackInliers(&inliers_), trackRes(&res_), start_{true}, start(&start_) {}
};

int main(int argc, char **argv)
{
    if (argc < 3) { printf("usage: %s <GlobalStateParam.txt> <outdir>\n", argv[0]); return 2; }
    const std::string outdir = argv[2];
    float confidence, depth, icp, icpErrThresh, covThresh, photoThresh;
    int framesToSkip, timeDelta, icpCountThresh, start, end;
    bool so3, fastOdom = false, frameToFrameRGB = false;
    HRBFFusion *hrbfFusion = nullptr;
    try {
    // ---- BEGIN: MainController::MainController (GUI/src/HRBF_fusion.cpp:35-54,87-100,174-181) -------------------------
    //load application parameter from file
    ParameterFile pf(argv[1]);

    //set parameter to global variable
    GlobalStateParam::getInstance().readMembers(pf);

    //Load camera parameters from open CV settings file and set global variables
    hrbf_mi355::CameraFile fSettings = hrbf_mi355::CameraFile::fromFile(GlobalStateParam::get().parameterFileCvFormat);
    float fx = fSettings.fx;
    float fy = fSettings.fy;
    float cx = fSettings.cx;
    float cy = fSettings.cy;
    int width = fSettings.width;
    int height = fSettings.height;
    Resolution::getInstance(width, height);
    Intrinsics::getInstance(fx, fy, cx, cy);

    //default parameter settings
    confidence = GlobalStateParam::get().globalConfidenceThreshold;
    depth = GlobalStateParam::get().globalDepthCutoff;
    icp = GlobalStateParam::get().registrationJointICPWeight;
    icpErrThresh = GlobalStateParam::get().registrationICPErrorThreshold;
    covThresh = GlobalStateParam::get().registrationICPCovarianceThreshold;
    photoThresh = GlobalStateParam::get().registrationColorPhotoThreshold;
    framesToSkip = GlobalStateParam::get().globalFrameToSkip;
    timeDelta = 200;
    icpCountThresh = 40000;
    start = GlobalStateParam::get().globalStartFrame;
    so3 = GlobalStateParam::get().registrationPreAlignSO3;
    end = std::numeric_limits<unsigned short>::max();     //Funny bound, since we predict times in this format really!
    if(GlobalStateParam::get().globalEndFrame > 0)
        end = GlobalStateParam::get().globalEndFrame;

    hrbfFusion = new HRBFFusion(icpCountThresh,
                                icpErrThresh,
                                confidence,
                                depth,
                                icp,
                                fastOdom,
                                so3,
                                frameToFrameRGB);
    // ---- END ---------------------------------------------------------------------------------------------------------------
    (void)covThresh; (void)photoThresh; (void)timeDelta;
    printf("constructed: %d x %d, tick %d\n", Resolution::getInstance().width(), Resolution::getInstance().height(), hrbfFusion->getTick());
    SyntheticLogReader reader(width, height, 4), *logReader = &reader;
    // ---- BEGIN: MainController::run (GUI/src/HRBF_fusion.cpp:190-239) -------------------------------------------------------
    while(logReader->hasMore() && hrbfFusion->getTick() < end)
    {
                logReader->getNext();

                if(hrbfFusion->getTick() < start)
                {
                    hrbfFusion->setTick(start);
                    logReader->fastForward(start);
                    logReader->getNext();
                }

                float weightMultiplier = framesToSkip + 1;

                if(framesToSkip > 0)
                {
                    hrbfFusion->setTick(hrbfFusion->getTick() + framesToSkip);
                    logReader->fastForward(logReader->currentFrame + framesToSkip);
                    framesToSkip = 0;
                }

                hrbfFusion->processFrame(logReader->rgb, logReader->depth, logReader->timestamp, weightMultiplier);
    }
    // ---- END ---------------------------------------------------------------------------------------------------------------
    // what the GUI polls and exports (GUI/src/HRBF_fusion.cpp:235-239,448-456,470-497)
    hrbfFusion->setRgbOnly(false);
    hrbfFusion->setPyramid(true);
    hrbfFusion->setFastOdom(fastOdom);
    hrbfFusion->setConfidenceThreshold(confidence);
    hrbfFusion->setDepthCutoff(depth);
    hrbfFusion->setIcpWeight(icp);
    hrbfFusion->setSo3(so3);
    hrbfFusion->setFrameToFrameRGB(frameToFrameRGB);
    GuiStandIn gui_, *gui = &gui_;
    bool aStep = false;
    // ---- BEGIN: MainController::run, the tracking read-outs (GUI/src/HRBF_fusion.cpp:283-297) ----------------------------------
        //Tracking inliers in histgram
        std::stringstream stri;
        stri << hrbfFusion->getFrameToModel().lastICPCount;
        gui->trackInliers->Ref().Set(stri.str());
        //Tracking ICP error in histgram.
        std::stringstream stre;

        stre << (std::isnan(hrbfFusion->getFrameToModel().lastICPError) ? 0 : hrbfFusion->getFrameToModel().lastICPError);
        gui->trackRes->Ref().Set(stre.str());

        if(gui->start->Get()|| aStep) {
            gui->resLog.Log((std::isnan(hrbfFusion->getFrameToModel().lastICPError) ? std::numeric_limits<float>::max() : hrbfFusion->getFrameToModel().lastICPError), icpErrThresh);
            gui->inLog.Log(hrbfFusion->getFrameToModel().lastICPCount, icpCountThresh);
            aStep = false;         
        }
    // ---- END ---------------------------------------------------------------------------------------------------------------
    printf("read-outs: inliers '%s' error '%s' logged %g / %g\n", gui->inliers_.s.c_str(), gui->res_.s.c_str(), gui->resLog.v[0], gui->inLog.v[0]);
    /* the synthetic source repeats one frame: every pixel is an inlier and the residual is zero */
    if (!(gui->inLog.v[0] > 1000.0f) || !(gui->resLog.v[0] >= 0.0f && gui->resLog.v[0] < 1e-2f) || gui->inLog.v[0] != hrbfFusion->lastICPCount()) return 22;
    const float *currPose = hrbfFusion->getCurrPoseData();
    printf("tick %d surfels %u icp error %g count %g pose t = %g %g %g\n", hrbfFusion->getTick(), hrbfFusion->getGlobalModel().lastCount(),
           hrbfFusion->lastICPError(), hrbfFusion->lastICPCount(), currPose[12], currPose[13], currPose[14]);
    hrbfFusion->savePly(outdir + "/ref_caller.ply", GlobalStateParam::get().globalOutputSavePointCloudConfThreshold);
    hrbfFusion->trajectory_manager->SaveTrajectoryToFile(GlobalStateParam::get().globalOutputSaveTrjectoryFileType, outdir + "/ref_caller.freiburg",
                                                         GlobalStateParam::get().globalInputICLNUIMDataset);
    if (hrbfFusion->getTick() != 5 || hrbfFusion->getGlobalModel().lastCount() == 0) return 20;
    delete hrbfFusion;
    printf("GPU-OK\n");
    } catch (const std::runtime_error &e) {
        printf("threw: %s\n", e.what());
        if (argc > 3 && !strcmp(argv[3], "gpu")) return 21;
        printf("NOGPU-OK\n");
    }
    return 0;
}
