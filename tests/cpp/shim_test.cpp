// Compiles the header-only HRBFFusion shim with a plain host compiler and exercises the parts that
// need no GPU: trajectory writers, quaternion conversion, loud failure of the constructor.
// With a GPU (argv[1] = "gpu") it also runs processFrame on a synthetic plane and writes a PLY.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "HRBFFusion.h"

using namespace hrbf_mi355;

int main(int argc, char **argv)
{
    TrajectoryManager tm;
    Pose I; memset(&I, 0, sizeof(I)); I.m[0] = I.m[5] = I.m[10] = I.m[15] = 1.0f;
    Pose R = I;   // 90 deg about z, t = (1,2,3)
    R.m[0] = 0; R.m[1] = 1; R.m[4] = -1; R.m[5] = 0; R.m[12] = 1; R.m[13] = 2; R.m[14] = 3;
    tm.poses.push_back(I); tm.poses.push_back(R);
    tm.timstamp.push_back(1000000); tm.timstamp.push_back(2500000);
    const char *dir = argc > 2 ? argv[2] : "/tmp";
    std::string base(dir);
    if (!tm.SaveTrajectoryToFile("TUM", base + "/t.freiburg")) return 10;
    if (!tm.SaveTrajectoryToFile("TUM", base + "/t_icl.freiburg", true)) return 11;
    if (!tm.SaveTrajectoryToFile("zhou", base + "/t.log")) return 12;
    if (!tm.SaveTrajectoryToFile("lefloch", base + "/t_lef.txt")) return 13;
    float q[4]; TrajectoryManager::quaternion(R.m, q);
    if (std::fabs(q[2] - std::sqrt(0.5f)) > 1e-6f || std::fabs(q[3] - std::sqrt(0.5f)) > 1e-6f || q[0] != 0 || q[1] != 0) return 14;
    const bool want_gpu = argc > 1 && !strcmp(argv[1], "gpu");
    try {
        HRBFFusion f(160, 120, 132.f, 132.f, 80.f, 60.f, 1.0f / 5000.f, 35000, 5e-5f, 5.0f, 3.5f, 10.f, false, true, false, 1 << 16);
        if (!want_gpu) { printf("constructed with a GPU present\n"); }
        std::vector<unsigned char> rgb(160 * 120 * 3, 128);
        std::vector<unsigned short> d(160 * 120, 7500);
        for (int i = 0; i < 160 * 120; ++i) rgb[i * 3] = (unsigned char)(40 + (i * 7) % 150);
        f.processFrame(rgb.data(), d.data(), 0);
        f.processFrame(rgb.data(), d.data(), 33333);
        printf("tick %d count %u pose t = %g %g %g\n", f.getTick(), f.getGlobalModel().lastCount(), f.getCurrPoseData()[12],
               f.getCurrPoseData()[13], f.getCurrPoseData()[14]);
        if (f.getTick() != 3 || f.getGlobalModel().lastCount() == 0) return 20;
        // the trajectory is pulled lazily from the device-written ring: nothing was synchronised by processFrame itself
        if (f.getTrajectory().size() != 2 || f.trajectory_manager->timstamp.size() != 1) return 22;
        if (memcmp(f.getTrajectory()[1].m, f.getCurrPoseData(), sizeof(float) * 16) != 0) return 23;
        if (f.framesCompleted() != 2) return 24;
        // replayed trajectory (globalInputLoadTrajectory): the frame is processed at the given pose and pushes nothing
        f.setLoadTrajectory(true);
        f.setPose(f.getCurrPoseData());
        f.processFrame(rgb.data(), d.data(), 66666);
        if (f.getTrajectory().size() != 2 || f.trajectory_manager->timstamp.size() != 1 || f.framesCompleted() != 3) return 25;
        f.setLoadTrajectory(false);
        f.savePly(base + "/m.ply");
        f.trajectory_manager->SaveTrajectoryToFile("TUM", base + "/run.freiburg");
        printf("GPU-OK\n");
    } catch (const std::runtime_error &e) {
        printf("ctor threw: %s\n", e.what());
        if (want_gpu) return 21;
        printf("NOGPU-OK\n");
    }
    return 0;
}
