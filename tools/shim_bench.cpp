// shim_bench.cpp — frames/s through the C++ class (include/HRBFFusion.h), the way the reference's caller drives it
// (GUI/src/HRBF_fusion.cpp:235: hrbfFusion->processFrame(logReader->rgb, logReader->depth, logReader->timestamp, ...)).
//
//   shim_bench <frames.bin> <map.bin|-> <W> <H> <fx> <fy> <cx> <cy> <warmup> <steps>
//
// frames.bin: N x { W*H*3 bytes rgb, W*H uint16 depth }, host memory (what a LogReader hands over);
// map.bin   : optional AoS surfel map (20 floats per surfel) uploaded before the run ("-" = start from an empty map),
//             followed in the file by one 16-float column-major pose for the first frame.
// Prints one JSON object: frames/s of processFrame(host pointers) with NO per-frame synchronisation (the trajectory is
// read once at the end from the device-written ring), and the same loop with a getCurrPose() after every frame (what
// a GUI that draws the camera every frame pays).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "HRBFFusion.h"

using namespace hrbf_mi355;

static std::vector<unsigned char> slurp(const char *path)
{
    std::vector<unsigned char> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 11) { fprintf(stderr, "usage: %s frames.bin map.bin|- W H fx fy cx cy warmup steps\n", argv[0]); return 2; }
    const int W = atoi(argv[3]), H = atoi(argv[4]);
    const float fx = (float)atof(argv[5]), fy = (float)atof(argv[6]), cx = (float)atof(argv[7]), cy = (float)atof(argv[8]);
    const int warm = atoi(argv[9]), steps = atoi(argv[10]);
    const size_t frame_bytes = (size_t)W * H * 5;
    std::vector<unsigned char> frames = slurp(argv[1]);
    const int nf = (int)(frames.size() / frame_bytes);
    if (nf < 2 + warm + steps) { fprintf(stderr, "frames.bin holds %d frames, need %d\n", nf, 2 + warm + steps); return 3; }
    std::vector<unsigned char> mapfile;
    if (strcmp(argv[2], "-") != 0) mapfile = slurp(argv[2]);
    const size_t nsurf = mapfile.size() >= 64 ? (mapfile.size() - 64) / 80 : 0;
    try {
        auto run = [&](bool pose_every_frame) -> double {
            HRBFFusion f(W, H, fx, fy, cx, cy, 1.0f / 5000.f, 40000, 5e-5f, 5.0f, 3.5f, 10.f, false, true, false,
                         (int)(nsurf + (size_t)(W / 2) * (H / 2) * (size_t)(4 + warm + steps) + 200000));
            auto rgb = [&](int k) { return frames.data() + (size_t)k * frame_bytes; };
            auto dep = [&](int k) { return (const unsigned short *)(frames.data() + (size_t)k * frame_bytes + (size_t)W * H * 3); };
            int k = 0;
            if (nsurf) {
                if (hrbf_upload_map(f.handle(), (const float *)mapfile.data(), nsurf) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
                f.setPose((const float *)(mapfile.data() + nsurf * 80));
                if (hrbf_bootstrap(f.handle(), rgb(0), dep(0)) != HRBF_OK) throw std::runtime_error(hrbf_last_error());
                k = 1;
            }
            for (int i = 0; i < warm; ++i, ++k) f.processFrame(rgb(k), dep(k), (int64_t)k * 33333);
            f.synchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < steps; ++i, ++k) {
                f.processFrame(rgb(k), dep(k), (int64_t)k * 33333);
                if (pose_every_frame) (void)f.getCurrPoseData();
            }
            const size_t np = f.getTrajectory().size();     // one synchronisation for the whole run
            const auto t1 = std::chrono::steady_clock::now();
            if (np == 0) throw std::runtime_error("empty trajectory");
            return steps / std::chrono::duration<double>(t1 - t0).count();
        };
        const double fps_async = run(false);
        const double fps_sync = run(true);
        printf("{\"cpp_shim_fps\": %.2f, \"cpp_shim_fps_pose_every_frame\": %.2f, \"steps\": %d, \"warmup\": %d, \"surfels\": %zu}\n",
               fps_async, fps_sync, steps, warm, nsurf);
    } catch (const std::runtime_error &e) {
        fprintf(stderr, "shim_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
