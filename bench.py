#!/usr/bin/env python3
"""bench.py — frames/sec of the HRBF-Fusion per-frame hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one hrbf_process_frame over one synthetic 640x480 RGB-D frame (pre-processing, SO3 +
3-level joint ICP/RGB registration, index-map projection x3, data association + merge, clean +
compact + append, HRBF ray-cast prediction + fill-in) against a map pre-seeded to >= 1 M surfels.
Inputs are resident in HBM before the timed region (torch tensors; PyTorch is plumbing only).

N > 1: one process per GPU, each an independent replica of the same sequence (SURVEY.md §8e last row: the
per-frame path of one sequence does not shard below VGA); no data-path collective, "scaling": "weak",
value = frames all ranks processed / max-over-ranks time.  Started by hand (`python bench.py --gpus N`, no
WORLD_SIZE in the environment) it re-executes itself under `python -m torch.distributed.run`, one rank per GPU; under
the driver's launcher it checks WORLD_SIZE == N.  The line carries the rank count RCCL itself saw (`ranks_observed`).
At N > 1 a second leg times the SHARDED design — all ranks on ONE sequence, the surfel map owned by
spatial hash, registration row-sharded, RCCL all-reduce of the 6 x 6 limb sums: BASELINE config 4 (640 x 480, 4.3 M
surfels) at N = 2..7, config 5 (1280 x 960, 8.8 M) at N = 8 — in child processes with a time limit, so that nothing
there can take the replica line down (`sharded_one_sequence`, "scaling": "strong", per-rank fuse-pass times).

One JSON line on rank 0, with `roofline` (fuse streaming kernel, HIP events on the library's own
stream) and `cpu_baseline` (the CPU oracle on a bounded sample; baseline, not target).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed frames (SURVEY §8d: 200 measured frames after 20 warm-up)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--surfels", type=int, default=1_050_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--noise", action="store_true", help="Kinect-style depth noise + 3%% drop-outs")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the CPU-oracle sample (OpenMP), ~10 s of host time (0 = skip)")
    ap.add_argument("--cpu-frames-1t", type=int, default=2, help="frames of the single-thread CPU-oracle sample (~10 s; 0 = skip)")
    ap.add_argument("--big-surfels", type=int, default=8_800_000,
                    help="third leg (rank 0, N=1): BASELINE config 5's largest single-GPU shape, a 1280x960 stream against a map of this "
                         "many surfels (0 = skip)")
    ap.add_argument("--big-frames", type=int, default=10)
    ap.add_argument("--shard-odometry", action="store_true",
                    help="N > 1: all ranks track ONE sequence, registration reductions row-sharded + RCCL all-reduce "
                         "(SURVEY §8e sharding 1; strong scaling, a latency cost at VGA). Default: independent replicas")
    ap.add_argument("--shard-map", action="store_true",
                    help="all ranks track ONE sequence against ONE surfel map cut into contiguous ranges over the ranks "
                         "(SURVEY §8e sharding 2; add --shard-odometry for the row-sharded registration too; strong scaling; pays off "
                         "for maps far beyond 1 M surfels). Default: independent replicas")
    ap.add_argument("--partition", choices=["ranges", "hash"], default="ranges",
                    help="ownership of a sharded map: contiguous ranges of the global order, or spatial hash of the surfel's cell "
                         "(SURVEY §8e; the surfels in view then spread over the ranks)")
    ap.add_argument("--virtual-shards", type=int, default=0,
                    help="single process: play G map shards in turn on one GPU (measures the sharded path's extra "
                         "kernels without any interconnect); not a benchmark configuration")
    ap.add_argument("--worst-surfels", type=int, default=4_300_000,
                    help="second roofline leg (rank 0, N=1): a map of this many surfels (> the 256 MiB Infinity Cache) with "
                         "--worst-frac of them unstable and stale, spread from index 0, so that the in-place compaction "
                         "really moves the whole map (0 = skip)")
    ap.add_argument("--worst-frac", type=float, default=0.05)
    ap.add_argument("--worst-samples", type=int, default=5)
    ap.add_argument("--only-worst", action="store_true", help="run only the worst-case fuse leg (profiling)")
    ap.add_argument("--no-cpp-shim", action="store_true", help="skip the C++-class leg (g++ compile + run of tools/shim_bench.cpp)")
    ap.add_argument("--ring-stride", type=int, default=4,
                    help="bracket the fuse pass with HIP events on every n-th frame of the timed region (each bracketed frame costs the stream ~22 us)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 --pmc passes that measure the fuse pass's HBM traffic")
    ap.add_argument("--no-fit-leg", action="store_true", help="skip the extension leg (true Hermite-RBF fit on the matrix core)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="N > 1: skip the one-sequence sharded leg (child processes)")
    ap.add_argument("--sharded-leg-timeout", type=float, default=420.0, help="seconds the sharded leg's child processes may take")
    ap.add_argument("--one-sequence-child", action="store_true", help=argparse.SUPPRESS)   # set by the parent for the sharded leg
    ap.add_argument("--dataset", default=None,
                    help="run a RECORDED sequence instead of the synthetic stream: a directory in the TUM RGB-D layout (rgb/, depth/, "
                         "rgb.txt + depth.txt or associations.txt, groundtruth.txt) or the ICL-NUIM TUM-compatible PNG package, or a root "
                         "that holds `rgbd_dataset_freiburg1_desk/` / `living_room_traj2_frei_png/` (see --sequence); BASELINE configs "
                         "2 / 3, with the reference's GUI settings (hrbffusion3d_amd/datasets.py); same line shape, data: \"real\"")
    ap.add_argument("--sequence", choices=["icl_nuim_lr_kt2", "tum_fr1_desk"], default=None, help="with --dataset ROOT: which sequence")
    ap.add_argument("--dataset-kind", choices=["tum", "icl"], default=None, help="with --dataset DIR: override the layout detection")
    ap.add_argument("--dataset-label", default=None, help=argparse.SUPPRESS)    # tests: say what the files really are
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: start the ranks (gloo), run the barrier / max-over-ranks / rank-count plumbing and the child-leg "
                         "launch, print the line's skeleton (tests/test_multigpu_gloo.py)")
    return ap.parse_args()


REGIONS = ("Initialization", "Registration", "Integration", "Prediction")   # HRBFFusion.cpp:1016,1063,1196,1248


def _cpu_run(args, seed, frames, poses, n, omp):
    from oracle_lib import Oracle
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.params import default_params
    fx, fy, cx, cy = synth.intrinsics(args.width, args.height)
    p = default_params(args.width, args.height, fx, fy, cx, cy, max_surfels=int(seed.shape[0] + 200_000 + 80_000 * (n + 1)))
    o = Oracle(p, omp=omp)
    o.upload_map(seed)
    o.set_pose(poses[0])
    o.bootstrap(frames[0][0], frames[0][1])
    reg = np.zeros(4)
    traj = []
    t0 = time.perf_counter()
    for k in range(1, n + 1):
        o.process_frame(frames[k][0], frames[k][1])
        reg += o.timings()[:4]
        traj.append(o.get_pose())       # a 64-byte copy out of the context
    dt = time.perf_counter() - t0
    o.close()
    return n / dt, {r: float(v / n) for r, v in zip(REGIONS, reg)}, traj


def cpu_baseline(args, seed, frames, poses, gpu_traj=None):
    """the CPU oracle on the same workload, (a) single thread, (b) OpenMP over pixels / surfels on ALL host cores of
    this box (count stated), ms per frame in the reference's four Stopwatch regions.  Bounded samples: bootstrap + n frames each.
    The oracle's loops are over pixels / surfels of one frame: beyond a few dozen threads the fork/join of its ~60 parallel regions
    per frame outweighs the work, so the all-core run is reported NEXT TO a 32-thread run and `value` is the faster of the two
    (`cores` = the threads that run used).  gpu_traj: the HIP path's poses of the same frames -> `ate_vs_oracle_mm` (the
    bit-parity claim where the driver can see it: must be 0.0)."""
    cores = len(os.sched_getaffinity(0))
    n = args.cpu_frames
    os.environ["OMP_NUM_THREADS"] = str(cores)     # read by libgomp when the OpenMP build is first loaded
    sweep = []
    traj = None
    gomp = None
    for threads in sorted(set([cores, min(cores, 32)]), reverse=True):
        if gomp is not None:
            gomp.omp_set_num_threads(threads)
        fps, reg, traj = _cpu_run(args, seed, frames, poses, n, True)
        sweep.append({"threads": threads, "value": fps, "region_ms": reg})
        if gomp is None:
            gomp = C.CDLL("libgomp.so.1")      # loaded by the OpenMP build of the oracle by now
    best = max(sweep, key=lambda r: r["value"])
    out = {"value": best["value"], "unit": "frames/s", "cores": best["threads"], "kind": "port", "host_cores": cores,
           "sample": "%d frames of the same %dx%d stream against the same %d-surfel map (oracle, OpenMP; all %d host cores and "
                     "32 threads timed, the faster is `value`)" % (n, args.width, args.height, seed.shape[0], cores),
           "region_ms": best["region_ms"], "threads_sweep": sweep}
    if gpu_traj is not None and traj:
        m = min(len(traj), len(gpu_traj))
        e = np.asarray([t[:3, 3] for t in traj[:m]], np.float64) - np.asarray([t[:3, 3] for t in gpu_traj[:m]], np.float64)
        out["ate_vs_oracle_mm"] = float(1000.0 * np.sqrt((e ** 2).sum(1).mean()))
        out["ate_vs_oracle_frames"] = int(m)
        out["poses_bit_identical"] = bool(all(np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
                                              for a, b in zip(traj[:m], gpu_traj[:m])))
    if args.cpu_frames_1t > 0:
        fps1, reg1, _ = _cpu_run(args, seed, frames, poses, args.cpu_frames_1t, False)
        out["single_thread"] = {"value": fps1, "unit": "frames/s", "cores": 1, "region_ms": reg1,
                                "sample": "%d frames, same workload, scalar build of the oracle" % args.cpu_frames_1t}
    return out


def big_leg(args, local_rank):
    """BASELINE config 5's largest single-GPU shape: 1280x960 frames against an 8.8 M-surfel map on ONE GPU (the 8-GPU split of
    that config is the driver's to run): ms per full processFrame, inputs resident in HBM."""
    import torch
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import default_params
    W, H = 1280, 960
    K = synth.intrinsics(W, H)
    seed = synth.seed_map(args.big_surfels, t_now=1, width=W)
    warm, n = 3, args.big_frames
    fr = _frames(range(1 + warm + n), W, H, False)
    p = default_params(W, H, *K, max_surfels=int(seed.shape[0] + (W // 2) * (H // 2) * (warm + n + 8)))
    fus = HRBFFusion(p, device=local_rank)
    fus.upload_map(seed); fus.set_pose(fr[0][2]); fus.bootstrap(fr[0][0], fr[0][1])
    d = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1].view(np.int16)).cuda()) for f in fr]
    for k in range(1, 1 + warm):
        fus.process_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), k)
    fus.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + warm, 1 + warm + n):
        fus.process_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), k)
    fus.synchronize()
    dt = time.perf_counter() - t0
    fus.enable_timing(1)
    fus.process_frame_device(d[warm + n][0].data_ptr(), d[warm + n][1].data_ptr(), warm + n)
    tm = fus.timings()
    T = fus.get_pose()
    out = {"workload": "synthetic %dx%d stream, map pre-seeded to %d surfels, one GPU" % (W, H, seed.shape[0]), "frames": n,
           "ms_per_frame": 1000.0 * dt / n, "frames_per_s": n / dt, "surfels_end": fus.surfel_count(), "status": fus.status(),
           "final_translation_error_mm": float(1000.0 * np.linalg.norm(T[:3, 3] - fr[warm + n][2][:3, 3])),
           "last_frame_region_ms": {r: float(v) for r, v in zip(REGIONS + ("fuse_stream_pass",), tm[:5])}}
    fus.close()
    return out


def _one_frame(a):
    from hrbffusion3d_amd import synth
    return synth.frame(a[0], a[1], a[2], noise=a[3])


def _frames(ks, W, H, noise):
    """synthetic frames (rgb, depth, pose); generated by a pool of forked workers — numpy only, 0.26 s per VGA frame"""
    ks = list(ks)
    nproc = int(os.environ.get("HRBF_BENCH_GEN_PROCS", "0")) or max(1, min(len(os.sched_getaffinity(0)), 16, len(ks)))
    if nproc > 1:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(nproc)
        try:    # a forked worker can hang under a profiler's preloaded tool (seen once under rocprofv3): fall back, never stall
            out = pool.map_async(_one_frame, [(k, W, H, noise) for k in ks], chunksize=max(1, len(ks) // (4 * nproc))).get(timeout=180)
            pool.close(); pool.join()
            return out
        except Exception:
            pool.terminate()
    return [_one_frame((k, W, H, noise)) for k in ks]


def cpp_shim_leg(args, seed, frames, poses):
    """frames/s through the C++ class (include/HRBFFusion.h -> tools/shim_bench.cpp, compiled here with g++): the
    reference's call site, HRBFFusion::processFrame(host rgb, host depth, timestamp) per frame, nothing else"""
    import subprocess
    import tempfile
    from hrbffusion3d_amd import build as hb
    so = hb.build()
    warm, steps = 5, min(40, len(frames) - 7)
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "shim_bench")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "shim_bench.cpp"), "-o", exe, so,
                               "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
        with open(os.path.join(tmp, "frames.bin"), "wb") as f:
            for rgb, d in frames[:2 + warm + steps]:
                f.write(np.ascontiguousarray(rgb).tobytes()); f.write(np.ascontiguousarray(d).tobytes())
        with open(os.path.join(tmp, "map.bin"), "wb") as f:
            f.write(np.ascontiguousarray(seed, np.float32).tobytes())
            f.write(np.ascontiguousarray(np.asarray(poses[0], np.float32).T).tobytes())
        fx, fy, cx, cy = (str(v) for v in __import__("hrbffusion3d_amd.synth", fromlist=["x"]).intrinsics(args.width, args.height))
        out = subprocess.run([exe, os.path.join(tmp, "frames.bin"), os.path.join(tmp, "map.bin"), str(args.width),
                              str(args.height), fx, fy, cx, cy, str(warm), str(steps)], capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    return json.loads(out.stdout.strip().split("\n")[-1])


def frustum_counts(m, T_wc, K, W, H, conf_thr, max_depth=20.0):
    """(in view, out of view but unstable) surfel counts of AoS map m for the pose T_wc: what decides how many bytes
    pass A of the fuse really reads (16 B per surfel + 16 B colour/time for in-view or unstable + 16 B normal/radius
    for in-view ones)"""
    fx, fy, cx, cy = K
    Ti = np.linalg.inv(T_wc.astype(np.float64))
    pc = m[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
    z = pc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = fx * pc[:, 0] / z + cx; v = fy * pc[:, 1] / z + cy
    inv = (z > 0) & (z < max_depth) & (u > 0) & (v > 0) & (u < W) & (v < H)
    return int(inv.sum()), int((~inv & (m[:, 3] < conf_thr)).sum())


def fuse_real_bytes(n_in, n_view, n_unst_out, merged, appended, moved, Q, P, full_check=False):
    """MODELLED HBM/L2 bytes the three fuse kernels touch per frame (the measured figure is the PMC one in profiles/):
    F2 k_apply_merges: every record lane reads its 80-B record + flag + best (8 B); a winning record reads and rewrites
       its surfel (160 B) and its slot word;
    pass A k_clean_flags: 1 class byte per surfel (left by the projection in front of the pass), 32 B position +
       colour/time if in view or unstable, +16 normal/radius if in view, +32 curvature if merged; with full_check (after a
       map upload) 16 B per surfel and no class; 80 B + 4 B per record, the 16-B clean texel image once (L2-resident
       afterwards), 1 keep byte per item;
    pass B k_fuse_stream: 1 keep byte per item of the moving tiles, 160 B per surfel that changes slot, 160 B per
       appended record."""
    f2 = Q * (80 + 8) + merged * (160 + 8)
    a_surf = (n_in * 16 + (n_view + n_unst_out) * 16 + n_view * 16 + n_in * 32) if full_check else \
             (n_in * 1 + (n_view + n_unst_out) * 32 + n_view * 16 + merged * 32)
    a = a_surf + Q * 84 + P * 16 + (n_in + Q)
    b = (moved + Q) * 1 + moved * 160 + appended * 160
    return float(f2 + a + b)


def worst_case_leg(args, local_rank):
    """The fuse pass when it cannot hide: a map larger than the 256 MiB Infinity Cache whose first tile already loses
    surfels, so every survivor is re-read and written `shift` slots to the left (copy_unstable.vert:159-165 removes
    unstable surfels not seen for 200 frames).  Per sample: upload the seed, one frame at tick 2 (takes the
    after-upload full curvature check and the merges), jump the clock by 300 frames, one frame = the measured one."""
    import torch
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import default_params
    W, H = args.width, args.height
    K = synth.intrinsics(W, H)
    seed = synth.seed_map(args.worst_surfels, t_now=1, width=W)
    n = seed.shape[0]
    rng = np.random.default_rng(99)
    stale = np.zeros(n, bool)
    stale[rng.choice(n, int(args.worst_frac * n), replace=False)] = True
    stale[0:64:3] = True                      # removals start in the very first tile
    seed[stale, 3] = 1.0                      # unstable: confidence below the threshold (5)
    Q = (W // 2) * (H // 2)
    p = default_params(W, H, *K, max_surfels=n + 8 * Q)
    fus = HRBFFusion(p, device=local_rank)
    f0, f1, f2 = (synth.frame(k, W, H) for k in range(3))
    d1 = (torch.from_numpy(f1[0]).cuda(), torch.from_numpy(f1[1].view(np.int16)).cuda())
    d2 = (torch.from_numpy(f2[0]).cuda(), torch.from_numpy(f2[1].view(np.int16)).cuda())
    fus.enable_timing(2)
    rows = []
    for it in range(args.worst_samples + 1):   # sample 0 is a warm-up (first-touch of every buffer)
        fus.upload_map(seed); fus.set_pose(f0[2]); fus.bootstrap(f0[0], f0[1])
        fus.process_frame_device(d1[0].data_ptr(), d1[1].data_ptr(), 1)
        fus.synchronize()
        fus.set_tick(300)
        fus.reset_fuse_ring()
        fus.process_frame_device(d2[0].data_ptr(), d2[1].data_ptr(), 2)
        fus.synchronize()
        mm, ms, st = fus.fuse_ring_parts(1)
        if it > 0:
            rows.append((float(mm[0]), float(ms[0]), st[0].copy()))
    m_end = fus.download_map()
    n_view, n_unst_out = frustum_counts(seed, f2[2], K, W, H, p.confidence_threshold)
    status = fus.status()
    fus.close()
    mm = np.array([r[0] for r in rows]); ms = np.array([r[1] for r in rows]); st = rows[-1][2]
    n_in, merged, appended, n_out, moved = int(st[0]), int(st[1]), int(st[2]), int(st[3]), int(st[6])
    t = float((mm + ms).mean()) * 1e-3
    B_alg = 80.0 * (n_in + n_out + merged + appended)
    B_real = fuse_real_bytes(n_in, n_view, n_unst_out, merged, appended, moved, Q, W * H)
    return {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
            "workload": "%d-surfel map (%.0f MB of planes, beyond the 256 MiB Infinity Cache), %.1f %% of the surfels unstable and "
                        "stale, spread from index 0" % (n, n * 80 / 1e6, 100.0 * stale.mean()),
            "kernel": "k_apply_merges + k_clean_flags + k_fuse_stream", "samples": len(rows),
            "merge_ms": float(mm.mean()), "clean_compact_ms": float(ms.mean()), "avg_kernel_ms": t * 1e3,
            "surfels_in": n_in, "surfels_out": n_out, "removed": n_in + appended - n_out, "moved": moved,
            "merged": merged, "appended": appended, "map_end": int(m_end.shape[0]), "status": status,
            "bytes_per_launch": B_alg, "achieved": B_alg / t / 1e9, "frac": B_alg / t / 1e9 / 8000.0,
            "real_bytes": B_real, "achieved_real": B_real / t / 1e9, "frac_real": B_real / t / 1e9 / 8000.0,
            "real_bytes_source": "model (fuse_real_bytes in bench.py); PMC FETCH_SIZE + WRITE_SIZE of the same command: profiles/"}


FUSE_KERNELS = ("k_apply_merges", "k_clean_flags", "k_fuse_stream")


def _run_group(cmd, cwd, env, timeout_s):
    """subprocess.run(capture_output, timeout) for a child that has children of its own (rocprofv3 -> python): the child leads its own
    session and on a timeout the WHOLE process group is killed, so no orphan keeps the GPU busy under the legs timed afterwards"""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        so, se = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.communicate()
        raise
    return subprocess.CompletedProcess(cmd, p.returncode, so, se)


def pmc_fuse_traffic(child_args, last_n, timeout_s=300.0, skip=None):
    """HBM-side bytes of the fuse pass's three kernels, MEASURED IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    cannot share one: 3 + 2 of the 4 TCC slots, /opt/skills/guides/MI355X_MICROARCH.md) over a short child of this script, nothing
    but --pmc in each.  Returns {kernel: {"FETCH_KB": mean per dispatch over the last `last_n` dispatches — or, with `skip`, over
    dispatches skip .. skip + last_n of the kernel, i.e. the timed frames behind the warm-up —, "WRITE_KB": ...}} or
    {"error": ...}; None when rocprofv3 is not on the box.  Counter values are KB (1024 B) per dispatch."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None
    # this process is itself being profiled (somebody ran `rocprofv3 ... bench.py`): no counter passes underneath a tracing tool
    # (the pool refuses --pmc combined with trace domains for a reason), and its tool libraries must not leak into a child
    tooling = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) or
               (k in ("HSA_TOOLS_LIB", "LD_PRELOAD") and "rocprof" in os.environ[k].lower())]
    if tooling:
        return {"error": "bench.py is running under a profiler (%s): counter passes skipped" % ", ".join(sorted(tooling)[:3])}
    out = {k: {} for k in FUSE_KERNELS}
    env = dict(os.environ, TMPDIR="/tmp", HRBF_BENCH_GEN_PROCS="1")     # no forked frame generators under the profiler
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    tmp = tempfile.mkdtemp(prefix="hrbf_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "x", "--", sys.executable, os.path.abspath(__file__)] + child_args
            r = _run_group(cmd, cwd="/tmp", env=env, timeout_s=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": "rocprofv3 --pmc %s: exit code %d, %d csv files: %s" % (counter, r.returncode, len(files), (r.stderr or "")[-300:])}
            per = {k: [] for k in FUSE_KERNELS}
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].split("(")[0]
                    if name in per and row["Counter_Name"] == counter:
                        per[name].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for k in FUSE_KERNELS:
                v = [x for _, x in sorted(per[k])]
                v = v[-last_n:] if skip is None else v[skip:skip + last_n]
                if not v:
                    return {"error": "no %s dispatches of %s in the profile" % (counter, k)}
                out[k][counter.split("_")[0] + "_KB"] = float(np.mean(v))
                out[k]["dispatches"] = len(v)
    except subprocess.TimeoutExpired:
        return {"error": "rocprofv3 pass ran into its %.0f s limit" % timeout_s}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def respawn_under_launcher(n):
    """`python bench.py --gpus N` by hand: become `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`"""
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def sharded_leg_command(args, world):
    """the one-sequence sharded leg: BASELINE config 4 at 2..7 ranks, config 5 at 8"""
    if world >= 8:
        shape = ["--width", "1280", "--height", "960", "--surfels", "8800000"]
        name = "BASELINE config 5: synthetic 1280x960 stream, 8.8 M surfels hash-owned over %d ranks" % world
    else:
        shape = ["--width", "640", "--height", "480", "--surfels", "4300000"]
        name = "BASELINE config 4: synthetic 640x480 stream, 4.3 M surfels hash-owned over %d ranks" % world
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--one-sequence-child", "--shard-map", "--partition", "hash",
           "--shard-odometry", "--steps", str(min(args.steps, 50)), "--warmup", str(min(args.warmup, 10)), "--cpu-frames", "0",
           "--worst-surfels", "0", "--big-surfels", "0", "--no-cpp-shim", "--no-sharded-leg", "--no-fit-leg"] + shape
    if args.dry_run:
        cmd.append("--dry-run")
    return cmd, name


def sharded_leg(args, rank, world, barrier, any_rank):
    """Every rank starts ONE child (same RANK / LOCAL_RANK / WORLD_SIZE, its own rendezvous port) and watches it; the children form
    their own communicator, so a failure or a hang in there ends with an `error` entry — the parent's process group is never inside
    the children's collectives.  The parents poll: every 2 s they agree (any_rank: a max-all-reduce over the parents) on whether some
    child has failed or the time limit has passed — then every parent kills its child at once instead of waiting for the limit
    while its child sits in a collective with a dead peer.  Returns the child line (rank 0) or {"error": ...}."""
    import subprocess
    import tempfile
    cmd, name = sharded_leg_command(args, world)
    env = dict(os.environ)
    # the children's rendezvous port: a FREE one picked by rank 0 and agreed through the parents' communicator (a fixed offset from
    # MASTER_PORT could be taken)
    import socket
    port = 0
    if rank == 0:
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env["MASTER_PORT"] = str(any_rank(port))
    env["MASTER_ADDR"] = "127.0.0.1"
    for k in [k for k in env if k.startswith("TORCHELASTIC")]:      # the children rendezvous on their own TCP store, not the agent's
        del env[k]
    barrier()
    t0 = time.perf_counter()
    res = None
    fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
    try:
        child = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, text=True, start_new_session=True)
    except Exception as e:
        child, res = None, {"error": repr(e)}
    while True:
        rc = child.poll() if child is not None else 1
        late = time.perf_counter() - t0 > args.sharded_leg_timeout
        failed = child is None or (rc is not None and rc != 0) or late
        stop = any_rank(1 if failed else 0)            # somebody's child failed (or the limit passed): everybody stops
        running = any_rank(1 if (rc is None) else 0)
        if stop or not running:
            break
        time.sleep(2.0)
    if child is not None and child.poll() is None:
        import signal
        try:
            os.killpg(child.pid, signal.SIGKILL)     # the child and whatever it started
        except ProcessLookupError:
            pass
        child.wait()
        res = {"error": "time limit of %.0f s" % args.sharded_leg_timeout if time.perf_counter() - t0 > args.sharded_leg_timeout
               else "stopped: another rank's child failed"}
    if child is not None and res is None:
        fo.seek(0); fe.seek(0)
        so, se = fo.read(), fe.read()
        if child.returncode:
            res = {"error": "child exit code %d: %s" % (child.returncode, se[-400:])}
        elif rank == 0:
            line = [l for l in so.splitlines() if l.startswith("{")]
            res = json.loads(line[-1]) if line else {"error": "no line from the child: " + se[-400:]}
    # rank 0 reports; if its own child was fine but another rank's failed, say so
    worst = any_rank(1 if (res is not None and "error" in res) else 0)
    barrier()
    if rank != 0:
        return None
    if worst and not (res and "error" in res):
        res = {"error": "a rank's child failed or was stopped"}
    res = dict(res or {})
    res["workload"] = name
    res["wall_s_incl_setup"] = time.perf_counter() - t0
    return res


def collective_latencies(dist, torch, W, H, world, iters=50):
    """microseconds of the collectives one sharded frame issues, timed on this communicator with torch.distributed calls of the same
    sizes and types (the library issues its own through librccl on its stream; this is the wire + launch cost, not a trace of them)"""
    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); dist.barrier()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t) / iters
    keys = torch.zeros(W * H, dtype=torch.int64, device="cuda")
    limbs = torch.zeros(29 * 3, dtype=torch.int64, device="cuda")
    word = torch.zeros(1, dtype=torch.int64, device="cuda")
    counts = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
    out = {"allreduce_min_keys_us": timed(lambda: dist.all_reduce(keys, op=dist.ReduceOp.MIN)),
           "allreduce_sum_limbs_us": timed(lambda: dist.all_reduce(limbs)),
           "allreduce_one_word_us": timed(lambda: dist.all_reduce(word)),
           "allgather_counts_us": timed(lambda: dist.all_gather(counts, word))}
    # per frame: 3 projections (key min-reduce + one meeting word each), 1 counts all-gather, 19 Gauss-Newton
    # iterations x 2 limb all-reduces + <= 10 SO3 iterations x 1
    out["per_frame_us_estimate"] = 3 * (out["allreduce_min_keys_us"] + out["allreduce_one_word_us"]) + out["allgather_counts_us"] + \
        (19 * 2 + 10) * out["allreduce_sum_limbs_us"]
    out["what"] = "same-size torch.distributed collectives on this RCCL communicator, %d iterations each" % iters
    return out


def dry_run(args, rank, world):
    """the N-rank plumbing without a GPU (gloo): rank count, barrier + max-over-ranks, and the sharded leg's child launch"""
    import torch
    import torch.distributed as dist
    fault = os.environ.get("HRBF_BENCH_TEST_CHILD", "") if args.one_sequence_child else ""     # tests: what the parents do when the
    if fault == "fail" and rank == world - 1:                                                   # sharded leg breaks or hangs
        raise SystemExit("injected failure of the sharded leg's rank %d" % rank)
    if fault == "hang":
        time.sleep(3600)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    one = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(one)
    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sharded = None
    if world > 1 and not args.one_sequence_child and not args.no_sharded_leg:
        def any_rank(v):
            t = torch.tensor([int(v)], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return int(t.item())
        sharded = sharded_leg(args, rank, world, barrier, any_rank)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_observed": int(one.item()), "max_over_ranks_s": float(t.item()),
                          "one_sequence_child": bool(args.one_sequence_child), "scaling": "strong" if args.one_sequence_child else "weak",
                          "sharded_one_sequence": sharded}))
        sys.stdout.flush()
    barrier()
    if world > 1:
        dist.destroy_process_group()


def dataset_leg(args):
    """BASELINE configs 2 / 3: a recorded sequence from its own files through the same timed loop (inputs decoded and resident in HBM
    before timing, W untimed + exactly K timed frames of the whole processFrame, one synchronise either side), started from an empty
    map like the reference's caller does, with the reference's GUI settings; ATE against the dataset's ground truth by the benchmark's
    rule; roofline of the fuse pass from the event ring; the oracle on the first frames as cpu_baseline and bit-parity witness."""
    import tempfile
    import torch
    from hrbffusion3d_amd import datasets as ds
    from hrbffusion3d_amd.api import HRBFFusion
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.gpus != 1:
        raise SystemExit("bench.py --dataset: one recorded sequence is one dependent chain of frames; run it with --gpus 1")
    seq, name = args.dataset, args.sequence
    if name is None:
        for cand in ds.SEQUENCES:
            if ds.find_sequence(args.dataset, cand):
                name = cand
                break
    if name is not None and ds.find_sequence(args.dataset, name):
        seq = ds.find_sequence(args.dataset, name)
    if not (os.path.isdir(os.path.join(seq, "rgb")) and os.path.isdir(os.path.join(seq, "depth"))):
        raise SystemExit("bench.py --dataset %s: no rgb/ and depth/ directories there (nor a known sequence below it)" % args.dataset)
    kind = args.dataset_kind or (ds.SEQUENCES[name][0] if name else
                                 ("icl" if any(os.path.isfile(os.path.join(seq, n)) for n in ds.GT_NAMES["icl"][:3]) else "tum"))
    K, Wm = args.steps, args.warmup
    work = tempfile.mkdtemp(prefix="hrbf_bench_dataset_")
    info = ds.prepare(seq, work, kind, max_frames=Wm + K)
    if info["frames"] < Wm + K:
        raise SystemExit("bench.py --dataset: the sequence has %d associated frames, --warmup %d + --steps %d need %d" % (info["frames"], Wm, K, Wm + K))
    t_dec = time.perf_counter()
    frames = list(ds.read_frames(info, Wm + K))
    t_dec = time.perf_counter() - t_dec
    cam = info["camera"]
    W, H = cam["width"], cam["height"]
    cap = 4 * 1024 * 1024 + (W // 2) * (H // 2) * 8
    prm = ds.params_for(info, max_surfels=cap)
    torch.cuda.set_device(0)
    fus = HRBFFusion(prm, device=0)
    d_rgb = [torch.from_numpy(f[1]).cuda() for f in frames]
    d_dep = [torch.from_numpy(f[2].view(np.int16)).cuda() for f in frames]
    torch.cuda.synchronize()
    for k in range(Wm):
        fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), frames[k][0])
    fus.synchronize()
    count0 = fus.surfel_count()
    fus.enable_timing(2)
    fus.set_fuse_ring_stride(max(1, args.ring_stride))
    fus.reset_fuse_ring()
    torch.cuda.synchronize(); fus.synchronize()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), frames[k][0])
    t_enq = time.perf_counter() - t0
    fus.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mm, ms2, st = fus.fuse_ring_parts(K)
    ms = mm + ms2
    ok = (mm >= 0) & (ms2 > 0)
    B = 80.0 * (st[:, 0].astype(np.float64) + st[:, 3] + st[:, 1] + st[:, 2])
    gbps = float((B[ok] / (ms[ok] * 1e-3)).mean() / 1e9) if ok.any() else 0.0
    fuse_ms = float(ms[ok].mean()) if ok.any() else 0.0
    merge_ms = float(mm[ok].mean()) if ok.any() else 0.0
    count1 = fus.surfel_count()
    traj = fus.pose_log(0, Wm + K)
    status = fus.status()
    fus.enable_timing(False)
    ate = None
    stamps = info["stamps_s"][:len(traj)]
    if info["groundtruth"]:
        gs, gp = ds.load_groundtruth(info["groundtruth"])
        est = [np.asarray(T, np.float64).copy() for T in traj]
        if info["icl_nuim"]:       # what the reference's writer does before anybody compares (TrajectoryManager.cpp:329)
            for T in est:
                T[1, 3] = -T[1, 3]
        ate = ds.evaluate_ate(stamps, est, gs, gp)
        ate_timed = ds.evaluate_ate(stamps[Wm:], est[Wm:], gs, gp)
    # the oracle on the first frames of the same files: cpu_baseline (bounded sample) and the bit-parity witness
    cpu = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "skipped (--cpu-frames 0)"}
    if args.cpu_frames > 0:
        try:
            from oracle_lib import Oracle
            n = min(args.cpu_frames, Wm + K)
            cores = min(len(os.sched_getaffinity(0)), 32)
            os.environ["OMP_NUM_THREADS"] = str(cores)
            o = Oracle(prm, omp=True)
            t1 = time.perf_counter()
            otraj = []
            for k in range(n):
                o.process_frame(frames[k][1], frames[k][2], frames[k][0])
                otraj.append(o.get_pose())
            t1 = time.perf_counter() - t1
            o.close()
            same = all(np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32)) for a, b in zip(otraj, traj[:n]))
            e = np.asarray([a[:3, 3] for a in otraj], np.float64) - np.asarray([b[:3, 3] for b in traj[:n]], np.float64)
            cpu = {"value": n / t1, "unit": "frames/s", "cores": cores, "kind": "port", "host_cores": len(os.sched_getaffinity(0)),
                   "sample": "the first %d frames of the same sequence from an empty map (oracle, OpenMP, %d threads)" % (n, cores),
                   "ate_vs_oracle_mm": float(1000.0 * np.sqrt((e ** 2).sum(1).mean())), "ate_vs_oracle_frames": n, "poses_bit_identical": bool(same)}
        except Exception as e:
            cpu = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    label = args.dataset_label or {"icl_nuim_lr_kt2": "ICL-NUIM living-room kt2", "tum_fr1_desk": "TUM fr1/desk"}.get(name, os.path.basename(os.path.normpath(seq)))
    out = {
        "metric": "frames/sec at 640x480, 1M-surfel map, 1 MI355X; ATE vs reference",
        "value": K / dt, "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": 1000.0 * dt / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "real" if not args.dataset_label else "synthetic",
        "config": {"workload": "%s, %dx%d, %s layout read from %s, the reference's GUI settings with the sparse back-end off, from an empty map, "
                               "full processFrame per step" % (label, W, H, "TUM RGB-D" if kind == "tum" else "ICL-NUIM (TUM-compatible PNG)", seq),
                   "parallelism": "single GPU", "surfels_start": int(count0), "surfels_end": int(count1),
                   "ate_rmse_mm": None if not ate or ate["rmse_m"] is None else 1000.0 * ate["rmse_m"], "ate_frames": None if not ate else ate["pairs"],
                   "ate_timed_frames_rmse_mm": None if not ate or ate_timed["rmse_m"] is None else 1000.0 * ate_timed["rmse_m"],
                   "ate_against": "the dataset's ground truth (%s): stamps associated within 20 ms, Horn alignment, translation RMSE" % info["groundtruth"],
                   "ate_vs_oracle_mm": cpu.get("ate_vs_oracle_mm"),
                   "intrinsics": [cam["fx"], cam["fy"], cam["cx"], cam["cy"]], "depth_factor": cam["factor"],
                   "host_submit_ms_per_frame": 1e3 * t_enq / K, "png_decode_s_before_timing": t_dec},
        "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "traffic": None,
                     "kernel": "k_apply_merges + k_clean_flags + k_fuse_stream (F2 + F3, every kernel of the pass)",
                     "avg_kernel_ms": fuse_ms, "merge_ms": merge_ms, "clean_compact_ms": fuse_ms - merge_ms, "launches_timed": int(ok.sum()),
                     "timed_every_nth_frame": max(1, args.ring_stride), "bytes_per_launch": float(B[ok].mean()) if ok.any() else 0.0,
                     "note": "a map grown from one sequence (well under 1 M surfels) is cache-resident: three dependent launches of latency, "
                             "not an HBM measurement — the HBM-bound leg is the default run's roofline_worst_case", "status": status},
        "cpu_baseline": cpu,
    }
    fus.close()
    print(json.dumps(out)); sys.stdout.flush()


def main():
    args = parse()
    if args.dataset:
        return dataset_leg(args)
    if args.only_worst:
        print(json.dumps({"roofline_worst_case": worst_case_leg(args, int(os.environ.get("LOCAL_RANK", "0")))}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.virtual_shards <= 1:
        respawn_under_launcher(args.gpus)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus) and args.virtual_shards <= 1:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE = %d ranks" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    t_gen = time.perf_counter()
    gen = _frames(range(1 + args.warmup + args.steps), args.width, args.height, args.noise)   # forked workers: before the HIP runtime starts
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ranks_observed = 1
    if dist is not None:       # what RCCL itself connects: one contribution per rank
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one)
        ranks_observed = int(one.item())
        if ranks_observed != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: --gpus %d but RCCL sees %d ranks" % (args.gpus, ranks_observed))

    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import default_params

    W, H = args.width, args.height
    K, Wm = args.steps, args.warmup
    nframes = 1 + Wm + K
    frames = [(g[0], g[1]) for g in gen]
    poses = [g[2] for g in gen]
    del gen
    seed = synth.seed_map(args.surfels, t_now=1, width=W)
    t_gen = time.perf_counter() - t_gen

    fx, fy, cx, cy = synth.intrinsics(W, H)
    cap = int(seed.shape[0] + 200_000 + (W // 2) * (H // 2) * 4)
    cap += (W // 2) * (H // 2) * min(nframes, 64)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=cap)
    fus = HRBFFusion(p, device=local_rank)
    one_sequence = args.shard_odometry or args.shard_map
    if args.virtual_shards > 1:
        fus.comm_init(-1, args.virtual_shards)
        fus.map_shard_init(True, partition=args.partition)
    elif one_sequence:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(HRBFFusion.comm_unique_id()), dtype=torch.uint8, device="cuda")
        if dist is not None:
            dist.broadcast(uid, src=0)
        fus.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
        if args.shard_map:
            fus.map_shard_init(True, partition=args.partition)       # every rank then keeps its slice of the uploaded map
            fus.set_row_sharding(args.shard_odometry)   # the 39 registration all-reduces only when asked for
    fus.upload_map(seed)
    fus.set_pose(poses[0])
    fus.bootstrap(frames[0][0], frames[0][1])

    # inputs resident in HBM before timing
    d_rgb = [torch.from_numpy(f[0]).cuda() for f in frames]
    d_dep = [torch.from_numpy(f[1].view(np.int16)).cuda() for f in frames]
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    for k in range(1, 1 + Wm):
        fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), k)
    fus.synchronize()
    count0 = fus.surfel_count()
    fus.enable_timing(2)     # only the events around the fuse pass (roofline); region events stay off
    fus.set_fuse_ring_stride(max(1, args.ring_stride))
    fus.reset_fuse_ring()
    fus.comm_stats(reset=True)      # the library's collective counters cover exactly the timed frames

    barrier(); torch.cuda.synchronize(); fus.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + Wm, 1 + Wm + K):
        fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), k)
    t_enqueued = time.perf_counter() - t0       # the host has submitted every frame; the device may still be working
    fus.synchronize(); torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0

    comm_timed = fus.comm_stats()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # roofline of the fuse pass (F2 + F3) from the event ring recorded during the timed region:
    # merge_ms = k_apply_merges, stream_ms = k_clean_flags + k_fuse_stream — every kernel the pass consists of
    mm, ms2, st = fus.fuse_ring_parts(K)
    ms = mm + ms2
    ok = (mm >= 0) & (ms2 > 0)
    B = 80.0 * (st[:, 0].astype(np.float64) + st[:, 3] + st[:, 1] + st[:, 2])   # 80(N_in + N_out + M + A), SURVEY §8d
    gbps = float((B[ok] / (ms[ok] * 1e-3)).mean() / 1e9) if ok.any() else 0.0
    fuse_ms = float(ms[ok].mean()) if ok.any() else 0.0
    merge_ms = float(mm[ok].mean()) if ok.any() else 0.0
    count1 = fus.surfel_count()
    P_end = fus.get_pose()
    # the trajectory of every frame so far from the device-written pose ring: log entry f = stream frame f + 1 (the bootstrap is not logged)
    traj_all = fus.pose_log(0, Wm + K)
    ate_mm = 1000.0 * synth.ate_rmse(traj_all[Wm:Wm + K], poses[1 + Wm:1 + Wm + K]) if len(traj_all) == Wm + K else None
    m_timed = fus.download_map() if (world == 1 and args.virtual_shards <= 1) else None   # for the real-bytes model below
    # A sharded run checks ITSELF (the claim of DESIGN.md §7 where the driver can see it, on whatever hardware this runs): every rank
    # ends with the same pose bits, and rank 0 replays the same frames against the same map on ONE unsharded context — same pose
    # bits, same global surfel count.  Outside the timed region; ~0.1 s.
    self_check = None
    if one_sequence or args.virtual_shards > 1:
        pb = np.ascontiguousarray(P_end, np.float32).view(np.uint32).astype(np.int64).ravel()
        same_on_all = True
        if dist is not None:
            mine_p = torch.from_numpy(pb).cuda()
            all_p = [torch.zeros_like(mine_p) for _ in range(world)]
            dist.all_gather(all_p, mine_p)
            same_on_all = all(bool(torch.equal(a, mine_p)) for a in all_p)
        if rank == 0:
            try:
                ref = HRBFFusion(p, device=local_rank)
                ref.upload_map(seed); ref.set_pose(poses[0]); ref.bootstrap(frames[0][0], frames[0][1])
                for k in range(1, 1 + Wm + K):
                    ref.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), k)
                ref.synchronize()
                rb = np.ascontiguousarray(ref.get_pose(), np.float32).view(np.uint32).astype(np.int64).ravel()
                self_check = {"pose_bits_equal_on_all_ranks": bool(same_on_all), "pose_bits_equal_one_unsharded_gpu": bool(np.array_equal(rb, pb)),
                              "surfel_count_equal_one_unsharded_gpu": bool(ref.surfel_count() == count1), "frames_replayed": int(Wm + K),
                              "status_unsharded": int(ref.status())}
                ref.close()
            except Exception as e:      # the check must never take the line down
                self_check = {"error": repr(e), "pose_bits_equal_on_all_ranks": bool(same_on_all)}
    tm = np.zeros(8, np.float32)
    # PCIe-inclusive rate of the host-pointer entry point (never `value`): 1.5 MB upload + sync per frame
    pcie_fps = None
    update_model = None
    if world == 1:
        nh = min(20, K)
        fus.enable_timing(False)
        for k in range(1 + Wm, 1 + Wm + min(6, K)):      # first use allocates the pinned staging ring; clocks back up after the read-back
            fus.process_frame(frames[k][0], frames[k][1], k)
        fus.synchronize()
        t1 = time.perf_counter()
        for k in range(1 + Wm + K - nh, 1 + Wm + K):
            fus.process_frame(frames[k][0], frames[k][1], k)
        fus.synchronize()
        pcie_fps = nh / (time.perf_counter() - t1)
        # Stopwatch regions of one more frame, outside every timed loop (twelve event records per frame)
        fus.enable_timing(1)
        fus.process_frame_device(d_rgb[Wm + K].data_ptr(), d_dep[Wm + K].data_ptr(), Wm + K)
        tm = fus.timings()
        fus.enable_timing(False)
        # the caller-side map correction (GlobalModel::updateModel), after everything that is reported:
        # wall time per call including the 64-byte matrix upload and its sync; the kernel alone is in profiles/
        eye = np.eye(4, dtype=np.float32)[None]
        fus.update_model(eye); fus.synchronize()
        t2 = time.perf_counter()
        for _ in range(20):
            fus.update_model(eye)
        fus.synchronize()
        um_ms = 1000.0 * (time.perf_counter() - t2) / 20
        n_now = fus.surfel_count()
        update_model = {"ms_per_call_incl_upload": um_ms, "surfels": int(n_now),
                        "algorithmic_GBps_160B_per_surfel": 160.0 * n_now / (um_ms * 1e-3) / 1e9}
    status = fus.status()
    real_bytes = None
    if m_timed is not None and ok.any():
        m_now = m_timed
        n_view, n_unst_out = frustum_counts(m_now, P_end, (fx, fy, cx, cy), W, H, p.confidence_threshold)
        sm = st[ok].astype(np.float64).mean(axis=0)
        real_bytes = fuse_real_bytes(sm[0], n_view, n_unst_out, sm[1], sm[2], sm[6], (W // 2) * (H // 2), W * H)
        del m_now
    err_mm = float(1000.0 * np.linalg.norm(P_end[:3, 3] - poses[Wm + K][:3, 3]))

    # HBM traffic of the fuse pass: PMC counters cannot be collected from inside this process; the figure is the
    # one measured with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) over this very command
    # and committed with its calibration under profiles/ (null if the file is missing)
    traffic = None
    traffic_src = None
    traffic_kernels = None
    if rank == 0 and world == 1 and args.virtual_shards <= 1 and not one_sequence and not args.no_traffic:
        fus.synchronize()
        kp = min(K, 10)
        traffic_kernels = pmc_fuse_traffic(["--steps", str(kp), "--warmup", "3", "--surfels", str(args.surfels), "--width", str(W), "--height", str(H),
                                            "--cpu-frames", "0", "--worst-surfels", "0", "--big-surfels", "0", "--no-cpp-shim", "--no-traffic", "--no-fit-leg"] +
                                           (["--noise"] if args.noise else []), last_n=kp, skip=3)
        if traffic_kernels is not None and "error" not in traffic_kernels:
            # raw counters, no gfx950 x2: pass A mixes a 16-B stream with scattered texel gathers and the 86 MB map is resident in the
            # 256 MiB Infinity Cache at this size (the guide's x2 is calibrated for wide coalesced streams; see roofline_worst_case)
            traffic = 1024.0 * sum(v["FETCH_KB"] + v["WRITE_KB"] for v in traffic_kernels.values())
            traffic_src = "this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over a %d-frame child of this command, mean per dispatch of the three kernels" % kp
    if traffic is None:      # no profiler on the box (or a failed pass): the figure committed with its calibration, and say so
        for name in ("r04_fuse_traffic.json", "r03_fuse_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    traffic = float(json.load(f)["traffic_bytes_per_launch"])
                traffic_src = "profiles/" + name + " (not measured in this run%s)" % (": " + traffic_kernels["error"] if traffic_kernels else ": no rocprofv3 on this box")
                break
            except Exception:
                traffic = None
    shim = None
    if rank == 0 and world == 1 and args.virtual_shards <= 1 and not one_sequence and not args.no_cpp_shim:
        fus.synchronize()
        try:
            shim = cpp_shim_leg(args, seed, frames, poses)
        except Exception as e:
            shim = {"error": repr(e)}
    worst = None
    if rank == 0 and world == 1 and args.worst_surfels > 0 and args.virtual_shards <= 1 and not one_sequence:
        fus.synchronize()
        try:
            worst = worst_case_leg(args, local_rank)
            wk = None if args.no_traffic else pmc_fuse_traffic(
                ["--only-worst", "--worst-samples", "3", "--worst-surfels", str(args.worst_surfels), "--worst-frac", str(args.worst_frac),
                 "--width", str(W), "--height", str(H)], last_n=1)
            if wk is not None and "error" not in wk:
                # the LAST dispatch of each kernel is the measured frame (the whole map moves).  k_fuse_stream's reads are a wide
                # coalesced stream: FETCH_SIZE tallies its 128-B requests at 64 B on gfx950 -> x2 (MI355X_MICROARCH.md, HBM);
                # writes and the other two kernels (gathers, sparse updates) as counted
                tw = 1024.0 * (2.0 * wk["k_fuse_stream"]["FETCH_KB"] + wk["k_fuse_stream"]["WRITE_KB"] +
                               sum(wk[k]["FETCH_KB"] + wk[k]["WRITE_KB"] for k in ("k_apply_merges", "k_clean_flags")))
                worst["traffic"] = tw
                worst["achieved_traffic"] = tw / (worst["avg_kernel_ms"] * 1e-3) / 1e9
                worst["frac_traffic"] = worst["achieved_traffic"] / 8000.0
                worst["traffic_kernels_KB"] = wk
                worst["traffic_source"] = "this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --only-worst --worst-samples 3`, last dispatch of each kernel; k_fuse_stream FETCH x2 (gfx950 correction for coalesced streams)"
            else:
                try:
                    with open(os.path.join(ROOT, "profiles", "r04_fuse_traffic.json")) as f:
                        tw = float(json.load(f)["worst_case_leg_4.34M_surfels_all_moved"]["traffic_bytes_per_launch"])
                    if args.worst_surfels == 4_300_000 and abs(args.worst_frac - 0.05) < 1e-9 and (W, H) == (640, 480):
                        worst["traffic"] = tw
                        worst["achieved_traffic"] = tw / (worst["avg_kernel_ms"] * 1e-3) / 1e9
                        worst["traffic_source"] = "profiles/r04_fuse_traffic.json (not measured in this run%s)" % (": " + wk["error"] if wk else "")
                except Exception:
                    pass
        except Exception as e:   # the second leg must never take the bench line down
            worst = {"error": repr(e)}

    big = None
    if rank == 0 and world == 1 and args.big_surfels > 0 and args.virtual_shards <= 1 and not one_sequence:
        fus.synchronize()
        try:
            big = big_leg(args, local_rank)
        except Exception as e:   # the extra leg must never take the bench line down
            big = {"error": repr(e)}

    # EXTENSION leg (no reference counterpart, not part of `value`): the true Hermite-RBF fit on the matrix core over the live frame
    # the context holds (hrbf_fit_curvature, csrc/k_fit.hip) — BASELINE config 5's "batched-HRBF small-GEMM on MFMA"
    fit = None
    if rank == 0 and world == 1 and args.virtual_shards <= 1 and not one_sequence and not args.no_fit_leg:
        try:
            fus.synchronize()
            fus.fit_curvature(timed=True)
            ts = [fus.fit_curvature(timed=True) for _ in range(5)]
            c1 = fus.get_image("FIT_CURV1")
            fitted = int((c1[..., 3] != 1000.0).sum())
            ms_fit = float(np.mean(ts))
            n_sys = 100                      # 25 centres x 4 unknowns (+ the right-hand-side row; padded to 7 x 7 blocks of 16)
            flops_alg = fitted * (n_sys ** 3 / 3.0 + 2.0 * n_sys ** 2)      # Cholesky + two triangular solves of the 100 x 100 system
            flops_mfma = fitted * 77 * 2.0 * 16 ** 3                        # what the matrix core executes: 56 update + 21 panel products of 16 x 16 x 16
            fit = {"what": "EXTENSION, no reference counterpart: Hermite-RBF fit (5x5 window, 100x100 SPD system per pixel, blocked left-looking Cholesky "
                           "with the whole system in registers, panel solves and trailing updates on v_mfma_f32_16x16x4_f32) over the %dx%d live frame; "
                           "never part of `value`" % (W, H),
                   "ms_per_call": ms_fit, "pixels_fitted": fitted, "systems_per_s": fitted / (ms_fit * 1e-3),
                   "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3, "achieved": flops_alg / (ms_fit * 1e-3) / 1e12,
                                "frac": flops_alg / (ms_fit * 1e-3) / 1e12 / 157.3,
                                "achieved_executed_mfma": flops_mfma / (ms_fit * 1e-3) / 1e12,
                                "note": "f32-input MFMA peak at 2.4 GHz (= the packed-f32 vector peak; tools/probes/mfma_f32_rate.hip measures 151-155 TFLOP/s for "
                                        "v_mfma_f32_16x16x4_f32 in runs long enough for the clock to settle, 2.39 GHz under the load: the guide's figure). `achieved` "
                                        "counts the algorithm's flops (n^3/3 + 2n^2, n = 100), `achieved_executed_mfma` the padded 16-blocks the matrix core really "
                                        "multiplies. On this part f32 MFMA and VALU instructions do not overlap, also not across the waves of a SIMD (their times "
                                        "add, profiles/r06_mfma_f32_rate.txt): the kernel is its 308 MFMAs at the full rate (1.24 ms per 640x480 frame) plus its "
                                        "vector work (1.1 ms) plus the MFMA -> VALU -> MFMA dependency stalls four waves per SIMD do not hide (~0.45 ms)"}}
        except Exception as e:
            fit = {"error": repr(e)}

    # the LIBRARY's own account of its communicator and of what it issued over the timed frames (hrbf_comm_stats): what RCCL itself
    # reports as world size and rank for the library's communicator, and the exchange steps per frame (DESIGN.md §7's model)
    lib_comm = None
    if one_sequence or args.virtual_shards > 1:
        cs = comm_timed
        keys = ("world", "rank", "frames", "limb_allreduce", "limb_allreduce_bytes", "key_min_reduce", "key_min_reduce_bytes", "allgather",
                "allgather_bytes", "word_allreduce", "send", "send_bytes", "recv", "recv_bytes", "host_barriers")
        rows = [[cs[k] for k in keys]]
        if dist is not None:
            mine_cs = torch.tensor(rows[0], dtype=torch.int64, device="cuda")
            all_cs = [torch.zeros_like(mine_cs) for _ in range(world)]
            dist.all_gather(all_cs, mine_cs)
            rows = [[int(v) for v in a.tolist()] for a in all_cs]
        fr = max(1, rows[0][2])
        lib_comm = {"transport": cs["transport"], "per_rank": [dict(zip(keys, r)) for r in rows],
                    "world_sizes_reported_by_the_library": sorted(set(r[0] for r in rows)),
                    "ranks_reported_by_the_library": sorted(r[1] for r in rows),
                    "per_frame_rank0": {"limb_allreduce": rows[0][3] / fr, "key_min_reduce": rows[0][5] / fr, "allgather": rows[0][7] / fr,
                                        "word_allreduce": rows[0][9] / fr, "send": rows[0][10] / fr, "recv": rows[0][12] / fr},
                    "model": "per frame: 3 key min-reduces (one per projection), 1 counts all-gather (+ 1 first-id all-gather under hash "
                             "ownership), 10 + 2 x 19 limb all-reduces when the registration is row-sharded, 1 one-word all-reduce per "
                             "projection with peer-mapped images (or the packed records over send / recv)"}
    per_rank_fuse_ms = None
    coll = None
    if dist is not None:
        mine = torch.tensor([fuse_ms, merge_ms, float(count1)], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_fuse_ms = [[float(x) for x in a.tolist()] for a in allr]
        if one_sequence:
            try:
                coll = collective_latencies(dist, torch, W, H, world)
            except Exception as e:
                coll = {"error": repr(e)}
    sharded = None
    if dist is not None and not one_sequence and not args.no_sharded_leg:
        fus.synchronize(); torch.cuda.synchronize()
        def any_rank(v):
            t = torch.tensor([int(v)], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return int(t.item())
        sharded = sharded_leg(args, rank, world, barrier, any_rank)
        if sharded is not None and "error" not in sharded:      # keep what the leg is for; the rest of the child's line repeats this one's
            sharded = {k: sharded.get(k) for k in ("workload", "value", "unit", "ms_per_step", "scaling", "n_gpus", "ranks_observed", "steps", "warmup",
                                                   "per_rank_fuse_ms", "collectives", "library_comm", "sharded_self_check", "wall_s_incl_setup")} | {
                "parallelism": sharded.get("config", {}).get("parallelism"), "surfels_end_rank0": sharded.get("config", {}).get("surfels_end"),
                "final_translation_error_mm": sharded.get("config", {}).get("final_translation_error_mm"),
                "status": sharded.get("roofline", {}).get("status")}

    if rank == 0:
        out = {
            "metric": "frames/sec at 640x480, 1M-surfel map, 1 MI355X; ATE vs reference",
            "value": (K if one_sequence else world * K) / dt, "unit": "frames/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "ms_per_step": 1000.0 * dt / K, "higher_is_better": True,
            "scaling": "strong" if one_sequence else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d RGB-D stream (room+sphere+relief, Lissajous path, seed 12345), "
                                   "map pre-seeded to %d surfels, full processFrame per step" % (W, H, seed.shape[0]),
                       "surfels_start": int(count0), "surfels_end": int(count1),
                       "parallelism": ("surfel map in %d %s shards%s (RCCL)" % (world, "hash-owned" if args.partition == "hash" else "contiguous", " + row-sharded registration" if args.shard_odometry else "")) if args.shard_map
                                      else ("row-sharded registration x%d (RCCL int64 all-reduce)" % world) if args.shard_odometry
                                      else ("%d virtual %s map shards on one GPU" % (args.virtual_shards, "hash-owned" if args.partition == "hash" else "contiguous")) if args.virtual_shards > 1
                                      else ("replicas x%d" % world if world > 1 else "single GPU"),
                       # ATE: translation RMSE of the K timed frames against the stream's ANALYTIC camera poses (same world frame: tracking
                       # starts from the analytic pose of frame 0).  "vs reference" (the reference's own trajectory) cannot be measured
                       # here — it cannot run in this image; ate_vs_oracle_mm (cpu_baseline) is the HIP path against the restatement
                       "ate_rmse_mm": ate_mm, "ate_frames": K, "ate_against": "analytic poses of the synthetic stream",
                       "ate_vs_oracle_mm": None,
                       "final_translation_error_mm": err_mm, "pcie_inclusive_fps": pcie_fps,
                       "host_submit_ms_per_frame": 1e3 * t_enqueued / K,
                       "cpp_shim": shim,
                       "update_model": update_model,
                       "last_frame_region_ms": {"Initialization": float(tm[0]), "Registration": float(tm[1]),
                                                "Integration": float(tm[2]), "Prediction": float(tm[3]),
                                                "fuse_stream_pass": float(tm[4])}},
            "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_kernels_KB": traffic_kernels,
                         "kernel": "k_apply_merges + k_clean_flags + k_fuse_stream (F2 + F3, every kernel of the pass)",
                         "avg_kernel_ms": fuse_ms, "merge_ms": merge_ms, "clean_compact_ms": fuse_ms - merge_ms,
                         "launches_timed": int(ok.sum()), "timed_every_nth_frame": max(1, args.ring_stride),
                         "bytes_per_launch": float(B[ok].mean()) if ok.any() else 0.0,
                         "real_bytes": real_bytes,
                         "achieved_real": (real_bytes / (fuse_ms * 1e-3) / 1e9) if real_bytes and fuse_ms > 0 else None,
                         "real_bytes_source": "model (fuse_real_bytes in bench.py) from this run's item statistics",
                         "note": "the 1 M-surfel map (86 MB) sits inside the 256 MiB Infinity Cache and nothing is removed on this "
                                 "stream, so the in-place pass moves almost nothing: `achieved` is in SURVEY §8d's algorithmic "
                                 "currency; roofline_worst_case is the HBM-bound measurement",
                         "moved_per_frame": float(st[ok][:, 6].mean()) if ok.any() else 0.0, "status": status},
            "roofline_worst_case": worst,
            "config5_single_gpu": big,
            "extension_hrbf_fit_mfma": fit,
            "ranks_observed": ranks_observed,
            "per_rank_fuse_ms": per_rank_fuse_ms,      # [fuse pass ms, of which merge ms, live surfels] per rank
            "collectives": coll,
            "library_comm": lib_comm,
            "sharded_self_check": self_check,
            "sharded_one_sequence": sharded,
        }
        if args.cpu_frames > 0 and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, seed, frames, poses, list(traj_all))
                out["config"]["ate_vs_oracle_mm"] = out["cpu_baseline"].get("ate_vs_oracle_mm")
            except Exception as e:  # the baseline leg must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        else:
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                   "sample": "skipped (rank 0 at N=1 only)"}
        if args.verbose:
            sys.stderr.write("gen %.1fs, timed %.3fs, fuse kernel %.4f ms, %s\n" % (t_gen, dt, fuse_ms, st[-1]))
    # RCCL prints a version banner through C stdio when a communicator is created; when stdout is a pipe it would be
    # flushed at exit, i.e. AFTER the JSON line.  Push everything out first, print the line, then silence fd 1.
    libc = C.CDLL(None)
    sys.stdout.flush(); libc.fflush(None)
    barrier()
    if rank == 0:
        print(json.dumps(out)); sys.stdout.flush()
    barrier()
    devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
    fus.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
