#!/usr/bin/env python3
"""bench.py — frames/sec of the HRBF-Fusion per-frame hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one hrbf_process_frame over one synthetic 640x480 RGB-D frame (pre-processing, SO3 +
3-level joint ICP/RGB registration, index-map projection x3, data association + merge, clean +
compact + append, HRBF ray-cast prediction + fill-in) against a map pre-seeded to >= 1 M surfels.
Inputs are resident in HBM before the timed region (torch tensors; PyTorch is plumbing only).

N > 1: one process per GPU (torchrun), each an independent replica of the same sequence
(SURVEY.md §8e last row: the per-frame path of one sequence does not shard below VGA); no data-path
collective, "scaling": "weak", value = frames all ranks processed / max-over-ranks time.

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
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--surfels", type=int, default=1_050_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--noise", action="store_true", help="Kinect-style depth noise + 3%% drop-outs")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the CPU-oracle sample, ~10 s of host time (0 = skip)")
    ap.add_argument("--shard-odometry", action="store_true",
                    help="N > 1: all ranks track ONE sequence, registration reductions row-sharded + RCCL all-reduce "
                         "(SURVEY §8e sharding 1; strong scaling, a latency cost at VGA). Default: independent replicas")
    ap.add_argument("--shard-map", action="store_true",
                    help="all ranks track ONE sequence against ONE surfel map cut into contiguous ranges over the ranks "
                         "(SURVEY §8e sharding 2; add --shard-odometry for the row-sharded registration too; strong scaling; pays off "
                         "for maps far beyond 1 M surfels). Default: independent replicas")
    ap.add_argument("--virtual-shards", type=int, default=0,
                    help="single process: play G map shards in turn on one GPU (measures the sharded path's extra "
                         "kernels without any interconnect); not a benchmark configuration")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, seed, frames, poses):
    """Oracle (OpenMP build) on the same workload, bounded sample: bootstrap + n frames."""
    from oracle_lib import Oracle
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.params import default_params
    n = args.cpu_frames
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(cores, 32))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    fx, fy, cx, cy = synth.intrinsics(args.width, args.height)
    p = default_params(args.width, args.height, fx, fy, cx, cy, max_surfels=int(seed.shape[0] + 200_000 + 80_000 * (n + 1)))
    o = Oracle(p, omp=True)
    o.upload_map(seed)
    o.set_pose(poses[0])
    o.bootstrap(frames[0][0], frames[0][1])
    t0 = time.perf_counter()
    for k in range(1, n + 1):
        o.process_frame(frames[k][0], frames[k][1])
    dt = time.perf_counter() - t0
    o.close()
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames of the same %dx%d stream against the same %d-surfel map (oracle, OpenMP)" %
                      (n, args.width, args.height, seed.shape[0])}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import default_params

    W, H = args.width, args.height
    K, Wm = args.steps, args.warmup
    nframes = 1 + Wm + K
    t_gen = time.perf_counter()
    frames, poses = [], []
    for k in range(nframes):
        rgb, depth, T = synth.frame(k, W, H, noise=args.noise)
        frames.append((rgb, depth)); poses.append(T)
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
        fus.map_shard_init(True)
    elif one_sequence:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(HRBFFusion.comm_unique_id()), dtype=torch.uint8, device="cuda")
        if dist is not None:
            dist.broadcast(uid, src=0)
        fus.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
        if args.shard_map:
            fus.map_shard_init(True)       # every rank then keeps its slice of the uploaded map
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
    fus.enable_timing(2)     # only the two events around the fuse pass (roofline); region events stay off
    fus.reset_fuse_ring()

    barrier(); torch.cuda.synchronize(); fus.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + Wm, 1 + Wm + K):
        fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), k)
    fus.synchronize(); torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # roofline of the fuse streaming kernel from the event ring recorded during the timed region
    ms, st = fus.fuse_ring(K)
    ok = ms > 0
    B = 80.0 * (st[:, 0].astype(np.float64) + st[:, 3] + st[:, 1] + st[:, 2])   # 80(N_in + N_out + M + A), SURVEY §8d
    gbps = float((B[ok] / (ms[ok] * 1e-3)).mean() / 1e9) if ok.any() else 0.0
    fuse_ms = float(ms[ok].mean()) if ok.any() else 0.0
    count1 = fus.surfel_count()
    P_end = fus.get_pose()
    tm = np.zeros(8, np.float32)
    # PCIe-inclusive rate of the host-pointer entry point (never `value`): 1.5 MB upload + sync per frame
    pcie_fps = None
    update_model = None
    if world == 1:
        nh = min(20, K)
        fus.enable_timing(False)
        for k in range(1 + Wm, 1 + Wm + min(3, K)):      # first use allocates the pinned staging ring
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
    err_mm = float(1000.0 * np.linalg.norm(P_end[:3, 3] - poses[Wm + K][:3, 3]))

    # HBM traffic of the fuse pass: PMC counters cannot be collected from inside this process; the figure is the
    # one measured with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) over this very command
    # and committed with its calibration under profiles/ (null if the file is missing)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_fuse_traffic.json")) as f:
            traffic = float(json.load(f)["traffic_bytes_per_launch"])
    except Exception:
        traffic = None

    if rank == 0:
        out = {
            "metric": "frames/sec at 640x480, 1M-surfel map, 1 MI355X",
            "value": (K if one_sequence else world * K) / dt, "unit": "frames/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "ms_per_step": 1000.0 * dt / K, "higher_is_better": True,
            "scaling": "strong" if one_sequence else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d RGB-D stream (room+sphere+relief, Lissajous path, seed 12345), "
                                   "map pre-seeded to %d surfels, full processFrame per step" % (W, H, seed.shape[0]),
                       "surfels_start": int(count0), "surfels_end": int(count1),
                       "parallelism": ("surfel map in %d contiguous shards%s (RCCL)" % (world, " + row-sharded registration" if args.shard_odometry else "")) if args.shard_map
                                      else ("row-sharded registration x%d (RCCL int64 all-reduce)" % world) if args.shard_odometry
                                      else ("%d virtual map shards on one GPU" % args.virtual_shards) if args.virtual_shards > 1
                                      else ("replicas x%d" % world if world > 1 else "single GPU"),
                       "final_translation_error_mm": err_mm, "pcie_inclusive_fps": pcie_fps,
                       "update_model": update_model,
                       "last_frame_region_ms": {"Initialization": float(tm[0]), "Registration": float(tm[1]),
                                                "Integration": float(tm[2]), "Prediction": float(tm[3]),
                                                "fuse_stream_pass": float(tm[4])}},
            "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                         "traffic": traffic, "kernel": "k_clean_flags + k_fuse_stream", "avg_kernel_ms": fuse_ms,
                         "bytes_per_launch": float(B[ok].mean()) if ok.any() else 0.0},
        }
        if args.cpu_frames > 0 and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, seed, frames, poses)
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
