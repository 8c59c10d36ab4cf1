"""hrbf_comm_stats: the library's own account of its communicator and of the exchange steps it issues, held to the model of
SURVEY.md §8e / DESIGN.md §7 — per tracked frame of a sharded run:

    3 key min-reduces over W*H u64 keys            (one per projection: HRBFFusion.cpp:1195,1215,1247)
    1 all-gather of the live counts (+ 1 of the first ids under hash ownership)
    10 + 2 x 19 all-reduces of int64 limb sums      (SO3: 33 words x 10; per Gauss-Newton iteration the residual count, 2 words, which the
                                                     photometric weight needs first, then the 174 words of the joint system)
    peer-mapped images (real ranks): 1 one-word all-reduce per projection ("every owner has written")

Counted where the sharded path issues them, whichever transport carries them — so the same numbers are asserted with one process
playing all shards (transport "virtual"), with RCCL at world size 1, and between two processes over the shared-memory rendezvous
(tests/test_peer_shards_gpu.py prints them); tests/test_real_ranks_gpu.py asserts them per rank where >= 2 devices exist."""
import numpy as np
import pytest

from hrbffusion3d_amd import synth
from hrbffusion3d_amd.params import default_params

W, H = 160, 120
FRAMES = 4
LIMB_CALLS = 10 + 2 * 19
LIMB_BYTES = 8 * (10 * 33 + 19 * (2 + 174))


def _frames(n):
    return [synth.frame(k, W, H, noise=True) for k in range(n)]


def _run(setup, n=FRAMES, **params):
    from hrbffusion3d_amd.api import HRBFFusion
    K = synth.intrinsics(W, H)
    g = HRBFFusion(default_params(W, H, *K, max_surfels=1 << 17, **params))
    try:
        setup(g)
        fr = _frames(n + 1)
        g.process_frame(fr[0][0], fr[0][1])
        first = g.comm_stats(reset=True)
        for k in range(1, n + 1):
            g.process_frame(fr[k][0], fr[k][1])
        g.synchronize()
        s = g.comm_stats()
        again = g.comm_stats(reset=True)
        zero = g.comm_stats()
        return first, s, again, zero, np.ascontiguousarray(g.get_pose()).view(np.uint32).copy(), g.status()
    finally:
        g.close()


@pytest.mark.gpu
def test_a_single_map_issues_nothing(gpu_available):
    first, s, again, zero, _, status = _run(lambda g: None)
    assert status == 0 and s["transport"] == "none" and s["world"] == 1 and s["rank"] == 0 and s["frames"] == FRAMES and first["frames"] == 1
    for k, v in s.items():
        if k not in ("transport", "world", "rank", "frames"):
            assert v == 0, k
    assert again == s and zero["frames"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["ranges", "hash"])
@pytest.mark.parametrize("G", [2, 4])
def test_virtual_shards_issue_the_modelled_exchange_steps(gpu_available, G, partition):
    ref = _run(lambda g: None)
    def setup(g):
        g.comm_init(-1, G); g.map_shard_init(True, partition=partition)
    first, s, again, zero, pose, status = _run(setup)
    assert status == 0 and np.array_equal(pose, ref[4])            # and the sharded run is the single map's run
    assert s["transport"] == "virtual" and s["world"] == G and s["rank"] == 0 and s["frames"] == FRAMES
    assert s["key_min_reduce"] == 3 * FRAMES and s["key_min_reduce_bytes"] == 3 * FRAMES * 8 * W * H
    assert s["allgather"] == (2 if partition == "hash" else 1) * FRAMES and s["allgather_bytes"] == 4 * s["allgather"]
    # one process playing `G` ranks also row-shards the registration among them: the limb all-reduces are issued (and need no wire)
    assert s["limb_allreduce"] == LIMB_CALLS * FRAMES and s["limb_allreduce_bytes"] == LIMB_BYTES * FRAMES
    assert s["word_allreduce"] == 0 and s["send"] == 0 and s["recv"] == 0 and s["host_barriers"] == 0
    assert first["frames"] == 1 and first["key_min_reduce"] == 1 and first["limb_allreduce"] == 0     # frame 1: the seeding projection only
    assert again == s
    for k, v in zero.items():
        if k not in ("transport", "world", "rank"):
            assert v == 0, k


@pytest.mark.gpu
def test_registration_modes_change_the_limb_count_as_modelled(gpu_available):
    """no SO3 pre-alignment: 10 fewer; fast odometry: 3 / 5 / 4 iterations.  The geometric term alone (icp_weight 100 switches the
    photometric rows off, RGBDOdometry.cpp:807) issues the SAME sequence: the launch sequence of a frame is data independent (one
    captured graph per configuration), a term that is switched off contributes zeros"""
    vs = lambda g: g.comm_init(-1, 2)
    _, s, *_ = _run(vs, so3=0)
    assert s["limb_allreduce"] == 2 * 19 * FRAMES and s["key_min_reduce"] == 0 and s["allgather"] == 0       # rows only: the map is not sharded
    _, s, *_ = _run(vs, icp_weight=100.0)
    assert s["limb_allreduce"] == LIMB_CALLS * FRAMES
    _, s, *_ = _run(vs, fast_odom=1)                                # 3 / 5 / 4 iterations instead of 10 / 5 / 4
    assert s["limb_allreduce"] == (10 + 2 * 12) * FRAMES


@pytest.mark.gpu
def test_rccl_at_world_size_one_reports_what_the_communicator_says(gpu_available):
    """the one RCCL configuration a single-device box can run: the library asks ITS communicator for count and rank (ncclCommCount /
    ncclCommUserRank) and issues every collective of the sharded paths on it"""
    from hrbffusion3d_amd.api import HRBFFusion
    def setup(g):
        g.comm_init(0, 1, HRBFFusion.comm_unique_id()); g.map_shard_init(True, partition="hash"); g.set_row_sharding(True)
    ref = _run(lambda g: None)
    first, s, again, zero, pose, status = _run(setup)
    assert status == 0 and np.array_equal(pose, ref[4])
    assert s["transport"] == "rccl" and s["world"] == 1 and s["rank"] == 0 and s["frames"] == FRAMES
    assert s["key_min_reduce"] == 3 * FRAMES and s["limb_allreduce"] == LIMB_CALLS * FRAMES and s["limb_allreduce_bytes"] == LIMB_BYTES * FRAMES
    assert s["allgather"] == 2 * FRAMES and s["host_barriers"] == 0
    # a world of one has no peer to map: the packed-record path, whose exchange loop has nobody to send to
    assert s["send"] == 0 and s["recv"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["virtual_hash4", "rccl_world1_hash_rows"])
def test_a_sharded_bench_line_carries_the_librarys_counters_and_checks_itself(gpu_available, mode):
    """bench.py's one-sequence modes: `library_comm` (what the library's communicator reports and issued over the timed frames) and
    `sharded_self_check` (all ranks end on the same pose bits; rank 0 replays the frames on one unsharded context: same pose bits,
    same surfel count) — the two things a first run on a real node has to show, exercised on the shapes one device can run"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    extra = ["--virtual-shards", "4", "--partition", "hash"] if mode == "virtual_hash4" else ["--shard-map", "--partition", "hash", "--shard-odometry"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "3", "--surfels", "300000", "--width", "320", "--height", "240",
                          "--cpu-frames", "0", "--worst-surfels", "0", "--big-surfels", "0", "--no-cpp-shim", "--no-traffic", "--no-fit-leg"] + extra,
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    lc, sc = j["library_comm"], j["sharded_self_check"]
    assert sc == {"pose_bits_equal_on_all_ranks": True, "pose_bits_equal_one_unsharded_gpu": True, "surfel_count_equal_one_unsharded_gpu": True,
                  "frames_replayed": 11, "status_unsharded": 0}, sc
    assert lc["transport"] == ("virtual" if mode == "virtual_hash4" else "rccl")
    assert lc["world_sizes_reported_by_the_library"] == [4 if mode == "virtual_hash4" else 1] and lc["ranks_reported_by_the_library"] == [0]
    pf = lc["per_frame_rank0"]
    assert lc["per_rank"][0]["frames"] == 8 and pf["limb_allreduce"] == 48.0 and pf["key_min_reduce"] == 3.0 and pf["allgather"] == 2.0
    assert j["scaling"] == ("weak" if mode == "virtual_hash4" else "strong") and j["roofline"]["status"] == 0
