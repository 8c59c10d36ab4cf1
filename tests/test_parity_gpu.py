"""Parity of the HIP path (through the C-ABI) with the CPU oracle.  Everything here is BIT-EXACT:
integer images, float images (compared as raw bits, NaN included), the surfel map and the pose.
The arithmetic contract (include/hrbf_detmath.h) is what makes that possible; the only tolerance in
this file is on the standalone icpStep seam against an independent fp64 numpy reference."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.params import IMAGES, default_params

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gputest_pair_expected.npz")


def bits(a):
    """raw bits with NaNs canonicalised: x86 SSE produces the negative quiet NaN (0xFFC00000) for 0/0,
    gfx950 the positive one (0x7FC00000); both are NaN to every consumer, payload/sign carry no meaning."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        u = a.view(np.uint32).copy()
        u[np.isnan(a)] = 0x7FC00000
        return u
    return a.view(np.uint8)


def assert_same_state(o, g, tag="", images=None):
    for name in (images or IMAGES):
        a, b = o.get_image(name), g.get_image(name)
        ba, bb = bits(a), bits(b)
        if not np.array_equal(ba, bb):
            w = np.argwhere(ba != bb)[:5]
            raise AssertionError("%s image %s differs in %d values, first at %s: oracle %s gpu %s" % (
                tag, name, int((ba != bb).sum()), w.tolist(), [a[tuple(i)] for i in w], [b[tuple(i)] for i in w]))
    assert o.surfel_count() == g.surfel_count(), tag
    assert np.array_equal(bits(o.download_map()), bits(g.download_map())), tag + " map"
    assert np.array_equal(bits(o.get_pose()), bits(g.get_pose())), tag + " pose"


@pytest.fixture()
def pair(oracle_lib_built, gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    made = []

    def make(params, omp=True):
        o = oracle_lib_built.Oracle(params, omp=omp); g = HRBFFusion(params)
        made.append((o, g))
        return o, g
    yield make
    for o, g in made:
        o.close(); g.close()


def test_native_library_is_the_one_running(gpu_available):
    from hrbffusion3d_amd import api
    api.load_library()
    maps = open("/proc/self/maps").read()
    assert "libhrbf_mi355.so" in maps


def test_png_pair_matches_oracle_and_golden(pair, png_pair):
    """GPUTest/{1c,1d,2c,2d}.png (the reference's only real-data fixture), 640x480, four frames."""
    import hashlib
    o, g = pair(default_params(max_surfels=1 << 20))
    exp = np.load(GOLD)
    seq = [png_pair[0], png_pair[1], png_pair[0], png_pair[1]]
    for k, (rgb, d) in enumerate(seq):
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "frame %d" % (k + 1))
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
        assert o.last_icp() == g.last_icp() and o.get_weighting() == g.get_weighting()
        if k < 2:   # committed golden fixture
            tag = "f%d_" % (k + 1)
            assert np.array_equal(bits(g.get_pose()), bits(exp[tag + "pose"]))
            assert g.surfel_count() == int(exp[tag + "count"][0])
            sha = np.frombuffer(hashlib.sha256(bits(g.download_map()).tobytes()).digest(), np.uint8)
            assert np.array_equal(sha, exp[tag + "map_sha"])
            for name in IMAGES:
                sha = np.frombuffer(hashlib.sha256(bits(g.get_image(name)).tobytes()).digest(), np.uint8)
                assert np.array_equal(sha, exp[tag + "sha_" + name]), name


def test_png_pair_variants_match_golden_digests(gpu_available, png_pair):
    """the GPU path alone against the committed fixture tests/golden/gputest_pair_variants.npz (made by
    tests/golden/make_golden.py from the oracle): six registration / pre-processing options on the reference's PNG pair"""
    import hashlib
    import importlib.util
    from hrbffusion3d_amd.api import HRBFFusion
    gdir = os.path.dirname(GOLD)
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gdir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    exp = np.load(os.path.join(gdir, "gputest_pair_variants.npz"))
    sha = lambda a: np.frombuffer(hashlib.sha256(bits(a).tobytes()).digest(), np.uint8)   # NaN payloads canonicalised
    for name, kw in mg.VARIANTS.items():
        g = HRBFFusion(default_params(max_surfels=1 << 20, **kw))
        for rgb, d in png_pair:
            g.process_frame(rgb, d)
        assert np.array_equal(bits(g.get_pose()), bits(exp[name + "_pose"])), name
        assert np.array_equal(g.fuse_stats(), exp[name + "_stats"]), name
        assert np.array_equal(bits(np.array(g.last_icp(), np.float32)), bits(exp[name + "_icp"])), name
        assert np.array_equal(sha(g.download_map()), exp[name + "_map_sha"]), name
        assert np.array_equal(sha(g.get_image("PRED_VERTEX")), exp[name + "_pred_sha"]), name
        g.close()


def test_per_frame_weight_multiplier_never_recaptures_the_gn_graph(pair):
    """the reference's caller passes weightMultiplier = framesToSkip + 1, a different value on any frame (GUI/src/HRBF_fusion.cpp:225).
    The captured Gauss-Newton graph reads it from a device word, not from a launch argument: 20 frames with 1, 2, 1, 3, ... are
    bit-identical to the oracle and the loop is captured exactly twice (once per image-pointer parity), never again;
    a setter that changes the configuration does re-capture (once per parity)."""
    W, H = 320, 240
    seed = synth.seed_map(100_000, width=W)
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=seed.shape[0] + 400_000)
    o, g = pair(p)
    rgb, d, T = synth.frame(0, W, H, noise=True)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
    wm = [1.0, 2.0, 1.0, 3.0]
    for k in range(1, 21):
        rgb, d, T = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d, k, wm[k % 4]); g.process_frame(rgb, d, k, wm[k % 4])
        assert bits(np.float32(o.get_weighting())) == bits(np.float32(g.get_weighting())), k
        if k in (1, 2, 3, 4, 12, 20):
            assert_same_state(o, g, "frame %d (weightMultiplier %g)" % (k, wm[k % 4]))
        if k >= 2:
            assert g.gn_graph_captures() == 2, (k, g.gn_graph_captures())
    assert np.array_equal(bits(o.download_map()), bits(g.download_map()))
    g.set_fast_odom(True)       # 3 instead of 10 iterations on level 0: another graph (the oracle has no live setters)
    for k in range(21, 25):
        rgb, d, T = synth.frame(k, W, H, noise=True)
        g.process_frame(rgb, d, k, wm[k % 4])
    assert g.gn_graph_captures() == 4 and g.status() == 0
    assert np.linalg.norm(g.get_pose()[:3, 3] - T[:3, 3]) < 0.05


@pytest.mark.parametrize("noise", [False, True])
def test_tracked_synthetic_stream(pair, noise):
    """QVGA synthetic stream against a pre-seeded map, tracking ON: 10 frames, every image, the map
    (content AND order) and the trajectory bit-identical to the oracle."""
    W, H = 320, 240
    fx, fy, cx, cy = synth.intrinsics(W, H)
    seed = synth.seed_map(150_000, width=W)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=seed.shape[0] + 400_000)
    o, g = pair(p)
    rgb, d, T = synth.frame(0, W, H, noise=noise)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
    assert_same_state(o, g, "bootstrap")
    for k in range(1, 11):
        rgb, d, T = synth.frame(k, W, H, noise=noise)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "frame %d" % k)
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
    err = np.linalg.norm(g.get_pose()[:3, 3] - T[:3, 3])
    assert err < 0.05


def test_tracked_vga_stream_against_large_map(pair):
    """The benchmark's geometry — 640x480, tracking against a pre-seeded 400 k-surfel map, noisy input — compared with
    the oracle bit for bit over 6 frames: the 1200-workgroup reduction grids, several fuse tiles per workgroup and the
    hipGraph replay (frames 3+ replay the captured loop) are only exercised at this size."""
    W, H = 640, 480
    fx, fy, cx, cy = synth.intrinsics(W, H)
    seed = synth.seed_map(400_000, width=W)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=seed.shape[0] + 800_000)
    o, g = pair(p)
    rgb, d, T = synth.frame(0, W, H, noise=True)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
    for k in range(1, 7):
        rgb, d, T = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "vga frame %d" % k)
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
    assert np.linalg.norm(g.get_pose()[:3, 3] - T[:3, 3]) < 0.02


DATASET_K = {"tum_fr1": (synth.TUM_FR1, (640, 480)), "icl_nuim": (synth.ICL_NUIM, (640, 480)),
             "icl_nuim_neg_fy": (synth.ICL_NUIM_NEG, (640, 480)), "kinect2_512x424": (synth.KINECT2_512x424, (512, 424))}


@pytest.mark.parametrize("name", list(DATASET_K))
def test_tracked_stream_with_dataset_intrinsics(pair, name):
    """BASELINE configs 2 / 3 geometry: streams rendered with the TUM fr1 (517.3, 516.5, 318.6, 255.3) and ICL-NUIM
    (481.2, +-480, 319.5, 239.5) intrinsics at 640x480 and with a non-4:3 sensor (512x424, fx != fy, off-centre) —
    noisy depth, 32 tracked frames against a pre-seeded map.  Pose and surfel count bit-identical to the oracle after
    EVERY frame, every image and the whole map (content and order) at frames 1, 2, 8, 16, 24 and 32, and the
    trajectory stays near the analytic ground truth.  With fx == fy and a centred principal point (every other test)
    a swapped fx/fy or cx/cy, or a W-for-H slip, would cancel; here it cannot (see also test_projection_site_kats)."""
    K, (W, H) = DATASET_K[name]
    seed = synth.seed_map(400_000, width=W, K=K)
    p = default_params(W, H, *K, max_surfels=seed.shape[0] + 40 * (W // 2) * (H // 2))
    o, g = pair(p)
    rgb, d, T = synth.frame(0, W, H, noise=True, K=K)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
    worst = 0.0
    for k in range(1, 33):
        rgb, d, T = synth.frame(k, W, H, noise=True, K=K)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert np.array_equal(bits(o.get_pose()), bits(g.get_pose())), "%s frame %d pose" % (name, k)
        assert o.surfel_count() == g.surfel_count(), "%s frame %d count" % (name, k)
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
        if k in (1, 2, 8, 16, 24, 32):
            assert_same_state(o, g, "%s frame %d" % (name, k))
        worst = max(worst, float(np.linalg.norm(g.get_pose()[:3, 3] - T[:3, 3])))
    assert worst < 0.03, worst
    assert g.status() == 0


@pytest.mark.parametrize("K", ["tum_quarter", "skewed"])
def test_projection_site_kats(gpu_available, K):
    """the HIP path against the independent numpy known answers of tests/kat_projection.py (the same checks pin the
    oracle in tests/test_intrinsics_kat.py): back-projection, index-map projection and prediction rays with fx != fy and
    an off-centre principal point; each must also DISAGREE with the expectation evaluated for swapped intrinsics."""
    import kat_projection as kp
    from hrbffusion3d_amd.api import HRBFFusion
    K = kp.K_TUM_Q if K == "tum_quarter" else kp.K_SKEWED
    W, H = 160, 120
    g = HRBFFusion(default_params(W, H, *K, max_surfels=1 << 15))
    out = kp.run_back_projection(g, W, H, K)
    assert kp.back_projection_error(out, K, W, H) < 2e-6
    assert kp.back_projection_error(out, kp.swapped(K), W, H) > 1e-2
    T, m, px, py, pc = kp.make_projection_case(W, H, K)
    out = kp.run_projection(g, T, m)
    bad, err, hits = kp.projection_mismatch(out, T, m, K, W, H)
    assert hits == px.size and bad == 0 and err < 5e-6
    assert kp.projection_mismatch(out, T, m, kp.swapped(K), W, H)[0] > px.size // 2
    out = kp.run_prediction_rays(g, W, H, K)
    err, plane = kp.prediction_ray_error(out, K, W, H)
    assert err < 2e-6 and plane < 1e-4
    assert kp.prediction_ray_error(out, kp.swapped(K), W, H)[0] > 1e-2
    g.close()


def test_long_sequence_trajectory_and_determinism(gpu_available):
    """150 noisy QVGA frames from an empty map (frame 0 seeds it): the trajectory stays near the ground truth (ATE, the
    north star's other metric, against the stream's analytic poses) and a second run reproduces pose and map bit for
    bit (the property that lets the short oracle comparisons stand for long sequences).
    The bound is loose on purpose: while the map is young the model is the filled-in previous frame and the joint
    registration is dominated by the photometric rows, whose nearest-texel residual (reduce.cu:1027-1046) resolves the
    pixel-or-two of inter-frame motion only to about half a pixel (~5 mm / 0.1 deg per frame at QVGA); the offset stops
    growing once the model is predicted from stable surfels, at about 6 cm at QVGA (2 cm at VGA).  Located by
    tests/test_tracking_accuracy.py and tools/probes/rgb_term_emulation.py."""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 320, 240
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 21)
    frames = [synth.frame(k, W, H, noise=True) for k in range(150)]

    def run():
        g = HRBFFusion(p)
        g.set_pose(frames[0][2])
        est = []
        for rgb, d, _ in frames:
            g.process_frame(rgb, d)
            est.append(g.get_pose())
        m = g.download_map()
        g.close()
        return est, m

    est, m = run()
    gt = [f[2] for f in frames]
    ate = synth.ate_rmse(est, gt)
    assert 0.03 < ate < 0.08, ate          # 0.0596 in round 4 (0.058 in round 3): the photometric term's saturating offset, see the docstring
    assert len(m) > 0.8 * W * H
    est2, m2 = run()
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(est, est2))
    assert np.array_equal(bits(m), bits(m2))


def test_icp_only_registration_tracks_the_same_stream_to_a_centimetre(gpu_available):
    """the stream of test_long_sequence_trajectory_and_determinism with icp_weight 100 instead of 10, which switches the
    photometric term off (`rgb = rgbOnly || icpWeight < 100`, RGBDOdometry.cpp:807): the offset goes from ~6 cm to ~1 cm — the accuracy of the rest of the
    pipeline (pre-processing, fusion, prediction, ICP) on this stream, and the evidence that the loose bound of the default
    configuration is the photometric term's"""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 320, 240
    fx, fy, cx, cy = synth.intrinsics(W, H)
    frames = [synth.frame(k, W, H, noise=True) for k in range(150)]
    ate = {}
    for w in (10.0, 100.0):
        g = HRBFFusion(default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 21, icp_weight=w))
        g.set_pose(frames[0][2])
        est = []
        for rgb, d, _ in frames:
            g.process_frame(rgb, d)
            est.append(g.get_pose())
        g.close()
        ate[w] = synth.ate_rmse(est, [f[2] for f in frames])
    assert ate[100.0] < 0.02 and ate[100.0] < ate[10.0] / 3.0, ate


@pytest.mark.parametrize("variant", ["gauss_filter", "central_diff_normals", "no_so3_no_pyramid", "conf_eval", "rgb_only",
                                     "icp_only", "corr_search", "sparse_icp_corr_search", "clean_window_1", "clean_window_4", "clean_window_2_25",
                                     "frame_to_frame_rgb", "rgb_grad_weight", "icp_unweighted", "predict_small",
                                     "curv_window_2", "thresholds"])
def test_parameter_variants(pair, variant):
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    kw = {"gauss_filter": dict(use_bilateral=0), "central_diff_normals": dict(normal_estimation_pca=0.0),
          "no_so3_no_pyramid": dict(so3=0, pyramid=0, fast_odom=1), "conf_eval": dict(use_conf_eval=1),
          "rgb_only": dict(rgb_only=1), "icp_only": dict(icp_weight=100.0), "corr_search": dict(icp_use_corr_search=1),
          "sparse_icp_corr_search": dict(use_sparse_icp=1, icp_use_corr_search=1),
          "clean_window_1": dict(clean_window_multiplier=1.0), "clean_window_4": dict(clean_window_multiplier=4.0),
          "clean_window_2_25": dict(clean_window_multiplier=2.25),      # 2 wm not an integer: ceil(4.5) = 5 samples per axis
          "frame_to_frame_rgb": dict(frame_to_frame_rgb=1), "rgb_grad_weight": dict(rgb_use_grad_weight=1),
          "icp_unweighted": dict(icp_use_weighted=0),
          "predict_small": dict(predict_window_multiplier=2.0, predict_min_neighbors=4, predict_max_neighbors=6),
          "curv_window_2": dict(curv_estimation_window=2.0),
          "thresholds": dict(confidence_threshold=2.0, depth_cutoff=2.5, curv_valid_threshold=150.0, dense_enough_thresh=0.99,
                             predict_conf_threshold=1.0, init_radius_multiplier=3.0)}[variant]
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17, **kw)
    o, g = pair(p)
    for k in range(5):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "%s frame %d" % (variant, k))


def test_random_parameter_combinations(oracle_lib_built, gpu_available):
    """every switch drawn at random TOGETHER (tests/gpu_fuzz_params.py; the variants above turn one at a time): the first 10 draws
    of seed 1 — windowed search with the sparse variant without the pyramid, an empty depth image in the middle, ... — HIP == oracle
    on every image, the map and the pose after every frame.  profiles/r06_param_fuzz.txt holds a run of many hundred draws."""
    import gpu_fuzz_params as F
    for i in range(10):
        kw, plan = F.draw(1, i)
        r = F.run_one(oracle_lib_built, kw, plan)
        assert r is None, (i, r, kw, plan)
    # the one difference the long runs found (profiles/r06_param_fuzz.txt): a surfel merged at total confidence 0 (confidence evaluation
    # on, every weight 0): position, normal and colour are 0 / 0, and the colour WORD is an int conversion of NaN — undefined in C,
    # -2^31 on the host, 0 on the device.  Stated since (hd_cvt_i32, hrbf_detmath.h); sharded as drawn and on the single map
    kw = {'use_bilateral': 0, 'normal_estimation_pca': 0.0, 'so3': 1, 'pyramid': 0, 'fast_odom': 0, 'use_conf_eval': 1, 'conf_eval_epsilon': 2500.0,
          'rgb_only': 0, 'icp_weight': 25.0, 'icp_use_corr_search': 0, 'icp_search_radius': 2, 'use_sparse_icp': 0, 'clean_window_multiplier': 2.0,
          'frame_to_frame_rgb': 1, 'rgb_use_grad_weight': 0, 'icp_use_weighted': 1, 'icp_curv_weight_lambda': 10.0, 'predict_window_multiplier': 2.0,
          'predict_min_neighbors': 6, 'curv_estimation_window': 3.0, 'curv_valid_threshold': 100.0, 'confidence_threshold': 5.0, 'depth_cutoff': 5.0,
          'dense_enough_thresh': 0.99, 'predict_conf_threshold': 1.0, 'init_radius_multiplier': 4.0, 'max_depth_processed': 20.0, 'predict_max_neighbors': 8}
    plan = {'size': (160, 120), 'start': 113, 'step': 1, 'frames': 3, 'noise': 1, 'odd_frame': 'far', 'odd_at': 2, 'shards': 3, 'partition': 'ranges',
            'row_sharding': 1, 'rebalance_at': None, 'seed_map': 0, 'tick_jump': None}
    for shards in (3, 0):
        r = F.run_one(oracle_lib_built, kw, dict(plan, shards=shards))
        assert r is None, (shards, r)
    o = oracle_lib_built.Oracle(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 17, **kw), omp=True)
    for k in range(2):
        rgb, d, _ = synth.frame(113 + k, 160, 120, noise=True)
        o.process_frame(rgb, d)
    m = o.download_map(); o.close()
    nan = np.isnan(m[:, 0])
    assert nan.sum() >= 1 and (m[nan, 4] == 0).all() and (m[nan, 3] == 0).all()          # the case does occur in this run


def test_adversarial_pixels_at_the_stage_seams(oracle_lib_built, gpu_available):
    """NaN / inf / negative / denormal pixels in the images one stage reads (tests/gpu_fuzz_stages.py): 40 trials of seed 1 and the two
    trials of seed 3 that differed before uint() of a negative or huge float was stated (hd_cvt_u32) — the init time of a predicted
    pixel was 0xFFFFFFFF on the host and 0 on the device.  profiles/r06_stage_fuzz.txt: 3 000 trials, 0 mismatches."""
    import gpu_fuzz_stages as S
    for seed, i in [(1, k) for k in range(40)] + [(3, 80), (3, 141)]:
        r = S.trial(oracle_lib_built, seed, i)
        assert r is None, (seed, i, r)


def test_no_entry_point_dies_on_zero_arguments(gpu_available):
    """every handle-taking entry point on a LIVE context (two frames in) with all other arguments zero / NULL, one child process each
    (tests/gpu_probe_abi_zero_args.py): an error code or a harmless success, never a signal, and the context processes the next frame.
    hrbf_icp_step read its host matrices unchecked until this survey called it."""
    import gpu_probe_abi_zero_args as Z
    eps, bad = Z.survey(warm_states=(1,))
    assert len(eps) >= 70 and not bad, bad


def test_wrong_values_at_the_boundary_are_refused_not_fatal(gpu_available):
    """tests/gpu_probe_abi_bad_values.py: sizes that do not match, enums out of range, counts beyond the capacity, NaN / zero cameras and
    units, windows the kernels are not built for — 36 calls on a live context and 21 creations, one child process each: an error status
    where one is due, never a signal, and the next frame runs.  The first survey found hrbf_update_model reading n x 64 bytes for any n
    (now 0 <= n <= 1200, the reference's texture) and hrbf_create accepting fx = 0, depth_scale = 0 and NaN parameters."""
    import gpu_probe_abi_bad_values as B
    findings = B.survey()
    assert not findings, findings


def test_random_api_call_sequences(oracle_lib_built, gpu_available):
    """10-24 random API calls per context (tests/gpu_fuzz_api.py): frames by host and device pointer, pose / tick / weighting / the
    run-time switches set in between, maps re-uploaded re-ordered or thinned, updateModel, submap masks, stages, images set back
    unchanged, timing and the shard cut changed on the library's side only — compared with the oracle after every call.  Trials 0-5 of
    seed 1 and the two trials of seed 5 that found something (below).  profiles/r06_api_fuzz.txt: 600 trials, 0 mismatches."""
    import gpu_fuzz_api as A
    for seed, i in [(1, k) for k in range(6)] + [(5, 304), (5, 305)]:
        r = A.trial(oracle_lib_built, seed, i)
        assert r is None, (seed, i, r)


def test_timings_read_before_any_timed_frame_leave_no_error_behind(pair):
    """found by the API fuzzer: hrbf_get_timings right after hrbf_enable_timing queries events no frame has recorded; the query's
    'invalid resource handle' was tolerated but stayed in the runtime's last-error slot, and the NEXT frame's launch check reported
    it as its own (status -2).  Same for the fuse ring read before a frame."""
    W, H = 160, 120
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 17)
    o, g = pair(p)
    g.enable_timing(1)
    assert np.all(g.timings() == 0)
    g.fuse_ring_parts(4)
    for k in range(3):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "frame %d" % k)
    assert g.status() == 0 and g.timings()[5] > 0


def test_stage_seams_refuse_raw_images_the_context_does_not_hold(pair):
    """found by the API fuzzer: after hrbf_process_frame_device (the caller's buffers are read in place and not kept) a stage seam that
    reads the raw frame ran — silently — on the images of the last HOST-pointer frame.  It now refuses until hrbf_upload_frame()."""
    import torch
    from hrbffusion3d_amd.api import HrbfError
    W, H = 160, 120
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 17)
    o, g = pair(p)
    for k in range(3):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d)
        if k < 2:
            g.process_frame(rgb, d)
        else:
            tr, td = torch.from_numpy(rgb).cuda(), torch.from_numpy(d.view(np.int16)).cuda()
            g.process_frame_device(tr.data_ptr(), td.data_ptr(), 0); g.synchronize()
            del tr, td
    assert_same_state(o, g, "device-pointer frame")
    for s in ("FILLIN", "FILTER_DEPTH", "FUSE", "INITIALISE"):
        with pytest.raises(HrbfError, match="hrbf_upload_frame"):
            g.run_stage(s)
    for x in (o, g):
        x.run_stage("PREDICT_HRBF")             # reads no raw image: runs
    for x in (o, g):
        x.upload_frame(rgb, d); x.run_stage("FILLIN")
    assert_same_state(o, g, "fill-in after the upload")


@pytest.mark.parametrize("size", [(320, 240), (1280, 960)])
def test_other_resolutions(pair, size):
    """QVGA (BASELINE config 1 geometry) and 1280x960 (config 5 geometry): same kernels, other grid shapes — the
    level-2 SO3 grid is 19 / 300 workgroups, the fuse tiles and the pyramids change size."""
    W, H = size
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << (19 if W < 1000 else 22))
    o, g = pair(p)
    for k in range(3):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "%dx%d frame %d" % (W, H, k))
    assert g.surfel_count() > 0.5 * W * H


def test_edge_cases_empty_and_invalid_depth(pair):
    """all-zero depth (no valid pixel), depth beyond the cut-off, and a frame after an empty one"""
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 16)
    o, g = pair(p)
    rgb = scenes.gray_rgb(W, H)
    zero = np.zeros((H, W), np.uint16)
    far = np.full((H, W), 60000, np.uint16)
    good = synth.frame(0, W, H)[1]
    ragged = good.copy(); ragged[::3, ::2] = 0; ragged[:7] = 0; ragged[:, -5:] = 0
    for k, d in enumerate([zero, far, good, ragged, zero, good]):
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "edge frame %d" % k)
    assert g.surfel_count() > 0


def test_two_contexts_interleaved(gpu_available):
    """two contexts in one process (different resolutions, their own streams, graphs and maps) fed alternately give the
    same bits as each of them run alone: nothing is shared between contexts but the library's constant tables"""
    from hrbffusion3d_amd.api import HRBFFusion

    def params(W, H):
        fx, fy, cx, cy = synth.intrinsics(W, H)
        return default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 18)

    sizes = [(160, 120), (320, 240)]
    frames = {sz: [synth.frame(k, sz[0], sz[1], noise=True) for k in range(6)] for sz in sizes}
    alone = {}
    for sz in sizes:
        g = HRBFFusion(params(*sz))
        for rgb, d, _ in frames[sz]:
            g.process_frame(rgb, d)
        alone[sz] = (g.get_pose(), g.download_map(), g.get_image("PRED_VERTEX"))
        g.close()
    gs = {sz: HRBFFusion(params(*sz)) for sz in sizes}
    for k in range(6):
        for sz in sizes:
            gs[sz].process_frame(frames[sz][k][0], frames[sz][k][1])
    for sz in sizes:
        pose, m, pv = alone[sz]
        assert np.array_equal(bits(pose), bits(gs[sz].get_pose()))
        assert np.array_equal(bits(m), bits(gs[sz].download_map()))
        assert np.array_equal(bits(pv), bits(gs[sz].get_image("PRED_VERTEX")))
        gs[sz].close()


def test_two_contexts_concurrent_with_a_device_filling_co_runner(gpu_available):
    """1000 iterations of two contexts enqueued back to back WITHOUT synchronisation (each on its own stream, inputs
    resident in HBM, so their kernels really overlap on the device) while a third stream keeps every CU busy with
    large GEMMs.  The fuse pass (ticketed tiles, tile_done epochs) and the persistent SO3 kernel (ticketed chunks) must
    not depend on their workgroups being co-resident: no hang, no status bit, and poses / maps bit-identical to each
    context run alone.  Noisy QVGA streams against 150 k-surfel maps: ~110 fuse tiles per frame, surfels removed in
    mid-array on most frames (the in-place compaction's wait chain is exercised), 75 SO3 chunks."""
    import torch
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 320, 240
    K = synth.intrinsics(W, H)
    NF, ITERS = 40, 1000
    tri = lambda i: (i % (2 * NF - 2)) if (i % (2 * NF - 2)) < NF else (2 * NF - 2) - (i % (2 * NF - 2))    # 0..39..1 0..: continuous motion
    streams = {}
    for name, off in (("a", 0), ("b", 60)):
        fr = [synth.frame(off + k, W, H, noise=True) for k in range(NF)]
        streams[name] = dict(frames=fr, seed=synth.seed_map(150_000, width=W),
                             dev=[(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1].view(np.int16)).cuda()) for f in fr])

    def start(name):
        st = streams[name]
        g = HRBFFusion(default_params(W, H, *K, max_surfels=st["seed"].shape[0] + 600_000))
        g.upload_map(st["seed"]); g.set_pose(st["frames"][0][2]); g.bootstrap(st["frames"][0][0], st["frames"][0][1])
        return g

    def feed(g, name, i):
        d = streams[name]["dev"][tri(i)]
        g.process_frame_device(d[0].data_ptr(), d[1].data_ptr(), i)

    alone = {}
    moved_frames = 0
    for name in streams:
        g = start(name)
        g.enable_timing(2)
        for i in range(1, ITERS + 1):
            feed(g, name, i)
        g.synchronize()
        _, _, st8 = g.fuse_ring_parts(1024)
        moved_frames += int((st8[:, 6] > 0).sum())
        assert g.status() == 0
        alone[name] = (g.get_pose(), g.download_map())
        g.close()
    assert moved_frames > 200      # the moving branch of the compaction ran on a good share of the frames
    side = torch.cuda.Stream()
    a = torch.randn(6144, 6144, device="cuda"); b = torch.randn(6144, 6144, device="cuda"); c = torch.empty_like(a)
    gs = {name: start(name) for name in streams}
    for i in range(1, ITERS + 1):
        if i % 25 == 1:
            with torch.cuda.stream(side):
                for _ in range(6):
                    torch.mm(a, b, out=c)          # a few hundred workgroups with big LDS tiles: keeps the CUs occupied
        for name in streams:
            feed(gs[name], name, i)
    for name in streams:
        gs[name].synchronize()
    torch.cuda.synchronize()
    for name in streams:
        assert gs[name].status() == 0
        pose, m = alone[name]
        assert np.array_equal(bits(pose), bits(gs[name].get_pose())), name
        assert np.array_equal(bits(m), bits(gs[name].download_map())), name
        gs[name].close()


def test_stale_surfel_purge_matches_oracle(pair):
    """copy_unstable.vert:159-165 — unstable surfels not seen for 200 frames are dropped — against the oracle, with the
    removals starting in the first tile so that every later tile of the in-place compaction moves: 6 % of a 150 k-surfel
    map is unstable, the clock jumps by 300 frames after the bootstrap, five tracked noisy frames follow.  Images, map
    (content and order), statistics and pose bit for bit; the pass reports that it moved (nearly) the whole map."""
    W, H = 320, 240
    K = synth.intrinsics(W, H)
    seed = synth.seed_map(150_000, width=W)
    n = len(seed)
    stale = np.zeros(n, bool)
    stale[np.random.default_rng(5).choice(n, int(0.06 * n), replace=False)] = True
    stale[0:64:3] = True
    seed[stale, 3] = 1.0
    p = default_params(W, H, *K, max_surfels=n + 300_000)
    o, g = pair(p)
    g.enable_timing(2)
    rgb, d, T = synth.frame(0, W, H, noise=True)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d); x.set_tick(300)
    for k in range(1, 6):
        rgb, d, T = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "purge frame %d" % k)
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
        if k == 1:
            st = g.fuse_stats().astype(np.int64)
            assert st[0] + st[2] - st[3] > 0.8 * stale.sum()
    _, _, st8 = g.fuse_ring_parts(5)
    assert st8[0][6] > 0.9 * n and g.status() == 0


def test_clean_pass_classes_on_a_map_that_is_mostly_elsewhere(pair):
    """Pass A of the clean pass takes its first decision per surfel from a class byte the projection in front of it leaves
    (stable and outside the frustum: kept unread; unstable and outside: position + colour/time; in view: + normal/radius, all
    in one round — DESIGN 5 item 2).  A map whose surfels are mostly NOT where the camera looks — behind it, beyond the depth
    limit, far to the side, stable and unstable, fresh and stale, interleaved with a seeded room so that every wave holds all
    three classes — through tracked noisy frames: frame 1 runs the full check after the upload (no classes), the later ones
    the classified path.  Images, map (content and order), statistics and pose bit for bit against the oracle."""
    W, H = 320, 240
    K = synth.intrinsics(W, H)
    room = synth.seed_map(60_000, width=W)
    rng = np.random.default_rng(17)
    n_x = 90_000
    extra = room[rng.integers(0, len(room), n_x)].copy()
    extra[:, 0:3] = rng.uniform(-12.0, 12.0, (n_x, 3)).astype(np.float32)          # anywhere in a 24 m box around the room
    extra[:, 3] = np.where(rng.random(n_x) < 0.4, 1.0, rng.uniform(5.0, 20.0, n_x)).astype(np.float32)   # 40 % unstable
    extra[:, 7] = rng.choice(np.array([1.0, 50.0, 250.0], np.float32), n_x)        # last seen: stale (tick 300) or not
    extra[:, 6] = 1.0
    seed = np.concatenate([room, extra])
    seed = seed[rng.permutation(len(seed))]
    n = len(seed)
    p = default_params(W, H, *K, max_surfels=n + 200_000)
    o, g = pair(p)
    rgb, d, T = synth.frame(0, W, H, noise=True)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d); x.set_tick(300)
    removed = 0
    for k in range(1, 6):
        rgb, d, T = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "classes frame %d" % k)
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
        st = g.fuse_stats().astype(np.int64)
        removed += int(st[0] + st[2] - st[3])
    assert removed > 0.2 * n_x * 0.4 and g.status() == 0      # the stale unstable surfels outside the frustum went
    # the stage API's clean pass (position first, no classes) on the same state gives the same map as the frame path did
    for x in (o, g):
        x.run_stage("PREDICT_INDICES"); x.run_stage("CLEAN")
    assert_same_state(o, g, "stage-API clean after the classified frames", images=[])


def test_upload_of_a_larger_map_after_frames_ran(pair):
    """a count read-back armed by earlier frames must not survive hrbf_upload_map / initialise — it
    describes the OLD map, and folding it into the host bound later would size the next fuse pass (LDS tile counts,
    tile counters) for fewer surfels than the device holds.  Frames from an empty map (small count), then a 60 k-surfel
    map is uploaded into the running context, then more frames: bit-identical to the oracle, no status bit."""
    W, H = 160, 120
    K = synth.intrinsics(W, H)
    p = default_params(W, H, *K, max_surfels=1 << 18)
    o, g = pair(p)
    for k in range(3):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)      # no synchronisation on the GPU side: read-backs stay armed
    seed = synth.seed_map(60_000, width=W)
    T = synth.frame(3, W, H)[2]
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T)
    for k in range(3, 9):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
    assert_same_state(o, g, "after upload")
    assert np.array_equal(o.fuse_stats(), g.fuse_stats())
    assert g.status() == 0 and g.surfel_count() > 60_000


@pytest.mark.parametrize("cap", [9000, 20000])
def test_map_at_capacity(pair, cap):
    """maximum size: a map whose capacity is hit by the seed frame (cap 9000 < first frame's surfels) or by the
    appends of later frames (cap 20000): the count saturates at the capacity, nothing is written past it, and images,
    map and pose stay bit-identical to the oracle, which applies the same clamp"""
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=cap)
    o, g = pair(p)
    for k in range(8):
        rgb, d, _ = synth.frame(3 * k, W, H, noise=True)     # larger steps: more new surface per frame
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "cap %d frame %d" % (cap, k))
        assert g.surfel_count() <= cap
    assert g.surfel_count() > 0.9 * cap if cap == 9000 else True


def test_stage_seams_in_isolation(pair):
    """operator-level seams: each stage run alone on injected inputs"""
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    o, g = pair(p)
    rgb, d, T = synth.frame(3, W, H, noise=True)
    for x in (o, g):
        x.upload_frame(rgb, d)
        x.run_stage("FILTER_DEPTH"); x.run_stage("METRICISE")
    assert_same_state(o, g, "filter", ["DEPTH_FILTERED", "DEPTH_METRIC", "DEPTH_METRIC_FILTERED"])
    for x in (o, g):
        x.run_stage("VERTEX_NORMAL_RADIUS")
    assert_same_state(o, g, "vnr", ["VERTEX_RAW", "VERTEX_FILTERED", "NORMAL", "NORMAL_PCA", "RADIUS"])
    for x in (o, g):
        x.run_stage("CURVATURE")
    assert_same_state(o, g, "curv", ["CURV1", "CURV2", "GRADIENT_MAG", "NORMAL"])
    for x in (o, g):
        x.set_weighting(0.75); x.run_stage("CONFIDENCE")
    assert_same_state(o, g, "conf", ["CONFIDENCE"])
    seed = synth.seed_map(60_000, width=W)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.set_tick(5); x.run_stage("PREDICT_INDICES")
    assert_same_state(o, g, "indices", [n for n in IMAGES if n.startswith("INDEX")])
    for x in (o, g):
        x.run_stage("PREDICT_HRBF"); x.run_stage("FILLIN")
    assert_same_state(o, g, "predict", [n for n in IMAGES if n.startswith(("PRED", "FILL"))])
    for x in (o, g):
        x.run_stage("FUSE")
    assert np.array_equal(o.fuse_stats()[:2], g.fuse_stats()[:2])
    assert_same_state(o, g, "fuse", ["INDEX"])
    for x in (o, g):
        x.run_stage("PREDICT_INDICES"); x.run_stage("CLEAN")
    assert_same_state(o, g, "clean", ["INDEX"])
    assert np.array_equal(o.fuse_stats(), g.fuse_stats())
    # idempotence: a second clean without a fuse appends nothing and keeps the map
    before = g.download_map()
    for x in (o, g):
        x.run_stage("PREDICT_INDICES"); x.run_stage("CLEAN")
    assert_same_state(o, g, "clean twice", ["INDEX"])
    assert g.fuse_stats()[2] == 0 and len(g.download_map()) <= len(before)


def test_frame_path_with_and_without_the_fused_per_pixel_passes(pair):
    """process_frame runs level 0 of the registration pyramids from the curvature kernel's tail and the fill-in from the
    ray cast's — unless the shouldFillIn flag on the device may be stale, e.g. after hrbf_set_image, when it falls back to
    the separate kernels.  Alternating between the two must not show anywhere."""
    W, H = 320, 240
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 19)
    o, g = pair(p)
    for k in range(10):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        if k in (3, 4, 7):   # same content written back: marks the prediction as touched
            for x in (o, g):
                x.set_image("PRED_TIME", x.get_image("PRED_TIME"))
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "frame %d" % k)


def test_fuse_ring_stride(gpu_available):
    """hrbf_set_fuse_ring_stride: only frames whose time stamp is a multiple of the stride are bracketed by events"""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    g = HRBFFusion(default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 16))
    g.enable_timing(2); g.set_fuse_ring_stride(3); g.reset_fuse_ring()
    for k in range(12):
        rgb, d, _ = synth.frame(k, W, H)
        g.process_frame(rgb, d)
    mm, ms, st = g.fuse_ring_parts(64)
    assert len(ms) == 4 and (ms > 0).all() and (st[:, 3] > 0).all()       # frames 0 (first: initialise only) .. 11 -> ticks 3, 6, 9 (+ one of 0 / 12)
    g.set_fuse_ring_stride(1); g.reset_fuse_ring()
    for k in range(12, 16):
        rgb, d, _ = synth.frame(k, W, H)
        g.process_frame(rgb, d)
    assert len(g.fuse_ring_parts(64)[1]) == 4
    g.close()


def test_sqrt_shortcut_is_exhaustively_exact(gpu_available):
    """k_predict_hrbf replaces the compiler's sqrtf expansion by v_sqrt_f32 + two residual tests (no rescaling of tiny
    arguments, no 0 / inf re-check).  Checked here over EVERY non-negative finite float on the device under test."""
    from hrbffusion3d_amd.api import HRBFFusion
    g = HRBFFusion(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 12))
    h = g.probe_sqrt_rounding()
    g.close()
    assert int(h[:4].sum()) == 0x7f800000, h
    assert h[3] <= 0x7fffff, h     # further off than one ulp only for (some of) the 2^23 - 1 denormal arguments: covered by h[5]
    assert h[4] == 0 and h[5] == 0, h


def test_exp_scaling_shortcut_is_exhaustively_exact(gpu_available):
    """the bilateral filter's exp scales by 2^k with v_ldexp_f32 instead of hd_expf's two multiplications: same bits for
    every polynomial value in [0.5, 2) and every k the filter can produce"""
    from hrbffusion3d_amd.api import HRBFFusion
    g = HRBFFusion(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 12))
    bad, total = g.probe_exp_scaling()
    g.close()
    assert total == (1 << 24) * 161 and bad == 0, (bad, total)


def test_unscaled_division_matches_the_compilers(gpu_available):
    """2^32 pseudo-random operand pairs over the tame ranges of k_curvature: the FMA division without v_div_scale /
    v_div_fixup gives the compiler's correctly rounded quotient every time (scalar and packed form)"""
    from hrbffusion3d_amd.api import HRBFFusion
    g = HRBFFusion(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 12))
    bad, total = g.probe_division()
    g.close()
    assert total == 1 << 32 and bad == 0, (bad, total)


@pytest.mark.parametrize("kind", ["tame_near_duplicates", "negative_zero", "tiny_coordinate", "huge_coordinate", "radius_zero",
                                  "radius_tiny", "radius_huge", "radius_nan", "normal_huge", "position_nan"])
def test_curvature_with_degenerate_texels(pair, kind):
    """k_curvature runs packed arithmetic with unscaled divisions on tiles whose texels are all within tame ranges and the
    literal code otherwise (and, inside a tame tile, for neighbours closer than 2^-20 support radii).  Each kind poisons
    a few texels of the vertex / normal images the stage reads."""
    W, H = 160, 120
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 12)
    o, g = pair(p)
    rgb, d, _ = synth.frame(3, W, H, noise=True)
    for x in (o, g):
        x.upload_frame(rgb, d)
        for st in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS"):
            x.run_stage(st)
    v = o.get_image("VERTEX_FILTERED").copy(); n = o.get_image("NORMAL").copy()
    spots = [(30, 40), (31, 41), (60, 80), (61, 80), (90, 120), (100, 20), (17, 17), (64, 64), (65, 65)]
    for (y, x) in spots:
        if kind == "tame_near_duplicates":      # two texels one ulp apart at small coordinates: d2 / T^2 below 2^-40
            v[y, x, :3] = (0.01, 0.01, v[y, x, 2] if v[y, x, 2] > 0.3 else 1.0)
            v[y, x + 1, :3] = (np.nextafter(np.float32(0.01), np.float32(1)), 0.01, v[y, x, 2])
            n[y, x + 1] = n[y, x]
        elif kind == "negative_zero": v[y, x, 0] = -0.0
        elif kind == "tiny_coordinate": v[y, x, 1] = 1e-30
        elif kind == "huge_coordinate": v[y, x, 0] = 1e12
        elif kind == "radius_zero": n[y, x, 3] = 0.0
        elif kind == "radius_tiny": n[y, x, 3] = 1e-9
        elif kind == "radius_huge": n[y, x, 3] = 1e5
        elif kind == "radius_nan": n[y, x, 3] = float("nan")
        elif kind == "normal_huge": n[y, x, :3] *= 1e20
        elif kind == "position_nan": v[y, x, 2] = float("nan")
    for x in (o, g):
        x.set_image("VERTEX_FILTERED", v); x.set_image("NORMAL", n); x.run_stage("CURVATURE")
    assert_same_state(o, g, kind, ["CURV1", "CURV2", "GRADIENT_MAG", "NORMAL"])
    assert (np.abs(g.get_image("CURV1")[..., 3]) < 300).sum() > 1000      # the rest of the image has curvatures


@pytest.mark.parametrize("radius", [0.0, float("nan"), 1e-25, 1e25, float("inf"), 0.02])
def test_predict_with_degenerate_texels(pair, radius):
    """The ray-cast kernel runs a select-free inner loop on tiles whose texels are all finite and tame and the literal one
    otherwise.  Poisoned index-map texels: a texel ON the optical axis of the principal pixel (its first sample coincides
    with the centre: d2 == 0, the getWeightD special case) with a zero / NaN / tiny / huge / infinite support radius, and
    non-finite or huge positions and normals scattered over other tiles."""
    W, H = 160, 120
    p = default_params(W, H, 150.0, 150.0, 80.5, 60.5, max_surfels=1 << 17)
    o, g = pair(p)
    _, _, T = synth.frame(3, W, H)
    seed = synth.seed_map(60_000, width=W)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.set_tick(5); x.run_stage("PREDICT_INDICES")
    v = o.get_image("INDEX_VERTCONF").copy(); n = o.get_image("INDEX_NORMRAD").copy()
    z0 = float(np.median(v[50:70, 70:90, 2][v[50:70, 70:90, 2] > 0.1]))
    v[60, 80] = (0.0, 0.0, z0, 50.0); n[60, 80] = (0.0, 0.0, 1.0, radius)
    v[61, 81, 3] = 50.0; n[61, 81, 3] = radius                                   # the same radius off the axis
    v[20, 30, 0] = 1e20; v[25, 100, 1] = float("nan"); v[90, 40, 2] = float("inf"); v[100, 120, :3] *= 1e16
    n[30, 130, :3] *= 1e35; n[95, 20, 0] = float("nan"); n[15, 75, 3] = -0.0
    for x in (o, g):
        x.set_image("INDEX_VERTCONF", v); x.set_image("INDEX_NORMRAD", n); x.run_stage("PREDICT_HRBF"); x.run_stage("FILLIN")
    assert_same_state(o, g, "radius %r" % radius, [k for k in IMAGES if k.startswith(("PRED", "FILL"))])
    assert (g.get_image("PRED_VERTEX")[..., 2] > 0).sum() > 1000               # the rest of the image is still predicted


def test_named_map_operators(pair):
    """GlobalModel::{initialise,fuse,clean} / IndexMap::{predictIndices,predictHRBF} under their own names with the
    explicit pose / time / cut-off arguments of the reference (GlobalModel.h:50-107, IndexMap.h:43-68) against the
    oracle driven through set_pose / set_tick / run_stage: the map pipeline of one frame, operator by operator."""
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    o, g = pair(p)
    pre = ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE", "CONFIDENCE")
    rgb, d, T0 = synth.frame(0, W, H, noise=True)
    for x in (o, g):
        x.upload_frame(rgb, d)
        for st in pre:
            x.run_stage(st)
    o.set_pose(T0); o.run_stage("INITIALISE")
    g.initialise(T0)
    assert_same_state(o, g, "initialise", ["INDEX"])
    rgb, d, T1 = synth.frame(1, W, H, noise=True)
    for x in (o, g):
        x.upload_frame(rgb, d)
        for st in pre:
            x.run_stage(st)
    o.set_pose(T1); o.set_tick(2)
    o.run_stage("PREDICT_INDICES"); g.predict_indices(T1, 2, p.max_depth_processed, 0)
    assert_same_state(o, g, "predictIndices", [n for n in IMAGES if n.startswith("INDEX")])
    o.run_stage("FUSE"); g.fuse(T1, 2, p.max_depth_processed, 0)
    o.run_stage("PREDICT_INDICES"); g.predict_indices(T1, 2)
    o.run_stage("CLEAN"); g.clean(T1, 2, p.confidence_threshold, p.max_depth_processed)
    assert np.array_equal(o.fuse_stats(), g.fuse_stats()) and g.fuse_stats()[1] > 0
    o.run_stage("PREDICT_INDICES"); g.predict_indices()
    o.run_stage("PREDICT_HRBF"); g.predict_hrbf()
    assert_same_state(o, g, "operators", [n for n in IMAGES if n.startswith(("INDEX", "PRED"))])


def test_update_model_and_submap_mask(pair):
    """GlobalModel::updateModel and the active-submap mask, GPU vs oracle bit for bit, inside a
    tracked sequence (the corrected map is projected, fused, cleaned and predicted from afterwards)."""
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    o, g = pair(p)
    for k in range(3):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
    for x in (o, g):
        x.set_index_submap(1)                     # surfels created from now on belong to submap 1
    for k in range(3, 5):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
    assert_same_state(o, g, "two submaps")
    sm = g.download_map()[:, 5]
    assert (sm == 0).any() and (sm == 1).any()
    a, b = 0.004, -0.003
    T0 = np.eye(4, dtype=np.float32); T0[:3, 3] = (0.002, -0.001, 0.003)
    T1 = np.array([[np.cos(a), -np.sin(a), 0, 0.001], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, b], [0, 0, 0, 1]], np.float32)
    for x in (o, g):
        x.update_model(np.stack([T0, T1]))
    assert np.array_equal(bits(o.download_map()), bits(g.download_map()))
    for x in (o, g):
        x.set_active_submaps([1, 0])              # submap 1 goes inactive
    for k in range(5, 7):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "masked frame %d" % k)
    for x in (o, g):
        x.set_active_submaps(None)
    rgb, d, _ = synth.frame(7, W, H, noise=True)
    o.process_frame(rgb, d); g.process_frame(rgb, d)
    assert_same_state(o, g, "mask removed")


@pytest.mark.parametrize("mode", ["virtual3", "rccl1"])
def test_row_sharded_registration(pair, mode):
    """SURVEY §8e sharding 1: registration reductions over row strips + all-reduce of the exact limb sums give the
    single-GPU bits.  virtual3: one process plays three ranks in turn (strip arithmetic, no collective);
    rccl1: a real RCCL communicator of world size 1 (library binding, ncclAllReduce(int64, sum) on the stream)."""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    o, g = pair(p)
    if mode == "virtual3":
        g.comm_init(-1, 3)
    else:
        g.comm_init(0, 1, HRBFFusion.comm_unique_id())
    for k in range(5):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "%s frame %d" % (mode, k))
    g.comm_init(-1, 1)                                      # back to the single-GPU path, same context
    rgb, d, _ = synth.frame(5, W, H, noise=True)
    o.process_frame(rgb, d); g.process_frame(rgb, d)
    assert_same_state(o, g, mode + " back to single")


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_map_from_an_empty_map(pair, G):
    """SURVEY §8e sharding 2: the surfel map cut into G contiguous ranges of the global order, one process playing all
    shards (projection under global ids -> min-reduce of the z-buffer keys -> owner gathers, others write zeros ->
    integer sum-reduce; merges by the owner; clean + compaction per shard; appends on the last shard).  Every image,
    the concatenated map, the fuse statistics and the pose stay bit-identical to the oracle's single map — before and
    after re-cutting the ranges (frame 0 seeds everything on the last shard)."""
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    o, g = pair(p)
    g.comm_init(-1, G); g.map_shard_init(True)
    for k in range(7):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "G=%d frame %d" % (G, k))
        if k > 0:
            assert np.array_equal(o.fuse_stats(), g.fuse_stats()), k
        if k in (1, 4):                                     # frames 0-1 run with every shard but the last one empty
            g.map_rebalance()
            assert_same_state(o, g, "G=%d after rebalance at %d" % (G, k))
    assert g.local_surfel_count() == g.surfel_count() == o.surfel_count()


def test_sharded_map_rccl_world1(pair):
    """the real-mode code path on the one GPU a test box has: an RCCL communicator of world size 1, so every
    collective of the sharded map (allReduce min over u64 keys, grouped allReduce sum over the images as uint32,
    allGather of the counts, in place on the context's stream) is actually issued through librccl"""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    o, g = pair(p)
    g.comm_init(0, 1, HRBFFusion.comm_unique_id()); g.map_shard_init(True)
    for k in range(4):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "rccl1 frame %d" % k)
    g.map_rebalance()
    assert_same_state(o, g, "rccl1 after rebalance")
    g.set_row_sharding(False)                               # sharded map, replicated registration
    rgb, d, _ = synth.frame(4, W, H, noise=True)
    o.process_frame(rgb, d); g.process_frame(rgb, d)
    assert_same_state(o, g, "rccl1 rows replicated")


def test_sharded_map_uploaded_and_tracked(pair):
    """the same at QVGA against an uploaded 150 k-surfel map cut into 4 slices (several fuse tiles per shard, most of
    the view's winners owned by different shards), tracked and noisy, with the sparse-ICP option on top"""
    W, H, G = 320, 240, 4
    fx, fy, cx, cy = synth.intrinsics(W, H)
    seed = synth.seed_map(150_000, width=W)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=seed.shape[0] + 300_000, use_sparse_icp=1)   # capacity per shard
    o, g = pair(p)
    g.comm_init(-1, G); g.map_shard_init(True)
    rgb, d, T = synth.frame(0, W, H, noise=True)
    for x in (o, g):
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
    for k in range(1, 6):
        rgb, d, T = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "sharded upload frame %d" % k)
        assert np.array_equal(o.fuse_stats(), g.fuse_stats())
        if k == 3:
            g.map_rebalance()
            assert_same_state(o, g, "after rebalance")


@pytest.mark.parametrize("cfg", ["config4_vga_4M_4_shards", "config5_1280x960_8M_8_shards"])
def test_baseline_config_4_and_5_geometry_sharded_equals_single_equals_oracle(gpu_available, oracle_lib_built, cfg):
    """BASELINE configs 4 / 5 at their full sizes on the one GPU a test box has: a 640x480 stream against 4.3 M surfels
    cut into 4 shards, and a 1280x960 stream against 8.7 M surfels cut into 8 — every shard played in turn by one
    process with the winner-record exchange between them (key min-merge, pack, scatter).  Property: sharded == unsharded
    == the CPU oracle (about a second per frame and million surfels on its OpenMP build), bit for bit: pose, fuse
    statistics and the whole concatenated map (content and order) over tracked noisy frames with a stale-surfel purge in
    frame 1 (so that the shards' in-place compactions really move their ranges), then again after a re-cut of the ranges."""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H, n_seed, G = (640, 480, 4_300_000, 4) if cfg.startswith("config4") else (1280, 960, 8_700_000, 8)
    K = synth.intrinsics(W, H)
    seed = synth.seed_map(n_seed, width=W)
    n = len(seed)
    stale = np.zeros(n, bool)
    stale[np.random.default_rng(3).choice(n, n // 50, replace=False)] = True
    stale[0:64:3] = True
    seed[stale, 3] = 1.0
    frames = [synth.frame(k, W, H, noise=True) for k in range(4)]
    Q = (W // 2) * (H // 2)

    def run(shards, partition="ranges"):
        g = HRBFFusion(default_params(W, H, *K, max_surfels=int(1.15 * n / max(shards, 1)) + 8 * Q + 200_000 if shards else n + 8 * Q))
        if shards:
            g.comm_init(-1, shards); g.map_shard_init(True, partition=partition)
        g.upload_map(seed); g.set_pose(frames[0][2]); g.bootstrap(frames[0][0], frames[0][1])
        g.set_tick(300)
        out = []
        for k in range(1, 4):
            g.process_frame(frames[k][0], frames[k][1])
            out.append((g.get_pose(), g.fuse_stats(), g.surfel_count()))
            if shards and k == 2:
                g.map_rebalance()
        m = g.download_map()
        status = g.status()
        g.close()
        return out, m, status

    ref, m_ref, st_ref = run(0)
    got, m_got, st_got = run(G)
    assert st_ref == 0 and st_got == 0
    assert ref[0][1][0] + ref[0][1][2] - ref[0][1][3] > 0.8 * stale.sum()      # the purge happened
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(bits(a[0]), bits(b[0])), "%s frame %d pose" % (cfg, k + 1)
        assert np.array_equal(a[1], b[1]) and a[2] == b[2], "%s frame %d counts" % (cfg, k + 1)
    assert np.array_equal(bits(m_ref), bits(m_got))
    assert np.linalg.norm(got[-1][0][:3, 3] - frames[3][2][:3, 3]) < 0.03
    del m_got
    # the same shards owned by spatial hash instead of contiguous ranges (tests/test_hash_shards_gpu.py): merged by id, the same map
    got, m_got, st_got = run(G, partition="hash")
    assert st_got == 0
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(bits(a[0]), bits(b[0])), "%s frame %d pose (hash ownership)" % (cfg, k + 1)
        assert np.array_equal(a[1], b[1]) and a[2] == b[2], "%s frame %d counts (hash ownership)" % (cfg, k + 1)
    assert np.array_equal(bits(m_ref), bits(m_got))
    del m_got
    # the oracle on the same map and frames
    o = oracle_lib_built.Oracle(default_params(W, H, *K, max_surfels=n + 8 * Q), omp=True)
    try:
        o.upload_map(seed); o.set_pose(frames[0][2]); o.bootstrap(frames[0][0], frames[0][1])
        o.set_tick(300)
        for k in range(1, 4):
            o.process_frame(frames[k][0], frames[k][1])
            assert np.array_equal(bits(o.get_pose()), bits(ref[k - 1][0])), "%s frame %d pose vs oracle" % (cfg, k)
            assert np.array_equal(o.fuse_stats(), ref[k - 1][1]) and o.surfel_count() == ref[k - 1][2], "%s frame %d counts vs oracle" % (cfg, k)
        assert np.array_equal(bits(o.download_map()), bits(m_ref)), cfg + " map vs oracle"
    finally:
        o.close()


@pytest.mark.parametrize("sharded", [False, True])
def test_sparse_icp_with_outlier_slab(pair, sharded):
    """SURVEY §8f-4, use_sparse_icp: the ADMM variant of icpStep (multiplier image, shrink step, updateLambdaMap folded
    into the head of the next iteration on the GPU, a separate pass in the oracle).  Frame 3 carries a patch pushed
    9 cm back so the shrink step's non-trivial branch runs (the oracle counts it); everything stays bit-identical,
    also with the reductions split over three row strips."""
    W, H = 320, 240
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 19, use_sparse_icp=1)
    o, g = pair(p)
    if sharded:
        g.comm_init(-1, 3)
    for k in range(5):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        if k == 3:
            d = d.copy(); d[80:160, 100:220] += 450
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        assert_same_state(o, g, "sparse frame %d" % k)
        if k == 2:
            assert o.sparse_shrunk_count() == 0
    assert o.sparse_shrunk_count() > 1000


@pytest.mark.parametrize("K", [(132.0, 132.0, 80.0, 60.0), (129.325, 129.125, 79.65, 63.825), (141.0, 129.0, 71.5, 66.25)])
def test_icp_step_seam(oracle_lib_built, gpu_available, K):
    """hrbf_icp_step on caller-owned device maps == oracle (bit-exact sums) ~= fp64 numpy (1e-5); symmetric intrinsics,
    TUM fr1 proportions and a set with fx, fy 9 % apart (reduce.cu:326-331 projects with fx, cx / fy, cy separately)."""
    import torch
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    z = scenes.corner_depth(W, H, *K)
    r = scenes.pixel_rays(W, H, *K)
    P = r * z[..., None]
    dx = np.zeros_like(P); dy = np.zeros_like(P)
    dx[:, 1:-1] = P[:, 2:] - P[:, :-2]; dy[1:-1] = P[2:] - P[:-2]
    n = np.cross(dx, dy); ln = np.linalg.norm(n, axis=-1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-12), 0); n = np.where(n[..., 2:3] < 0, -n, n)
    v = np.stack([P[..., 0], P[..., 1], P[..., 2], np.ones_like(z)]).astype(np.float32)
    nn = np.stack([n[..., 0], n[..., 1], n[..., 2], np.ones_like(z)]).astype(np.float32)
    v[0][z <= 0] = np.nan; nn[0][ln[..., 0] <= 0] = np.nan
    kk = np.zeros_like(v); kk[3] = 0.5
    rng = np.random.default_rng(5)
    w = rng.uniform(0.1, 3.0, (H, W)).astype(np.float32); w[::7, ::5] = np.nan
    Rc = np.eye(3, dtype=np.float32); Rc[0, 1] = -0.004; Rc[1, 0] = 0.004
    tc = np.array([0.003, -0.002, 0.004], np.float32)
    I3 = np.eye(3, dtype=np.float32); t0 = np.zeros(3, np.float32)
    lib = oracle_lib_built.load()
    A0 = np.zeros(36); b0 = np.zeros(6); r0 = np.zeros(2)
    pp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.orc_icp_step(pp(Rc), pp(tc), pp(v), pp(nn), pp(kk), pp(kk), pp(I3), pp(t0), *K, pp(v), pp(nn), pp(kk), pp(kk),
                     pp(w), H, W, 0.1, 0.342, 1, pp(A0), pp(b0), pp(r0))
    g = HRBFFusion(default_params(W, H, *K, max_surfels=1024))
    dv, dn, dk, dw = (torch.from_numpy(a).cuda() for a in (v, nn, kk, w))
    A1 = np.zeros(36); b1 = np.zeros(6); r1 = np.zeros(2)
    dp = lambda t: C.c_void_p(t.data_ptr())
    rc = g.lib.hrbf_icp_step(g.h, pp(Rc), pp(tc), dp(dv), dp(dn), dp(dk), dp(dk), pp(I3), pp(t0), *K, dp(dv), dp(dn),
                             dp(dk), dp(dk), dp(dw), H, W, 0.1, 0.342, 1, pp(A1), pp(b1), pp(r1))
    assert rc == 0
    assert np.array_equal(A0, A1) and np.array_equal(b0, b1) and np.array_equal(r0, r1)
    assert r1[1] > 0.5 * W * H
    # independent fp64 reference of the point-to-plane system (reduce.cu:494-545)
    Pm = np.moveaxis(v[:3].astype(np.float64), 0, -1); Nm = np.moveaxis(nn[:3].astype(np.float64), 0, -1)
    s = Pm @ Rc.astype(np.float64).T + tc
    u = np.rint(s[..., 0] * K[0] / s[..., 2] + K[2]); vv = np.rint(s[..., 1] * K[1] / s[..., 2] + K[3])
    ok = np.isfinite(u) & np.isfinite(vv) & (u >= 0) & (vv >= 0) & (u < W) & (vv < H)
    ui = np.where(ok, u, 0).astype(int); vi = np.where(ok, vv, 0).astype(int)
    dm = Pm[vi, ui]; nm = Nm[vi, ui]; wm = w[vi, ui].astype(np.float64)
    ng = Nm @ Rc.astype(np.float64).T
    ok &= np.isfinite(dm[..., 0]) & np.isfinite(nm[..., 0]) & np.isfinite(Pm[..., 0]) & np.isfinite(Nm[..., 0])
    ok &= (np.linalg.norm(dm - s, axis=-1) <= 0.1) & (np.linalg.norm(np.cross(ng, nm), axis=-1) <= 0.342)
    wm = np.where(np.isnan(wm), 0.0, wm)
    J = np.concatenate([nm, np.cross(s, nm)], -1)[ok]; rr = ((s - dm) * nm).sum(-1)[ok]; ww = wm[ok]
    A_ref = (J * ww[:, None]).T @ J; b_ref = (J * ww[:, None]).T @ rr
    assert int(r1[1]) == int(ok.sum())
    # per-pixel rows are fp32 (like reduce.cu:494-545: s, d, n, cross products all in float), only the SUM is
    # exact; against an all-fp64 evaluation that leaves ~1e-5 relative on A and on the (cancelling) b
    np.testing.assert_allclose(A1.reshape(6, 6), A_ref, rtol=1e-4, atol=1e-5 * np.abs(A_ref).max())
    np.testing.assert_allclose(b1, b_ref, rtol=1e-4, atol=1e-5 * np.abs(b_ref).max())
    g.close()


@pytest.mark.parametrize("wscale", [1.0, 40.0, 3.0e3, 1.0e9, 1.0e15, 1.0e30])
def test_icp_step_exact_sums_across_reduction_paths(oracle_lib_built, gpu_available, wscale):
    """The wave stage of the exact reduction picks its form by the largest magnitude in the wave: doubles below 2^6, two
    / three / five 25-bit limbs above.  The same ICP system with the weight map scaled into each range: every sum bit-equal
    to the oracle's 128-bit accumulation."""
    import torch
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    K = synth.intrinsics(W, H)
    z = scenes.corner_depth(W, H, *K)
    P = scenes.pixel_rays(W, H, *K) * z[..., None]
    dx = np.zeros_like(P); dy = np.zeros_like(P)
    dx[:, 1:-1] = P[:, 2:] - P[:, :-2]; dy[1:-1] = P[2:] - P[:-2]
    n = np.cross(dx, dy); ln = np.linalg.norm(n, axis=-1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-12), 0); n = np.where(n[..., 2:3] < 0, -n, n)
    v = np.stack([P[..., 0], P[..., 1], P[..., 2], np.ones_like(z)]).astype(np.float32)
    nn = np.stack([n[..., 0], n[..., 1], n[..., 2], np.ones_like(z)]).astype(np.float32)
    v[0][z <= 0] = np.nan; nn[0][ln[..., 0] <= 0] = np.nan
    kk = np.zeros_like(v); kk[3] = 0.5
    rng = np.random.default_rng(6)
    w = (rng.uniform(0.1, 3.0, (H, W)) * wscale).astype(np.float32)
    w[:, : W // 2] *= np.float32(1e-3)          # half of the image three decades lower: waves of different ranges in one launch
    Rc = np.eye(3, dtype=np.float32); Rc[0, 1] = -0.004; Rc[1, 0] = 0.004
    tc = np.array([0.003, -0.002, 0.004], np.float32)
    I3 = np.eye(3, dtype=np.float32); t0 = np.zeros(3, np.float32)
    lib = oracle_lib_built.load()
    A0 = np.zeros(36); b0 = np.zeros(6); r0 = np.zeros(2)
    pp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.orc_icp_step(pp(Rc), pp(tc), pp(v), pp(nn), pp(kk), pp(kk), pp(I3), pp(t0), *K, pp(v), pp(nn), pp(kk), pp(kk),
                     pp(w), H, W, 0.1, 0.342, 1, pp(A0), pp(b0), pp(r0))
    g = HRBFFusion(default_params(W, H, *K, max_surfels=1024))
    dv, dn, dk, dw = (torch.from_numpy(a).cuda() for a in (v, nn, kk, w))
    A1 = np.zeros(36); b1 = np.zeros(6); r1 = np.zeros(2)
    dp = lambda t: C.c_void_p(t.data_ptr())
    rc = g.lib.hrbf_icp_step(g.h, pp(Rc), pp(tc), dp(dv), dp(dn), dp(dk), dp(dk), pp(I3), pp(t0), *K, dp(dv), dp(dn),
                             dp(dk), dp(dk), dp(dw), H, W, 0.1, 0.342, 1, pp(A1), pp(b1), pp(r1))
    g.close()
    assert rc == 0 and r1[1] > 0.5 * W * H
    assert np.array_equal(A0, A1) and np.array_equal(b0, b1) and np.array_equal(r0, r1)


def test_icp_step_sparse_seam(oracle_lib_built, gpu_available):
    """hrbf_icp_step_sparse / hrbf_update_lambda_map (icpStep with useSparse, updateLambdaMap) on caller-owned device
    images against the oracle: sums, z_thrinkMap, corresICP and the updated lambdaMap bit for bit, over three
    iterations.  The multiplier image starts with random vectors of up to 1.5 m / mu so that both branches of the
    shrink operator run; with lambda = 0 the first iteration equals the plain seam; an independent numpy evaluation
    checks z and the multiplier update of the matched pixels."""
    import torch
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    K = (132.0, 132.0, 80.0, 60.0)
    z = scenes.corner_depth(W, H, *K)
    r = scenes.pixel_rays(W, H, *K)
    P = r * z[..., None]
    dx = np.zeros_like(P); dy = np.zeros_like(P)
    dx[:, 1:-1] = P[:, 2:] - P[:, :-2]; dy[1:-1] = P[2:] - P[:-2]
    n = np.cross(dx, dy); ln = np.linalg.norm(n, axis=-1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-12), 0); n = np.where(n[..., 2:3] < 0, -n, n)
    v = np.stack([P[..., 0], P[..., 1], P[..., 2], np.ones_like(z)]).astype(np.float32)
    nn = np.stack([n[..., 0], n[..., 1], n[..., 2], np.ones_like(z)]).astype(np.float32)
    v[0][z <= 0] = np.nan; nn[0][ln[..., 0] <= 0] = np.nan
    kk = np.zeros_like(v); kk[3] = 0.5
    rng = np.random.default_rng(9)
    w = rng.uniform(0.1, 3.0, (H, W)).astype(np.float32)
    Rc = np.eye(3, dtype=np.float32); Rc[0, 1] = -0.004; Rc[1, 0] = 0.004
    tc = np.array([0.003, -0.002, 0.004], np.float32)
    I3 = np.eye(3, dtype=np.float32); t0 = np.zeros(3, np.float32)
    lib = oracle_lib_built.load()
    pp = lambda a: a.ctypes.data_as(C.c_void_p)
    dp = lambda t: C.c_void_p(t.data_ptr())
    g = HRBFFusion(default_params(W, H, *K, max_surfels=1024))
    dv, dn, dk, dw = (torch.from_numpy(a).cuda() for a in (v, nn, kk, w))

    def both(lam):
        zo = np.full((H, W, 3), 7.0, np.float32); co = np.full((H, W, 2), 5, np.int32)
        A0 = np.zeros(36); b0 = np.zeros(6); r0 = np.zeros(2)
        lib.orc_icp_step_sparse(pp(Rc), pp(tc), pp(v), pp(nn), pp(kk), pp(kk), pp(I3), pp(t0), *K, pp(v), pp(nn), pp(kk),
                                pp(kk), pp(w), H, W, 0.1, 0.342, 1, pp(lam), pp(zo), pp(co), pp(A0), pp(b0), pp(r0))
        dl = torch.from_numpy(lam).cuda(); dz = torch.full((H, W, 3), 7.0, device="cuda"); dc = torch.full((H, W, 2), 5, dtype=torch.int32, device="cuda")
        A1 = np.zeros(36); b1 = np.zeros(6); r1 = np.zeros(2)
        rc = g.lib.hrbf_icp_step_sparse(g.h, pp(Rc), pp(tc), dp(dv), dp(dn), dp(dk), dp(dk), pp(I3), pp(t0), *K, dp(dv),
                                        dp(dn), dp(dk), dp(dk), dp(dw), H, W, 0.1, 0.342, 1, dp(dl), dp(dz), dp(dc),
                                        pp(A1), pp(b1), pp(r1))
        assert rc == 0
        assert np.array_equal(A0, A1) and np.array_equal(b0, b1) and np.array_equal(r0, r1)
        assert np.array_equal(bits(zo), bits(dz.cpu().numpy())) and np.array_equal(co, dc.cpu().numpy())
        lam_o = lam.copy()
        lib.orc_update_lambda_map(pp(Rc), pp(tc), pp(v), pp(I3), pp(t0), pp(v), pp(co), pp(zo), pp(lam_o), H, W)
        assert g.lib.hrbf_update_lambda_map(g.h, pp(Rc), pp(tc), dp(dv), pp(I3), pp(t0), dp(dv), dp(dc), dp(dz), dp(dl), H, W) == 0
        assert np.array_equal(bits(lam_o), bits(dl.cpu().numpy()))
        return (A0, b0, r0), zo, co, lam_o

    # lambda = 0: the plain seam's system, z = 0 everywhere
    A_p = np.zeros(36); b_p = np.zeros(6); r_p = np.zeros(2)
    lib.orc_icp_step(pp(Rc), pp(tc), pp(v), pp(nn), pp(kk), pp(kk), pp(I3), pp(t0), *K, pp(v), pp(nn), pp(kk), pp(kk),
                     pp(w), H, W, 0.1, 0.342, 1, pp(A_p), pp(b_p), pp(r_p))
    (A0, b0, r0), zo, co, lam1 = both(np.zeros((H, W, 3), np.float32))
    assert np.array_equal(A0, A_p) and np.array_equal(b0, b_p) and np.array_equal(r0, r_p) and not zo.any()
    assert (co[..., 0] >= 0).sum() == int(r0[1])
    # random multipliers: both shrink branches, then two more iterations fed with the updated multipliers
    lam = (rng.standard_normal((H, W, 3)) * rng.uniform(0, 15.0, (H, W, 1))).astype(np.float32)
    for it in range(3):
        lam_in = lam
        _, zo, co, lam = both(lam_in)
        m = co[..., 0] >= 0
        hit = np.linalg.norm(zo, axis=-1) > 0
        if it == 0:
            assert hit.sum() > 100 and (m & ~hit).sum() > 100
        # independent fp64 check on the matched pixels: z = beta(|h|) h, lambda' = lambda + mu (s - d - z) where x > 0
        Pm = np.moveaxis(v[:3].astype(np.float64), 0, -1)
        s_ = Pm @ Rc.astype(np.float64).T + tc
        d_ = Pm[np.clip(co[..., 1], 0, H - 1), np.clip(co[..., 0], 0, W - 1)]
        h = s_ - d_ + lam_in.astype(np.float64) / 10.0
        hn = np.linalg.norm(h, axis=-1)
        alpha = 0.1 ** (2.0 / 3.0)
        beta = (alpha / np.maximum(hn, 1e-30) + 1.0) / 2.0
        for _ in range(3):
            beta = 1.0 - 0.05 * np.maximum(hn, 1e-30) ** -1.5 * beta ** -0.5
        beta = np.where(hn <= alpha + 0.05 / np.sqrt(alpha), 0.0, beta)
        near = np.abs(hn - (alpha + 0.05 / np.sqrt(alpha))) < 1e-4          # fp32 / fp64 may disagree at the threshold
        sel = m & ~near
        np.testing.assert_allclose(zo[sel], (beta[..., None] * h)[sel], rtol=2e-4, atol=2e-5)
        upd = sel & (co[..., 0] > 0)
        np.testing.assert_allclose(lam[upd], (lam_in + 10.0 * (s_ - d_ - zo))[upd], rtol=2e-4, atol=2e-3)
        keep = m & (co[..., 0] == 0)
        assert np.array_equal(lam[keep], lam_in[keep]) and np.array_equal(lam[~m], lam_in[~m])
    g.close()


def test_rgb_and_so3_step_seams(oracle_lib_built, gpu_available):
    """hrbf_so3_step, hrbf_rgb_residual, hrbf_rgb_step on caller-owned device images == oracle: the correspondence
    image byte for byte, count / sigma and every normal-equation entry bit for bit."""
    import torch
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    f0, f1 = synth.frame(3, W, H, noise=True), synth.frame(4, W, H, noise=True)
    grey = lambda rgb: (0.114 * rgb[..., 0] + 0.299 * rgb[..., 1] + 0.587 * rgb[..., 2]).astype(np.uint8)
    last_img, next_img = np.ascontiguousarray(grey(f0[0])), np.ascontiguousarray(grey(f1[0]))
    last_img[5:9, 7:30] = 0; next_img[40:44, 90:95] = 0                       # invalid intensities
    dep = lambda d: np.where(d > 0, d.astype(np.float32) / 5000.0, np.nan).astype(np.float32)
    last_d, next_d = dep(f0[1]), dep(f1[1])
    gi = next_img.astype(np.int32)
    dIdx = np.zeros((H, W), np.int16); dIdy = np.zeros((H, W), np.int16)
    dIdx[:, 1:-1] = 4 * (gi[:, 2:] - gi[:, :-2]); dIdy[1:-1] = 4 * (gi[2:] - gi[:-2])
    a = 0.01
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    krk = (K @ R @ np.linalg.inv(K)).astype(np.float32); kt = (K @ np.array([0.004, -0.002, 0.003])).astype(np.float32)
    kinv = np.linalg.inv(K).astype(np.float32); krlr = (K @ R).astype(np.float32)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    z = np.nan_to_num(last_d, nan=0.0)
    cloud = np.ascontiguousarray(np.stack([(xs - cx) * z / fx, (ys - cy) * z / fy, z], -1).astype(np.float32))
    pp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib = oracle_lib_built.load()
    g = HRBFFusion(default_params(W, H, fx, fy, cx, cy, max_surfels=1024))
    dev = lambda a: torch.from_numpy(a).cuda()
    dp = lambda t: C.c_void_p(t.data_ptr())
    d_last, d_next, d_ld, d_nd, d_ix, d_iy, d_cloud = map(dev, (last_img, next_img, last_d, next_d, dIdx, dIdy, cloud))

    # --- so3Step
    A0, b0, r0 = np.zeros(9), np.zeros(3), np.zeros(2); A1, b1, r1 = np.zeros(9), np.zeros(3), np.zeros(2)
    lib.orc_so3_step(pp(last_img), pp(next_img), H, W, pp(krk), pp(kinv), pp(krlr), pp(A0), pp(b0), pp(r0))
    assert g.lib.hrbf_so3_step(g.h, dp(d_last), dp(d_next), H, W, pp(krk), pp(kinv), pp(krlr), pp(A1), pp(b1), pp(r1)) == 0
    assert np.array_equal(A0, A1) and np.array_equal(b0, b1) and np.array_equal(r0, r1) and r1[1] > 0.5 * W * H

    # --- computeRgbResidual
    co0 = np.zeros((H * W, 6), np.int16); df0 = np.zeros(H * W, np.float32)
    c0, s0 = C.c_longlong(), C.c_longlong()
    lib.orc_rgb_residual(25.0, pp(dIdx), pp(dIdy), pp(last_d), pp(next_d), pp(last_img), pp(next_img), H, W, pp(kt), pp(krk),
                         pp(co0), pp(df0), C.byref(c0), C.byref(s0))
    d_co = torch.zeros((H * W, 6), dtype=torch.int16, device="cuda"); d_df = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    c1, s1 = C.c_longlong(), C.c_longlong()
    assert g.lib.hrbf_rgb_residual(g.h, 25.0, dp(d_ix), dp(d_iy), dp(d_ld), dp(d_nd), dp(d_last), dp(d_next), H, W, pp(kt),
                                   pp(krk), dp(d_co), dp(d_df), C.byref(c1), C.byref(s1)) == 0
    assert (c0.value, s0.value) == (c1.value, s1.value) and c1.value > 100
    assert np.array_equal(d_co.cpu().numpy(), co0) and np.array_equal(bits(d_df.cpu().numpy()), bits(df0))

    # --- rgbStep, plain and gradient-weighted, robust and sigma = -1
    for sigma, use_grad in ((np.float32(np.sqrt(c0.value)), 0), (np.float32(-1.0), 0), (np.float32(30.0), 1)):
        A0, b0, r0 = np.zeros(36), np.zeros(6), np.zeros(2); A1, b1, r1 = np.zeros(36), np.zeros(6), np.zeros(2)
        lib.orc_rgb_step(pp(co0), pp(df0), float(sigma), pp(cloud), fx, fy, pp(dIdx), pp(dIdy), use_grad, H, W, pp(A0), pp(b0), pp(r0))
        assert g.lib.hrbf_rgb_step(g.h, dp(d_co), dp(d_df), float(sigma), dp(d_cloud), fx, fy, dp(d_ix), dp(d_iy), use_grad, H, W,
                                   pp(A1), pp(b1), pp(r1)) == 0
        assert np.array_equal(bits(A0), bits(A1)) and np.array_equal(bits(b0), bits(b1)) and np.array_equal(bits(r0), bits(r1))
        assert r1[1] == c0.value
    g.close()


def test_full_size_properties_1M(gpu_available):
    """BASELINE sizes (640x480, > 1 M surfels): size-independent properties instead of the oracle:
    count conservation, order preservation of the compaction, in-place invariance of untouched
    surfels, z-buffer correctness of the index map, and run-to-run determinism."""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 640, 480
    fx, fy, cx, cy = synth.intrinsics(W, H)
    seed = synth.seed_map(1_050_000)
    seed[:, 6] = np.arange(len(seed)) % 16000 + 1      # initTime doubles as an order tag (kept < 2^24)
    tag = [r.tobytes() for r in np.ascontiguousarray(seed[:, 0:3])]      # 96-bit position key
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=seed.shape[0] + 600_000)

    def run():
        g = HRBFFusion(p)
        g.upload_map(seed); g.set_pose(synth.camera_pose(0))
        rgb, d, _ = synth.frame(0, W, H); g.bootstrap(rgb, d)
        maps, stats = [], []
        for k in range(1, 4):
            rgb, d, T = synth.frame(k, W, H)
            g.process_frame(rgb, d)
            maps.append(g.download_map()); stats.append(g.fuse_stats().astype(np.int64))
        out = (maps, stats, g.get_pose(), g.get_image("INDEX"), g.get_image("INDEX_VERTCONF"), T)
        g.close()
        return out

    maps, stats, pose, idx, vcf, T = run()
    n_prev = len(seed)
    for m, st in zip(maps, stats):
        assert st[0] == n_prev and st[3] == len(m)
        removed = st[0] + st[2] - st[3]
        assert removed >= 0 and st[1] > 30000 and st[2] <= (W // 2) * (H // 2)
        n_prev = len(m)
    # order preservation: the seeded surfels that survive appear in their original relative order.
    # Unmerged surfels keep (x,y,z) bit for bit -> their keys form a subsequence of the seed keys.
    m = maps[-1]
    old = m[m[:, 7] == 1.0]                      # never merged, never appended
    pos = {t: i for i, t in enumerate(tag)}
    assert len(pos) == len(tag)
    where = np.array([pos[r.tobytes()] for r in np.ascontiguousarray(old[::97, 0:3])])
    assert np.all(np.diff(where) > 0)
    assert np.array_equal(bits(old[::97]), bits(seed[where]))      # untouched surfels are bit-identical
    # appended surfels carry this run's time stamps and sit at the tail
    tail = m[len(seed) - int(sum(s[0] + s[2] - s[3] for s in stats)):]
    assert np.all(tail[:, 6] >= 2.0)
    # index map = z-buffer: each stored index points at a surfel that projects into that pixel with that depth
    ys, xs = np.nonzero(idx)
    sel = slice(None, None, 211)
    Tinv = np.linalg.inv(pose.astype(np.float64))
    P = maps[-1][idx[ys[sel], xs[sel]], :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]
    assert np.abs(P[:, 2] - vcf[ys[sel], xs[sel], 2]).max() < 1e-5
    u = np.floor(fx * P[:, 0] / P[:, 2] + cx); v = np.floor(fy * P[:, 1] / P[:, 2] + cy)
    assert (np.abs(u - xs[sel]) <= 1).all() and (np.abs(v - ys[sel]) <= 1).all()
    assert np.linalg.norm(pose[:3, 3] - T[:3, 3]) < 0.02
    # determinism: atomics (z-buffer, slots, look-back) do not leak scheduling order into the result
    maps2, stats2, pose2, idx2, _, _ = run()
    assert np.array_equal(bits(maps[-1]), bits(maps2[-1])) and np.array_equal(bits(pose), bits(pose2))
    assert np.array_equal(idx, idx2)


def test_full_size_forced_early_removals_1M(gpu_available):
    """The in-place compaction at BASELINE size when it has to move the WHOLE map: 3 % of a 1.05 M-surfel map is unstable
    and stale (copy_unstable.vert:159-165 drops unstable surfels not seen for 200 frames), spread from index 0, noisy
    input.  Properties: every unmerged stale surfel is gone, the survivors keep their relative order and their bits
    (they were re-read and written `shift` slots to the left across ~520 ticketed tiles), counts add up, the pass
    reports that it moved (nearly) everything, and a second run reproduces map and pose bit for bit."""
    from hrbffusion3d_amd.api import HRBFFusion
    W, H = 640, 480
    K = synth.intrinsics(W, H)
    seed = synth.seed_map(1_050_000)
    n = len(seed)
    stale = np.zeros(n, bool)
    stale[np.random.default_rng(7).choice(n, int(0.03 * n), replace=False)] = True
    stale[0:64:3] = True
    seed[stale, 3] = 1.0
    tag = {r.tobytes(): i for i, r in enumerate(np.ascontiguousarray(seed[:, 0:3]))}
    assert len(tag) == n
    p = default_params(W, H, *K, max_surfels=n + 600_000)

    def run():
        g = HRBFFusion(p)
        g.enable_timing(2)
        g.upload_map(seed); g.set_pose(synth.camera_pose(0))
        rgb, d, _ = synth.frame(0, W, H, noise=True); g.bootstrap(rgb, d)
        g.set_tick(300)
        stats = []
        for k in range(1, 4):
            rgb, d, T = synth.frame(k, W, H, noise=True)
            g.process_frame(rgb, d)
            stats.append(g.fuse_stats().astype(np.int64))
        _, _, st8 = g.fuse_ring_parts(3)
        out = (g.download_map(), stats, st8.astype(np.int64), g.get_pose(), g.status())
        g.close()
        return out

    m, stats, st8, pose, status = run()
    assert status == 0
    removed1 = stats[0][0] + stats[0][2] - stats[0][3]
    assert stats[0][0] == n and removed1 > 0.8 * stale.sum()
    assert st8[0][6] > 0.9 * n                       # the first frame moved (nearly) the whole map
    assert stats[-1][3] == len(m)
    old = m[m[:, 7] == 1.0]                          # never merged, never appended
    assert not np.any(old[:, 3] == 1.0)              # no stale unstable surfel survived unmerged
    where = np.array([tag[r.tobytes()] for r in np.ascontiguousarray(old[:, 0:3])])
    assert np.all(np.diff(where) > 0)                # order preserved over the whole map, not a sample
    assert np.array_equal(bits(old), bits(seed[where]))
    assert len(old) > 0.85 * n
    m2, stats2, _, pose2, _ = run()
    assert np.array_equal(bits(m), bits(m2)) and np.array_equal(bits(pose), bits(pose2))


def test_run_cli_over_a_klg_log(gpu_available, tmp_path):
    """The caller's side (`python -m hrbffusion3d_amd.run`, the MainController/RawLogReader loop without the GUI):
    a synthetic QVGA stream written as a .klg log, replayed through process_frame; the trajectory file and the PLY are
    written, the map grows, and the ATE against the stream's ground truth (saved as a TUM file) is small."""
    from hrbffusion3d_amd import run
    from hrbffusion3d_amd.io import write_klg, save_trajectory, load_trajectory_tum
    W, H, N = 320, 240, 30
    fx, fy, cx, cy = synth.intrinsics(W, H)
    fr = [synth.frame(k, W, H) for k in range(N)]
    klg = str(tmp_path / "s.klg")
    write_klg(klg, [(k * 33333, f[0], f[1]) for k, f in enumerate(fr)])
    gt = str(tmp_path / "gt.txt")
    save_trajectory(gt, [f[2] for f in fr], stamps_us=[k * 33333 for k in range(N)])
    out, ply = str(tmp_path / "traj.txt"), str(tmp_path / "map.ply")
    rep = run.main(["--klg", klg, "--width", str(W), "--height", str(H), "--fx", str(fx), "--fy", str(fy),
                    "--cx", str(cx), "--cy", str(cy), "--max-surfels", str(1 << 20), "--out", out, "--ply", ply,
                    "--groundtruth", gt])
    assert rep["frames"] == N and rep["surfels"] > 0.8 * W * H and rep["ply_vertices"] > 0
    assert rep["ate_pairs"] == N and rep["ate_rmse_m"] < 0.05, rep
    st, ps = load_trajectory_tum(out)
    assert len(ps) == N and np.isfinite(np.asarray(ps)).all()
    assert open(ply, "rb").read(3) == b"ply"
