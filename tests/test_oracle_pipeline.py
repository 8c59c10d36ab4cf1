"""BASELINE config 1: the GPUTest two-frame pair through the CPU-only path, end to end, against the
committed golden fixture (tests/golden/gputest_pair_expected.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np

from hrbffusion3d_amd.params import default_params, IMAGES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gputest_pair_expected.npz")


def test_png_pair_end_to_end_matches_golden(oracle_lib_built, png_pair):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    got = mg.run(lambda p: oracle_lib_built.Oracle(p, omp=False))     # single-thread build
    exp = np.load(GOLD)
    assert set(got) == set(exp.files)
    for k in exp.files:
        a, b = np.asarray(got[k]), exp[k]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), k


def test_png_pair_variants_match_golden(oracle_lib_built, png_pair):
    """the same pair under six registration / pre-processing options (sparse ICP, windowed search, RGB only, ICP only,
    no SO3 / pyramid, Gauss filter + central-difference normals) against the committed digests"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    got = mg.run_variants(lambda p: oracle_lib_built.Oracle(p, omp=True))
    exp = np.load(os.path.join(os.path.dirname(GOLD), "gputest_pair_variants.npz"))
    assert set(got) == set(exp.files)
    for k in exp.files:
        a, b = np.asarray(got[k]), exp[k]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), k
    poses = [exp[n + "_pose"] for n in mg.VARIANTS]
    assert len({p.tobytes() for p in poses}) == len(poses)        # every option really changes the result
    for p in poses:
        assert np.linalg.norm(p[:3, 3]) < 0.05


def test_png_pair_sanity(oracle_lib_built):
    """Pass criteria of SURVEY §8d config 1: runs end to end, small motion, map not empty."""
    exp = np.load(GOLD)
    T = exp["f2_pose"]
    assert np.linalg.norm(T[:3, 3]) < 0.05
    ang = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3]) - 1) / 2, -1, 1)))
    assert ang < 5.0
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-5)
    assert exp["f1_count"][0] > 200000 and exp["f2_count"][0] > exp["f1_count"][0]
    st = exp["f2_stats"]
    assert st[0] == exp["f1_count"][0] and st[3] == exp["f2_count"][0] and st[1] > 50000
    assert exp["f2_icp"][1] > 200000


def test_omp_and_single_thread_oracle_agree(oracle_lib_built):
    """the exact accumulator makes the OpenMP build bit-identical to the scalar one"""
    import scenes
    from hrbffusion3d_amd import synth
    W, H = 160, 120
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17)
    outs = []
    for omp in (False, True):
        o = oracle_lib_built.Oracle(p, omp=omp)
        for k in range(3):
            rgb, d, _ = synth.frame(k, W, H)
            o.process_frame(rgb, d)
        outs.append((o.get_pose(), o.download_map(), o.get_image("PRED_VERTEX")))
        o.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
