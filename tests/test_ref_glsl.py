"""Parity with the reference's own GLSL, executed (oracle/_ref).

tests/golden/ref_glsl/*.npz hold what the shaders under /root/reference/Core/src/Shaders wrote when Mesa's GLSL compiler
built them and llvmpipe ran them in the build container (tests/golden/make_ref_glsl.py + oracle/ref_glsl/refgl.py, which
issue the reference host code's GL calls).  Nothing here needs the reference or a GL context: only the fixtures travel.

  CPU suite : the C oracle against the fixtures — this is what pins the oracle (and with it the 90 bit-exact GPU parity
              tests of tests/test_parity_gpu.py) to the reference.
  GPU suite : the HIP library, through the C-ABI, against the same fixtures.

Rows of SURVEY.md §8a covered: P1-P5, M1, F1-F4, H1-H3, f-3 (every GLSL row).  tests/ref_glsl_check.py states the bounds.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_glsl_check as R  # noqa: E402

sys.path.insert(0, HERE)
SCENES = ["pair", "sphere"]


def scene_params(scene, **kw):
    import make_ref_glsl as M
    return M.params(scene, **kw)


@pytest.mark.parametrize("scene", SCENES)
def test_fixture_is_the_reference_shaders_output(scene):
    """the fixture carries every pass of the GLSL rows and says which shader executions it came from"""
    fx = R.load(scene)
    for k in ("f2_DEPTH_FILTERED", "f2_NORMAL_P3", "f2_CURV1", "f2_CONFIDENCE", "f1_map", "f2_a_INDEX", "f2_records", "f2_keep",
              "x_extra", "x_keep", "x_p_INDEX", "x_PRED_VERTEX", "x_FILL_VERTEX", "x_map_updated_head", "f2_init_head"):
        assert k in fx, k
    removed = int((~np.unpackbits(fx["x_keep"])[:fx["f1_map"].shape[0] + fx["x_extra"].shape[0]].astype(bool)).sum())
    assert removed >= 200, "the stable-map flow must exercise the three removal rules"
    assert int((fx["x_PRED_VERTEX"][..., 2] != 0).sum()) > fx["x_PRED_VERTEX"].shape[0] * fx["x_PRED_VERTEX"].shape[1] // 2


@pytest.mark.parametrize("scene", SCENES)
def test_oracle_matches_the_executed_reference_shaders(scene, oracle_lib_built):
    fx = R.load(scene)
    o = oracle_lib_built.Oracle(scene_params(scene), omp=True)
    try:
        rep = R.run(o, fx, R.Report(strict=True))
    finally:
        o.close()
    assert len(rep.rows) > 60 and all(ok for _, ok, _ in rep.rows)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
def test_hip_path_matches_the_executed_reference_shaders(scene, gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    fx = R.load(scene)
    g = HRBFFusion(scene_params(scene))
    try:
        rep = R.run(g, fx, R.Report(strict=True))
    finally:
        g.close()
    assert len(rep.rows) > 60 and all(ok for _, ok, _ in rep.rows)


def _thumb_params(W, H):
    from hrbffusion3d_amd.params import default_params
    return default_params(width=W, height=H, fx=float(W), fy=float(W), cx=W / 2.0, cy=H / 2.0, max_surfels=1 << 12)


def test_oracle_dense_enough_reads_the_texels_the_executed_resize_shader_read(oracle_lib_built):
    """Resize::vertex + denseEnough: cell size 20, NEAREST at the cell centre, z > 0, `per > 0.75`"""
    fx = R.load("thumbnail")
    assert [tuple(s) for s in fx["sizes"].tolist()] == [(640, 480), (160, 120), (256, 128), (128, 128)]
    rep = R.run_thumbnail(lambda W, H: oracle_lib_built.Oracle(_thumb_params(W, H)), fx, R.Report(strict=True))
    assert len(rep.rows) == 20 and all(ok for _, ok, _ in rep.rows)


@pytest.mark.gpu
def test_hip_dense_enough_reads_the_texels_the_executed_resize_shader_read(gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    rep = R.run_thumbnail(lambda W, H: HRBFFusion(_thumb_params(W, H)), R.load("thumbnail"), R.Report(strict=True))
    assert len(rep.rows) == 20 and all(ok for _, ok, _ in rep.rows)


def _nonpow2_params(fx):
    from hrbffusion3d_amd.params import default_params
    W, H, fx_, fy_, cx, cy = fx["geom"]
    return default_params(width=int(W), height=int(H), fx=float(fx_), fy=float(fy_), cx=float(cx), cy=float(cy), max_surfels=1 << 16)


def test_oracle_matches_the_executed_shaders_at_a_size_that_is_not_a_power_of_two(oracle_lib_built):
    """160 x 120: the float-stepped window loops drop their last sample at 88 columns / 17 rows — the rule of hd_window_axis"""
    fx = R.load("qqvga_pre")
    assert int((fx["win_x"] == 6).sum()) == 88 and int((fx["win_y"] == 6).sum()) == 17
    o = oracle_lib_built.Oracle(_nonpow2_params(fx), omp=True)
    try:
        rep = R.run_nonpow2_pre(o, fx, R.Report(strict=True))
    finally:
        o.close()
    assert len(rep.rows) > 15 and all(ok for _, ok, _ in rep.rows)


@pytest.mark.gpu
def test_hip_path_matches_the_executed_shaders_at_a_size_that_is_not_a_power_of_two(gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    fx = R.load("qqvga_pre")
    g = HRBFFusion(_nonpow2_params(fx))
    try:
        rep = R.run_nonpow2_pre(g, fx, R.Report(strict=True))
    finally:
        g.close()
    assert len(rep.rows) > 15 and all(ok for _, ok, _ in rep.rows)


def test_oracle_matches_the_executed_map_passes_at_a_size_that_is_not_a_power_of_two(oracle_lib_built):
    """160 x 120: the vertex shaders' host-computed uv attribute is an ulp off the fragment texcoord at 43 columns / 19 rows —
    data.vert's own normal, position and ray follow the attribute (hd_uv_attribute; found at 640 x 480)"""
    fx = R.load("qqvga_map")
    assert len(fx["uv_cols_differ"]) == 43 and len(fx["uv_rows_differ"]) == 19
    o = oracle_lib_built.Oracle(_nonpow2_params(fx), omp=True)
    try:
        rep = R.run_nonpow2_map(o, fx, R.Report(strict=True))
    finally:
        o.close()
    assert len(rep.rows) > 15 and all(ok for _, ok, _ in rep.rows)


@pytest.mark.gpu
def test_hip_path_matches_the_executed_map_passes_at_a_size_that_is_not_a_power_of_two(gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    fx = R.load("qqvga_map")
    g = HRBFFusion(_nonpow2_params(fx))
    try:
        rep = R.run_nonpow2_map(g, fx, R.Report(strict=True))
    finally:
        g.close()
    assert len(rep.rows) > 15 and all(ok for _, ok, _ in rep.rows)


# ---- 640 x 480: the size BASELINE's metric is quoted on ---------------------------------------------------------------------------
_VGA = {}


def _vga():
    if not _VGA:
        import ref_glsl_vga as V
        _VGA.update(V.decode())
    return _VGA


def _vga_params():
    from hrbffusion3d_amd.params import default_params
    return default_params(max_surfels=1 << 20)        # 640 x 480, K = (528, 528, 320, 240), 1 / 5000: the GPUTest pair


def test_vga_fixture_decodes_to_the_recorded_bits_and_covers_every_pass():
    """tests/ref_glsl_vga.py rebuilds 177 MB of shader outputs from 8.4 MB (numpy and staged-oracle predictors + XOR residuals) and checks a CRC per array"""
    fx = _vga()
    assert "llvmpipe" in fx["_info"]["renderer"]
    assert fx["f2_depth"].shape == (480, 640) and fx["f1_map"].shape[0] > 250000
    for k in ("f2_DEPTH_FILTERED", "f2_NORMAL_P3", "f2_CURV1", "f2_CONFIDENCE", "f2_a_INDEX", "f2_records", "f2_fused_vals", "f2_keep", "x_keep",
              "x_p_INDEX_CURVMAX", "x_PRED_VERTEX", "x_FILL_ICPWEIGHT", "x_map_updated_head", "f2_init_head", "tc"):
        assert k in fx, k
    removed = int((~np.unpackbits(fx["x_keep"])[:fx["f1_map"].shape[0] + fx["x_extra"].shape[0]].astype(bool)).sum())
    assert removed >= 500 and int((fx["f2_records"][:, 7] == -1).sum()) > 60000 and int((fx["x_PRED_VERTEX"][..., 2] != 0).sum()) > 250000


def test_oracle_matches_the_executed_reference_shaders_at_640x480(oracle_lib_built):
    fx = _vga()
    o = oracle_lib_built.Oracle(_vga_params(), omp=True)
    try:
        rep = R.run_vga(o, fx, R.Report(strict=True))
    finally:
        o.close()
    assert len(rep.rows) > 90 and all(ok for _, ok, _ in rep.rows)


def test_oracle_matches_every_pixel_at_640x480_given_the_rasterisers_texcoords(oracle_lib_built):
    """the one implementation-defined input of the fragment passes — the interpolated texcoord — taken from the execution (test hook
    orc_set_fragment_texcoords): no mask is left, all 307 200 pixels of P3 / P4 meet the power-of-two bounds"""
    fx = _vga()
    o = oracle_lib_built.Oracle(_vga_params(), omp=True)
    try:
        rep = R.run_vga(o, fx, R.Report(strict=True), rasteriser_texcoords=True)
    finally:
        o.close()
    assert len(rep.rows) > 80 and all(ok for _, ok, _ in rep.rows)


@pytest.mark.gpu
def test_hip_path_matches_the_executed_reference_shaders_at_640x480(gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    fx = _vga()
    g = HRBFFusion(_vga_params())
    try:
        rep = R.run_vga(g, fx, R.Report(strict=True))
    finally:
        g.close()
    assert len(rep.rows) > 90 and all(ok for _, ok, _ in rep.rows)


VARIANT_NAMES = sorted(R.VARIANTS)


def test_variant_fixture_covers_every_variant_and_differs_from_the_default():
    var, base = R.load("sphere_variants"), R.load("sphere")
    for name in VARIANT_NAMES:
        ks = [k for k in var if k.startswith(name + "__")]
        assert ks, name
        for k in ks:   # only outputs that the variant changes are stored
            assert not np.array_equal(var[k], base[k.split("__", 1)[1]]), k
    # the removal rules react to the window and the thresholds as the shader's loops say they must
    n = {name: int(var[name + "__x_map_count"][0]) for name in VARIANT_NAMES if name.startswith("clean_")}
    assert n["clean_window_1"] > int(base["x_map_count"][0]) > n["clean_window_4"], n


@pytest.mark.parametrize("name", VARIANT_NAMES)
def test_oracle_matches_the_executed_shaders_under_the_references_parameter_variants(name, oracle_lib_built):
    """the reference's switches of the GLSL rows (Gauss filter, central-difference normals, 5 x 5 curvature window, confidence
    evaluation, clean windows 1 / 2.25 / 4, thresholds, prediction window and neighbour bounds), each executed on llvmpipe"""
    base, var = R.load("sphere"), R.load("sphere_variants")
    o = oracle_lib_built.Oracle(scene_params("sphere", **R.VARIANTS[name][0]), omp=True)
    try:
        rep = R.run_variant(o, base, var, name, R.Report(strict=True))
    finally:
        o.close()
    assert rep.rows and all(ok for _, ok, _ in rep.rows)


@pytest.mark.gpu
@pytest.mark.parametrize("name", VARIANT_NAMES)
def test_hip_path_matches_the_executed_shaders_under_the_references_parameter_variants(name, gpu_available):
    from hrbffusion3d_amd.api import HRBFFusion
    base, var = R.load("sphere"), R.load("sphere_variants")
    g = HRBFFusion(scene_params("sphere", **R.VARIANTS[name][0]))
    try:
        rep = R.run_variant(g, base, var, name, R.Report(strict=True))
    finally:
        g.close()
    assert rep.rows and all(ok for _, ok, _ in rep.rows)


def test_glsl_harness_source_fixes_are_token_level():
    """the harness may only respell what Mesa's compiler rejects: four spellings, no arithmetic"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    from ref_glsl import refgl      # importing needs neither GL nor the reference
    assert len(refgl.SOURCE_FIXES) == 4
    for fname, old, new, why in refgl.SOURCE_FIXES:
        assert fname in ("hrbfbase.glsl", "index_map.vert", "copy_unstable.vert", "resize.frag") and why
        assert (old, new) == ("active", "active_") or old.replace("return 0;", "return 0.0;") == new or \
            (fname == "resize.frag" and old.replace("texture2D(", "texture(") == new)
