"""Generates tests/golden/ref_glsl/*.npz: outputs of the REFERENCE'S OWN GLSL SHADERS, executed.

Runs only in the build container (needs /root/reference and Mesa's llvmpipe): oracle/ref_glsl/refgl.py loads the shader
text from /root/reference/Core/src/Shaders at run time, compiles it with Mesa's GLSL compiler and drives it with the GL
calls the reference's host code makes.  The fixtures are data: per pass, the inputs that were bound and the images /
surfel buffers the shaders wrote.  tests/test_ref_glsl.py feeds the same inputs to the C oracle (CPU suite) and to the
HIP library (GPU suite) and compares.

    python tests/golden/make_ref_glsl.py            # writes the power-of-two fixtures
    python tests/golden/make_ref_glsl.py --vga-fixture  # the whole GPUTest pair at 640 x 480 (vga.npz, coded by tests/ref_glsl_vga.py)
    python tests/golden/make_ref_glsl.py --vga-report   # what differs between the two executions at 640 x 480, and why
    python tests/ref_glsl_report.py oracle|hip           # per-pass comparison of an implementation with the fixtures

Scenes are power-of-two sized (256 x 128 and 128 x 128).  There every texture coordinate, loop bound and half-pixel step of the
shaders is exact in fp32, so the passes have no implementation-defined sampling (DESIGN.md §8 lists what is
implementation-defined at 640 x 480 and how the two executions differ there).
  pair    : a 256 x 128 window of the reference's GPUTest pair (frame 1 seeds the map, frame 2 is fused), the camera
            pose of frame 2 from the oracle's own registration of the two crops (an input here: registration is CUDA)
  sphere  : an analytic scene (sphere on a slanted plane) under fx != fy and an off-centre principal point, second view
            from a different pose

This script writes plain <scene>.npz files; what the tree keeps (and what travels to the GPU box) are their lossless re-codings
<scene>.fxz: run `python tests/fixture_codec.py --encode` afterwards (it verifies the round trip bit for bit) and delete the .npz.
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from hrbffusion3d_amd.params import default_params  # noqa: E402

OUT = os.path.join(HERE, "ref_glsl")
X0, Y0 = 256, 224          # window of the 640 x 480 GPUTest frames
# scene -> (W, H, fx, fy, cx, cy)
GEOM = {"pair": (256, 128, 528.0, 528.0, 320.0 - X0, 240.0 - Y0),
        "sphere": (128, 128, 150.0, 170.0, 60.3, 66.9),    # fx != fy, off-centre principal point
        "vga": (640, 480, 528.0, 528.0, 320.0, 240.0),
        "qqvga_map": (160, 120, 190.0, 200.0, 77.3, 61.9)}  # not a power of two: the vertex shaders' uv attribute differs from the fragment texcoord     # the whole GPUTest frames: report only (--vga-map-report), no fixture


def params(scene, **kw):
    W, H, fx, fy, cx, cy = GEOM[scene]
    return default_params(width=W, height=H, fx=fx, fy=fy, cx=cx, cy=cy, max_surfels=1 << (20 if W * H > (1 << 17) else 17), **kw)


def crop(name):
    W, H = GEOM["pair"][:2]
    return np.ascontiguousarray(np.array(Image.open(os.path.join(HERE, name + ".png")))[Y0:Y0 + H, X0:X0 + W])


def scene_pair():
    from oracle_lib import Oracle
    f1, f2 = (crop("1c"), crop("1d")), (crop("2c"), crop("2d"))
    o = Oracle(params("pair"), omp=True)
    o.process_frame(*f1); o.process_frame(*f2)
    T2, w2 = o.get_pose(), o.get_weighting()
    o.close()
    return f1, f2, T2.astype(np.float32), float(w2)


def scene_sphere(scene="sphere"):
    import scenes
    W, H, FX, FY, CX, CY = GEOM[scene]

    def view(T):
        # depth of a sphere in front of a slanted plane, rendered for camera pose T (camera-to-world)
        Tinv = np.linalg.inv(T)
        c = (Tinv @ np.array([0.05, 0.02, 1.3, 1.0]))[:3]
        n = Tinv[:3, :3] @ np.array([0.25, -0.15, 1.0]); n /= np.linalg.norm(n)
        d0 = 1.9 * n[2] + n @ (Tinv[:3, 3] * 0)   # plane through (0,0,1.9)-ish in this view
        zs = scenes.sphere_depth(W, H, FX, FY, CX, CY, c, 0.35)
        zp = scenes.plane_depth(W, H, FX, FY, CX, CY, n, d0)
        z = np.where(zs > 0, zs, zp)
        # sensor-like noise: without it the reprojected grid of frame 1 lines up with frame 2's rays along whole columns
        # and the association's `dist < bestDist` is decided between exactly equidistant candidates
        z = z + rng.normal(0.0, 0.0006, z.shape) * (z > 0)
        return scenes.gray_rgb(W, H, seed=3), scenes.to_u16(z)
    rng = np.random.default_rng(5)
    T1 = np.eye(4)
    T2 = np.eye(4); T2[:3, 3] = [0.006, -0.004, 0.005]
    a = 0.004; T2[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    return view(T1), view(T2), T2.astype(np.float32), 0.8


def run_reference(scene, f1, f2, T2, w2, prm_over=None, keep_frame1=False):
    """the reference's GL passes in processFrame order (HRBFFusion.cpp:991-1260) for two frames; returns {name: array}"""
    from ref_glsl import refgl
    W, H, FX, FY, CX, CY = GEOM[scene]
    big = W * H > (1 << 17)
    p = refgl.RefPipeline(W, H, FX, FY, CX, CY, 1.0 / 5000.0, prm=prm_over, tex_dim=1024 if big else 512, max_surfels=1 << (20 if big else 17))
    out = {}
    I4 = np.eye(4, dtype=np.float32)

    def pre(tag, rgb, depth):
        p.upload_frame(rgb, depth)
        out[tag + "rgb"], out[tag + "depth"] = rgb, depth
        p.filter_depth(); out[tag + "DEPTH_FILTERED"] = p.get("DEPTH_FILTERED")
        p.metricise_depth()
        out[tag + "DEPTH_METRIC"], out[tag + "DEPTH_METRIC_FILTERED"] = p.get("DEPTH_METRIC"), p.get("DEPTH_METRIC_FILTERED")
        p.compute_vertex_normal_radius()
        for n in ("VERTEX_RAW", "VERTEX_FILTERED", "RADIUS"):
            out[tag + n] = p.get(n)
        out[tag + "NORMAL_P3"] = p.get("NORMAL")                       # NORMAL after computeVertexNormalRadius
        p.compute_curvature_gradient()
        out[tag + "CURV1"], out[tag + "CURV2"] = p.get("PRINCIPAL_CURV1"), p.get("PRINCIPAL_CURV2")
        out[tag + "GRADIENT_MAG"] = p.get("GRADIENT_MAG")
        p.update_normal_rad(); out[tag + "NORMAL"] = p.get("NORMAL")   # NORMAL after updateNormalRad

    def predict(tag, pose, tick, weighting):
        p.predict_indices(pose, tick)
        for k, v in p.index_images().items():
            out[tag + "p_" + k] = v
        p.predict_hrbf()
        for k, v in p.prediction_images().items():
            out[tag + k] = v
        p.fill_in(tick, weighting)
        for k, v in p.fill_images().items():
            out[tag + k] = v

    # frame 1 (tick 1): pre-processing, initialise.  Only the frame and the map it seeds are kept (the passes are compared
    # on frame 2); its prediction at the identity pose is the degenerate raster case (every surfel exactly on a pixel
    # corner) and is not recorded.
    pre("f1_", *f1)
    # (keep_frame1: the 640 x 480 fixture's coder keeps them as helper arrays — the seed map is a gather from them)
    for k in [k for k in out if k.startswith("f1_") and k not in ("f1_rgb", "f1_depth") and not keep_frame1]:
        del out[k]
    p.initialise(I4); out["f1_map"] = p.download_map()
    # frame 2 (tick 2) at pose T2
    pre("f2_", *f2)
    p.vertex_confidence(w2); out["f2_CONFIDENCE"] = p.get("CONFIDENCE")
    out["f2_pose"], out["f2_weighting"] = T2, np.float32(w2)
    def map_flow(tag, map_in):
        """predictIndices, fuse, predictIndices, clean on `map_in` with frame 2 bound (HRBFFusion.cpp:1186-1228)"""
        p.upload_map(map_in)
        p.predict_indices(T2, 2)
        for k, v in p.index_images().items():
            if k in ("INDEX", "INDEX_VERTCONF", "INDEX_NORMRAD"):          # what data.vert reads of the index map
                out[tag + "a_" + k] = v
        p.fuse(T2, 2, w2)
        rec = out[tag + "records"] = p.fuse_records()                      # stage 1: merge marks and new surfels
        fused = p.download_map()                                           # stage 2: only merged surfels change
        ch = np.nonzero((fused.view(np.uint32) != map_in.view(np.uint32)).any(1))[0]
        out[tag + "fused_rows"], out[tag + "fused_vals"] = ch.astype(np.uint32), fused[ch]
        p.predict_indices(T2, 2)
        out[tag + "c_INDEX"] = p.index_images()["INDEX"]                   # index map the clean pass reads (ids only)
        p.clean(T2, 2)
        final = p.download_map()
        # the clean pass copies: survivors of the fused map in order, then the new surfels among the records in order, their
        # time stamp set.  Stored as that structure (keep mask, record picks) and checked to reproduce the buffer bit for bit.
        fb, ob = fused.view(np.uint32), final.view(np.uint32)
        keep = np.zeros(fused.shape[0], bool)
        j = 0
        for i in range(fused.shape[0]):
            if j < final.shape[0] and np.array_equal(fb[i], ob[j]):
                keep[i] = True; j += 1
        picks = []
        rec_new = rec.copy(); rec_new[rec_new[:, 7] == -2.0, 7] = 2.0      # vColor.w = time (copy_unstable.vert:152-155)
        rn = rec_new.view(np.uint32)
        for i in range(rec_new.shape[0]):
            if j < final.shape[0] and rec[i, 7] == -2.0 and np.array_equal(rn[i], ob[j]):
                picks.append(i); j += 1
        assert j == final.shape[0], (j, final.shape)
        out[tag + "keep"] = np.packbits(keep); out[tag + "new_picks"] = np.array(picks, np.uint32)
        out[tag + "map_count"] = np.array([final.shape[0]], np.uint32)
        return final

    # (1) the plain second frame: young map (confidence ~1), nothing is removed, new surfels are appended
    map_flow("f2_", out["f1_map"])
    # (2) a map that has been observed for a while (confidence + 6: stable), plus surfels the clean pass must remove:
    #     floating outliers 4 cm in front of the surface (free-space violation, copy_unstable.vert:124-134), newer duplicates
    #     2 mm in front of stable older surfels (:112-122) and long-unseen unstable surfels (:158-161)
    m = out["f1_map"].copy()
    m[:, 3] += 6.0
    rng = np.random.default_rng(11)
    Tinv = np.linalg.inv(T2.astype(np.float64))
    cam = Tinv[:3, :3] @ m[:, 0:3].T.astype(np.float64) + Tinv[:3, 3:4]
    rays = (cam / np.linalg.norm(cam, axis=0)).T                           # unit view rays in the camera frame
    R = T2[:3, :3].astype(np.float64)

    def shifted(sel, dist, init_time):
        x = m[sel].copy()
        x[:, 0:3] -= (dist * (R @ rays[sel].T).T).astype(np.float32)       # towards the camera
        x[:, 8:11] = (-(R @ rays[sel].T).T).astype(np.float32)             # facing it (|n_z| > 0.85 in the camera frame)
        x[:, 6] = init_time
        return x
    n0 = m.shape[0]
    sel = rng.permutation(n0)
    extra = np.concatenate([shifted(sel[:400], 0.04, 1.0), shifted(sel[400:800], 0.002, 2.0)])
    old = sel[800:1000]
    m[old, 3] = 1.0; m[old, 7] = -250.0                                   # unstable and not seen for > 200 frames
    out["x_extra"], out["x_old"] = extra, old.astype(np.uint32)           # x_map = stable_map(f1_map, x_old) ++ x_extra
    x_map = np.concatenate([m, extra])
    final = map_flow("x_", x_map)
    p.upload_map(final)
    predict("x_", T2, 2, w2)
    # updateModel on the final map: one correction per submap id (only submap 0 exists)
    D = np.eye(4, dtype=np.float32)
    a = 0.01
    D[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    D[:3, 3] = [0.01, -0.02, 0.005]
    p.update_model([D]); out["x_delta"] = D; out["x_map_updated_head"] = p.download_map()[:4096]
    out["tc"] = interpolated_texcoords(p)
    # initialise from frame 2's images at pose T2 (GlobalModel::initialise takes any init_pose)
    p.initialise(T2)
    im = p.download_map()
    out["f2_init_count"], out["f2_init_head"] = np.array([im.shape[0]], np.uint32), im[:4096]
    return out


def clean_structure(fused, rec, final):
    """the clean pass copies survivors of the fused map in order, then the new surfels among the records in order"""
    fb, ob = fused.view(np.uint32), final.view(np.uint32)
    keep = np.zeros(fused.shape[0], bool)
    j = 0
    for i in range(fused.shape[0]):
        if j < final.shape[0] and np.array_equal(fb[i], ob[j]):
            keep[i] = True; j += 1
    picks = []
    rec_new = rec.copy(); rec_new[rec_new[:, 7] == -2.0, 7] = 2.0
    rn = rec_new.view(np.uint32)
    for i in range(rec_new.shape[0]):
        if j < final.shape[0] and rec[i, 7] == -2.0 and np.array_equal(rn[i], ob[j]):
            picks.append(i); j += 1
    assert j == final.shape[0], (j, final.shape)
    return np.packbits(keep), np.array(picks, np.uint32), np.array([final.shape[0]], np.uint32)


def run_reference_variants(f1, f2, T2, w2, base):
    """The reference's parameter variants of the GLSL rows (ref_glsl_check.VARIANTS) on the sphere scene: each variant re-runs
    ONE part of the pipeline with its uniforms changed, on the state the default run (= the committed sphere fixture, asserted)
    leaves in front of that part.  Only that part's outputs are stored."""
    import ref_glsl_check as R
    from ref_glsl import refgl
    W, H, FX, FY, CX, CY = GEOM["sphere"]
    p = refgl.RefPipeline(W, H, FX, FY, CX, CY, 1.0 / 5000.0, tex_dim=512, max_surfels=1 << 17)
    default = dict(p.p)
    out = {}

    def same(a, b, what):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), what

    def pre(upto):
        """frame 2 through the default passes in front of `upto`"""
        p.p.clear(); p.p.update(default)
        p.upload_frame(*f2)
        if upto == "filter":
            return
        p.filter_depth(); p.metricise_depth()
        same(p.get("DEPTH_METRIC_FILTERED"), base["f2_DEPTH_METRIC_FILTERED"], "default run differs from the committed fixture")
        if upto == "vnr":
            return
        p.compute_vertex_normal_radius()
        if upto == "curv":
            return
        p.compute_curvature_gradient(); p.update_normal_rad()
        same(p.get("NORMAL"), base["f2_NORMAL"], "default run differs from the committed fixture")
        if upto == "conf":
            return
        p.vertex_confidence(w2)
        same(p.get("CONFIDENCE"), base["f2_CONFIDENCE"], "default run differs from the committed fixture")

    xm = R.stable_map_with_outliers(base)
    fused_ref = xm.copy(); fused_ref[base["x_fused_rows"]] = base["x_fused_vals"]
    final_ref = R.reference_final(base, "x_", xm)
    for name, (kw, part) in R.VARIANTS.items():
        tag = name + "__"
        pre(part if part in ("filter", "vnr", "curv", "conf") else "all")
        p.p.update({k: v for k, v in kw.items() if k in default})   # frame_to_frame_rgb is an argument of fill_in, not a uniform set
        if part == "filter":
            p.filter_depth(); out[tag + "f2_DEPTH_FILTERED"] = p.get("DEPTH_FILTERED")
            p.metricise_depth()
            out[tag + "f2_DEPTH_METRIC"], out[tag + "f2_DEPTH_METRIC_FILTERED"] = p.get("DEPTH_METRIC"), p.get("DEPTH_METRIC_FILTERED")
        elif part == "vnr":
            p.compute_vertex_normal_radius()
            for n in ("VERTEX_RAW", "VERTEX_FILTERED", "RADIUS"):
                out[tag + "f2_" + n] = p.get(n)
            out[tag + "f2_NORMAL_P3"] = p.get("NORMAL")
        elif part == "curv":
            p.compute_curvature_gradient()
            out[tag + "f2_CURV1"], out[tag + "f2_CURV2"] = p.get("PRINCIPAL_CURV1"), p.get("PRINCIPAL_CURV2")
            out[tag + "f2_GRADIENT_MAG"] = p.get("GRADIENT_MAG")
            p.update_normal_rad(); out[tag + "f2_NORMAL"] = p.get("NORMAL")
        elif part == "conf":
            p.vertex_confidence(w2); out[tag + "f2_CONFIDENCE"] = p.get("CONFIDENCE")
        elif part == "clean":
            # the default association and merge (asserted to reproduce the fixture), then the variant's clean pass
            p.p.clear(); p.p.update(default)
            p.upload_map(xm); p.predict_indices(T2, 2); p.fuse(T2, 2, w2)
            rec = p.fuse_records()
            same(rec, base["x_records"], "records differ from the committed fixture")
            same(p.download_map(), fused_ref, "fused map differs from the committed fixture")
            p.predict_indices(T2, 2)
            p.p.update(kw)
            p.clean(T2, 2)
            out[tag + "x_keep"], out[tag + "x_new_picks"], out[tag + "x_map_count"] = clean_structure(fused_ref, rec, p.download_map())
        elif part == "fuse":
            # association + merge + clean with the variant's uniforms in data.vert (the frame's images are the default run's)
            p.upload_map(xm); p.predict_indices(T2, 2); p.fuse(T2, 2, w2)
            rec = out[tag + "x_records"] = p.fuse_records()
            fused = p.download_map()
            ch = np.nonzero((fused.view(np.uint32) != xm.view(np.uint32)).any(1))[0]
            out[tag + "x_fused_rows"], out[tag + "x_fused_vals"] = ch.astype(np.uint32), fused[ch]
            p.predict_indices(T2, 2)
            out[tag + "x_c_INDEX"] = p.index_images()["INDEX"]
            p.clean(T2, 2)
            out[tag + "x_keep"], out[tag + "x_new_picks"], out[tag + "x_map_count"] = clean_structure(fused, rec, p.download_map())
        elif part == "predict":
            p.upload_map(final_ref); p.predict_indices(T2, 2)
            same(p.index_images()["INDEX"], base["x_p_INDEX"], "index map differs from the committed fixture")
            p.predict_hrbf()
            for k, v in p.prediction_images().items():
                out[tag + "x_" + k] = v
            p.fill_in(2, w2, frame_to_frame_rgb=bool(kw.get("frame_to_frame_rgb", 0)))
            for k, v in p.fill_images().items():
                out[tag + "x_" + k] = v
        print("  variant %-22s %s" % (name, kw))
    # the variants must differ from the default where they are meant to
    for name, (kw, part) in R.VARIANTS.items():
        ks = [k for k in out if k.startswith(name + "__")]
        diff = sum(int(not np.array_equal(np.ascontiguousarray(out[k]).view(np.uint8), np.ascontiguousarray(base[k.split("__", 1)[1]]).view(np.uint8))) for k in ks)
        print("  %-22s %d of %d stored outputs differ from the default run" % (name, diff, len(ks)))
        assert diff > 0, name
    # outputs a variant leaves as the default run wrote them are not stored twice (variant_fixture falls back to the default)
    for k in list(out):
        if np.array_equal(np.ascontiguousarray(out[k]).view(np.uint8), np.ascontiguousarray(base[k.split("__", 1)[1]]).view(np.uint8)):
            del out[k]
    return out


def fp32_window_counts(n, tc, win=3.0):
    """iterations of `for (i = max(0, t - s win); i <= min(1, t + s win); i += s)` in fp32 for every pixel, given the texture
    coordinates tc[p] (hd_window_axis in include/hrbf_detmath.h is this loop with tc = the correctly rounded (p + 0.5) / n)"""
    f = np.float32
    s_ = f(1) / f(n)
    out = []
    for p in range(n):
        t = f(tc[p])
        lo, hi = max(f(0), f(t - f(s_ * f(win)))), min(f(1), f(t + f(s_ * f(win))))
        i, k = lo, 0
        while i <= hi:
            k += 1; i = f(i + s_)
        out.append(k)
    return np.array(out)


def interpolated_texcoords(p):
    """the texture coordinate the rasteriser interpolates for every pixel of the full-screen quad (a fragment shader that writes
    it, behind the reference's quad.geom)"""
    import ctypes as C
    from ref_glsl import refgl, glbind as G
    gl, W, H = p.gl, p.W, p.H
    src = b"#version 330 core\nin vec2 texcoord; out vec4 o; void main(){ o = vec4(texcoord, 0.0, 1.0); }\n"
    pid = gl.glCreateProgram()
    for kind, text in ((G.GL_VERTEX_SHADER, refgl.shader_source("empty.vert").encode()), (G.GL_GEOMETRY_SHADER, refgl.shader_source("quad.geom").encode()),
                       (G.GL_FRAGMENT_SHADER, src)):
        sid = gl.glCreateShader(kind); b = C.c_char_p(text); gl.glShaderSource(sid, 1, C.byref(b), None); gl.glCompileShader(sid); gl.glAttachShader(pid, sid)
    gl.glLinkProgram(pid)

    class P:
        def bind(self): gl.glUseProgram(pid)
        def unbind(self): gl.glUseProgram(0)
        def set(self, k, v): pass
    t = refgl.tex_rgba32f(gl, W, H)
    p._quad_pass(P(), refgl.Fbo(gl, W, H, [t]), [], [])
    tc = refgl.get_f4(t)
    return np.ascontiguousarray(tc[..., :2])      # (H, W, 2); NOT separable in general: the two triangles of the strip are set up separately


def run_reference_nonpow2():
    """P1-P5 at 160 x 120 — NOT a power of two: here the float-stepped window loops of getNormalPCA and the curvature pass take 6
    instead of 7 samples at 88 of the 160 columns and 17 of the 120 rows, and 4 rows of the bilateral filter's taps land a texel
    low.  The fixture records, next to the passes' inputs and outputs, the texture coordinates llvmpipe interpolated, so that the
    check can tell the columns / rows where ITS window differs from the one the correctly rounded coordinate gives."""
    import ctypes as C
    from ref_glsl import refgl, glbind as G
    W, H = 160, 120
    X1, Y1 = 240, 180
    fx = fy = 264.0
    cx, cy = 320.0 / 2 - 40.0, 240.0 / 2 - 30.0     # the GPUTest frames decimated by 2, window at (40, 30) of the 320 x 240 image
    rgb = np.ascontiguousarray(np.array(Image.open(os.path.join(HERE, "2c.png")))[::2, ::2][30:30 + H, 40:40 + W])
    depth = np.ascontiguousarray(np.array(Image.open(os.path.join(HERE, "2d.png")))[::2, ::2][30:30 + H, 40:40 + W])
    p = refgl.RefPipeline(W, H, fx, fy, cx, cy, 1.0 / 5000.0, tex_dim=64, max_surfels=1024)
    out = {"geom": np.array([W, H, fx, fy, cx, cy], np.float64), "rgb": rgb, "depth": depth}
    p.upload_frame(rgb, depth)
    p.filter_depth(); out["DEPTH_FILTERED"] = p.get("DEPTH_FILTERED")
    p.metricise_depth(); out["DEPTH_METRIC"], out["DEPTH_METRIC_FILTERED"] = p.get("DEPTH_METRIC"), p.get("DEPTH_METRIC_FILTERED")
    p.compute_vertex_normal_radius()
    for n in ("VERTEX_RAW", "VERTEX_FILTERED", "RADIUS"):
        out[n] = p.get(n)
    out["NORMAL_P3"] = p.get("NORMAL")
    p.compute_curvature_gradient()
    out["CURV1"], out["CURV2"], out["GRADIENT_MAG"] = p.get("PRINCIPAL_CURV1"), p.get("PRINCIPAL_CURV2"), p.get("GRADIENT_MAG")
    p.update_normal_rad(); out["NORMAL"] = p.get("NORMAL")
    tc = interpolated_texcoords(p)
    assert (tc[..., 0] == tc[0, :, 0][None, :]).all() and (tc[..., 1] == tc[:, 0, 1][:, None]).all()     # separable at this size
    out["tc_x"], out["tc_y"] = tc[0, :, 0].copy(), tc[:, 0, 1].copy()
    f = np.float32
    ideal_x = ((np.arange(W, dtype=f) + f(0.5)) / f(W)).astype(f); ideal_y = ((np.arange(H, dtype=f) + f(0.5)) / f(H)).astype(f)
    cxi, cxl = fp32_window_counts(W, ideal_x), fp32_window_counts(W, out["tc_x"])
    cyi, cyl = fp32_window_counts(H, ideal_y), fp32_window_counts(H, out["tc_y"])
    out["win_x"], out["win_y"] = cxi.astype(np.uint8), cyi.astype(np.uint8)                 # iterations under the correctly rounded coordinate
    out["tie_cols"], out["tie_rows"] = np.nonzero(cxi != cxl)[0].astype(np.int32), np.nonzero(cyi != cyl)[0].astype(np.int32)
    c = np.arange(H, dtype=f)
    out["tap_rows_low"] = np.nonzero(np.floor((c / f(H)) * f(H)) != c)[0].astype(np.int32)  # bilateral taps one texel low under fp32 floor
    return out


THUMB_SIZES = [(640, 480), (160, 120), (256, 128), (128, 128)]


def run_reference_thumbnail():
    """Resize::vertex (resize.frag behind quad.geom into a (W / 20) x (H / 20) target, NEAREST) on a predicted vertex map whose
    texels carry their own coordinates: which texel every cell of denseEnough's thumbnail reads."""
    from ref_glsl import refgl
    out = {"sizes": np.array(THUMB_SIZES, np.int32)}
    for W, H in THUMB_SIZES:
        p = refgl.RefPipeline(W, H, float(W), float(W), W / 2.0, H / 2.0, 1.0 / 5000.0, tex_dim=64, max_surfels=1024)
        ys, xs = np.mgrid[0:H, 0:W]
        v = np.zeros((H, W, 4), np.float32); v[..., 0] = xs; v[..., 1] = ys; v[..., 2] = 1.0; v[..., 3] = xs + ys * W
        p.pr_vertex.upload(v)
        t = p.dense_thumbnail()
        assert t.shape == (H // 20, W // 20, 4) and (t[..., 2] == 1.0).all()
        assert (t[..., 0] == t[0, :, 0][None]).all() and (t[..., 1] == t[:, 0, 1][:, None]).all()     # separable
        assert (t[..., 3] == t[..., 0] + t[..., 1] * W).all()                                             # one texel, not a blend
        out["sx_%dx%d" % (W, H)] = t[0, :, 0].astype(np.int32); out["sy_%dx%d" % (W, H)] = t[:, 0, 1].astype(np.int32)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-thumbnail" in sys.argv:
        q = run_reference_thumbnail()
        path = os.path.join(OUT, "thumbnail.npz")
        np.savez_compressed(path, **q)
        print("thumbnail ->", path, {k: v.tolist() for k, v in q.items() if k.startswith("sy_")})
        return
    q = run_reference_nonpow2()
    path = os.path.join(OUT, "qqvga_pre.npz")
    np.savez_compressed(path, **q)
    print("qqvga_pre ->", path, "%.1f MB" % (os.path.getsize(path) / 1e6), "6-sample columns / rows:", int((q["win_x"] == 6).sum()), int((q["win_y"] == 6).sum()),
          "tie columns / rows:", q["tie_cols"].tolist(), q["tie_rows"].tolist(), "low tap rows:", q["tap_rows_low"].tolist())
    if "--only-nonpow2" in sys.argv:
        return
    if "--only-nonpow2-map" in sys.argv:
        # the map passes of the second frame at 160 x 120: association (data.vert with the host-computed uv attribute), merge,
        # index map, clean — only what those checks read is kept
        f1, f2, T2, w2 = scene_sphere("qqvga_map")
        full = run_reference("qqvga_map", f1, f2, T2, w2)
        keep = ["f1_map", "f2_rgb", "f2_depth", "f2_pose", "f2_weighting", "f2_DEPTH_FILTERED", "f2_DEPTH_METRIC", "f2_DEPTH_METRIC_FILTERED",
                "f2_VERTEX_RAW", "f2_VERTEX_FILTERED", "f2_RADIUS", "f2_NORMAL_P3", "f2_NORMAL", "f2_CURV1", "f2_CURV2", "f2_GRADIENT_MAG",
                "f2_CONFIDENCE", "f2_a_INDEX", "f2_a_INDEX_VERTCONF", "f2_a_INDEX_NORMRAD", "f2_records", "f2_fused_rows", "f2_fused_vals",
                "f2_c_INDEX", "f2_keep", "f2_new_picks", "f2_map_count", "f2_init_count", "f2_init_head",
                # the stable map with planted outliers / duplicates / stale surfels: the removal rules of copy_unstable.vert
                "x_extra", "x_old", "x_a_INDEX", "x_a_INDEX_VERTCONF", "x_a_INDEX_NORMRAD", "x_records", "x_fused_rows", "x_fused_vals",
                "x_c_INDEX", "x_keep", "x_new_picks", "x_map_count"]
        out = {k: full[k] for k in keep}
        out["geom"] = np.array(GEOM["qqvga_map"], np.float64)
        f = np.float32
        for n, key in ((160, "uv_cols_differ"), (120, "uv_rows_differ")):
            i = np.arange(n)
            ta = ((i.astype(f) / f(n)).astype(np.float64) + 1.0 / float(2 * f(n))).astype(f)
            out[key] = np.nonzero(ta != ((i.astype(f) + f(0.5)) / f(n)).astype(f))[0].astype(np.int32)
        path = os.path.join(OUT, "qqvga_map.npz")
        np.savez_compressed(path, **out)
        print("qqvga_map ->", path, "%.1f MB" % (os.path.getsize(path) / 1e6), "surfels", out["f1_map"].shape[0], "->", int(out["f2_map_count"][0]),
              "merge marks", int((out["f2_records"][:, 7] == -1).sum()), "| stable flow removed",
              int((~np.unpackbits(out["x_keep"])[:out["f1_map"].shape[0] + out["x_extra"].shape[0]].astype(bool)).sum()), "| uv attribute differs at", len(out["uv_cols_differ"]), "columns /", len(out["uv_rows_differ"]), "rows")
        return
    if "--only-variants" in sys.argv:
        import ref_glsl_check as R
        f1, f2, T2, w2 = scene_sphere()
        out = run_reference_variants(f1, f2, T2, w2, R.load("sphere"))
        path = os.path.join(OUT, "sphere_variants.npz")
        np.savez_compressed(path, **out)
        print("sphere_variants ->", path, "%.1f MB" % (os.path.getsize(path) / 1e6), len(out), "arrays")
        return
    for name, scene in (("pair", scene_pair), ("sphere", scene_sphere)):
        f1, f2, T2, w2 = scene()
        out = run_reference(name, f1, f2, T2, w2)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "->", path, "%.1f MB" % (os.path.getsize(path) / 1e6), "surfels:", out["f1_map"].shape[0], "->",
              int(out["f2_map_count"][0]), "records:", out["f2_records"].shape[0], "| stable map + outliers:", out["f1_map"].shape[0] + out["x_extra"].shape[0], "->",
              int(out["x_map_count"][0]), "predicted pixels:", int((out["x_PRED_VERTEX"][..., 2] != 0).sum()))


def vga_report():
    """What is implementation-defined at 640 x 480: frame 1 of the GPUTest pair through P1-P4 on llvmpipe and
    on the oracle, each pass on the reference's own input images."""
    from ref_glsl import refgl
    from oracle_lib import Oracle
    import ref_glsl_check as R
    W, H = 640, 480
    rgb = np.array(Image.open(os.path.join(HERE, "1c.png"))); d = np.array(Image.open(os.path.join(HERE, "1d.png")))
    prm = default_params(max_surfels=1 << 20)
    p = refgl.RefPipeline(W, H, prm.fx, prm.fy, prm.cx, prm.cy, prm.depth_scale, tex_dim=64, max_surfels=1024)
    o = Oracle(prm, omp=True)
    p.upload_frame(rgb, d); o.upload_frame(rgb, d)
    f = np.float32
    print("== texel-boundary taps: texture(s, vec2(float(c) / n, ..)) with NEAREST filtering samples floor(fl(fl(c / n) * n))")
    for n in (640, 480):
        c = np.arange(n, dtype=f)
        print("   n = %d: columns/rows whose tap lands one texel low under fp32 floor: %s" % (n, np.nonzero(np.floor((c / f(n)) * f(n)) != c)[0].tolist()))
    p.filter_depth(); o.run_stage("FILTER_DEPTH")
    a, b = p.get("DEPTH_FILTERED").astype(np.float64), o.get_image("DEPTH_FILTERED").astype(np.float64)
    rel = np.abs(a - b) / np.maximum(b, 1.0)
    rows = np.nonzero(rel.max(1) > 1e-5)[0]
    print("P1 bilateral: rows with a relative difference > 1e-5: %s" % rows.tolist())
    print("   everywhere else: max relative difference %.2e" % np.delete(rel, rows, 0).max())
    print("== float-stepped window loops: for (i = t - 3 s; i <= t + 3 s; i += s), s = 1 / n, t = (p + 0.5) / n, all fp32")
    counts = {}
    for n in (640, 480, 256, 128):
        s_ = f(1) / f(n); cnt = []
        for px in range(n):
            t = f(f(px) + f(0.5)) / f(n)
            lo, hi = max(f(0), f(t - f(s_ * f(3)))), min(f(1), f(t + f(s_ * f(3))))
            i, k = lo, 0
            while i <= hi:
                k += 1; i = f(i + s_)
            cnt.append(k)
        cnt = counts[n] = np.array(cnt)
        print("   n = %d: iterations per pixel 7 (nominal): %d, 6: %d, 5 / 4 (image border): %d" % (n, (cnt == 7).sum(), (cnt == 6).sum(), (cnt < 6).sum()))
    o.set_image("DEPTH_FILTERED", p.get("DEPTH_FILTERED"))
    p.metricise_depth(); o.run_stage("METRICISE")
    p.compute_vertex_normal_radius(); o.run_stage("VERTEX_NORMAL_RADIUS")
    for nme in ("VERTEX_RAW", "VERTEX_FILTERED"):
        print("P3 %s xyz identical: %s" % (nme, bool((R.ulp_diff(p.get(nme)[..., :3], o.get_image(nme)[..., :3]) == 0).all())))
    a, b = p.get("NORMAL"), o.get_image("NORMAL")
    v = (np.linalg.norm(a[..., :3], axis=-1) > 0.5) & (np.linalg.norm(b[..., :3], axis=-1) > 0.5)
    ang = np.degrees(np.arccos(np.clip((a[..., :3] * b[..., :3]).sum(-1)[v], -1, 1)))
    print("P3 PCA normal, 7 x 7 nominal window (oracle) vs fp32-stepped window (llvmpipe): validity differs at %d pixels; angle median %.2f deg, p90 %.2f, p99 %.2f" % (
        int(((np.linalg.norm(a[..., :3], axis=-1) > 0.5) != (np.linalg.norm(b[..., :3], axis=-1) > 0.5)).sum()), np.median(ang), np.percentile(ang, 90), np.percentile(ang, 99)))
    full = ((counts[640] == 7)[None, :] & (counts[480] == 7)[:, None])   # pixels whose fp32 window is the nominal 7 x 7
    angf = np.degrees(np.arccos(np.clip((a[..., :3] * b[..., :3]).sum(-1), -1, 1)))
    print("   pixels with the nominal window under fp32 (%.1f%% of the image): angle median %.3f deg, p99 %.2f; the others: median %.2f, p99 %.2f" % (
        100 * full.mean(), np.median(angf[v & full]), np.percentile(angf[v & full], 99), np.median(angf[v & ~full]), np.percentile(angf[v & ~full], 99)))
    print("   (the sample positions of getNormalPCA are fl(i * cols) with i the accumulated loop variable, geometry.glsl:209: ~1e-5 pixel of\n"
          "    noise at 640 x 480 that the covariance's cancellation amplifies; exact at power-of-two sizes, where the normals agree to 3 ulp)")
    o.set_image("VERTEX_FILTERED", p.get("VERTEX_FILTERED")); o.set_image("NORMAL", p.get("NORMAL"))
    p.compute_curvature_gradient(); o.run_stage("CURVATURE")
    gm, gmo = p.get("GRADIENT_MAG"), o.get_image("GRADIENT_MAG")
    ok = (gm != 0) & (gmo != 0)
    print("P4 gradient magnitude: within 1024 ulp at %.1f%% of the pixels; at %.2f%% of those with the nominal window under fp32 (%.1f%% of the image)" % (
        100 * (R.ulp_diff(gm, gmo)[ok] <= 1024).mean(), 100 * (R.ulp_diff(gm, gmo)[ok & full] <= 1024).mean(), 100 * full.mean()))
    k, ko = p.get("PRINCIPAL_CURV1")[..., 3], o.get_image("CURV1")[..., 3]
    ok = (k != 1000) & (ko != 1000) & np.isfinite(k) & np.isfinite(ko)
    err = np.abs(k[ok] - ko[ok]) / (np.abs(ko[ok]) + 1)
    print("P4 k1: |dk| / (|k| + 1): median %.2e, p90 %.2e, p99 %.2e" % (np.median(err), np.percentile(err, 90), np.percentile(err, 99)))
    okf = ok & full
    errf = np.abs(k[okf] - ko[okf]) / (np.abs(ko[okf]) + 1)
    print("   restricted to pixels with the nominal window: median %.2e, p90 %.2e, p99 %.2e" % (np.median(errf), np.percentile(errf, 90), np.percentile(errf, 99)))


def vga_map_report():
    """Every GLSL pass at 640 x 480 — the benchmark's resolution — on the whole GPUTest pair: the reference's shaders on llvmpipe
    against the oracle, through the same checks as the committed (power-of-two) fixtures, printed instead of asserted: which bounds
    hold unchanged at a size that is not a power of two, and what the implementation-defined taps cost where they do not."""
    import ref_glsl_check as R
    from oracle_lib import Oracle
    from PIL import Image as _I
    f1 = (np.array(_I.open(os.path.join(HERE, "1c.png"))), np.array(_I.open(os.path.join(HERE, "1d.png"))))
    f2 = (np.array(_I.open(os.path.join(HERE, "2c.png"))), np.array(_I.open(os.path.join(HERE, "2d.png"))))
    o = Oracle(params("vga"), omp=True)
    o.process_frame(*f1); o.process_frame(*f2)
    T2, w2 = o.get_pose().astype(np.float32), float(o.get_weighting())
    o.close()
    fx = run_reference("vga", f1, f2, T2, w2)
    print("reference shaders at 640 x 480: %d surfels seeded, %d after frame 2; stable map + outliers %d -> %d; %d predicted pixels" % (
        fx["f1_map"].shape[0], int(fx["f2_map_count"][0]), fx["f1_map"].shape[0] + fx["x_extra"].shape[0], int(fx["x_map_count"][0]),
        int((fx["x_PRED_VERTEX"][..., 2] != 0).sum())))
    o = Oracle(params("vga"), omp=True)
    rep = R.run(o, fx, R.Report(strict=False, verbose=True), own_pca_normals=True)
    o.close()
    bad = [w for w, ok, _ in rep.rows if not ok]
    print("%d checks, %d outside the bounds of the power-of-two fixtures: %s" % (len(rep.rows), len(bad), bad))


QQVGA_VARIANTS = [dict(clean_window_multiplier=1.0), dict(clean_window_multiplier=2.25), dict(clean_window_multiplier=3.0),
                  dict(predict_window_multiplier=2.0, predict_min_neighbors=4, predict_max_neighbors=6), dict(predict_window_multiplier=4.0),
                  dict(normal_estimation_pca=0.0), dict(init_radius_multiplier=3.0)]


def qqvga_variants_report():
    """Parameter variants x a size that is not a power of two: the whole two-frame flow of `qqvga_map` (160 x 120) through the reference's
    shaders with the variant's uniforms, against the oracle with the same parameters — printed, not a fixture.  The window walks that
    are literal fp32 loops (clean pass, HRBF windows) are exact at power-of-two sizes, where the committed variants live; this is the
    combination they do not cover.  Pre-processing rows are left out (llvmpipe's interpolated coordinate decides ties there)."""
    import ref_glsl_check as R
    from oracle_lib import Oracle
    f1, f2, T2, w2 = scene_sphere("qqvga_map")
    for kw in QQVGA_VARIANTS:
        fx = run_reference("qqvga_map", f1, f2, T2, w2, prm_over=kw)
        if "normal_estimation_pca" in kw:
            fx["_normal_abs_floor"] = 2e-5          # central differences: the cross product cancels (as in the committed fuse_central_diff variant)
        o = Oracle(params("qqvga_map", **kw), omp=True)
        rep = R.run(o, fx, R.Report(strict=False, verbose=False), own_pca_normals=True)
        o.close()
        rows = [(w, ok, d) for w, ok, d in rep.rows if not w.startswith(("P1", "P2", "P3", "P4", "P5"))]
        bad = [(w, d) for w, ok, d in rows if not ok]
        print("%s: %d map-pass checks, %d outside the bounds of the power-of-two fixtures %s" % (kw, len(rows), len(bad), bad))


def vga_inputs():
    """the whole GPUTest pair at 640 x 480 and the pose / weighting the oracle's registration finds for frame 2"""
    from oracle_lib import Oracle
    f1 = (np.array(Image.open(os.path.join(HERE, "1c.png"))), np.array(Image.open(os.path.join(HERE, "1d.png"))))
    f2 = (np.array(Image.open(os.path.join(HERE, "2c.png"))), np.array(Image.open(os.path.join(HERE, "2d.png"))))
    o = Oracle(params("vga"), omp=True)
    o.process_frame(*f1); o.process_frame(*f2)
    T2, w2 = o.get_pose().astype(np.float32), float(o.get_weighting())
    o.close()
    return f1, f2, T2, w2


def dump_vga(path):
    """the full, unreduced output of every pass at 640 x 480 (hundreds of MB: scratch, never committed)"""
    from ref_glsl import refgl
    f1, f2, T2, w2 = vga_inputs()
    fx = run_reference("vga", f1, f2, T2, w2)
    from ref_glsl import glbind as G
    fx["renderer"] = np.array(G.GL(compat=True).glGetString(G.GL_RENDERER).decode())     # glh_init is idempotent: the run's context
    print("GL_RENDERER:", fx["renderer"])
    np.savez(path, **fx)
    print("vga dump ->", path, "%.1f MB" % (os.path.getsize(path) / 1e6))


def vga_rasteriser_report():
    """Every GLSL pass at 640 x 480 on the rasteriser GALLIUM_DRIVER selects (llvmpipe by default, softpipe: Mesa's second software
    rasteriser — C code, no JIT, its own texture addressing, interpolation and point rasterisation) against the oracle, through the
    checks of the committed fixture (ref_glsl_check.run_vga), printed instead of asserted.  What the two executions agree on is evidence
    about the reference; where they differ from each other the behaviour is the GL implementation's."""
    import ref_glsl_check as R
    from oracle_lib import Oracle
    from ref_glsl import glbind as G
    f1, f2, T2, w2 = vga_inputs()
    fx = run_reference("vga", f1, f2, T2, w2, keep_frame1=True)
    renderer = G.GL(compat=True).glGetString(G.GL_RENDERER).decode()
    print("GL_RENDERER:", renderer)
    W, H = 640, 480
    f = np.float32
    ys, xs = np.mgrid[0:H, 0:W]
    ideal = np.stack([(xs.astype(f) + f(0.5)) / f(W), (ys.astype(f) + f(0.5)) / f(H)], -1).astype(f)
    tc = fx["tc"]
    u = lambda a, b: np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32))
    print("interpolated texcoord vs the correctly rounded (p + 0.5) / n: x differs at %d pixels (max %d ulp), y at %d (max %d ulp); both exact at %.1f%%" % (
        int((tc[..., 0] != ideal[..., 0]).sum()), int(u(tc[..., 0], ideal[..., 0]).max()), int((tc[..., 1] != ideal[..., 1]).sum()),
        int(u(tc[..., 1], ideal[..., 1]).max()), 100 * (tc == ideal).all(-1).mean()))
    ids = fx["f2_a_INDEX"]
    print("index map: largest surfel id drawn %d of %d surfels%s" % (int(ids.max()), fx["f1_map"].shape[0],
          "  <-- gl_VertexID restarts every 4096 vertices on this rasteriser: its map passes are not comparable" if ids.max() < fx["f1_map"].shape[0] // 2 else ""))
    for mode in (False, True):
        print("---- oracle %s" % ("given this rasteriser's interpolated texcoords (test hook)" if mode else "with correctly rounded texcoords + the exact-texcoord mask"))
        o = Oracle(params("vga"), omp=True)
        try:
            rep = R.run_vga(o, fx, R.Report(strict=False, verbose=True), rasteriser_texcoords=mode)
        except Exception as e:      # a rasteriser whose map passes are broken can trip the checks' own assumptions
            print("   (stopped: %r)" % (e,))
            rep = None
        o.close()
        if rep is not None:
            bad = [w for w, ok, _ in rep.rows if not ok]
            print("%d checks, %d outside the bounds: %s" % (len(rep.rows), len(bad), bad))


VGA_VARIANTS = QQVGA_VARIANTS + [dict(use_bilateral=0), dict(curv_estimation_window=2.0), dict(use_conf_eval=1), dict(depth_cutoff=1.5),
                                 dict(confidence_threshold=9.0, curv_valid_threshold=40.0), dict(predict_conf_threshold=6.6)]


def vga_variants_report():
    """The reference's switches x the benchmark's resolution: the whole GPUTest pair at 640 x 480 through the reference's shaders with
    each variant's uniforms, against the oracle with the same parameters, through run_vga's checks (the oracle is given the
    rasteriser's texcoords, so no mask) — printed, not a fixture (the committed variants live on a 128 x 128 scene)."""
    import ref_glsl_check as R
    from oracle_lib import Oracle
    f1, f2, T2, w2 = vga_inputs()
    for kw in VGA_VARIANTS:
        fx = run_reference("vga", f1, f2, T2, w2, prm_over={k: v for k, v in kw.items()})
        if "normal_estimation_pca" in kw:
            fx["_normal_abs_floor"] = 2e-5
        o = Oracle(params("vga", **kw), omp=True)
        try:
            rep = R.run_vga(o, fx, R.Report(strict=False, verbose=False), rasteriser_texcoords=True)
            rows = rep.rows
            bad = [(w, d) for w, ok, d in rows if not ok]
            print("%s: %d checks, %d outside the bounds %s" % (kw, len(rows), len(bad), bad))
        except Exception as e:
            print("%s: stopped: %r" % (kw, e))
        o.close()


def vga_fixture():
    """tests/golden/ref_glsl/vga.npz: every pass on the WHOLE GPUTest pair at 640 x 480 (the benchmark's resolution), coded
    losslessly by tests/ref_glsl_vga.py (177 MB of arrays -> 8 MB: most are exact functions of the others, the rest within ulps of the oracle's)."""
    import ref_glsl_vga as V
    from ref_glsl import glbind as G
    f1, f2, T2, w2 = vga_inputs()
    fx = run_reference("vga", f1, f2, T2, w2, keep_frame1=True)
    renderer = G.GL(compat=True).glGetString(G.GL_RENDERER).decode()
    path = os.path.join(OUT, "vga.npz")
    sizes = V.encode(fx, path, {"renderer": renderer, "scene": "GPUTest 1c/1d -> 2c/2d, 640 x 480, K = (528, 528, 320, 240), pose of frame 2 from the oracle's registration"})
    kinds = {}
    for k, (kind, sz, nb, _) in sizes.items():
        a = kinds.setdefault(kind, [0, 0, 0]); a[0] += 1; a[1] += sz; a[2] += nb
    print("vga ->", path, "%.1f MB on" % (os.path.getsize(path) / 1e6), renderer, "|", ", ".join("%s: %d arrays, %.1f MB stored for %.1f MB" % (k, v[0], v[1] / 1e6, v[2] / 1e6) for k, v in kinds.items()))
    back = V.decode(path)
    for k, v in back.items():
        if k != "_info":
            assert np.array_equal(np.ascontiguousarray(v).view(np.uint8).ravel(), np.ascontiguousarray(fx[k]).view(np.uint8).ravel()), k
    print("decodes to the run's bits: %d arrays; %d surfels seeded, %d after frame 2; stable map + outliers %d -> %d; %d predicted pixels" % (
        len(fx), fx["f1_map"].shape[0], int(fx["f2_map_count"][0]), fx["f1_map"].shape[0] + fx["x_extra"].shape[0], int(fx["x_map_count"][0]),
        int((fx["x_PRED_VERTEX"][..., 2] != 0).sum())))


if __name__ == "__main__":
    if "--vga-fixture" in sys.argv:
        vga_fixture()
    elif "--vga-rasteriser-report" in sys.argv:
        vga_rasteriser_report()
    elif "--vga-variants-report" in sys.argv:
        vga_variants_report()
    elif "--dump-vga" in sys.argv:
        dump_vga(sys.argv[sys.argv.index("--dump-vga") + 1])
    elif "--qqvga-variants-report" in sys.argv:
        qqvga_variants_report()
    elif "--vga-map-report" in sys.argv:
        vga_map_report()
    elif "--vga-report" in sys.argv:
        vga_report()
    else:
        main()
