"""TEST INFRASTRUCTURE — oracle/_ref: the reference's own GLSL, executed.

The GLSL rows of SURVEY.md §8a (P1-P5, M1, F1-F4, H1-H3, f-3) are shader programs the reference loads from
Core/src/Shaders/*.{vert,geom,frag,glsl} at run time.  This module loads those files FROM /root/reference (they are
never copied into this repository), compiles them with Mesa's GLSL compiler and runs them on llvmpipe through an X-less
context (gl_headless.c), issuing the GL calls the reference's host code issues (cited per method).  What comes out are
the reference's own results for a given input — the vectors tests/golden/make_ref_glsl.py commits as fixtures and that
pin both the C oracle and the HIP path.

What the harness owns (and the reference gets from Pangolin, which is not vendored):
  * include expansion: `#include "x"` → text of Shaders/x, as pangolin::GlSlProgram::ParseGLSL does
    (Core/src/Shaders/Shaders.h:74-116 passes the shader directory as include path);
  * texture objects: pangolin::GlTexture(w, h, internal, sampling_linear, border, fmt, type) = glTexImage2D + MIN/MAG
    filter NEAREST (LINEAR when `draw`) + WRAP_S/T CLAMP_TO_EDGE;
  * frame buffers: pangolin::GlFramebuffer::AttachColour → COLOR_ATTACHMENTn + glDrawBuffers(n+1),
    AttachDepth → DEPTH_COMPONENT24 render buffer.
  * global state of the GUI's context (GUI/src/Tools/GUI.h:53-71): UNPACK/PACK_ALIGNMENT 1, DEPTH_TEST on, GL_LESS,
    depth writes on.

Token-level source fixes (SOURCE_FIXES): Mesa's compiler is stricter than NVIDIA's on three spellings; none touches
arithmetic.  Each fix is applied only where listed and is asserted to hit.

Harness-side deviations from the reference's host constants, chosen so that a CPU rasteriser finishes:
  * GlobalModel::TEXTURE_DIMENSION (4596, GlobalModel.cpp:21) is a parameter `tex_dim` (default 1024): the 5 x RGBA32F
    update textures are tex_dim^2 texels; surfel ids only need id < tex_dim^2.  update.vert / data.vert receive it as
    the uniform `texDim` exactly as the reference passes its constant.
  * MAX_VERTICES-sized vertex buffers are allocated for `max_surfels` records instead of 4596^2.
  * DELTA_TRANS_DIMENSION / ACTIVE_KEYFRAME_DIMENSION (19200, GlobalModel.cpp:25-26, IndexMap.cpp:24) are 16384 here
    (llvmpipe's GL_MAX_TEXTURE_SIZE); the shaders take them as uniforms and only address texel (id + 0.5) / dimension.
  * LUMINANCE32F / LUMINANCEnnUI colour attachments are not colour-renderable on Mesa; those textures are created as
    R32F / RnnUI.  Every shader reads them through float(texture(...)) / uint(texture(...)), i.e. the first channel.
"""
import ctypes as C
import os
import re

import numpy as np

from . import glbind as G

SHADER_DIR = "/root/reference/Core/src/Shaders"

# (file, literal text, replacement, reason).  Applied to the text of `file` only; must match exactly once.
SOURCE_FIXES = [
    ("hrbfbase.glsl", "        return 0;\n    float r = sqrt(d2 / T2);", "        return 0.0;\n    float r = sqrt(d2 / T2);",
     "hrbfbase.glsl:11 `return 0;` in `float getWeight`: int->float conversion of a return value is a GLSL >= 4.20 rule"),
    ("index_map.vert", "active", "active_",
     "index_map.vert:43,45 identifier `active` is a reserved word in GLSL 3.30 (Mesa enforces it)"),
    ("copy_unstable.vert", "active", "active_",
     "copy_unstable.vert:101,133 identifier `active` is a reserved word in GLSL 3.30"),
    ("resize.frag", "texture2D(eSampler, texcoord.xy)", "texture(eSampler, texcoord.xy)",
     "resize.frag:31 `texture2D` was renamed `texture` in GLSL 1.30 and is gone from the 3.30 core profile (NVIDIA still accepts it); same function"),
]

_INC = re.compile(r'^[ \t]*#include[ \t]*"([^"]+)"[ \t]*\r?$', re.M)


def shader_source(name, _depth=0):
    """Text of Shaders/<name> with includes expanded (pangolin ParseGLSL) and SOURCE_FIXES applied."""
    with open(os.path.join(SHADER_DIR, name), "r", encoding="latin-1") as f:
        txt = f.read()
    for fname, old, new, _why in SOURCE_FIXES:
        if fname == name:
            if old == "active":
                txt, n = re.subn(r"\bactive\b", new, txt)
                assert n >= 2, (name, n)
            else:
                assert txt.count(old) == 1, (name, old)
                txt = txt.replace(old, new)
    assert _depth < 4
    return _INC.sub(lambda m: shader_source(m.group(1), _depth + 1), txt)


class Tex:
    """pangolin::GlTexture as GPUTexture builds it (Core/src/GPUTexture.cpp:41-63)."""

    def __init__(self, gl, w, h, internal, fmt, typ, linear=False):
        self.gl, self.w, self.h, self.internal, self.fmt, self.typ = gl, w, h, internal, fmt, typ
        t = C.c_uint(0)
        gl.glGenTextures(1, C.byref(t))
        self.tid = t.value
        gl.glBindTexture(G.GL_TEXTURE_2D, self.tid)
        gl.glTexImage2D(G.GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, typ, None)
        flt = G.GL_LINEAR if linear else G.GL_NEAREST
        gl.glTexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_MIN_FILTER, flt)
        gl.glTexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_MAG_FILTER, flt)
        gl.glTexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_WRAP_S, G.GL_CLAMP_TO_EDGE)
        gl.glTexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_WRAP_T, G.GL_CLAMP_TO_EDGE)
        gl.glBindTexture(G.GL_TEXTURE_2D, 0)

    def upload(self, a, fmt=None, typ=None):
        a = np.ascontiguousarray(a)
        gl = self.gl
        gl.glBindTexture(G.GL_TEXTURE_2D, self.tid)
        gl.glTexSubImage2D(G.GL_TEXTURE_2D, 0, 0, 0, self.w, self.h, fmt or self.fmt, typ or self.typ,
                           a.ctypes.data_as(C.c_void_p))
        gl.glBindTexture(G.GL_TEXTURE_2D, 0)

    def download(self, fmt, typ, dtype, ch):
        o = np.zeros((self.h, self.w, ch) if ch > 1 else (self.h, self.w), dtype)
        gl = self.gl
        gl.glBindTexture(G.GL_TEXTURE_2D, self.tid)
        gl.glGetTexImage(G.GL_TEXTURE_2D, 0, fmt, typ, o.ctypes.data_as(C.c_void_p))
        gl.glBindTexture(G.GL_TEXTURE_2D, 0)
        return o


def tex_rgba32f(gl, w, h):   # GL_RGBA32F, GL_LUMINANCE, GL_FLOAT (HRBFFusion.cpp:826-878)
    return Tex(gl, w, h, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT)


def tex_f1(gl, w, h):        # GL_LUMINANCE32F_ARB in the reference; R32F here (see module docstring)
    return Tex(gl, w, h, G.GL_R32F, G.GL_RED, G.GL_FLOAT)


def tex_u16(gl, w, h):       # GL_LUMINANCE16UI_EXT
    return Tex(gl, w, h, G.GL_R16UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT)


def tex_u32(gl, w, h):       # GL_LUMINANCE32UI_EXT
    return Tex(gl, w, h, G.GL_R32UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_INT)


def tex_rgba8(gl, w, h, linear=False):   # GL_RGBA, GL_RGB, GL_UNSIGNED_BYTE
    return Tex(gl, w, h, G.GL_RGBA, G.GL_RGB, G.GL_UNSIGNED_BYTE, linear)


def get_f4(t):
    return t.download(G.GL_RGBA, G.GL_FLOAT, np.float32, 4)


def get_f1(t):
    return t.download(G.GL_RED, G.GL_FLOAT, np.float32, 1)


def get_u32(t):
    return t.download(G.GL_RED_INTEGER, G.GL_UNSIGNED_INT, np.uint32, 1)


def get_rgba8(t):
    return t.download(G.GL_RGBA, G.GL_UNSIGNED_BYTE, np.uint8, 4)


class Fbo:
    """pangolin::GlFramebuffer + GlRenderBuffer (depth)."""

    def __init__(self, gl, w, h, colours):
        self.gl, self.w, self.h = gl, w, h
        f = C.c_uint(0)
        gl.glGenFramebuffers(1, C.byref(f))
        self.fid = f.value
        gl.glBindFramebuffer(G.GL_FRAMEBUFFER, self.fid)
        for n, t in enumerate(colours):
            gl.glFramebufferTexture2D(G.GL_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0 + n, G.GL_TEXTURE_2D, t.tid, 0)
        bufs = (C.c_uint * len(colours))(*[G.GL_COLOR_ATTACHMENT0 + n for n in range(len(colours))])
        gl.glDrawBuffers(len(colours), bufs)
        r = C.c_uint(0)
        gl.glGenRenderbuffers(1, C.byref(r))
        gl.glBindRenderbuffer(G.GL_RENDERBUFFER, r.value)
        gl.glRenderbufferStorage(G.GL_RENDERBUFFER, G.GL_DEPTH_COMPONENT24, w, h)
        gl.glFramebufferRenderbuffer(G.GL_FRAMEBUFFER, G.GL_DEPTH_ATTACHMENT, G.GL_RENDERBUFFER, r.value)
        st = gl.glCheckFramebufferStatus(G.GL_FRAMEBUFFER)
        assert st == G.GL_FRAMEBUFFER_COMPLETE, hex(st)
        gl.glBindFramebuffer(G.GL_FRAMEBUFFER, 0)

    def bind(self):
        self.gl.glBindFramebuffer(G.GL_FRAMEBUFFER, self.fid)

    def unbind(self):
        self.gl.glBindFramebuffer(G.GL_FRAMEBUFFER, 0)


XFB5 = ["vPosition0", "vColor0", "vNormRad0", "curv_map_max0", "curv_map_min0"]


class Program:
    """Shader (Core/src/Shaders/Shaders.h:28-72) built by loadProgram*FromFile (:74-116)."""

    def __init__(self, gl, vert, frag=None, geom=None, xfb=None):
        self.gl = gl
        self.name = "+".join(x for x in (vert, geom, frag) if x)
        self.pid = gl.glCreateProgram()
        for kind, fname in ((G.GL_VERTEX_SHADER, vert), (G.GL_GEOMETRY_SHADER, geom), (G.GL_FRAGMENT_SHADER, frag)):
            if not fname:
                continue
            sid = gl.glCreateShader(kind)
            src = shader_source(fname).encode("latin-1")
            buf = C.c_char_p(src)
            gl.glShaderSource(sid, 1, C.byref(buf), None)
            gl.glCompileShader(sid)
            ok = C.c_int(0)
            gl.glGetShaderiv(sid, G.GL_COMPILE_STATUS, C.byref(ok))
            if not ok.value:
                log = C.create_string_buffer(16384)
                gl.glGetShaderInfoLog(sid, 16384, None, log)
                raise RuntimeError("%s does not compile:\n%s" % (fname, log.value.decode()))
            gl.glAttachShader(self.pid, sid)
        if xfb:
            # the reference uses NV_transform_feedback after linking (GlobalModel.cpp:113-170); the core-profile
            # equivalent names the same five varyings before linking
            arr = (C.c_char_p * len(xfb))(*[s.encode() for s in xfb])
            gl.glTransformFeedbackVaryings(self.pid, len(xfb), arr, G.GL_INTERLEAVED_ATTRIBS)
        gl.glLinkProgram(self.pid)
        ok = C.c_int(0)
        gl.glGetProgramiv(self.pid, G.GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(16384)
            gl.glGetProgramInfoLog(self.pid, 16384, None, log)
            raise RuntimeError("%s does not link:\n%s" % (self.name, log.value.decode()))

    def bind(self):
        self.gl.glUseProgram(self.pid)

    def unbind(self):
        self.gl.glUseProgram(0)

    def set(self, name, v):
        """Shader::setUniform (Shaders.h:39-71): the GL call follows the C++ type of the Uniform's value."""
        gl = self.gl
        loc = gl.glGetUniformLocation(self.pid, name.encode())
        if loc < 0:
            return   # the reference issues glUniform* on location -1 too (silently ignored by GL)
        if isinstance(v, (bool, np.bool_)):
            gl.glUniform1i(loc, int(v))
        elif isinstance(v, (int, np.integer)):
            gl.glUniform1i(loc, int(v))
        elif isinstance(v, (float, np.floating)):
            gl.glUniform1f(loc, float(v))
        else:
            a = np.asarray(v, np.float32)
            if a.shape == (4, 4):
                m = np.ascontiguousarray(a.T)   # Eigen storage is column-major; transpose = GL_FALSE
                gl.glUniformMatrix4fv(loc, 1, 0, m.ctypes.data_as(C.c_void_p))
            elif a.shape == (4,):
                gl.glUniform4f(loc, *[float(x) for x in a])
            elif a.shape == (3,):
                gl.glUniform3f(loc, *[float(x) for x in a])
            elif a.shape == (2,):
                gl.glUniform2f(loc, *[float(x) for x in a])
            else:
                raise TypeError(name)


def f32(x):
    return float(np.float32(x))


class RefPipeline:
    """The GL objects HRBFFusion, GlobalModel, IndexMap and FillIn own, and the passes they run.

    Parameters are the GlobalStateParam fields / constructor arguments the reference reads (same names as hrbf_params).
    """

    def __init__(self, W, H, fx, fy, cx, cy, depth_scale, prm=None, tex_dim=1024, max_surfels=1 << 20):
        gl = self.gl = G.GL(compat=True)
        self.W, self.H = W, H
        self.fx, self.fy, self.cx, self.cy = f32(fx), f32(fy), f32(cx), f32(cy)
        self.depth_factor = f32(depth_scale)          # mDepthMapFactor after inversion (HRBFFusion.cpp:772-780)
        p = dict(depth_cutoff=3.5, max_depth_processed=20.0, confidence_threshold=5.0, use_bilateral=1,
                 init_radius_multiplier=4.0, curv_estimation_window=3.0, curv_valid_threshold=300.0,
                 normal_estimation_pca=1.0, use_conf_eval=0, conf_eval_epsilon=1000.0, icp_curv_weight_lambda=10.0,
                 predict_window_multiplier=3.0, predict_min_neighbors=6, predict_max_neighbors=10,
                 predict_conf_threshold=3.0, clean_window_multiplier=2.0)
        p.update(prm or {})
        self.p = p
        self.tex_dim, self.max_surfels = tex_dim, max_surfels
        # GUI/src/Tools/GUI.h:53-71
        gl.glPixelStorei(G.GL_UNPACK_ALIGNMENT, 1)
        gl.glPixelStorei(G.GL_PACK_ALIGNMENT, 1)
        gl.glEnable(G.GL_DEPTH_TEST)
        gl.glDepthMask(1)
        gl.glDepthFunc(G.GL_LESS)
        self._textures()
        self._compute_packs()
        self._global_model()
        self._index_map()
        self._fill_in()

    # ---- HRBFFusion::createTextures (HRBFFusion.cpp:783-895) -------------------------------------------------
    def _textures(self):
        gl, W, H = self.gl, self.W, self.H
        t = self.tex = {}
        t["RGB"] = tex_rgba8(gl, W, H, linear=True)       # draw = true
        t["DEPTH_RAW"] = tex_u16(gl, W, H)
        for k in ("DEPTH_FILTERED", "DEPTH_METRIC", "DEPTH_METRIC_FILTERED", "GRADIENT_MAG", "RADIUS", "CONFIDENCE"):
            t[k] = tex_f1(gl, W, H)
        for k in ("VERTEX_RAW", "VERTEX_FILTERED", "NORMAL", "NORMAL_OPT", "PRINCIPAL_CURV1", "PRINCIPAL_CURV2"):
            t[k] = tex_rgba32f(gl, W, H)

    # ---- HRBFFusion::createCompute (HRBFFusion.cpp:897-940) --------------------------------------------------
    def _compute_packs(self):
        gl, W, H, t = self.gl, self.W, self.H, self.tex
        cp = self.cp = {}

        def pack(frag, *targets):
            return (Program(gl, "empty.vert", frag, "quad.geom"), Fbo(gl, W, H, [t[k] for k in targets]))
        cp["FILTER_BILATERAL"] = pack("depth_bilateral.frag", "DEPTH_FILTERED")
        cp["FILTER_GAUSS"] = pack("depth_guass.frag", "DEPTH_FILTERED")
        cp["METRIC"] = pack("depth_metric_raw.frag", "DEPTH_METRIC")
        cp["METRIC_FILTERED"] = pack("depth_metric_filtered.frag", "DEPTH_METRIC_FILTERED")
        cp["VERTEX_NORMAL_RADIUS"] = pack("depth_vertex_normal_radius.frag", "VERTEX_RAW", "VERTEX_FILTERED", "NORMAL", "RADIUS")
        cp["CURVATURE"] = pack("depth_curvature_gradient.frag", "PRINCIPAL_CURV1", "PRINCIPAL_CURV2", "GRADIENT_MAG", "NORMAL_OPT")
        cp["UPDATE_NORMALRAD"] = pack("depth_update_normalrad.frag", "NORMAL")
        cp["CONFIDENCE_EVALUATION"] = pack("depth_confidence_evaluation.frag", "CONFIDENCE")

    def _quad_pass(self, prog, fbo, inputs, uniforms):
        """ComputePack::compute / compute_2input (Shaders/ComputePack.cpp:63-133) and the FillIn / predictHRBF passes
        of the same shape: bind fbo, viewport, clear colour+depth, uniforms, textures on units 0.., one GL_POINTS
        vertex that quad.geom expands to the full-screen strip."""
        gl = self.gl
        fbo.bind()
        gl.glViewport(0, 0, fbo.w, fbo.h)
        gl.glClearColor(0, 0, 0, 0)
        gl.glClear(G.GL_COLOR_BUFFER_BIT | G.GL_DEPTH_BUFFER_BIT)
        prog.bind()
        for k, v in uniforms:
            prog.set(k, v)
        for u, tx in enumerate(inputs):
            gl.glActiveTexture(G.GL_TEXTURE0 + u)
            gl.glBindTexture(G.GL_TEXTURE_2D, tx.tid)
        gl.glDrawArrays(G.GL_POINTS, 0, 1)
        for u in range(len(inputs)):
            gl.glActiveTexture(G.GL_TEXTURE0 + u)
            gl.glBindTexture(G.GL_TEXTURE_2D, 0)
        gl.glActiveTexture(G.GL_TEXTURE0)
        fbo.unbind()
        prog.unbind()
        gl.glFinish()

    def _cam_inv(self):
        # Eigen::Vector4f(cx, cy, 1.0 / fx, 1.0 / fy): double division, rounded to float by the Vector4f ctor
        return np.array([self.cx, self.cy, 1.0 / float(self.fx), 1.0 / float(self.fy)], np.float32)   # float(): numpy would divide in float32

    def upload_frame(self, rgb, depth):
        """HRBFFusion.cpp:1008-1010: Upload(depth, GL_LUMINANCE_INTEGER_EXT, GL_UNSIGNED_SHORT); Upload(rgb, GL_RGB, ..)."""
        self.tex["DEPTH_RAW"].upload(np.asarray(depth, np.uint16).reshape(self.H, self.W))
        self.tex["RGB"].upload(np.asarray(rgb, np.uint8).reshape(self.H, self.W, 3), G.GL_RGB, G.GL_UNSIGNED_BYTE)

    def filter_depth(self):
        """HRBFFusion::filterDepth (HRBFFusion.cpp:1272-1280)."""
        u = [("cols", float(self.W)), ("rows", float(self.H)), ("maxD", f32(self.p["depth_cutoff"])),
             ("depthFactor", self.depth_factor)]
        prog, fbo = self.cp["FILTER_BILATERAL" if self.p["use_bilateral"] else "FILTER_GAUSS"]
        self._quad_pass(prog, fbo, [self.tex["DEPTH_RAW"]], u)

    def metricise_depth(self):
        """HRBFFusion::metriciseDepth (HRBFFusion.cpp:1263-1270)."""
        u = [("maxD", f32(self.p["depth_cutoff"])), ("depthFactor", self.depth_factor)]
        self._quad_pass(*self.cp["METRIC"], [self.tex["DEPTH_RAW"]], u)
        self._quad_pass(*self.cp["METRIC_FILTERED"], [self.tex["DEPTH_FILTERED"]], u)

    def compute_vertex_normal_radius(self):
        """HRBFFusion::computeVertexNormalRadius (HRBFFusion.cpp:1329-1345)."""
        u = [("cols", float(self.W)), ("rows", float(self.H)), ("cam", self._cam_inv()),
             ("depthRawSampler", 0), ("depthFilteredSampler", 1),
             ("radius_multiplier", f32(self.p["init_radius_multiplier"])),
             ("PCAforNormalEstimation", f32(self.p["normal_estimation_pca"]))]
        self._quad_pass(*self.cp["VERTEX_NORMAL_RADIUS"], [self.tex["DEPTH_METRIC"], self.tex["DEPTH_METRIC_FILTERED"]], u)

    def compute_curvature_gradient(self):
        """HRBFFusion::computeCurvatureGradient (HRBFFusion.cpp:1282-1300)."""
        u = [("cols", float(self.W)), ("rows", float(self.H)), ("cam", self._cam_inv()),
             ("maxD", f32(self.p["depth_cutoff"])), ("winMultiply", f32(self.p["curv_estimation_window"])),
             ("VertexFiltered", 0), ("NormalRadSampler", 1)]
        self._quad_pass(*self.cp["CURVATURE"], [self.tex["VERTEX_FILTERED"], self.tex["NORMAL"]], u)

    def update_normal_rad(self):
        """HRBFFusion::updateNormalRad (HRBFFusion.cpp:1301-1310)."""
        u = [("cols", float(self.W)), ("rows", float(self.H)), ("NormalOptSampler", 0), ("VertexSampler", 1)]
        self._quad_pass(*self.cp["UPDATE_NORMALRAD"], [self.tex["NORMAL_OPT"], self.tex["VERTEX_FILTERED"]], u)

    def vertex_confidence(self, weighting):
        """HRBFFusion::VertexConfidence (HRBFFusion.cpp:1311-1327)."""
        u = [("cols", float(self.W)), ("rows", float(self.H)), ("cam", self._cam_inv()),
             ("useConfidenceEvaluation", float(self.p["use_conf_eval"])), ("epsilon", f32(self.p["conf_eval_epsilon"])),
             ("gradient_mag", 0), ("depthSampler", 1), ("weighting", f32(weighting))]
        self._quad_pass(*self.cp["CONFIDENCE_EVALUATION"], [self.tex["GRADIENT_MAG"], self.tex["DEPTH_METRIC"]], u)

    # ---- GlobalModel (GlobalModel.cpp:30-205) -----------------------------------------------------------------
    def _buffer(self, nbytes, data=None, usage=G.GL_STREAM_DRAW):
        gl = self.gl
        b = C.c_uint(0)
        gl.glGenBuffers(1, C.byref(b))
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, b.value)
        if data is None:
            data = np.zeros(nbytes, np.uint8)
        gl.glBufferData(G.GL_ARRAY_BUFFER, nbytes, data.ctypes.data_as(C.c_void_p), usage)
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, 0)
        return b.value

    def _xfb(self):
        t = C.c_uint(0)
        self.gl.glGenTransformFeedbacks(1, C.byref(t))
        return t.value

    def _global_model(self):
        gl, W, H, D = self.gl, self.W, self.H, self.tex_dim
        VS = 80   # Vertex::SIZE (Shaders/Vertex.cpp:21-44): 5 x vec4
        self.VS = VS
        self.target, self.render_source = 0, 1
        self.count = 0
        self.vbos = [(self._buffer(self.max_surfels * VS), self._xfb()) for _ in range(2)]
        self.new_unstable_vbo, self.new_unstable_fid = self._buffer(W * H * VS), self._xfb()
        # uv buffer: column-major walk over the image, texel centres (GlobalModel.cpp:82-98)
        ii, jj = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
        uv = np.empty((W * H, 2), np.float32)
        # `((float)i / (float)width) + 1.0 / (2 * (float)width)`: a float quotient, a DOUBLE reciprocal (1.0 is a double literal), a
        # double sum, stored as float.  (float(...) keeps numpy's weak-scalar rules from rounding the reciprocal to float32.)
        uv[:, 0] = (ii.ravel().astype(np.float32) / np.float32(W)).astype(np.float64) + 1.0 / float(2 * np.float32(W))
        uv[:, 1] = (jj.ravel().astype(np.float32) / np.float32(H)).astype(np.float64) + 1.0 / float(2 * np.float32(H))
        self.uv_size = W * H
        self.uvo = self._buffer(uv.nbytes, uv, G.GL_STATIC_DRAW)
        self.update_maps = [tex_rgba32f(gl, D, D) for _ in range(5)]
        self.gm_fbo = Fbo(gl, D, D, self.update_maps)
        self.delta_trans_dim = 16384   # DELTA_TRANS_DIMENSION = 19200 exceeds llvmpipe's GL_MAX_TEXTURE_SIZE (16384)
        self.delta_trans = tex_f1(gl, self.delta_trans_dim, 1)
        self.kf_dim = 16384            # ACTIVE_KEYFRAME_DIMENSION = 19200, same limit; both reach the shaders as uniforms
        self.gm_kfid = tex_f1(gl, self.kf_dim, 1)
        self.active_kf = [0]      # lActiveKFID after the first frame (HRBFFusion.cpp:1053-1055)
        self.init_prog = Program(gl, "init_unstableTex.vert", None, "init_unstableTex.geom", XFB5)
        self.data_prog = Program(gl, "data.vert", "data.frag", "data.geom", XFB5)
        self.update_prog = Program(gl, "update.vert", None, None, XFB5)
        self.unstable_prog = Program(gl, "copy_unstable.vert", None, "copy_unstable.geom", XFB5)
        self.delta_prog = Program(gl, "update_delta_trans.vert", None, "update_delta_trans.geom", XFB5)
        q = C.c_uint(0)
        gl.glGenQueries(1, C.byref(q))
        self.count_query = q.value
        # "Empty both transform feedbacks" (GlobalModel.cpp:172-187)
        self.init_prog.bind()
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        for vbo, fid in self.vbos:
            gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, fid)
            gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, vbo)
            gl.glBeginTransformFeedback(G.GL_POINTS)
            gl.glDrawArrays(G.GL_POINTS, 0, 0)
            gl.glEndTransformFeedback()
            gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        self.init_prog.unbind()

    def _bind_textures(self, texs):
        gl = self.gl
        for u, tx in enumerate(texs):
            gl.glActiveTexture(G.GL_TEXTURE0 + u)
            gl.glBindTexture(G.GL_TEXTURE_2D, tx.tid)

    def _unbind_textures(self, n):
        gl = self.gl
        for u in range(n):
            gl.glActiveTexture(G.GL_TEXTURE0 + u)
            gl.glBindTexture(G.GL_TEXTURE_2D, 0)
        gl.glActiveTexture(G.GL_TEXTURE0)

    def _attribs5(self, vbo, on=True):
        gl = self.gl
        if on:
            gl.glBindBuffer(G.GL_ARRAY_BUFFER, vbo)
            for a in range(5):
                gl.glEnableVertexAttribArray(a)
                gl.glVertexAttribPointer(a, 4, G.GL_FLOAT, 0, self.VS, C.c_void_p(16 * a))
        else:
            for a in range(5):
                gl.glDisableVertexAttribArray(a)
            gl.glBindBuffer(G.GL_ARRAY_BUFFER, 0)

    def _query_count(self):
        c = C.c_uint(0)
        self.gl.glGetQueryObjectuiv(self.count_query, G.GL_QUERY_RESULT, C.byref(c))
        return c.value

    def initialise(self, init_pose):
        """GlobalModel::initialise (GlobalModel.cpp:214-288)."""
        gl, pr, t = self.gl, self.init_prog, self.tex
        pr.bind()
        for k, v in (("vertexSampler", 0), ("normalSampler", 1), ("colorSampler", 2), ("curv1Sampler", 3),
                     ("curv2Sampler", 4), ("gradientMagSampler", 5), ("cols", float(self.W)), ("rows", float(self.H)),
                     ("cam", self._cam_inv()), ("curvature_valid_threshold", f32(self.p["curv_valid_threshold"])),
                     ("useConfidenceEvaluation", float(self.p["use_conf_eval"])),
                     ("epsilon", f32(self.p["conf_eval_epsilon"])), ("init_pose", np.asarray(init_pose, np.float32))):
            pr.set(k, v)
        gl.glEnableVertexAttribArray(0)
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, self.uvo)
        gl.glVertexAttribPointer(0, 2, G.GL_FLOAT, 0, 0, None)
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        vbo, fid = self.vbos[self.target]
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, fid)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, vbo)
        self._bind_textures([t["VERTEX_RAW"], t["NORMAL"], t["RGB"], t["PRINCIPAL_CURV1"], t["PRINCIPAL_CURV2"], t["GRADIENT_MAG"]])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glBeginQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, self.count_query)
        gl.glDrawArrays(G.GL_POINTS, 0, self.uv_size)
        self._unbind_textures(6)
        gl.glEndTransformFeedback()
        gl.glEndQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN)
        self.count = self._query_count()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        gl.glDisableVertexAttribArray(0)
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, 0)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        pr.unbind()
        gl.glFinish()

    def download_map(self):
        """GlobalModel::downloadMap (GlobalModel.cpp:775-804) — reads vbos[target] (the reference reads renderSource after
        an even number of swaps; `target` is the buffer model() hands to every consumer)."""
        gl = self.gl
        o = np.zeros((self.count, 20), np.float32)
        if self.count:
            gl.glBindBuffer(G.GL_ARRAY_BUFFER, self.vbos[self.target][0])
            gl.glGetBufferSubData(G.GL_ARRAY_BUFFER, 0, o.nbytes, o.ctypes.data_as(C.c_void_p))
            gl.glBindBuffer(G.GL_ARRAY_BUFFER, 0)
        return o

    def upload_map(self, m):
        """Harness-only: seed vbos[target] with `m` (n x 20) through a pass-through transform feedback, so that the
        feedback object's vertex count (what glDrawTransformFeedback draws) equals n."""
        gl = self.gl
        m = np.ascontiguousarray(m, np.float32)
        n = m.shape[0]
        assert n <= self.max_surfels
        src = self._buffer(max(m.nbytes, 80), m if n else None)
        if not hasattr(self, "_copy_prog"):
            self._copy_prog = _passthrough_program(gl)
        vbo, fid = self.vbos[self.target]
        gl.glUseProgram(self._copy_prog)
        self._attribs5(src)
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, fid)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, vbo)
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.glEndTransformFeedback()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        self._attribs5(0, on=False)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        gl.glUseProgram(0)
        b = C.c_uint(src)
        gl.glDeleteBuffers(1, C.byref(b))
        gl.glFinish()
        self.count = n

    def _upload_kfid(self, tex):
        lafk = np.zeros(self.kf_dim, np.float32)
        lafk[self.active_kf] = 1
        tex.upload(lafk.reshape(1, -1))

    # ---- IndexMap (IndexMap.cpp:27-267) ------------------------------------------------------------------------
    def _index_map(self):
        gl, W, H = self.gl, self.W, self.H
        self.index_prog = Program(gl, "index_map.vert", "index_map.frag")
        self.im_index = tex_u32(gl, W, H)
        self.im_vertconf, self.im_colortime, self.im_normrad, self.im_curvmax, self.im_curvmin = \
            [tex_rgba32f(gl, W, H) for _ in range(5)]
        self.index_fbo = Fbo(gl, W, H, [self.im_index, self.im_vertconf, self.im_colortime, self.im_normrad,
                                        self.im_curvmax, self.im_curvmin])
        self.im_kfid = tex_f1(gl, self.kf_dim, 1)
        self.im_depth = tex_f1(gl, W, H)     # depthTexture: only written by renderDepth, which processFrame never calls
        self.predict_prog = Program(gl, "empty.vert", "predict_hrbf.frag", "quad.geom")
        self.pr_image = tex_rgba8(gl, W, H)
        self.pr_vertex, self.pr_normal, self.pr_curv1, self.pr_curv2 = [tex_rgba32f(gl, W, H) for _ in range(4)]
        self.pr_time = tex_u16(gl, W, H)
        self.pr_icpw = tex_f1(gl, W, H)
        self.predict_fbo = Fbo(gl, W, H, [self.pr_image, self.pr_vertex, self.pr_normal, self.pr_curv1, self.pr_curv2,
                                          self.pr_time, self.pr_icpw])

    def predict_indices(self, pose, time, depth_cutoff=None, insert_submap=0, index_submap=0):
        """IndexMap::predictIndices (IndexMap.cpp:193-267)."""
        gl, pr = self.gl, self.index_prog
        depth_cutoff = self.p["max_depth_processed"] if depth_cutoff is None else depth_cutoff
        self.index_fbo.bind()
        gl.glViewport(0, 0, self.W, self.H)
        gl.glClearColor(0, 0, 0, 0)
        gl.glClear(G.GL_COLOR_BUFFER_BIT | G.GL_DEPTH_BUFFER_BIT)
        pr.bind()
        t_inv = np.linalg.inv(np.asarray(pose, np.float32)).astype(np.float32)
        for k, v in (("t_inv", t_inv), ("cam", np.array([self.cx, self.cy, self.fx, self.fy], np.float32)),
                     ("maxDepth", f32(depth_cutoff)), ("cols", float(self.W)), ("rows", float(self.H)),
                     ("time", int(time)), ("insertSubmap", int(insert_submap)), ("indexSubmap", int(index_submap)),
                     ("curvature_valid_threshold", f32(self.p["curv_valid_threshold"]))):
            pr.set(k, v)
        self._upload_kfid(self.im_kfid)
        pr.set("KeyFrameIDMap", 0)
        pr.set("KeyFrameIDDimen", float(self.kf_dim))
        gl.glActiveTexture(G.GL_TEXTURE0)
        gl.glBindTexture(G.GL_TEXTURE_2D, self.im_kfid.tid)
        vbo, fid = self.vbos[self.target]
        self._attribs5(vbo)
        gl.glDrawTransformFeedback(G.GL_POINTS, fid)
        self._attribs5(0, on=False)
        gl.glBindTexture(G.GL_TEXTURE_2D, 0)
        gl.glActiveTexture(G.GL_TEXTURE0)
        self.index_fbo.unbind()
        pr.unbind()
        gl.glFinish()

    def index_images(self):
        return dict(INDEX=get_u32(self.im_index), INDEX_VERTCONF=get_f4(self.im_vertconf),
                    INDEX_COLORTIME=get_f4(self.im_colortime), INDEX_NORMRAD=get_f4(self.im_normrad),
                    INDEX_CURVMAX=get_f4(self.im_curvmax), INDEX_CURVMIN=get_f4(self.im_curvmin))

    def set_index_images(self, d):
        self.im_index.upload(np.asarray(d["INDEX"], np.uint32))
        for k, t in (("INDEX_VERTCONF", self.im_vertconf), ("INDEX_COLORTIME", self.im_colortime),
                     ("INDEX_NORMRAD", self.im_normrad), ("INDEX_CURVMAX", self.im_curvmax), ("INDEX_CURVMIN", self.im_curvmin)):
            t.upload(np.asarray(d[k], np.float32))

    def predict_hrbf(self):
        """IndexMap::predictHRBF(ACTIVE) (IndexMap.cpp:413-518)."""
        cam = np.array([self.cx, self.cy, 1.0 / float(self.fx), 1.0 / float(self.fy)], np.float32)   # double division (IndexMap.cpp:451-452)
        u = [("cam", cam), ("cols", float(self.W)), ("rows", float(self.H)), ("scale", 1.0),
             ("predict_minimum_neighbors", int(self.p["predict_min_neighbors"])),
             ("predict_maximum_neighbors", int(self.p["predict_max_neighbors"])),
             ("winMultiply", f32(self.p["predict_window_multiplier"])),
             ("predict_confidence_threshold", f32(self.p["predict_conf_threshold"])),
             ("indexSampler", 0), ("vertConfSampler", 1), ("colorTimeSampler", 2), ("normRadSampler", 3),
             ("curv_maxSampler", 4), ("curv_minSampler", 5),
             ("icp_weight_lambda", f32(self.p["icp_curv_weight_lambda"]))]
        self._quad_pass(self.predict_prog, self.predict_fbo,
                        [self.im_index, self.im_vertconf, self.im_colortime, self.im_normrad, self.im_curvmax, self.im_curvmin], u)

    def dense_thumbnail(self):
        """Resize::vertex(indexMap.vertexTexHRBF(), verticesBuff) (Shaders/Resize.cpp:106-134, resize.frag): the predicted vertex
        map sampled (NEAREST: vertexTextureHRBF is built with draw = false, IndexMap.cpp:85-87) at the centres of a
        (W / consSample) x (H / consSample) grid, consSample = 20 (HRBFFusion.cpp:27-31) — what denseEnough counts."""
        if not hasattr(self, "_resize"):
            w, h = self.W // 20, self.H // 20
            self._resize_tex = tex_rgba32f(self.gl, w, h)
            self._resize = (Program(self.gl, "empty.vert", "resize.frag", "quad.geom"), Fbo(self.gl, w, h, [self._resize_tex]))
        self._quad_pass(*self._resize, [self.pr_vertex], [("eSampler", 0)])
        return get_f4(self._resize_tex)

    def prediction_images(self):
        return dict(PRED_IMAGE=get_rgba8(self.pr_image), PRED_VERTEX=get_f4(self.pr_vertex), PRED_NORMAL=get_f4(self.pr_normal),
                    PRED_CURV1=get_f4(self.pr_curv1), PRED_CURV2=get_f4(self.pr_curv2),
                    PRED_TIME=self.pr_time.download(G.GL_RED_INTEGER, G.GL_UNSIGNED_INT, np.uint32, 1),
                    PRED_ICPWEIGHT=get_f1(self.pr_icpw))

    # ---- GlobalModel::fuse / clean (GlobalModel.cpp:355-688) ---------------------------------------------------
    def fuse(self, pose, time, weighting, depth_cutoff=None, insert_submap=False, index_submap=0.0):
        gl, t, D = self.gl, self.tex, self.tex_dim
        depth_cutoff = self.p["max_depth_processed"] if depth_cutoff is None else depth_cutoff
        pose = np.asarray(pose, np.float32)
        # stage 1: data association (GlobalModel.cpp:375-468)
        self.gm_fbo.bind()
        gl.glViewport(0, 0, D, D)
        gl.glClearColor(0, 0, 0, 0)
        gl.glClear(G.GL_COLOR_BUFFER_BIT | G.GL_DEPTH_BUFFER_BIT)
        pr = self.data_prog
        pr.bind()
        for k, v in (("cSampler", 0), ("drSampler", 1), ("drfSampler", 2), ("new_curv1_Samp", 3), ("new_curv2_Samp", 4),
                     ("new_conf_Samp", 5), ("indexSampler", 6), ("vertConfSampler", 7), ("colorTimeSampler", 8),
                     ("normRadSampler", 9), ("time", float(time)), ("weighting", f32(weighting)), ("cam", self._cam_inv()),
                     ("cols", float(self.W)), ("rows", float(self.H)), ("scale", 1.0), ("texDim", float(D)),
                     ("pose", pose), ("maxDepth", f32(depth_cutoff)), ("indexSubmap", float(index_submap)),
                     ("insertSubmap", bool(insert_submap)),
                     ("RadiusMultiplier", f32(self.p["init_radius_multiplier"])),
                     ("PCAforNormalEstimation", f32(self.p["normal_estimation_pca"]))):
            pr.set(k, v)
        gl.glEnableVertexAttribArray(0)
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, self.uvo)
        gl.glVertexAttribPointer(0, 2, G.GL_FLOAT, 0, 0, None)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, self.new_unstable_fid)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, self.new_unstable_vbo)
        self._bind_textures([t["RGB"], t["DEPTH_METRIC"], t["DEPTH_METRIC_FILTERED"], t["PRINCIPAL_CURV1"], t["PRINCIPAL_CURV2"],
                             t["CONFIDENCE"], self.im_index, self.im_vertconf, self.im_colortime, self.im_normrad])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glBeginQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, self.count_query)   # harness: count the records
        gl.glDrawArrays(G.GL_POINTS, 0, self.uv_size)
        gl.glEndQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN)
        gl.glEndTransformFeedback()
        self.n_records = self._query_count()
        self.gm_fbo.unbind()
        self._unbind_textures(10)
        gl.glDisableVertexAttribArray(0)
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, 0)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        pr.unbind()
        gl.glFinish()
        # stage 2: merge into the map (GlobalModel.cpp:472-548)
        pr = self.update_prog
        pr.bind()
        for k, v in (("vertSamp", 0), ("colorSamp", 1), ("normSamp", 2), ("curv_map_maxSamp", 3), ("curv_map_minSamp", 4),
                     ("texDim", float(D)), ("time", int(time)), ("cam", self._cam_inv()), ("pose", pose),
                     ("pose_inv", np.linalg.inv(pose).astype(np.float32)), ("cols", float(self.W)), ("rows", float(self.H))):
            pr.set(k, v)
        self._attribs5(self.vbos[self.target][0])
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, self.vbos[self.render_source][1])
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, self.vbos[self.render_source][0])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        self._bind_textures(self.update_maps)
        gl.glDrawTransformFeedback(G.GL_POINTS, self.vbos[self.target][1])
        gl.glEndTransformFeedback()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        self._unbind_textures(5)
        self._attribs5(0, on=False)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        pr.unbind()
        self.target, self.render_source = self.render_source, self.target
        gl.glFinish()

    def fuse_records(self):
        """Harness-only: the records stage 1 wrote to newUnstableVbo (merge marks, vColor.w = -1, and new surfels, -2)."""
        gl = self.gl
        o = np.zeros((self.n_records, 20), np.float32)
        if self.n_records:
            gl.glBindBuffer(G.GL_ARRAY_BUFFER, self.new_unstable_vbo)
            gl.glGetBufferSubData(G.GL_ARRAY_BUFFER, 0, o.nbytes, o.ctypes.data_as(C.c_void_p))
            gl.glBindBuffer(G.GL_ARRAY_BUFFER, 0)
        return o

    def clean(self, pose, time, conf_threshold=None, max_depth=None):
        gl = self.gl
        pose = np.asarray(pose, np.float32)
        conf_threshold = self.p["confidence_threshold"] if conf_threshold is None else conf_threshold
        max_depth = self.p["max_depth_processed"] if max_depth is None else max_depth
        pr = self.unstable_prog
        pr.bind()
        for k, v in (("time", int(time)), ("confThreshold", f32(conf_threshold)), ("scale", 1.0), ("indexSampler", 0),
                     ("vertConfSampler", 1), ("colorTimeSampler", 2), ("normRadSampler", 3), ("depthSampler", 4),
                     ("maxDepth", f32(max_depth)), ("window_multiplier", f32(self.p["clean_window_multiplier"])),
                     ("curvature_valid_threshold", f32(self.p["curv_valid_threshold"])),
                     ("t_inv", np.linalg.inv(pose).astype(np.float32)), ("pose", pose),
                     ("cam", np.array([self.cx, self.cy, self.fx, self.fy], np.float32)),
                     ("cols", float(self.W)), ("rows", float(self.H))):
            pr.set(k, v)
        self._upload_kfid(self.gm_kfid)
        pr.set("KeyFrameIDMap", 5)
        pr.set("KeyFrameIDDimen", float(self.kf_dim))
        self._attribs5(self.vbos[self.target][0])
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, self.vbos[self.render_source][1])
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, self.vbos[self.render_source][0])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        self._bind_textures([self.im_index, self.im_vertconf, self.im_colortime, self.im_normrad, self.im_depth, self.gm_kfid])
        gl.glBeginQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, self.count_query)
        gl.glDrawTransformFeedback(G.GL_POINTS, self.vbos[self.target][1])
        self._attribs5(self.new_unstable_vbo)
        gl.glDrawTransformFeedback(G.GL_POINTS, self.new_unstable_fid)
        gl.glEndQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN)
        self.count = self._query_count()
        gl.glEndTransformFeedback()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        self._unbind_textures(6)
        self._attribs5(0, on=False)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        pr.unbind()
        self.target, self.render_source = self.render_source, self.target
        gl.glFinish()

    def update_model(self, deltas):
        """GlobalModel::updateModel (GlobalModel.cpp:690-767); deltas: list of 4x4 (DeltaTransformKF)."""
        gl = self.gl
        dt = np.zeros(self.delta_trans_dim, np.float32)
        flat = np.concatenate([np.asarray(m, np.float32).T.ravel() for m in deltas])   # m(k, j), j outer: column-major
        dt[:flat.size] = flat
        # the reference uploads only DTFK.size() texels (glTexSubImage2D width = DTFK.size()); the rest keep their content
        self.delta_trans.upload(dt.reshape(1, -1))
        pr = self.delta_prog
        pr.bind()
        pr.set("DeltaTransformKF", 0)
        pr.set("DeltaTransDimen", float(self.delta_trans_dim))
        self._attribs5(self.vbos[self.target][0])
        self._bind_textures([self.delta_trans])
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, self.vbos[self.render_source][1])
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, self.vbos[self.render_source][0])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glBeginQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, self.count_query)
        gl.glDrawTransformFeedback(G.GL_POINTS, self.vbos[self.target][1])
        gl.glEndQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN)
        self.count = self._query_count()
        gl.glEndTransformFeedback()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        self._attribs5(0, on=False)
        self._unbind_textures(1)
        gl.glBindTransformFeedback(G.GL_TRANSFORM_FEEDBACK, 0)
        pr.unbind()
        self.target, self.render_source = self.render_source, self.target
        gl.glFinish()

    # ---- FillIn (Shaders/FillIn.cpp:21-297) ---------------------------------------------------------------------
    def _fill_in(self):
        gl, W, H = self.gl, self.W, self.H
        self.fi_image = tex_rgba8(gl, W, H)
        self.fi_vertex, self.fi_normal, self.fi_curv1, self.fi_curv2 = [tex_rgba32f(gl, W, H) for _ in range(4)]
        self.fi_icpw = tex_f1(gl, W, H)
        self.fi_image_pack = (Program(gl, "empty.vert", "fill_rgb.frag", "quad.geom"), Fbo(gl, W, H, [self.fi_image]))
        self.fi_vertex_pack = (Program(gl, "empty.vert", "fill_vertex.frag", "quad.geom"), Fbo(gl, W, H, [self.fi_vertex, self.fi_icpw]))
        self.fi_normal_pack = (Program(gl, "empty.vert", "fill_normal.frag", "quad.geom"), Fbo(gl, W, H, [self.fi_normal]))
        self.fi_curv_pack = (Program(gl, "empty.vert", "fill_curvature.frag", "quad.geom"), Fbo(gl, W, H, [self.fi_curv1, self.fi_curv2]))

    def fill_in(self, time, weight, lost=False, frame_to_frame_rgb=False):
        """HRBFFusion::predict's four FillIn calls (HRBFFusion.cpp:1252-1259)."""
        t = self.tex
        cam = np.array([self.cx, self.cy, np.float32(1.0) / np.float32(self.fx), np.float32(1.0) / np.float32(self.fy)], np.float32)
        # FillIn::vertex (FillIn.cpp:127-195)
        u = [("eSampler", 0), ("filteredSampler", 1), ("rck1Sampler", 2), ("rck2Sampler", 3), ("eicpweightSampler", 4),
             ("confidenceSampler", 5), ("passthrough", int(lost)), ("curvature_valid_threshold", f32(self.p["curv_valid_threshold"])),
             ("cam", cam), ("cols", float(self.W)), ("rows", float(self.H)), ("weight", f32(weight)), ("time", int(time)),
             ("icp_weight_lambda", f32(self.p["icp_curv_weight_lambda"]))]
        self._quad_pass(*self.fi_vertex_pack, [self.pr_vertex, t["VERTEX_FILTERED"], t["PRINCIPAL_CURV1"], t["PRINCIPAL_CURV2"],
                                               self.pr_icpw, t["CONFIDENCE"]], u)
        # FillIn::normal (FillIn.cpp:197-243)
        u = [("eSampler", 0), ("rSampler", 1), ("passthrough", int(lost)), ("cam", cam), ("cols", float(self.W)), ("rows", float(self.H))]
        self._quad_pass(*self.fi_normal_pack, [self.pr_normal, t["NORMAL"]], u)
        # FillIn::curvature (FillIn.cpp:245-297)
        u = [("ecurvk1Sampler", 0), ("ecurvk2Sampler", 1), ("rcurvk1Sampler", 2), ("rcurvk2Sampler", 3), ("passthrough", int(lost)),
             ("cam", cam), ("cols", float(self.W)), ("rows", float(self.H))]
        self._quad_pass(*self.fi_curv_pack, [self.pr_curv1, self.pr_curv2, t["PRINCIPAL_CURV1"], t["PRINCIPAL_CURV2"]], u)
        # FillIn::image (FillIn.cpp:93-125)
        u = [("eSampler", 0), ("rSampler", 1), ("passthrough", int(lost or frame_to_frame_rgb))]
        self._quad_pass(*self.fi_image_pack, [self.pr_image, t["RGB"]], u)

    def fill_images(self):
        return dict(FILL_IMAGE=get_rgba8(self.fi_image), FILL_VERTEX=get_f4(self.fi_vertex), FILL_NORMAL=get_f4(self.fi_normal),
                    FILL_CURV1=get_f4(self.fi_curv1), FILL_CURV2=get_f4(self.fi_curv2), FILL_ICPWEIGHT=get_f1(self.fi_icpw))

    # ---- image access by the library's image names ---------------------------------------------------------------
    _F1 = {"DEPTH_FILTERED", "DEPTH_METRIC", "DEPTH_METRIC_FILTERED", "GRADIENT_MAG", "RADIUS", "CONFIDENCE"}

    def get(self, name):
        t = self.tex[name]
        return get_f1(t) if name in self._F1 else get_f4(t)

    def put(self, name, a):
        self.tex[name].upload(np.asarray(a, np.float32))


def _passthrough_program(gl):
    src = b"""#version 330 core
layout (location = 0) in vec4 a0; layout (location = 1) in vec4 a1; layout (location = 2) in vec4 a2;
layout (location = 3) in vec4 a3; layout (location = 4) in vec4 a4;
out vec4 o0; out vec4 o1; out vec4 o2; out vec4 o3; out vec4 o4;
void main() { o0 = a0; o1 = a1; o2 = a2; o3 = a3; o4 = a4; }
"""
    pid = gl.glCreateProgram()
    sid = gl.glCreateShader(G.GL_VERTEX_SHADER)
    buf = C.c_char_p(src)
    gl.glShaderSource(sid, 1, C.byref(buf), None)
    gl.glCompileShader(sid)
    gl.glAttachShader(pid, sid)
    names = [b"o0", b"o1", b"o2", b"o3", b"o4"]
    arr = (C.c_char_p * 5)(*names)
    gl.glTransformFeedbackVaryings(pid, 5, arr, G.GL_INTERLEAVED_ATTRIBS)
    gl.glLinkProgram(pid)
    ok = C.c_int(0)
    gl.glGetProgramiv(pid, G.GL_LINK_STATUS, C.byref(ok))
    assert ok.value
    return pid
