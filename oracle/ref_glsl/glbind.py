"""TEST INFRASTRUCTURE.  Minimal ctypes OpenGL binding over oracle/_ref/libgl_headless.so (an X-less llvmpipe context).

Used only by refgl.py / tests/golden/make_ref_glsl.py, in the build container, to execute the reference's GLSL.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "..", "_ref", "libgl_headless.so")

# ---- enums -------------------------------------------------------------------------------------------------
GL_NO_ERROR = 0
GL_POINTS = 0x0000
GL_DEPTH_BUFFER_BIT = 0x0100
GL_COLOR_BUFFER_BIT = 0x4000
GL_LESS = 0x0201
GL_DEPTH_TEST = 0x0B71
GL_UNPACK_ALIGNMENT = 0x0CF5
GL_PACK_ALIGNMENT = 0x0D05
GL_TEXTURE_2D = 0x0DE1
GL_UNSIGNED_BYTE = 0x1401
GL_UNSIGNED_SHORT = 0x1403
GL_INT = 0x1404
GL_UNSIGNED_INT = 0x1405
GL_FLOAT = 0x1406
GL_RED = 0x1903
GL_RGB = 0x1907
GL_RGBA = 0x1908
GL_LUMINANCE = 0x1909
GL_NEAREST = 0x2600
GL_LINEAR = 0x2601
GL_TEXTURE_MAG_FILTER = 0x2800
GL_TEXTURE_MIN_FILTER = 0x2801
GL_TEXTURE_WRAP_S = 0x2802
GL_TEXTURE_WRAP_T = 0x2803
GL_CLAMP_TO_EDGE = 0x812F
GL_TEXTURE0 = 0x84C0
GL_VERTEX_PROGRAM_POINT_SIZE = 0x8642
GL_POINT_SPRITE = 0x8861
GL_DEPTH_COMPONENT24 = 0x81A6
GL_RGBA32F = 0x8814
GL_LUMINANCE32F_ARB = 0x8818
GL_R32F = 0x822E
GL_R16UI = 0x8234
GL_R32UI = 0x8236
GL_R32I = 0x8235
GL_RED_INTEGER = 0x8D94
GL_LUMINANCE32UI_EXT = 0x8D74
GL_LUMINANCE16UI_EXT = 0x8D7A
GL_LUMINANCE_INTEGER_EXT = 0x8D9C
GL_ARRAY_BUFFER = 0x8892
GL_STREAM_DRAW = 0x88E0
GL_STATIC_DRAW = 0x88E4
GL_FRAGMENT_SHADER = 0x8B30
GL_VERTEX_SHADER = 0x8B31
GL_GEOMETRY_SHADER = 0x8DD9
GL_COMPILE_STATUS = 0x8B81
GL_LINK_STATUS = 0x8B82
GL_INFO_LOG_LENGTH = 0x8B84
GL_QUERY_RESULT = 0x8866
GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN = 0x8C88
GL_INTERLEAVED_ATTRIBS = 0x8C8C
GL_TRANSFORM_FEEDBACK_BUFFER = 0x8C8E
GL_RASTERIZER_DISCARD = 0x8C89
GL_TRANSFORM_FEEDBACK = 0x8E22
GL_FRAMEBUFFER = 0x8D40
GL_RENDERBUFFER = 0x8D41
GL_COLOR_ATTACHMENT0 = 0x8CE0
GL_DEPTH_ATTACHMENT = 0x8D00
GL_FRAMEBUFFER_COMPLETE = 0x8CD5
GL_VERSION = 0x1F02
GL_RENDERER = 0x1F01

_v = None
_i, _u, _f, _p, _sz = C.c_int, C.c_uint, C.c_float, C.c_void_p, C.c_ssize_t
_SIGS = {
    "glGetString": (C.c_char_p, [_u]),
    "glGetError": (_u, []),
    "glEnable": (_v, [_u]), "glDisable": (_v, [_u]),
    "glViewport": (_v, [_i, _i, _i, _i]),
    "glClearColor": (_v, [_f, _f, _f, _f]), "glClear": (_v, [_u]), "glFinish": (_v, []),
    "glPixelStorei": (_v, [_u, _i]), "glDepthFunc": (_v, [_u]), "glDepthMask": (_v, [C.c_ubyte]),
    "glGenTextures": (_v, [_i, _p]), "glBindTexture": (_v, [_u, _u]), "glDeleteTextures": (_v, [_i, _p]),
    "glTexImage2D": (_v, [_u, _i, _i, _i, _i, _i, _u, _u, _p]),
    "glTexSubImage2D": (_v, [_u, _i, _i, _i, _i, _i, _u, _u, _p]),
    "glTexParameteri": (_v, [_u, _u, _i]), "glActiveTexture": (_v, [_u]),
    "glGetTexImage": (_v, [_u, _i, _u, _u, _p]),
    "glGenFramebuffers": (_v, [_i, _p]), "glBindFramebuffer": (_v, [_u, _u]),
    "glFramebufferTexture2D": (_v, [_u, _u, _u, _u, _i]),
    "glFramebufferRenderbuffer": (_v, [_u, _u, _u, _u]),
    "glCheckFramebufferStatus": (_u, [_u]), "glDrawBuffers": (_v, [_i, _p]),
    "glGenRenderbuffers": (_v, [_i, _p]), "glBindRenderbuffer": (_v, [_u, _u]),
    "glRenderbufferStorage": (_v, [_u, _u, _i, _i]),
    "glCreateShader": (_u, [_u]), "glShaderSource": (_v, [_u, _i, _p, _p]), "glCompileShader": (_v, [_u]),
    "glGetShaderiv": (_v, [_u, _u, _p]), "glGetShaderInfoLog": (_v, [_u, _i, _p, _p]),
    "glCreateProgram": (_u, []), "glAttachShader": (_v, [_u, _u]), "glLinkProgram": (_v, [_u]),
    "glGetProgramiv": (_v, [_u, _u, _p]), "glGetProgramInfoLog": (_v, [_u, _i, _p, _p]),
    "glUseProgram": (_v, [_u]), "glGetUniformLocation": (_i, [_u, C.c_char_p]),
    "glUniform1i": (_v, [_i, _i]), "glUniform1f": (_v, [_i, _f]), "glUniform2f": (_v, [_i, _f, _f]),
    "glUniform3f": (_v, [_i, _f, _f, _f]), "glUniform4f": (_v, [_i, _f, _f, _f, _f]),
    "glUniformMatrix4fv": (_v, [_i, _i, C.c_ubyte, _p]),
    "glTransformFeedbackVaryings": (_v, [_u, _i, _p, _u]),
    "glGenBuffers": (_v, [_i, _p]), "glBindBuffer": (_v, [_u, _u]), "glDeleteBuffers": (_v, [_i, _p]),
    "glBufferData": (_v, [_u, _sz, _p, _u]), "glBufferSubData": (_v, [_u, _sz, _sz, _p]),
    "glGetBufferSubData": (_v, [_u, _sz, _sz, _p]), "glBindBufferBase": (_v, [_u, _u, _u]),
    "glEnableVertexAttribArray": (_v, [_u]), "glDisableVertexAttribArray": (_v, [_u]),
    "glVertexAttribPointer": (_v, [_u, _i, _u, C.c_ubyte, _i, _p]),
    "glGenTransformFeedbacks": (_v, [_i, _p]), "glBindTransformFeedback": (_v, [_u, _u]),
    "glBeginTransformFeedback": (_v, [_u]), "glEndTransformFeedback": (_v, []),
    "glDrawTransformFeedback": (_v, [_u, _u]), "glDrawArrays": (_v, [_u, _i, _i]),
    "glGenQueries": (_v, [_i, _p]), "glBeginQuery": (_v, [_u, _u]), "glEndQuery": (_v, [_u]),
    "glGetQueryObjectuiv": (_v, [_u, _u, _p]),
}


class GL:
    """gl = GL(); gl.glViewport(0, 0, w, h) ... every call is followed by a glGetError check."""

    def __init__(self, compat=True):
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        self._lib = C.CDLL(_SO)
        self._lib.glh_init.argtypes = [C.c_char_p, C.c_int]
        rc = self._lib.glh_init(None, 1 if compat else 0)
        if rc:
            raise RuntimeError("no llvmpipe context (glh_init = %d)" % rc)
        self._lib.glh_proc.restype = C.c_void_p
        self._lib.glh_proc.argtypes = [C.c_char_p]
        self._fn = {}
        self._err = self._raw("glGetError")

    def _raw(self, name):
        res, args = _SIGS[name]
        addr = self._lib.glh_proc(name.encode())
        if not addr:
            raise RuntimeError("GL entry point missing: " + name)
        return C.CFUNCTYPE(res, *args)(addr)

    def __getattr__(self, name):
        if not name.startswith("gl"):
            raise AttributeError(name)
        f = self._fn.get(name)
        if f is None:
            raw = self._raw(name)

            def f(*a, _raw=raw, _name=name):
                r = _raw(*a)
                e = self._err()
                if e:
                    raise RuntimeError("%s -> GL error 0x%04x" % (_name, e))
                return r
            self._fn[name] = f
        return f
