"""Lossless re-coding of the executed-shader fixtures tests/golden/ref_glsl/{sphere,sphere_variants,pair,qqvga_pre,qqvga_map}:
`<scene>.npz` (deflate over raw floats, 36.7 MB together) -> `<scene>.fxz` (20 MB), because every driver run pushes the tree to a
GPU box (round-4 / round-5 brief, hygiene).  Two generic steps, no knowledge of what the arrays mean:

    byte planes + LZMA : a float32 image compresses better when its four byte planes are stored one after the other
    sibling XOR        : many arrays repeat another one almost bit for bit (the stable flow's records and the young flow's, a
                         parameter variant and its base scene, an image before and after a pass that touches few pixels): the array
                         is stored as bits(array) XOR bits(sibling) for the EARLIER array of the same shape and dtype that makes the
                         result smallest (any file earlier in ORDER), or plain when no sibling helps

A CRC32 of every decoded array is kept: `load` returns the bits the reference's shaders wrote or raises — the coding decides the
file's size, never its content.  tests/golden/ref_glsl/vga.npz has its own coder with predictors (tests/ref_glsl_vga.py).

    python tests/fixture_codec.py --encode      # build container: <scene>.npz -> <scene>.fxz (tests/golden/make_ref_glsl.py writes the .npz)
    python tests/fixture_codec.py --verify      # decode every .fxz and compare with the .npz beside it, where one exists
"""
import json
import lzma
import os
import sys
import zlib

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glsl")
ORDER = ["sphere", "sphere_variants", "pair", "qqvga_pre", "qqvga_map"]      # a file may refer to arrays of the files before it
_cache = {}


def _planes(a):
    a = np.ascontiguousarray(a)
    return np.ascontiguousarray(a.view(np.uint8).reshape(-1, a.dtype.itemsize).T).tobytes()


def _unplanes(raw, dtype, shape):
    dt = np.dtype(dtype)
    b = np.frombuffer(raw, np.uint8)
    return np.ascontiguousarray(b.reshape(dt.itemsize, -1).T).view(dt).reshape(shape)


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def _pack(a):
    return lzma.compress(_planes(a), preset=6)


def encode_all(verbose=True):
    pool = []         # (file, key, array) encoded so far
    for scene in ORDER:
        src = dict(np.load(os.path.join(GOLD, scene + ".npz")))
        meta, blobs = [], {}
        for k, a0 in src.items():
            shape0 = list(np.asarray(a0).shape)          # (np.ascontiguousarray makes a 0-d array 1-d)
            a = np.ascontiguousarray(a0)
            best, ref = _pack(a), None
            if a.size >= 64:
                for f2, k2, b in pool:
                    if b.shape == a.shape and b.dtype == a.dtype:
                        c = _pack(_bits(a) ^ _bits(b))
                        if len(c) < len(best):
                            best, ref = c, [f2, k2]
            meta.append({"name": k, "dtype": a.dtype.str, "shape": shape0, "crc": zlib.crc32(a.tobytes()), "ref": ref})
            blobs["b%d" % (len(meta) - 1)] = np.frombuffer(best, np.uint8)
            pool.append((scene, k, a))
        blobs["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
        out = os.path.join(GOLD, scene + ".fxz")
        with open(out, "wb") as f:
            np.savez(f, **blobs)
        if verbose:
            print("%-16s %6.2f MB -> %6.2f MB  (%d arrays, %d stored against a sibling)" % (
                scene, os.path.getsize(os.path.join(GOLD, scene + ".npz")) / 1e6, os.path.getsize(out) / 1e6, len(meta), sum(1 for m in meta if m["ref"])))


def load(scene):
    """dict name -> array of a scene; decodes the scenes it refers to on the way (cached)"""
    if scene in _cache:
        return dict(_cache[scene])
    path = os.path.join(GOLD, scene + ".fxz")
    if not os.path.exists(path):          # not one of the re-coded scenes (thumbnail) or a tree that still has the plain file
        d = dict(np.load(os.path.join(GOLD, scene + ".npz")))
        _cache[scene] = d
        return dict(d)
    z = np.load(path)
    meta = json.loads(z["meta"].tobytes().decode())
    out = {}
    for i, m in enumerate(meta):
        a = _unplanes(lzma.decompress(z["b%d" % i].tobytes()), m["dtype"], m["shape"])
        if m["ref"]:
            f2, k2 = m["ref"]
            sib = out[k2] if f2 == scene else load(f2)[k2]
            a = (_bits(a).reshape(-1) ^ _bits(sib).reshape(-1)).view(np.dtype(m["dtype"])).reshape(m["shape"])
        if zlib.crc32(np.ascontiguousarray(a).tobytes()) != m["crc"]:
            raise ValueError("%s: %s does not decode to the recorded bits" % (path, m["name"]))
        out[m["name"]] = a
    _cache[scene] = out
    return dict(out)


def verify():
    ok = True
    for scene in ORDER:
        p = os.path.join(GOLD, scene + ".npz")
        got = load(scene)
        if not os.path.exists(p):
            print("%-16s decodes (%d arrays, CRCs hold); no .npz beside it to compare with" % (scene, len(got)))
            continue
        ref = dict(np.load(p))
        same = set(ref) == set(got) and all(ref[k].dtype == got[k].dtype and ref[k].shape == got[k].shape and ref[k].tobytes() == got[k].tobytes() for k in ref)
        print("%-16s %s" % (scene, "identical to the .npz" if same else "DIFFERS"))
        ok = ok and same
    return ok


if __name__ == "__main__":
    if "--encode" in sys.argv:
        encode_all()
    if "--verify" in sys.argv or "--encode" in sys.argv:
        sys.exit(0 if verify() else 1)
