"""GPU-only determinism probe: repeat a short sequence, report the first quantity that differs from run 0."""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params, IMAGES
W, H = 160, 120
fx, fy, cx, cy = synth.intrinsics(W, H)
variants = {"gauss_filter": dict(use_bilateral=0), "default": dict(), "no_so3": dict(so3=0)}
frames = [synth.frame(k, W, H, noise=True) for k in range(4)]
def run(kw):
    g = HRBFFusion(default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 17, **kw))
    out = []
    for k in range(4):
        g.process_frame(frames[k][0], frames[k][1])
        rec = {"pose": g.get_pose().tobytes(), "count": g.surfel_count()}
        for name in IMAGES:
            rec[name] = hashlib.md5(np.ascontiguousarray(g.get_image(name)).view(np.uint8).tobytes()).hexdigest()
        out.append(rec)
    g.close()
    return out
for vname, kw in variants.items():
    ref = run(kw)
    nbad = 0
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        cur = run(kw)
        for k in range(4):
            diff = [n for n in ref[k] if ref[k][n] != cur[k][n]]
            if diff:
                nbad += 1
                print(vname, "run", r, "frame", k, "differs:", diff[:12])
                break
    print(vname, "bad runs:", nbad)
