"""Every entry point of include/hrbf_mi355.h that takes a handle, called on a LIVE context with every other argument zero / NULL — each in its
own process, so that a crash is a finding and not the end of the survey.  The boundary's error convention (int status, never
exit) asks for an error code or a harmless success, never a signal.  (With a NULL handle all 74 return HRBF_ERR_INVALID: tests/test_abi.py.)

    python tests/gpu_probe_abi_zero_args.py        # prints one line per entry point that died or hung
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def handle_entry_points():
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hrbf_mi355.h")).read(), flags=re.S)
    protos = re.findall(r"\b(int|uint32_t|void|const char \*|float)\s*\*?\s*(hrbf_\w+)\s*\(([^;{]*?)\)\s*;", hdr)
    return [(rt, n, len([x for x in a.split(",") if x.strip()])) for rt, n, a in protos if a.strip().startswith("hrbf_handle")]


CHILD = """
import ctypes as C, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params
g = HRBFFusion(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 16))
if %d:
    for k in range(2):
        rgb, d, _ = synth.frame(k, 160, 120, noise=True); g.process_frame(rgb, d)
f = getattr(g.lib, %r)
f.restype = C.c_int; f.argtypes = None
r = f(*([g.h] + [C.c_void_p(0)] * %d))
print("RET", r, flush=True)
if %r != "hrbf_destroy":
    rgb, d, _ = synth.frame(2, 160, 120, noise=True)
    try:
        g.process_frame(rgb, d); g.synchronize(); print("NEXT frame ok", flush=True)
    except Exception as e:
        print("NEXT frame:", repr(e)[:120], flush=True)
else:
    g.h = None
"""


def survey(warm_states=(0, 1)):
    bad = []
    eps = handle_entry_points()
    for warm in warm_states:
        for rt, n, nargs in eps:
            code = CHILD % (ROOT, os.path.join(ROOT, "tests"), warm, n, nargs - 1, n)
            try:
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
            except subprocess.TimeoutExpired:
                bad.append("%s (context %s): hung" % (n, "after two frames" if warm else "fresh")); continue
            lines = [l for l in p.stdout.splitlines() if l.startswith(("RET", "NEXT"))]
            if p.returncode != 0:
                err = [l for l in p.stderr.strip().splitlines() if l.strip()]
                bad.append("%s (context %s): exit %d after %s | %s" % (n, "after two frames" if warm else "fresh", p.returncode, lines, err[-1][:160] if err else ""))
    return eps, bad


def main():
    eps, bad = survey()
    print("%d entry points x {fresh context, context after two frames}: %d died or hung" % (len(eps), len(bad)))
    for b in bad:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
