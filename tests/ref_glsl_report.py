"""Prints the per-pass comparison of an implementation with the executed reference shaders (tests/golden/ref_glsl).
    python tests/ref_glsl_report.py oracle|hip [scene ...]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_glsl_check as R  # noqa: E402
import make_ref_glsl as M  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
    for scene in sys.argv[2:] or ("pair", "sphere"):
        print("=====", which, scene)
        fx = R.load(scene)
        if which == "hip":
            from hrbffusion3d_amd.api import HRBFFusion
            impl = HRBFFusion(M.params(scene))
        else:
            from oracle_lib import Oracle
            impl = Oracle(M.params(scene), omp=True)
        rep = R.run(impl, fx, R.Report(strict=False, verbose=True))
        print("failed:", [w for w, ok, _ in rep.rows if not ok])
