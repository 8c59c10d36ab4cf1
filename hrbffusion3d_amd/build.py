"""Build libhrbf_mi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m hrbffusion3d_amd.build [--force]

-ffp-contract=off is part of the arithmetic contract (include/hrbf_detmath.h): fused multiply-adds
are written explicitly so that device results are bit-identical to the CPU oracle.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libhrbf_mi355.so")
SOURCES = ["abi.hip", "k_pre.hip", "k_map.hip", "k_predict.hip", "k_odo.hip", "k_fit.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("hrbf_mi355.h", "hrbf_detmath.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: experiment variants, e.g. build(True, defines=["-DFUSE_IPT=4"], out="libhrbf_v1.so") — they land in
    hrbffusion3d_amd/_build/ (git-ignored), never next to the product library (load one with HRBF_LIB=<path>)"""
    global OUT
    if out is not None:
        os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
        return _build_to(os.path.join(HERE, "_build", os.path.basename(out)), list(defines), verbose, "_" + os.path.splitext(os.path.basename(out))[0])
    if not force and not _stale():
        return OUT
    return _build_to(OUT, [], verbose, "")


def _build_to(OUT, defines, verbose, suffix):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
        cmd = [hipcc, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj] + FLAGS + defines
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s ==\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    if suffix:      # an experiment variant: its objects are not a cache for anything, do not leave them in csrc/
        for obj in objs:
            os.remove(obj)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
