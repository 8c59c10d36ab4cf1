#!/bin/bash
# A/B of experiment builds on the GPU box: quick bench legs (headline stream + worst-case fuse leg), no CPU baseline.
# usage: tools/ab_bench.sh [--reps N] default <variant> ...   where <variant> names hrbffusion3d_amd/_build/libhrbf_v_<variant>.so
# (build with hrbffusion3d_amd.build.build(True, defines=[...], out="libhrbf_v_<variant>.so")); runs alternate to average drift out.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
REPS=1
if [ "$1" = "--reps" ]; then REPS=$2; shift 2; fi
Q="--steps 120 --warmup 20 --cpu-frames 0 --cpu-frames-1t 0 --big-surfels 0 --no-cpp-shim"
for r in $(seq 1 $REPS); do
  for v in "$@"; do
    if [ "$v" = default ]; then lib=libhrbf_mi355.so; else lib=_build/libhrbf_v_$v.so; fi
    HRBF_LIB=$lib python bench.py $Q > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    python - "$v" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % name).read().strip().splitlines()[-1])
    r, w = d["roofline"], d.get("roofline_worst_case", {})
    print("%-14s fps %7.1f  headline frac %.3f merge %.1f + stream %.1f us | worst frac %.3f merge %.1f + stream %.1f us" % (
        name, d["value"], r["frac"], r.get("merge_ms", 0) * 1e3, r.get("clean_compact_ms", 0) * 1e3,
        w.get("frac", 0), w.get("merge_ms", 0) * 1e3, w.get("clean_compact_ms", 0) * 1e3))
except Exception as e:
    print(name, "FAILED", e)
PY
  done
done
