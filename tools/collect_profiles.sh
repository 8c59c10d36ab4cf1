#!/bin/bash
# Per-leg kernel statistics of bench.py (rocprofv3 --kernel-trace --stats), one CSV per leg so that the roofline fractions can be
# reproduced without subtracting legs.  Run on the GPU box:  bash tools/collect_profiles.sh <outdir>
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; export HRBF_BENCH_GEN_PROCS=1   # no forked frame generators under the profiler
OUT=${1:-gpurun_out/r05/prof}; mkdir -p "$OUT"
leg() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o x -- "$@" > "$OUT/$name.log" 2>&1 || echo "leg $name failed";
        f=$(find "$OUT/$name" -name '*kernel_trace.csv' | head -1); python tools/prof_summary.py "$f" 40 "$OUT/${name}_kernel_stats.csv" > "$OUT/${name}_summary.txt";
        s=$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" "$OUT/${name}_rocprof_stats.csv"; }
# headline leg only: 640x480 stream, 1.06 M-surfel map (no worst-case leg, no 1280x960 leg, no CPU baseline, no C++ leg)
leg bench python bench.py --steps 100 --warmup 20 --cpu-frames 0 --worst-surfels 0 --big-surfels 0 --no-cpp-shim --no-traffic --no-fit-leg
# the extension leg (true Hermite-RBF fit on the matrix core), on its own so that it does not dilute the frame's shares
leg fit python bench.py --steps 5 --warmup 3 --cpu-frames 0 --worst-surfels 0 --big-surfels 0 --no-cpp-shim --no-traffic
# worst-case fuse leg only: 4.34 M surfels, every survivor moves
leg worst python bench.py --only-worst --worst-samples 5
# the hash-owned map played as 4 virtual shards on this one GPU (BASELINE config 4's shape: 640x480, 4.3 M surfels): per-shard kernel times
# are what the spatial hash is meant to balance across ranks (round-3 verdict item 8); grid sizes tell the shards apart
leg hash4 python bench.py --virtual-shards 4 --partition hash --surfels 4300000 --steps 60 --warmup 10 --cpu-frames 0 --worst-surfels 0 --big-surfels 0 --no-cpp-shim --no-traffic --no-fit-leg
leg ranges4 python bench.py --virtual-shards 4 --partition ranges --surfels 4300000 --steps 60 --warmup 10 --cpu-frames 0 --worst-surfels 0 --big-surfels 0 --no-cpp-shim --no-traffic --no-fit-leg
find "$OUT" -name '*kernel_trace.csv' -size +30M -delete
tail -3 "$OUT/bench.log" | head -1 | cut -c1-400
head -32 "$OUT/bench_summary.txt"; head -12 "$OUT/worst_summary.txt"; head -16 "$OUT/hash4_summary.txt"; head -16 "$OUT/ranges4_summary.txt"
