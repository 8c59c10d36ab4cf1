#!/bin/bash
# PMC evidence for the ALU-bound kernels and HBM traffic of the fuse pass: one rocprofv3 --pmc pass per counter group
# (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass), nothing but --pmc in each (the pool refuses
# counter collection combined with the trace domains).  Run on the GPU box:  bash tools/collect_pmc.sh <outdir> [lib]
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; export HRBF_BENCH_GEN_PROCS=1   # no forked frame generators under the profiler
OUT=${1:-gpurun_out/r05/pmc}; mkdir -p "$OUT"
[ -n "${2:-}" ] && export HRBF_LIB=$2
CMD="python bench.py --steps 10 --warmup 3 --cpu-frames 0 --worst-surfels 0 --big-surfels 0 --no-cpp-shim --no-traffic --no-fit-leg"
run() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o x -- $CMD > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run s1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run s2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run s3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64
run g1 GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES
run t1 FETCH_SIZE
run t2 WRITE_SIZE
python tools/pmc_summary.py "$OUT/summary_frame.csv" "$OUT/s1" "$OUT/s2" "$OUT/s3" "$OUT/g1" "$OUT/t1" "$OUT/t2"
# the worst-case fuse leg (4.3 M surfels, whole map moved): HBM traffic of its three kernels
CMD="python bench.py --only-worst --worst-samples 3 --no-traffic"
run w1 FETCH_SIZE
run w2 WRITE_SIZE
python tools/pmc_summary.py "$OUT/summary_worst.csv" "$OUT/w1" "$OUT/w2"
du -sh "$OUT"; find "$OUT" -name '*counter_collection.csv' -size +20M -delete
