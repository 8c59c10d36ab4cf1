#!/bin/bash
# Measurement: hardware counters of k_hrbf_fit per wave (two rocprofv3 --pmc passes over tools/probes/fit_time.py) -> $1/fit_pmc_raw.txt
OUT=${1:-gpurun_out/fit_pmc}; mkdir -p $OUT; ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d $ROOT/$OUT/$n -o x --output-format csv -- python $ROOT/tools/probes/fit_time.py > $ROOT/$OUT/$n.log 2>&1; }
pass p1 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES
pass p2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_hrbf_fit" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/fit_pmc_raw.txt", "w") as o:
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            line = "%-26s %-32s dispatches %3d  mean %14.1f  per wave %10.1f" % (k, c, len(v), sum(v) / len(v), sum(v) / len(v) / 307200.0)
            print(line); o.write(line + "\n")
PY
