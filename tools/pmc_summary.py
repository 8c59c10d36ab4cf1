#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one *_counter_collection.csv per pass) into one per-kernel table.

    pmc_summary.py out.csv pass1_dir pass2_dir ...

Every counter is averaged over the dispatches of a (kernel, grid) pair; the durations are the averages of the
pass that carried the counter (profiled passes run at a lower clock than un-profiled ones — compare counters with
counters, not with the kernel-trace times).  Derived columns (only when their inputs were collected):
  valu_active_frac   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   share of the resident waves' cycles spent issuing VALU (quad-cycles)
  wait_frac          SQ_WAIT_ANY / SQ_WAVE_CYCLES           waves parked (s_waitcnt / barrier)
  issue_stall_frac   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict_cycles_per_lds_cycle  SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS   conflict cycles per cycle an LDS instruction is active (a ratio, can exceed 1)
  fp32_gflop         (2 FMA + ADD + MUL + TRANS) x 64 lanes per wave instruction
  fp32_tflops        fp32_gflop / duration                  to be read against the 157 TFLOP/s fp32 vector peak
"""
import collections
import csv
import glob
import os
import sys


def main():
    out = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))   # (kernel, grid) -> counter -> values
    dur = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.defaultdict(set)
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"].split("(")[0], r["Grid_Size"])
                c = r["Counter_Name"]
                acc[k][c].append(float(r["Counter_Value"]))
                key = (f, r["Dispatch_Id"])
                if key not in seen[(k, c)]:
                    seen[(k, c)].add(key)
                    dur[k][c].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = sorted({c for v in acc.values() for c in v})
    derived = ["valu_active_frac", "wait_frac", "issue_stall_frac", "lds_conflict_cycles_per_lds_cycle", "fp32_gflop", "fp32_tflops", "valu_insts_per_wave"]
    with open(out, "w") as fo:
        fo.write(",".join(["kernel", "grid", "dispatches", "avg_us_profiled"] + counters + derived) + "\n")
        rows = []
        for k, cs in acc.items():
            a = {c: sum(v) / len(v) for c, v in cs.items()}
            any_c = next(iter(cs))
            n = len(cs[any_c])
            alld = [x for c in dur[k] for x in dur[k][c]]
            us = sum(alld) / len(alld)
            g = lambda name: a.get(name)
            dv = {}
            wc = g("SQ_WAVE_CYCLES")
            if wc:
                for name, src in (("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY")):
                    if g(src) is not None:
                        dv[name] = g(src) / wc
            if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_ACTIVE_INST_LDS"):
                dv["lds_conflict_cycles_per_lds_cycle"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS")
            if g("SQ_INSTS_VALU_FMA_F32") is not None:
                fl = 64.0 * (2 * g("SQ_INSTS_VALU_FMA_F32") + (g("SQ_INSTS_VALU_ADD_F32") or 0) + (g("SQ_INSTS_VALU_MUL_F32") or 0) +
                             (g("SQ_INSTS_VALU_TRANS_F32") or 0))
                dv["fp32_gflop"] = fl / 1e9
                d3 = dur[k].get("SQ_INSTS_VALU_FMA_F32")
                if d3:
                    dv["fp32_tflops"] = fl / (sum(d3) / len(d3) * 1e-6) / 1e12
            if g("SQ_INSTS_VALU") is not None and g("SQ_WAVES"):
                dv["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
            rows.append((us * n, [k[0], k[1], str(n), "%.2f" % us] + ["%.6g" % a[c] if c in a else "" for c in counters] +
                         ["%.4g" % dv[x] if x in dv else "" for x in derived]))
        for _, r in sorted(rows, key=lambda t: -t[0]):
            fo.write(",".join(r) + "\n")
    print("wrote", out, len(rows), "kernel rows,", len(counters), "counters")


if __name__ == "__main__":
    main()
