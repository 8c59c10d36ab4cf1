#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count, average / min duration, share.
usage: prof_summary.py <..._kernel_trace.csv> [top_n] [out.csv]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].split("(")[0]
        acc[(n, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in acc.values())
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
    if out:
        out.write("kernel,grid_x,calls,avg_us,min_us,total_us,share_pct\n")
    for k, v in rows:
        line = "%s,%s,%d,%.2f,%.2f,%.1f,%.2f" % (k[0], k[1], len(v), sum(v) / len(v), min(v), sum(v), 100 * sum(v) / tot)
        if out:
            out.write(line + "\n")
    for k, v in rows[:top]:
        print("%-40s grid=%-8s n=%-5d avg=%8.1f us  min=%8.1f  share=%5.1f%%" % (k[0][:40], k[1], len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
    print("total kernel time %.1f us" % tot)


if __name__ == "__main__":
    main()
