#!/usr/bin/env python3
"""Per-shard kernel times of `bench.py --virtual-shards G` from a rocprofv3 --kernel-trace CSV: one GPU plays the G map shards in turn,
so the i-th dispatch (mod G) of a per-shard kernel belongs to shard i.  What a real node would run side by side on G ranks is here a
sequence; max / mean over the shards is the imbalance the ownership function leaves (round-3 verdict item 8).
usage: per_shard_times.py <x_kernel_trace.csv> G"""
import collections
import csv
import sys

KERNELS = ("k_project", "k_keys_global", "k_resolve", "k_apply_merges", "k_clean_flags", "k_fuse_stream")


def main():
    path, G = sys.argv[1], int(sys.argv[2])
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].split("(")[0]
        if n in KERNELS:
            per[n].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    tot_max = tot_sum = 0.0
    for n in KERNELS:
        if n not in per:
            continue
        v = [d for _, d in sorted(per[n])]
        v = v[-(len(v) // G // 2 * G):]                   # the second half of the run
        calls = len(v) // G
        sh = [sum(v[i::G]) / len(v[i::G]) for i in range(G)]
        per_frame = len(per[n]) // G
        print("%-16s per shard, mean us: %s   max / mean %.2f   (%d dispatches per shard)" % (n, "  ".join("%5.1f" % x for x in sh), max(sh) / (sum(sh) / G), calls))
        k = 3 if n in ("k_project", "k_keys_global", "k_resolve") else 1      # three projections per frame
        tot_max += k * max(sh); tot_sum += k * sum(sh)
    print("map passes of one frame: slowest shard %.0f us, all shards one after the other %.0f us" % (tot_max, tot_sum))


if __name__ == "__main__":
    main()
