#!/usr/bin/env python3
"""Dependent load chains the compiler leaves in a kernel ("dependent round trips the compiler adds").

    python tools/isa_load_chains.py [k_odo k_map ...]      # default: every .hip file of hrbffusion3d_amd/csrc

Compiles each file with the library's flags and -save-temps into /tmp, then reports per kernel: the number of vector loads, the
number of `s_waitcnt vmcnt(..)` that follow at least one new load (an upper bound on the dependent round trips of the longest
path: every branch of the kernel is counted), and the longest run of "ONE load, then s_waitcnt vmcnt(0)" — the signature of a
tested field fetched alone, of a load under a branch, or of a loop that loads, waits and stores.  A latency-bound kernel (one
wave per SIMD, a chain of round trips) with a long run is worth reading; a VALU-bound tile kernel is not (measured)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hrbffusion3d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-save-temps=obj"]


def main():
    names = sys.argv[1:] or [os.path.splitext(os.path.basename(f))[0] for f in sorted(glob.glob(os.path.join(CSRC, "k_*.hip")))]
    for n in names:
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "-c", os.path.join(CSRC, n + ".hip"), "-o", "/tmp/%s.o" % n] + FLAGS,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernel, seq, rows = None, [], []
        for line in open("/tmp/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % n):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kernel, seq = m.group(1), []
                continue
            if kernel is None:
                continue
            if re.search(r"\b(global_load|buffer_load|flat_load)", line):
                seq.append("L")
            elif "s_waitcnt" in line and "vmcnt" in line:
                seq.append("W" if "vmcnt(0)" in line else "w")
            elif "s_endpgm" in line:
                s = "".join(seq)
                rounds = len(re.findall(r"L[^Ww]*[Ww]", s))
                run = max([len(x.group(0)) // 2 for x in re.finditer(r"(?:LW){2,}", s)] or [0])
                rows.append((kernel, s.count("L"), rounds, run))
                kernel = None
        for k, loads, rounds, run in rows:
            if loads:
                print("%-10s %-64s loads %3d  waits-after-loads %3d  longest single-load chain %2d" % (n, k[:64], loads, rounds, run))


if __name__ == "__main__":
    main()
