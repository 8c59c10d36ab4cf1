"""Do the metamorphic tests of tests/test_registration_metamorphic.py have teeth?  Runs them against 26 deliberately MISREAD
builds of the oracle's registration (oracle/orc_odo.c, `#if ORC_MUTANT == k`; `make -C oracle mutants`) and reports which tests
fail on which misreading.  A misreading no test fails on is a blind spot of the suite — it is listed as such.

    python tools/mutation_report.py > profiles/r05_metamorphic_mutation_report.txt        (build container or any CPU host; ~20 minutes)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MUTANTS = {
    1: "ICP: the matched normal left in the tracker's world frame (no Rprev_inv on n; reduce.cu:488)",
    2: "ICP: residual n.(d - s) instead of n.(s - d) (reduce.cu:507)",
    3: "RGB: rotational columns of the photometric row with the opposite sign (reduce.cu:752-754)",
    4: "RGB residual: model image looked up at the truncated instead of the nearest texel (reduce.cu:1027-1028)",
    5: "RGB step: Sobel scale 1/4 instead of 1/8 (RGBDOdometry.cpp:51)",
    6: "joint system: A_rgb + w A_icp instead of w^2 (RGBDOdometry.cpp:1171)",
    7: "joint system: b_rgb + w^2 b_icp instead of w (RGBDOdometry.cpp:1172) - the 'consistent' weighting the reference does NOT use",
    8: "sigma = rms residual instead of sqrt(count) (the precedence quirk of RGBDOdometry.cpp:1017 'corrected')",
    9: "pose composition T_prev * dT instead of T_prev * dT^-1 (RGBDOdometry.cpp:1198)",
    10: "pyramid intrinsics: principal point not divided by 2^level (CameraModel::operator(), types.cuh:84-87)",
    11: "SO3: residual with the opposite sign (reduce.cu:1249)",
    12: "RGB step: gradient read at the model pixel `zero` instead of the live pixel `one` (reduce.cu:745-746)",
    13: "ICP: rotational columns n x s instead of s x n (reduce.cu:497-503)",
    14: "ICP: associated texel by truncation instead of __float2int_rn (reduce.cu:411-412)",
    15: "ICP: live vertex projected without the previous pose (no Rprev_inv (v - tprev); reduce.cu:404-409)",
    16: "RGB residual: depth gate on the live pixel's own depth instead of its depth in the model camera (reduce.cu:1033)",
    17: "RGB residual: warp built from resultRt itself instead of its inverse (RGBDOdometry.cpp:981)",
    18: "RGB step: the depth column of the row without its second division by z (reduce.cu:749)",
    19: "RGB step: weight 1/sigma, no down-weighting of large residuals (reduce.cu:733-735)",
    20: "depth pyramid by plain subsampling instead of the NaN-aware 5x5 binomial (cudafuncs.cu:493-524)",
    21: "iteration schedule 10/5/4 given to the levels the other way round (RGBDOdometry.cpp:897-903)",
    22: "increment composed on the right: resultRt * update (OdometryProvider.h:91)",
    23: "gradient threshold compared unsquared with the squared magnitude (RGBDOdometry.cpp:991)",
    24: "SO3 homography with the rotation transposed, K R^T K^-1 (RGBDOdometry.cpp:832)",
    25: "a different but CONSISTENT intensity image: textbook luma instead of the reference's 0.114 R + 0.299 G + 0.587 B (cudafuncs.cu:896-911)",
    26: "Sobel kernels swapped: dIdx holds the vertical derivative (cudafuncs.cu:927-954)",
}


def run(mutant, module="tests/test_registration_metamorphic.py"):
    env = dict(os.environ)
    if mutant:
        env["HRBF_ORACLE_MUTANT"] = str(mutant)
    r = subprocess.run([sys.executable, "-m", "pytest", module, "-m", "not gpu", "-q", "-p", "no:cacheprovider",
                        "-rf", "--tb=no"], cwd=ROOT, env=env, capture_output=True, text=True)
    failed = sorted(set(re.findall(r"FAILED %s::(\S+)" % re.escape(module), r.stdout)))
    m = re.search(r"(\d+) passed", r.stdout)
    return failed, int(m.group(1)) if m else 0


# the builder's own second readings (numpy restatements / hand-computed seams): what ELSE in the CPU suite a misreading trips
RESTATEMENTS = ["tests/test_registration_fp64.py", "tests/test_intrinsics_kat.py"]


def mutant_needs_second_look(failed):
    return len(failed) <= 1


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "mutants"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    failed, passed = run(0)
    print("# tools/mutation_report.py: the CPU tests of tests/test_registration_metamorphic.py against deliberate misreadings of the registration")
    print("oracle as it is: %d passed, %d failed" % (passed, len(failed)))
    assert not failed, failed
    caught = 0
    for k, what in MUTANTS.items():
        failed, passed = run(k)
        caught += bool(failed)
        print("\nmutant %2d  %s\n  -> %d of %d tests fail%s" % (k, what, len(failed), len(failed) + passed, "" if failed else "   ** NOT CAUGHT: a blind spot of these tests **"))
        for f in failed:
            print("       " + f)
        if mutant_needs_second_look(failed):
            for mod in RESTATEMENTS:
                f2, p2 = run(k, mod)
                print("     (%s: %d of %d fail)" % (mod, len(f2), len(f2) + p2))
    print("\n%d of %d misreadings are caught by at least one test" % (caught, len(MUTANTS)))


if __name__ == "__main__":
    main()
