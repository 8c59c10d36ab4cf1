"""Do the metamorphic tests of tests/test_registration_metamorphic.py have teeth?  Runs them against 26 deliberately MISREAD
builds of the oracle's registration — 58 since round 6 — (oracle/orc_odo.c, orc_ctx.c, `#if ORC_MUTANT == k`; `make -C oracle mutants`) and reports which tests
fail on which misreading.  A misreading no test fails on is a blind spot of the suite — it is listed as such.

    python tools/mutation_report.py [k ...] > profiles/r06_mutation_report.txt        (build container or any CPU host; ~40 minutes)
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
    # round 6: the CUDA rows round 5's misreadings did not touch (map building, rejection, search, sparse ICP, guard, weighting)
    27: "model normals moved like points: R n + t (tranformMapsKernel, cudafuncs.cu:246)",
    28: "model vertices rotated only: R v without t (cudafuncs.cu:230)",
    29: "the model's principal directions left in the camera frame (transformCurvMaps skipped, cudafuncs.cu:279-322)",
    30: "2x2 resize as a NaN-aware mean of the valid taps instead of 'NaN if any tap is NaN' (cudafuncs.cu:549-556)",
    31: "resized normals not renormalised (cudafuncs.cu:573-581)",
    32: "copyMaps validity on the vertex alone: `nsrc.w > 0` dropped (cudafuncs.cu:366)",
    33: "curvature validity `kappa < thr` without `> -thr` (cudafuncs.cu:420)",
    34: "icp weight kept when >= 0 instead of > 0 (cudafuncs.cu:462)",
    35: "ICP distance threshold met by the squared distance (reduce.cu:383)",
    36: "ICP angle threshold met by 1 - cosine instead of the sine (reduce.cu:383)",
    37: "ICP distance threshold on the depth difference instead of the Euclidean distance (reduce.cu:381-383)",
    38: "search ties go to the LAST candidate: `<=` for `<` (reduce.cu:429)",
    39: "search cost: D_p normalised by the distance threshold instead of the window's largest accepted distance (reduce.cu:397-398,421)",
    40: "sparse ICP: h = s - d - lambda / mu (reduce.cu:482)",
    41: "sparse ICP: target moved by z + lambda / mu instead of z - lambda / mu (reduce.cu:485)",
    42: "updateLambdaMap: lambda - mu * Delta (cudafuncs.cu:1066-1067)",
    43: "no 0.3 m guard (RGBDOdometry.cpp:1232-1236)",
    44: "velocity weighting from the translation alone: no max with the rotation angle (HRBFFusion.cpp:1116)",
    45: "velocity weighting without the lower clamp minWeight (HRBFFusion.cpp:1124)",
    46: "velocity weighting with weightMultiplier inside the clamp (HRBFFusion.cpp:1124)",
    47: "sparse ICP: the l1 soft threshold instead of the l_p (p = 0.5) shrink operator (reduce.cu:302-315,652)",
    48: "intensity pyramid averaging every tap, also the black (no data) pixels (pyrDownKernelIntensityGauss, cudafuncs.cu:836-841)",
    49: "verticesToDepth without the far cut-off: only z <= 0 invalid (cudafuncs.cu:874-885, populateRGBDData's 6 m)",
    50: "RGB residual: the 'not an isolated pixel' window dropped (reduce.cu:1003-1010)",
    51: "RGB residual: a black model pixel accepted (`lastImage != 0` dropped, reduce.cu:1039)",
    52: "RGB residual: no border margin (`j0 < cols - 5 && i < rows - 1` dropped, reduce.cu:999)",
    53: "SO3 row: image gradient from the warped live image alone instead of the mean of both images' (reduce.cu:1224-1225)",
    54: "RGB step: the row's 3-D point read at the live pixel (`one`) instead of the model pixel (`zero`) (reduce.cu:744-746)",
    55: "RGB step: the rgbOnly signal sigma == -1 not honoured (reduce.cu:737-740)",
    56: "RGB step: gradient weight exp(-0.5 (grad / 10)^2) instead of exp(-0.5 (10 / grad)^2) (reduce.cu:757-758)",
    57: "Sobel: the kernel entry taken from the tap's offset, without the running index that slips where the border cuts the window (cudafuncs.cu:937-946)",
    58: "resizeCMap renormalising the averaged principal direction like a normal (cudafuncs.cu:618-674 takes the plain mean of all four planes)",
}
MODULES = ["tests/test_registration_metamorphic.py", "tests/test_registration_metamorphic2.py"]


def run(mutant, module):
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
    only = [int(a) for a in sys.argv[1:]] or list(MUTANTS)
    print("# tools/mutation_report.py: the CPU tests of %s against deliberate misreadings of the registration" % " + ".join(MODULES))
    for mod in MODULES:
        failed, passed = run(0, mod)
        print("oracle as it is, %s: %d passed, %d failed" % (mod, passed, len(failed)))
        assert not failed, failed
    caught = 0
    for k in only:
        what = MUTANTS[k]
        failed, passed = [], 0
        for mod in MODULES:
            f, p = run(k, mod)
            failed += [os.path.basename(mod)[:-3].replace("test_registration_", "") + "::" + t for t in f]; passed += p
        caught += bool(failed)
        print("\nmutant %2d  %s\n  -> %d of %d tests fail%s" % (k, what, len(failed), len(failed) + passed, "" if failed else "   ** NOT CAUGHT: a blind spot of these tests **"))
        for f in failed:
            print("       " + f)
        sys.stdout.flush()
        if mutant_needs_second_look(failed):
            for mod in RESTATEMENTS:
                f2, p2 = run(k, mod)
                print("     (%s: %d of %d fail)" % (mod, len(f2), len(f2) + p2))
    print("\n%d of %d misreadings are caught by at least one test" % (caught, len(only)))


if __name__ == "__main__":
    main()
