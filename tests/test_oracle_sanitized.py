"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (+ float-cast-overflow): a checker that reads out of bounds or leans on
undefined behaviour pins nothing.  `make -C oracle san` builds oracle/_build/liboracle_san.so; a child process (LD_PRELOAD = libasan)
runs the fuzzers' draws with the oracle on both sides (tests/oracle_only.py): adversarial pixels at the stage seams, random switch
combinations with ragged / empty frames and garbage map rows, random API call sequences.  Any report aborts the child.

Round 6: the first run found (long long) of a 1.9e20 quadrant count in hd_sincos (an SE3 step solved from garbage images) — the shared
header now states that conversion (hd_cvt_i64), like the two the GPU fuzzers found (hd_cvt_i32, hd_cvt_u32)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def san_env():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan.so beside this gcc")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", HRBF_ORACLE_SAN="1")
    env.pop("HRBF_ORACLE_MUTANT", None)
    return env


@pytest.mark.parametrize("what,n,seed", [("stages", 24, 3), ("params", 4, 11), ("api", 2, 5)])
def test_oracle_runs_the_fuzzers_draws_without_a_sanitizer_report(san_env, what, n, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_only.py"), what, str(n), str(seed)], env=san_env, capture_output=True, text=True, timeout=1500)
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-1500:]
    assert p.returncode == 0 and ("%s: %d runs, 0 differences" % (what, n)) in p.stdout, (p.returncode, p.stdout[-300:], p.stderr[-600:])
