"""The fuzzers' drivers with the oracle on BOTH sides (no GPU): used to check that every draw is something the oracle survives before it
is sent to a GPU box, and by tests/test_oracle_sanitized.py to run the same draws through the oracle built with ASan + UBSan."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import hrbffusion3d_amd.api as api  # noqa: E402


class OracleAsLibrary(oracle_lib.Oracle):
    """the oracle behind the library's Python surface: what only the library has (shards, timing, the ring) does nothing"""
    def __init__(self, p, device=0):
        super().__init__(p, omp=False)

    def comm_init(self, *a): pass
    def map_shard_init(self, *a, **k): pass
    def set_row_sharding(self, *a): pass
    def map_rebalance(self): pass
    def enable_timing(self, *a): pass
    def set_fuse_ring_stride(self, *a): pass
    def reset_fuse_ring(self): pass
    def synchronize(self): pass
    def status(self): return 0
    def local_surfel_count(self): return self.surfel_count()

    def process_frame_device(self, *a):
        raise RuntimeError("no device")

    def __getattr__(self, n):
        if n.startswith("set_") and n[4:] in oracle_lib.Oracle.SWITCHES:
            return lambda v: self.set_switch(n[4:], v)
        raise AttributeError(n)


def install():
    os.environ["HRBF_FUZZ_NO_DEVICE"] = "1"
    api.HRBFFusion = OracleAsLibrary
    return oracle_lib


def main():
    """python tests/oracle_only.py params|stages|api|pair N [seed]"""
    import numpy as np
    what, n = sys.argv[1], int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ol = install()
    if what == "pair":
        from PIL import Image
        from hrbffusion3d_amd.params import default_params
        g = os.path.join(ROOT, "tests", "golden")
        o = oracle_lib.Oracle(default_params(max_surfels=1 << 20), omp=False)
        for a, b in (("1c", "1d"), ("2c", "2d")):
            o.process_frame(np.array(Image.open(os.path.join(g, a + ".png"))), np.array(Image.open(os.path.join(g, b + ".png"))))
        print("pair: %d surfels" % o.surfel_count()); o.close()
        return 0
    if what == "params":
        import gpu_fuzz_params as F
        run = lambda i: F.run_one(ol, *F.draw(seed, i))
    elif what == "stages":
        import gpu_fuzz_stages as S
        run = lambda i: S.trial(ol, seed, i)
    else:
        import gpu_fuzz_api as A
        run = lambda i: A.trial(ol, seed, i)
    bad = 0
    for i in range(n):
        r = run(i)
        if r is not None and "no device" not in r and "HIP" not in r:
            bad += 1; print(what, i, r[:300])
    print("%s: %d runs, %d differences between two oracles" % (what, n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
