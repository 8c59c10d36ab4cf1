"""N > 1 path on CPU (gloo, world_size 2).

(a) bench.py's replica harness: barrier + max-over-ranks timing + whole-job aggregation.
(b) the arithmetic that makes a sharded reduction safe: ICP normal equations accumulated as exact
    int64 limbs per rank (rows split across ranks), all-reduced with SUM, equal the single-process
    result BIT FOR BIT (all-reduce of the 6x6 system over xGMI)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def _limbs(v):
    """hd_limbs_from_f32 restated in Python ints (round_half_even(p * 2^40) split in 40-bit limbs)"""
    from fractions import Fraction
    out = np.zeros((len(v), 3), np.int64)
    for i, x in enumerate(v):
        q = int(round(Fraction(float(x)) * (1 << 40)))
        q &= (1 << 128) - 1
        l0 = q & ((1 << 40) - 1); l1 = (q >> 40) & ((1 << 40) - 1); l2 = q >> 80
        if l2 >= 1 << 47:
            l2 -= 1 << 48
        out[i] = (l0, l1, l2)
    return out


def _combine(s):
    q = int(s[0]) + (int(s[1]) << 40) + (int(s[2]) << 80)
    return q / float(1 << 40)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    v = (rng.standard_normal(4000) * np.exp(rng.uniform(-12, 12, 4000))).astype(np.float32)
    mine = v[rank::world]
    part = torch.from_numpy(_limbs(mine).sum(0))
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    # replica timing harness
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((part.numpy().copy(), float(t.item())))
    dist.destroy_process_group()


def test_limb_allreduce_is_exact_and_timing_is_max_over_ranks(oracle_lib_built):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    part, tmax = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(tmax - 0.020) < 1e-12
    # single-process reference through the C accumulator of the oracle build
    rng = np.random.default_rng(7)
    v = (rng.standard_normal(4000) * np.exp(rng.uniform(-12, 12, 4000))).astype(np.float32)
    lib = oracle_lib_built.load()
    out = C.c_double()
    lib.orc_acc_test(v.ctypes.data_as(C.c_void_p), v.size, C.byref(out))
    assert _combine(part) == out.value


def test_bench_json_contract_fields():
    """static check of the bench line's keys (the values need a GPU)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "workload", "achieved", "peak",
                "frac", "traffic", "cores", "kind", "sample"):
        assert '"%s"' % key in src, key
