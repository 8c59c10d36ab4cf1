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


def _shard_worker(rank, world, port, q):
    """the exchange protocol of the sharded surfel map (csrc/abi.hip st_indices / hrbf_map_rebalance) on two real
    processes: global ids in the z-buffer keys, MIN over the keys, then the WINNER-RECORD exchange (every rank packs
    {pixel index | updated bit, attributes} of the pixels whose winner it owns, the record counts are all-gathered, the
    records travel as variable-length send / recv and are scattered into the zero-filled images), all-gather of the
    counts, and the re-cut of the ranges with the moves of hrbf_rebalance_plan as send / recv."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hrbffusion3d_amd import api
    rng = np.random.default_rng(11)
    N, P = 5000, 64 * 48
    pix = rng.integers(0, P, N); depth = rng.integers(1, 200, N).astype(np.float32) * 0.01   # many exact depth ties
    attr = rng.standard_normal((N, 4)).astype(np.float32); attr[::7] = -0.0                   # -0.0 must survive the sum
    counts = np.array([1200, 3800], np.int64)                                                  # uneven on purpose
    off = int(counts[:rank].sum()); n = int(counts[rank])
    EMPTY = np.iinfo(np.int64).max
    z = np.full(P, EMPTY, np.int64)
    keys = (depth[off:off + n].view(np.uint32).astype(np.int64) << 32) | np.arange(off, off + n, dtype=np.int64)
    np.minimum.at(z, pix[off:off + n], keys)
    zt = torch.from_numpy(z); dist.all_reduce(zt, op=dist.ReduceOp.MIN)
    win = zt.numpy() & 0xFFFFFFFF
    hit = zt.numpy() != EMPTY
    img = np.zeros((P, 4), np.float32)
    own = hit & (win >= off) & (win < off + n)
    img[own] = attr[win[own]]                      # the rank's own k_resolve: its winners, zeros elsewhere
    # pack (k_resolve's record branch): pixel index with the "updated" flag in bit 31 + one attribute plane
    updated = (win % 5 == 0) & own
    order = rng.permutation(np.nonzero(own)[0])    # the pack order is arbitrary on the GPU (one atomic per workgroup)
    rec_idx = (order.astype(np.uint32) | (updated[order].astype(np.uint32) << 31)).astype(np.int64)
    rec_f = attr[win[order]].copy()
    cnt_t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(cnt_t, torch.tensor([len(order)], dtype=torch.int64))
    counts_rec = [int(t.item()) for t in cnt_t]
    mask = np.zeros(P, bool); mask[np.nonzero(updated)[0]] = True
    reqs = []
    for p in range(world):
        if p == rank:
            continue
        if counts_rec[rank]:
            reqs.append(dist.isend(torch.from_numpy(rec_idx.copy()), dst=p))
            reqs.append(dist.isend(torch.from_numpy(rec_f.view(np.int32).copy()), dst=p))
        if counts_rec[p]:
            ridx = torch.zeros(counts_rec[p], dtype=torch.int64); dist.recv(ridx, src=p)
            rf = torch.zeros((counts_rec[p], 4), dtype=torch.int32); dist.recv(rf, src=p)
            pix_r = (ridx.numpy() & 0x7FFFFFFF).astype(np.int64)
            assert not own[pix_r].any()            # one owner per pixel: a received record never lands on an own winner
            img[pix_r] = rf.numpy().view(np.float32)          # k_winner_unpack
            mask[pix_r[(ridx.numpy() >> 31) & 1 == 1]] = True
    for r in reqs:
        r.wait()
    it = torch.from_numpy(np.concatenate([img.view(np.int32), mask.astype(np.int32)[:, None]], 1))
    assert sum(counts_rec) == int(hit.sum())       # every hit pixel has exactly one record
    # counts all-gather + even re-cut
    mine = torch.tensor([n], dtype=torch.int64); allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, mine)
    cnt = np.array([int(t.item()) for t in allc], np.uint32)
    new, moves = api.rebalance_plan(cnt)
    shard = np.arange(off, off + n, dtype=np.int64)
    out = np.full(int(new[rank]), -1, np.int64)
    reqs = []
    for a, b, so, do, ln in moves:
        if a == rank and b == rank:
            out[do:do + ln] = shard[so:so + ln]
        elif a == rank:
            reqs.append(dist.isend(torch.from_numpy(shard[so:so + ln].copy()), dst=b))
        elif b == rank:
            t = torch.zeros(ln, dtype=torch.int64); dist.recv(t, src=a); out[do:do + ln] = t.numpy()
    for r in reqs:
        r.wait()
    q.put((rank, zt.numpy().copy(), it.numpy().copy(), out, [int(v) for v in new]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_map_exchange_protocol_world2():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict()
    for _ in range(2):
        r = q.get(timeout=120); got[r[0]] = r[1:]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: one z-buffer over all surfels, ties -> lower global id, direct gather
    rng = np.random.default_rng(11)
    N, P = 5000, 64 * 48
    pix = rng.integers(0, P, N); depth = rng.integers(1, 200, N).astype(np.float32) * 0.01
    attr = rng.standard_normal((N, 4)).astype(np.float32); attr[::7] = -0.0
    EMPTY = np.iinfo(np.int64).max
    z = np.full(P, EMPTY, np.int64)
    np.minimum.at(z, pix, (depth.view(np.uint32).astype(np.int64) << 32) | np.arange(N, dtype=np.int64))
    img = np.zeros((P, 4), np.float32); hit = z != EMPTY; img[hit] = attr[(z & 0xFFFFFFFF)[hit]]
    for r in (0, 1):
        zr, ir, out, new = got[r]
        assert np.array_equal(zr, z) and np.array_equal(ir[:, :4], img.view(np.int32))      # bit-exact incl. -0.0
        assert np.array_equal(ir[:, 4] != 0, hit & ((z & 0xFFFFFFFF) % 5 == 0))               # the "updated" mask
    assert got[0][3] == [2500, 2500]
    assert np.array_equal(np.concatenate([got[0][2], got[1][2]]), np.arange(N))


def _hash_worker(rank, world, port, q):
    """the exchange of a HASH-OWNED map (hrbf_map_shard_init(h, 2), csrc/abi.hip st_indices) on two real processes: private
    z-test keyed {depth, local index}, {depth, id of the private winner} MIN-reduced over the ranks, a rank owns a pixel iff its
    private winner's global key equals the reduced key, the owner's attributes reach every rank (one owner per pixel: exact integer
    sum), the smallest id alive is all-gathered, every rank appends the new surfels of its own cells with ids g_next + q"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hrbffusion3d_amd import api
    lib = api.load_library()
    rng = np.random.default_rng(23)
    N, P, Q = 6000, 64 * 48, 400
    pos = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
    pix = rng.integers(0, P, N); depth = rng.integers(1, 120, N).astype(np.float32) * 0.01      # many exact depth ties
    attr = rng.standard_normal((N, 4)).astype(np.float32)
    gid = np.sort(rng.choice(4 * N, N, replace=False)).astype(np.int64)                           # ids with gaps: never renumbered
    owner = np.array([lib.hrbf_hash_owner(float(x), float(y), float(z), 0.25, world) for x, y, z in pos])
    mine = np.nonzero(owner == rank)[0]                                                           # local index -> row, ascending in id
    EMPTY = np.iinfo(np.int64).max
    zp = np.full(P, EMPTY, np.int64)
    np.minimum.at(zp, pix[mine], (depth[mine].view(np.uint32).astype(np.int64) << 32) | np.arange(len(mine), dtype=np.int64))
    hitp = zp != EMPTY
    keys = np.full(P, EMPTY, np.int64)
    keys[hitp] = (zp[hitp] & ~np.int64(0xFFFFFFFF)) | gid[mine[zp[hitp] & 0xFFFFFFFF]]         # k_keys_global
    zt = torch.from_numpy(keys.copy()); dist.all_reduce(zt, op=dist.ReduceOp.MIN)
    zred = zt.numpy()
    own = hitp & (keys == zred)                                                                   # pixel_winner
    img = np.zeros((P, 4), np.float32)
    img[own] = attr[mine[zp[own] & 0xFFFFFFFF]]
    it = torch.from_numpy(img.view(np.int32).copy()); dist.all_reduce(it, op=dist.ReduceOp.SUM)   # one owner per pixel: zeros elsewhere
    nown = torch.tensor([int(own.sum())]); dist.all_reduce(nown)
    # the smallest id alive travels with the counts
    g0 = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(g0, torch.tensor([int(gid[mine[0]]) if len(mine) else EMPTY], dtype=torch.int64))
    gfirst = min(int(t.item()) for t in g0)
    # appends: every rank takes the records of its own cells, ids g_next + q
    rpos = rng.uniform(-3, 3, (Q, 3)).astype(np.float32)
    rown = np.array([lib.hrbf_hash_owner(float(x), float(y), float(z), 0.25, world) for x, y, z in rpos])
    g_next = 4 * N
    new_ids = g_next + np.nonzero(rown == rank)[0]
    q.put((rank, zred.copy(), it.numpy().copy(), int(nown.item()), gfirst, gid[mine], new_ids))
    dist.barrier()
    dist.destroy_process_group()


def test_hash_owned_map_exchange_protocol_world2():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_hash_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict()
    for _ in range(2):
        r = q.get(timeout=120); got[r[0]] = r[1:]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: one z-buffer over all surfels keyed {depth, id}
    rng = np.random.default_rng(23)
    N, P, Q = 6000, 64 * 48, 400
    rng.uniform(-3, 3, (N, 3))
    pix = rng.integers(0, P, N); depth = rng.integers(1, 120, N).astype(np.float32) * 0.01
    attr = rng.standard_normal((N, 4)).astype(np.float32)
    gid = np.sort(rng.choice(4 * N, N, replace=False)).astype(np.int64)
    EMPTY = np.iinfo(np.int64).max
    z = np.full(P, EMPTY, np.int64)
    np.minimum.at(z, pix, (depth.view(np.uint32).astype(np.int64) << 32) | gid)
    hit = z != EMPTY
    img = np.zeros((P, 4), np.float32); img[hit] = attr[np.searchsorted(gid, z[hit] & 0xFFFFFFFF)]
    for r in (0, 1):
        zr, ir, nown, gfirst, ids, new_ids = got[r]
        assert np.array_equal(zr, z) and np.array_equal(ir, img.view(np.int32))
        assert nown == int(hit.sum()) and gfirst == int(gid[0])          # every hit pixel has exactly one owner
        assert (np.diff(ids) > 0).all()
    assert np.array_equal(np.sort(np.concatenate([got[0][4], got[1][4]])), gid)
    allnew = np.sort(np.concatenate([got[0][5], got[1][5]]))
    assert np.array_equal(allnew, 4 * N + np.arange(Q))                   # every record is appended by exactly one rank
    assert min(len(got[0][4]), len(got[1][4])) > 0.35 * N                 # the hash splits the map


def test_bench_json_contract_fields():
    """static check of the bench line's keys (the values need a GPU)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "workload", "achieved", "peak",
                "frac", "traffic", "cores", "kind", "sample"):
        assert '"%s"' % key in src, key


def _bench_line(cmd):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("form", ["by hand", "driver"])
def test_bench_gpus_2_starts_two_ranks_and_times_the_sharded_design_in_children(form):
    """`bench.py --gpus N` must USE N ranks (round-3 verdict: the flag was parsed and never read).  By hand it re-executes itself under
    torch.distributed.run; under the driver's launcher it checks WORLD_SIZE.  Either way the line carries the rank count the
    communicator saw, and at N > 1 the one-sequence sharded leg runs in child processes with their own rendezvous (dry mode: gloo,
    no GPU work)."""
    import socket
    if form == "by hand":
        cmd = [sys.executable, "bench.py", "--gpus", "2", "--dry-run"]
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    line = _bench_line(cmd)
    assert line["n_gpus"] == 2 and line["ranks_observed"] == 2 and line["scaling"] == "weak"
    child = line["sharded_one_sequence"]
    assert "error" not in child, child
    assert child["n_gpus"] == 2 and child["ranks_observed"] == 2 and child["scaling"] == "strong" and child["one_sequence_child"]
    assert "config 4" in child["workload"] and "hash" in child["workload"]


def test_bench_refuses_a_launcher_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


@pytest.mark.parametrize("fault", ["fail", "hang"])
def test_a_broken_or_hanging_sharded_leg_cannot_take_the_replica_line_down(fault):
    """the one-sequence leg runs code no test box can run (RCCL with a real peer, IPC between devices): whatever happens in its child
    processes — a rank that dies (its peers then wait in a collective), a hang — the parents wait with a time limit, kill their
    child, meet again and print the replica line with an `error` entry"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HRBF_BENCH_TEST_CHILD"] = fault
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--sharded-leg-timeout", "8"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_observed"] == 2
    err = line["sharded_one_sequence"]["error"]
    assert ("time limit" in err) if fault == "hang" else ("exit code" in err or "child failed" in err), err
    assert line["sharded_one_sequence"]["wall_s_incl_setup"] < (25 if fault == "hang" else 15)      # a dead rank stops every parent at once
