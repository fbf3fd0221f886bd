"""CPU-only: bank sharding + min-reduce of packed keys (the path's one exchange step), rehearsed
with gloo at world_size 2 and 3.  The per-shard search is done by the oracle here (no GPU); the
GPU test tests/test_gpu_knn.py::test_shard_invariance covers the kernel side of the same contract."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_bank, batch_size, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import ganleaks_amd
    from ganleaks_amd import shard
    import c_oracle
    import oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    case = ganleaks_amd.synth.attack_case(71, n_bank, 12, 12, 16)
    q = np.concatenate([case["pos"], case["neg"]])
    bounds = shard.shard_bounds(n_bank, batch_size, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    if hi > lo:
        _, idx, ssd = c_oracle.knn_l2_u8(case["bank"][lo:hi], q, 1)
        keys = oracle.pack_key(ssd, idx + lo).view(np.uint64)
    else:
        keys = np.full(len(q), np.iinfo(np.uint64).max >> 1, np.uint64)   # empty shard: neutral element (int64 max)
    merged = shard.allreduce_min_keys_host(keys)
    if rank == 0:
        np.save(os.path.join(out_dir, "merged_%d.npy" % world), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_min_reduce_equals_single_device(world, tmp_path, oracle, synth):
    import torch.multiprocessing as mp
    n_bank, bs = 1000, 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_bank, bs, str(tmp_path)), nprocs=world, join=True)
    merged = np.load(tmp_path / ("merged_%d.npy" % world))
    case = synth.attack_case(71, n_bank, 12, 12, 16)
    q = np.concatenate([case["pos"], case["neg"]])
    dist, idx, ssd = oracle.knn_l2_u8(case["bank"], q, bs)
    s, i = oracle.unpack_key(merged.view(np.int64))
    assert np.array_equal(i, idx) and np.array_equal(s, ssd)
    assert i.max() < 960           # global truncation happened before the split


def test_shard_bounds_global_truncation():
    sys.path.insert(0, ROOT)
    from ganleaks_amd import shard
    assert shard.shard_bounds(100000, 64, 8) == [99968 * r // 8 for r in range(9)]
    assert shard.shard_bounds(100000, 64, 8)[-1] == 99968
    assert shard.shard_bounds(1000, 64, 1) == [0, 960]
    assert shard.shard_bounds(10, 64, 4) == [0, 0, 0, 0, 0]
    b = shard.shard_bounds(1000, 30, 7)
    assert b[0] == 0 and b[-1] == 990 and all(b[k] <= b[k + 1] for k in range(7))
    k = [np.array([5, 9], np.uint64), np.array([7, 2], np.uint64), np.array([6, 3], np.uint64)]
    assert shard.merge_keys_host(k).tolist() == [5, 2]


def test_weighted_bounds():
    """shards proportional to measured rank speeds: contiguous, cover [0, n_eff), interior cuts on multiples of the batch size"""
    import ganleaks_amd  # noqa: F401
    from ganleaks_amd import shard
    b = shard.weighted_bounds(99968, [1.0] * 8, 64)
    assert b[0] == 0 and b[-1] == 99968 and all(x % 64 == 0 for x in b) and max(np.diff(b)) - min(np.diff(b)) <= 64
    w = [1 / 8.3, 1 / 7.5, 1 / 8.0, 1 / 7.9]
    b = shard.weighted_bounds(99968, w, 64)
    sizes = np.diff(b)
    assert b[0] == 0 and b[-1] == 99968 and (sizes > 0).all() and all(x % 64 == 0 for x in b[:-1])
    assert np.abs(sizes / 99968 - np.array(w) / sum(w)).max() < 1e-3
    assert np.argmax(sizes) == 1 and np.argmin(sizes) == 0
    assert shard.weighted_bounds(128, [1, 1, 1], 64) in ([0, 64, 64, 128], [0, 64, 128, 128], [0, 0, 64, 128])
    for bad in ([], [1, 0], [1, float("nan")]):
        with pytest.raises(ValueError):
            shard.weighted_bounds(100, bad)


def test_host_merge_survives_repeated_calls_with_skewed_threads():
    """shard.HostMerge (the in-process fallback of attack_on_devices): 3 threads, 40 merges in a row, every thread sleeping at a different point
    -- a fast rank's next deposit must not replace what a slow rank is still merging (one rendezvous was not enough)"""
    import threading
    import time
    import ganleaks_amd  # noqa: F401
    from ganleaks_amd import shard
    world, rounds, nq = 3, 40, 257
    rng = np.random.default_rng(3)
    data = rng.integers(0, 2 ** 62, size=(rounds, world, nq), dtype=np.uint64)
    hm = shard.HostMerge(world)
    out = [[None] * rounds for _ in range(world)]

    def work(rank):
        for r in range(rounds):
            if (r + rank) % 3 == 0:
                time.sleep(0.002)                      # late to deposit
            m = hm.merge(rank, data[r, rank])
            if (r + rank) % 3 == 1:
                time.sleep(0.002)                      # slow to use the result while the others race ahead
            out[rank][r] = m.copy()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r in range(rounds):
        want = data[r].min(axis=0)
        for rank in range(world):
            assert np.array_equal(out[rank][r], want), (r, rank)


def _make_comm_worker(rank, world, port, scenario, out_dir):
    """make_comm must leave every rank the same way: rank 0 failing to draw the id, or one rank failing to join, raises
    GanLeaksError on ALL ranks, and the process group's next collective still matches."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import ganleaks_amd  # noqa: F401
    from ganleaks_amd import _lib, shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    aborted = []

    class FakeComm:
        def __init__(self, ctx, uid, r, n):
            if scenario == "join" and r == 1:
                raise _lib.GanLeaksError(_lib.GL_ERR_RCCL, "ncclCommInitRank failed on purpose")
            assert uid == b"u" * 128

        def abort(self):
            aborted.append(rank)

        @staticmethod
        def unique_id():
            if scenario == "id":
                raise _lib.GanLeaksError(_lib.GL_ERR_RCCL, "RCCL is not available: on purpose")
            return b"u" * 128

    _lib.Comm = FakeComm
    outcome = "ok"
    try:
        c = shard.make_comm(None)
        assert isinstance(c, FakeComm)
    except _lib.GanLeaksError as e:
        assert e.code == _lib.GL_ERR_RCCL
        outcome = "rccl-error"
    # what bench.py does next: a vote over the launcher's group -- must not hang or mismatch
    t = torch.tensor([1 if outcome == "ok" else 0], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    with open(os.path.join(out_dir, "%s_%d.txt" % (scenario, rank)), "w") as f:
        f.write("%s %d %d" % (outcome, int(t.item()), len(aborted)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["id", "join", "fine"])
def test_make_comm_fails_on_every_rank_or_none(scenario, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_make_comm_worker, args=(world, _free_port(), scenario, str(tmp_path)), nprocs=world, join=True)
    got = [open(tmp_path / ("%s_%d.txt" % (scenario, r))).read().split() for r in range(world)]
    if scenario == "fine":
        assert got == [["ok", "1", "0"]] * 2
    else:
        assert [g[:2] for g in got] == [["rccl-error", "0"]] * 2
        if scenario == "join":
            assert got[0][2] == "1" and got[1][2] == "0"        # the rank that had joined dropped its communicator
