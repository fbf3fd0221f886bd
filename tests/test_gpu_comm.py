"""GPU: the path's one collective -- the minimum over ranks of the packed keys -- through the C ABI's own RCCL binding
(gl_comm_*, gl_allreduce_min_keys), exercised as far as ONE GPU allows:
  * a one-rank communicator: the real ncclCommInitRank / ncclAllReduce(ncclMin, ncclUint64) calls and their ordering on the context's
    stream behind the search kernel and ahead of the unpack kernel;
  * the torch.distributed nccl route on a world of one (the fallback of bench.py): the aliased int64 view of the key buffer;
  * gl_comm_init_all refusing two ranks on one device, and attack_on_devices falling back to the host merge (several reduces per attack);
  * bench.py --gpus 2 as two processes sharing the GPU (keys through gloo): sharding, balancing, barrier and max-over-ranks of the bench.
Ranks on different GPUs are the driver's 8-GPU run; the gloo world-2/3 CPU tests (tests/test_shard_cpu.py) cover the merge logic."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_one_rank_communicator_on_the_context_stream(synth):
    import c_oracle
    import ganleaks_amd as gl
    from ganleaks_amd._lib import Comm, check
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    ctx = gl.Context.get()
    lib = ctx.lib
    uid = Comm.unique_id()
    assert isinstance(uid, bytes) and len(uid) == 128 and uid != bytes(128)
    comm = Comm(ctx, uid, 0, 1)
    assert (comm.rank, comm.nranks) == (0, 1)
    case = synth.attack_case(301, 2048, 300, 300, 32)
    q = np.concatenate([case["pos"], case["neg"]])
    od, oi, _ = c_oracle.knn_l2_u8(case["bank"], q, 64)
    # through attack(): search kernel -> all-reduce -> unpack, all queued on the context's stream
    d, i = gl.attack(q, case["bank"], batch_size=64, reduce_fn=comm.allreduce_min_keys)
    assert np.array_equal(i, oi) and np.array_equal(d, od)
    # by hand, no host synchronisation between the launches; repeated so that a mis-ordered reduce would show
    bank = Bank.from_images(case["bank"], ctx)
    qb = Bank.from_images(q, ctx)
    for _ in range(5):
        keys, _, _ = knn_keys(bank, qb, 2048)
        assert comm.allreduce_min_keys(keys) is keys
        d2, i2 = unpack_keys(ctx, keys, len(q), bank.d, "u8")
        assert np.array_equal(i2, oi) and np.array_equal(d2, od)
    # the raw ABI: bad arguments are refused, not crashed on
    p = ctypes.c_void_p
    assert lib.gl_allreduce_min_keys(p(0), p(keys.ptr), 4) == -1
    assert lib.gl_allreduce_min_keys(comm.handle, p(0), 4) == -1
    assert lib.gl_allreduce_min_keys(comm.handle, p(keys.ptr), 0) == 0
    with pytest.raises(ValueError):
        Comm(ctx, uid[:10], 0, 1)
    with pytest.raises(gl.GanLeaksError):
        Comm(ctx, uid, 3, 2)
    comm.destroy()
    comm.destroy()                       # idempotent


def test_init_all_refuses_two_ranks_on_one_device_and_attack_on_devices_falls_back(synth, golden_dir):
    import c_oracle
    import ganleaks_amd as gl
    from ganleaks_amd import shard
    from ganleaks_amd._lib import Comm, Context, GL_ERR_RCCL
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    a, b = Context(0), Context(0)
    with pytest.raises(gl.GanLeaksError) as e:
        Comm.init_all([a, b])
    assert e.value.code == GL_ERR_RCCL and "one rank per GPU" in str(e.value)
    one = Comm.init_all([a])             # a single context is a valid communicator
    assert len(one) == 1 and one[0].nranks == 1
    one[0].destroy()
    a.destroy()
    b.destroy()
    # three contexts on the one device: RCCL is out, the host merge takes over; l2-lpips with query slices makes it reduce several times
    sd = synth.dcgan_state_dict(1234)

    def make_generator(ctx):
        g = Generator(100, 3, 64, ctx)
        g.load_state_dict(sd)
        return g

    z = synth.latent(31, 700)
    ref = make_generator(gl.Context.get())
    bank = ref.generate_u8(z).numpy()
    q = np.concatenate([synth.perturb_u8(7, bank[[5, 300, 699]], 4.0), synth.lowpass_u8_images(8, 5, 64)])
    d, i = shard.attack_on_devices(q, make_generator, z, devices=[0, 0, 0], batch_size=64)
    od, oi, _ = c_oracle.knn_l2_u8(bank, q, 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od) and i[0] == 5 and i[1] == 300 and i[2] != 699
    from ganleaks_amd.lpips import LpipsModel
    lin = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    lind = {"lin%d" % k: lin["lin%d" % k] for k in range(5)}

    def make_lpips(ctx):
        return LpipsModel(ctx).load_state_dicts(synth.vgg16_state_dict(7), lind)

    model = make_lpips(gl.Context.get())
    d0, i0 = gl.attack(q, bank, distance="l2-lpips", batch_size=64, lpips=model)
    os.environ["GANLEAKS_CHUNK_GB"] = "%g" % (3.2 * 2 * int(gl.Context.get().lib.gl_lpips_search_dim(64, 64)) / 2 ** 30)     # 3 query rows per slice
    try:
        d1, i1 = shard.attack_on_devices(q, make_generator, z, devices=[0, 0], distance="l2-lpips", batch_size=64, make_lpips=make_lpips)
    finally:
        del os.environ["GANLEAKS_CHUNK_GB"]
    assert np.array_equal(i1, i0) and np.array_equal(d1, d0)


def test_torch_nccl_route_on_a_world_of_one(synth):
    """shard.allreduce_min_keys over torch.distributed's nccl backend (= RCCL): the device buffer aliased as an int64 tensor"""
    import torch
    import torch.distributed as dist
    import c_oracle
    import ganleaks_amd as gl
    from ganleaks_amd import shard
    ctx = gl.Context.get()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", ctx.device))
    try:
        case = synth.attack_case(302, 640, 20, 20, 32)
        q = np.concatenate([case["pos"], case["neg"]])
        d, i = gl.attack(q, case["bank"], batch_size=64, reduce_fn=lambda k: shard.allreduce_min_keys(k, _even_alone=True))
        od, oi, _ = c_oracle.knn_l2_u8(case["bank"], q, 64)
        assert np.array_equal(i, oi) and np.array_equal(d, od)
        assert shard.make_comm(ctx) is None           # a world of one needs no communicator
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_sharing_the_gpu():
    """bench.py --gpus 2 under the launcher, keys reduced through gloo (two ranks cannot share a device under RCCL): the JSON line of a
    sharded run, parity against the oracle included"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--collective", "gloo", "--queries", "512",
           "--bank", "8192", "--cpu-queries", "0"]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, cwd=ROOT, env=env, timeout=600).stdout.decode()
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and sum(d["config"]["shard_rows"]) == 8192 and len(d["config"]["shard_rows"]) == 2
    assert d["parity"]["idx_equal"] is True and d["parity"]["max_abs_dist_err"] == 0.0
    assert "gloo" in d["config"]["parallelism"]
