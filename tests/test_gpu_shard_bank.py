"""GPU: the sharded attack on a MATERIALISED bank -- what the reference's entry point reads from image_*.png
(attack_models/fbb.py:133-135) -- through shard.DeviceGroup / attack_on_devices(bank=...) and `fbb.py --ngpu / --devices`,
driven the way ONE GPU allows: several contexts on device 0 (RCCL refuses that, the host merge takes over).  The failure paths
(a rank failing during its setup, a rank failing after the rendezvous) must raise, not hang."""
import os

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


def _lpips_factory(synth, golden_dir):
    from ganleaks_amd.lpips import LpipsModel
    lin = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    lind = {"lin%d" % k: lin["lin%d" % k] for k in range(5)}
    vgg = synth.vgg16_state_dict(7)
    return lambda ctx: LpipsModel(ctx).load_state_dicts(vgg, lind)


def test_materialised_bank_sharded_equals_single_device(synth, golden_dir):
    import c_oracle
    import ganleaks_amd as gl
    from ganleaks_amd import shard
    case = synth.attack_case(411, 1000, 40, 40, 32)            # 1000 rows: n_eff = 960, shards of 320 (not multiples of 64 apart from 0)
    q = np.concatenate([case["pos"], case["neg"]])
    od, oi, _ = c_oracle.knn_l2_u8(case["bank"], q, 64)
    for devices in ([0], [0, 0], [0, 0, 0]):
        d, i = shard.attack_on_devices(q, bank=case["bank"], devices=devices, batch_size=64)
        assert np.array_equal(i, oi) and np.array_equal(d, od), devices
    assert oi.max() < 960
    # float images on the lattice and a DeviceArray bank are accepted too; weights change the cut, never the result
    f = (2.0 * (case["bank"] / 255.0) - 1.0).astype(np.float32)
    d, i = shard.attack_on_devices(q, bank=gl.Context.get().to_device(f), devices=[0, 0], batch_size=64, weights=[1.0, 2.5])
    assert np.array_equal(i, oi) and np.array_equal(d, od)
    # more ranks than full batches: empty shards take part in the reduction
    d, i = shard.attack_on_devices(q, bank=case["bank"][:130], devices=[0, 0, 0], batch_size=64)
    od2, oi2, _ = c_oracle.knn_l2_u8(case["bank"][:130], q, 64)
    assert np.array_equal(i, oi2) and np.array_equal(d, od2)
    # l2-lpips: search rows per shard, indices global; a group kept across banks featurises the queries once per context
    make_lpips = _lpips_factory(synth, golden_dir)
    model = make_lpips(gl.Context.get())
    bank64 = synth.lowpass_u8_images(21, 200, 64)
    q64 = np.concatenate([synth.perturb_u8(22, bank64[[3, 77, 191]], 3.0), synth.lowpass_u8_images(23, 4, 64)])
    d0, i0 = gl.attack(q64, bank64, distance="l2-lpips", batch_size=64, lpips=model)
    with shard.DeviceGroup([0, 0]) as group:
        assert group.collective == "host-merge"
        d1, i1 = group.attack(q64, bank=bank64, distance="l2-lpips", batch_size=64, make_lpips=make_lpips)
        rows = [qq[1] for qq in group._queries]
        d2, i2 = group.attack(q64, bank=bank64[::-1].copy(), distance="l2-lpips", batch_size=64, make_lpips=make_lpips)
        assert all(a[1] is b for a, b in zip(group._queries, rows))            # the query rows of the first call were reused
    assert np.array_equal(i1, i0) and np.array_equal(d1, d0) and list(i0[:3]) == [3, 77, 191]
    d3, i3 = gl.attack(q64, bank64[::-1].copy(), distance="l2-lpips", batch_size=64, lpips=model)
    assert np.array_equal(i2, i3) and np.array_equal(d2, d3)


def test_a_failing_rank_raises_instead_of_hanging(synth):
    import ganleaks_amd as gl
    from ganleaks_amd import shard
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    sd = synth.dcgan_state_dict(1234)
    z = synth.latent(5, 256)
    q = synth.lowpass_u8_images(9, 6, 64)
    calls = []

    def make_generator(ctx):
        calls.append(ctx)
        if len(calls) == 2:
            raise RuntimeError("rank setup failed on purpose")
        g = Generator(100, 3, 64, ctx)
        g.load_state_dict(sd)
        return g

    with pytest.raises(RuntimeError, match="on purpose"):          # during setup: nobody has queued a collective
        shard.attack_on_devices(q, make_generator, z, devices=[0, 0, 0], batch_size=64)

    class Flaky:
        """a generator whose second chunk fails: the rank dies after the rendezvous, while the others wait in the merge"""
        def __init__(self, ctx, fail):
            self.ctx, self.fail = ctx, fail
            self.g = Generator(100, 3, 64, ctx)
            self.g.load_state_dict(sd)

        def generate_u8(self, zz, **kw):
            if self.fail:
                raise MemoryError("bank chunk failed on purpose")
            return self.g.generate_u8(zz, **kw)

    made = []

    def make_flaky(ctx):
        made.append(ctx)
        return Flaky(ctx, len(made) == 1)

    with pytest.raises(MemoryError, match="on purpose"):
        shard.attack_on_devices(q, make_flaky, z, devices=[0, 0], batch_size=64)
    # the process is still healthy
    d, i = shard.attack_on_devices(q, lambda ctx: Flaky(ctx, False), z, devices=[0, 0], batch_size=64)
    assert len(d) == 6 and i.max() < 256
    with pytest.raises(ValueError):
        shard.attack_on_devices(q, devices=[0])                      # neither a generator nor a bank
    with pytest.raises(ValueError):
        shard.attack_on_devices(q, make_flaky, z, devices=[0], bank=np.zeros((64, 3, 64, 64), np.uint8))


def test_fbb_ngpu_flag_matches_single_device(tmp_path, monkeypatch, synth, golden_dir):
    import PIL.Image
    from ganleaks_amd import lpips
    from ganleaks_amd.attack_models import fbb
    bank = synth.lowpass_u8_images(31, 140, 32)
    pos = synth.perturb_u8(32, bank[[1, 50, 127, 139]], 3.0)
    neg = synth.lowpass_u8_images(33, 5, 32)
    for name, imgs in (("syn", bank), ("pos", pos), ("neg", neg)):
        os.makedirs(tmp_path / name)
        for k, im in enumerate(imgs):
            PIL.Image.fromarray(im.transpose(1, 2, 0)).save(tmp_path / name / ("image_%d.png" % k))
    monkeypatch.chdir(tmp_path)
    base = ["--syn_data_path", str(tmp_path / "syn"), "--pos_data_dir", str(tmp_path / "pos"), "--neg_data_dir", str(tmp_path / "neg"),
            "--resolution", "32", "--BATCH_SIZE", "64"]
    factory = _lpips_factory(synth, golden_dir)
    import ganleaks_amd as gl
    lpips.set_default_model(factory(gl.Context.get()))
    lpips.set_default_factory(factory)
    try:
        for distance in ("l2", "l2-lpips"):
            fbb.main(fbb.parse_arguments(base + ["--exp_name", "one_" + distance, "--distance", distance]))
            a = fbb.parse_arguments(base + ["--exp_name", "two_" + distance, "--distance", distance, "--devices", "0,0,0"])
            assert fbb.shard_devices(a) == [0, 0, 0]
            fbb.main(a)
            for f in ("pos_loss", "neg_loss", "pos_nn_idx", "neg_nn_idx", "pos_idx"):
                one = np.load(tmp_path / "fbb_attack" / ("one_" + distance) / (f + ".npy"))
                two = np.load(tmp_path / "fbb_attack" / ("two_" + distance) / (f + ".npy"))
                assert np.array_equal(one, two), (distance, f)
            assert (tmp_path / "fbb_attack" / ("two_" + distance) / "0pos.png").exists()
    finally:
        lpips.set_default_model(None)
        lpips.set_default_factory(None)
    assert fbb.shard_devices(fbb.parse_arguments(base + ["--ngpu", "4"])) == [0, 1, 2, 3]
    assert fbb.shard_devices(fbb.parse_arguments(base)) is None


def test_sharded_query_features_equal_replicated_ones(synth, golden_dir):
    """lpips.features_sharded: rank r featurises queries [r per, (r + 1) per) into its block, all-gathers rows and norms.  One GPU cannot run
    several RCCL ranks, so the gather is played by a stand-in that fills the other ranks' blocks the way they would (each from its own
    slice); what is checked is the block arithmetic -- ragged last rank, ranks without queries -- and that the result is bit for bit what
    every rank gets by featurising all queries itself.  The real collective runs on a one-rank communicator (in-place copy)."""
    import ganleaks_amd as gl
    from ganleaks_amd import lpips as lp
    from ganleaks_amd._lib import Comm
    from ganleaks_amd.attack import prepare_queries
    ctx = gl.Context.get()
    model = _lpips_factory(synth, golden_dir)(ctx)
    q = synth.lowpass_u8_images(77, 43, 32)
    full = model.features(q, role="query")
    K1 = full.K

    class Stand_in:
        def __init__(self, rank, nranks):
            self.rank, self.nranks, self.calls = rank, nranks, 0

        def allgather_rows(self, buf, bytes_per_rank):
            # the other ranks' blocks, as they would have produced them; rows gathered first, norms second
            per = -(-len(q) // self.nranks)
            for r in range(self.nranks):
                lo, hi = min(r * per, len(q)), min((r + 1) * per, len(q))
                if r == self.rank or hi <= lo:
                    continue
                other = model.features(q[lo:hi], role="query")
                src = other.V if self.calls == 0 else other.norms
                host = src.numpy()[:hi - lo]
                ctx.to_device(host)          # (exercise the upload path)
                from ganleaks_amd._lib import check
                import ctypes
                check(ctx.lib.gl_memcpy_h2d(ctx.handle, ctypes.c_void_p(buf.ptr + r * bytes_per_rank), host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
            self.calls += 1
            return buf

    for world in (2, 3, 8):
        for rank in (0, world - 1):
            fb = lp.features_sharded(model, q, Stand_in(rank, world))
            assert fb.n == len(q) and fb.K == K1 and fb.fmt == "lattice" and fb.role == "query"
            assert np.array_equal(fb.rows_numpy(), full.rows_numpy()) and np.array_equal(fb.norms.numpy()[:len(q)], full.norms.numpy()), (world, rank)
    # more ranks than queries would leave ranks empty: prepare_queries keeps such small sets replicated
    one = Comm(ctx, Comm.unique_id(), 0, 1)
    fb1 = lp.features_sharded(model, q, one)                       # the real ncclAllGather, one rank: in-place
    ctx.sync()
    assert np.array_equal(fb1.rows_numpy(), full.rows_numpy()) and np.array_equal(fb1.norms.numpy()[:len(q)], full.norms.numpy())
    assert prepare_queries(q, "l2-lpips", ctx, model, comm=one).V.shape[0] == len(q)     # one rank: the replicated form
    one.destroy()
    with pytest.raises(ValueError):
        lp.features_sharded(model, q.astype(np.float32), Stand_in(0, 2))


def test_fbb_sweep_on_a_device_group(tmp_path, monkeypatch, synth):
    """fbb.py --hyperparameter_search --devices ...: one DeviceGroup serves every bank of the sweep (attack_models/fbb.py:114-123 walks the
    sub-folders), the results equal the single-device sweep's file for file"""
    import PIL.Image
    from ganleaks_amd.attack_models import fbb
    banks = {"lr_0.1": synth.lowpass_u8_images(41, 130, 16), "lr_0.2": synth.lowpass_u8_images(42, 70, 16)}
    pos = synth.perturb_u8(43, banks["lr_0.1"][[0, 64, 127]], 2.0)
    neg = synth.lowpass_u8_images(44, 4, 16)
    sweep = tmp_path / "png_images" / "lr"
    for name, imgs in list(banks.items()) + [("pos", pos), ("neg", neg)]:
        d = (sweep / name) if name in banks else (tmp_path / name)
        os.makedirs(d)
        for k, im in enumerate(imgs):
            PIL.Image.fromarray(im.transpose(1, 2, 0)).save(d / ("image_%d.png" % k))
    monkeypatch.chdir(tmp_path)
    base = ["--syn_data_path", str(sweep), "--pos_data_dir", str(tmp_path / "pos"), "--neg_data_dir", str(tmp_path / "neg"), "--resolution", "16",
            "--BATCH_SIZE", "64", "--distance", "l2", "--hyperparameter_search", "1"]
    one = fbb.main(fbb.parse_arguments(base + ["--exp_name", "single"]))
    two = fbb.main(fbb.parse_arguments(base + ["--exp_name", "sharded", "--devices", "0,0"]))
    assert len(one) == len(two) == 2
    for (d1, p1, n1, pi1, ni1), (d2, p2, n2, pi2, ni2) in zip(sorted(one), sorted(two)):
        assert os.path.basename(d1) == os.path.basename(d2)
        assert np.array_equal(p1, p2) and np.array_equal(n1, n2) and np.array_equal(pi1, pi2) and np.array_equal(ni1, ni2)
    lr1 = [r for r in two if r[0].endswith("lr_0.1")][0]
    # indices refer to the sorted() path order of the bank files (image_10 < image_2), as in the reference (utils.py:57)
    order = sorted(range(130), key=lambda k: str(sweep / "lr_0.1" / ("image_%d.png" % k)))[:128]
    porder = sorted(range(3), key=lambda k: str(tmp_path / "pos" / ("image_%d.png" % k)))
    want = [order.index([0, 64, 127][k]) for k in porder]
    assert [int(v) for v in lr1[3][:, 0]] == want


def _gloo_rank(rank, world, port, out_dir):
    """one process per rank sharing the GPU: its shard of a MATERIALISED bank (rows [bounds[r], bounds[r+1]) of the host array, index_base =
    bounds[r]) searched for all queries, keys min-reduced through torch.distributed (gloo: ranks on one device cannot form an RCCL communicator)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    import ganleaks_amd as gl
    from ganleaks_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    case = gl.synth.attack_case(512, 1000, 30, 30, 32)
    q = np.concatenate([case["pos"], case["neg"]])
    bounds = shard.shard_bounds(len(case["bank"]), 64, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    d, i = gl.attack(q, case["bank"][lo:hi], batch_size=64, index_base=lo, reduce_fn=lambda k: shard.allreduce_min_keys(k))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gloo_%d.npz" % world), d=d, i=i)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_materialised_bank_one_process_per_rank_over_gloo(world, tmp_path, synth):
    import socket
    import torch.multiprocessing as mp
    import c_oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / ("gloo_%d.npz" % world))
    case = synth.attack_case(512, 1000, 30, 30, 32)
    od, oi, _ = c_oracle.knn_l2_u8(case["bank"], np.concatenate([case["pos"], case["neg"]]), 64)
    assert np.array_equal(got["i"], oi) and np.array_equal(got["d"], od) and oi.max() < 960
