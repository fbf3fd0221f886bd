"""GPU parity tests of the L2 nearest-neighbour path: HIP (through the C ABI) vs the oracle and
vs golden vectors made by the reference's custom_knn.  Bit-exact indices; distances equal to the
oracle's fp32 value bit for bit and within 1e-6 of the reference's."""
import ctypes
import os
import types

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    return ganleaks_amd


@pytest.fixture(scope="module")
def coracle():
    import c_oracle
    return c_oracle


@pytest.mark.parametrize("name", ["knn_c1", "knn_b30", "knn_res32", "knn_res16"])
def test_attack_matches_reference_goldens(name, gl, synth, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    case = synth.attack_case(int(g["seed"]), int(g["n_bank"]), int(g["n_pos"]), int(g["n_neg"]), int(g["res"]))
    bs = int(g["batch_size"])
    for kind in ("pos", "neg"):
        dist, idx = gl.attack(case[kind], case["bank"], distance="l2", batch_size=bs)
        assert dist.dtype == np.float32 and idx.dtype == np.int64
        assert np.array_equal(idx, g[kind + "_idx"])
        np.testing.assert_allclose(dist.astype(np.float64), g[kind + "_dist"], rtol=0, atol=1e-6)
        od, oi, _ = oracle.knn_l2_u8(case["bank"], case[kind], bs) if int(g["n_bank"]) <= 300 else (None, None, None)
        if od is not None:
            assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_float_lattice_inputs_and_custom_knn(gl, synth, oracle, golden_dir):
    """the drop-in signature: float NCHW tensors as fbb.main builds them, one query at a time"""
    from ganleaks_amd.attack_models.fbb import custom_knn
    from ganleaks_amd.attack_models.utils import Loss
    g = np.load(os.path.join(golden_dir, "knn_res32.npz"))
    case = synth.attack_case(int(g["seed"]), int(g["n_bank"]), int(g["n_pos"]), int(g["n_neg"]), int(g["res"]))
    bank_f = oracle.dequantize_u8(case["bank"])
    q_f = oracle.dequantize_u8(case["pos"])
    loss = Loss("l2")
    args = types.SimpleNamespace(BATCH_SIZE=64)
    for k in range(6):
        d, i = custom_knn(bank_f, q_f[k], loss, args)
        assert isinstance(d, float) and isinstance(i, int)
        assert i == int(g["pos_idx"][k]) and abs(d - float(g["pos_dist"][k])) < 1e-6
    # torch CPU tensors, as the reference passes them
    import torch
    d, i = custom_knn(torch.from_numpy(bank_f), torch.from_numpy(q_f[7]), loss, args)
    assert i == int(g["pos_idx"][7])
    # off-lattice floats are never rounded to the lattice: they take the fixed-order fp32 path
    import c_oracle
    bad = bank_f.copy()
    bad[3, 0, 0, 0] += 1e-3
    d, i = gl.attack(q_f[:5], bad, batch_size=64)
    od, oi = c_oracle.knn_l2_f32(bad, q_f[:5], 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od)


def test_loss_forward_vector(gl, synth, oracle):
    from ganleaks_amd.attack_models.utils import Loss
    case = synth.attack_case(31, 70, 3, 3, 32)
    loss = Loss("l2")
    x_hat = case["bank"][:64]
    v = loss(x_hat, case["pos"][:1])
    ref = (oracle.ssd_u8(x_hat, case["pos"][0]).astype(np.float64) * oracle.l2_scale(3 * 32 * 32)).astype(np.float32)
    assert np.array_equal(v, ref)
    assert np.array_equal(loss.vec_loss, ref) and loss.loss_lpips == 0.0
    v2 = loss(x_hat[:3], case["pos"][:3])       # pairwise form
    for k in range(3):
        assert v2[k] == np.float32(float(oracle.ssd_u8(x_hat[k:k + 1], case["pos"][k])[0]) * oracle.l2_scale(3072))
    with pytest.raises(gl.GanLeaksError):
        loss(x_hat[:3], case["pos"][:2])


def test_ties_truncation_and_empty(gl, synth, golden_dir):
    g = np.load(os.path.join(golden_dir, "knn_ties.npz"))
    base = synth.lowpass_u8_images(77, 1000, 64)
    bank = base.copy()
    bank[700] = bank[5]
    bank[300] = bank[5]
    queries = np.stack([bank[5], bank[990], synth.perturb_u8(1, bank[700:701], 3.0)[0], bank[959], bank[0]])
    dist, idx = gl.attack(queries, bank, batch_size=64)
    assert np.array_equal(idx, g["idx"])
    assert idx[0] == 5 and dist[0] == 0.0 and idx[1] < 960 and idx[3] == 959
    np.testing.assert_allclose(dist.astype(np.float64), g["dist"], rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        gl.attack(queries, bank[:10], batch_size=64)
    d0, i0 = gl.attack(queries[:0], bank, batch_size=64)
    assert len(d0) == 0 and len(i0) == 0


@pytest.mark.parametrize("shape", [(3, 10, 10), (1, 28, 28), (3, 16, 16), (1071,)])
@pytest.mark.parametrize("nq,nb", [(1, 64), (129, 257), (200, 1000)])
def test_ragged_shapes_vs_oracle(shape, nq, nb, gl, coracle):
    rng = np.random.default_rng(nq * 1000 + nb + len(shape))
    bank = rng.integers(0, 256, size=(nb,) + shape, dtype=np.uint8)
    q = rng.integers(0, 256, size=(nq,) + shape, dtype=np.uint8)
    q[0] = bank[min(nb - 1, 63)]
    for bs in (64, 30, 1):
        if nb < bs:
            continue
        dist, idx = gl.attack(q, bank, batch_size=bs)
        od, oi, _ = coracle.knn_l2_u8(bank, q, bs)
        assert np.array_equal(idx, oi)
        assert np.array_equal(dist, od)


def test_extreme_values_int32_exact(gl, coracle):
    """all-0 vs all-255 images: S = 255^2 * D, the largest value the int32 path must carry"""
    d = (3, 64, 64)
    bank = np.zeros((128,) + d, np.uint8)
    bank[64:] = 255
    bank[100, 0, 0, 0] = 254
    q = np.stack([np.full(d, 255, np.uint8), np.zeros(d, np.uint8)])
    dist, idx = gl.attack(q, bank[:64], batch_size=64)        # only the all-zero half
    assert idx.tolist() == [0, 0]
    assert dist[0] == np.float32(255.0 ** 2 * 12288 * (4.0 / (65025.0 * 12288))) and dist[1] == 0
    dist, idx = gl.attack(q, bank, batch_size=64)
    od, oi, _ = coracle.knn_l2_u8(bank, q, 64)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


@pytest.mark.parametrize("shape", [(3, 128, 128), (3, 256, 256), (262143,)])
def test_large_images_exact(shape, gl, coracle):
    """d > 32768: S no longer fits 31 bits.  3x128x128 runs modulo 2^32 (S < 2^32, key shift 31); 3x256x256 and the largest
    supported d flush the int32 accumulators into 64-bit totals every 64 KiB of K (key shift 29).  All-0 against all-255 rows
    give the largest S = 255^2 d."""
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    from ganleaks_amd.attack_models.utils import Loss
    rng = np.random.default_rng(shape[-1])
    bank = rng.integers(0, 256, size=(70,) + shape, dtype=np.uint8)
    bank[3] = 0
    bank[5] = 255
    q = np.stack([np.full(shape, 255, np.uint8), np.zeros(shape, np.uint8), bank[40], rng.integers(0, 256, size=shape, dtype=np.uint8),
                  rng.integers(100, 140, size=shape, dtype=np.uint8)])
    dist, idx = gl.attack(q, bank, batch_size=32)
    od, oi, ssd = coracle.knn_l2_u8(bank, q, 32)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert idx[:3].tolist() == [5, 3, 40] and dist[:3].tolist() == [0, 0, 0]
    # the far extreme: only rows of the opposite colour in the bank
    d2, i2 = gl.attack(q[:2], np.zeros((16,) + shape, np.uint8), batch_size=16)
    assert i2.tolist() == [0, 0] and d2[0] == np.float32(4.0) and d2[1] == 0
    # Loss('l2').forward rows (gl_l2_rows_u8) at this d
    v = Loss("l2")(bank[:8], q[:1])
    ref = (coracle.ssd_row_u8(bank[:8], q[0]).astype(np.float64) * (4.0 / (65025.0 * q[0].size))).astype(np.float32)
    assert np.array_equal(np.asarray(v, np.float32), ref)
    # shards carry global indices through the narrower index field
    ctx = gl.Context.get()
    qb = Bank.from_images(q, ctx)
    keys = None
    for lo, hi in ((32, 64), (0, 32)):
        keys, _, _ = knn_keys(Bank.from_images(bank[lo:hi], ctx, index_base=lo), qb, keys=keys)
    d3, i3 = unpack_keys(ctx, keys, qb.n, qb.d)
    assert np.array_equal(i3, oi) and np.array_equal(d3, od)
    if shape == (3, 256, 256):
        with pytest.raises(gl.GanLeaksError):
            knn_keys(Bank.from_images(bank[:32], ctx, index_base=(1 << 29) - 8), qb)
    with pytest.raises(gl.GanLeaksError):
        Bank.from_images(np.zeros((2, 262144), np.uint8), ctx)


def test_integer_tables_take_the_exact_path(gl, coracle):
    """binary / count rows given as floats (medGAN's thresholded samples, medgan/train.py:306-312) are searched on the int8 matrix
    cores; the result equals the fixed-order fp32 path bit for bit (all sums are exact integers) and the C oracle's S / F"""
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    rng = np.random.default_rng(11)
    F = 1071
    for hi in (2, 12):                                                   # {0,1} and small counts
        bank = rng.integers(0, hi, size=(300, F)).astype(np.float32)
        q = rng.integers(0, hi, size=(37, F)).astype(np.float32)
        q[3] = bank[170]
        ctx = gl.Context.get()
        assert Bank.from_images(bank, ctx).kind == "int"
        d, i = gl.attack(q, bank, batch_size=64)
        _, oi, ssd = coracle.knn_l2_u8(bank.astype(np.uint8), q.astype(np.uint8), 64)
        assert np.array_equal(i, oi) and np.array_equal(d, (ssd.astype(np.float64) / F).astype(np.float32))
        assert i[3] == 170 and d[3] == 0 and i.max() < 256
        kf, qf, kind = knn_keys(Bank.from_images(bank[:256], ctx, force_kind="f32"), Bank.from_images(q, ctx, force_kind="f32"))
        df, if_ = unpack_keys(ctx, kf, 37, F, kind)
        assert kind == "f32" and np.array_equal(if_, i) and np.array_equal(df, d)
        ds, is_ = gl.attack(q, bank, batch_size=64, chunk_bytes=50 * 2 * F)          # streamed, same path
        assert np.array_equal(is_, i) and np.array_equal(ds, d)
    # one non-integer query value: everything moves to the fp32 path, same answer up to fp32 rounding of that row
    q2 = q.copy()
    q2[0, 0] += 0.25
    d2, i2 = gl.attack(q2, bank, batch_size=64)
    assert np.array_equal(i2[1:], i[1:]) and np.array_equal(d2[1:], d[1:])
    with pytest.raises(ValueError):
        Bank.from_images(q2, gl.Context.get(), force_kind="int")


def test_shard_invariance(gl, synth, coracle):
    """any split of the bank into index-based shards gives bit-identical (dist, idx)"""
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    case = synth.attack_case(41, 1500, 40, 40, 32)
    q = np.concatenate([case["pos"], case["neg"]])
    n_eff = (1500 // 64) * 64
    ref_d, ref_i = gl.attack(q, case["bank"], batch_size=64)
    od, oi, _ = coracle.knn_l2_u8(case["bank"], q, 64)
    assert np.array_equal(ref_i, oi) and np.array_equal(ref_d, od)
    ctx = gl.Context.get()
    qb = Bank.from_images(q, ctx)
    for world in (2, 3, 8):
        bounds = [n_eff * r // world for r in range(world + 1)]
        keys = None
        for r in reversed(range(world)):      # order must not matter
            shard = Bank.from_images(case["bank"][bounds[r]:bounds[r + 1]], ctx, index_base=bounds[r])
            keys, _, _ = knn_keys(shard, qb, keys=keys)
        d, i = unpack_keys(ctx, keys, qb.n, qb.d)
        assert np.array_equal(i, ref_i) and np.array_equal(d, ref_d)


@pytest.mark.parametrize("shape", [(3, 16, 16), (3, 10, 10), (1071,), (5,)])
def test_float_path_vs_oracle(shape, gl, coracle):
    """arbitrary fp32 images: device result equals the oracle's fixed-order fp32 chain bit for bit"""
    from ganleaks_amd.attack_models.utils import Loss
    rng = np.random.default_rng(len(shape) + shape[0])
    bank = rng.uniform(-1, 1, size=(333,) + shape).astype(np.float32)
    q = rng.uniform(-1, 1, size=(70,) + shape).astype(np.float32)
    q[3] = bank[17]
    bank[200] = bank[40]
    q[4] = bank[200]
    for bs in (64, 30):
        d, i = gl.attack(q, bank, batch_size=bs)
        od, oi = coracle.knn_l2_f32(bank, q, bs)
        assert np.array_equal(i, oi) and np.array_equal(d, od)
        assert i[3] == 17 and d[3] == 0 and i[4] == 40
    v = Loss("l2")(bank[:64], q[:1])
    ref = np.array([coracle.l2_pair_f32(q[0], bank[k]) for k in range(64)], np.float32)
    assert np.array_equal(v, ref)
    # mixed: u8 bank, off-lattice float queries
    ub = rng.integers(0, 256, size=(128,) + shape, dtype=np.uint8)
    d, i = gl.attack(q[:9], ub, batch_size=64)
    fb = (2.0 * (ub / 255.0) - 1.0).astype(np.float32)
    od, oi = coracle.knn_l2_f32(fb, q[:9], 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od)
    # shards of a float bank merge exactly like the integer path
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    ctx = gl.Context.get()
    keys = None
    qb = Bank.from_images(q, ctx, force_kind="f32")
    for lo, hi in ((192, 320), (0, 100), (100, 192)):
        keys, _, kind = knn_keys(Bank.from_images(bank[lo:hi], ctx, index_base=lo, force_kind="f32"), qb, keys=keys)
    d, i = unpack_keys(ctx, keys, qb.n, qb.d, kind)
    od, oi = coracle.knn_l2_f32(bank, q, 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od)


@pytest.mark.parametrize("shape", [(3, 32, 32), (1071,), (37,)])
def test_float_rows_on_the_matrix_cores(shape, gl, coracle):
    """float_path='mfma': |y|^2 + |x|^2 - 2 y.x with split-fp16 operands and a per-row power-of-two scale.  Not bit-reproducible
    against the oracle's fixed fp32 order, so: same indices where the top-2 gap exceeds the error, distances within
    4e-6 * mean(x^2) (fp32 accumulation of the dot product, exposed by the cancellation) + 1e-6 relative."""
    rng = np.random.default_rng(shape[-1])
    bank = rng.uniform(-1, 1, size=(700,) + shape).astype(np.float32)
    q = rng.uniform(-1, 1, size=(150,) + shape).astype(np.float32)
    # rows of very different magnitude (tabular counts next to fractions), an exact copy, an all-zero row
    bank[100:200] *= 300.0
    q[10:20] *= 300.0
    bank[5] = 0.0
    q[3] = bank[17]
    q[4] = bank[150]
    de, ie = gl.attack(q, bank, batch_size=64, float_path="exact")
    dm, im = gl.attack(q, bank, batch_size=64, float_path="mfma")
    od, oi = coracle.knn_l2_f32(bank, q, 64)
    assert np.array_equal(ie, oi) and np.array_equal(de, od)
    assert np.array_equal(im, ie)
    scale = np.maximum(1.0, (q.reshape(len(q), -1) ** 2).mean(axis=1))          # error scales with |x|^2 / d
    assert (np.abs(dm - de) <= 4e-6 * scale + 1e-6 * de).all(), np.abs(dm - de).max()
    assert dm[3] <= 2e-6 and dm[4] <= 2e-6 * 300 ** 2 and im[3] == 17 and im[4] == 150
    ds, is_ = gl.attack(q, bank, batch_size=64, float_path="mfma", chunk_bytes=100 * 4 * bank[0].size)     # streamed, same kernel
    assert np.array_equal(is_, im) and np.array_equal(ds, dm)


def test_medium_vs_c_oracle(gl, synth, coracle):
    case = synth.attack_case(51, 6000, 150, 150, 64)
    q = np.concatenate([case["pos"], case["neg"]])
    dist, idx = gl.attack(q, case["bank"], batch_size=64)
    od, oi, _ = coracle.knn_l2_u8(case["bank"], q, 64)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_host_one_call_abi(gl, synth, coracle):
    """gl_fbb_knn_l2_host straight through ctypes, as a foreign host would call it"""
    from ganleaks_amd import _lib
    lib = _lib.load()
    ctx = gl.Context.get()
    case = synth.attack_case(61, 700, 20, 20, 32)
    q = np.ascontiguousarray(np.concatenate([case["pos"], case["neg"]]))
    bank = np.ascontiguousarray(case["bank"])
    dist = np.empty(len(q), np.float32)
    idx = np.empty(len(q), np.int64)
    p = ctypes.c_void_p
    rc = lib.gl_fbb_knn_l2_host(ctx.handle, bank.ctypes.data_as(p), len(bank), q.ctypes.data_as(p), len(q), 3 * 32 * 32, 64,
                                dist.ctypes.data_as(p), idx.ctypes.data_as(p))
    assert rc == 0, lib.gl_last_error()
    od, oi, _ = coracle.knn_l2_u8(bank, q, 64)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    rc = lib.gl_fbb_knn_l2_host(ctx.handle, bank.ctypes.data_as(p), 10, q.ctypes.data_as(p), len(q), 3 * 32 * 32, 64,
                                dist.ctypes.data_as(p), idx.ctypes.data_as(p))
    assert rc == -5 and b"no full batch" in lib.gl_last_error()


def test_codec_kernels(gl, synth, oracle, coracle):
    from ganleaks_amd import _lib
    lib = _lib.load()
    ctx = gl.Context.get()
    p = ctypes.c_void_p
    u = synth.lowpass_u8_images(3, 5, 32)
    x = oracle.dequantize_u8(u)
    # decode
    du = ctx.to_device(u)
    dx = ctx.empty(x.shape, np.float32)
    assert lib.gl_decode_u8(ctx.handle, p(du.ptr), u.size, p(dx.ptr)) == 0
    assert np.array_equal(dx.numpy(), x)
    # quantise (both modes) on arbitrary floats, incl. the ends of the range
    rng = np.random.default_rng(0)
    f = rng.uniform(-1, 1, size=10007).astype(np.float32)
    f[:4] = [-1.0, 1.0, 0.0, -0.0]
    df = ctx.to_device(f)
    dq = ctx.empty(f.shape, np.uint8)
    for mode, name in ((0, "normalize"), (1, "half")):
        assert lib.gl_quantize_f32(ctx.handle, p(df.ptr), f.size, mode, p(dq.ptr)) == 0
        assert np.array_equal(dq.numpy(), oracle.quantize_to_u8(f, name))
    # prepare: biased rows + norms
    d = u[0].size
    rows = ctx.empty((len(u), lib.gl_l2_row_stride(d)), np.int8)
    norms = ctx.empty((len(u),), np.int32)
    assert lib.gl_l2_prepare(ctx.handle, p(du.ptr), len(u), d, p(rows.ptr), p(norms.ptr)) == 0
    assert np.array_equal(norms.numpy(), coracle.row_norms_u8(u))
    assert np.array_equal(rows.numpy()[:, :d].astype(np.int16) + 128, u.reshape(len(u), -1).astype(np.int16))


def test_randomised_shapes_vs_c_oracle(gl, coracle):
    """40 seeded random problem shapes (odd row lengths, single rows, batch sizes that do not divide the bank, chunked and resident) against
    the C oracle: indices and distances bit for bit"""
    rng = np.random.default_rng(20240)
    for case in range(40):
        d = int(rng.choice([1, 3, 7, 48, 100, 127, 128, 129, 300, 768, 1071, 3072, 5000]))
        nb = int(rng.integers(1, 700))
        nq = int(rng.integers(1, 300))
        bs = int(rng.choice([1, 2, 7, 30, 64]))
        if nb < bs:
            nb = bs
        lo, hi = (0, 256) if case % 3 else (100, 140)                 # narrow ranges produce many exact ties
        bank = rng.integers(lo, hi, size=(nb, d), dtype=np.uint8)
        q = rng.integers(lo, hi, size=(nq, d), dtype=np.uint8)
        q[0] = bank[int(rng.integers(0, nb))]
        od, oi, _ = coracle.knn_l2_u8(bank, q, bs)
        kw = {"chunk_bytes": int(rng.integers(1, 40)) * 2 * d} if case % 4 == 0 else {}
        dist, idx = gl.attack(q, bank, batch_size=bs, **kw)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), (case, d, nb, nq, bs)
