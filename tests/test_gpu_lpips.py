"""GPU parity tests of the 'l2-lpips' path (VGG16 + LPIPS on fp32 MFMA, |V_q - V_n|^2 contraction + argmin)
against vectors produced by the reference's PNetLin (tests/golden/lpips_*.npz) and the fp64 oracle.
Tolerance: 1e-4 on distances (north_star); measured ~1e-6.  Backbone weights are seeded random (the ImageNet
VGG16 weights are not available offline), lin weights are the reference's vendored ones."""
import os
import types

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu
ATOL = 1e-4


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    return ganleaks_amd


@pytest.fixture(scope="module")
def lin(golden_dir):
    z = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    return {"lin%d" % i: z["lin%d" % i] for i in range(5)}


@pytest.fixture(scope="module")
def model(gl, synth, lin):
    from ganleaks_amd.lpips import LpipsModel
    return LpipsModel().load_state_dicts(synth.vgg16_state_dict(7), lin)


def _case(g, synth):
    case = synth.attack_case(int(g["seed"]), int(g["n_bank"]), int(g["n_pos"]), int(g["n_neg"]), int(g["res"]), sigma=20.0)
    return case["bank"], np.concatenate([case["pos"], case["neg"]])


@pytest.mark.parametrize("search_rows", ["fp16", "split"])
@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("name", ["lpips_res32", "lpips_res64"])
def test_attack_l2_lpips_matches_reference(name, precision, search_rows, gl, synth, model, golden_dir):
    """precision 1 (default): VGG16 convolutions as split-fp16; 0: fp32 MFMA.  search_rows 'fp16' (default): one half per
    LPIPS value + hi/lo image part, gl_feat_knn_h1; 'split': hi + lo of everything, gl_feat_knn.  Same bound for all four."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    bank, q = _case(g, synth)
    model.set_precision(precision)
    model.search_rows = search_rows
    try:
        dist, idx = gl.attack(q, bank, distance="l2-lpips", batch_size=int(g["batch_size"]), lpips=model)
    finally:
        model.set_precision(1)
        model.search_rows = "fp16"
    assert np.array_equal(idx, g["idx"])
    err = np.abs(dist.astype(np.float64) - g["dist"]).max()
    assert err < ATOL, err
    assert err < 5e-6, err       # what fp32 actually delivers


@pytest.mark.parametrize("name", ["lpips_res32"])
def test_loss_forward_and_custom_knn(name, gl, synth, model, oracle, golden_dir):
    from ganleaks_amd.attack_models.fbb import custom_knn
    from ganleaks_amd.attack_models.utils import Loss
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    bank, q = _case(g, synth)
    loss = Loss("l2-lpips", lpips=model)
    bs = int(g["batch_size"])
    v = loss(bank[:bs], q[:1])                                   # the reference's call shape: [B] vs [1]
    assert np.abs(loss.loss_lpips - g["lpips"][0, :bs]).max() < 5e-6
    l2 = (oracle.ssd_u8(bank[:bs], q[0]).astype(np.float64) * oracle.l2_scale(q[0].size)).astype(np.float32)
    assert np.abs(loss.loss_l2 - l2).max() < 1e-6
    assert np.abs(v - (0.2 * g["lpips"][0, :bs] + l2)).max() < 5e-6
    bank_f = oracle.dequantize_u8(bank)
    q_f = oracle.dequantize_u8(q)
    args = types.SimpleNamespace(BATCH_SIZE=bs)
    for k in (0, 5):
        d, i = custom_knn(bank_f, q_f[k], loss, args)
        assert isinstance(d, float) and isinstance(i, int)
        assert i == int(g["idx"][k]) and abs(d - float(g["dist"][k])) < 5e-6


def test_perceptual_loss_drop_in(gl, synth, model, golden_dir):
    """lpips_pytorch.PerceptualLoss(model='net-lin', net='vgg').forward(pred, target) -> [N,1,1,1], the surface utils.py:157,171 uses"""
    import torch
    from ganleaks_amd.attack_models import lpips_pytorch as ps
    g = np.load(os.path.join(golden_dir, "lpips_res32.npz"))
    bank, q = _case(g, synth)
    loss = ps.PerceptualLoss(model='net-lin', net='vgg', use_gpu=True, lpips_model=model)
    bs = int(g["batch_size"])
    pred = 2.0 * (bank[:bs].astype(np.float32) / 255.0) - 1.0
    target = 2.0 * (q[:1].astype(np.float32) / 255.0) - 1.0
    d = loss.forward(pred, target)
    assert d.shape == (bs, 1, 1, 1) and d.dtype == np.float32
    assert np.abs(d.reshape(-1) - g["lpips"][0, :bs]).max() < 5e-6
    dt = loss(torch.from_numpy((pred + 1) / 2), torch.from_numpy((target + 1) / 2), normalize=True)       # [0,1] images
    assert isinstance(dt, torch.Tensor) and tuple(dt.shape) == (bs, 1, 1, 1)
    assert np.abs(dt.numpy().reshape(-1) - g["lpips"][0, :bs]).max() < 2e-5      # (x+1)/2 and back is not exact in fp32
    with pytest.raises(NotImplementedError):
        ps.PerceptualLoss(model='net', net='alex')


def test_vs_fp64_oracle_ragged_and_shards(gl, synth, model, lin, oracle):
    """sizes that do not fill tiles; off-lattice float images; shard merge"""
    import lpips_oracle
    from ganleaks_amd.lpips import feat_knn_keys
    from ganleaks_amd.attack import unpack_keys
    sd = synth.vgg16_state_dict(7)
    linl = [lin["lin%d" % i] for i in range(5)]
    case = synth.attack_case(91, 37, 3, 2, 16, sigma=25.0)
    bank = case["bank"]
    q = np.concatenate([case["pos"], case["neg"]])
    od, oi, tot = lpips_oracle.knn_l2_lpips(sd, linl, oracle.dequantize_u8(bank), oracle.dequantize_u8(q), 8)
    d, i = gl.attack(q, bank, distance="l2-lpips", batch_size=8, lpips=model)
    assert np.array_equal(i, oi) and np.abs(d - od).max() < 5e-6
    assert i.max() < 32
    # off-lattice floats go through the fp32 feature kernel
    rng = np.random.default_rng(0)
    bf = np.clip(oracle.dequantize_u8(bank) + rng.normal(0, 0.01, bank.shape).astype(np.float32), -1, 1)
    qf = np.clip(oracle.dequantize_u8(q) + rng.normal(0, 0.01, q.shape).astype(np.float32), -1, 1)
    od, oi, _ = lpips_oracle.knn_l2_lpips(sd, linl, bf, qf, 8)
    d, i = gl.attack(qf, bf, distance="l2-lpips", batch_size=8, lpips=model)
    assert np.array_equal(i, oi) and np.abs(d - od).max() < 5e-6
    # shards, both row formats
    for roles in ((None, None), ("bank", "query")):
        fq = model.features(q, role=roles[1])
        keys = None
        for lo, hi in ((16, 32), (0, 16)):
            keys = feat_knn_keys(model.features(bank[lo:hi], index_base=lo, role=roles[0]), fq, keys=keys)
        ds, is_ = unpack_keys(gl.Context.get(), keys, fq.n, fq.K, "f32")
        model.search_rows = "fp16" if roles[0] else "split"
        try:
            d0, i0 = gl.attack(q, bank, distance="l2-lpips", batch_size=8, lpips=model)
        finally:
            model.search_rows = "fp16"
        assert np.array_equal(is_, i0) and np.array_equal(ds, d0)
    with pytest.raises(ValueError):
        feat_knn_keys(model.features(bank[:8]), model.features(q, role="query"))


def test_larger_images(gl, synth, model, lin, oracle):
    """128 x 128 against the fp64 oracle (10 images: a few seconds of CPU); 256 x 256 (a search row is 17 MB, PGGAN-256's size) only
    for agreement between the two row formats and between the streamed and the resident form"""
    import lpips_oracle
    sd = synth.vgg16_state_dict(7)
    linl = [lin["lin%d" % i] for i in range(5)]
    case = synth.attack_case(131, 8, 1, 1, 128, sigma=20.0)
    bank, q = case["bank"], np.concatenate([case["pos"], case["neg"]])
    od, oi, _ = lpips_oracle.knn_l2_lpips(sd, linl, oracle.dequantize_u8(bank), oracle.dequantize_u8(q), 4)
    d, i = gl.attack(q, bank, distance="l2-lpips", batch_size=4, lpips=model)
    assert np.array_equal(i, oi) and np.abs(d - od).max() < 5e-6
    case = synth.attack_case(133, 12, 2, 1, 256, sigma=20.0)
    bank, q = case["bank"], np.concatenate([case["pos"], case["neg"]])
    res = {}
    for rows in ("fp16", "split"):
        model.search_rows = rows
        try:
            res[rows] = gl.attack(q, bank, distance="l2-lpips", batch_size=4, lpips=model)
        finally:
            model.search_rows = "fp16"
    assert np.array_equal(res["fp16"][1], res["split"][1]) and np.abs(res["fp16"][0] - res["split"][0]).max() < 2e-6
    row = 2 * int(gl.Context.get().lib.gl_lpips_search_dim(256, 256))
    ds, is_ = gl.attack(q, bank, distance="l2-lpips", batch_size=4, lpips=model, chunk_bytes=2 * row)      # bank in 6 chunks, queries in 2 slices
    assert np.array_equal(is_, res["fp16"][1]) and np.array_equal(ds, res["fp16"][0])


def test_host_one_call_lpips_abi(gl, synth, model):
    """gl_fbb_knn_lpips_host straight through ctypes (what a non-Python host would bind), resident and streamed in 3 chunks"""
    import ctypes
    from ganleaks_amd import _lib
    lib = _lib.load()
    ctx = gl.Context.get()
    case = synth.attack_case(93, 70, 5, 4, 32, sigma=20.0)
    bank = np.ascontiguousarray(case["bank"])
    q = np.ascontiguousarray(np.concatenate([case["pos"], case["neg"]]))
    d0, i0 = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=model)
    p = ctypes.c_void_p
    row = 2 * int(lib.gl_lpips_search_dim(32, 32))
    for budget in (0, 30 * row):
        dist = np.empty(len(q), np.float32)
        idx = np.empty(len(q), np.int64)
        rc = lib.gl_fbb_knn_lpips_host(ctx.handle, model._handle, bank.ctypes.data_as(p), len(bank), q.ctypes.data_as(p), len(q), 32, 32, 16, budget,
                                       dist.ctypes.data_as(p), idx.ctypes.data_as(p))
        assert rc == 0, lib.gl_last_error()
        assert np.array_equal(idx, i0) and np.array_equal(dist, d0)
    rc = lib.gl_fbb_knn_lpips_host(ctx.handle, model._handle, bank.ctypes.data_as(p), 10, q.ctypes.data_as(p), len(q), 32, 32, 16, 0, dist.ctypes.data_as(p),
                                   idx.ctypes.data_as(p))
    assert rc == -5 and b"no full batch" in lib.gl_last_error()


def test_search_rows_multi_tile(gl, synth, model):
    """600 bank rows x 300 queries: several 256 x 256 tiles with ragged edges in both directions.  No CPU oracle at this size
    (VGG16 in fp64 takes minutes); the split-row search, itself pinned to the oracle above, is the reference."""
    case = synth.attack_case(97, 600, 150, 150, 32, sigma=20.0)
    bank = case["bank"]
    q = np.concatenate([case["pos"], case["neg"]])
    out = {}
    for rows in ("split", "fp16"):
        model.search_rows = rows
        try:
            out[rows] = gl.attack(q, bank, distance="l2-lpips", batch_size=64, lpips=model)
        finally:
            model.search_rows = "fp16"
    assert out["fp16"][1].max() < 576
    assert np.array_equal(out["fp16"][1], out["split"][1])
    assert np.abs(out["fp16"][0] - out["split"][0]).max() < 2e-6


def test_fbb_main_default_distance_with_local_weights(tmp_path, monkeypatch, gl, synth, lin, model):
    """the reference's default: Loss('l2-lpips') (fbb.py:148), weights from local files"""
    import torch
    import PIL.Image
    from ganleaks_amd import lpips
    from ganleaks_amd.attack_models import fbb
    torch.save({"features.%s" % k: torch.from_numpy(v) for k, v in synth.vgg16_state_dict(7).items()}, tmp_path / "vgg16.pth")
    torch.save({"lin%d.model.1.weight" % i: torch.from_numpy(lin["lin%d" % i]).view(1, -1, 1, 1) for i in range(5)}, tmp_path / "vgg_lin.pth")
    monkeypatch.setenv("GANLEAKS_VGG16_PATH", str(tmp_path / "vgg16.pth"))
    monkeypatch.setenv("GANLEAKS_LPIPS_LIN_PATH", str(tmp_path / "vgg_lin.pth"))
    lpips.set_default_model(None)
    case = synth.attack_case(95, 40, 5, 4, 32, sigma=20.0)
    for name in ("bank", "pos", "neg"):
        d = tmp_path / name
        d.mkdir()
        for k, im in enumerate(case[name]):
            PIL.Image.fromarray(im.transpose(1, 2, 0)).save(d / ("image_%02d.png" % k))
    monkeypatch.chdir(tmp_path)
    args = fbb.parse_arguments(["--exp_name", "lp", "--syn_data_path", str(tmp_path / "bank"), "--pos_data_dir", str(tmp_path / "pos"),
                                "--neg_data_dir", str(tmp_path / "neg"), "--resolution", "32", "--BATCH_SIZE", "16"])
    assert args.distance == "l2-lpips"
    fbb.main(args)
    out = tmp_path / "fbb_attack" / "lp"
    d_ref, i_ref = gl.attack(case["pos"], case["bank"], distance="l2-lpips", batch_size=16, lpips=model)
    assert np.array_equal(np.load(out / "pos_nn_idx.npy")[:, 0], i_ref)
    assert np.allclose(np.load(out / "pos_loss.npy")[:, 0], d_ref.astype(np.float64), atol=1e-7)
    lpips.set_default_model(None)


@pytest.mark.parametrize("hw", [(32, 48), (48, 16), (16, 16)])
def test_non_square_images(hw, gl, synth, model, lin, oracle):
    """H != W and the smallest supported size (four 2x2 pools need multiples of 16) against the fp64 oracle, both row formats"""
    import lpips_oracle
    H, W = hw
    rng = np.random.default_rng(H * 100 + W)
    bank = rng.integers(0, 256, (24, 3, H, W), dtype=np.uint8)
    q = np.clip(bank[[3, 17]].astype(int) + rng.integers(-9, 10, (2, 3, H, W)), 0, 255).astype(np.uint8)
    od, oi, _ = lpips_oracle.knn_l2_lpips(synth.vgg16_state_dict(7), [lin["lin%d" % i] for i in range(5)], oracle.dequantize_u8(bank), oracle.dequantize_u8(q), 8)
    for rows in ("fp16", "split"):
        model.search_rows = rows
        try:
            d, i = gl.attack(q, bank, distance="l2-lpips", batch_size=8, lpips=model)
        finally:
            model.search_rows = "fp16"
        assert np.array_equal(i, oi) and np.abs(d - od).max() < 5e-6
    with pytest.raises(ValueError):
        gl.attack(q[:, :, :, :8], bank[:, :, :, :8], distance="l2-lpips", batch_size=8, lpips=model)


@pytest.mark.parametrize("name", ["lpips_res256", "lpips_res128x256"])
def test_config3_image_size_matches_reference(name, gl, synth, model, golden_dir):
    """256 x 256 (PGGAN-256's images, BASELINE configs[3]) and a non-square 128 x 256 against the reference's own PNetLin + custom_knn
    (tests/golden/make_golden.py make_lpips_big): both VGG16 arithmetic modes, both row formats, resident and streamed"""
    from ganleaks_amd.attack_models.utils import Loss
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    bank, q = synth.lpips_big_case(name)
    H, W = bank.shape[2:]
    bs = int(g["batch_size"])
    row = 2 * int(gl.Context.get().lib.gl_lpips_search_dim(H, W))
    errs = {}
    for precision in (1, 0):
        for rows in ("fp16", "split"):
            for budget in (None, 3 * row):                  # None: resident; 3 rows: bank in chunks, queries in slices
                model.set_precision(precision)
                model.search_rows = rows
                try:
                    kw = {} if budget is None else {"chunk_bytes": budget * (1 if rows == "fp16" else 2)}
                    dist, idx = gl.attack(q, bank, distance="l2-lpips", batch_size=bs, lpips=model, **kw)
                finally:
                    model.set_precision(1)
                    model.search_rows = "fp16"
                assert np.array_equal(idx, g["idx"]), (precision, rows, budget)
                errs[(precision, rows, budget is not None)] = float(np.abs(dist.astype(np.float64) - g["dist"]).max())
    print(name, errs)
    assert max(errs.values()) < ATOL, errs                  # north_star's bound
    # K is 8.5 M values per row at 256 x 256: a single fp32 accumulation chain over it was 1.9e-5 off; the kernels sum in segments
    # (two-level), measured 6e-7 .. 1.4e-6
    assert max(errs.values()) < 5e-6, errs
    # the pure LPIPS matrix row of query 0 through Loss.forward (split rows)
    loss = Loss("l2-lpips", lpips=model)
    loss(bank[:bs], q[:1])
    assert np.abs(loss.loss_lpips - g["lpips"][0, :bs]).max() < 5e-6


def test_host_one_call_falls_back_when_split_fp16_saturates(gl, synth, lin):
    """VGG16 weights with a large first-layer gain push activations beyond the fp16 range of the split layout: gl_fbb_knn_lpips_host must notice
    (gl_ctx_h3_saturations) and redo the features with fp32 products, not return clamped results with GL_OK"""
    import ctypes
    from ganleaks_amd import _lib
    from ganleaks_amd.lpips import LpipsModel
    lib = _lib.load()
    ctx = gl.Context.get()
    sd = {k: v.copy() for k, v in synth.vgg16_state_dict(7).items()}
    sd["0.weight"] *= 3.0e4
    hot = LpipsModel().load_state_dicts(sd, lin)
    hot.set_calibration(False)           # the fixed activation scale of rounds 1-2: the calibrated scales adapt to such weights (test below)
    case = synth.attack_case(94, 48, 4, 3, 32, sigma=20.0)
    bank = np.ascontiguousarray(case["bank"])
    q = np.ascontiguousarray(np.concatenate([case["pos"], case["neg"]]))
    ref = LpipsModel().load_state_dicts(sd, lin)
    ref.set_precision(0)
    d0, i0 = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=ref)
    # the split path alone does saturate on these weights
    ctx.h3_saturations()
    check = hot.ctx.lib.gl_lpips_search_features_u8
    fb = ctx.empty((len(q), int(lib.gl_lpips_search_dim(32, 32))), np.float16)
    nb = ctx.empty((len(q),), np.float32)
    assert check(hot._handle, ctypes.c_void_p(ctx.to_device(q).ptr), len(q), 32, 32, 0, ctypes.c_void_p(fb.ptr), ctypes.c_void_p(nb.ptr)) == 0
    assert ctx.h3_saturations() > 0
    p = ctypes.c_void_p
    dist = np.empty(len(q), np.float32)
    idx = np.empty(len(q), np.int64)
    rc = lib.gl_fbb_knn_lpips_host(ctx.handle, hot._handle, bank.ctypes.data_as(p), len(bank), q.ctypes.data_as(p), len(q), 32, 32, 16, 0,
                                   dist.ctypes.data_as(p), idx.ctypes.data_as(p))
    assert rc == 0, lib.gl_last_error()
    assert np.array_equal(idx, i0) and np.abs(dist - d0).max() < 5e-6
    assert hot._precision == 1           # the caller's setting is restored
    # and the Python entry does the same
    with pytest.warns(UserWarning):
        d1, i1 = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=hot)
    assert np.array_equal(i1, i0) and np.abs(d1 - d0).max() < 5e-6


@pytest.mark.parametrize("role,res,n,passes", [("bank", 64, 288, (16, 96, 288)), (None, 64, 288, (16, 96, 288)), ("bank", 256, 20, (4, 20))])
def test_feature_rows_do_not_depend_on_the_pass_size(role, res, n, passes, gl, synth, lin):
    """The images-per-pass setting picks the kernels: small passes run every tap as its own kernel behind the tap-gather convolution,
    64+ images of 64 x 64 put conv1_2 / conv2_2 on the halo kernel with tap + max-pool in its epilogue (conv3_3, on the 256-channel tile, keeps
    its stand-alone tap); at 256 x 256, 4 images are enough for conv1_2 and 16 for conv2_x.  All of them must produce the same bits (a streamed bank is
    featurised in passes of whatever size is left)."""
    from ganleaks_amd.lpips import LpipsModel
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, size=(n, 3, res, res), dtype=np.uint8)
    rows = {}
    for per_pass in passes:
        m = LpipsModel().load_state_dicts(synth.vgg16_state_dict(7), lin)
        m.set_chunk(per_pass)
        fb = m.features(imgs, role=role)
        rows[per_pass] = ((fb.rows_numpy() if role else fb.V.numpy()).copy(), fb.norms.numpy()[:n].copy())
    for per_pass in passes[1:]:
        assert np.array_equal(rows[passes[0]][0].view(np.uint8), rows[per_pass][0].view(np.uint8)), \
            "V differs between passes of %d and %d images" % (passes[0], per_pass)
        assert np.array_equal(rows[passes[0]][1], rows[per_pass][1])


def test_lattice_and_hilo_search_rows_agree(gl, synth, model, lin, oracle):
    """8-bit images give lattice search rows (image part exact in one fp16 K segment); the hi / lo layout takes any float image.  Same
    neighbours, distances within the fp16 rounding of the LPIPS part; a mixed pair (8-bit bank, off-lattice float queries) is searched in
    the hi / lo layout on both sides, and mismatched layouts are refused."""
    import lpips_oracle
    from ganleaks_amd.lpips import feat_knn_keys
    from ganleaks_amd.attack import unpack_keys
    case = synth.attack_case(97, 40, 3, 3, 32, sigma=20.0)
    bank, q = case["bank"], np.concatenate([case["pos"], case["neg"]])
    ctx = gl.Context.get()
    res = {}
    for fmt in ("lattice", "hilo"):
        fb, fq = model.features(bank, role="bank", fmt=fmt), model.features(q, role="query", fmt=fmt)
        assert fb.fmt == fmt and fq.fmt == fmt
        res[fmt] = unpack_keys(ctx, feat_knn_keys(fb, fq), fq.n, fq.K, "f32")
    lib = ctx.lib
    assert model.features(q, role="query").fmt == "lattice" and model.features(q, role="query").K == int(lib.gl_lpips_lattice_dim(32, 32))
    assert int(lib.gl_lpips_lattice_dim(32, 32)) == int(lib.gl_lpips_feature_dim(32, 32))          # K_lpips + D: the algorithmic length
    assert np.array_equal(res["lattice"][1], res["hilo"][1]) and np.abs(res["lattice"][0] - res["hilo"][0]).max() < 2e-6
    sd = synth.vgg16_state_dict(7)
    od, oi, _ = lpips_oracle.knn_l2_lpips(sd, [lin["lin%d" % i] for i in range(5)], oracle.dequantize_u8(bank), oracle.dequantize_u8(q), 8)
    assert np.array_equal(res["lattice"][1], oi) and np.abs(res["lattice"][0] - od).max() < 5e-6
    # mixed: the bank stays 8-bit, the queries leave the lattice
    qf = np.clip(oracle.dequantize_u8(q) + np.random.default_rng(1).normal(0, 0.01, q.shape).astype(np.float32), -1, 1)
    od, oi, _ = lpips_oracle.knn_l2_lpips(sd, [lin["lin%d" % i] for i in range(5)], oracle.dequantize_u8(bank), qf, 8)
    for kw in ({}, {"chunk_bytes": 3 * 2 * int(lib.gl_lpips_search_dim(32, 32))}):       # resident, and streamed in chunks of 3 rows
        d, i = gl.attack(qf, bank, distance="l2-lpips", batch_size=8, lpips=model, **kw)
        assert np.array_equal(i, oi) and np.abs(d - od).max() < 5e-6
    with pytest.raises(ValueError):
        model.features(qf, role="query", fmt="lattice")
    with pytest.raises(ValueError):
        feat_knn_keys(model.features(bank, role="bank", fmt="hilo"), model.features(q, role="query"))


def test_split_path_on_weights_with_imagenet_like_dynamic_range(gl, synth, lin, oracle):
    """every split-fp16 VGG16 result so far came from a Kaiming backbone with O(1) activations.  With weights that have the
    dynamic range of a trained network (synth.vgg16_state_dict(imagenet_like=True): activations from ~1e-4 to ~6e3, layer levels rising to
    hundreds and falling back) the split path must (a) actually run -- no saturation, no fallback to fp32 products -- and (b) stay fp32-class:
    its error against the fp64 oracle at most 1.5 x the fp32-MFMA path's.  The per-layer activation scales come from the calibration pass
    (gl_lpips_set_calibration); with the fixed scale of rounds 1-2 the same weights lose accuracy or clamp."""
    import lpips_oracle
    from ganleaks_amd import lpips as lp
    from ganleaks_amd.lpips import LpipsModel
    ctx = gl.Context.get()
    sd = synth.vgg16_state_dict(7, imagenet_like=True)
    lins = [lin["lin%d" % i] for i in range(5)]
    rng = np.random.default_rng(12)
    bank = np.concatenate([synth.lowpass_u8_images(61, 10, 32), rng.integers(0, 256, size=(6, 3, 32, 32), dtype=np.uint8)])      # smooth + noise images
    q = synth.perturb_u8(62, bank[[0, 5, 11, 14]], 6.0)
    f = lambda u: (2.0 * (u / 255.0) - 1.0).astype(np.float32)
    ref = lpips_oracle.lpips_matrix(sd, lins, f(q), f(bank))                    # float64 [4, 16]
    errs = {}
    for precision in (1, 0):
        m = LpipsModel().load_state_dicts(sd, lin)
        m.set_precision(precision)
        ctx.h3_saturations()
        fb = m.features(bank)
        got = np.stack([lp.rows_dist(fb, m.features(q[k:k + 1]))[0] for k in range(len(q))])
        assert ctx.h3_saturations() == 0 and m._precision == precision, "the split path saturated on imagenet-like weights"
        errs[precision] = float(np.abs(got.astype(np.float64) - ref).max())
    print("lpips error vs fp64, imagenet-like weights: split %.3g, fp32 %.3g (values %.3g .. %.3g)" % (errs[1], errs[0], ref.min(), ref.max()))
    assert errs[1] <= 1.5 * errs[0] + 2e-7, errs
    assert errs[1] < 5e-6, errs
    # calibration is a function of the weights alone: a second model, used in another order, gives the same bits
    a = LpipsModel().load_state_dicts(sd, lin)
    b = LpipsModel().load_state_dicts(sd, lin)
    b.features(rng.integers(0, 256, size=(3, 3, 64, 64), dtype=np.uint8), role="bank")
    va, vb = a.features(bank, role="bank"), b.features(bank, role="bank")
    assert np.array_equal(va.rows_numpy(), vb.rows_numpy()) and np.array_equal(va.norms.numpy(), vb.norms.numpy())
    # and the search on these weights agrees with the oracle's nearest neighbours
    od, oi, _ = lpips_oracle.knn_l2_lpips(sd, lins, f(bank), f(q), 16)
    d, i = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=a)
    assert np.array_equal(i, oi) and np.abs(d - od).max() < 5e-6


def test_k_blocked_rows_in_both_layouts_and_on_the_float_path(gl, synth, model, lin, oracle):
    """Search rows of 2 MiB or more are stored K-blocked (include/ganleaks.h gl_lpips_search_rows_capacity).  At 128 x 128 both the lattice rows
    (4.1 MB) and the hi / lo rows (4.3 MB) are: same neighbours in both, equal to the fp64 oracle's, also when the queries are off-lattice
    floats (hi / lo on both sides, the fp32 feature entry points) and when the bank is streamed in chunks smaller than one block of 256 rows."""
    import lpips_oracle
    from ganleaks_amd.lpips import feat_knn_keys
    from ganleaks_amd.attack import unpack_keys
    ctx = gl.Context.get()
    case = synth.attack_case(131, 20, 2, 2, 128, sigma=20.0)
    bank, q = case["bank"], np.concatenate([case["pos"], case["neg"]])
    sd = synth.vgg16_state_dict(7)
    lins = [lin["lin%d" % i] for i in range(5)]
    od, oi, _ = lpips_oracle.knn_l2_lpips(sd, lins, oracle.dequantize_u8(bank), oracle.dequantize_u8(q), 4)
    res = {}
    for fmt in ("lattice", "hilo"):
        fb, fq = model.features(bank, role="bank", fmt=fmt), model.features(q, role="query", fmt=fmt)
        assert fb.blocked and fq.blocked and fb.V.shape[0] == 256 and fb.rows_numpy().shape == (20, fb.K)
        res[fmt] = unpack_keys(ctx, feat_knn_keys(fb, fq), fq.n, fq.K, "f32")
        assert np.array_equal(res[fmt][1], oi) and np.abs(res[fmt][0] - od).max() < 5e-6, fmt
    qf = np.clip(oracle.dequantize_u8(q) + np.random.default_rng(2).normal(0, 0.01, q.shape).astype(np.float32), -1, 1)
    odf, oif, _ = lpips_oracle.knn_l2_lpips(sd, lins, oracle.dequantize_u8(bank), qf, 4)
    row = 2 * int(ctx.lib.gl_lpips_search_dim(128, 128))
    for kw in ({}, {"chunk_bytes": 6 * row}):
        d, i = gl.attack(qf, bank, distance="l2-lpips", batch_size=4, lpips=model, **kw)
        assert np.array_equal(i, oif) and np.abs(d - odf).max() < 5e-6, kw
