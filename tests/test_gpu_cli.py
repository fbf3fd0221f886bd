"""GPU: the fbb / eval_roc command-line drivers end to end on PNG directories (the reference's data
format between the generate branch and the attack)."""
import os
import types

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


def _write_pngs(d, imgs_u8_nchw, prefix="image_"):
    import PIL.Image
    os.makedirs(d, exist_ok=True)
    for i, im in enumerate(imgs_u8_nchw):
        PIL.Image.fromarray(im.transpose(1, 2, 0)).save(os.path.join(d, "%s%d.png" % (prefix, i)))


def test_fbb_main_and_eval_roc(tmp_path, monkeypatch, synth, oracle):
    import c_oracle
    from ganleaks_amd.attack_models import eval_roc, fbb, utils
    case = synth.attack_case(81, 150, 25, 22, 16)
    _write_pngs(tmp_path / "syn", case["bank"])
    _write_pngs(tmp_path / "pos", case["pos"])
    _write_pngs(tmp_path / "neg", case["neg"])
    monkeypatch.chdir(tmp_path)
    args = fbb.parse_arguments(["--exp_name", "t", "--syn_data_path", str(tmp_path / "syn"), "--pos_data_dir", str(tmp_path / "pos"),
                                "--neg_data_dir", str(tmp_path / "neg"), "--resolution", "16", "--BATCH_SIZE", "64", "--distance", "l2"])
    fbb.main(args)
    out = tmp_path / "fbb_attack" / "t"
    pos_loss = np.load(out / "pos_loss.npy")
    neg_loss = np.load(out / "neg_loss.npy")
    assert pos_loss.shape == (25, 1) and pos_loss.dtype == np.float64 and neg_loss.shape == (22, 1)
    assert np.array_equal(np.load(out / "pos_idx.npy"), np.arange(25).reshape(-1, 1))
    assert np.array_equal(np.load(out / "neg_idx.npy"), np.arange(25).reshape(-1, 1))     # the reference's len(pos_loss) quirk
    for f in ("params.txt", "params.pkl", "0pos.png", "19pos.png", "0neg.png", "19neg.png"):
        assert (out / f).exists(), f
    # bank order is the sorted() path-string order (image_10 < image_2), indices refer to it
    order = [int(os.path.basename(p)[6:-4]) for p in utils.get_filepaths_from_dir(str(tmp_path / "syn"), "png")]
    bank_sorted = case["bank"][order]
    pos_sorted = case["pos"][[int(os.path.basename(p)[6:-4]) for p in utils.get_filepaths_from_dir(str(tmp_path / "pos"), "png")]]
    od, oi, _ = c_oracle.knn_l2_u8(bank_sorted, pos_sorted, 64)
    assert np.array_equal(np.load(out / "pos_nn_idx.npy")[:, 0], oi)
    assert np.array_equal(pos_loss[:, 0], od.astype(np.float64))
    # eval
    ev = eval_roc.parse_arguments(["--result_load_dir", str(out), "--attack_type", "fbb"])
    auc, ap, prec = eval_roc.main(ev)
    _, _, _, oauc, oap, oprec = oracle.plot_roc(-pos_loss, -neg_loss)
    assert abs(auc - oauc) < 1e-12 and abs(ap - oap) < 1e-12 and prec == oprec
    assert (out / "roc.png").exists()
    # default distance is the reference's 'l2-lpips'; without local weight files it says which are missing
    args2 = fbb.parse_arguments(["--syn_data_path", str(tmp_path / "syn"), "--pos_data_dir", str(tmp_path / "pos"),
                                 "--neg_data_dir", str(tmp_path / "neg"), "--resolution", "16", "--BATCH_SIZE", "64"])
    monkeypatch.delenv("GANLEAKS_VGG16_PATH", raising=False)
    from ganleaks_amd import lpips
    lpips.set_default_model(None)
    with pytest.raises(FileNotFoundError):
        fbb.main(args2)


def test_generate_branch_png_bank_and_sweep(tmp_path, monkeypatch, synth):
    """generator -> PNG bank (generate branch) -> fbb over a hyper-parameter sweep directory (fbb.py:114-123)"""
    import ganleaks_amd as gl
    from ganleaks_amd import bank_io
    from ganleaks_amd.attack_models import fbb
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    sweep = tmp_path / "sweep"
    banks = {}
    for tag, seed in (("lr_a", 1234), ("lr_b", 1235)):
        g = Generator(100, 3, 64)
        g.load_state_dict(synth.dcgan_state_dict(seed))
        u8 = g.generate_u8(synth.latent(3, 70))
        bank_io.save_png_bank(u8, str(sweep / tag))
        banks[tag] = u8.numpy()
    loaded, paths = bank_io.load_png_bank(str(sweep / "lr_a"), 64)
    order = bank_io.generation_order(paths)
    assert order[:3].tolist() == [0, 1, 10] and np.array_equal(loaded, banks["lr_a"][order])    # PNG is lossless; sorted() order
    queries = synth.perturb_u8(4, banks["lr_a"][[3, 40]], 4.0)
    bank_io.save_png_bank(queries, str(tmp_path / "pos"))
    bank_io.save_png_bank(synth.lowpass_u8_images(5, 2, 64), str(tmp_path / "neg"))
    monkeypatch.chdir(tmp_path)
    args = fbb.parse_arguments(["--exp_name", "hp", "--syn_data_path", str(sweep), "--pos_data_dir", str(tmp_path / "pos"), "--neg_data_dir",
                                str(tmp_path / "neg"), "--BATCH_SIZE", "64", "--distance", "l2"])
    args.hyperparameter_search = True
    res = fbb.main(args)
    assert len(res) == 2
    for tag in ("lr_a", "lr_b"):
        out = tmp_path / "fbb_attack" / ("hp__sweep") / tag
        assert (out / "pos_loss.npy").exists() and (out / "params.txt").exists(), out
        nn = np.load(out / "pos_nn_idx.npy")[:, 0]
        d_direct, i_direct = gl.attack(queries, banks[tag][bank_io.generation_order(bank_io.load_png_bank(str(sweep / tag), 64)[1])], batch_size=64)
        assert np.array_equal(nn, i_direct)
        assert np.array_equal(np.load(out / "pos_loss.npy")[:, 0], d_direct.astype(np.float64))


def test_generate_branch_command_line(tmp_path, synth, oracle):
    """the generate branch of gan_models/dcgan/train_torch.py (:138-174): generator.pth -> npz_images / npz_noise / png_images/<timestamp>,
    driven through a YAML config like the reference's; the PNG bank is what fbb.py then reads"""
    import subprocess
    import sys
    import torch
    import yaml
    from ganleaks_amd import bank_io
    from ganleaks_amd.gan_models.dcgan import train_torch
    model_dir = tmp_path / "model"
    model_dir.mkdir()
    sd = synth.dcgan_state_dict(1234)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, model_dir / "generator.pth")
    cfg = {"training": False, "generate": True, "saved_model_name": str(model_dir), "num_generated": 70, "PATH_syn_data": str(tmp_path / "syn")}
    (tmp_path / "gen.yaml").write_text(yaml.safe_dump(cfg))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-m", "ganleaks_amd.gan_models.dcgan.train_torch", "--local_config", str(tmp_path / "gen.yaml")], check=True,
                   cwd=root)
    stamps = os.listdir(tmp_path / "syn" / "png_images")
    assert len(stamps) == 1 and stamps[0].startswith("_20")
    fake = np.load(tmp_path / "syn" / "npz_images" / stamps[0] / "dcgan_synthetic_data.npz")["fake"]
    noise = np.load(tmp_path / "syn" / "npz_noise" / stamps[0] / "dcgan_noise.npz")["noise"]
    assert fake.shape == (70, 3, 64, 64) and fake.dtype == np.float32 and noise.shape == (70, 100, 1, 1)
    ref = oracle.dcgan_generator_forward(sd, noise[:3])
    assert np.abs((fake[:3] * 2 - 1) - ref).max() < 2e-5 and fake.min() >= 0 and fake.max() <= 1
    bank, paths = bank_io.load_png_bank(str(tmp_path / "syn" / "png_images" / stamps[0]), 64)
    order = bank_io.generation_order(paths)
    # ToPILImage: the [0,1] float image times 255, truncated (train_torch.py:172): the PNG bytes are exactly that of the saved `fake`
    assert np.array_equal(bank, np.clip(np.floor(fake * np.float32(255.0)), 0, 255).astype(np.uint8)[order])
    args = train_torch.parse_arguments([])
    args.training = True
    with pytest.raises(NotImplementedError):
        train_torch.main(args)


def test_generate_branches_wgangp_and_pggan(tmp_path, synth):
    """wgangp/train.py:139-174 (draws `batch_size` images) and pggan/train.py:205-257 (gen(noise, 4, 1) * 0.5 + 0.5)"""
    import torch
    from ganleaks_amd import bank_io
    from ganleaks_amd.gan_models.pggan import train as pg_train
    from ganleaks_amd.gan_models.wgangp import train as wg_train
    wdir = tmp_path / "w"
    wdir.mkdir()
    torch.save({k: torch.from_numpy(v) for k, v in synth.dcgan_state_dict(77).items()}, wdir / "generator.pth")
    a = wg_train.parse_arguments(["--saved_model_name", str(wdir), "--PATH_syn_data", str(tmp_path / "wsyn"), "--batch_size", "33"])
    png, npz_i, npz_n = wg_train.main(a)
    fake = np.load(npz_i)["fake"]
    assert os.path.basename(npz_i) == "wgangp_synthetic_data.npz" and os.path.basename(npz_n) == "wgangp_noise.npz" and fake.shape == (33, 3, 64, 64)
    bank, paths = bank_io.load_png_bank(png, 64)
    assert np.array_equal(bank, np.clip(np.floor(fake * np.float32(255.0)), 0, 255).astype(np.uint8)[bank_io.generation_order(paths)])
    pdir = tmp_path / "p"
    pdir.mkdir()
    torch.save({k: torch.from_numpy(v) for k, v in synth.pggan_state_dict(5, 128, 128).items()}, pdir / "generator.pth")
    a = pg_train.parse_arguments(["--saved_model_name", str(pdir), "--PATH_syn_data", str(tmp_path / "psyn"), "--num_generated", "21", "--nz", "128",
                                  "--in_channels", "128"])
    png, npz_i, npz_n = pg_train.main(a)
    fake = np.load(npz_i)["fake"]
    assert os.path.basename(npz_i) == "pggan_images.npz" and fake.shape == (21, 3, 64, 64) and np.load(npz_n)["noise"].shape == (21, 128, 1, 1)
    bank, paths = bank_io.load_png_bank(png, 64)
    assert np.array_equal(bank, np.clip(np.floor(fake * np.float32(255.0)), 0, 255).astype(np.uint8)[bank_io.generation_order(paths)])
    a.training = True
    with pytest.raises(NotImplementedError):
        pg_train.main(a)


def test_priv_generate_sweep_feeds_fbb(tmp_path, monkeypatch, synth, oracle):
    """privDCGAN / privPGGAN generate branches (privDCGAN.py:168-215, privPGGAN.py:372-437): gen.pth of a generator stack -> generator 0 ->
    png_images/<params_keys>/<params_values>/image_{i}.png for every combination of a hyper-parameter sweep -> fbb.main(hyperparameter_search)"""
    import torch
    import yaml
    import ganleaks_amd as gl
    from ganleaks_amd import bank_io
    from ganleaks_amd.attack_models import fbb
    from ganleaks_amd.gan_models.dcgan import privDCGAN
    from ganleaks_amd.gan_models.pggan import privPGGAN
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGen
    models = tmp_path / "models"
    sweep = {"lr": [0.1, 0.2], "privacy_ratio": [0.5]}
    (tmp_path / "sweep.yaml").write_text(yaml.safe_dump(sweep))
    seeds = {"0.1-0.5": 1234, "0.2-0.5": 2234}
    for vals, seed in seeds.items():
        d = models / "lr-privacy_ratio" / vals
        d.mkdir(parents=True)
        sd = {}
        for gi in range(2):
            sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in synth.dcgan_state_dict(seed + gi, prefix="gen.%d.gen." % gi).items()})
        torch.save(sd, d / "gen.pth")
    (tmp_path / "gen.yaml").write_text(yaml.safe_dump({"training": False, "generate": True, "num_generated": 70, "PATH": str(models),
                                                       "PATH_syn_data": str(tmp_path / "syn"), "N_splits": 2}))
    args = privDCGAN.parse_arguments(["--local_config", str(tmp_path / "gen.yaml"), "--hyperparameter_search", str(tmp_path / "sweep.yaml")])
    for k, v in yaml.safe_load((tmp_path / "gen.yaml").read_text()).items():
        setattr(args, k, v)
    torch.manual_seed(5)
    outs = privDCGAN.main(args)
    assert len(outs) == 2
    root = tmp_path / "syn" / "png_images" / "lr-privacy_ratio"
    assert sorted(os.listdir(root)) == ["0.1-0.5", "0.2-0.5"]
    # the bank of the first combination is generator 0 of ITS stack on the stored noise
    noise = np.load(tmp_path / "syn" / "npz_noise" / "lr-privacy_ratio" / "0.1-0.5" / "dcgan_noise.npz")["noise"]
    g0 = Generator(100, 3, 64)
    g0.load_state_dict(synth.dcgan_state_dict(1234))
    want = g0.generate_u8(noise).numpy()
    loaded, paths = bank_io.load_png_bank(str(root / "0.1-0.5"), 64)
    assert len(paths) == 70 and np.array_equal(loaded, want[bank_io.generation_order(paths)])
    fake = np.load(tmp_path / "syn" / "npz_images" / "lr-privacy_ratio" / "0.1-0.5" / "dcgan_synthetic_data.npz")["fake"]
    assert fake.shape == (70, 3, 64, 64) and np.abs(fake * 255.0 - want).max() < 1.0 + 1e-3      # u8 = trunc(255 * fake)
    # ... and the tree is what fbb's sweep mode walks
    queries = synth.perturb_u8(4, want[[3, 40]], 4.0)
    bank_io.save_png_bank(queries, str(tmp_path / "pos"))
    bank_io.save_png_bank(synth.lowpass_u8_images(5, 2, 64), str(tmp_path / "neg"))
    monkeypatch.chdir(tmp_path)
    a = fbb.parse_arguments(["--exp_name", "priv", "--syn_data_path", str(root), "--pos_data_dir", str(tmp_path / "pos"), "--neg_data_dir",
                             str(tmp_path / "neg"), "--BATCH_SIZE", "64", "--distance", "l2"])
    a.hyperparameter_search = True
    res = fbb.main(a)
    assert len(res) == 2
    by = {os.path.basename(r[0]): r for r in res}
    d_direct, i_direct = gl.attack(queries, loaded, batch_size=64)
    assert np.array_equal(by["0.1-0.5"][3][:, 0], i_direct) and np.array_equal(by["0.1-0.5"][1][:, 0], d_direct.astype(np.float64))
    # privPGGAN: one run without a sweep (timestamp folder), generator 0 of a 2-stack, Normalize(-1, 2) bytes
    pm = tmp_path / "pmodel"
    pm.mkdir()
    sd = {}
    for gi in range(2):
        sd.update({k: torch.from_numpy(v) for k, v in synth.pggan_state_dict(100 + gi, 64, 64, prefix="gen.%d." % gi).items()})
    torch.save(sd, pm / "gen.pth")
    pa = privPGGAN.parse_arguments([])
    pa.training, pa.generate, pa.num_generated, pa.nz, pa.in_channels = False, True, 40, 64, 64
    pa.saved_model_name, pa.PATH_syn_data = str(pm), str(tmp_path / "psyn")
    outs = privPGGAN.main(pa)
    png_dir = outs[0][0]
    noise = np.load(outs[0][2])["noise"]
    pg = PGen(64, 64, 3)
    pg.load_state_dict(synth.pggan_state_dict(100, 64, 64))
    want = oracle.quantize_to_u8(pg(noise, 4, 1.0))                     # Normalize(-1, 2) + ToPILImage bytes
    loaded, paths = bank_io.load_png_bank(png_dir, 64)
    assert len(paths) == 40 and np.array_equal(loaded, want[bank_io.generation_order(paths)])
    with pytest.raises(NotImplementedError):
        pa.training = True
        privPGGAN.main(pa)
