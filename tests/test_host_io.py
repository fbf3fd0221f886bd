"""CPU-only host logic: file helpers, read_image, CLI surface, eval_roc metrics of the PACKAGE
(not the oracle) against vectors made by the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    return ganleaks_amd


def test_sorted_paths_and_read_image_match_reference(gl, golden_dir):
    from ganleaks_amd.attack_models import utils
    g = np.load(os.path.join(golden_dir, "png_case.npz"))
    d = os.path.join(golden_dir, "png_case")
    paths = utils.get_filepaths_from_dir(d, ext="png")
    assert [os.path.relpath(p, d) for p in paths] == list(g["rel_paths"])      # string order: image_10 < image_2
    arr = np.array([utils.read_image(f, 16) for f in paths])
    assert arr.dtype == np.float64 and arr.shape == (12, 16, 16, 3)
    assert np.array_equal(arr, g["images"])                                    # incl. the resized 24x24 file
    u8 = utils.read_images_u8_nchw(paths, 16)
    assert np.array_equal(2.0 * (u8.transpose(0, 2, 3, 1) / 255.0) - 1.0, g["images"])
    # the worker-process form (used for banks of thousands of files) returns the same bytes in the same order
    assert np.array_equal(utils.read_images_u8_nchw(paths, 16, workers=3), u8)


def test_png_bank_round_trip_with_worker_processes(gl, tmp_path):
    """save_png_bank -> load_png_bank: file names of the generate branch, string-sorted load order, identical bytes for any worker count"""
    from ganleaks_amd import bank_io
    imgs = np.random.default_rng(5).integers(0, 256, (23, 3, 16, 16), dtype=np.uint8)
    for workers, sub in ((1, "a"), (4, "b")):
        d = bank_io.save_png_bank(imgs, str(tmp_path / sub), workers=workers)
        assert sorted(os.listdir(d)) == sorted("image_%d.png" % i for i in range(23))
        bank, paths = bank_io.load_png_bank(d, 16, workers=workers)
        assert np.array_equal(bank, imgs[bank_io.generation_order(paths)])
    with pytest.raises(RuntimeError):
        from ganleaks_amd.attack_models import utils
        utils.read_images_u8_nchw([str(tmp_path / "missing.png")] * 4, 16, workers=2)


@pytest.mark.parametrize("name", ["roc_sep", "roc_ties", "roc_small", "roc_knn_c1"])
def test_package_plot_roc_matches_reference(name, gl, golden_dir):
    from ganleaks_amd.attack_models.eval_roc import plot_roc
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    fpr, tpr, thr, auc, ap, prec = plot_roc(-g["pos_loss"], -g["neg_loss"])
    assert abs(auc - float(g["auc"])) < 1e-12 and abs(ap - float(g["ap"])) < 1e-12 and abs(prec - float(g["precision"])) < 1e-12
    np.testing.assert_allclose(fpr, g["fpr"], atol=1e-12)
    np.testing.assert_allclose(tpr, g["tpr"], atol=1e-12)
    assert np.array_equal(thr, g["thr"])


def test_eval_roc_main_files(gl, golden_dir, tmp_path):
    from ganleaks_amd.attack_models import eval_roc
    g = np.load(os.path.join(golden_dir, "roc_sep.npz"))
    np.save(tmp_path / "pos_loss.npy", g["pos_loss"])
    np.save(tmp_path / "neg_loss.npy", g["neg_loss"])
    args = eval_roc.parse_arguments(["--result_load_dir", str(tmp_path), "--attack_type", "fbb"])
    auc, ap, prec = eval_roc.main(args)
    assert abs(auc - float(g["auc"])) < 1e-12
    assert (tmp_path / "roc.png").exists()
    # calibrated branch (the reference's own is broken: 6 values unpacked into 5, eval_roc.py:101)
    ref = tmp_path / "ref"
    ref.mkdir()
    np.save(ref / "pos_loss.npy", g["pos_loss"] * 0.5)
    np.save(ref / "neg_loss.npy", g["neg_loss"] * 0.5)
    args.reference_load_dir = str(ref)
    eval_roc.main(args)


def test_plot_hist_writes_a_figure(gl, tmp_path):
    pytest.importorskip("matplotlib")
    from ganleaks_amd.attack_models.eval_roc import plot_hist
    rng = np.random.default_rng(0)
    out = tmp_path / "hist.png"
    plot_hist(rng.normal(0.05, 0.01, (200, 1)), rng.normal(0.08, 0.01, (180, 1)), str(out))
    assert out.stat().st_size > 1000


def test_figure_helpers(gl, tmp_path):
    pytest.importorskip("matplotlib")
    from ganleaks_amd.attack_models import utils
    rng = np.random.default_rng(1)
    imgs = rng.uniform(-1, 1, (7, 16, 16, 3))
    assert np.allclose(utils.inverse_transform(np.array([-1.0, 0.0, 1.0])), [0.0, 0.5, 1.0]) and utils.NCOLS == 5
    utils.visualize_gt(imgs, str(tmp_path))
    utils.visualize_progress(imgs, rng.uniform(0, 1, 7), str(tmp_path), 3)
    utils.visualize_samples(rng.uniform(0, 1, (64, 8, 8, 3)), str(tmp_path))
    for name in ("input.png", "output_3.png", "samples.png"):
        assert (tmp_path / name).stat().st_size > 1000


def test_fbb_cli_surface(gl):
    from ganleaks_amd.attack_models import fbb
    a = fbb.parse_arguments([])
    # the reference's defaults (attack_models/fbb.py:20-37)
    assert (a.exp_name, a.data_num, a.resolution, a.K, a.BATCH_SIZE) == ("debug", 20000, 64, 5, 30)
    assert a.hyperparameter_search is False and a.params is None and a.wandb is None and a.local_config is None
    assert a.distance == "l2-lpips"
    fbb.update_args(a, {"BATCH_SIZE": 64, "K": 1, "exp_name": "x"})
    assert a.BATCH_SIZE == 64 and a.exp_name == "x"
    r = subprocess.run([sys.executable, "-m", "ganleaks_amd.attack_models.fbb", "--help"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "--syn_data_path" in r.stdout and "--BATCH_SIZE" in r.stdout


def test_check_args_writes_params(gl, tmp_path, monkeypatch):
    from ganleaks_amd.attack_models import fbb
    monkeypatch.chdir(tmp_path)
    a = fbb.parse_arguments(["--syn_data_path", str(tmp_path), "--exp_name", "e1"])
    a, save_dir = fbb.check_args(a)
    assert save_dir == os.path.join(str(tmp_path), "fbb_attack", "e1")
    txt = open(os.path.join(save_dir, "params.txt")).read()
    assert "BATCH_SIZE:30" in txt and os.path.exists(os.path.join(save_dir, "params.pkl"))


def test_generator_state_dict_validation_without_gpu(gl):
    """constructing the Python object needs no GPU; using it does"""
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator, stackGenerators
    g = Generator(100, 3, 64)
    st = stackGenerators(100, 3, 64, 2)
    assert st.num_generators == 2 and len(st.gen) == 2
    if gl.device_count() == 0:
        with pytest.raises(gl.GanLeaksError):
            g.load_state_dict(gl.synth.dcgan_state_dict(1))


def test_custom_knn_with_a_foreign_loss_callable(gl, oracle):
    """custom_knn(syn_imgs, sample, loss, args) accepts any callable like the reference (fbb.py:73-88); only Loss instances go to the device"""
    import types
    from ganleaks_amd.attack_models.fbb import custom_knn
    rng = np.random.default_rng(3)
    bank = rng.uniform(-1, 1, size=(75, 3, 8, 8)).astype(np.float32)
    q = rng.uniform(-1, 1, size=(3, 8, 8)).astype(np.float32)
    bank[40] = q
    bank[10] = q                                                   # a tie: the first index wins
    args = types.SimpleNamespace(BATCH_SIZE=32)
    d, i = custom_knn(bank, q, lambda x, y: ((y - x) ** 2).mean(axis=(1, 2, 3)), args)
    assert isinstance(d, float) and isinstance(i, int) and (d, i) == (0.0, 10)
    od, oi = oracle.custom_knn_literal(bank, q, 32)
    assert (od, oi) == (d, i)
    import torch
    d2, i2 = custom_knn(torch.from_numpy(bank), torch.from_numpy(q), lambda x, y: torch.mean((y - x) ** 2, dim=[1, 2, 3]), args)
    assert (d2, i2) == (0.0, 10)
    with pytest.raises(ValueError):
        custom_knn(bank[:31], q, lambda x, y: ((y - x) ** 2).mean(axis=(1, 2, 3)), args)
    with pytest.raises(TypeError):
        custom_knn(bank, q, 3.0, args)
