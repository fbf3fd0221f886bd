"""Pin the CPU oracle against vectors produced by the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

KNN_CASES = ["knn_c1", "knn_b30", "knn_res32", "knn_res16"]


@pytest.mark.parametrize("name", KNN_CASES)
def test_knn_exact_matches_reference(name, oracle, synth, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    case = synth.attack_case(int(g["seed"]), int(g["n_bank"]), int(g["n_pos"]), int(g["n_neg"]), int(g["res"]))
    bs = int(g["batch_size"])
    for kind in ("pos", "neg"):
        dist, idx, _ = oracle.knn_l2_u8(case["bank"], case[kind], bs)
        assert np.array_equal(idx, g[kind + "_idx"]), "nearest-neighbour indices must be bit-exact"
        # reference value is an fp32 torch.mean widened to python float (fbb.py:88)
        np.testing.assert_allclose(dist.astype(np.float64), g[kind + "_dist"], rtol=0, atol=1e-6)
        assert idx.max() < oracle.n_effective(int(g["n_bank"]), bs)


def test_knn_ties_truncation(oracle, synth, golden_dir):
    g = np.load(os.path.join(golden_dir, "knn_ties.npz"))
    base = synth.lowpass_u8_images(77, 1000, 64)
    bank = base.copy()
    bank[700] = bank[5]
    bank[300] = bank[5]
    queries = np.stack([bank[5], bank[990], synth.perturb_u8(1, bank[700:701], 3.0)[0], bank[959], bank[0]])
    dist, idx, ssd = oracle.knn_l2_u8(bank, queries, 64)
    assert np.array_equal(idx, g["idx"])
    assert idx[0] == 5 and ssd[0] == 0            # first of three identical samples
    assert idx[1] < 960                           # twin in the dropped tail is not found
    np.testing.assert_allclose(dist.astype(np.float64), g["dist"], rtol=0, atol=1e-6)


def test_knn_empty_bank_error(oracle, synth, golden_dir):
    with open(os.path.join(golden_dir, "knn_empty_error.txt")) as f:
        ref_err = f.read().strip()
    assert ref_err == "ValueError"
    bank = synth.lowpass_u8_images(1, 10, 16)
    with pytest.raises(ValueError):
        oracle.knn_l2_u8(bank, bank[:1], 64)


def test_float_path_and_literal_agree_with_exact(oracle, synth):
    case = synth.attack_case(21, 200, 6, 6, 32)
    bank_f = oracle.dequantize_u8(case["bank"])
    q_f = oracle.dequantize_u8(case["pos"])
    d_u8, i_u8, _ = oracle.knn_l2_u8(case["bank"], case["pos"], 64)
    d_f, i_f = oracle.knn_l2_f32(bank_f, q_f, 64)
    assert np.array_equal(i_u8, i_f)
    np.testing.assert_allclose(d_u8, d_f, atol=1e-6)
    for k in range(len(q_f)):
        d, i = oracle.custom_knn_literal(bank_f, q_f[k], 64)
        assert i == i_u8[k]
        assert abs(d - d_u8[k]) < 1e-6


def test_lattice_detection(oracle, synth):
    u = synth.lowpass_u8_images(3, 4, 16)
    ok, back = oracle.is_on_u8_lattice(oracle.dequantize_u8(u))
    assert ok and np.array_equal(back, u)
    x = oracle.dequantize_u8(u).copy()
    x.flat[5] += 1e-4
    assert not oracle.is_on_u8_lattice(x)[0]


def test_quantize_roundtrip(oracle):
    u = np.arange(256, dtype=np.uint8)
    # an image written by the generate branch and read back by read_image keeps its codes
    x = oracle.dequantize_u8(u)
    # (x+1)/2*255 is not exactly u in fp32 for every code; quantisation truncates, so re-encoding a
    # decoded value may lose one code: document by checking the bound rather than equality
    back = oracle.quantize_to_u8(x).astype(int)
    assert np.all((u.astype(int) - back >= 0) & (u.astype(int) - back <= 1))
    assert oracle.quantize_to_u8(np.array([-1.0, 1.0], np.float32)).tolist() == [0, 255]
    assert oracle.quantize_to_u8(np.array([0.0], np.float32), "half").tolist() == [127]


@pytest.mark.parametrize("name", ["roc_sep", "roc_ties", "roc_small", "roc_knn_c1"])
def test_plot_roc_matches_reference(name, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    fpr, tpr, thr, auc, ap, prec = oracle.plot_roc(-g["pos_loss"], -g["neg_loss"])
    assert abs(auc - float(g["auc"])) < 1e-12
    assert abs(ap - float(g["ap"])) < 1e-12
    assert abs(prec - float(g["precision"])) < 1e-12
    np.testing.assert_allclose(fpr, g["fpr"], atol=1e-12)
    np.testing.assert_allclose(tpr, g["tpr"], atol=1e-12)
    np.testing.assert_allclose(thr, g["thr"], atol=0)


def test_dcgan_generator_matches_reference(oracle, synth, golden_dir):
    g = np.load(os.path.join(golden_dir, "dcgan_gen.npz"))
    sd = synth.dcgan_state_dict(int(g["weight_seed"]))
    z = synth.latent(int(g["z_seed"]), int(g["n"]))
    out = oracle.dcgan_generator_forward(sd, z)
    assert out.shape == (8, 3, 64, 64)
    np.testing.assert_allclose(out, g["out"], atol=5e-6)
    np.testing.assert_allclose(out, g["out_wgangp"], atol=5e-6)
    sd1 = synth.dcgan_state_dict(int(g["weight_seed"]) + 1, prefix="gen.1.gen.")
    out1 = oracle.dcgan_generator_forward(sd1, z, prefix="gen.1.gen.")
    np.testing.assert_allclose(out1, g["stack_out_g1"], atol=5e-6)


def test_pack_key_order(oracle):
    ssd = np.array([5, 5, 4, 2 ** 30], np.int64)
    idx = np.array([9, 3, 100, 0], np.int64)
    k = oracle.pack_key(ssd, idx)
    assert int(np.argmin(k)) == 2
    assert np.argsort(k).tolist() == [2, 1, 0, 3]
    s, i = oracle.unpack_key(k)
    assert np.array_equal(s, ssd) and np.array_equal(i, idx)
