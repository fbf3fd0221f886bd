"""Pin the LPIPS oracle against vectors produced by the reference's PNetLin (CPU only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.mark.parametrize("name", ["lpips_res32", "lpips_res64"])
def test_lpips_oracle_matches_reference(name, synth, oracle, golden_dir):
    import lpips_oracle
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    lin = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    lin = [lin["lin%d" % i] for i in range(5)]
    assert [len(w) for w in lin] == synth.LPIPS_CHANNELS and all((w >= 0).all() for w in lin)
    sd = synth.vgg16_state_dict(int(g["vgg_seed"]))
    case = synth.attack_case(int(g["seed"]), int(g["n_bank"]), int(g["n_pos"]), int(g["n_neg"]), int(g["res"]), sigma=20.0)
    bank = oracle.dequantize_u8(case["bank"])
    q = oracle.dequantize_u8(np.concatenate([case["pos"], case["neg"]]))
    taps = lpips_oracle.vgg16_taps(sd, lpips_oracle.scale_input(q[:1]))
    assert np.array_equal(np.array([list(t.shape[1:]) for t in taps]), g["tap_shapes"])
    np.testing.assert_allclose([float(t.sum()) for t in taps], g["tap_sums"], rtol=2e-5)
    lp = lpips_oracle.lpips_matrix(sd, lin, q, bank)
    np.testing.assert_allclose(lp, g["lpips"], atol=2e-6)
    d, i, _ = lpips_oracle.knn_l2_lpips(sd, lin, bank, q, int(g["batch_size"]))
    assert np.array_equal(i, g["idx"])
    np.testing.assert_allclose(d.astype(np.float64), g["dist"], atol=2e-6)


@pytest.mark.parametrize("name", ["lpips_res256", "lpips_res128x256"])
def test_lpips_oracle_matches_reference_at_config3_size(name, synth, oracle, golden_dir):
    """256 x 256 (PGGAN-256's images, BASELINE configs[3]) and a non-square size against the reference's PNetLin + custom_knn"""
    import lpips_oracle
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    lin = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    lin = [lin["lin%d" % i] for i in range(5)]
    sd = synth.vgg16_state_dict(int(g["vgg_seed"]))
    bank_u8, q_u8 = synth.lpips_big_case(name)
    assert tuple(bank_u8.shape[2:]) == tuple(g["shape"])
    bank, q = oracle.dequantize_u8(bank_u8), oracle.dequantize_u8(q_u8)
    d, i, tot = lpips_oracle.knn_l2_lpips(sd, lin, bank, q, int(g["batch_size"]))
    assert np.array_equal(i, g["idx"])
    np.testing.assert_allclose(d.astype(np.float64), g["dist"], atol=2e-6)
    lp = lpips_oracle.lpips_matrix(sd, lin, q, bank)
    np.testing.assert_allclose(lp, g["lpips"], atol=2e-6)


def test_torch_port_l2_lpips_matches_reference(synth, oracle, golden_dir):
    """oracle/torch_port.py's fp32 restatement of Loss('l2-lpips').forward (the timed CPU baseline of bench.py's l2-lpips line) against the
    reference's PNetLin + custom_knn vectors"""
    import torch
    import torch_port
    g = np.load(os.path.join(golden_dir, "lpips_res32.npz"))
    lin = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    loss = torch_port.make_l2_lpips_loss(synth.vgg16_state_dict(int(g["vgg_seed"])), [lin["lin%d" % i] for i in range(5)])
    case = synth.attack_case(int(g["seed"]), int(g["n_bank"]), int(g["n_pos"]), int(g["n_neg"]), int(g["res"]), sigma=20.0)
    bank = torch_port.dequantize(case["bank"])
    q = torch_port.dequantize(np.concatenate([case["pos"], case["neg"]]))
    bs = int(g["batch_size"])
    for k in (0, 3, 7):
        d, i = torch_port.custom_knn(bank, q[k], loss, bs)
        assert i == int(g["idx"][k]) and abs(d - float(g["dist"][k])) < 1e-6
