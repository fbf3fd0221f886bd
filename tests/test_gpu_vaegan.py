"""GPU parity of the VAEGAN generator (spectral-norm ConvTranspose stack + self-attention) against two consecutive
forwards of the reference's own Generator (tests/golden/vaegan_gen.npz).  Tolerance 5e-5 on outputs in (-1,1)."""
import os

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [1, 0])
def test_vaegan_generator_two_forwards(precision, synth, golden_dir):
    """precision 1 (default): split-fp16 convolutions; 0: fp32 MFMA.  The attention block (one q|k|v GEMM + the fp32-MFMA
    attention kernel) is shared."""
    from ganleaks_amd.gan_models.vaegan.train import Generator
    g = np.load(os.path.join(golden_dir, "vaegan_gen.npz"))
    gen = Generator(100, 64)
    assert "matched" in gen.load_state_dict(synth.vaegan_state_dict(777, 100, 64))
    gen.set_precision(precision)
    z = synth.latent(4, 6)
    out1 = gen.eval()(z)
    assert out1.shape == (6, 3, 64, 64)
    e1 = np.abs(out1 - g["out1"]).max()
    assert e1 < 5e-5, e1
    out2 = gen(z)                                       # the spectral-norm state advanced, like the reference's
    e2 = np.abs(out2 - g["out2"]).max()
    assert e2 < 5e-5, e2
    assert np.abs(gen.state_dict()["deconv1.module.weight_u"] - g["u1"]).max() < 1e-5
    assert np.abs(gen.state_dict()["deconv4.module.weight_v"] - g["v4"]).max() < 1e-5
    assert gen._precision == precision                  # no saturation fallback happened


def test_vaegan_oracle_and_chunks(synth):
    import vaegan_oracle
    from ganleaks_amd._lib import check
    from ganleaks_amd.gan_models.vaegan.train import Generator
    sd = synth.vaegan_state_dict(778, 64, 32)            # d = 32: attention width 64
    z = synth.latent(6, 70, 64)
    ref = vaegan_oracle.VaeganOracle(sd).forward(z[:4])
    gen = Generator(64, 32)
    gen.load_state_dict(sd)
    check(gen.ctx.lib.gl_dcgan_set_chunk(gen._ensure(), 32))    # 70 = 32 + 32 + 6
    out = gen(z)
    assert np.abs(out[:4] - ref).max() < 5e-5
    gen2 = Generator(64, 32)
    gen2.load_state_dict(sd)
    assert np.array_equal(gen2(z), out)
    gen3 = Generator(64, 32)
    gen3.load_state_dict(sd)
    gen3.set_precision(0)
    assert np.abs(gen3(z) - out).max() < 2e-5


def test_sample_script(tmp_path, synth):
    """vaegan/sample.py: seed 1000, batches of 100, one forward (and one spectral-norm step) per batch; generated.npz + samples.png"""
    import torch
    from ganleaks_amd.gan_models.vaegan import sample
    from ganleaks_amd.gan_models.vaegan.train import Generator
    sd = synth.vaegan_state_dict(779, 64, 32)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, tmp_path / "netG.pt")
    out = sample.main(sample.parse_args(["--model_dir", str(tmp_path), "--num_samples", "150"]))
    g = np.load(out)
    assert g["noise"].shape == (150, 64) and g["img_r01"].shape == (150, 64, 64, 3) and g["img_r01"].dtype == np.float32
    assert (tmp_path / "samples.png").stat().st_size > 1000
    # the same schedule by hand: two forwards of 100 latents each
    torch.manual_seed(1000)
    ref = Generator(64, 32)
    ref.load_state_dict(sd)
    z1, z2 = torch.randn(100, 64, 1, 1), torch.randn(100, 64, 1, 1)
    a, b = ref(z1.numpy()), ref(z2.numpy())
    assert np.array_equal(g["noise"][:100], z1.numpy().reshape(100, 64))
    want = (np.concatenate([a, b])[:150] + np.float32(1)) / np.float32(2)
    assert np.array_equal(g["img_r01"], want.transpose(0, 2, 3, 1))
    torch.save(torch.nn.Linear(2, 2), tmp_path / "netG.pt")       # a pickled module is refused, not executed
    with pytest.raises(ValueError):
        sample.main(sample.parse_args(["--model_dir", str(tmp_path)]))


def test_vaegan_bank_attack_against_c_oracle(synth):
    """the image half of BASELINE configs[4]: VAEGAN generate_u8 -> 8-bit bank -> exact L2 1-NN, against the C oracle
    (mirror of the DCGAN / PGGAN bank tests)"""
    import c_oracle
    import ganleaks_amd as gl
    import oracle
    from ganleaks_amd.gan_models.vaegan.train import Generator
    gen = Generator(100, 64)
    gen.load_state_dict(synth.vaegan_state_dict(777, 100, 64))
    z = synth.latent(14, 200)
    bank = gen.eval().generate_u8(z)
    hb = bank.numpy()
    assert hb.shape == (200, 3, 64, 64) and hb.dtype == np.uint8
    q = np.concatenate([synth.perturb_u8(3, hb[[7, 150, 199]], 5.0), synth.lowpass_u8_images(5, 3, 64)])
    d, i = gl.attack(q, bank, batch_size=64)
    od, oi, _ = c_oracle.knn_l2_u8(hb, q, 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od) and i[0] == 7 and i[1] == 150 and i[2] != 199
    # and the codes are the quantisation of what forward() returns for the same spectral-norm state
    gen2 = Generator(100, 64)
    gen2.load_state_dict(synth.vaegan_state_dict(777, 100, 64))
    f = gen2.eval()(z)
    diff = hb.astype(np.int32) - oracle.quantize_to_u8(f).astype(np.int32)
    assert np.abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3          # forward() and generate_u8 are separate launches of the same arithmetic
