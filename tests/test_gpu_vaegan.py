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
