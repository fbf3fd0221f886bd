"""GPU parity of the medGAN generator + decoder against outputs of the reference's own modules
(tests/golden/medgan_gen.npz).  Tolerance 2e-5 (fp32, different summation order)."""
import os

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


def test_medgan_generator_and_decoder(synth, golden_dir):
    import ganleaks_amd as gl
    from ganleaks_amd.gan_models.medgan.model import Autoencoder, Generator, generate_synthetic
    g = np.load(os.path.join(golden_dir, "medgan_gen.npz"))
    gsd, asd = synth.medgan_state_dicts(555, 1071)
    z = np.random.default_rng(8).standard_normal((37, 128)).astype(np.float32)
    gen = Generator(128, 128)
    gen.load_state_dict(gsd)
    h = gen.eval()(z)
    assert h.shape == (37, 128) and np.abs(h - g["hidden"]).max() < 2e-5
    for binary in (True, False):
        ae = Autoencoder(1071, 128, binary=binary)
        ae.load_state_dict(asd)
        dec = ae.decode(h)
        assert dec.shape == (37, 1071)
        assert np.abs(dec - g["decoded_binary%d" % int(binary)]).max() < 2e-5
        assert np.array_equal(ae.decoder(h), dec)
    ae = Autoencoder(1071, 128, binary=True)
    ae.load_state_dict(asd)
    rows = generate_synthetic(gen, ae, z)
    ref = (g["decoded_binary1"] >= 0.5).astype(np.float32)
    far = np.abs(g["decoded_binary1"] - 0.5) > 1e-4            # away from the threshold the bits must agree
    assert rows.dtype == np.float32 and set(np.unique(rows)) <= {0.0, 1.0}
    assert np.array_equal(rows[far], ref[far]) and far.mean() > 0.999
    # tabular rows go through the exact integer search (whole numbers 0..255, no image lattice)
    d, i = gl.attack(rows[:5], rows, batch_size=1)
    assert i.tolist() == [0, 1, 2, 3, 4] and np.all(d == 0)
    with pytest.raises(gl.GanLeaksError):
        Generator(100, 128)._ensure()


def test_medgan_bank_attack_against_c_oracle(synth):
    """the tabular half of BASELINE configs[4]: medGAN rows (F = 1071 binary columns) as the bank, perturbed rows and fresh rows as
    queries, against the C oracle: indices bit-exact, dist = fl32(S / F).  S = number of differing columns for 0/1 data, so the byte
    oracle (which squares differences of codes) applies with codes 0 / 1."""
    import c_oracle
    import ganleaks_amd as gl
    from ganleaks_amd.attack_models.utils import Loss
    from ganleaks_amd.gan_models.medgan.model import Autoencoder, Generator, generate_synthetic
    gsd, asd = synth.medgan_state_dicts(555, 1071)
    gen = Generator(128, 128)
    gen.load_state_dict(gsd)
    ae = Autoencoder(1071, 128, binary=True)
    ae.load_state_dict(asd)
    rng = np.random.default_rng(17)
    bank = generate_synthetic(gen.eval(), ae, rng.standard_normal((700, 128)).astype(np.float32))
    assert bank.shape == (700, 1071) and set(np.unique(bank)) <= {0.0, 1.0}
    q = bank[[5, 77, 650, 699]].copy()
    flip = rng.random(q.shape) < 0.03
    q = np.where(flip, 1.0 - q, q).astype(np.float32)
    fresh = generate_synthetic(gen, ae, rng.standard_normal((6, 128)).astype(np.float32))
    q = np.concatenate([q, fresh])
    d, i = gl.attack(q, bank, batch_size=64)                   # 700 -> 640 rows used (fbb.py:77)
    _, oi, ssd = c_oracle.knn_l2_u8(bank.astype(np.uint8), q.astype(np.uint8), 64)
    assert np.array_equal(i, oi) and i.max() < 640
    assert np.array_equal(d, (ssd.astype(np.float64) / 1071.0).astype(np.float32))
    assert i[0] == 5 and i[1] == 77 and i[2] != 650            # 650, 699 sit in the truncated tail
    # Loss('l2').forward on whole-number float rows (0/1 tables): mean((y - x)^2) = S / F
    loss = Loss("l2")
    v = loss(bank[:64], q[:1])
    want = (c_oracle.ssd_row_u8(bank[:64].astype(np.uint8), q[0].astype(np.uint8)).astype(np.float64) / 1071.0).astype(np.float32)
    assert v.shape == (64,) and np.array_equal(v, want)
    z0 = np.zeros((3, 1071), np.float32)
    assert np.array_equal(loss(z0, z0[:1]), np.zeros(3, np.float32))
    raw = rng.integers(0, 256, (5, 3, 8, 8)).astype(np.float32)   # raw 0..255 floats: also whole numbers, not image codes
    want = ((raw[:, None] - raw[None, :1]) ** 2).reshape(5, -1).mean(axis=1).astype(np.float32)
    assert np.allclose(loss(raw, raw[:1]), want[:, ] if want.ndim == 1 else want, rtol=1e-6)
