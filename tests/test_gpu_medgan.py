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
    # tabular rows go through the fp32 L2 search (no image lattice)
    d, i = gl.attack(rows[:5], rows, batch_size=1)
    assert i.tolist() == [0, 1, 2, 3, 4] and np.all(d == 0)
    with pytest.raises(gl.GanLeaksError):
        Generator(100, 128)._ensure()
