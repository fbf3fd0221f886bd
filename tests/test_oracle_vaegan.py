"""Pin the VAEGAN oracle (incl. the stateful spectral norm) against the reference's Generator (CPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_vaegan_oracle_matches_reference(synth, golden_dir):
    import vaegan_oracle
    g = np.load(os.path.join(golden_dir, "vaegan_gen.npz"))
    o = vaegan_oracle.VaeganOracle(synth.vaegan_state_dict(777, 100, 64))
    z = synth.latent(4, 6)
    out1 = o.forward(z)
    out2 = o.forward(z)            # second call: u, v advanced once more
    assert np.abs(out1 - g["out1"]).max() < 2e-5
    assert np.abs(out2 - g["out2"]).max() < 2e-5
    assert np.abs(out1 - out2).max() > 1e-4          # the drift is real: parity must fix the number of prior forwards
    assert np.abs(o.sd["deconv1.module.weight_u"].numpy() - g["u1"]).max() < 1e-5
    assert np.abs(o.sd["deconv4.module.weight_v"].numpy() - g["v4"]).max() < 1e-5
