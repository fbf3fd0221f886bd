"""Pin the PGGAN oracle against outputs of the reference's Generator (CPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_pggan_oracle_matches_reference(synth, golden_dir):
    import pggan_oracle
    g = np.load(os.path.join(golden_dir, "pggan_gen.npz"))
    for ci, (z_dim, C, steps, alpha) in enumerate(g["cases"]):
        z_dim, C, steps = int(z_dim), int(C), int(steps)
        sd = synth.pggan_state_dict(4321 + C, z_dim, C)
        out = pggan_oracle.pggan_forward(sd, synth.latent(3, 4, z_dim), steps, float(alpha))
        ref = g["case%d" % ci]
        assert out.shape == ref.shape == (4, 3, 4 * 2 ** steps, 4 * 2 ** steps)
        assert np.abs(out - ref).max() < 2e-5, (ci, np.abs(out - ref).max())
    sd = synth.pggan_state_dict(101, 64, 64, prefix="gen.1.")
    out = pggan_oracle.pggan_forward(sd, synth.latent(3, 4, 64), 2, 1.0, prefix="gen.1.")
    assert np.abs(out - g["stack_g1"]).max() < 2e-5


def test_pggan_oracle_matches_reference_at_128_and_256(synth, golden_dir):
    """steps 5 and 6 (BASELINE configs[3] is PGGAN-256 = steps 6): tests/golden/pggan_gen_big.npz, make_golden.py make_pggan_big.
    The 512-channel case (14 GMAC in fp64) is left to the GPU test."""
    import pggan_oracle
    g = np.load(os.path.join(golden_dir, "pggan_gen_big.npz"))
    for ci, (z_dim, C, steps, alpha, n) in enumerate(g["cases"]):
        z_dim, C, steps, n = int(z_dim), int(C), int(steps), int(n)
        if C > 256:
            continue
        sd = synth.pggan_state_dict(4321 + C, z_dim, C)
        out = pggan_oracle.pggan_forward(sd, synth.latent(5, n, z_dim), steps, float(alpha))
        ref = g["case%d" % ci]
        assert out.shape == ref.shape == (n, 3, 4 * 2 ** steps, 4 * 2 ** steps)
        assert np.abs(out - ref).max() < 2e-5, (ci, np.abs(out - ref).max())
