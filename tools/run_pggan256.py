#!/usr/bin/env python3
"""PGGAN-256 (in_channels 512, steps 6) generation of N images a few times: a target for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ganleaks_amd as gl  # noqa: E402
from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGGAN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = gl.Context.get()
g = PGGAN(512, 512, 3)
g.load_state_dict(gl.synth.pggan_state_dict(1, 512, 512))
z = ctx.to_device(gl.synth.latent(2, n, 512).reshape(n, 512))
for _ in range(3):
    g.forward_device(z, 6, 1.0, False, True)
ctx.sync()
