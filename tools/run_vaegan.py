#!/usr/bin/env python3
"""VAEGAN-64 generator, 16384 images x 5 forwards (for rocprofv3 --kernel-trace --stats): python tools/run_vaegan.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ganleaks_amd as gl
from ganleaks_amd.gan_models.vaegan.train import Generator
g = Generator(100, 64)
g.load_state_dict(gl.synth.vaegan_state_dict(777, 100, 64))
z = gl.synth.latent(3, 16384)
g.generate_u8(z)
ctx = gl.Context.get()
ctx.sync(); t0 = time.perf_counter()
for _ in range(5):
    g.generate_u8(z)
ctx.sync()
print("%.1f k img/s" % (5 * 16384 / (time.perf_counter() - t0) / 1e3))
