#!/usr/bin/env python3
"""PGGAN-256 (in_channels 512, steps 6) generation of N images a few times: a target for rocprofv3 --kernel-trace --stats, and a
timer of the images-per-pass setting.

    python tools/run_pggan256.py [N] [images_per_pass ...]       (0 = the library's default pass size)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ganleaks_amd as gl  # noqa: E402
from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGGAN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
passes = [int(a) for a in sys.argv[2:]] or [0]
ctx = gl.Context.get()
g = PGGAN(512, 512, 3)
g.load_state_dict(gl.synth.pggan_state_dict(1, 512, 512))
z = ctx.to_device(gl.synth.latent(2, n, 512).reshape(n, 512))
for per_pass in passes:
    g.set_chunk(per_pass)
    g.forward_device(z, 6, 1.0, False, True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        g.forward_device(z, 6, 1.0, False, True)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"images": n, "images_per_pass": per_pass, "ms": round(dt * 1e3, 2), "images_per_s": round(n / dt, 1),
                      "alg_tflops": round(n * 56.3e9 / dt / 1e12, 1)}), flush=True)
