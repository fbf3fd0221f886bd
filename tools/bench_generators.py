#!/usr/bin/env python3
"""Throughput of the generator ports (images/s, algorithmic TFLOP/s) on one MI355X, both arithmetic modes.
Secondary measurement (the headline is bench.py); writes one JSON line per generator."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ganleaks_amd as gl  # noqa: E402
from ganleaks_amd.gan_models.dcgan.model_torch import Generator as DCGAN  # noqa: E402
from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGGAN  # noqa: E402
from ganleaks_amd.gan_models.vaegan.train import Generator as VAEGAN  # noqa: E402

synth = gl.synth
ctx = gl.Context.get()


def timed(fn, reps=3):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def main():
    out = []
    n = 16384
    g = DCGAN(100, 3, 64)
    g.load_state_dict(synth.dcgan_state_dict(1234))
    z = ctx.to_device(synth.latent(1, n).reshape(n, 100))
    for mode in (1, 0):
        g.set_precision(mode)
        t = timed(lambda: g.forward_device(z, False, True))
        out.append({"generator": "DCGAN-64", "precision": mode, "images": n, "images_per_s": n / t, "alg_tflops": n * 0.821e9 / t / 1e12})
    n = 2048
    g = PGGAN(512, 512, 3)
    g.load_state_dict(synth.pggan_state_dict(1, 512, 512))
    z = ctx.to_device(synth.latent(2, n, 512).reshape(n, 512))
    for mode in (1, 0):
        g.set_precision(mode)
        t = timed(lambda: g.forward_device(z, 4, 1.0, False, True))
        out.append({"generator": "PGGAN-64 (in_channels 512, steps 4)", "precision": mode, "images": n, "images_per_s": n / t, "alg_tflops": n * 27.3e9 / t / 1e12})
    n = 512
    z = ctx.to_device(synth.latent(2, n, 512).reshape(n, 512))
    g.set_precision(1)
    t = timed(lambda: g.forward_device(z, 6, 1.0, False, True))
    out.append({"generator": "PGGAN-256 (in_channels 512, steps 6)", "precision": 1, "images": n, "images_per_s": n / t, "alg_tflops": n * 56.3e9 / t / 1e12})
    n = 16384
    g = VAEGAN(100, 64)
    g.load_state_dict(synth.vaegan_state_dict(777, 100, 64))
    z = ctx.to_device(synth.latent(3, n).reshape(n, 100))
    for mode in (1, 0):
        g.set_precision(mode)
        t = timed(lambda: g.forward_device(z, False, True))
        out.append({"generator": "VAEGAN-64 (spectral norm + self-attention)", "precision": mode, "images": n, "images_per_s": n / t, "alg_tflops": n * 0.24e9 / t / 1e12})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
