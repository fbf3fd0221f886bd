#!/usr/bin/env python3
"""images per VGG16 pass (LpipsModel.set_chunk) against feature throughput at 64 x 64: python tools/sweep_vgg_chunk.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ganleaks_amd as gl
from ganleaks_amd.lpips import LpipsModel
ctx = gl.Context.get()
lin = np.load(os.path.join(ROOT, "tests", "golden", "lpips_lin_v0.1.npz"))
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, size=(16384, 3, 64, 64), dtype=np.uint8))
for chunk in (1024, 1536, 2048, 2560, 2816):
    m = LpipsModel().load_state_dicts(gl.synth.vgg16_state_dict(7), {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
    m.set_chunk(chunk)
    fb = m.features(imgs, role="bank")
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(3):
        fb = m.features(imgs, role="bank", out=fb)
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"images_per_pass": chunk, "ms_per_16384_images": round(dt * 1e3, 2), "images_per_s": round(16384 / dt)}), flush=True)
    del m, fb
