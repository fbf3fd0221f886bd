#!/usr/bin/env python3
"""Time LpipsModel.features (VGG16 + taps + norms) on N random 64 x 64 images: python tools/time_vgg_features.py [N] [repeats]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if any(k in os.environ for k in ("GL_TAP_FUSE", "GL_H3_HALO", "GL_H3_TILE128", "GL_HALO_RING", "GL_H3_T4")):      # tuning switches exist only in the tuning build
    os.environ.setdefault("GANLEAKS_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gan-leaks_amd", "libganleaks_hip_tuning.so"))
import ganleaks_amd as gl  # noqa: E402
from ganleaks_amd.lpips import LpipsModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lin = np.load(os.path.join(root, "tests", "golden", "lpips_lin_v0.1.npz"))
m = LpipsModel().load_state_dicts(gl.synth.vgg16_state_dict(7), {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
ctx = gl.Context.get()
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, size=(n, 3, 64, 64), dtype=np.uint8))
fb = m.features(imgs, role="bank")
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    fb = m.features(imgs, role="bank", out=fb)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"images": n, "ms": round(dt * 1e3, 3), "images_per_s": round(n / dt, 1), "GL_TAP_FUSE": os.environ.get("GL_TAP_FUSE", "1"), "GL_HALO_RING": os.environ.get("GL_HALO_RING", "")}))
