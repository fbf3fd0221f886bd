#!/usr/bin/env python3
"""A/B of the halo kernel's weight ring (GL_HALO_PREFX, tuning build) in one process, alternating: VGG16 features at 64 x 64 and PGGAN-256."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GANLEAKS_LIB", os.path.join(ROOT, "gan-leaks_amd", "libganleaks_hip_tuning.so"))
import ganleaks_amd as gl
from ganleaks_amd.lpips import LpipsModel
from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGGAN
ctx = gl.Context.get()
lin = np.load(os.path.join(ROOT, "tests", "golden", "lpips_lin_v0.1.npz"))
m = LpipsModel().load_state_dicts(gl.synth.vgg16_state_dict(7), {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, size=(8192, 3, 64, 64), dtype=np.uint8))
fb = m.features(imgs, role="bank")
pg = PGGAN(512, 512, 3)
pg.load_state_dict(gl.synth.pggan_state_dict(1, 512, 512))
zp = gl.synth.latent(2, 512, 512)
pg.generate_u8(zp, steps=6, alpha=1.0)
res = {"vgg": {0: [], 1: []}, "pggan256": {0: [], 1: []}}
for r in range(4):
    for v in (0, 1):
        os.environ["GL_HALO_PREFX"] = str(v)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            fb = m.features(imgs, role="bank", out=fb)
        ctx.sync(); res["vgg"][v].append((time.perf_counter() - t0) / 3 * 1e3)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(2):
            pg.generate_u8(zp, steps=6, alpha=1.0)
        ctx.sync(); res["pggan256"][v].append((time.perf_counter() - t0) / 2 * 1e3)
for k, d in res.items():
    print(json.dumps({"workload": k, "off_ms": [round(x, 2) for x in d[0]], "on_ms": [round(x, 2) for x in d[1]],
                      "median_ratio": round(float(np.median(d[0]) / np.median(d[1])), 4)}))
