"""GPU: the kernels the product runs against their simpler siblings, bit for bit (the software-pipelined main loop of
gl_pair256.h places its fragment reads and waits by hand; keep a check against the plainly written kernels).  The siblings exist only in
the tuning build (libganleaks_hip_tuning.so, -DGL_TUNING), which is loaded in a child process through $GANLEAKS_LIB:

  int8 L2 search    : round-1 kernel (variant 0) == pipelined kernel (variant 1; what the shipped library runs) -- exact integers
  fp16 LPIPS search : persistent cluster kernel (3; shipped) == persistent kernel without clusters (5; what a device with fewer than
                      256 CUs gets): the same K segments and totals, so the same bits on any device
  convolutions      : the 4 x 4-grid tile order that skips the MFMAs of outside taps (GL_H3_T4=1; shipped) == the raster order (0)
"""
import json
import os
import subprocess
import sys

import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import ganleaks_amd as gl
from ganleaks_amd import _lib
from ganleaks_amd.attack import Bank, knn_keys
assert _lib.LIB_PATH.endswith("libganleaks_hip_tuning.so")
ctx = gl.Context.get()
rng = np.random.default_rng(5)
out = {}
# int8: 2100 x 8200 rows of 3 x 32 x 32 codes (ragged last tiles; enough 256-tiles for the large-tile kernel)
bank = Bank.from_images(rng.integers(0, 256, size=(8200, 3, 32, 32), dtype=np.uint8), ctx)
q = Bank.from_images(rng.integers(0, 256, size=(2100, 3, 32, 32), dtype=np.uint8), ctx)
keys = {}
for v in (0, 1, 4, 2):
    os.environ["GL_PAIR_VARIANT"] = str(v)
    keys[v] = knn_keys(bank, q)[0].numpy().copy()
out["l2_equal"] = [bool(np.array_equal(keys[0], keys[v])) for v in (1, 4, 2)]
# fp16 search rows: K = 2 segments + a ragged one (2048 slices of 64 halves per segment)
import ctypes
p = ctypes.c_void_p
K = 64 * (2 * 2048 + 100)
nb, nq = 700, 600
bv = ctx.to_device((rng.standard_normal((nb, K)) * 40).astype(np.float16))
qv = ctx.to_device((rng.standard_normal((nq, K)) * 40).astype(np.float16))
bn = ctx.to_device((bv.numpy().astype(np.float32) ** 2).sum(1).astype(np.float32) / 16384.0 ** 2)
qn = ctx.to_device((qv.numpy().astype(np.float32) ** 2).sum(1).astype(np.float32) / 16384.0 ** 2)
fk = {}
for v in (3, 5, 1):
    os.environ["GL_PAIR_VARIANT"] = str(v)
    k = ctx.empty((nq,), np.uint64)
    _lib.check(ctx.lib.gl_keys_init(ctx.handle, p(k.ptr), nq))
    _lib.check(ctx.lib.gl_feat_knn_h1(ctx.handle, p(bv.ptr), p(bn.ptr), nb, 0, p(qv.ptr), p(qn.ptr), nq, K, p(k.ptr)))
    fk[v] = k.numpy().copy()
out["feat_cluster_equals_plain_persistent"] = bool(np.array_equal(fk[3], fk[5]))
d3 = (fk[3] >> np.uint64(32)).astype(np.uint32).view(np.float32)
d1 = (fk[1] >> np.uint64(32)).astype(np.uint32).view(np.float32)
out["feat_unsegmented_rel_diff"] = float(np.max(np.abs(d3 - d1) / np.maximum(d3, 1e-30)))
out["feat_idx_equal_unsegmented"] = bool(np.array_equal(fk[3] & np.uint64(0xFFFFFFFF), fk[1] & np.uint64(0xFFFFFFFF)))
# the 4 x 4-grid tile order of gather_conv_h3 (T4: one position of 16 images per MFMA fragment, fragments of outside taps skipped) against the
# raster order: DCGAN's first strided layer and VGG16's conv5_x take it when the pass is large enough for the 256 x 256 tile
from ganleaks_amd.gan_models.dcgan.model_torch import Generator
from ganleaks_amd.lpips import LpipsModel
g = Generator(100, 3, 64, ctx)
g.load_state_dict(gl.synth.dcgan_state_dict(1234))
z = gl.synth.latent(11, 1100)                       # 1100 images: a ragged last tile of 12 images
lin = np.load(os.path.join(%(root)r, "tests", "golden", "lpips_lin_v0.1.npz"))
m = LpipsModel(ctx).load_state_dicts(gl.synth.vgg16_state_dict(7), {"lin%%d" %% i: lin["lin%%d" %% i] for i in range(5)})
imgs = rng.integers(0, 256, size=(2100, 3, 64, 64), dtype=np.uint8)
res = {}
for v in (1, 0):
    os.environ["GL_H3_T4"] = str(v)
    f32, u8 = g.forward_device(z, True, True)
    fb = m.features(imgs, role="bank")
    res[v] = (f32.numpy().copy(), u8.numpy().copy(), fb.rows_numpy().copy(), fb.norms.numpy().copy())
out["t4_equals_raster"] = [bool(np.array_equal(a, b)) for a, b in zip(res[1], res[0])]
# the halo kernel's ring of three weight slices with counted waits against its two-buffer form (VGG16 conv1_2 / conv2_x at 64 x 64, and a PGGAN block
# through x2 upsampling at 128 x 128)
from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGGAN
pg = PGGAN(128, 128, 3)
pg.load_state_dict(gl.synth.pggan_state_dict(3, 128, 128))
zp = gl.synth.latent(12, 40, 128)
hr = {}
for v in (1, 0):
    os.environ["GL_HALO_RING"] = str(v)
    fb = m.features(imgs, role="bank")
    hr[v] = (fb.rows_numpy().copy(), fb.norms.numpy().copy(), pg.generate_u8(zp, steps=5, alpha=1.0).numpy().copy())
os.environ.pop("GL_HALO_RING")
out["halo_ring_equals_two_buffers"] = [bool(np.array_equal(a, b)) for a, b in zip(hr[1], hr[0])]
print("RESULT " + json.dumps(out))
'''


def test_shipped_kernels_match_their_plain_siblings_bit_for_bit():
    tuning = os.path.join(ROOT, "gan-leaks_amd", "libganleaks_hip_tuning.so")
    if not os.path.exists(tuning):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gan-leaks_amd", "csrc"), "-j8", "tuning"], check=True)
    env = dict(os.environ, GANLEAKS_LIB=tuning)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert out["l2_equal"] == [True, True, True], out
    assert out["feat_cluster_equals_plain_persistent"] is True, out
    assert out["feat_unsegmented_rel_diff"] < 1e-4 and out["feat_idx_equal_unsegmented"], out
    assert out["t4_equals_raster"] == [True, True, True, True], out
    assert out["halo_ring_equals_two_buffers"] == [True, True, True], out
