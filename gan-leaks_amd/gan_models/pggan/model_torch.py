"""Counterpart of the reference's gan_models/pggan/model_torch.py for inference: `Generator`
(model_torch.py:49-88) and `stackGenerators` (:207-216) on the HIP generator (csrc/gl_pggan.hip).
The discriminator and training are out of scope.

`Generator(z_dim, in_channels, img_channels=3)(x, steps, alpha)`: x [N,z_dim,1,1] -> [N,img_channels,4*2^steps,...].
Weights via `load_state_dict` with the reference's key names (initial.1.*, initial.3.*, initial_rgb.*,
prog_blocks.{i}.conv{1,2}.*, rgb_layers.{j}.*).  The reference generates with steps=4, alpha=1
(gan_models/pggan/train.py:238).
"""
from __future__ import annotations

import ctypes

import numpy as np

from ..._lib import Context, as_device, check

_p = ctypes.c_void_p
FACTORS = [1, 1, 1, 1, 1 / 2, 1 / 4, 1 / 8, 1 / 16, 1 / 32]


def _np(v):
    if type(v).__module__.startswith("torch"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class Generator:
    def __init__(self, z_dim, in_channels, img_channels=3, ctx=None):
        self.z_dim, self.in_channels, self.img_channels = int(z_dim), int(in_channels), int(img_channels)
        self._ctx = ctx
        self._handle = None
        self._loaded = False
        self._precision = 1

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context.get()
        return self._ctx

    def _ensure(self):
        if self._handle is None:
            h = _p()
            check(self.ctx.lib.gl_pggan_create(self.ctx.handle, self.z_dim, self.in_channels, self.img_channels, ctypes.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self.ctx.lib.gl_pggan_destroy(self._handle)
            except Exception:  # noqa: BLE001
                pass

    def load_state_dict(self, state_dict, strict=True, prefix=""):
        h = self._ensure()
        lib = self.ctx.lib
        C, nc = self.in_channels, self.img_channels
        sd = state_dict

        def get(name, shape):
            if prefix + name not in sd:
                raise KeyError("missing key in state_dict: %s" % (prefix + name))
            a = _np(sd[prefix + name])
            if a.shape != tuple(shape):
                raise ValueError("%s has shape %s, expected %s" % (prefix + name, a.shape, tuple(shape)))
            return a

        used = set()

        def ptr(name, shape):
            used.add(prefix + name)
            a = get(name, shape)
            self._keep.append(a)
            return a.ctypes.data_as(_p)

        self._keep = []
        check(lib.gl_pggan_set_initial(h, ptr("initial.1.weight", (self.z_dim, C, 4, 4)), ptr("initial.1.bias", (C,)),
                                       ptr("initial.3.conv.weight", (C, C, 3, 3)), ptr("initial.3.bias", (C,))))
        check(lib.gl_pggan_set_rgb(h, 0, ptr("initial_rgb.conv.weight", (nc, C, 1, 1)), ptr("initial_rgb.bias", (nc,))))
        used.update({prefix + "rgb_layers.0.conv.weight", prefix + "rgb_layers.0.bias"})
        for i in range(len(FACTORS) - 1):
            ci, co = int(C * FACTORS[i]), int(C * FACTORS[i + 1])
            if co < 1:
                break
            check(lib.gl_pggan_set_block(h, i, ptr(f"prog_blocks.{i}.conv1.conv.weight", (co, ci, 3, 3)), ptr(f"prog_blocks.{i}.conv1.bias", (co,)),
                                         ptr(f"prog_blocks.{i}.conv2.conv.weight", (co, co, 3, 3)), ptr(f"prog_blocks.{i}.conv2.bias", (co,))))
            check(lib.gl_pggan_set_rgb(h, i + 1, ptr(f"rgb_layers.{i + 1}.conv.weight", (nc, co, 1, 1)), ptr(f"rgb_layers.{i + 1}.bias", (nc,))))
        self._keep = []
        if strict:
            extra = [k for k in sd if k.startswith(prefix) and k not in used]
            if extra:
                raise KeyError("unexpected keys in state_dict: %s" % extra[:5])
        self._loaded = True
        return "<All keys matched successfully>"

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):                 # vaegan/sample.py:36 calls .cuda(); the weights already live on the device
        return self

    def train(self, mode=True):              # inference only: BatchNorm always uses its running statistics
        return self

    def set_precision(self, mode):
        """1 (default) = split-fp16 convolutions, 0 = fp32 MFMA"""
        check(self.ctx.lib.gl_pggan_set_precision(self._ensure(), int(mode)))
        self._precision = int(mode)

    def set_chunk(self, images_per_pass):
        check(self.ctx.lib.gl_pggan_set_chunk(self._ensure(), int(images_per_pass)))

    def forward_device(self, x, steps, alpha, want_f32=True, want_u8=False, check_range=True):
        if not self._loaded:
            raise RuntimeError("Generator: load_state_dict() has not been called")
        z = as_device(self.ctx, x, np.float32)
        n = z.shape[0]
        if int(np.prod(z.shape[1:], dtype=np.int64)) != self.z_dim:
            raise ValueError("expected z of shape [N,%d,1,1], got %s" % (self.z_dim, z.shape))
        R = 4 * 2 ** int(steps)
        shape = (n, self.img_channels, R, R)
        f32 = self.ctx.empty(shape, np.float32) if want_f32 else None
        u8 = self.ctx.empty(shape, np.uint8) if want_u8 else None
        def run():
            check(self.ctx.lib.gl_pggan_forward(self._handle, _p(z.ptr), n, int(steps), ctypes.c_float(alpha), _p(f32.ptr if f32 else 0),
                                                _p(u8.ptr if u8 else 0)))
        run()
        if check_range and self._precision == 1 and self.ctx.h3_saturations() > 0:
            import warnings
            warnings.warn("split-fp16 generator path saturated for these weights; falling back to fp32 MFMA products")
            self.set_precision(0)
            run()
        return f32, u8

    def forward(self, x, steps, alpha):
        f32, _ = self.forward_device(x, steps, alpha, True, False)
        out = f32.numpy()
        if type(x).__module__.startswith("torch"):
            import torch
            return torch.from_numpy(out).to(x.device)
        return out

    __call__ = forward

    def generate_u8(self, x, steps=4, alpha=1.0):
        """the bank of the generate branch (gan_models/pggan/train.py:222-249): gen(noise, 4, 1) * 0.5 + 0.5 -> bytes"""
        _, u8 = self.forward_device(x, steps, alpha, False, True)
        return u8


class stackGenerators:
    """model_torch.py:207-216; keys `gen.{i}.{...}`; forward(x, steps, alpha, i)"""

    def __init__(self, z_dim, in_channels, img_channels, num_generators, ctx=None):
        self.num_generators = int(num_generators)
        self.gen = [Generator(z_dim, in_channels, img_channels, ctx) for _ in range(self.num_generators)]

    def load_state_dict(self, state_dict, strict=True):
        for i, g in enumerate(self.gen):
            g.load_state_dict(state_dict, strict=strict, prefix="gen.%d." % i)
        return "<All keys matched successfully>"

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):                 # vaegan/sample.py:36 calls .cuda(); the weights already live on the device
        return self

    def train(self, mode=True):              # inference only: BatchNorm always uses its running statistics
        return self

    def forward(self, x, steps, alpha, i):
        return self.gen[i](x, steps, alpha)

    __call__ = forward
