"""Counterpart of the reference's gan_models/dcgan/model_torch.py for inference:
`Generator` (model_torch.py:75-96) and `stackGenerators` (:99-108), running on the HIP generator
stack (csrc/gl_dcgan.hip).  Discriminators and training are out of scope.

Weights come from `load_state_dict` with the reference's key names
    gen.{0..3}.0.weight, gen.{0..3}.1.{weight,bias,running_mean,running_var}, gen.4.{weight,bias}
as numpy arrays or torch tensors (PyTorch is only used by the caller to read the .pth).
The module always behaves as in `.eval()` (BatchNorm uses running statistics), which is how the
generate branch runs it (gan_models/dcgan/train_torch.py:150).
"""
from __future__ import annotations

import ctypes

import numpy as np

from ..._lib import Context, DeviceArray, as_device, check

_p = ctypes.c_void_p


def _np(v):
    if type(v).__module__.startswith("torch"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class Generator:
    def __init__(self, z_dim, channels_img, features_g, ctx=None):
        self.z_dim, self.channels_img, self.features_g = int(z_dim), int(channels_img), int(features_g)
        self._ctx = ctx
        self._handle = None
        self._loaded = False
        self._precision = 1

    # ---- plumbing
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context.get()
        return self._ctx

    def _ensure(self):
        if self._handle is None:
            h = _p()
            check(self.ctx.lib.gl_dcgan_create(self.ctx.handle, self.z_dim, self.channels_img, self.features_g, ctypes.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self.ctx.lib.gl_dcgan_destroy(self._handle)
            except Exception:  # noqa: BLE001
                pass

    # ---- nn.Module-like surface
    def load_state_dict(self, state_dict, strict=True, prefix="gen."):
        h = self._ensure()
        lib = self.ctx.lib
        fg = self.features_g
        chans = [self.z_dim, fg * 16, fg * 8, fg * 4, fg * 2, self.channels_img]
        need = []
        for l in range(4):
            need += [f"{prefix}{l}.0.weight"] + [f"{prefix}{l}.1.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
        need += [f"{prefix}4.weight", f"{prefix}4.bias"]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError("missing keys in state_dict: %s" % missing)
        if strict:
            extra = [k for k in state_dict if k.startswith(prefix) and k not in need and not k.endswith("num_batches_tracked")]
            if extra:
                raise KeyError("unexpected keys in state_dict: %s" % extra)
        for l in range(5):
            w = _np(state_dict[f"{prefix}{l}.0.weight" if l < 4 else f"{prefix}4.weight"])
            if w.shape != (chans[l], chans[l + 1], 4, 4):
                raise ValueError("layer %d weight has shape %s, expected %s" % (l, w.shape, (chans[l], chans[l + 1], 4, 4)))
            check(lib.gl_dcgan_set_conv_weight(h, l, w.ctypes.data_as(_p)))
        for l in range(4):
            g, b, m, v = (_np(state_dict[f"{prefix}{l}.1.{k}"]) for k in ("weight", "bias", "running_mean", "running_var"))
            for a in (g, b, m, v):
                if a.shape != (chans[l + 1],):
                    raise ValueError("layer %d BatchNorm vector has shape %s" % (l, a.shape))
            check(lib.gl_dcgan_set_bn(h, l, g.ctypes.data_as(_p), b.ctypes.data_as(_p), m.ctypes.data_as(_p), v.ctypes.data_as(_p),
                                      ctypes.c_float(1e-5)))
        bias = _np(state_dict[f"{prefix}4.bias"])
        check(lib.gl_dcgan_set_out_bias(h, bias.ctypes.data_as(_p)))
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
        """0 = fp32 MFMA products, 1 (default) = split-fp16 (hi + lo halves, three fp16 MFMAs per product)"""
        check(self.ctx.lib.gl_dcgan_set_precision(self._ensure(), int(mode)))
        self._precision = int(mode)

    def set_chunk(self, images_per_pass):
        check(self.ctx.lib.gl_dcgan_set_chunk(self._ensure(), int(images_per_pass)))

    def _z_device(self, x):
        z = as_device(self.ctx, x, np.float32)
        n = z.shape[0]
        if int(np.prod(z.shape[1:], dtype=np.int64)) != self.z_dim:
            raise ValueError("expected z of shape [N,%d,1,1], got %s" % (self.z_dim, z.shape))
        return z, n

    def forward_device(self, x, want_f32=True, want_u8=False, check_range=True):
        """z -> (f32 DeviceArray [N,C,64,64] or None, u8 DeviceArray or None).  With check_range (default) the call
        synchronises once to make sure the split-fp16 path did not clamp any activation."""
        if not self._loaded:
            raise RuntimeError("Generator: load_state_dict() has not been called")
        z, n = self._z_device(x)
        shape = (n, self.channels_img, 64, 64)
        f32 = self.ctx.empty(shape, np.float32) if want_f32 else None
        u8 = self.ctx.empty(shape, np.uint8) if want_u8 else None
        check(self.ctx.lib.gl_dcgan_forward(self._handle, _p(z.ptr), n, _p(f32.ptr if f32 else 0), _p(u8.ptr if u8 else 0)))
        if check_range and self._precision == 1 and self.ctx.h3_saturations() > 0:
            # an activation left the fp16 range of the split layout: redo this call with fp32 products
            import warnings
            warnings.warn("split-fp16 generator path saturated for these weights; falling back to fp32 MFMA products")
            self.set_precision(0)
            check(self.ctx.lib.gl_dcgan_forward(self._handle, _p(z.ptr), n, _p(f32.ptr if f32 else 0), _p(u8.ptr if u8 else 0)))
        return f32, u8

    def forward(self, x):
        """Generator.forward (model_torch.py:95-96): x [N,z_dim,1,1] -> [N,channels_img,64,64] in (-1,1).
        numpy in -> numpy out; torch in -> torch tensor on the input's device."""
        f32, _ = self.forward_device(x, True, False)
        out = f32.numpy()
        if type(x).__module__.startswith("torch"):
            import torch
            return torch.from_numpy(out).to(x.device)
        return out

    __call__ = forward

    def generate_u8(self, x):
        """the bank the generate branch writes as PNG (train_torch.py:152-174): u8 DeviceArray
        [N,C,64,64], bank index = generation index.  Stays on the device."""
        _, u8 = self.forward_device(x, False, True)
        return u8


class stackGenerators:
    """model_torch.py:99-108: num_generators independent generators; forward(x, i) runs generator i.
    state_dict keys are `gen.{i}.gen.{...}` (the reference generates from i = 0, privDCGAN.py:192)."""

    def __init__(self, z_dim, channels_img, features_g, num_generators, ctx=None):
        self.num_generators = int(num_generators)
        self.gen = [Generator(z_dim, channels_img, features_g, ctx) for _ in range(self.num_generators)]

    def load_state_dict(self, state_dict, strict=True):
        for i, g in enumerate(self.gen):
            g.load_state_dict(state_dict, strict=strict, prefix=f"gen.{i}.gen.")
        return "<All keys matched successfully>"

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):                 # vaegan/sample.py:36 calls .cuda(); the weights already live on the device
        return self

    def train(self, mode=True):              # inference only: BatchNorm always uses its running statistics
        return self

    def forward(self, x, i):
        return self.gen[i](x)

    __call__ = forward
