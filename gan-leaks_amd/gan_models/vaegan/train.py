"""Counterpart of the `Generator` class of the reference's gan_models/vaegan/train.py:109-135 (with `SpectralNorm`
and `SelfAttention` from gan_models/vaegan/ops.py:23-75,86-120) for inference.  Encoder, discriminators and
training are out of scope.

    Generator(z_dim, d=64)(x[N,z_dim,1,1]) -> [N,3,64,64]
    4 x [SpectralNorm(ConvTranspose2d + bias) -> BatchNorm2d -> ReLU], SelfAttention after the third, ConvTranspose2d -> tanh

SpectralNorm is STATEFUL at inference in the reference: every forward runs one power iteration that overwrites
`weight_u` / `weight_v` (ops.py:32-44,73-75), also under `.eval()`.  This class reproduces that: each call of
`forward` advances u, v once per wrapped layer (host side, fp32 like the reference), folds 1/sigma, the
ConvTranspose bias and BatchNorm(eval) into the convolution epilogue and runs the stack on the HIP generator.
`state_dict()` returns the advanced u, v so a run can be continued or compared.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ..._lib import Context, as_device, check

_p = ctypes.c_void_p


def _np(v):
    if type(v).__module__.startswith("torch"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class Generator:
    def __init__(self, z_dim, d=64, ctx=None):
        self.z_dim, self.d = int(z_dim), int(d)
        if self.d % 32 != 0:
            raise ValueError("d must be a multiple of 32")
        self._ctx = ctx
        self._handle = None
        self._loaded = False
        self.power_iterations = 1
        self._precision = 1
        self.chans = [self.z_dim, 8 * self.d, 4 * self.d, 2 * self.d, self.d, 3]

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context.get()
        return self._ctx

    def _ensure(self):
        if self._handle is None:
            h = _p()
            # the ConvTranspose stack is the DCGAN one with features_g = d / 2
            check(self.ctx.lib.gl_dcgan_create(self.ctx.handle, self.z_dim, 3, self.d // 2, ctypes.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self.ctx.lib.gl_dcgan_destroy(self._handle)
            except Exception:  # noqa: BLE001
                pass

    def load_state_dict(self, sd, strict=True):
        h = self._ensure()
        lib = self.ctx.lib
        for l in range(4):
            name = "deconv%d" % (l + 1)
            w = _np(sd[name + ".module.weight_bar"])
            if w.shape != (self.chans[l], self.chans[l + 1], 4, 4):
                raise ValueError("%s.module.weight_bar has shape %s" % (name, w.shape))
            u, v, bias = (_np(sd[name + ".module." + k]) for k in ("weight_u", "weight_v", "bias"))
            g, b, mu, var = (_np(sd["%s_bn.%s" % (name, k)]) for k in ("weight", "bias", "running_mean", "running_var"))
            # BatchNorm(eval) and the ConvTranspose bias fold into y = conv(x, w_bar) * (bn_s / sigma) + shift; sigma is advanced on the device
            bn_s = g.astype(np.float64) / np.sqrt(var.astype(np.float64) + 1e-5)
            shift = ((bias.astype(np.float64) - mu) * bn_s + b).astype(np.float32)
            bn_s = bn_s.astype(np.float32)
            check(lib.gl_dcgan_set_spectral_norm(h, l, w.ctypes.data_as(_p), u.ctypes.data_as(_p), v.ctypes.data_as(_p), bn_s.ctypes.data_as(_p),
                                                 shift.ctypes.data_as(_p), int(self.power_iterations)))
        w5, b5 = _np(sd["deconv5.weight"]), _np(sd["deconv5.bias"])
        check(lib.gl_dcgan_set_conv_weight(h, 4, w5.ctypes.data_as(_p)))
        check(lib.gl_dcgan_set_out_bias(h, b5.ctypes.data_as(_p)))
        C = 2 * self.d
        att = [_np(sd["sa1.%s_conv.%s" % (n, k)]) for n in ("query", "key", "value") for k in ("weight", "bias")]
        wq, bq, wk, bk, wv, bv = att
        if wq.shape != (C // 8, C, 1, 1) or wv.shape != (C, C, 1, 1):
            raise ValueError("sa1 conv weights have shapes %s / %s" % (wq.shape, wv.shape))
        gamma = float(_np(sd["sa1.gamma"]).reshape(-1)[0])
        check(lib.gl_dcgan_set_attention(h, *[a.ctypes.data_as(_p) for a in (wq, bq, wk, bk, wv, bv)], ctypes.c_float(gamma)))
        self._loaded = True
        return "<All keys matched successfully>"

    def state_dict(self):
        """the spectral-norm state after the forwards run so far (it lives on the device)"""
        out = {}
        for l in range(4):
            u = np.empty(self.chans[l], np.float32)
            v = np.empty(self.chans[l + 1] * 16, np.float32)
            check(self.ctx.lib.gl_dcgan_get_spectral_state(self._handle, l, u.ctypes.data_as(_p), v.ctypes.data_as(_p)))
            out["deconv%d.module.weight_u" % (l + 1)] = u
            out["deconv%d.module.weight_v" % (l + 1)] = v
        return out

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):                 # vaegan/sample.py:36 calls .cuda(); the weights already live on the device
        return self

    def train(self, mode=True):              # inference only: BatchNorm always uses its running statistics
        return self

    def set_precision(self, mode):
        """1 (default) = split-fp16 convolutions (three fp16 MFMAs per product), 0 = fp32 MFMA; the attention block is fp32 MFMA in both"""
        check(self.ctx.lib.gl_dcgan_set_precision(self._ensure(), int(mode)))
        self._precision = int(mode)

    def forward_device(self, x, want_f32=True, want_u8=False):
        if not self._loaded:
            raise RuntimeError("Generator: load_state_dict() has not been called")
        z = as_device(self.ctx, x, np.float32)
        n = z.shape[0]
        if int(np.prod(z.shape[1:], dtype=np.int64)) != self.z_dim:
            raise ValueError("expected z of shape [N,%d,1,1], got %s" % (self.z_dim, z.shape))
        # gl_dcgan_forward advances the spectral-norm state by one power iteration per call on the device, as the reference does
        shape = (n, 3, 64, 64)
        f32 = self.ctx.empty(shape, np.float32) if want_f32 else None
        u8 = self.ctx.empty(shape, np.uint8) if want_u8 else None
        check(self.ctx.lib.gl_dcgan_forward(self._handle, _p(z.ptr), n, _p(f32.ptr if f32 else 0), _p(u8.ptr if u8 else 0)))
        if self._precision == 1 and self.ctx.h3_saturations() > 0:
            # an activation left the fp16 range of the split layout: redo this call (same spectral-norm state) with fp32 products
            import warnings
            warnings.warn("split-fp16 generator path saturated for these weights; falling back to fp32 MFMA products")
            self.set_precision(0)
            check(self.ctx.lib.gl_dcgan_set_spectral_hold(self._handle, 1))
            try:
                check(self.ctx.lib.gl_dcgan_forward(self._handle, _p(z.ptr), n, _p(f32.ptr if f32 else 0), _p(u8.ptr if u8 else 0)))
            finally:
                check(self.ctx.lib.gl_dcgan_set_spectral_hold(self._handle, 0))
        return f32, u8

    def forward(self, input):
        f32, _ = self.forward_device(input, True, False)
        out = f32.numpy()
        if type(input).__module__.startswith("torch"):
            import torch
            return torch.from_numpy(out).to(input.device)
        return out

    __call__ = forward

    def generate_u8(self, x):
        """8-bit bank of the generator's samples (u8 DeviceArray [N,3,64,64], bank index = latent index), quantised as the image generate
        branches do (gan_models/dcgan/train_torch.py:154-158,172).  The reference's VAEGAN sampler keeps floats (vaegan/sample.py:55-59)
        and has no reader in fbb.py; this is what `attack(queries, GeneratedBank(generator, z))` consumes.  One forward =
        one spectral-norm step, like forward()."""
        return self.forward_device(x, False, True)[1]
