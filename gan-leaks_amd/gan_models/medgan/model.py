"""Counterpart of the reference's gan_models/medgan/model.py for inference: `Generator` (model.py:44-73) and
`Autoencoder.decode` (model.py:13-41), as the generate branch chains them (gan_models/medgan/train.py:306-312):
    gen_samples = decoder(generator(z));  gen_samples >= 0.5 -> 1.0 else 0.0
Discriminator, encoder training and the CSV dataset are out of scope.

state_dict keys (reference names): Generator: gen_block{1,2}.0.{weight,bias}, gen_block{1,2}.1.{weight,bias,
running_mean,running_var}; Autoencoder: decoder.0.{weight,bias} (encoder.0.* accepted and ignored).
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


def _ret(out, like):
    if type(like).__module__.startswith("torch"):
        import torch
        return torch.from_numpy(out).to(like.device)
    return out


class _Base:
    def __init__(self, z_dim, hidden, input_size, binary, ctx):
        self._ctx = ctx
        self._args = (int(z_dim), int(hidden), int(input_size), 1 if binary else 0)
        self._handle = None
        self._loaded = False

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context.get()
        return self._ctx

    def _ensure(self):
        if self._handle is None:
            h = _p()
            check(self.ctx.lib.gl_medgan_create(self.ctx.handle, *self._args, ctypes.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self.ctx.lib.gl_medgan_destroy(self._handle)
            except Exception:  # noqa: BLE001
                pass

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):                 # vaegan/sample.py:36 calls .cuda(); the weights already live on the device
        return self

    def train(self, mode=True):              # inference only: BatchNorm always uses its running statistics
        return self


class Generator(_Base):
    def __init__(self, z_dim, hidden_size, ctx=None):
        super().__init__(z_dim, hidden_size, 0, False, ctx)
        self.z_dim, self.hidden_size, self.genDim = int(z_dim), int(hidden_size), 128

    def load_state_dict(self, sd, strict=True):
        h = self._ensure()
        for blk, name in enumerate(("gen_block1", "gen_block2")):
            a = [_np(sd[f"{name}.0.weight"]), _np(sd[f"{name}.0.bias"])] + [_np(sd[f"{name}.1.{k}"]) for k in ("weight", "bias", "running_mean", "running_var")]
            check(self.ctx.lib.gl_medgan_set_gen_block(h, blk, *[x.ctypes.data_as(_p) for x in a], ctypes.c_float(0.001)))
        self._loaded = True
        return "<All keys matched successfully>"

    def forward_device(self, x):
        if not self._loaded:
            raise RuntimeError("Generator: load_state_dict() has not been called")
        z = as_device(self.ctx, x, np.float32)
        if len(z.shape) != 2 or z.shape[1] != self.z_dim:
            raise ValueError("expected z of shape [N,%d], got %s" % (self.z_dim, z.shape))
        out = self.ctx.empty((z.shape[0], 128), np.float32)
        check(self.ctx.lib.gl_medgan_generate(self._handle, _p(z.ptr), z.shape[0], _p(out.ptr)))
        return out

    def forward(self, x):
        return _ret(self.forward_device(x).numpy(), x)

    __call__ = forward


class Autoencoder(_Base):
    def __init__(self, input_size, hidden_size, binary=False, ctx=None):
        super().__init__(128, hidden_size, input_size, binary, ctx)
        self.input_size, self.hidden_size, self.binary = int(input_size), int(hidden_size), bool(binary)

    def load_state_dict(self, sd, strict=True):
        w, b = _np(sd["decoder.0.weight"]), _np(sd["decoder.0.bias"])
        if w.shape != (self.input_size, self.hidden_size):
            raise ValueError("decoder.0.weight has shape %s" % (w.shape,))
        check(self.ctx.lib.gl_medgan_set_decoder(self._ensure(), w.ctypes.data_as(_p), b.ctypes.data_as(_p)))
        self._loaded = True
        return "<All keys matched successfully>"

    def decode_device(self, x, want_binary=False):
        if not self._loaded:
            raise RuntimeError("Autoencoder: load_state_dict() has not been called")
        hdn = as_device(self.ctx, x, np.float32)
        n = hdn.shape[0]
        out = self.ctx.empty((n, self.input_size), np.float32)
        binr = self.ctx.empty((n, self.input_size), np.float32) if want_binary else None
        check(self.ctx.lib.gl_medgan_decode(self._handle, _p(hdn.ptr), n, _p(out.ptr), _p(binr.ptr if binr else 0)))
        return out, binr

    def decode(self, x):
        """Autoencoder.decode (model.py:35-36)"""
        return _ret(self.decode_device(x)[0].numpy(), x)

    decoder = decode          # `decoder = autoencoder.decoder; decoder(x)` at medgan/train.py:296,307

    def forward(self, x):
        raise NotImplementedError("the encoder (training side) is out of scope; use decode()")


def generate_synthetic(generator, autoencoder, z):
    """medgan/train.py:303-315: decoder(generator(z)) thresholded at 0.5 -> float32 {0,1} rows"""
    hidden = generator.forward_device(z)
    _, binr = autoencoder.decode_device(hidden, want_binary=True)
    return binr.numpy()
