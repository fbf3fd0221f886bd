"""CPU ORACLE for the PGGAN generator forward -- TEST INFRASTRUCTURE ONLY.

Restates in float64 (torch CPU conv2d as the convolution primitive):
  Generator.forward(x, steps, alpha)   gan_models/pggan/model_torch.py:49-88
  WSConv2d     :8-22   conv(x * sqrt(2 / (C_in k^2))) + bias  (the INPUT is scaled, the bias is not)
  PixelNorm    :25-31  x / sqrt(mean_c x^2 + 1e-8)
  ConvBlock    :33-47  [WSConv3x3 -> LeakyReLU(0.2) -> PixelNorm] x 2
  fade_in      :71-72  tanh(alpha * rgb[steps](out) + (1 - alpha) * rgb[steps-1](upscaled))
Pinned by tests/golden/pggan_gen.npz (outputs of the reference's own Generator / stackGenerators).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.asarray(a)).double()


def _ws(sd, name, x, k):
    w = _t(sd[name + ".conv.weight"])
    scale = (2.0 / (w.shape[1] * k * k)) ** 0.5
    return F.conv2d(x * scale, w, None, padding=k // 2) + _t(sd[name + ".bias"]).view(1, -1, 1, 1)


def _pn(x):
    return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def pggan_forward(sd, z, steps, alpha, prefix=""):
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = _pn(_t(z).reshape(len(z), -1, 1, 1))
    x = F.conv_transpose2d(x, _t(sd["initial.1.weight"]), _t(sd["initial.1.bias"]))
    x = F.leaky_relu(x, 0.2)
    x = _pn(F.leaky_relu(_ws(sd, "initial.3", x, 3), 0.2))
    if steps == 0:
        return _ws(sd, "initial_rgb", x, 1).float().numpy()
    out = x
    for step in range(steps):
        up = F.interpolate(out, scale_factor=2, mode="nearest")
        out = _pn(F.leaky_relu(_ws(sd, f"prog_blocks.{step}.conv1", up, 3), 0.2))
        out = _pn(F.leaky_relu(_ws(sd, f"prog_blocks.{step}.conv2", out, 3), 0.2))
    a = _ws(sd, f"rgb_layers.{steps}", out, 1)
    b = _ws(sd, f"rgb_layers.{steps - 1}", up, 1)
    return torch.tanh(alpha * a + (1 - alpha) * b).float().numpy()
