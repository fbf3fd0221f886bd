"""CPU ORACLE for the VAEGAN generator forward -- TEST INFRASTRUCTURE ONLY.

Restates in float64:
  Generator.forward       gan_models/vaegan/train.py:129-135
  SpectralNorm._update_u_v  gan_models/vaegan/ops.py:32-44 (one power iteration per forward, state advances;
                            height = weight.shape[0] = C_in of the ConvTranspose2d weight [C_in, C_out, 4, 4])
  SelfAttention.forward   gan_models/vaegan/ops.py:101-120
Pinned by tests/golden/vaegan_gen.npz (two consecutive forwards of the reference's own Generator).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.asarray(a)).double()


def _l2n(v, eps=1e-12):
    return v / (v.norm() + eps)


class VaeganOracle:
    def __init__(self, sd):
        self.sd = {k: _t(v) for k, v in sd.items() if not k.endswith("num_batches_tracked")}

    def _sn_weight(self, name):
        w = self.sd[name + ".module.weight_bar"]
        h = w.shape[0]
        u, v = self.sd[name + ".module.weight_u"], self.sd[name + ".module.weight_v"]
        w2 = w.reshape(h, -1)
        v = _l2n(w2.t().mv(u))
        u = _l2n(w2.mv(v))
        self.sd[name + ".module.weight_u"], self.sd[name + ".module.weight_v"] = u, v
        sigma = u.dot(w2.mv(v))
        return w / sigma

    def _bn(self, x, name):
        s = self.sd
        return F.batch_norm(x, s[name + ".running_mean"], s[name + ".running_var"], s[name + ".weight"], s[name + ".bias"], False, 0.0, 1e-5)

    def _attention(self, x):
        s = self.sd
        b, c, w, h = x.shape
        q = F.conv2d(x, s["sa1.query_conv.weight"], s["sa1.query_conv.bias"]).view(b, -1, w * h).permute(0, 2, 1)
        k = F.conv2d(x, s["sa1.key_conv.weight"], s["sa1.key_conv.bias"]).view(b, -1, w * h)
        att = torch.softmax(torch.bmm(q, k), dim=-1)
        v = F.conv2d(x, s["sa1.value_conv.weight"], s["sa1.value_conv.bias"]).view(b, -1, w * h)
        out = torch.bmm(v, att.permute(0, 2, 1)).view(b, c, w, h)
        return s["sa1.gamma"] * out + x

    def forward(self, z):
        s = self.sd
        x = _t(z).reshape(len(z), -1, 1, 1)
        x = F.relu(self._bn(F.conv_transpose2d(x, self._sn_weight("deconv1"), s["deconv1.module.bias"], 1, 0), "deconv1_bn"))
        x = F.relu(self._bn(F.conv_transpose2d(x, self._sn_weight("deconv2"), s["deconv2.module.bias"], 2, 1), "deconv2_bn"))
        x = self._attention(F.relu(self._bn(F.conv_transpose2d(x, self._sn_weight("deconv3"), s["deconv3.module.bias"], 2, 1), "deconv3_bn")))
        x = F.relu(self._bn(F.conv_transpose2d(x, self._sn_weight("deconv4"), s["deconv4.module.bias"], 2, 1), "deconv4_bn"))
        return torch.tanh(F.conv_transpose2d(x, s["deconv5.weight"], s["deconv5.bias"], 2, 1)).float().numpy()
