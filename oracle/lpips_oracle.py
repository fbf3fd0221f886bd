"""CPU ORACLE for the LPIPS part of the path -- TEST INFRASTRUCTURE ONLY.

Restates in float64 (torch CPU conv2d is used as the convolution primitive):
  PNetLin.forward        attack_models/lpips_pytorch/models/networks_basic.py:134-181 (v0.1, net-lin, vgg, eval)
  NetLinLayer            networks_basic.py:222-230 (1x1 conv, no bias; dropout is identity in eval)
  vgg16 slices           attack_models/lpips_pytorch/models/pretrained_networks.py:96-134
  normalize_tensor       attack_models/lpips_pytorch/util/util.py:70-73  (eps added AFTER the sqrt)
  Loss('l2-lpips')       attack_models/utils.py:166-176   0.2 * lpips + mean((y-x)^2)
Pinned by tests/golden/lpips_*.npz, produced by the reference's own PNetLin with the vendored lin
weights and a seeded random backbone (torchvision's ImageNet VGG16 weights are not available
offline: real-weight parity is UNPINNED).
"""
import numpy as np
import torch
import torch.nn.functional as F

SHIFT = np.array([-.030, -.088, -.188], np.float32)   # networks_basic.py:115
SCALE = np.array([.458, .448, .450], np.float32)      # networks_basic.py:116
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
VGG_KEYS = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
TAP_AFTER_CONV = {1, 3, 6, 9, 12}                     # relu1_2, 2_2, 3_3, 4_3, 5_3 (0-based conv counter)


def scale_input(x):
    """(in - shift) / scale in fp32, as the reference does (networks_basic.py:135-136)."""
    x = np.asarray(x, np.float32)
    return ((x - SHIFT[None, :, None, None]) / SCALE[None, :, None, None]).astype(np.float32)


def vgg16_taps(sd, x_scaled):
    h = torch.from_numpy(np.asarray(x_scaled)).double()
    taps = []
    ci = 0
    for v in VGG_CFG:
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            continue
        k = VGG_KEYS[ci]
        h = F.relu(F.conv2d(h, torch.from_numpy(sd[f"{k}.weight"]).double(), torch.from_numpy(sd[f"{k}.bias"]).double(), padding=1))
        if ci in TAP_AFTER_CONV:
            taps.append(h)
        ci += 1
    return taps


def normalize_tensor(f, eps=1e-10):
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + eps)


def lpips_matrix(sd, lin, queries_f32, bank_f32):
    """pure LPIPS distances [Q, N] (float64) between images in [-1,1]."""
    tq = [normalize_tensor(t) for t in vgg16_taps(sd, scale_input(queries_f32))]
    tb = [normalize_tensor(t) for t in vgg16_taps(sd, scale_input(bank_f32))]
    out = torch.zeros(len(queries_f32), len(bank_f32), dtype=torch.float64)
    for l in range(5):
        w = torch.from_numpy(np.asarray(lin[l], np.float64)).view(1, -1, 1, 1)
        for qi in range(len(queries_f32)):
            d = (tq[l][qi:qi + 1] - tb[l]) ** 2
            out[qi] += (d * w).sum(dim=1).mean(dim=(1, 2))
    return out.numpy()


def l2_lpips_matrix(sd, lin, queries_f32, bank_f32):
    """0.2 * lpips + mean((y-x)^2)  (attack_models/utils.py:176), [Q, N] float64"""
    lp = lpips_matrix(sd, lin, queries_f32, bank_f32)
    q = np.asarray(queries_f32, np.float64).reshape(len(queries_f32), -1)
    b = np.asarray(bank_f32, np.float64).reshape(len(bank_f32), -1)
    l2 = ((q[:, None, :] - b[None, :, :]) ** 2).mean(axis=2)
    return 0.2 * lp + l2, lp, l2


def knn_l2_lpips(sd, lin, bank_f32, queries_f32, batch_size):
    n_eff = (len(bank_f32) // batch_size) * batch_size
    if n_eff == 0:
        raise ValueError("bank smaller than BATCH_SIZE")
    tot, _, _ = l2_lpips_matrix(sd, lin, queries_f32, bank_f32[:n_eff])
    idx = tot.argmin(axis=1)
    return tot[np.arange(len(tot)), idx].astype(np.float32), idx.astype(np.int64), tot
