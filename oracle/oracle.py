"""CPU ORACLE for the GAN-Leaks full-black-box (fbb) attack path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import it; the shipped
package (gan-leaks_amd/) never does and fails loudly without its HIP library.

It restates, in plain numpy, what the reference computes on this path.  Every
function cites the reference file:line (relative to /root/reference) it follows.
Pinning: tests/golden/*.npz were produced by importing the reference's own
modules in the build container (tests/golden/make_golden.py); tests/test_oracle.py
checks this file against them.  Pieces that no reference file can pin here
(torchvision's ToPILImage / VGG16 weights, sklearn internals) are marked
"parity unpinned" where they occur.

Arithmetic contract (shared with the HIP path):
  * images on the 8-bit lattice (everything the reference ever feeds fbb.py,
    attack_models/utils.py:71-82) are compared in EXACT integer arithmetic:
    S = sum_k (uq_k - ub_k)^2, distance = fl32( S * 4 / (255^2 * D) ).
    The reference's fp32 `torch.mean((y-x)**2)` equals this to ~1e-7; where two
    bank images tie in exact arithmetic the smallest index wins, as torch.min does
    (attack_models/fbb.py:86).
  * arbitrary float images are compared with fp32 differences, fp64
    accumulation, and the argmin taken on the fp64 values.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# 8-bit codec  (bank writer + reader)
# --------------------------------------------------------------------------

def dequantize_u8(u8):
    """u8 image -> float32 in [-1,1].

    attack_models/utils.py:82  `img = 2. * (img / 255.) - 1.` (float64), then
    attack_models/fbb.py:134-135 `.float()`.
    """
    u8 = np.asarray(u8)
    return (2.0 * (u8.astype(np.float64) / 255.0) - 1.0).astype(np.float32)


def quantize_to_u8(x, mode="normalize"):
    """generator output in [-1,1] (float32) -> u8, as the generate branches do.

    gan_models/dcgan/train_torch.py:154-158,172: Normalize(mean=-1,std=2) i.e.
    fl32((x - (-1)) / 2), then torchvision ToPILImage on a float tensor =
    `pic.mul(255).byte()` (fp32 multiply, truncation toward zero).
    mode="half": gan_models/pggan/train.py:238  `gen(...) * 0.5 + 0.5`.
    torchvision is not installed here, so the ToPILImage step is restated from its
    documented behaviour: QUANTISATION PARITY UNPINNED.
    """
    x = np.asarray(x, dtype=np.float32)
    if mode == "normalize":
        t = (x + np.float32(1.0)) / np.float32(2.0)
    elif mode == "half":
        t = x * np.float32(0.5) + np.float32(0.5)
    else:
        raise ValueError(mode)
    t = (t * np.float32(255.0)).astype(np.float32)
    # .byte() on values in [0,255]: truncation.  Clamp guards |x|>1 inputs (wraps in torch).
    return np.clip(np.trunc(t), 0, 255).astype(np.uint8)


def is_on_u8_lattice(x):
    """(bool, u8 array): is every float exactly dequantize_u8(u) for some u?"""
    x = np.asarray(x, dtype=np.float32)
    u = np.clip(np.rint((x.astype(np.float64) + 1.0) * 127.5), 0, 255).astype(np.uint8)
    return bool(np.array_equal(dequantize_u8(u), x)), u


# --------------------------------------------------------------------------
# distance + 1-NN  (attack_models/fbb.py:73-88, attack_models/utils.py:153-177)
# --------------------------------------------------------------------------

def n_effective(n, batch_size):
    """attack_models/fbb.py:77  `range(len(syn_imgs) // args.BATCH_SIZE)`: the tail
    n % batch_size bank samples are never compared."""
    return (int(n) // int(batch_size)) * int(batch_size)


def l2_scale(d):
    """distance = S * l2_scale(D): (2/255)^2 / D as a double."""
    return 4.0 / (65025.0 * float(d))


def ssd_u8(bank_u8, query_u8):
    """exact integer sum of squared differences, int64 [N]."""
    b = np.asarray(bank_u8).reshape(len(bank_u8), -1).astype(np.int32)
    q = np.asarray(query_u8).reshape(-1).astype(np.int32)
    d = b - q[None, :]
    return np.einsum("nk,nk->n", d, d, dtype=np.int64)


def knn_l2_u8(bank_u8, queries_u8, batch_size):
    """1-NN of every query over the truncated bank, exact-integer L2.

    Restates custom_knn (attack_models/fbb.py:73-88) with
    Loss('l2').forward (attack_models/utils.py:163,171-177:
    mean((y-x)**2, dim=[1,2,3])) for lattice inputs.
    Returns (dist float32 [Q], idx int64 [Q], ssd int64 [Q]).
    Raises like the reference when no full batch exists (torch.cat of an empty
    list -> ValueError at fbb.py:83; golden knn_empty_error.txt).
    """
    bank_u8 = np.asarray(bank_u8)
    queries_u8 = np.asarray(queries_u8)
    n_eff = n_effective(len(bank_u8), batch_size)
    if n_eff == 0:
        raise ValueError("bank smaller than BATCH_SIZE: no batch to compare (fbb.py:77-83; torch.cat([]) raises ValueError)")
    bank = bank_u8[:n_eff].reshape(n_eff, -1)
    d = bank.shape[1]
    qs = queries_u8.reshape(len(queries_u8), -1)
    idx = np.empty(len(qs), np.int64)
    ssd = np.empty(len(qs), np.int64)
    for i, q in enumerate(qs):
        s = ssd_u8(bank, q)
        j = int(np.argmin(s))          # first occurrence, as torch.min (fbb.py:86)
        idx[i] = j
        ssd[i] = s[j]
    dist = (ssd.astype(np.float64) * l2_scale(d)).astype(np.float32)
    return dist, idx, ssd


def knn_l2_f32(bank, queries, batch_size):
    """general float inputs: fp32 difference, fp64 accumulate, argmin on fp64."""
    bank = np.asarray(bank, np.float32)
    queries = np.asarray(queries, np.float32)
    n_eff = n_effective(len(bank), batch_size)
    if n_eff == 0:
        raise ValueError("bank smaller than BATCH_SIZE: no batch to compare (fbb.py:77-83; torch.cat([]) raises ValueError)")
    b = bank[:n_eff].reshape(n_eff, -1)
    qs = queries.reshape(len(queries), -1)
    d = b.shape[1]
    idx = np.empty(len(qs), np.int64)
    dist = np.empty(len(qs), np.float64)
    for i, q in enumerate(qs):
        diff = (q[None, :] - b).astype(np.float32).astype(np.float64)
        s = np.einsum("nk,nk->n", diff, diff) / d
        j = int(np.argmin(s))
        idx[i] = j
        dist[i] = s[j]
    return dist.astype(np.float32), idx


def custom_knn_literal(bank_f32, query_f32, batch_size):
    """Reference-literal loop in fp32 numpy (one query): per batch
    mean((y-x)**2) in fp32, concatenate, argmin.  attack_models/fbb.py:73-88.
    Used to cross-check the exact path and as a numpy CPU baseline."""
    n = len(bank_f32)
    nb = n // batch_size
    if nb == 0:
        raise ValueError("no full batch")
    out = []
    x_gt = query_f32[None]
    for i in range(nb):
        xb = bank_f32[i * batch_size:(i + 1) * batch_size]
        out.append(np.mean((x_gt - xb) ** 2, axis=(1, 2, 3), dtype=np.float32))
    dist = np.concatenate(out)
    j = int(np.argmin(dist))
    return float(dist[j]), j


# --------------------------------------------------------------------------
# cross-shard merge (not in the reference, which is single-device)
# --------------------------------------------------------------------------

def pack_key(ssd, idx):
    """(S << 32) | global_index : order-preserving, smallest index wins ties."""
    return (np.asarray(ssd, np.int64) << 32) | np.asarray(idx, np.int64)


def unpack_key(key):
    key = np.asarray(key, np.int64)
    return key >> 32, key & 0xFFFFFFFF


# --------------------------------------------------------------------------
# eval_roc metrics (attack_models/eval_roc.py:14-25)
# --------------------------------------------------------------------------

def plot_roc(pos_results, neg_results):
    """labels 0=neg, 1=pos; score = the arrays as given (caller passes -distance,
    eval_roc.py:78).  Returns fpr, tpr, thresholds, auc, ap, precision, matching
    sklearn.metrics.roc_curve(drop_intermediate=True) / roc_auc_score /
    average_precision_score / precision_score(score > -0.14).
    sklearn is third-party; pinned through goldens made with the reference's own
    plot_roc under scikit-learn 1.7.2 (requirements.txt pins 1.2.1).
    """
    pos = np.asarray(pos_results, np.float64).reshape(-1)
    neg = np.asarray(neg_results, np.float64).reshape(-1)
    labels = np.concatenate((np.zeros(len(neg)), np.ones(len(pos))))
    scores = np.concatenate((neg, pos))

    # _binary_clf_curve
    order = np.argsort(scores, kind="mergesort")[::-1]
    s = scores[order]
    y = labels[order]
    distinct = np.where(np.diff(s))[0]
    thr_idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[thr_idx]
    fps = 1 + thr_idx - tps
    thr = s[thr_idx]

    # average precision (step-wise sum over distinct thresholds)
    ps = tps + fps
    precision_c = np.where(ps != 0, tps / np.where(ps == 0, 1, ps), 0.0)
    recall_c = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
    prec_r = np.r_[precision_c[::-1], 1.0]
    rec_r = np.r_[recall_c[::-1], 0.0]
    ap = float(-np.sum(np.diff(rec_r) * prec_r[:-1]))

    # roc_curve: drop collinear points, prepend (0,0) with threshold inf
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps_k, tps_k, thr_k = fps[keep], tps[keep], thr[keep]
    else:
        fps_k, tps_k, thr_k = fps, tps, thr
    tps_k = np.r_[0, tps_k]
    fps_k = np.r_[0, fps_k]
    thr_k = np.r_[np.inf, thr_k]
    fpr = fps_k / fps_k[-1] if fps_k[-1] > 0 else np.full(fps_k.shape, np.nan)
    tpr = tps_k / tps_k[-1] if tps_k[-1] > 0 else np.full(tps_k.shape, np.nan)

    # AUROC: trapezoid over the full curve == Mann-Whitney U with half-credit ties
    tps_f = np.r_[0, tps]
    fps_f = np.r_[0, fps]
    auc = float(np.trapezoid(tps_f / tps_f[-1], fps_f / fps_f[-1]))

    pred = scores > -0.14
    tp = float(np.sum(pred & (labels == 1)))
    pp = float(np.sum(pred))
    precision = tp / pp if pp > 0 else 0.0
    return fpr, tpr, thr_k, auc, ap, precision


# --------------------------------------------------------------------------
# DCGAN / WGAN-GP generator forward (gan_models/dcgan/model_torch.py:75-96,
# gan_models/wgangp/model.py:37-58), eval mode
# --------------------------------------------------------------------------

def conv_transpose2d(x, w, stride, padding, bias=None):
    """x [N,Ci,H,W], w [Ci,Co,kH,kW] (torch ConvTranspose2d layout), float64 math."""
    n, ci, h, wd = x.shape
    _, co, kh, kw = w.shape
    ho = (h - 1) * stride - 2 * padding + kh
    wo = (wd - 1) * stride - 2 * padding + kw
    full = np.zeros((n, co, (h - 1) * stride + kh, (wd - 1) * stride + kw), np.float64)
    x64 = x.astype(np.float64)
    w64 = w.astype(np.float64)
    for ky in range(kh):
        for kx in range(kw):
            contrib = np.einsum("nchw,co->nohw", x64, w64[:, :, ky, kx])
            full[:, :, ky:ky + (h - 1) * stride + 1:stride, kx:kx + (wd - 1) * stride + 1:stride] += contrib
    out = full[:, :, padding:padding + ho, padding:padding + wo]
    if bias is not None:
        out = out + bias.astype(np.float64)[None, :, None, None]
    return out


def batchnorm_eval(x, weight, bias, running_mean, running_var, eps=1e-5):
    """nn.BatchNorm2d in eval mode (gen.eval() at dcgan/train_torch.py:150)."""
    inv = weight.astype(np.float64) / np.sqrt(running_var.astype(np.float64) + eps)
    return (x - running_mean[None, :, None, None]) * inv[None, :, None, None] + bias[None, :, None, None]


def dcgan_generator_forward(sd, z, prefix="gen."):
    """state_dict `sd` (numpy arrays, reference key names) , z [N,nz,1,1] -> [N,3,64,64] f32.
    Layers: 4 x [ConvT(bias=False) -> BN -> ReLU] (k4: s1p0 then s2p1 x3), ConvT k4s2p1 + bias, tanh."""
    x = np.asarray(z, np.float64)
    specs = [(1, 0), (2, 1), (2, 1), (2, 1)]
    for i, (s, p) in enumerate(specs):
        x = conv_transpose2d(x, sd[f"{prefix}{i}.0.weight"], s, p)
        x = batchnorm_eval(x, sd[f"{prefix}{i}.1.weight"], sd[f"{prefix}{i}.1.bias"],
                           sd[f"{prefix}{i}.1.running_mean"], sd[f"{prefix}{i}.1.running_var"])
        x = np.maximum(x, 0.0)
    x = conv_transpose2d(x, sd[f"{prefix}4.weight"], 2, 1, sd[f"{prefix}4.bias"])
    return np.tanh(x).astype(np.float32)
