#!/usr/bin/env python3
"""CPU study for VERDICT r2 "Next 1": would Winograd F(2x2,3x3) / F(2x2,2x2) on the split-fp16 operands keep the accuracy of the
direct split-fp16 convolution (csrc/gl_conv_h3.hip)?  Emulates both in numpy with the device's arithmetic:

  operands   x*S = hi + lo, hi = fp16(x*S), lo = fp16(x*S - hi)             (gl_conv_h3.hip `split_store`)
  product    hi*hi + hi*lo + lo*hi, fp32 accumulation over C_in (and taps)   (three v_mfma_f32_16x16x32_f16)
  Winograd   U = G g G^T in double, then split; V = B^T d B in fp32 from the stored x = hi + lo, then split; 16 (9) GEMMs over C_in;
             Y = A^T M A on the fp32 accumulators

against an fp64 convolution of the same stored inputs.  Prints max / rms error of both forms and their ratio; the kill criterion of
the review is ratio > 2.

    python tools/winograd_numerics.py            # a table on stdout (json lines)
"""
import json

import numpy as np

f16, f32, f64 = np.float16, np.float32, np.float64


def split(x, S):
    xs = (x.astype(f32) * f32(S)).astype(f32)
    hi = xs.astype(f16)
    lo = (xs - hi.astype(f32)).astype(f16)
    return hi.astype(f32), lo.astype(f32)


def gemm3(ah, al, bh, bl):
    """[M,K] x [K,N] with the three-product split form, fp32 accumulate"""
    return (ah @ bh + ah @ bl + al @ bh).astype(f32)


def pow2_scale(x, target=2048.0):
    m = float(np.abs(x).max())
    if m == 0:
        return 1.0
    return float(2.0 ** np.floor(np.log2(target / m)))


# --- F(2x2, 3x3) -------------------------------------------------------------------------------------------------------------
BT3 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], f64)
G3 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], f64)
AT3 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], f64)
# --- F(2x2, 2x2): y0 = d0 g0 + d1 g1, y1 = d1 g0 + d2 g1 ---------------------------------------------------------------------
BT2 = np.array([[1, -1, 0], [0, 1, 0], [0, -1, 1]], f64)
G2 = np.array([[1, 0], [1, 1], [0, 1]], f64)
AT2 = np.array([[1, 1, 0], [0, 1, 1]], f64)


def conv_direct_f64(x, w):
    """x [C,H,W] (already padded), w [O,C,r,r] -> [O,H-r+1,W-r+1], correlation"""
    O, C, r, _ = w.shape
    H, W = x.shape[1] - r + 1, x.shape[2] - r + 1
    out = np.zeros((O, H, W), f64)
    for a in range(r):
        for b in range(r):
            out += np.einsum("oc,chw->ohw", w[:, :, a, b].astype(f64), x[:, a:a + H, b:b + W].astype(f64))
    return out


def conv_direct_split(x, w, Sx, Sw):
    O, C, r, _ = w.shape
    H, W = x.shape[1] - r + 1, x.shape[2] - r + 1
    xh, xl = split(x, Sx)
    wh, wl = split(w, Sw)
    # one GEMM over K = taps * C, as the device does (fp32 accumulation across everything)
    cols_h = np.concatenate([xh[:, a:a + H, b:b + W].reshape(C, -1) for a in range(r) for b in range(r)], 0)
    cols_l = np.concatenate([xl[:, a:a + H, b:b + W].reshape(C, -1) for a in range(r) for b in range(r)], 0)
    Wh = np.concatenate([wh[:, :, a, b] for a in range(r) for b in range(r)], 1)
    Wl = np.concatenate([wl[:, :, a, b] for a in range(r) for b in range(r)], 1)
    acc = gemm3(Wh, Wl, cols_h, cols_l)
    return (acc.astype(f64) / (Sx * Sw)).reshape(O, H, W)


def conv_winograd_split(x, w, Sx, Sw, BT, G, AT, vgrow):
    """vgrow: power of two by which V values may exceed x (4 for F(2,3): sums of 4 inputs; 4 for F(2,2)) -> V is split at Sx/vgrow"""
    O, C, r, _ = w.shape
    m = 2
    t = m + r - 1
    H, W = x.shape[1] - r + 1, x.shape[2] - r + 1
    assert H % m == 0 and W % m == 0
    U = np.einsum("ia,ocab,jb->ijoc", G, w.astype(f64), G)                    # [t,t,O,C] in double
    Su = Sw
    Uh, Ul = split(U.astype(f32), Su)
    xs = x.astype(f32)                                                       # stored value hi + lo, exact in fp32
    # tiles [C, th, tw, t, t]
    th, tw = H // m, W // m
    d = np.stack([np.stack([xs[:, m * i:m * i + t, m * j:m * j + t] for j in range(tw)], 1) for i in range(th)], 1)
    BTf = BT.astype(f32)
    # fp32 adds in two stages, like a VALU implementation: rows then columns
    tmp = np.einsum("ia,cyxab->cyxib", BTf, d).astype(f32)
    V = np.einsum("cyxib,jb->cyxij", tmp, BTf).astype(f32)
    Sv = Sx / vgrow
    Vh, Vl = split(V, Sv)
    M = np.zeros((t, t, O, th * tw), f32)
    for i in range(t):
        for j in range(t):
            M[i, j] = gemm3(Uh[i, j], Ul[i, j], Vh[:, :, :, i, j].reshape(C, -1), Vl[:, :, :, i, j].reshape(C, -1))
    ATf = AT.astype(f32)
    tmp = np.einsum("ai,ijop->ajop", ATf, M).astype(f32)
    Y = np.einsum("ajop,bj->abop", tmp, ATf).astype(f32)                    # [m,m,O,tiles]
    out = Y.reshape(m, m, O, th, tw).transpose(2, 3, 0, 4, 1).reshape(O, H, W)
    return out.astype(f64) / (Su * Sv)


def study(name, x, w, kind):
    r = w.shape[2]
    Sx, Sw = pow2_scale(x), pow2_scale(w)
    # the stored activation is the split value
    xh, xl = split(x, Sx)
    xst = ((xh.astype(f64) + xl.astype(f64)) / Sx)
    ref = conv_direct_f64(xst, w)
    d = conv_direct_split(xst, w, Sx, Sw)
    if kind == 3:
        wg = conv_winograd_split(xst, w, Sx, Sw, BT3, G3, AT3, 4.0)
    else:
        wg = conv_winograd_split(xst, w, Sx, Sw, BT2, G2, AT2, 4.0)
    # an fp32 direct convolution for scale (what cuDNN's non-Winograd algorithms give the reference)
    O, C = w.shape[:2]
    H, W = ref.shape[1:]
    cols = np.concatenate([xst[:, a:a + H, b:b + W].reshape(C, -1) for a in range(r) for b in range(r)], 0).astype(f32)
    Wm = np.concatenate([w[:, :, a, b] for a in range(r) for b in range(r)], 1).astype(f32)
    f = (Wm @ cols).reshape(O, H, W).astype(f64)
    rms = float(np.sqrt((ref ** 2).mean()))
    res = {"case": name, "out_rms": rms}
    for k, v in (("direct_split", d), ("winograd_split", wg), ("direct_fp32", f)):
        e = np.abs(v - ref)
        res[k + "_max"] = float(e.max())
        res[k + "_rms"] = float(np.sqrt((e ** 2).mean()))
    res["ratio_max"] = res["winograd_split_max"] / res["direct_split_max"]
    res["ratio_rms"] = res["winograd_split_rms"] / res["direct_split_rms"]
    print(json.dumps(res))
    return res


def main():
    rng = np.random.default_rng(0)
    # VGG-like: ReLU activations, Kaiming weights (synth.vgg16_state_dict), pad 1
    for C, O, HW in ((64, 64, 32), (128, 64, 32), (256, 64, 16), (512, 64, 8)):
        x = np.maximum(rng.standard_normal((C, HW, HW)), 0).astype(f32)
        x = np.pad(x, ((0, 0), (1, 1), (1, 1)))
        w = (rng.standard_normal((O, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(f32)
        study("vgg3x3 C=%d %dx%d" % (C, HW, HW), x, w, 3)
    # activations with a wide dynamic range (ImageNet-like: a few large channels)
    C, O, HW = 256, 64, 16
    gains = np.exp(rng.uniform(np.log(1e-2), np.log(30.0), C)).astype(f32)
    x = np.maximum(rng.standard_normal((C, HW, HW)), 0).astype(f32) * gains[:, None, None]
    x = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    w = (rng.standard_normal((O, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(f32)
    study("vgg3x3 wide-range C=256", x, w, 3)
    # PGGAN-like: pixel-normalised leaky activations (both signs), N(0,1) weights * sqrt(2/(9C))
    C, O, HW = 512, 64, 16
    x = rng.standard_normal((C, HW, HW)).astype(f32)
    x = np.where(x > 0, x, 0.2 * x)
    x = (x / np.sqrt((x ** 2).mean(0, keepdims=True) + 1e-8)).astype(f32)
    x = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    w = (rng.standard_normal((O, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(f32)
    study("pggan3x3 C=512", x, w, 3)
    # DCGAN ConvT k4s2p1 phase = 2x2-tap convolution of ReLU(BN) activations, weights N(0, 0.02) at three scales
    for scale in (0.05, 1.0, 4.0):
        for C, O, HW in ((1024, 64, 4), (512, 64, 8), (256, 64, 16)):
            x = np.maximum(rng.standard_normal((C, HW, HW)) * 0.5 + 0.1, 0).astype(f32)
            x = np.pad(x, ((0, 0), (1, 0), (1, 0)))                 # phase (0,0) of ConvT k4s2p1: taps at (i-1..i, j-1..j); HW x HW outputs
            w = (rng.standard_normal((O, C, 2, 2)) * 0.02 * scale).astype(f32)
            study("dcgan2x2 scale=%g C=%d %dx%d" % (scale, C, HW, HW), x, w, 2)


if __name__ == "__main__":
    main()


def winograd_fp32_only(x, w, BT, G, AT):
    """the same Winograd with fp32 operands (no split): what an fp32 Winograd convolution (cuDNN's, under the reference) gives"""
    O, C, r, _ = w.shape
    m, t = 2, 2 + r - 1
    H, W = x.shape[1] - r + 1, x.shape[2] - r + 1
    U = np.einsum("ia,ocab,jb->ijoc", G, w.astype(f64), G).astype(f32)
    xs = x.astype(f32)
    th, tw = H // m, W // m
    d = np.stack([np.stack([xs[:, m * i:m * i + t, m * j:m * j + t] for j in range(tw)], 1) for i in range(th)], 1)
    BTf = BT.astype(f32)
    V = np.einsum("cyxib,jb->cyxij", np.einsum("ia,cyxab->cyxib", BTf, d).astype(f32), BTf).astype(f32)
    M = np.zeros((t, t, O, th * tw), f32)
    for i in range(t):
        for j in range(t):
            M[i, j] = U[i, j] @ V[:, :, :, i, j].reshape(C, -1)
    ATf = AT.astype(f32)
    Y = np.einsum("ajop,bj->abop", np.einsum("ai,ijop->ajop", ATf, M).astype(f32), ATf).astype(f32)
    return Y.reshape(m, m, O, th, tw).transpose(2, 3, 0, 4, 1).reshape(O, H, W).astype(f64)
