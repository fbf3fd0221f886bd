"""Deterministic synthetic inputs (numpy PCG64 only, no torch) shared by tests, bench.py and
the golden-vector generator.  Definitions follow SURVEY.md 8(d) "Synthetic inputs".

There is no dataset or checkpoint in the reference tree (its .gitignore excludes them), so every
measurement and parity test runs on these.
"""
from __future__ import annotations

import numpy as np

DCGAN_CHANNELS = [(100, 1024), (1024, 512), (512, 256), (256, 128)]  # gen.0..gen.3, features_g=64


def lowpass_u8_images(seed, n, res=64, ch=3, cell=8):
    """u8 images [n, ch, res, res] with spatial structure: a coarse random field, bilinearly
    enlarged, plus fine noise.  Spans the full 0..255 range."""
    rng = np.random.default_rng(seed)
    g = res // cell + 1
    coarse = rng.uniform(0.0, 255.0, size=(n, ch, g, g))
    t = (np.arange(res) / cell)
    i0 = np.floor(t).astype(int)
    f = t - i0
    rows = coarse[:, :, i0, :] * (1 - f)[None, None, :, None] + coarse[:, :, i0 + 1, :] * f[None, None, :, None]
    img = rows[:, :, :, i0] * (1 - f)[None, None, None, :] + rows[:, :, :, i0 + 1] * f[None, None, None, :]
    img = img + rng.normal(0.0, 6.0, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def perturb_u8(seed, imgs_u8, sigma=10.0):
    """noisy copies (re-quantised): members that sit close to, not on, a bank sample."""
    rng = np.random.default_rng(seed)
    x = imgs_u8.astype(np.float64) + rng.normal(0.0, sigma, size=imgs_u8.shape)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def attack_case(seed, n_bank, n_pos, n_neg, res=64, sigma=10.0):
    """bank + positive queries (perturbed copies of randomly chosen bank images) + negative queries
    (fresh images).  Returns dict of u8 arrays [.,3,res,res] and `pos_src` (bank index each positive
    was derived from; with the default sigma it is also its nearest neighbour)."""
    bank = lowpass_u8_images(seed * 1000 + 1, n_bank, res)
    rng = np.random.default_rng(seed * 1000 + 2)
    src = rng.integers(0, n_bank, size=n_pos)
    pos = perturb_u8(seed * 1000 + 3, bank[src], sigma)
    neg = lowpass_u8_images(seed * 1000 + 4, n_neg, res)
    return {"bank": bank, "pos": pos, "neg": neg, "pos_src": src}


def lpips_big_case(name):
    """(bank [8,3,H,W], queries [3,3,H,W]) u8 of the BASELINE configs[3]-shaped LPIPS fixtures `lpips_res256` (256 x 256, PGGAN-256's
    image size) and `lpips_res128x256` (non-square): two queries are perturbed bank images, one is fresh"""
    if name == "lpips_res256":
        case = attack_case(41, 8, 2, 1, 256, sigma=20.0)
        return case["bank"], np.concatenate([case["pos"], case["neg"]])
    if name != "lpips_res128x256":
        raise KeyError(name)
    H, W = 128, 256
    bank = np.stack([lowpass_u8_images(4200 + k, 1, 256)[0][:, 64:64 + H, :W] for k in range(8)])
    q = np.concatenate([perturb_u8(43, bank[[6, 1]], 20.0), lowpass_u8_images(4300, 1, 256)[:, :, :H, :W]])
    return np.ascontiguousarray(bank), np.ascontiguousarray(q)


def dcgan_state_dict(seed=1234, z_dim=100, channels_img=3, features_g=64, prefix="gen.", gain=1.0):
    """Random DCGAN/WGAN-GP generator weights under the reference's key names
    (gan_models/dcgan/model_torch.py:75-96): gen.{0..3}.0.weight [Ci,Co,4,4],
    gen.{0..3}.1.{weight,bias,running_mean,running_var,num_batches_tracked}, gen.4.{weight,bias}.
    Conv weights are scaled so activations stay O(1) through the stack and the tanh output uses
    most of (-1,1); BN statistics are randomised so that folding them is exercised."""
    rng = np.random.default_rng(seed)
    fg = features_g
    chans = [(z_dim, fg * 16), (fg * 16, fg * 8), (fg * 8, fg * 4), (fg * 4, fg * 2)]
    sd = {}
    for i, (ci, co) in enumerate(chans):
        taps = 16 if i == 0 else 4          # contributing taps per output pixel
        std = gain * np.sqrt(2.0 / (ci * taps))
        sd[f"{prefix}{i}.0.weight"] = rng.normal(0.0, std, size=(ci, co, 4, 4)).astype(np.float32)
        sd[f"{prefix}{i}.1.weight"] = rng.normal(1.0, 0.1, size=co).astype(np.float32)
        sd[f"{prefix}{i}.1.bias"] = rng.normal(0.0, 0.1, size=co).astype(np.float32)
        sd[f"{prefix}{i}.1.running_mean"] = rng.normal(0.0, 0.1, size=co).astype(np.float32)
        sd[f"{prefix}{i}.1.running_var"] = rng.uniform(0.5, 1.5, size=co).astype(np.float32)
        sd[f"{prefix}{i}.1.num_batches_tracked"] = np.array(1, np.int64)
    ci = fg * 2
    sd[f"{prefix}4.weight"] = rng.normal(0.0, 1.5 * np.sqrt(1.0 / (ci * 4)), size=(ci, channels_img, 4, 4)).astype(np.float32)
    sd[f"{prefix}4.bias"] = rng.normal(0.0, 0.1, size=channels_img).astype(np.float32)
    return sd


def latent(seed, n, z_dim=100):
    """z ~ N(0,1), float32 [n, z_dim, 1, 1] (dcgan/train_torch.py:153)."""
    return np.random.default_rng(seed).standard_normal((n, z_dim)).astype(np.float32).reshape(n, z_dim, 1, 1)


VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]   # features[0:30]
VGG16_CONV_KEYS = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]     # torchvision vgg16().features indices of the convs
LPIPS_CHANNELS = [64, 128, 256, 512, 512]                               # taps relu1_2, 2_2, 3_3, 4_3, 5_3


# imagenet_like=True: how much every layer amplifies its input (second-moment gain of the Kaiming-normal layer times this factor).  The
# cumulative products -- 2, 8, 24, 72, 180, 360, 360, 180, 90, 22, 11, 2.8, 0.35 -- follow the rise and fall of the activation
# magnitudes of a trained VGG16 (O(1) at conv1_1, hundreds in conv3 / conv4, back to O(1) at conv5_3); with the per-channel spread below
# activations span roughly 1e-3 ... 3e4.
VGG16_LAYER_GAINS = [2.0, 4.0, 3.0, 3.0, 2.5, 2.0, 1.0, 0.5, 0.5, 0.25, 0.5, 0.25, 0.125]


def vgg16_state_dict(seed=7, bias_std=0.05, imagenet_like=False):
    """Random VGG16 conv weights under torchvision's `vgg16().features` key names ({idx}.weight
    [Cout,Cin,3,3], {idx}.bias).  The ImageNet weights the reference downloads
    (attack_models/lpips_pytorch/models/pretrained_networks.py:99) are not available offline
    (SURVEY.md D10); Kaiming-normal weights keep activations O(1) through the 13 layers.

    imagenet_like=True gives the weights the DYNAMIC RANGE of a trained network instead (not its features): every layer's rows are scaled by
    VGG16_LAYER_GAINS[l] and by a per-channel factor drawn log-uniformly from [1/16, 16] (normalised to unit mean square, so the layer gain
    stays what the table says), biases scale with the layer's activation level.  Used to check that the split-fp16 VGG16 path neither
    saturates nor loses accuracy when activations are far from O(1) (tests/test_gpu_lpips.py)."""
    rng = np.random.default_rng(seed)
    sd = {}
    cin = 3
    it = iter(VGG16_CONV_KEYS)
    level = 1.0
    for li, v in enumerate([c for c in VGG16_CFG if c != "M"]):
        k = next(it)
        w = rng.normal(0.0, np.sqrt(2.0 / (cin * 9)), size=(v, cin, 3, 3))
        b = rng.normal(0.0, bias_std, size=v)
        if imagenet_like:
            ch = np.exp(rng.uniform(np.log(1 / 16.0), np.log(16.0), size=v))
            ch /= np.sqrt(np.mean(ch ** 2))
            w *= (VGG16_LAYER_GAINS[li] * ch)[:, None, None, None]
            level *= VGG16_LAYER_GAINS[li]
            b *= level * ch
        sd[f"{k}.weight"] = w.astype(np.float32)
        sd[f"{k}.bias"] = b.astype(np.float32)
        cin = v
    return sd


PGGAN_FACTORS = [1, 1, 1, 1, 1 / 2, 1 / 4, 1 / 8, 1 / 16, 1 / 32]      # gan_models/pggan/model_torch.py:6


def pggan_state_dict(seed=4321, z_dim=512, in_channels=512, img_channels=3, prefix=""):
    """Random PGGAN generator weights under the reference's key names (gan_models/pggan/model_torch.py:49-69):
    initial.1.{weight [z,C,4,4], bias}, initial.3.{conv.weight [C,C,3,3], bias}, initial_rgb.{conv.weight, bias},
    prog_blocks.{i}.conv{1,2}.{conv.weight, bias}, rgb_layers.{j}.{conv.weight, bias} with rgb_layers.0 the SAME
    tensors as initial_rgb.  WSConv weights ~ N(0,1) as in the reference's init (model_torch.py:17); biases are
    random instead of zero so that they are exercised."""
    rng = np.random.default_rng(seed)
    C = in_channels
    sd = {}

    def ws(name, cin, cout, k):
        sd[prefix + name + ".conv.weight"] = rng.normal(0.0, 1.0, size=(cout, cin, k, k)).astype(np.float32)
        sd[prefix + name + ".bias"] = rng.normal(0.0, 0.1, size=cout).astype(np.float32)

    sd[prefix + "initial.1.weight"] = rng.normal(0.0, 1.0 / np.sqrt(z_dim), size=(z_dim, C, 4, 4)).astype(np.float32)
    sd[prefix + "initial.1.bias"] = rng.normal(0.0, 0.1, size=C).astype(np.float32)
    ws("initial.3", C, C, 3)
    ws("initial_rgb", C, img_channels, 1)
    sd[prefix + "rgb_layers.0.conv.weight"] = sd[prefix + "initial_rgb.conv.weight"]
    sd[prefix + "rgb_layers.0.bias"] = sd[prefix + "initial_rgb.bias"]
    for i in range(len(PGGAN_FACTORS) - 1):
        cin, cout = int(C * PGGAN_FACTORS[i]), int(C * PGGAN_FACTORS[i + 1])
        ws(f"prog_blocks.{i}.conv1", cin, cout, 3)
        ws(f"prog_blocks.{i}.conv2", cout, cout, 3)
        ws(f"rgb_layers.{i + 1}", cout, img_channels, 1)
    return sd


def medgan_state_dicts(seed=555, input_size=1071):
    """(generator sd, autoencoder sd) under the reference's key names (gan_models/medgan/model.py)."""
    rng = np.random.default_rng(seed)
    g = {}
    for name in ("gen_block1", "gen_block2"):
        g[f"{name}.0.weight"] = rng.normal(0.0, 1.0 / np.sqrt(128), size=(128, 128)).astype(np.float32)
        g[f"{name}.0.bias"] = rng.normal(0.0, 0.1, size=128).astype(np.float32)
        g[f"{name}.1.weight"] = rng.normal(1.0, 0.1, size=128).astype(np.float32)
        g[f"{name}.1.bias"] = rng.normal(0.0, 0.1, size=128).astype(np.float32)
        g[f"{name}.1.running_mean"] = rng.normal(0.0, 0.1, size=128).astype(np.float32)
        g[f"{name}.1.running_var"] = rng.uniform(0.5, 1.5, size=128).astype(np.float32)
        g[f"{name}.1.num_batches_tracked"] = np.array(1, np.int64)
    a = {"encoder.0.weight": rng.normal(0.0, 0.05, size=(128, input_size)).astype(np.float32),
         "encoder.0.bias": rng.normal(0.0, 0.1, size=128).astype(np.float32),
         "decoder.0.weight": rng.normal(0.0, 1.0 / np.sqrt(128), size=(input_size, 128)).astype(np.float32),
         "decoder.0.bias": rng.normal(0.0, 0.3, size=input_size).astype(np.float32)}
    return g, a


def vaegan_state_dict(seed=777, z_dim=100, d=64):
    """Random VAEGAN generator weights under the reference's key names (gan_models/vaegan/train.py:109-123 with
    SpectralNorm's weight_bar / weight_u / weight_v, ops.py:56-71).  gamma is non-zero so attention is exercised."""
    rng = np.random.default_rng(seed)
    ch = [z_dim, 8 * d, 4 * d, 2 * d, d]
    sd = {}
    for l in range(4):
        name = "deconv%d" % (l + 1)
        cin, cout = ch[l], ch[l + 1]
        taps = 16 if l == 0 else 4
        sd[name + ".module.weight_bar"] = rng.normal(0.0, np.sqrt(2.0 / (cin * taps)), size=(cin, cout, 4, 4)).astype(np.float32)
        sd[name + ".module.bias"] = rng.normal(0.0, 0.05, size=cout).astype(np.float32)
        u = rng.normal(0.0, 1.0, size=cin)
        w2 = sd[name + ".module.weight_bar"].reshape(cin, -1).astype(np.float64)
        for _ in range(8):                      # a partly converged power iteration, like a trained checkpoint
            v = w2.T @ u
            v /= np.linalg.norm(v) + 1e-12
            u = w2 @ v
            u /= np.linalg.norm(u) + 1e-12
        sd[name + ".module.weight_u"] = u.astype(np.float32)
        sd[name + ".module.weight_v"] = v.astype(np.float32)
        # spectral normalisation divides by sigma ~ the largest singular value: BatchNorm statistics sized accordingly
        sd[name + "_bn.weight"] = rng.normal(1.0, 0.1, size=cout).astype(np.float32)
        sd[name + "_bn.bias"] = rng.normal(0.0, 0.1, size=cout).astype(np.float32)
        sd[name + "_bn.running_mean"] = rng.normal(0.0, 0.02, size=cout).astype(np.float32)
        sd[name + "_bn.running_var"] = rng.uniform(0.05, 0.15, size=cout).astype(np.float32)
        sd[name + "_bn.num_batches_tracked"] = np.array(1, np.int64)
    sd["deconv5.weight"] = rng.normal(0.0, 0.6 * np.sqrt(1.0 / (d * 4)), size=(d, 3, 4, 4)).astype(np.float32)
    sd["deconv5.bias"] = rng.normal(0.0, 0.1, size=3).astype(np.float32)
    C = 2 * d
    for n, co in (("query", C // 8), ("key", C // 8), ("value", C)):
        sd["sa1.%s_conv.weight" % n] = rng.normal(0.0, 1.0 / np.sqrt(C), size=(co, C, 1, 1)).astype(np.float32)
        sd["sa1.%s_conv.bias" % n] = rng.normal(0.0, 0.1, size=co).astype(np.float32)
    sd["sa1.gamma"] = np.array([0.7], np.float32)
    return sd
