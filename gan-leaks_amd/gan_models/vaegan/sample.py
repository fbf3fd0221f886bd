"""Counterpart of the reference's `gan_models/vaegan/sample.py` (:10-68), its bank writer for VAEGAN: same flags (`--model_dir`,
`--out_dir`, `--num_samples`), same sampling schedule -- torch.manual_seed(1000), latents drawn on the CPU in batches of 100, ONE
generator forward per batch (so the spectral-norm state advances once per 100 images, as in the reference) -- and the same outputs:

    <out_dir>/generated.npz   noise [N, z_dim] float32,  img_r01 [N, 64, 64, 3] float32 in [0, 1]   (np.savez_compressed)
    <out_dir>/samples.png     10 x 10 grid of the first 100 images

One difference: `<model_dir>/netG.pt` must hold the generator's STATE DICT (torch.save(netG.state_dict(), ...)).  The reference
pickles the whole module object; unpickling executes code from the file, so it is not loaded here (weights_only=True).

    python -m ganleaks_amd.gan_models.vaegan.sample --model_dir runs/vaegan --num_samples 20000
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .train import Generator


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--model_dir', required=True, type=str, help='folder that holds netG.pt (a state dict)')
    parser.add_argument('--out_dir', type=str, help='where generated.npz and samples.png go (default: the model folder)')
    parser.add_argument('--num_samples', type=int, default=20000, help='how many images to draw')
    return parser.parse_args(argv)


def save_image_grid(images, path, drange=(-1, 1), grid=(10, 10)):
    """grid[0] columns x grid[1] rows of NCHW images, `drange` mapped to 0..255 with rounding"""
    import PIL.Image
    n, c, h, w = images.shape
    canvas = np.zeros((grid[1] * h, grid[0] * w, c), np.float32)
    for i in range(min(n, grid[0] * grid[1])):
        y, x = divmod(i, grid[0])
        canvas[y * h:(y + 1) * h, x * w:(x + 1) * w] = images[i].transpose(1, 2, 0)
    scale = 255.0 / (drange[1] - drange[0])
    out = np.clip(np.rint((canvas - drange[0]) * scale), 0, 255).astype(np.uint8)
    PIL.Image.fromarray(out).save(path)


def sample(netG, num_samples, z_dim, batch_size=100, seed=1000):
    import torch
    torch.manual_seed(seed)
    noise, imgs = [], []
    for _ in range(int(np.ceil(num_samples / batch_size))):
        z = torch.randn(batch_size, z_dim, 1, 1)
        imgs.append(np.asarray(netG(z.numpy())))
        noise.append(z.numpy())
    return np.concatenate(noise)[:num_samples].reshape(-1, z_dim), np.concatenate(imgs)[:num_samples]


def main(args):
    import torch
    path = os.path.join(args.model_dir, 'netG.pt')
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:                       # a pickled nn.Module: not loaded (it would execute code from the file)
        raise ValueError("%s must hold netG.state_dict(); a pickled module object is not loaded (%s)" % (path, type(e).__name__)) from e
    z_dim = int(sd["deconv1.module.weight_bar"].shape[0])
    d = int(sd["deconv4.module.weight_bar"].shape[1])
    netG = Generator(z_dim, d)
    netG.load_state_dict(sd)
    netG.eval()
    save_dir = args.model_dir if args.out_dir is None else args.out_dir
    os.makedirs(save_dir, exist_ok=True)
    noise, img = sample(netG, args.num_samples, z_dim)
    save_image_grid(img[:100], os.path.join(save_dir, 'samples.png'), [-1, 1], [10, 10])
    img_r01 = ((img + np.float32(1.0)) / np.float32(2.0)).transpose(0, 2, 3, 1)           # NCHW => NHWC
    np.savez_compressed(os.path.join(save_dir, 'generated.npz'), noise=noise, img_r01=img_r01)
    return os.path.join(save_dir, 'generated.npz')


if __name__ == '__main__':
    main(parse_args())
