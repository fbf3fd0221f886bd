"""The GENERATE branch of the reference's `gan_models/dcgan/train_torch.py` (:138-174) as a command line with the same flags and
output files: load `<saved_model_name>/generator.pth`, draw `num_generated` latents with torch.randn, run the generator on the
device and write

    <PATH_syn_data>/npz_images/<timestamp>/dcgan_synthetic_data.npz   fake  float32 [N,3,64,64] in [0,1]   (Normalize(-1, 2))
    <PATH_syn_data>/npz_noise/<timestamp>/dcgan_noise.npz             noise float32 [N,nz,1,1]
    <PATH_syn_data>/png_images/<timestamp>/image_{i}.png              the 8-bit bank fbb.py reads

Training (`--training`) is outside this repository's scope and is refused.  As in the reference the boolean flags are `type=bool`
(any non-empty string on the command line is True); use --local_config to set them from YAML, as the reference's configs do.

    python -m ganleaks_amd.gan_models.dcgan.train_torch --local_config generate.yaml
"""
from __future__ import annotations

import argparse
import datetime
import os

import numpy as np

from ...bank_io import save_png_bank
from .model_torch import Generator


def parse_arguments(argv=None):
    """train_torch.py:23-49 (the flags the generate branch reads; the training-only ones are accepted and ignored)"""
    p = argparse.ArgumentParser()
    p.add_argument('--lr', type=float, default=2e-4)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--image_size', type=int, default=64)
    p.add_argument('--nc', type=int, default=3, help='number of color channels in the input image, default=3')
    p.add_argument('--nz', type=int, default=100, help='size of the latent z vector, default=100')
    p.add_argument('--ngf', type=int, default=64, help='number of generator filters in first conv layer, default=64')
    p.add_argument('--ndf', type=int, default=64)
    p.add_argument('--input_size', type=int, default=64)
    p.add_argument('--num_epochs', type=int, default=5)
    p.add_argument('--out_size', type=int)
    p.add_argument('--beta1', type=float, default=0.5)
    p.add_argument('--beta2', type=float, default=0.999)
    p.add_argument('--data_path', type=str, default='miniCelebA')
    p.add_argument("--wandb", default=None)
    p.add_argument('--local_config', default=None, help='path to config file')
    p.add_argument('--num_generated', type=int, default=2040, help='number of generated images, default=2040')
    p.add_argument("--PATH", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'model_save', 'dcgan'))
    p.add_argument("--PATH_syn_data", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'syn_data', 'dcgan'),
                   help="Directory to save synthetic data")
    p.add_argument("--save_model", type=bool, default=True)
    p.add_argument("--saved_model_name", type=str, default=None, help="Saved model name")
    p.add_argument("--training", type=bool, default=False, help="Training status (not available here)")
    p.add_argument("--generate", type=bool, default=True, help="Generating Sythetic Data")
    p.add_argument('--ailab', type=bool, default=False)
    return p.parse_args(argv)


def update_args(args, config_dict):
    for key, val in config_dict.items():
        setattr(args, key, val)


def generate(args, noise=None, timestamp=None, pass_images=16384):
    """train_torch.py:138-174.  Returns (png_dir, npz_images_path, npz_noise_path)."""
    import torch
    if args.saved_model_name is None:
        raise AssertionError("Please specify the saved model name")
    if args.wandb is not None:
        raise AssertionError("No need to load anything to wand when only generating synthetic data")
    gen = Generator(args.nz, args.nc, args.ngf)
    gen.load_state_dict(torch.load(os.path.join(args.saved_model_name, "generator.pth"), map_location="cpu", weights_only=True))
    gen.eval()
    if noise is None:
        noise = torch.randn(args.num_generated, args.nz, 1, 1)
    noise_np = noise.numpy() if hasattr(noise, "numpy") else np.asarray(noise, np.float32)
    n = len(noise_np)
    fake = np.empty((n, args.nc, 64, 64), np.float32)
    codes = np.empty((n, args.nc, 64, 64), np.uint8)
    for lo in range(0, n, pass_images):                    # the reference runs all N in one forward (52 GB of activations at 100k)
        f32, u8 = gen.forward_device(noise_np[lo:lo + pass_images], True, True)
        x = f32.numpy()
        fake[lo:lo + len(x)] = (x + np.float32(1.0)) / np.float32(2.0)          # Normalize(mean=-1, std=2)
        codes[lo:lo + len(x)] = u8.numpy()
    timestamp = timestamp or datetime.datetime.now().strftime("_%Y_%m_%d__%H_%M_%S")
    d_img = os.path.join(args.PATH_syn_data, 'npz_images', timestamp)
    d_noise = os.path.join(args.PATH_syn_data, 'npz_noise', timestamp)
    d_png = os.path.join(args.PATH_syn_data, 'png_images', timestamp)
    os.makedirs(d_img, exist_ok=True)
    np.savez(os.path.join(d_img, "dcgan_synthetic_data.npz"), fake=fake)
    os.makedirs(d_noise, exist_ok=True)
    np.savez(os.path.join(d_noise, "dcgan_noise.npz"), noise=noise_np)
    save_png_bank(codes, d_png)
    return d_png, os.path.join(d_img, "dcgan_synthetic_data.npz"), os.path.join(d_noise, "dcgan_noise.npz")


def main(args):
    print(args)
    if args.training:
        raise NotImplementedError("training is outside the scope of this repository (the generate branch needs --training False, set it in "
                                  "the YAML config: argparse's type=bool turns any command-line string into True)")
    if args.generate:
        return generate(args)
    return None


if __name__ == '__main__':
    a = parse_arguments()
    if a.local_config is not None:
        import yaml
        with open(str(a.local_config), "r") as f:
            update_args(a, yaml.safe_load(f))
    main(a)
