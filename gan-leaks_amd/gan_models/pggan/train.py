"""The GENERATE branch of the reference's `gan_models/pggan/train.py` (:205-257) as a command line with the same flags and output
files: `gen(noise, 4, 1) * 0.5 + 0.5` for `num_generated` latents -> `pggan_images.npz`, `pggan_noise.npz`, `image_{i}.png` under
PATH_syn_data/{npz_images,npz_noise,png_images}/<timestamp>.  Training is outside this repository's scope.

    python -m ganleaks_amd.gan_models.pggan.train --local_config generate.yaml
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .._generate import refuse_training, run_cli, run_generate
from .model_torch import Generator


def parse_arguments(argv=None):
    """pggan/train.py:24-48 (training-only flags are accepted and ignored)"""
    p = argparse.ArgumentParser()
    p.add_argument('--num_epochs', type=int, default=30)
    p.add_argument('--lr', type=float, default=0.0002)
    p.add_argument('--batch_size', type=list, default=[16, 16, 16, 16, 16])
    p.add_argument('--image_size', type=int, default=64, help='the height / width of the generated images (the branch runs 4 steps: 64)')
    p.add_argument('--nc', type=int, default=3)
    p.add_argument('--nz', type=int, default=256, help='length of a latent vector')
    p.add_argument('--in_channels', type=int, default=256, help='channel count of the first block')
    p.add_argument('--start_img_size', type=int, default=4)
    p.add_argument('--num_generated', type=int, default=10000, help='how many images the generate branch draws')
    p.add_argument('--lambda_gp', type=float, default=10)
    p.add_argument('--data_path', type=str, default='miniCelebA')
    p.add_argument('--local_config', default=None, help='YAML file whose keys override these flags')
    p.add_argument("--wandb", default=None)
    p.add_argument("--PATH", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'model_save', 'dcgan'))
    p.add_argument("--PATH_syn_data", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'syn_data', 'dcgan'), help="root folder of the npz_images / npz_noise / png_images outputs")
    p.add_argument("--save_model", type=bool, default=True)
    p.add_argument("--saved_model_name", type=str, default=None, help="folder that holds generator.pth")
    p.add_argument("--training", type=bool, default=False, help="Training status (not available here)")
    p.add_argument("--generate", type=bool, default=True, help="run the generate branch")
    p.add_argument('--ailab', type=bool, default=False)
    return p.parse_args(argv)


def generate(args, noise=None, timestamp=None):
    if args.image_size != 64:
        raise ValueError("the generate branch runs gen(noise, 4, 1): 64 x 64 images (pggan/train.py:236); image_size=%d" % args.image_size)
    return run_generate(args, Generator(args.nz, args.in_channels, args.nc), args.num_generated, lambda g, z: g.forward_device(z, 4, 1.0, True, True),
                        lambda x: x * np.float32(0.5) + np.float32(0.5), "pggan_images.npz", "pggan_noise.npz", noise, timestamp, pass_images=2048)


def main(args):
    print(args)
    refuse_training(args)
    if args.generate:
        return generate(args)
    return None


if __name__ == '__main__':
    run_cli(parse_arguments, main)
