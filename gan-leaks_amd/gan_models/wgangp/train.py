"""The GENERATE branch of the reference's `gan_models/wgangp/train.py` (:139-174) as a command line with the same flags and output
files (`wgangp_synthetic_data.npz`, `wgangp_noise.npz`, `image_{i}.png` under PATH_syn_data/{npz_images,npz_noise,png_images}/<timestamp>).
As in the reference the number of generated images is `--batch_size` (wgangp/train.py:155 draws `torch.randn(args.batch_size, ...)`).
Training / resume / finetuning / evaluate are outside this repository's scope.

    python -m ganleaks_amd.gan_models.wgangp.train --local_config generate.yaml
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .._generate import refuse_training, run_cli, run_generate
from .model import Generator


def parse_arguments(argv=None):
    """wgangp/train.py:24-53 (training-only flags are accepted and ignored)"""
    p = argparse.ArgumentParser()
    p.add_argument('--lr', type=float, default=0.0004)
    p.add_argument('--batch_size', type=int, default=64, help='also the number of images the generate branch draws')
    p.add_argument('--image_size', type=int, default=64)
    p.add_argument('--nc', type=int, default=3)
    p.add_argument('--nz', type=int, default=100)
    p.add_argument('--ngf', type=int, default=64)
    p.add_argument('--ndf', type=int, default=64)
    p.add_argument('--critic_iter', type=int, default=5)
    p.add_argument('--lambda_gp', type=float, default=10)
    p.add_argument('--num_epochs', type=int, default=5)
    p.add_argument('--out_size', type=int)
    p.add_argument('--beta1', type=float, default=0.0)
    p.add_argument('--beta2', type=float, default=0.9)
    p.add_argument('--dataroot', type=str, default=1)
    p.add_argument('--data_name', type=str, default='miniCelebA')
    p.add_argument('--local_config', default=None, help='YAML file whose keys override these flags')
    p.add_argument("--wandb", default=None)
    p.add_argument("--PATH", type=str, default=os.path.join(os.getcwd(), 'model_save', 'wgangp'))
    p.add_argument("--PATH_syn_data", type=str, default=os.path.join(os.getcwd(), 'syn_data', 'wgangp'), help="root folder of the npz_images / npz_noise / png_images outputs")
    p.add_argument("--save_model", type=bool, default=True)
    p.add_argument("--saved_model_name", type=str, default=None, help="folder that holds generator.pth")
    p.add_argument("--training", type=bool, default=False, help="Training status (not available here)")
    p.add_argument("--resume", type=bool, default=False)
    p.add_argument("--finetuning", type=bool, default=False)
    p.add_argument("--generate", type=bool, default=True, help="run the generate branch")
    p.add_argument("--evaluate", type=bool, default=False)
    return p.parse_args(argv)


def generate(args, noise=None, timestamp=None):
    return run_generate(args, Generator(args.nz, args.nc, args.ngf), args.batch_size, lambda g, z: g.forward_device(z, True, True),
                        lambda x: (x + np.float32(1.0)) / np.float32(2.0), "wgangp_synthetic_data.npz", "wgangp_noise.npz", noise, timestamp)


def main(args):
    print(args)
    refuse_training(args)
    if args.resume or args.finetuning or args.evaluate:
        raise NotImplementedError("resume / finetuning / evaluate are outside the scope of this repository")
    if args.generate:
        return generate(args)
    return None


if __name__ == '__main__':
    run_cli(parse_arguments, main)
