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
import os

import numpy as np

from .._generate import refuse_training, run_cli, run_generate
from .model_torch import Generator


def parse_arguments(argv=None):
    """train_torch.py:23-49 (the flags the generate branch reads; the training-only ones are accepted and ignored)"""
    p = argparse.ArgumentParser()
    p.add_argument('--lr', type=float, default=2e-4)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--image_size', type=int, default=64)
    p.add_argument('--nc', type=int, default=3, help='image channels (3)')
    p.add_argument('--nz', type=int, default=100, help='length of a latent vector')
    p.add_argument('--ngf', type=int, default=64, help='width parameter of the generator (features_g)')
    p.add_argument('--ndf', type=int, default=64)
    p.add_argument('--input_size', type=int, default=64)
    p.add_argument('--num_epochs', type=int, default=5)
    p.add_argument('--out_size', type=int)
    p.add_argument('--beta1', type=float, default=0.5)
    p.add_argument('--beta2', type=float, default=0.999)
    p.add_argument('--data_path', type=str, default='miniCelebA')
    p.add_argument("--wandb", default=None)
    p.add_argument('--local_config', default=None, help='YAML file whose keys override these flags')
    p.add_argument('--num_generated', type=int, default=2040, help='how many images the generate branch draws')
    p.add_argument("--PATH", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'model_save', 'dcgan'))
    p.add_argument("--PATH_syn_data", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'syn_data', 'dcgan'),
                   help="root folder of the npz_images / npz_noise / png_images outputs")
    p.add_argument("--save_model", type=bool, default=True)
    p.add_argument("--saved_model_name", type=str, default=None, help="folder that holds generator.pth")
    p.add_argument("--training", type=bool, default=False, help="Training status (not available here)")
    p.add_argument("--generate", type=bool, default=True, help="run the generate branch")
    p.add_argument('--ailab', type=bool, default=False)
    return p.parse_args(argv)


def update_args(args, config_dict):
    for key, val in config_dict.items():
        setattr(args, key, val)


def generate(args, noise=None, timestamp=None):
    """train_torch.py:138-174.  Returns (png_dir, npz_images_path, npz_noise_path)."""
    return run_generate(args, Generator(args.nz, args.nc, args.ngf), args.num_generated, lambda g, z: g.forward_device(z, True, True),
                        lambda x: (x + np.float32(1.0)) / np.float32(2.0),            # Normalize(mean=-1, std=2)
                        "dcgan_synthetic_data.npz", "dcgan_noise.npz", noise, timestamp)


def main(args):
    print(args)
    refuse_training(args)
    if args.generate:
        return generate(args)
    return None


if __name__ == '__main__':
    run_cli(parse_arguments, main)
