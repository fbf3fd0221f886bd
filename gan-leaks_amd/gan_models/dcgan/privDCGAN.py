"""The GENERATE branch of the reference's `gan_models/dcgan/privDCGAN.py` (:168-215) as a command line with the same flags and output
files: `gen.pth` of a `stackGenerators(nz, nc, ngf, N_splits)` -> generator 0 of the stack (`gen(noise, 0)`) -> `num_generated` images ->

    <PATH_syn_data>/npz_images/<sub>/dcgan_synthetic_data.npz    fake  float32 [N,3,64,64] in [0,1]
    <PATH_syn_data>/npz_noise/<sub>/dcgan_noise.npz              noise float32 [N,nz,1,1]
    <PATH_syn_data>/png_images/<sub>/image_{i}.png               the 8-bit bank fbb.py reads

`<sub>` is the run's timestamp, or `<params_keys>/<params_values>` for every combination of a `--hyperparameter_search` YAML (:74-92):
`png_images/<params_keys>/` is then the directory `attack_models/fbb.py --hyperparameter_search` walks (fbb.py:114-123).  Inside a sweep
gen.pth is taken from `<PATH>/<params_keys>/<params_values>/` (where the reference's training run of that combination saved it) when it
exists there, else from `--saved_model_name`.  Training is outside this repository's scope and is refused.

    python -m ganleaks_amd.gan_models.dcgan.privDCGAN --local_config generate.yaml [--hyperparameter_search sweep.yaml]
"""
from __future__ import annotations

import argparse
import datetime
import os

import numpy as np

from .._generate import refuse_training, run_cli, run_generate, sweep_dirs, sweep_experiments
from .model_torch import stackGenerators


def parse_arguments(argv=None):
    """privDCGAN.py:22-56 (training-only flags are accepted and ignored; `num_generated` comes from the YAML config in the reference)"""
    p = argparse.ArgumentParser()
    p.add_argument('--lr', type=float, default=2e-4)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--image_size', type=int, default=64)
    p.add_argument('--nc', type=int, default=3, help='image channels (3)')
    p.add_argument('--nz', type=int, default=100, help='length of a latent vector')
    p.add_argument('--ngf', type=int, default=64, help='width parameter of the generators (features_g)')
    p.add_argument('--ndf', type=int, default=64)
    p.add_argument('--input_size', type=int, default=64)
    p.add_argument('--num_epochs', type=int, default=5)
    p.add_argument('--disc_epochs', type=int, default=2)
    p.add_argument('--dp_delay', type=int, default=100)
    p.add_argument('--out_size', type=int)
    p.add_argument('--beta1', type=float, default=0.5)
    p.add_argument('--beta2', type=float, default=0.999)
    p.add_argument('--data_path', type=str, default='miniCelebA')
    p.add_argument("--wandb", default=None)
    p.add_argument('--local_config', default=None, help='YAML file whose keys override these flags')
    p.add_argument('--hyperparameter_search', default=None, help='YAML file of lists: one run per combination')
    p.add_argument('--num_images', type=int, default=1000)
    p.add_argument('--num_generated', type=int, default=2040, help='how many images the generate branch draws')
    p.add_argument("--PATH", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'model_save', 'privDCGAN'), help="root folder of the saved models")
    p.add_argument("--PATH_syn_data", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'syn_data', 'privDCGAN'),
                   help="root folder of the npz_images / npz_noise / png_images outputs")
    p.add_argument("--save_model", type=bool, default=True)
    p.add_argument("--saved_model_name", type=str, default=None, help="folder that holds gen.pth")
    p.add_argument("--training", type=bool, default=False, help="Training status (not available here)")
    p.add_argument("--generate", type=bool, default=True, help="run the generate branch")
    p.add_argument("--N_splits", type=int, default=2, help="number of generator / discriminator pairs")
    p.add_argument('--privacy_ratio', type=float, default=0.5)
    p.add_argument('--ailab', type=bool, default=False)
    return p.parse_args(argv)


def generate(args, noise=None, timestamp=None):
    """privDCGAN.py:168-215 for the current experiment.  Returns (png_dir, npz_images_path, npz_noise_path)."""
    model_dir, sub = sweep_dirs(args, timestamp or datetime.datetime.now().strftime("_%Y_%m_%d__%H_%M_%S"))
    stack = stackGenerators(args.nz, args.nc, args.ngf, int(args.N_splits))
    return run_generate(args, stack, args.num_generated, lambda g, z: g.gen[0].forward_device(z, True, True),      # gen(noise, 0), :192
                        lambda x: (x + np.float32(1.0)) / np.float32(2.0), "dcgan_synthetic_data.npz", "dcgan_noise.npz", noise, sub,
                        model_file="gen.pth", model_dir=model_dir)


def main(args):
    out = []
    for _ in sweep_experiments(args):
        print(args)
        refuse_training(args)
        if args.generate:
            out.append(generate(args))
    return out


if __name__ == '__main__':
    run_cli(parse_arguments, main)
