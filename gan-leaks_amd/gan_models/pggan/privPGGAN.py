"""The GENERATE branch of the reference's `gan_models/pggan/privPGGAN.py` (:372-437) as a command line with the same flags and output
files: `gen.pth` of a `stackGenerators(nz, in_channels, nc, N_splits)` -> generator 0 at 64 x 64 (`gen(noise, 4, 1, 0)`, :407) ->
`pggan_images.npz`, `pggan_noise.npz`, `image_{i}.png` under PATH_syn_data/{npz_images,npz_noise,png_images}/<sub>, `<sub>` being the
timestamp or `<params_keys>/<params_values>` for every combination of a `--hyperparameter_search` YAML (:246-266) -- the tree
`attack_models/fbb.py --hyperparameter_search` consumes.  The images are mapped to [0,1] by Normalize(-1, 2) here (:408), not by
`* 0.5 + 0.5` as in pggan/train.py.  Training is outside this repository's scope.

    python -m ganleaks_amd.gan_models.pggan.privPGGAN --local_config generate.yaml [--hyperparameter_search sweep.yaml]
"""
from __future__ import annotations

import argparse
import datetime
import os

import numpy as np

from .._generate import refuse_training, run_cli, run_generate, sweep_dirs, sweep_experiments
from .model_torch import stackGenerators


def parse_arguments(argv=None):
    """privPGGAN.py:24-55 (training-only flags are accepted and ignored)"""
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
    p.add_argument('--hyperparameter_search', default=None, help='YAML file of lists: one run per combination')
    p.add_argument("--wandb", default=None)
    p.add_argument("--PATH", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'model_save', 'privPGGAN'), help="root folder of the saved models")
    p.add_argument("--PATH_syn_data", type=str, default=os.path.join(os.getcwd(), 'ersecki-thesis', 'syn_data', 'privPGGA'),
                   help="root folder of the npz_images / npz_noise / png_images outputs")
    p.add_argument("--save_model", type=bool, default=True)
    p.add_argument("--saved_model_name", type=str, default=None, help="folder that holds gen.pth")
    p.add_argument("--training", type=bool, default=False, help="Training status (not available here)")
    p.add_argument("--generate", type=bool, default=True, help="run the generate branch")
    p.add_argument('--ailab', type=bool, default=False)
    p.add_argument("--N_splits", type=int, default=2, help="number of generator / critic pairs")
    p.add_argument('--privacy_ratio', type=float, default=0.5)
    p.add_argument('--disc_epochs', type=int, default=2)
    p.add_argument('--dp_delay', type=int, default=100)
    return p.parse_args(argv)


def generate(args, noise=None, timestamp=None):
    if args.image_size != 64:
        raise ValueError("the generate branch runs gen(noise, 4, 1, 0): 64 x 64 images (privPGGAN.py:407); image_size=%d" % args.image_size)
    model_dir, sub = sweep_dirs(args, timestamp or datetime.datetime.now().strftime("_%Y_%m_%d__%H_%M_%S"))
    stack = stackGenerators(args.nz, args.in_channels, args.nc, int(args.N_splits))

    def forward(g, z):
        # the bytes of Normalize(-1, 2) + ToPILImage (privPGGAN.py:408-420; gl_quantize_f32 mode 0), not of pggan/train.py's x * 0.5 + 0.5
        import ctypes
        from ..._lib import check
        gen = g.gen[0]                                                   # gen(noise, 4, 1, 0), :407
        f32, _ = gen.forward_device(z, 4, 1.0, True, False)
        u8 = gen.ctx.empty(f32.shape, np.uint8)
        check(gen.ctx.lib.gl_quantize_f32(gen.ctx.handle, ctypes.c_void_p(f32.ptr), int(np.prod(f32.shape)), 0, ctypes.c_void_p(u8.ptr)))
        return f32, u8

    return run_generate(args, stack, args.num_generated, forward,
                        lambda x: (x + np.float32(1.0)) / np.float32(2.0), "pggan_images.npz", "pggan_noise.npz", noise, sub, pass_images=2048,
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
