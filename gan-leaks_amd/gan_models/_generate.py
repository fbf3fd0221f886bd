"""Shared body of the generate branches (gan_models/dcgan/train_torch.py:138-174, wgangp/train.py:139-174, pggan/train.py:205-257):
generator.pth -> latents -> images on the device -> npz_images / npz_noise / png_images/<timestamp> under PATH_syn_data."""
from __future__ import annotations

import datetime
import os

import numpy as np

from ..bank_io import save_png_bank


def run_generate(args, generator, count, forward, to_unit, npz_images, npz_noise, noise=None, timestamp=None, pass_images=16384,
                 model_file="generator.pth", model_dir=None):
    """forward(generator, z_chunk) -> (f32 DeviceArray in [-1,1], u8 DeviceArray); to_unit(x) maps [-1,1] to the [0,1] floats the
    reference stores.  `timestamp` names the sub-folder of the three outputs (the privGAN sweeps pass `<params_keys>/<params_values>`).
    Returns (png_dir, npz_images_path, npz_noise_path)."""
    import torch
    model_dir = model_dir or args.saved_model_name
    if model_dir is None:
        raise AssertionError("Please specify the saved model name")
    if args.wandb is not None:
        raise AssertionError("No need to load anything to wand when only generating synthetic data")
    generator.load_state_dict(torch.load(os.path.join(model_dir, model_file), map_location="cpu", weights_only=True))
    generator.eval()
    if noise is None:
        noise = torch.randn(count, args.nz, 1, 1)
    noise_np = noise.numpy() if hasattr(noise, "numpy") else np.asarray(noise, np.float32)
    fake, codes = None, None
    for lo in range(0, len(noise_np), pass_images):        # the reference runs all N in one forward (52 GB of activations at 100k, DCGAN)
        f32, u8 = forward(generator, noise_np[lo:lo + pass_images])
        x = f32.numpy()
        if fake is None:
            fake = np.empty((len(noise_np),) + x.shape[1:], np.float32)
            codes = np.empty((len(noise_np),) + x.shape[1:], np.uint8)
        fake[lo:lo + len(x)] = to_unit(x)
        codes[lo:lo + len(x)] = u8.numpy()
    timestamp = timestamp or datetime.datetime.now().strftime("_%Y_%m_%d__%H_%M_%S")
    d_img = os.path.join(args.PATH_syn_data, 'npz_images', timestamp)
    d_noise = os.path.join(args.PATH_syn_data, 'npz_noise', timestamp)
    d_png = os.path.join(args.PATH_syn_data, 'png_images', timestamp)
    os.makedirs(d_img, exist_ok=True)
    np.savez(os.path.join(d_img, npz_images), fake=fake)
    os.makedirs(d_noise, exist_ok=True)
    np.savez(os.path.join(d_noise, npz_noise), noise=noise_np)
    save_png_bank(codes, d_png)
    return d_png, os.path.join(d_img, npz_images), os.path.join(d_noise, npz_noise)


def refuse_training(args):
    if args.training:
        raise NotImplementedError("training is outside the scope of this repository (the generate branch needs `training: false` in the YAML "
                                  "config: argparse's type=bool turns any command-line string into True)")


def run_cli(parse_arguments, main):
    a = parse_arguments()
    if a.local_config is not None:
        import yaml
        with open(str(a.local_config), "r") as f:
            for key, val in yaml.safe_load(f).items():
                setattr(a, key, val)
    return main(a)


def sweep_experiments(args):
    """the hyper-parameter loop of the privGAN scripts (gan_models/dcgan/privDCGAN.py:74-92, pggan/privPGGAN.py:246-266): a YAML file of
    lists -> every combination; sets args.params_keys / args.params_values like the reference and yields once per experiment"""
    import itertools
    import yaml
    if args.hyperparameter_search is not None:
        with open(str(args.hyperparameter_search), "r") as f:
            config = yaml.safe_load(f)
        for key, val in config.items():
            setattr(args, key, val)
        keys, values = zip(*config.items())
        experiments = [dict(zip(keys, v)) for v in itertools.product(*values)]
        args.params_keys = '-'.join(keys)
    else:
        experiments = [{}]
        args.params_values = None
        args.params_keys = None
    for exp in experiments:
        if len(experiments) > 1:
            for key, val in exp.items():
                setattr(args, key, val)
            args.params_values = '-'.join([str(v) for v in exp.values()])
        yield exp


def sweep_dirs(args, timestamp):
    """(model folder, output sub-folder) of one experiment: `<PATH>/<keys>/<values>` and `<keys>/<values>` inside a sweep (where the
    reference's training run of that combination left gen.pth), `saved_model_name` and the timestamp otherwise"""
    if args.params_keys is not None and getattr(args, "params_values", None) is not None:
        sub = os.path.join(args.params_keys, args.params_values)
        own = os.path.join(args.PATH, sub)
        model_dir = own if (args.saved_model_name is None or os.path.exists(os.path.join(own, "gen.pth"))) else args.saved_model_name
        return model_dir, sub
    return args.saved_model_name, timestamp
