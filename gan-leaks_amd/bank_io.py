"""Sample-bank files: the on-disk format between the reference's generate branches and fbb.py.

  save_png_bank   gan_models/dcgan/train_torch.py:160-174 (also wgangp/train.py:139-174, pggan/train.py:244-249):
                  `image_{i}.png` in generation order, 8-bit RGB
  load_png_bank   attack_models/fbb.py:133-135 via utils.get_filepaths_from_dir + read_image: files in sorted
                  path-STRING order (image_10 < image_2), so bank index != generation index; `order` gives the map.
"""
from __future__ import annotations

import os

import numpy as np

from .attack_models.utils import _io_workers, get_filepaths_from_dir, read_images_u8_nchw, run_png_workers


def save_png_bank(images_u8, out_dir, prefix="image_", npz_name=None, noise=None, workers=None):
    """images_u8: [N,3,H,W] uint8 (numpy or DeviceArray).  Writes out_dir/{prefix}{i}.png; optionally the
    `fake` / `noise` .npz files the reference also stores (train_torch.py:164-168) next to them.  Large banks are encoded by worker
    processes (_png_worker.py; the reference's writer is a serial loop, train_torch.py:170-174); names and bytes do not depend on it."""
    import PIL.Image
    if hasattr(images_u8, "numpy") and not isinstance(images_u8, np.ndarray):
        images_u8 = images_u8.numpy()
    images_u8 = np.asarray(images_u8)
    if images_u8.dtype != np.uint8 or images_u8.ndim != 4 or images_u8.shape[1] != 3:
        raise ValueError("expected uint8 images [N,3,H,W]")
    os.makedirs(out_dir, exist_ok=True)
    n = len(images_u8)
    workers = _io_workers(workers, n)
    if workers == 1:
        for i, img in enumerate(images_u8):
            PIL.Image.fromarray(img.transpose(1, 2, 0)).save(os.path.join(out_dir, "%s%d.png" % (prefix, i)))
    else:
        import tempfile
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        with tempfile.TemporaryDirectory(dir=shm) as tmp:
            in_file = os.path.join(tmp, "bank.u8")
            np.ascontiguousarray(images_u8).tofile(in_file)
            step = -(-n // workers)
            run_png_workers([("encode", in_file, lo, min(lo + step, n), n, images_u8.shape[2], images_u8.shape[3], out_dir, prefix)
                             for lo in range(0, n, step)])
    if npz_name:
        np.savez(os.path.join(out_dir, npz_name), fake=images_u8.astype(np.float32) / 255.0, **({"noise": noise} if noise is not None else {}))
    return out_dir


def load_png_bank(data_dir, resolution=64, workers=None):
    """-> (uint8 [N,3,res,res] in the order fbb.py sees them, list of paths)"""
    paths = get_filepaths_from_dir(data_dir, ext="png")
    return read_images_u8_nchw(paths, resolution, workers), paths


def generation_order(paths, prefix="image_"):
    """generation index of every file in the loaded order (inverse of the sorted()-on-strings shuffle)"""
    return np.array([int(os.path.basename(p)[len(prefix):-4]) for p in paths], np.int64)
