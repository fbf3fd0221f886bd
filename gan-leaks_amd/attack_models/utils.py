"""Counterpart of the reference's attack_models/utils.py for the fbb path: file helpers and the
distance operator `Loss`.  The plotting helpers of the reference (utils.py:90-148) are not on the
attack path and are not provided.
"""
from __future__ import annotations

import ctypes
import fnmatch
import os

import numpy as np

from .. import _lib
from .._lib import Context, check

_p = ctypes.c_void_p


def check_folder(dir):
    """attack_models/utils.py:19-27"""
    if not os.path.exists(dir):
        os.makedirs(dir)
    return dir


def save_files(save_dir, file_name_list, array_list):
    """attack_models/utils.py:30-40"""
    assert len(file_name_list) == len(array_list)
    for i in range(len(file_name_list)):
        np.save(os.path.join(save_dir, file_name_list[i]), array_list[i], allow_pickle=False)


def get_filepaths_from_dir(data_dir, ext):
    """attack_models/utils.py:43-57: recursive, sorted on the path STRING (image_10 < image_2)."""
    pattern = "*." + ext
    path_list = []
    for d, s, fList in os.walk(data_dir):
        for filename in fList:
            if fnmatch.fnmatch(filename, pattern):
                path_list.append(os.path.join(d, filename))
    return sorted(path_list)


from .._png_worker import read_image_u8  # noqa: E402,F401  (one definition, shared with the worker processes)


def read_image(filepath, resolution=64, cx=89, cy=121):
    """attack_models/utils.py:60-84: image in [-1,1], shape (resolution, resolution, 3), float64."""
    return 2.0 * (read_image_u8(filepath, resolution) / 255.0) - 1.0


def _io_workers(workers, items):
    """worker processes for `items` files: $GANLEAKS_IO_PROCS or up to 16 (the CPU share of one GPU), one per 2048 files"""
    if workers is None:
        workers = int(os.environ.get("GANLEAKS_IO_PROCS", "0")) or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        workers = min(workers, items // 2048)
    return max(1, int(workers))


def run_png_workers(commands):
    """start one `python _png_worker.py ...` per command, wait for all, raise if any failed"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_png_worker.py")
    procs = [subprocess.Popen([sys.executable, script] + [str(a) for a in c], stderr=subprocess.PIPE) for c in commands]
    errs = [(p.wait(), p.stderr.read().decode(errors="replace")) for p in procs]
    bad = [e for rc, e in errs if rc != 0]
    if bad:
        raise RuntimeError("PNG worker failed: " + bad[0][-2000:])


def read_images_u8_nchw(paths, resolution=64, workers=None):
    """all files -> uint8 [N,3,res,res] (the NCHW order fbb.main permutes to, fbb.py:135).  The reference decodes the files one by
    one (fbb.py:133-135: ~20 s per 100k 64x64 PNGs); large lists are split over worker PROCESSES (_png_worker.py) that fill slices
    of one shared file.  Output order = input order; the bytes do not depend on the worker count."""
    n = len(paths)
    workers = _io_workers(workers, n)
    if workers == 1 or any("\n" in p for p in paths):
        out = np.empty((n, 3, resolution, resolution), np.uint8)
        for i, f in enumerate(paths):
            out[i] = read_image_u8(f, resolution).transpose(2, 0, 1)
        return out
    import tempfile
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm) as tmp:
        list_file, out_file = os.path.join(tmp, "paths.txt"), os.path.join(tmp, "bank.u8")
        with open(list_file, "w") as f:
            f.write("\n".join(paths))
        np.memmap(out_file, np.uint8, "w+", shape=(n, 3, resolution, resolution)).flush()
        step = -(-n // workers)
        run_png_workers([("decode", list_file, lo, min(lo + step, n), resolution, out_file) for lo in range(0, n, step)])
        return np.array(np.memmap(out_file, np.uint8, "r", shape=(n, 3, resolution, resolution)))


NCOLS = 5          # images per row in the figure helpers below (attack_models/utils.py:16)


def inverse_transform(imgs):
    """[-1, 1] images -> [0, 1]  (attack_models/utils.py:90-98)"""
    return (imgs + 1.) / 2.


def _image_grid(imgs, titles, path, figure):
    """NCOLS images per row, axes off, optional title per image; the figure helpers of attack_models/utils.py:101-138 (used by the
    white-box / partial-black-box attacks' progress dumps, not by fbb) share this body"""
    import matplotlib
    matplotlib.use('Agg')
    from matplotlib import pyplot
    shown = np.clip(inverse_transform(np.asarray(imgs)), 0., 1.)
    rows = int(np.ceil(len(shown) / float(NCOLS)))
    fig = pyplot.figure(figure)
    for k, picture in enumerate(shown):
        ax = fig.add_subplot(rows, NCOLS, k + 1)
        ax.imshow(picture)
        if titles is not None:
            ax.set_title(titles[k], fontdict={'fontsize': 8, 'color': 'blue'})
        ax.axis('off')
    fig.savefig(path)
    pyplot.close(fig)


def visualize_gt(imgs, save_dir):
    """HWC images in [-1, 1] -> <save_dir>/input.png  (attack_models/utils.py:101-116)"""
    _image_grid(imgs, None, os.path.join(save_dir, 'input.png'), 1)


def visualize_progress(imgs, loss, save_dir, counter):
    """the same with 'loss: %.4f' titles -> <save_dir>/output_<counter>.png  (attack_models/utils.py:119-138)"""
    _image_grid(imgs, ['loss: %.4f' % v for v in loss], os.path.join(save_dir, 'output_%d.png' % counter), 2)


def visualize_samples(img_r01, save_dir):
    """the first 64 HWC images in [0, 1] as an 8 x 8 sheet -> <save_dir>/samples.png  (attack_models/utils.py:141-148)"""
    import matplotlib
    matplotlib.use('Agg')
    from matplotlib import pyplot
    fig = pyplot.figure(figsize=(20, 20))
    for k in range(64):
        ax = fig.add_subplot(8, 8, k + 1)
        ax.imshow(img_r01[k])
        ax.axis('off')
    fig.tight_layout()
    fig.savefig(os.path.join(save_dir, 'samples.png'))
    pyplot.close(fig)


class Loss:
    """attack_models/utils.py:153-177.  Loss(distance, if_norm_reg=False); forward(x_hat, x_gt) -> [B].

    x_hat [B,C,H,W], x_gt [1,C,H,W] (broadcast) or [B,C,H,W].  Sets .loss_lpips, .loss_l2, .vec_loss like
    the reference.  Inputs may be numpy, torch (CPU / ROCm) or DeviceArray; the result is a numpy
    float32 vector, or a torch tensor on the input's device when x_hat is a torch tensor.
    'l2' is exact-integer for images on the 8-bit lattice and fixed-order fp32 otherwise;
    unlike the reference it does not
    build an LPIPS model it never uses (utils.py:157)."""

    def __init__(self, distance, if_norm_reg=False, ctx=None, lpips=None):
        if distance not in ("l2", "l2-lpips"):
            raise ValueError("distance must be 'l2' or 'l2-lpips'")
        self.distance = distance
        self.if_norm_reg = if_norm_reg
        self._ctx = ctx
        self.lpips_model = lpips          # the reference builds ps.PerceptualLoss() here (utils.py:157)
        if distance == "l2":
            print("Use distance: l2")
        else:
            print("Use distance: lpips + l2")
            if self.lpips_model is None:
                from .. import lpips as _lp
                self.lpips_model = _lp.default_model()     # local weight files; FileNotFoundError tells which

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = self.lpips_model.ctx if self.lpips_model is not None else Context.get()
        return self._ctx

    def _finish(self, x_hat, lp, l2):
        if type(x_hat).__module__.startswith("torch"):
            import torch
            l2 = torch.from_numpy(l2).to(x_hat.device)
            if not np.isscalar(lp):
                lp = torch.from_numpy(lp).to(x_hat.device)
        self.loss_lpips = lp
        self.loss_l2 = l2
        self.vec_loss = 0.2 * self.loss_lpips + self.loss_l2     # utils.py:176
        return self.vec_loss

    def forward(self, x_hat, x_gt):
        from ..attack import Bank
        ctx = self.ctx
        if self.distance == "l2-lpips":
            from .. import lpips as _lp
            a = self.lpips_model.features(x_hat)
            g = self.lpips_model.features(x_gt)
            if g.n not in (1, a.n):
                raise ValueError("x_gt must hold 1 image or as many as x_hat (broadcast rule of utils.py:168-169)")
            lp, l2 = _lp.rows_dist(a, g)
            return self._finish(x_hat, lp, l2)
        a = Bank.from_images(x_hat, ctx, keep_u8=True)
        g = Bank.from_images(x_gt, ctx, keep_u8=True, force_kind="f32" if a.kind == "f32" else None)
        if a.d != g.d:
            raise ValueError("image sizes differ")
        if a.kind != g.kind or a.kind == "int":
            # mixed lattices, or whole-number rows (0/1 tables, raw 0..255 floats): x = float(u) has no 4/255^2 scale, so the per-row
            # form runs on the fp32 kernel (its sums of small integers are exact, the result is fl32(S/d))
            a, g = a.as_f32(), g.as_f32()
        out = ctx.empty((max(a.n, 1),), np.float32)
        if a.kind == "u8":
            check(ctx.lib.gl_l2_rows_u8(ctx.handle, _p(a.u8.ptr), a.n, _p(g.u8.ptr), g.n, a.d, _p(out.ptr)))
        else:
            check(ctx.lib.gl_l2_rows_f32(ctx.handle, _p(a.rows_f32.ptr), a.n, _p(g.rows_f32.ptr), g.n, a.d, _p(out.ptr)))
        return self._finish(x_hat, 0.0, out.numpy()[:a.n])

    __call__ = forward
