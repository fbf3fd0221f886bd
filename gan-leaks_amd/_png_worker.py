"""PNG decode / encode of one slice of a sample bank, run as a separate PROCESS by attack_models.utils.read_images_u8_nchw and
bank_io.save_png_bank:

    python _png_worker.py decode <list.txt> <lo> <hi> <resolution> <out.u8>     fills rows [lo, hi) of the uint8 [N,3,res,res] file
    python _png_worker.py encode <in.u8>  <lo> <hi> <N> <H> <W> <out_dir> <prefix>

Why processes: PIL's per-file Python overhead holds the GIL, so threads make a bank of small PNGs load slower, not faster; why not
multiprocessing: spawn / forkserver re-import the caller's __main__, and forking a process that has initialised the GPU runtime is
not allowed.  This file imports numpy and PIL only and is started by path, so a worker is up in ~0.3 s.
The reference decodes serially (attack_models/utils.py:60-84, fbb.py:133-135)."""
import os
import sys

import numpy as np


def read_image_u8(filepath, resolution=64):
    """the 8-bit codes read_image decodes (attack_models/utils.py:71-80): PIL open, and a PIL resize
    (default filter) to resolution x resolution when the shape differs.  HWC uint8."""
    import PIL.Image
    img = np.asarray(PIL.Image.open(filepath))
    if img.shape != (resolution, resolution, 3):
        img = np.asarray(PIL.Image.fromarray(img).resize((resolution, resolution)))
    return img


def main(argv):
    mode = argv[1]
    if mode == "decode":
        list_file, lo, hi, res, out_file = argv[2], int(argv[3]), int(argv[4]), int(argv[5]), argv[6]
        with open(list_file) as f:
            paths = f.read().split("\n")
        out = np.memmap(out_file, np.uint8, "r+", shape=(len(paths), 3, res, res))
        for i in range(lo, hi):
            out[i] = read_image_u8(paths[i], res).transpose(2, 0, 1)
        out.flush()
    elif mode == "encode":
        import PIL.Image
        in_file, lo, hi, n, h, w, out_dir, prefix = argv[2], int(argv[3]), int(argv[4]), int(argv[5]), int(argv[6]), int(argv[7]), argv[8], argv[9]
        imgs = np.memmap(in_file, np.uint8, "r", shape=(n, 3, h, w))
        for i in range(lo, hi):
            PIL.Image.fromarray(np.ascontiguousarray(imgs[i].transpose(1, 2, 0))).save(os.path.join(out_dir, "%s%d.png" % (prefix, i)))
    else:
        raise SystemExit("unknown mode %r" % mode)


if __name__ == "__main__":
    main(sys.argv)
