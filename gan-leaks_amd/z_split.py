"""Counterpart of the reference's `z_split.py` (dataset preparation for the attack; host-only file work, no GPU): from CelebA's
aligned images and the identity annotations build

    output_dir1   the POSITIVE query set: images of identities with exactly `num_same_id` pictures ("private"), 128 x 128 centre crop
    output_dir2   the NEGATIVE query set: images of identities with fewer pictures ("public"), same crop
    output_dir0   the GAN training set: every positive image three times -- the crop, a random 128 x 128 crop (`_a1`) and the
                  mirrored crop (`_a2`)

with the reference's flags, file names, selection order (`num_images // 3` per class, identities in annotation order), crop centre
(89, 121) and use of numpy's GLOBAL random state for the `_a1` crops (z_split.py:10-139).  Like the reference it DELETES the three
output directories first.
"""
from __future__ import annotations

import argparse
import os
import shutil

import numpy as np


def parse_arguments(argv=None):
    """z_split.py:10-28 (same flags and defaults)"""
    p = argparse.ArgumentParser()
    p.add_argument('--num_images', type=int, default=10020, help='images taken in total: a third become positive queries, a third negative ones')
    p.add_argument('--identity_annotations', type=str, default='data/identities_ann.txt', help='text file with one `<identity> <image file>` pair per line')
    p.add_argument('--input_dir', type=str, default='data/img_align_celeba', help='folder holding the aligned 218 x 178 CelebA pictures')
    p.add_argument('--output_dir0', type=str, default='data/train', help='where the GAN training set (crop, random crop, mirrored crop) is written')
    p.add_argument('--output_dir1', type=str, default='data/celebAhuge_positive', help='where the member (positive) query crops are written')
    p.add_argument('--output_dir2', type=str, default='data/celebAhuge_negative', help='where the non-member (negative) query crops are written')
    p.add_argument('--img_size', type=int, default=64, help='kept for compatibility; the crops are always 128 x 128')
    p.add_argument('--local_config', default=None, help='YAML file whose keys override these flags')
    p.add_argument('--num_same_id', type=int, default=30, help='an identity is private when it has exactly this many pictures, public when it has fewer')
    return p.parse_args(argv)


def read_identities(path):
    """each line holds two fields; the FIRST is the grouping key and the second is collected under it, in file order -- so with the
    reference's `identities_ann.txt` the file must list `<identity> <image>` (z_split.py:35-39)"""
    groups = {}
    with open(path) as f:
        for line in f:
            key, member = line.strip().split()
            groups.setdefault(key, []).append(member)
    return groups


def select_images(groups, num_images, num_same_id):
    """(private_images, public_images): z_split.py:42-66"""
    if num_images % 30 != 0:
        raise AssertionError('num_images must be divisible by 30!, either 510, 1020, 2040, 10002, 20001')
    want = num_images // 3

    def take(keys):
        out = []
        for k in keys:
            if len(out) >= want:
                break
            out += groups[k][:want - len(out)]
        return out

    private = take([k for k in groups if len(groups[k]) == num_same_id])
    public = take([k for k in groups if len(groups[k]) < num_same_id])
    if set(private) & set(public):
        raise AssertionError('The two lists are not disjoint!')
    return private, public


def crops(img, cx=89, cy=121):
    """(centre crop, random crop, mirrored centre crop) of one aligned 218 x 178 picture (z_split.py:112-133).  The random crop draws
    its column offset first, then its row offset, from numpy's global state, as the reference does."""
    if img.shape != (218, 178, 3):
        raise AssertionError("expected a 218 x 178 RGB image, got %s" % (img.shape,))
    col = np.random.randint(img.shape[1] - 128)
    row = np.random.randint(img.shape[0] - 128)
    centre = img[cy - 64: cy + 64, cx - 64: cx + 64]
    return centre, img[row:row + 128, col:col + 128], np.fliplr(centre)


def main(args):
    import PIL.Image
    private, public = select_images(read_identities(args.identity_annotations), args.num_images, args.num_same_id)
    for d in (args.output_dir0, args.output_dir1, args.output_dir2):
        if os.path.exists(d):
            shutil.rmtree(d)
        os.makedirs(d, exist_ok=True)
    for name in private:
        stem = name.split('.')[0]
        centre, rand, flip = crops(np.asarray(PIL.Image.open(os.path.join(args.input_dir, name))))
        PIL.Image.fromarray(centre).save(os.path.join(args.output_dir1, stem + '.png'))
        PIL.Image.fromarray(centre).save(os.path.join(args.output_dir0, stem + '.png'))
        PIL.Image.fromarray(rand).save(os.path.join(args.output_dir0, stem + '_a1.png'))
        PIL.Image.fromarray(flip).save(os.path.join(args.output_dir0, stem + '_a2.png'))
    for name in public:
        stem = name.split('.')[0]
        centre, _, _ = crops(np.asarray(PIL.Image.open(os.path.join(args.input_dir, name))))
        PIL.Image.fromarray(centre).save(os.path.join(args.output_dir2, stem + '.png'))
    return private, public


def update_args(args, config_dict):
    for key, val in config_dict.items():
        setattr(args, key, val)


if __name__ == '__main__':
    a = parse_arguments()
    print(a)
    if a.local_config is not None:
        import yaml
        with open(str(a.local_config), "r") as f:
            update_args(a, yaml.safe_load(f))
    main(a)
