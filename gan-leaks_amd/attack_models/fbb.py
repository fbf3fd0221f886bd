"""Counterpart of the reference's attack_models/fbb.py (the full-black-box attack driver).

Same command line, YAML overlay, output files and `custom_knn` signature; underneath, the whole
query set is searched in one launch of the HIP pairwise kernel instead of Q x (N / BATCH_SIZE)
Python iterations (attack_models/fbb.py:156-159, 73-88).

Additions (flagged [build] in SURVEY.md 5): --distance {l2,l2-lpips} (the reference hard-wires
'l2-lpips', fbb.py:148); --ngpu N / --devices 0,1,... shard the PNG bank over GPUs (the reference is
single-device, fbb.py:40): rank r uploads and searches rows [bounds[r], bounds[r+1]) of the sorted file list,
one all-reduce(min) of the packed keys gives the single-device result bit for bit (ganleaks_amd.shard).
The reference has no `attack()`; the batched entry point named by the project brief lives in
ganleaks_amd.attack.attack and is re-exported here.
"""
from __future__ import annotations

import argparse
import os
import pickle
import warnings

import numpy as np

from ..attack import Bank, _budget_bytes, attack, prepare_queries  # noqa: F401  (re-export)
from .utils import (Loss, check_folder, get_filepaths_from_dir, read_images_u8_nchw, save_files)


# (flag names, argparse keywords): the reference's command line (attack_models/fbb.py:18-38) plus --distance
_FLAGS = (
    (('--exp_name', '-name'), dict(type=str, default='debug', help='experiment name; results go to ./fbb_attack/<exp_name>')),
    (('--syn_data_path',), dict(type=str, help='folder with the generated sample bank (image_*.png), or with one sub-folder per bank for a sweep')),
    (('--pos_data_dir',), dict(type=str, default=None, help='folder with the member (training) query images')),
    (('--neg_data_dir',), dict(type=str, default=None, help='folder with the non-member (held-out) query images')),
    (('--data_num', '-dnum'), dict(type=int, default=20000, help='number of query images (kept for compatibility)')),
    (('--resolution', '-resolution'), dict(type=int, default=64, help='images that differ are resized to this square size')),
    (('--K',), dict(type=int, default=5)),
    (('--BATCH_SIZE',), dict(type=int, default=30)),
    (('--local_config',), dict(type=str, default=None)),
    (('--hyperparameter_search',), dict(default=False, help='treat every sub-folder of syn_data_path as one bank')),
    (('--params',), dict(type=str, default=None, help='name of the hyper-parameter setting (set per sub-folder in a sweep)')),
    (('--wandb',), dict(default=None, help='accepted for compatibility; nothing is logged to WandB')),
    (('--distance',), dict(type=str, default='l2-lpips', choices=['l2', 'l2-lpips'],
                           help="[build] distance operator; the reference always uses 'l2-lpips' (fbb.py:148)")),
    (('--ngpu',), dict(type=int, default=1, help='[build] shard the bank over the first N GPUs (one context and host thread per GPU, RCCL min)')),
    (('--devices',), dict(type=str, default=None, help='[build] explicit device ordinals for the shards, e.g. 0,1,2,3 (overrides --ngpu)')),
)


def parse_arguments(argv=None):
    """attack_models/fbb.py:18-38: identical flags and defaults, plus --distance."""
    parser = argparse.ArgumentParser()
    data_root = os.path.join(os.getcwd(), 'data', 'miniCelebA')
    for names, kw in _FLAGS:
        kw = dict(kw)
        if names[0] == '--pos_data_dir':
            kw['default'] = os.path.join(data_root, 'train')
        elif names[0] == '--neg_data_dir':
            kw['default'] = os.path.join(data_root, 'test')
        parser.add_argument(*names, **kw)
    return parser.parse_args(argv)


def check_args(args):
    """attack_models/fbb.py:42-67: the bank folder must exist; results go to ./fbb_attack/<exp_name> -- in a sweep to
    ./fbb_attack/<exp_name>__<sweep folder>/<params> -- and the arguments are recorded there as params.txt ("key:value" lines, also
    echoed) and params.pkl (pickle protocol 2 of the same dict)."""
    assert os.path.exists(args.syn_data_path)
    root = os.path.join(os.getcwd(), 'fbb_attack')
    if args.hyperparameter_search and args.params is not None:
        sweep_folder = args.syn_data_path.split('/')[-2]
        save_dir = os.path.join(root, '%s__%s' % (args.exp_name, sweep_folder), args.params)
    else:
        save_dir = os.path.join(root, args.exp_name)
    check_folder(save_dir)
    record = vars(args)
    lines = ["%s:%s" % (key, value) for key, value in record.items()]
    with open(os.path.join(save_dir, 'params.txt'), 'w') as handle:
        handle.write("".join(line + "\n" for line in lines))
    print("\n".join(lines))
    with open(os.path.join(save_dir, 'params.pkl'), 'wb') as handle:
        pickle.dump(record, handle, protocol=2)
    return args, save_dir


_bank_cache = {"key": None, "bank": None}


def _cached_bank(syn_imgs, n_rows, loss, fmt=None):
    """custom_knn is called once per query with the same bank (fbb.py:156-159): prepare it once
    (int8 rows for 'l2'; VGG16/LPIPS feature vectors for 'l2-lpips')."""
    if isinstance(syn_imgs, Bank) or getattr(syn_imgs, "kind", None) == "feat":
        return syn_imgs
    ptr = syn_imgs.data_ptr() if hasattr(syn_imgs, "data_ptr") else (
        syn_imgs.ctypes.data if isinstance(syn_imgs, np.ndarray) else id(syn_imgs))
    key = (id(syn_imgs), ptr, tuple(syn_imgs.shape), n_rows, loss.distance, id(loss.lpips_model), fmt)
    if _bank_cache["key"] != key:
        if loss.distance == "l2-lpips":
            _bank_cache["bank"] = loss.lpips_model.features(syn_imgs[:n_rows], role=loss.lpips_model.search_role("bank"), fmt=fmt)
        else:
            _bank_cache["bank"] = Bank.from_images(syn_imgs[:n_rows], keep_u8=True)
        _bank_cache["key"] = key
    return _bank_cache["bank"]


def _custom_knn_foreign_loss(syn_imgs, sample, loss, args):
    """a caller-supplied distance function: the reference's loop as written (fbb.py:77-88).  The callable does the arithmetic
    (wherever it likes).  This is not a fallback of the device path: Loss instances -- everything the reference itself passes --
    never reach it."""
    dists = []
    for i in range(len(syn_imgs) // args.BATCH_SIZE):
        x_batch = syn_imgs[i * args.BATCH_SIZE:(i + 1) * args.BATCH_SIZE]
        x_gt = sample.unsqueeze(0) if hasattr(sample, "unsqueeze") else np.asarray(sample)[None]
        d = loss(x_batch, x_gt)
        if hasattr(d, "detach"):
            d = d.detach().cpu().numpy()
        dists.append(np.asarray(d).reshape(-1))
    if not dists:
        raise ValueError("torch.cat(): expected a non-empty list of Tensors")   # what fbb.py:83 raises
    dists = np.concatenate(dists)
    k = int(np.argmin(dists))                                                  # first minimum, as torch.min (fbb.py:86)
    return float(dists[k]), k


def custom_knn(syn_imgs, sample, loss, args):
    """attack_models/fbb.py:73-88.  syn_imgs [N,C,H,W], sample [C,H,W], loss = Loss(...) instance,
    args.BATCH_SIZE.  Returns (min distance as python float, index as python int); only the first
    (N // BATCH_SIZE) * BATCH_SIZE bank samples take part, first index wins ties."""
    distance = getattr(loss, "distance", None)
    if distance is None:
        if not callable(loss):
            raise TypeError("loss must be a ganleaks_amd Loss instance or a callable(x_batch, x_gt) -> [B]")
        return _custom_knn_foreign_loss(syn_imgs, sample, loss, args)
    n_rows = (len(syn_imgs) // args.BATCH_SIZE) * args.BATCH_SIZE
    if n_rows == 0:
        raise ValueError("torch.cat(): expected a non-empty list of Tensors")   # what fbb.py:83 raises
    q = sample.unsqueeze(0) if hasattr(sample, "unsqueeze") else np.asarray(sample)[None]
    bank = _cached_bank(syn_imgs, n_rows, loss, _bank_cache.get("fmt") if _bank_cache.get("fmt_for") == id(syn_imgs) else None)
    try:
        dist, idx = attack(q, bank, distance=distance, batch_size=args.BATCH_SIZE, lpips=loss.lpips_model)
    except ValueError:
        # an off-lattice float query against the lattice search rows of an 8-bit bank: rebuild the cached rows in the layout that takes
        # any float image (LpipsModel.features), and keep using it for this bank
        if getattr(bank, "fmt", None) != "lattice":
            raise
        _bank_cache["fmt"], _bank_cache["fmt_for"] = "hilo", id(syn_imgs)
        bank = _cached_bank(syn_imgs, n_rows, loss, "hilo")
        dist, idx = attack(q, bank, distance=distance, batch_size=args.BATCH_SIZE, lpips=loss.lpips_model)
    return float(dist[0]), int(idx[0])


def plot_closest_images(idx, query_imgs_u8, syn_imgs_u8, save_dir, class_type, num=20):
    """attack_models/fbb.py:91-106: query | nearest sample, side by side, <i><class_type>.png.
    Works on the 8-bit codes directly (the reference converts float -> uint8 with truncation)."""
    import PIL.Image
    for i in range(min(num, len(idx))):
        syn_img = syn_imgs_u8[int(idx[i][0])].transpose(1, 2, 0)
        query_img = query_imgs_u8[i].transpose(1, 2, 0)
        img = np.concatenate((query_img, syn_img), axis=1)
        f = (2.0 * (img / 255.0) - 1.0 + 1.0) / 2.0
        PIL.Image.fromarray(np.uint8(f * 255)).save(os.path.join(save_dir, str(i) + class_type + '.png'))


def shard_devices(args):
    """device ordinals the bank is sharded over, or None for the single-device path (--devices wins over --ngpu)"""
    devices = getattr(args, "devices", None)
    if devices:
        if isinstance(devices, str):
            devices = [int(d) for d in devices.replace(",", " ").split()]
        return [int(d) for d in devices]
    ngpu = int(getattr(args, "ngpu", 1) or 1)
    if ngpu < 1:
        raise ValueError("--ngpu must be at least 1")
    return list(range(ngpu)) if ngpu > 1 else None


def main(args):
    """attack_models/fbb.py:111-179."""
    devices = shard_devices(args)
    group = None
    if devices is not None:
        from ..shard import DeviceGroup
        group = DeviceGroup(devices)                 # one context per GPU for the whole sweep: models and query rows are kept
    try:
        return _main(args, group)
    finally:
        if group is not None:
            group.close()


def _main(args, group):
    if args.hyperparameter_search:
        subdirs = [os.path.join(args.syn_data_path, o) for o in os.listdir(args.syn_data_path)
                   if os.path.isdir(os.path.join(args.syn_data_path, o))]
    else:
        subdirs = [args.syn_data_path]
    distance = getattr(args, "distance", "l2-lpips")
    results = []
    queries = None
    for subdir in subdirs:
        args.syn_data_path = subdir
        args.params = subdir.split('/')[-1] if args.hyperparameter_search else args.params
        args, save_dir = check_args(args)
        print(args)
        print('exp_name: ', args.exp_name)
        print('params: ', args.params)
        resolution = args.resolution

        syn_imgs = read_images_u8_nchw(get_filepaths_from_dir(subdir, ext='png'), resolution)
        if group is not None:
            # [build] the bank sharded over GPUs: every context prepares the (replicated) queries itself, so nothing is prepared here
            if queries is None or queries[0] != (args.pos_data_dir, args.neg_data_dir, resolution, distance):
                pos_query_imgs = read_images_u8_nchw(get_filepaths_from_dir(args.pos_data_dir, ext='png'), resolution)
                neg_query_imgs = read_images_u8_nchw(get_filepaths_from_dir(args.neg_data_dir, ext='png'), resolution)
                queries = ((args.pos_data_dir, args.neg_data_dir, resolution, distance), pos_query_imgs, neg_query_imgs, None,
                           np.concatenate([pos_query_imgs, neg_query_imgs]))
        elif queries is None or queries[0] != (args.pos_data_dir, args.neg_data_dir, resolution, distance):
            # the query sets do not change across the banks of a hyper-parameter sweep (fbb.py:114-123 re-reads them for every
            # sub-directory): read and prepare them once -- int8 rows, or VGG16/LPIPS search rows -- and reuse them for every bank
            pos_query_imgs = read_images_u8_nchw(get_filepaths_from_dir(args.pos_data_dir, ext='png'), resolution)
            neg_query_imgs = read_images_u8_nchw(get_filepaths_from_dir(args.neg_data_dir, ext='png'), resolution)
            custom_loss = Loss(distance, if_norm_reg=False)      # fbb.py:148 (loads the LPIPS model for 'l2-lpips')
            both = np.concatenate([pos_query_imgs, neg_query_imgs])
            # prepared only when the rows fit the streaming budget; otherwise attack() slices the raw queries itself
            prepared = prepare_queries(both, distance, lpips=custom_loss.lpips_model)
            queries = ((args.pos_data_dir, args.neg_data_dir, resolution, distance), pos_query_imgs, neg_query_imgs, custom_loss, prepared)
        _, pos_query_imgs, neg_query_imgs, custom_loss, prepared = queries
        n_rows = (len(syn_imgs) // args.BATCH_SIZE) * args.BATCH_SIZE
        if n_rows == 0:
            raise ValueError("torch.cat(): expected a non-empty list of Tensors")
        # both query sets go through ONE pass over the bank (the reference loops over it per query, fbb.py:156-168): the bank's
        # prepared rows -- int8 rows or VGG16/LPIPS feature rows -- are built once, and streamed through HBM in chunks when
        # they would not fit (ganleaks_amd.attack, $GANLEAKS_CHUNK_GB)
        n_pos = len(pos_query_imgs)
        if group is not None:
            all_d, all_i = group.attack(prepared, bank=syn_imgs, distance=distance, batch_size=args.BATCH_SIZE)
        else:
            all_d, all_i = attack(prepared, syn_imgs, distance=distance, batch_size=args.BATCH_SIZE, lpips=custom_loss.lpips_model)
        pos_d, pos_i, neg_d, neg_i = all_d[:n_pos], all_i[:n_pos], all_d[n_pos:], all_i[n_pos:]
        pos_loss = pos_d.astype(np.float64).reshape(-1, 1)          # python floats -> float64 [Q,1] (fbb.py:160)
        plt_pos_idx = pos_i.reshape(-1, 1)
        save_files(save_dir, ['pos_loss', 'pos_idx'], [pos_loss, np.arange(len(pos_loss)).reshape(-1, 1)])

        neg_loss = neg_d.astype(np.float64).reshape(-1, 1)
        plt_neg_idx = neg_i.reshape(-1, 1)
        # the reference writes arange(len(pos_loss)) here as well (fbb.py:171): kept
        save_files(save_dir, ['neg_loss', 'neg_idx'], [neg_loss, np.arange(len(pos_loss)).reshape(-1, 1)])
        # [build] the real nearest-neighbour indices, which the reference keeps only in memory
        save_files(save_dir, ['pos_nn_idx', 'neg_nn_idx'], [plt_pos_idx, plt_neg_idx])

        plot_closest_images(plt_pos_idx, pos_query_imgs, syn_imgs, save_dir, 'pos')
        plot_closest_images(plt_neg_idx, neg_query_imgs, syn_imgs, save_dir, 'neg')
        results.append((save_dir, pos_loss, neg_loss, plt_pos_idx, plt_neg_idx))
    return results


def update_args(args, config_dict):
    """attack_models/fbb.py:182-184: keys of the YAML config override the command line"""
    for name in config_dict:
        setattr(args, name, config_dict[name])


if __name__ == '__main__':
    import yaml
    cli = parse_arguments()
    if cli.local_config is None:
        warnings.warn("No config file was provided. Using default parameters.")
    else:
        with open(str(cli.local_config)) as handle:
            update_args(cli, yaml.safe_load(handle))
    main(cli)
