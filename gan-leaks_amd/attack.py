"""Batched nearest-neighbour attack over a device-resident sample bank.

`attack()` is the batched form of the reference's per-query `custom_knn`
(attack_models/fbb.py:73-88): it must equal
    [custom_knn(bank, q, Loss(distance), args) for q in queries].
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import Context, DeviceArray, as_device, check

_p = ctypes.c_void_p


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def prepare_images(ctx, images):
    """images -> DeviceArray u8 [count, D].

    Accepts u8 arrays (numpy / torch / DeviceArray) of shape [count, ...] or float32 images in
    [-1,1] that sit exactly on the 8-bit lattice 2*(u/255.)-1 -- which is everything
    attack_models/utils.py:60-84 (read_image) can produce.  Off-lattice floats are refused: the
    exact-integer path would silently change their values.
    """
    if isinstance(images, DeviceArray):
        arr = images
    else:
        if _is_torch(images):
            kind = "f" if images.dtype.is_floating_point else "u"
            is_u8 = str(images.dtype) == "torch.uint8"
        else:
            images = np.asarray(images)
            kind = images.dtype.kind
            is_u8 = images.dtype == np.uint8
        if is_u8:
            arr = as_device(ctx, images, np.uint8)
        elif kind == "f":
            arr = as_device(ctx, images.float() if _is_torch(images) else images.astype(np.float32, copy=False), np.float32)
        else:
            raise TypeError("images must be uint8 or float, got %r" % (images.dtype,))
    count = arr.shape[0] if len(arr.shape) else 0
    d = int(np.prod(arr.shape[1:], dtype=np.int64)) if len(arr.shape) > 1 else 1
    if arr.dtype == np.uint8:
        return arr.view((count, d))
    if arr.dtype != np.float32:
        raise TypeError("device images must be uint8 or float32")
    out = ctx.empty((count, d), np.uint8)
    flag = ctx.zeros((1,), np.int32)
    check(ctx.lib.gl_encode_lattice_f32(ctx.handle, _p(arr.ptr), count * d, _p(out.ptr), _p(flag.ptr)))
    bad = int(flag.numpy()[0])
    if bad:
        raise NotImplementedError(
            "%d of %d pixel values are not on the 8-bit lattice 2*(u/255.)-1 that read_image produces "
            "(attack_models/utils.py:82); the exact-integer L2 path only accepts 8-bit images" % (bad, count * d))
    return out


class Bank:
    """a sample bank prepared for the L2 kernel: biased int8 rows + row norms, resident in HBM.

    `index_base` is the global index of row 0 (non-zero for a shard of a larger bank)."""

    def __init__(self, ctx, rows_i8, norms, n, d, index_base=0, u8=None):
        self.ctx, self.rows_i8, self.norms, self.n, self.d = ctx, rows_i8, norms, int(n), int(d)
        self.index_base = int(index_base)
        self.u8 = u8

    @classmethod
    def from_images(cls, images, ctx=None, index_base=0, keep_u8=False):
        ctx = ctx or Context.get()
        u8 = prepare_images(ctx, images)
        n, d = u8.shape
        stride = int(ctx.lib.gl_l2_row_stride(d))
        rows = ctx.empty((n, stride), np.int8)
        norms = ctx.empty((max(n, 1),), np.int32)
        check(ctx.lib.gl_l2_prepare(ctx.handle, _p(u8.ptr), n, d, _p(rows.ptr), _p(norms.ptr)))
        ctx.sync()
        return cls(ctx, rows, norms, n, d, index_base, u8 if keep_u8 else None)

    def __len__(self):
        return self.n


def knn_keys(bank, queries, n_rows=None, keys=None):
    """launch the pairwise kernel: returns the packed keys DeviceArray [Q] (uint64):
    (S << 32) | global index, min over bank rows [0, n_rows).  Asynchronous."""
    ctx = bank.ctx
    if not isinstance(queries, Bank):
        queries = Bank.from_images(queries, ctx)
    if queries.d != bank.d:
        raise ValueError("query images have %d values, bank images %d" % (queries.d, bank.d))
    n_rows = bank.n if n_rows is None else int(n_rows)
    if keys is None:
        keys = ctx.empty((max(queries.n, 1),), np.uint64)
        check(ctx.lib.gl_keys_init(ctx.handle, _p(keys.ptr), queries.n))
    check(ctx.lib.gl_l2_knn_i8(ctx.handle, _p(bank.rows_i8.ptr), _p(bank.norms.ptr), n_rows, bank.index_base,
                               _p(queries.rows_i8.ptr), _p(queries.norms.ptr), queries.n, bank.d, _p(keys.ptr)))
    return keys, queries


def unpack_keys(ctx, keys, nq, d):
    dist = ctx.empty((max(nq, 1),), np.float32)
    idx = ctx.empty((max(nq, 1),), np.int64)
    check(ctx.lib.gl_keys_unpack(ctx.handle, _p(keys.ptr), nq, d, _p(dist.ptr), _p(idx.ptr)))
    return dist.numpy()[:nq], idx.numpy()[:nq]


def attack(queries, bank, distance="l2", batch_size=64, ctx=None, reduce_fn=None):
    """nearest bank sample of every query.

    queries : [Q,C,H,W] images (u8, or float on the 8-bit lattice), numpy / torch / DeviceArray / Bank
    bank    : same, or a prepared `Bank` (then `batch_size` truncation applies to len(bank) unless the
              bank is a shard, index_base > 0 or reduce_fn given: shards are truncated by the caller,
              see shard.py)
    distance: 'l2' (attack_models/utils.py:161-164).  'l2-lpips' is the reference's fbb default
              (attack_models/fbb.py:148) and lands with the LPIPS kernels.
    returns (dist float32 [Q], idx int64 [Q]); idx < (N // batch_size) * batch_size (fbb.py:77),
    smallest index on ties (fbb.py:86).
    reduce_fn: optional callable(keys DeviceArray) -> keys DeviceArray, the cross-GPU min (shard.py).
    """
    if distance != "l2":
        raise NotImplementedError("distance %r: only 'l2' is implemented in this round" % (distance,))
    if isinstance(bank, Bank):
        ctx = bank.ctx
        n_rows = bank.n
        if reduce_fn is None and bank.index_base == 0:
            n_rows = (bank.n // int(batch_size)) * int(batch_size)
    else:
        ctx = ctx or Context.get()
        n_total = len(bank)
        n_rows = (n_total // int(batch_size)) * int(batch_size)
        if n_rows > 0:
            if isinstance(bank, DeviceArray):
                bank = bank.view((n_rows,) + tuple(bank.shape[1:]))
            else:
                bank = bank[:n_rows]
            bank = Bank.from_images(bank, ctx)
    if n_rows == 0 and reduce_fn is None:
        # the reference dies in torch.cat([]) (fbb.py:83) with ValueError
        raise ValueError("bank holds no full batch of %d samples (attack_models/fbb.py:77-83)" % int(batch_size))
    keys, q = knn_keys(bank, queries, n_rows)
    if reduce_fn is not None:
        keys = reduce_fn(keys)
    return unpack_keys(ctx, keys, q.n, bank.d)
