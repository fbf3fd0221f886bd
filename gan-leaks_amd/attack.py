"""Batched nearest-neighbour attack over a device-resident sample bank.

`attack()` is the batched form of the reference's per-query `custom_knn`
(attack_models/fbb.py:73-88): it must equal
    [custom_knn(bank, q, Loss(distance), args) for q in queries].

Arithmetic paths:
  * images on the 8-bit lattice 2*(u/255.)-1 -- everything attack_models/utils.py:60-84 (read_image)
    can produce -- are searched in exact integer arithmetic on the int8 matrix cores (Bank kind 'u8');
  * rows of small non-negative integers (binary / count tables, e.g. medGAN's thresholded samples) likewise (kind 'int');
  * any other fp32 rows take the fixed-order fp32 path (csrc/gl_l2f32.hip, kind 'f32').
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import Context, DeviceArray, as_device, check

_p = ctypes.c_void_p


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _to_device_rows(ctx, images):
    """-> DeviceArray [count, D] of dtype uint8 or float32 (no value change)."""
    if isinstance(images, DeviceArray):
        arr = images
    else:
        if _is_torch(images):
            is_u8 = str(images.dtype) == "torch.uint8"
            is_f = images.dtype.is_floating_point
        else:
            images = np.asarray(images)
            is_u8 = images.dtype == np.uint8
            is_f = images.dtype.kind == "f"
        if is_u8:
            arr = as_device(ctx, images, np.uint8)
        elif is_f:
            arr = as_device(ctx, images.float() if _is_torch(images) else images.astype(np.float32, copy=False), np.float32)
        else:
            raise TypeError("images must be uint8 or float, got %r" % (images.dtype,))
    if arr.dtype not in (np.dtype(np.uint8), np.dtype(np.float32)):
        raise TypeError("device images must be uint8 or float32")
    count = arr.shape[0] if len(arr.shape) else 0
    d = int(np.prod(arr.shape[1:], dtype=np.int64)) if len(arr.shape) > 1 else 1
    return arr.view((count, d))


def encode_if_lattice(ctx, rows_f32, integers=False):
    """float32 rows -> (u8 rows, off_lattice_count).  The u8 rows are only meaningful when the count is 0.
    integers=False: the image lattice 2*(u/255.)-1; True: x == float(u), u in 0..255."""
    count, d = rows_f32.shape
    out = ctx.empty((count, d), np.uint8)
    flag = ctx.zeros((1,), np.int32)
    fn = ctx.lib.gl_encode_integers_f32 if integers else ctx.lib.gl_encode_lattice_f32
    check(fn(ctx.handle, _p(rows_f32.ptr), count * d, _p(out.ptr), _p(flag.ptr)))
    return out, int(flag.numpy()[0])


def prepare_images(ctx, images):
    """images -> u8 DeviceArray [count, D]; raises if float values are off the 8-bit lattice."""
    rows = _to_device_rows(ctx, images)
    if rows.dtype == np.uint8:
        return rows
    out, bad = encode_if_lattice(ctx, rows)
    if bad:
        raise ValueError("%d pixel values are not on the 8-bit lattice 2*(u/255.)-1" % bad)
    return out


class Bank:
    """a sample bank (or query set) resident in HBM, prepared for one of the two L2 kernels.

    kind 'u8' : biased int8 rows + int32 row norms (exact path), bytes are image codes: x = 2*(u/255.)-1
    kind 'int': the same, bytes are the values themselves: x = float(u)  (float inputs only; uint8 input always means image codes)
    kind 'f32': the fp32 rows as given (fixed-order fp32 path)
    `index_base` is the global index of row 0 (non-zero for a shard of a larger bank)."""

    def __init__(self, ctx, kind, n, d, index_base=0, rows_i8=None, norms=None, rows_f32=None, u8=None):
        self.ctx, self.kind, self.n, self.d = ctx, kind, int(n), int(d)
        self.index_base = int(index_base)
        self.rows_i8, self.norms, self.rows_f32, self.u8 = rows_i8, norms, rows_f32, u8

    @classmethod
    def from_images(cls, images, ctx=None, index_base=0, keep_u8=False, force_kind=None):
        ctx = ctx or Context.get()
        rows = _to_device_rows(ctx, images)
        n, d = rows.shape
        kind = "u8"
        if rows.dtype == np.float32:
            u8, bad = None, 1
            for cand in (("u8", "int") if force_kind is None else (force_kind,)):
                if cand == "f32":
                    break
                u8, bad = encode_if_lattice(ctx, rows, integers=(cand == "int"))
                if bad == 0:
                    kind = cand
                    break
            if bad:
                if force_kind in ("u8", "int"):
                    raise ValueError("%d values are not on the %s" % (bad, "8-bit lattice" if force_kind == "u8" else "integer lattice 0..255"))
                return cls(ctx, "f32", n, d, index_base, rows_f32=rows)
            rows = u8
        elif force_kind == "f32":
            f = ctx.empty((n, d), np.float32)
            check(ctx.lib.gl_decode_u8(ctx.handle, _p(rows.ptr), n * d, _p(f.ptr)))
            return cls(ctx, "f32", n, d, index_base, rows_f32=f)
        elif force_kind == "int":
            raise ValueError("uint8 input is read as image codes; pass float rows for the integer lattice")
        stride = int(ctx.lib.gl_l2_row_stride(d))
        rows_i8 = ctx.empty((n, stride), np.int8)
        norms = ctx.empty((max(n, 1),), np.int32)
        check(ctx.lib.gl_l2_prepare(ctx.handle, _p(rows.ptr), n, d, _p(rows_i8.ptr), _p(norms.ptr)))
        ctx.sync()
        return cls(ctx, kind, n, d, index_base, rows_i8=rows_i8, norms=norms, u8=rows if keep_u8 else None)

    def split_rows(self):
        """(V, norms, scales) of an 'f32' bank for the matrix-core search (gl_rows_knn_split); built once, kept"""
        if self.kind != "f32":
            raise ValueError("split_rows() is for fp32 banks")
        if getattr(self, "_split", None) is None:
            ctx = self.ctx
            kp = int(ctx.lib.gl_rows_split_dim(self.d))
            V = ctx.empty((max(self.n, 1), kp), np.float32)
            norms = ctx.empty((max(self.n, 1),), np.float32)
            scales = ctx.empty((max(self.n, 1),), np.float32)
            check(ctx.lib.gl_rows_split_f32(ctx.handle, _p(self.rows_f32.ptr), self.n, self.d, _p(V.ptr), _p(norms.ptr), _p(scales.ptr)))
            self._split = (V, norms, scales)
        return self._split

    def as_f32(self):
        """an fp32 view of a u8 bank (needed when the other side of the comparison is off-lattice)."""
        if self.kind == "f32":
            return self
        if self.u8 is None:
            raise ValueError("bank was prepared without keep_u8; cannot convert to fp32")
        if self.kind == "int":
            f = self.ctx.empty((self.n, self.d), np.float32)
            check(self.ctx.lib.gl_decode_u8_integers(self.ctx.handle, _p(self.u8.ptr), self.n * self.d, _p(f.ptr)))
            return Bank(self.ctx, "f32", self.n, self.d, self.index_base, rows_f32=f)
        return Bank.from_images(self.u8, self.ctx, self.index_base, force_kind="f32")

    def __len__(self):
        return self.n


class GeneratedBank:
    """a bank that is never materialised: rows [lo, hi) are generated on demand, `generator.generate_u8(z[lo:hi], **kwargs)`,
    in the order the reference's generate branch would have written them as image_{i}.png (bank index = z index).
    `index_base` is the global index of z[0] when z is one shard of the latents (shard.py)."""
    kind = "generated"

    def __init__(self, generator, z, index_base=0, **generate_kwargs):
        if not hasattr(generator, "generate_u8"):
            raise TypeError("generator must provide generate_u8(z, ...) -> u8 DeviceArray")
        self.generator, self.z, self.index_base, self.kwargs = generator, z, int(index_base), generate_kwargs
        self.ctx = generator.ctx

    def __len__(self):
        return len(self.z)

    def rows(self, lo, hi):
        return self.generator.generate_u8(self.z[lo:hi], **self.kwargs)


def _budget_bytes():
    import os
    return int(float(os.environ.get("GANLEAKS_CHUNK_GB", "64")) * (1 << 30))


def _query_budget_bytes(chunk_bytes, ctx=None):
    """HBM the prepared QUERY rows of a streamed l2-lpips attack may occupy next to one bank chunk: $GANLEAKS_QUERY_GB if set, else what the
    device has available right now (gl_mem_info) minus the bank chunk and 12 GiB for the workspaces of the generator and of VGG16 (two
    activation buffers of up to 3 GiB each), the search scratch and the allocator's slack -- 191 GiB on an idle 288 GB MI355X with the
    default 64 GiB chunk.  Query rows that fit stay resident for the whole bank stream -- 10 000 search rows of 256 x 256 images are
    153 GiB -- so the bank is generated and featurised once; rows that do not fit go in slices, each against the whole (regenerated) bank
    stream."""
    import os
    if "GANLEAKS_QUERY_GB" in os.environ:
        return int(float(os.environ["GANLEAKS_QUERY_GB"]) * (1 << 30))
    if chunk_bytes != 64 * (1 << 30):
        return chunk_bytes                 # a caller who set its own chunk budget gets the same budget for the query rows
    if ctx is None:
        return 192 * (1 << 30)
    available, _ = ctx.mem_info()
    return max(available - int(chunk_bytes) - 12 * (1 << 30), min(int(chunk_bytes), available // 4))


def float_path(value=None):
    """how off-lattice fp32 rows are searched: 'exact' (default; VALU, one fixed fp32 order shared bit for bit with the oracle) or
    'mfma' (split-fp16 on the matrix cores, |y|^2 + |x|^2 - 2 y.x; distances agree to ~3e-6 * mean(x^2), 15-60x faster).  $GANLEAKS_FLOAT_PATH."""
    import os
    value = value or os.environ.get("GANLEAKS_FLOAT_PATH", "exact")
    if value not in ("exact", "mfma"):
        raise ValueError("float path must be 'exact' or 'mfma', got %r" % (value,))
    return value


def knn_keys(bank, queries, n_rows=None, keys=None, fpath=None):
    """launch the pairwise kernel: packed keys DeviceArray [Q] (uint64), min over bank rows [0, n_rows).
    u8 path: (S << 32) | global index; f32 path: (float_bits(dist) << 32) | global index.  Asynchronous."""
    ctx = bank.ctx
    if not isinstance(queries, Bank):
        queries = Bank.from_images(queries, ctx, keep_u8=True, force_kind="f32" if bank.kind == "f32" else None)
    if queries.d != bank.d:
        raise ValueError("query images have %d values, bank images %d" % (queries.d, bank.d))
    if queries.kind != bank.kind:
        # one side is off-lattice: compare in fp32 (the lattice side decodes exactly to what read_image yields)
        bank, queries = bank.as_f32(), queries.as_f32()
    n_rows = bank.n if n_rows is None else int(n_rows)
    if keys is None:
        keys = ctx.empty((max(queries.n, 1),), np.uint64)
        check(ctx.lib.gl_keys_init(ctx.handle, _p(keys.ptr), queries.n))
    if bank.kind in ("u8", "int"):
        check(ctx.lib.gl_l2_knn_i8(ctx.handle, _p(bank.rows_i8.ptr), _p(bank.norms.ptr), n_rows, bank.index_base,
                                   _p(queries.rows_i8.ptr), _p(queries.norms.ptr), queries.n, bank.d, _p(keys.ptr)))
    elif float_path(fpath) == "mfma":
        (bv, bn, bs), (qv, qn, qs) = bank.split_rows(), queries.split_rows()
        check(ctx.lib.gl_rows_knn_split(ctx.handle, _p(bv.ptr), _p(bn.ptr), _p(bs.ptr), n_rows, bank.index_base, _p(qv.ptr), _p(qn.ptr), _p(qs.ptr),
                                        queries.n, bank.d, _p(keys.ptr)))
    else:
        check(ctx.lib.gl_l2_knn_f32(ctx.handle, _p(bank.rows_f32.ptr), n_rows, bank.index_base, _p(queries.rows_f32.ptr), queries.n, bank.d,
                                    _p(keys.ptr)))
    return keys, queries, bank.kind


def unpack_keys(ctx, keys, nq, d, kind="u8"):
    dist = ctx.empty((max(nq, 1),), np.float32)
    idx = ctx.empty((max(nq, 1),), np.int64)
    if kind == "u8":
        check(ctx.lib.gl_keys_unpack(ctx.handle, _p(keys.ptr), nq, d, _p(dist.ptr), _p(idx.ptr)))
    elif kind == "int":
        check(ctx.lib.gl_keys_unpack_integers(ctx.handle, _p(keys.ptr), nq, d, _p(dist.ptr), _p(idx.ptr)))
    else:
        check(ctx.lib.gl_keys_unpack_f32(ctx.handle, _p(keys.ptr), nq, _p(dist.ptr), _p(idx.ptr)))
    return dist.numpy()[:nq], idx.numpy()[:nq]


def _feature_row_bytes(ctx, model, images):
    h, w = int(images.shape[2]), int(images.shape[3])
    if model.search_rows == "fp16":
        if getattr(images, "dtype", None) == np.uint8:               # 8-bit codes: lattice search rows (LpipsModel.features)
            return 2 * int(ctx.lib.gl_lpips_lattice_dim(h, w))
        return 2 * int(ctx.lib.gl_lpips_search_dim(h, w))
    return 4 * int(ctx.lib.gl_lpips_feature_dim(h, w))


def _attack_streamed(queries, bank, n_rows, distance, ctx, reduce_fn, model, chunk_bytes, fpath=None, index_base=0):
    """bank rows [0, n_rows) pass through HBM in chunks of at most `chunk_bytes` of prepared rows (int8 rows for 'l2', feature
    rows for 'l2-lpips'); the packed keys accumulate the minimum across chunks (atomicMin), so the result is the one the
    resident form gives.  `bank` is a GeneratedBank or a host array / DeviceArray of images (`index_base`: global index of its row 0)."""
    generated = getattr(bank, "kind", None) == "generated"
    base = bank.index_base if generated else int(index_base)

    def rows(lo, hi):
        if generated:
            return bank.rows(lo, hi)
        return bank.view((hi - lo,) + tuple(bank.shape[1:]), offset_bytes=lo * (bank.nbytes // max(len(bank), 1))) if isinstance(bank, DeviceArray) else bank[lo:hi]

    def finish(keys, nq, d, kind):
        if keys is None:                 # a shard without rows still takes part in the reduction
            keys = ctx.empty((max(nq, 1),), np.uint64)
            check(ctx.lib.gl_keys_init(ctx.handle, _p(keys.ptr), nq))
        if reduce_fn is not None:
            keys = reduce_fn(keys)
        return unpack_keys(ctx, keys, nq, d, kind)

    if distance == "l2-lpips":
        from . import lpips as _lp
        if getattr(queries, "kind", None) != "feat" and len(queries):
            # query rows that would not fit the budget either (256 x 256 images: 17 MB per search row) go in slices, each against
            # the whole bank stream -- the bank's features are then recomputed once per slice
            per_q = _feature_row_bytes(ctx, model, queries)
            q_step = max(1, int(_query_budget_bytes(chunk_bytes, ctx) // per_q))
            if len(queries) > q_step:
                parts = [_attack_streamed(queries[a:a + q_step], bank, n_rows, distance, ctx, reduce_fn, model, chunk_bytes, fpath, index_base)
                         for a in range(0, len(queries), q_step)]
                return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        raw_queries = getattr(queries, "kind", None) != "feat"
        fq = model.features(queries, role=model.search_role("query")) if raw_queries else queries
        # both sides must have one row layout: 8-bit codes give lattice rows, an off-lattice float chunk forces the hi / lo layout on
        # everything (the queries are then featurised again, which needs their images)
        for attempt in (0, 1):
            b_role = "bank" if getattr(fq, "role", None) else None
            step = max(1, int(chunk_bytes // (fq.K * (2 if fq.role else 4))))
            if fq.role:
                step = _lp.preferred_bank_rows(step, fq.n)
            keys, buf, ok = None, None, True
            for lo in range(0, n_rows, step):
                hi = min(lo + step, n_rows)
                try:
                    buf = model.features(rows(lo, hi), index_base=base + lo, role=b_role, out=buf, fmt=getattr(fq, "fmt", None))   # one buffer for every chunk
                except ValueError:
                    if attempt or not raw_queries or getattr(fq, "fmt", None) != "lattice":
                        raise
                    ok = False
                    break
                keys = _lp.feat_knn_keys(buf, fq, keys=keys)
                ctx.sync()
            if ok:
                return finish(keys, fq.n, fq.K, "f32")
            del buf
            fq = model.features(queries, role="query", fmt="hilo")
        raise AssertionError("unreachable")

    # 'l2': every chunk must take the same arithmetic path.  Exact integers unless the queries or some chunk are off the 8-bit lattice;
    # then everything is redone on the fixed-order fp32 path (what the resident form does for such inputs).
    fq = queries if isinstance(queries, Bank) else Bank.from_images(queries, ctx, keep_u8=True)
    for force in ((fq.kind, "f32") if fq.kind != "f32" else ("f32",)):
        q_side = fq if fq.kind == force else fq.as_f32()
        step = max(1, int(chunk_bytes // ((4 if force == "f32" else 2) * fq.d)))
        keys, ok = None, True
        for lo in range(0, n_rows, step):
            hi = min(lo + step, n_rows)
            try:
                b = Bank.from_images(rows(lo, hi), ctx, index_base=base + lo, force_kind=force)
            except ValueError:           # an off-lattice chunk
                ok = False
                break
            keys, _, _ = knn_keys(b, q_side, keys=keys, fpath=fpath)
            ctx.sync()
        if ok:
            return finish(keys, fq.n, fq.d, force)
    raise AssertionError("unreachable")


def prepare_queries(queries, distance, ctx=None, lpips=None, comm=None):
    """the query side of attack() prepared once, for callers that search several banks (a sweep, the chunks of a sharded bank) or want
    the fallible part (uploads, VGG16 features) done before a collective: int8 rows for 'l2', LPIPS search rows for 'l2-lpips' when
    they fit the streaming budget -- otherwise the images are returned as they are and attack() slices them itself.
    comm (a `_lib.Comm` of more than one rank; COLLECTIVE: every rank must call this with the same queries): the VGG16 features of 8-bit
    queries are computed Q / nranks per rank and all-gathered (lpips.features_sharded) instead of all of them on every rank."""
    if isinstance(queries, Bank) or getattr(queries, "kind", None) == "feat" or not len(queries):
        return queries
    if distance == "l2-lpips":
        from . import lpips as _lp
        model = lpips or _lp.default_model()
        if len(queries) * _feature_row_bytes(model.ctx, model, queries) > _budget_bytes():
            return queries
        if (comm is not None and comm.nranks > 1 and model.search_rows == "fp16" and isinstance(queries, np.ndarray) and queries.dtype == np.uint8
                and len(queries) >= 8 * comm.nranks):
            return _lp.features_sharded(model, queries, comm)
        return model.features(queries, role=model.search_role("query"))
    return Bank.from_images(queries, ctx or Context.get(), keep_u8=True)


def attack(queries, bank, distance="l2", batch_size=64, ctx=None, reduce_fn=None, lpips=None, chunk_bytes=None, float_path=None, index_base=0):
    """nearest bank sample of every query.

    queries : [Q,C,H,W] images, u8 or float; numpy / torch / DeviceArray / Bank / FeatureBank
    bank    : same, or a prepared `Bank` / `FeatureBank` (then `batch_size` truncation applies to len(bank)
              unless the bank is a shard -- index_base > 0 or reduce_fn given: shards are cut after the
              global truncation, see shard.py), or a `GeneratedBank(generator, z)` whose rows are generated,
              searched and dropped chunk by chunk.  Unprepared banks whose prepared rows would exceed `chunk_bytes`
              (default $GANLEAKS_CHUNK_GB = 64 GiB) are streamed through HBM the same way.
    distance: 'l2' (attack_models/utils.py:161-164) or 'l2-lpips' = 0.2*LPIPS + L2, the reference's fbb
              distance (attack_models/fbb.py:148, utils.py:166-176).  `lpips` is the LpipsModel to use
              (default: lpips.default_model(), weights from local files).
    returns (dist float32 [Q], idx int64 [Q]); idx < (N // batch_size) * batch_size (fbb.py:77),
    smallest index on ties (fbb.py:86).
    reduce_fn: optional callable(keys DeviceArray) -> keys DeviceArray, the cross-GPU min (shard.py).
    index_base: for an unprepared image array that is one shard of a larger bank: the global index of its row 0 (prepared and generated
              banks carry their own).  Like them, a shard (index_base > 0 or reduce_fn given) is not truncated again.
    float_path: 'exact' | 'mfma' for rows that are on neither lattice (see attack.float_path; default $GANLEAKS_FLOAT_PATH or 'exact').
    """
    if distance not in ("l2", "l2-lpips"):
        raise ValueError("distance must be 'l2' or 'l2-lpips', got %r" % (distance,))
    prepared = isinstance(bank, Bank) or getattr(bank, "kind", None) == "feat"
    generated = getattr(bank, "kind", None) == "generated"
    if prepared or generated:
        ctx = bank.ctx
        n_rows = len(bank)
        if reduce_fn is None and bank.index_base == 0:
            n_rows = (n_rows // int(batch_size)) * int(batch_size)
    else:
        ctx = ctx or Context.get()
        n_total = len(bank)
        index_base = int(index_base)
        n_rows = n_total if (reduce_fn is not None or index_base) else (n_total // int(batch_size)) * int(batch_size)
    if n_rows == 0 and reduce_fn is None:
        # the reference dies in torch.cat([]) (fbb.py:83) with ValueError
        raise ValueError("bank holds no full batch of %d samples (attack_models/fbb.py:77-83)" % int(batch_size))
    model = None
    if distance == "l2-lpips":
        from . import lpips as _lp
        model = lpips or _lp.default_model()
    if not prepared:
        chunk_bytes = _budget_bytes() if chunk_bytes is None else int(chunk_bytes)
        if generated or n_rows == 0:         # (an empty shard still takes part in the reduction: the streamed form handles it)
            need = chunk_bytes + 1
        else:
            per_img = int(np.prod(tuple(bank.shape[1:]), dtype=np.int64)) if len(bank) else 0
            if distance == "l2-lpips" and len(bank):
                per_img = _feature_row_bytes(ctx, model, bank)
                if getattr(queries, "kind", None) != "feat" and len(queries) * per_img > chunk_bytes:
                    need = chunk_bytes + 1           # the query rows alone exceed a chunk: streamed form (queries resident or in slices)
                else:
                    need = per_img * n_rows
            else:
                need = 2 * per_img * n_rows          # u8 codes + int8 rows
        if need > chunk_bytes:
            return _attack_streamed(queries, bank, n_rows, distance, ctx, reduce_fn, model, chunk_bytes, float_path, index_base)
    if not prepared and n_rows > 0:
        if isinstance(bank, DeviceArray):
            bank = bank.view((n_rows,) + tuple(bank.shape[1:]))
        else:
            bank = bank[:n_rows]

    if distance == "l2-lpips":
        q_feat = getattr(queries, "kind", None) == "feat"
        fb = bank if prepared else model.features(bank, index_base=index_base, role=model.search_role("bank"),
                                                  fmt=getattr(queries, "fmt", None) if q_feat else None)
        q_role = "query" if getattr(fb, "role", None) else None          # queries follow the bank's row format
        if q_feat:
            fq = queries
        else:
            try:
                fq = model.features(queries, role=q_role, fmt=getattr(fb, "fmt", None))
            except ValueError:
                # off-lattice float queries against lattice rows of an 8-bit bank: both sides in the hi / lo layout instead
                if prepared or getattr(fb, "fmt", None) != "lattice":
                    raise
                fb = model.features(bank, index_base=index_base, role="bank", fmt="hilo")
                fq = model.features(queries, role="query", fmt="hilo")
        keys = _lp.feat_knn_keys(fb, fq, n_rows)
        if reduce_fn is not None:
            keys = reduce_fn(keys)
        return unpack_keys(ctx, keys, fq.n, fb.K, "f32")

    if not prepared:
        bank = Bank.from_images(bank, ctx, index_base=index_base, keep_u8=True)
    keys, q, kind = knn_keys(bank, queries, n_rows, fpath=float_path)
    if reduce_fn is not None:
        keys = reduce_fn(keys)
    return unpack_keys(ctx, keys, q.n, bank.d, kind)
