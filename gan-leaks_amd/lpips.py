"""LPIPS (v0.1, net-lin, VGG16) on the device: counterpart of the reference's
`lpips_pytorch.PerceptualLoss(model='net-lin', net='vgg')` (attack_models/lpips_pytorch/__init__.py:9-32)
as used by `Loss('l2-lpips')` (attack_models/utils.py:157,166-176).

Weights are read from LOCAL files only (the reference downloads the VGG16 backbone through torchvision,
pretrained_networks.py:99 -- there is no network here):
  * backbone: a torchvision `vgg16` state dict (keys `features.{0,2,...,28}.{weight,bias}`) or the bare
    `features` state dict (`{0,2,...}.{weight,bias}`)
  * lin layers: the reference's `attack_models/lpips_pytorch/pretrained_models/v0.1/vgg.pth`
    (keys `lin{0..4}.model.1.weight`)
Default locations come from $GANLEAKS_VGG16_PATH and $GANLEAKS_LPIPS_LIN_PATH.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from ._lib import Context, DeviceArray, check
from .attack import _to_device_rows, encode_if_lattice

_p = ctypes.c_void_p
VGG16_CONV_KEYS = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
LPIPS_CHANNELS = [64, 128, 256, 512, 512]


def _np(v):
    if type(v).__module__.startswith("torch"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class FeatureBank:
    """V rows + |V|^2 of a set of images, resident in HBM (gl_lpips_features_*)."""

    def __init__(self, ctx, V, norms, n, K, K_lp, index_base=0, role=None, fmt=None, scale=16384.0):
        self.ctx, self.V, self.norms, self.n, self.K, self.K_lp = ctx, V, norms, int(n), int(K), int(K_lp)
        self.index_base = int(index_base)
        self.kind = "feat"
        # None: split rows (hi + lo halves of every value; any use).  'query' / 'bank': search rows (one half per LPIPS value),
        # usable only as the two sides of feat_knn_keys.  Search rows come in two layouts (`fmt`):
        #   'lattice' -- 8-bit images: the image part is the exact fp16 integer (2 code - 255) * 2^e, one K segment, K =
        #                gl_lpips_lattice_dim halves, row scale gl_lpips_lattice_scale; queries and bank rows look the same
        #   'hilo'    -- any float image: the image part as hi / lo halves in three K segments ([hi|hi|lo] for queries, [hi|lo|hi]
        #                for bank rows), K = gl_lpips_search_dim halves, row scale 2^14
        # both sides of a search must have the same layout.
        self.role = role
        self.fmt = fmt if role is not None else None
        self.scale = float(scale)

    @property
    def blocked(self):
        """fp16 search rows of 2 MiB or more (images from 96 x 96) are stored K-blocked -- [row / 256][K / 64][row % 256][64 halves], see
        include/ganleaks.h gl_lpips_search_rows_capacity -- and their buffer holds whole blocks of 256 rows"""
        return self.role is not None and self.K * 2 >= (2 << 20)

    def rows_numpy(self):
        """the n rows as a host array [n, K] whatever the device layout (tests / tools)"""
        raw = self.V.numpy()
        if not self.blocked:
            return raw.reshape(-1, self.K)[:self.n]
        nb = raw.size // (256 * self.K)
        return raw.reshape(nb, self.K // 64, 256, 64).transpose(0, 2, 1, 3).reshape(nb * 256, self.K)[:self.n]

    def __len__(self):
        return self.n


class LpipsModel:
    def __init__(self, ctx=None):
        self.ctx = ctx or Context.get()
        h = _p()
        check(self.ctx.lib.gl_lpips_create(self.ctx.handle, ctypes.byref(h)))
        self._handle = h
        self._loaded = False
        self._warm = False            # shard.DeviceGroup: the calibration pass has run
        self._precision = 1
        # rows attack() builds for the nearest-neighbour search: 'fp16' = search rows (gl_feat_knn_h1: one fp16 MFMA per
        # product, 1.07 MB per 64x64 image), 'split' = hi + lo rows (gl_feat_knn: three MFMAs, 2.05 MB)
        self.search_rows = os.environ.get("GANLEAKS_LPIPS_SEARCH", "fp16")

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self.ctx.lib.gl_lpips_destroy(self._handle)
            except Exception:  # noqa: BLE001
                pass

    def load_state_dicts(self, vgg_state_dict, lin_state_dict):
        lib = self.ctx.lib
        for i, k in enumerate(VGG16_CONV_KEYS):
            for prefix in ("features.%d." % k, "%d." % k):
                if prefix + "weight" in vgg_state_dict:
                    break
            else:
                raise KeyError("VGG16 state dict has no conv %d (features.%d.weight)" % (k, k))
            w, b = _np(vgg_state_dict[prefix + "weight"]), _np(vgg_state_dict[prefix + "bias"])
            if w.ndim != 4 or w.shape[2:] != (3, 3) or b.shape != (w.shape[0],):
                raise ValueError("VGG16 conv %d has shape %s / %s" % (k, w.shape, b.shape))
            check(lib.gl_lpips_set_conv(self._handle, i, w.ctypes.data_as(_p), b.ctypes.data_as(_p)))
        for i in range(5):
            key = "lin%d.model.1.weight" % i
            w = _np(lin_state_dict[key] if key in lin_state_dict else lin_state_dict["lin%d" % i]).reshape(-1)
            if w.shape != (LPIPS_CHANNELS[i],):
                raise ValueError("lin%d has %d weights, expected %d" % (i, w.size, LPIPS_CHANNELS[i]))
            check(lib.gl_lpips_set_lin(self._handle, i, w.ctypes.data_as(_p)))
        self._loaded = True
        return self

    @classmethod
    def from_files(cls, vgg_path=None, lin_path=None, ctx=None):
        vgg_path = vgg_path or os.environ.get("GANLEAKS_VGG16_PATH")
        lin_path = lin_path or os.environ.get("GANLEAKS_LPIPS_LIN_PATH")
        if not vgg_path or not lin_path or not os.path.exists(vgg_path) or not os.path.exists(lin_path):
            raise FileNotFoundError(
                "'l2-lpips' needs local weight files: a torchvision VGG16 state dict (GANLEAKS_VGG16_PATH) and the LPIPS lin "
                "weights attack_models/lpips_pytorch/pretrained_models/v0.1/vgg.pth (GANLEAKS_LPIPS_LIN_PATH). "
                "The reference downloads the backbone (pretrained_networks.py:99); no download is attempted here.")
        import torch
        vgg = torch.load(vgg_path, map_location="cpu", weights_only=True)
        lin = torch.load(lin_path, map_location="cpu", weights_only=True)
        return cls(ctx).load_state_dicts(vgg, lin)

    def set_precision(self, mode):
        """1 (default) = split-fp16 VGG16 convolutions, 0 = fp32 MFMA"""
        check(self.ctx.lib.gl_lpips_set_precision(self._handle, int(mode)))
        self._precision = int(mode)

    def set_calibration(self, enabled):
        """split-fp16 mode: True (default) = per-layer power-of-two activation scales from a calibration pass over fixed synthetic images
        (include/ganleaks.h gl_lpips_set_calibration); False = the fixed factor 4 of rounds 1-2"""
        check(self.ctx.lib.gl_lpips_set_calibration(self._handle, 1 if enabled else 0))

    def set_chunk(self, images_per_pass):
        check(self.ctx.lib.gl_lpips_set_chunk(self._handle, int(images_per_pass)))

    def search_role(self, role):
        """role to pass to features() for attack()'s two operands under the current `search_rows` setting"""
        if self.search_rows not in ("fp16", "split"):
            raise ValueError("LpipsModel.search_rows must be 'fp16' or 'split', got %r" % (self.search_rows,))
        return role if self.search_rows == "fp16" else None

    def features(self, images, index_base=0, role=None, out=None, fmt=None):
        """images [n,3,H,W] (u8, or float in [-1,1]) -> FeatureBank.  role None: split rows; 'query' / 'bank': search rows.
        fmt (search rows only): None = 'lattice' when the images are 8-bit codes (u8, or floats on the 8-bit lattice), else 'hilo';
        'hilo' / 'lattice' force one (ValueError if 'lattice' is asked for off-lattice floats) -- the two sides of a search must agree.
        out: a FeatureBank of the same role, layout and image size with at least n rows whose buffers are overwritten (streamed banks reuse one)."""
        if role not in (None, "query", "bank"):
            raise ValueError("role must be None, 'query' or 'bank', got %r" % (role,))
        if fmt not in (None, "hilo", "lattice"):
            raise ValueError("fmt must be None, 'hilo' or 'lattice', got %r" % (fmt,))
        if not self._loaded:
            raise RuntimeError("LpipsModel: weights not loaded")
        ctx = self.ctx
        shape = tuple(images.shape)
        if len(shape) != 4 or shape[1] != 3:
            raise ValueError("LPIPS needs images of shape [n,3,H,W], got %s" % (shape,))
        n, _, H, W = shape
        K = int(ctx.lib.gl_lpips_feature_dim(H, W))
        if K < 0:
            raise ValueError("LPIPS path needs H and W to be multiples of 16, got %dx%d" % (H, W))
        rows = _to_device_rows(ctx, images)
        if rows.dtype == np.float32:
            u8, bad = encode_if_lattice(ctx, rows)
            if bad == 0:
                rows = u8
        u8 = rows.dtype == np.uint8
        if role is not None:
            if fmt == "lattice" and not u8:
                raise ValueError("lattice search rows need 8-bit images; these floats are off the lattice")
            fmt = fmt or ("lattice" if u8 else "hilo")
        else:
            fmt = None
        lattice = fmt == "lattice"
        K1 = K if role is None else int(ctx.lib.gl_lpips_lattice_dim(H, W) if lattice else ctx.lib.gl_lpips_search_dim(H, W))
        scale = float(ctx.lib.gl_lpips_lattice_scale(H, W)) if lattice else 16384.0
        cap = n if role is None else int(ctx.lib.gl_lpips_search_rows_capacity(max(n, 1), K1))      # long search rows: whole blocks of 256
        if out is not None:
            if getattr(out, "role", None) != role or getattr(out, "fmt", None) != fmt or out.K != K1 or out.V.shape[0] < cap:
                raise ValueError("features(out=...): buffer of another role / layout / image size, or too small")
            V, norms = out.V, out.norms
        elif role is None:
            V = ctx.empty((max(n, 1), K), np.float32)
            norms = ctx.empty((max(n, 1),), np.float32)
        else:
            V = ctx.empty((max(cap, 1), K1), np.float16)
            norms = ctx.empty((max(cap, 1),), np.float32)
        r = 0 if role == "query" else 1

        def run():
            if role is None:
                fn = ctx.lib.gl_lpips_features_u8 if u8 else ctx.lib.gl_lpips_features_f32
                check(fn(self._handle, _p(rows.ptr), n, H, W, _p(V.ptr), _p(norms.ptr)))
            elif lattice:
                check(ctx.lib.gl_lpips_lattice_features_u8(self._handle, _p(rows.ptr), n, H, W, _p(V.ptr), _p(norms.ptr)))
            else:
                fn = ctx.lib.gl_lpips_search_features_u8 if u8 else ctx.lib.gl_lpips_search_features_f32
                check(fn(self._handle, _p(rows.ptr), n, H, W, r, _p(V.ptr), _p(norms.ptr)))
        run()
        if self._precision == 1 and ctx.h3_saturations() > 0:
            import warnings
            warnings.warn("split-fp16 VGG16 path saturated for these weights; falling back to fp32 MFMA products")
            self.set_precision(0)
            run()
        if role is not None:
            return FeatureBank(ctx, V, norms, n, K1, K - 3 * H * W, index_base, role, fmt, scale)
        return FeatureBank(ctx, V, norms, n, K, K - 3 * H * W, index_base)


def features_sharded(model, images_u8, comm):
    """query search rows with the VGG16 work split over the ranks of `comm` (the queries are replicated, so their features are
    the one part of the path that does not shrink with the number of GPUs -- 64 ms of a 273 ms one-rank share of configs[2] at 8 GPUs):
    rank r featurises images [r * per, (r + 1) * per), per = ceil(Q / nranks), straight into its block of the full row buffer; two in-place
    all-gathers (rows, norms) on the context's stream give every rank all Q rows -- the same bits as featurising everything locally.
    8-bit images only (lattice rows: the layout cannot differ between ranks).  `comm`: a `_lib.Comm` (or anything with rank, nranks,
    allgather_rows(buf, bytes_per_rank))."""
    ctx = model.ctx
    shape = tuple(images_u8.shape)
    if len(shape) != 4 or shape[1] != 3 or getattr(images_u8, "dtype", None) != np.uint8:
        raise ValueError("features_sharded needs 8-bit images [n,3,H,W]")
    Q, _, H, W = shape
    K = int(ctx.lib.gl_lpips_feature_dim(H, W))
    if K < 0:
        raise ValueError("LPIPS path needs H and W to be multiples of 16, got %dx%d" % (H, W))
    K1 = int(ctx.lib.gl_lpips_lattice_dim(H, W))
    scale = float(ctx.lib.gl_lpips_lattice_scale(H, W))
    world, rank = int(comm.nranks), int(comm.rank)
    per = -(-Q // world)
    if K1 * 2 >= (2 << 20):
        per = -(-per // 256) * 256                    # K-blocked rows: a rank's block is whole blocks of 256 rows (contiguous bytes)
    V = ctx.empty((world * per, K1), np.float16)
    norms = ctx.empty((world * per,), np.float32)
    lo, hi = min(rank * per, Q), min((rank + 1) * per, Q)
    if hi > lo:
        mine = FeatureBank(ctx, V.view((per, K1), offset_bytes=lo * K1 * 2), norms.view((per,), offset_bytes=lo * 4), hi - lo, K1,
                           K - 3 * H * W, 0, "query", "lattice", scale)
        model.features(images_u8[lo:hi], role="query", fmt="lattice", out=mine)
    comm.allgather_rows(V, per * K1 * 2)
    comm.allgather_rows(norms, per * 4)
    return FeatureBank(ctx, V, norms, Q, K1, K - 3 * H * W, 0, "query", "lattice", scale)


_default = {"model": None, "factory": None}


def set_default_model(model):
    _default["model"] = model


def set_default_factory(factory):
    """factory(ctx) -> LpipsModel with its weights loaded, for callers that need one model per context (shard.DeviceGroup: a context per
    GPU); None restores the default, LpipsModel.from_files(ctx=ctx) on the local weight files."""
    _default["factory"] = factory


def model_for(ctx):
    """the LPIPS model of a given context: the registered factory's, else the default model when it lives on that context, else a new
    one from the local weight files"""
    if _default["factory"] is not None:
        return _default["factory"](ctx)
    if _default["model"] is not None and _default["model"].ctx is ctx:
        return _default["model"]
    return LpipsModel.from_files(ctx=ctx)


def default_model():
    if _default["model"] is None:
        _default["model"] = LpipsModel.from_files()
    return _default["model"]


def preferred_bank_rows(max_rows, n_queries):
    """rows per bank chunk of a streamed search (<= max_rows): gl_feat_knn_h1's persistent form hands super-tiles of 1024 bank rows x 2048
    queries to 8 clusters of workgroups, so a chunk whose super-tile count is just below a multiple of 8 wastes least (with 10 000 queries
    of 256 x 256 images a 64 GiB chunk holds 3 995 rows = 20 super-tiles = 3 rounds at 83 %; 3 072 rows = 15 super-tiles = 2 rounds at 94 %)."""
    max_rows, n_queries = int(max_rows), int(n_queries)
    if max_rows < 2048 or n_queries <= 0:
        return max(1, max_rows)
    sup_q = -(-n_queries // 2048)
    best, best_eff = max_rows, 0.0
    for k in range(1, max_rows // 1024 + 1):
        s = k * sup_q
        eff = s / (8.0 * -(-s // 8))
        if eff >= best_eff - 1e-9:            # ties go to the larger chunk
            best, best_eff = k * 1024, eff
    full = -(-max_rows // 1024) * sup_q       # the largest chunk as it is (ragged last super-tile row)
    if full / (8.0 * -(-full // 8)) * (max_rows / (-(-max_rows // 1024) * 1024.0)) >= best_eff:
        return max_rows
    return best


def feat_knn_keys(bank, queries, n_rows=None, keys=None):
    ctx = bank.ctx
    roles = (getattr(bank, "role", None), getattr(queries, "role", None))
    fmts = (getattr(bank, "fmt", None), getattr(queries, "fmt", None))
    if fmts[0] != fmts[1]:
        raise ValueError("the two sides of a search have different row layouts: %r vs %r (LpipsModel.features(..., fmt=...))" % fmts)
    if queries.K != bank.K:
        raise ValueError("feature lengths differ: %d vs %d" % (queries.K, bank.K))
    if fmts[0] != "lattice" and roles not in ((None, None), ("bank", "query")):
        raise ValueError("feat_knn_keys needs two split FeatureBanks or search rows of roles ('bank', 'query'); got %r" % (roles,))
    if fmts[0] == "lattice" and None in roles:
        raise ValueError("feat_knn_keys: lattice rows on one side, split rows on the other")
    n_rows = bank.n if n_rows is None else int(n_rows)
    if keys is None:
        keys = ctx.empty((max(queries.n, 1),), np.uint64)
        check(ctx.lib.gl_keys_init(ctx.handle, _p(keys.ptr), queries.n))
    if roles[0]:
        check(ctx.lib.gl_feat_knn_h1_scaled(ctx.handle, _p(bank.V.ptr), _p(bank.norms.ptr), n_rows, bank.index_base, _p(queries.V.ptr),
                                            _p(queries.norms.ptr), queries.n, bank.K, _p(keys.ptr), bank.scale))
    else:
        check(ctx.lib.gl_feat_knn(ctx.handle, _p(bank.V.ptr), _p(bank.norms.ptr), n_rows, bank.index_base, _p(queries.V.ptr),
                                  _p(queries.norms.ptr), queries.n, bank.K, _p(keys.ptr)))
    return keys


def rows_dist(a, g):
    """(lpips [b], l2 [b]) between FeatureBanks a (x_hat) and g (x_gt, 1 row or b rows)."""
    ctx = a.ctx
    if getattr(a, "role", None) or getattr(g, "role", None):
        raise ValueError("rows_dist needs split FeatureBanks (features(..., role=None))")
    lp = ctx.empty((max(a.n, 1),), np.float32)
    l2 = ctx.empty((max(a.n, 1),), np.float32)
    check(ctx.lib.gl_feat_rows_dist(ctx.handle, _p(a.V.ptr), a.n, _p(g.V.ptr), g.n, a.K, a.K_lp, _p(lp.ptr), _p(l2.ptr)))
    return lp.numpy()[:a.n], l2.numpy()[:a.n]
