"""ctypes binding of libganleaks_hip.so (C ABI: include/ganleaks.h).

The product path has no CPU fallback: if the shared library is missing or no MI355X is visible,
everything that computes raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
import threading

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# $GANLEAKS_LIB: another build of the same ABI -- in practice libganleaks_hip_tuning.so (`make -C gan-leaks_amd/csrc tuning`), the only build
# that reads the GL_* tuning variables and holds the experiment kernels; the A/B tools under tools/ set it, nothing else should
LIB_PATH = os.environ.get("GANLEAKS_LIB") or os.path.join(_PKG_DIR, "libganleaks_hip.so")
TUNING_LIB_PATH = os.path.join(_PKG_DIR, "libganleaks_hip_tuning.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "ganleaks.h")

GL_OK = 0
GL_ERR_EMPTY_BANK = -5
GL_ERR_RCCL = -6
GL_COMM_ID_BYTES = 128


class GanLeaksError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libganleaks_hip: [%d] %s" % (code, msg))
        self.code = code


_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i = ctypes.c_int
_sz = ctypes.c_size_t
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); kept in step with include/ganleaks.h (tests/test_abi.py checks both ways)
SIGNATURES = {
    "gl_abi_version": (_i, []),
    "gl_last_error": (ctypes.c_char_p, []),
    "gl_device_count": (_i, [ctypes.POINTER(_i)]),
    "gl_ctx_create": (_i, [_i, _pp]),
    "gl_ctx_destroy": (_i, [_p]),
    "gl_ctx_set_stream": (_i, [_p, _p]),
    "gl_ctx_get_stream": (_i, [_p, _pp]),
    "gl_ctx_sync": (_i, [_p]),
    "gl_ctx_h3_saturations": (_i, [_p, ctypes.POINTER(_i64)]),
    "gl_malloc": (_i, [_p, _sz, _pp]),
    "gl_free": (_i, [_p, _p]),
    "gl_ctx_trim": (_i, [_p]),
    "gl_mem_info": (_i, [_p, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "gl_memcpy_h2d": (_i, [_p, _p, _p, _sz]),
    "gl_memcpy_d2h": (_i, [_p, _p, _p, _sz]),
    "gl_memset": (_i, [_p, _p, _i, _sz]),
    "gl_event_create": (_i, [_pp]),
    "gl_event_destroy": (_i, [_p]),
    "gl_event_record": (_i, [_p, _p]),
    "gl_event_elapsed_ms": (_i, [_p, _p, ctypes.POINTER(ctypes.c_float)]),
    "gl_prof_enable": (_i, [_p, _i]),
    "gl_prof_read": (_i, [_p, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64)]),
    "gl_prof_reset": (_i, [_p]),
    "gl_encode_lattice_f32": (_i, [_p, _p, _i64, _p, _p]),
    "gl_decode_u8": (_i, [_p, _p, _i64, _p]),
    "gl_quantize_f32": (_i, [_p, _p, _i64, _i, _p]),
    "gl_l2_row_stride": (_i64, [_i64]),
    "gl_l2_prepare": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "gl_keys_init": (_i, [_p, _p, _i64]),
    "gl_l2_knn_i8": (_i, [_p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _p]),
    "gl_keys_unpack": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "gl_encode_integers_f32": (_i, [_p, _p, _i64, _p, _p]),
    "gl_decode_u8_integers": (_i, [_p, _p, _i64, _p]),
    "gl_keys_unpack_integers": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "gl_l2_rows_u8": (_i, [_p, _p, _i64, _p, _i64, _i64, _p]),
    "gl_l2_knn_f32": (_i, [_p, _p, _i64, _i64, _p, _i64, _i64, _p]),
    "gl_keys_unpack_f32": (_i, [_p, _p, _i64, _p, _p]),
    "gl_l2_rows_f32": (_i, [_p, _p, _i64, _p, _i64, _i64, _p]),
    "gl_fbb_knn_l2_host": (_i, [_p, _p, _i64, _p, _i64, _i64, _i64, _p, _p]),
    "gl_dcgan_create": (_i, [_p, _i, _i, _i, _pp]),
    "gl_dcgan_destroy": (_i, [_p]),
    "gl_dcgan_set_conv_weight": (_i, [_p, _i, _p]),
    "gl_dcgan_set_bn": (_i, [_p, _i, _p, _p, _p, _p, ctypes.c_float]),
    "gl_dcgan_set_out_bias": (_i, [_p, _p]),
    "gl_dcgan_forward": (_i, [_p, _p, _i64, _p, _p]),
    "gl_dcgan_set_chunk": (_i, [_p, _i64]),
    "gl_dcgan_set_precision": (_i, [_p, _i]),
    "gl_dcgan_set_affine": (_i, [_p, _i, _p, _p]),
    "gl_dcgan_set_spectral_norm": (_i, [_p, _i, _p, _p, _p, _p, _p, _i]),
    "gl_dcgan_get_spectral_state": (_i, [_p, _i, _p, _p]),
    "gl_dcgan_set_spectral_hold": (_i, [_p, _i]),
    "gl_dcgan_set_fuse_tail": (_i, [_p, _i]),
    "gl_dcgan_set_attention": (_i, [_p, _p, _p, _p, _p, _p, _p, ctypes.c_float]),
    "gl_pggan_create": (_i, [_p, _i, _i, _i, _pp]),
    "gl_pggan_destroy": (_i, [_p]),
    "gl_pggan_set_initial": (_i, [_p, _p, _p, _p, _p]),
    "gl_pggan_set_block": (_i, [_p, _i, _p, _p, _p, _p]),
    "gl_pggan_set_rgb": (_i, [_p, _i, _p, _p]),
    "gl_pggan_set_chunk": (_i, [_p, _i64]),
    "gl_pggan_set_precision": (_i, [_p, _i]),
    "gl_pggan_forward": (_i, [_p, _p, _i64, _i, ctypes.c_float, _p, _p]),
    "gl_medgan_create": (_i, [_p, _i, _i, _i, _i, _pp]),
    "gl_medgan_destroy": (_i, [_p]),
    "gl_medgan_set_gen_block": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, ctypes.c_float]),
    "gl_medgan_set_decoder": (_i, [_p, _p, _p]),
    "gl_medgan_generate": (_i, [_p, _p, _i64, _p]),
    "gl_medgan_decode": (_i, [_p, _p, _i64, _p, _p]),
    "gl_lpips_create": (_i, [_p, _pp]),
    "gl_lpips_destroy": (_i, [_p]),
    "gl_lpips_set_conv": (_i, [_p, _i, _p, _p]),
    "gl_lpips_set_lin": (_i, [_p, _i, _p]),
    "gl_lpips_set_chunk": (_i, [_p, _i64]),
    "gl_lpips_set_precision": (_i, [_p, _i]),
    "gl_lpips_set_calibration": (_i, [_p, _i]),
    "gl_lpips_feature_dim": (_i64, [_i, _i]),
    "gl_lpips_features_u8": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "gl_lpips_features_f32": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "gl_feat_knn": (_i, [_p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _p]),
    "gl_fbb_knn_lpips_host": (_i, [_p, _p, _p, _i64, _p, _i64, _i, _i, _i64, _i64, _p, _p]),
    "gl_rows_split_dim": (_i64, [_i64]),
    "gl_rows_split_f32": (_i, [_p, _p, _i64, _i64, _p, _p, _p]),
    "gl_rows_knn_split": (_i, [_p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _p]),
    "gl_lpips_search_dim": (_i64, [_i, _i]),
    "gl_lpips_lattice_dim": (_i64, [_i, _i]),
    "gl_lpips_search_rows_capacity": (_i64, [_i64, _i64]),
    "gl_lpips_lattice_scale": (ctypes.c_float, [_i, _i]),
    "gl_lpips_lattice_features_u8": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "gl_lpips_search_features_u8": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p]),
    "gl_lpips_search_features_f32": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p]),
    "gl_feat_knn_h1": (_i, [_p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _p]),
    "gl_feat_knn_h1_scaled": (_i, [_p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _p, ctypes.c_float]),
    "gl_feat_rows_dist": (_i, [_p, _p, _i64, _p, _i64, _i64, _i64, _p, _p]),
    "gl_comm_unique_id": (_i, [_p]),
    "gl_comm_init_rank": (_i, [_p, _p, _i, _i, _pp]),
    "gl_comm_init_all": (_i, [_pp, _i, _pp]),
    "gl_comm_destroy": (_i, [_p]),
    "gl_comm_abort": (_i, [_p]),
    "gl_comm_rank": (_i, [_p, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "gl_allreduce_min_keys": (_i, [_p, _p, _i64]),
    "gl_allgather_rows": (_i, [_p, _p, _p, _i64]),
    "gl_comm_group_start": (_i, []),
    "gl_comm_group_end": (_i, []),
}

_lib = None
_lock = threading.Lock()


def build(verbose=False):
    """compile the HIP library in-tree (hipcc --offload-arch=gfx950; works without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG_DIR, "csrc"), "-j4"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libganleaks_hip.so failed:\n%s\n%s" % (res.stdout, res.stderr))
    return LIB_PATH


def load():
    """dlopen the library and declare every prototype.  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gan-leaks_amd/csrc`. "
                "There is no CPU fallback for the attack path." % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If this library pulls in the system
        # ROCm runtime first, a later `import torch` ends up with a runtime it cannot initialise ("No HIP GPUs are
        # available").  Loading torch first makes both share one runtime; without torch installed nothing changes.
        if "torch" not in sys.modules and not os.environ.get("GANLEAKS_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if lib.gl_abi_version() != 1:
            raise ImportError("libganleaks_hip.so ABI version %d != 1" % lib.gl_abi_version())
        _lib = lib
        return lib


def check(rc):
    if rc != GL_OK:
        msg = load().gl_last_error().decode("utf-8", "replace")
        if rc == GL_ERR_EMPTY_BANK:
            # the reference raises ValueError here: torch.cat([]) at attack_models/fbb.py:83
            raise ValueError(msg)
        raise GanLeaksError(rc, msg)


def device_count():
    n = _i(0)
    check(load().gl_device_count(ctypes.byref(n)))
    return n.value


# ------------------------------------------------------------------------------------------------
class Context:
    """one per GPU per process: a device + a private HIP stream (gl_ctx)."""

    _instances = {}

    def __init__(self, device=0):
        self.lib = load()
        h = _p()
        check(self.lib.gl_ctx_create(int(device), ctypes.byref(h)))
        self.handle = h
        self.device = int(device)

    @classmethod
    def get(cls, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
            n = device_count()
            if n > 0:
                device %= n
        if device not in cls._instances:
            cls._instances[device] = Context(device)
        return cls._instances[device]

    def sync(self):
        check(self.lib.gl_ctx_sync(self.handle))

    def mem_info(self):
        """(available, total) bytes of the context's GPU; available = free + what the arena keeps (gl_mem_info)"""
        a, t = _sz(0), _sz(0)
        check(self.lib.gl_mem_info(self.handle, ctypes.byref(a), ctypes.byref(t)))
        return int(a.value), int(t.value)

    def trim(self):
        """return the large device blocks the context keeps for reuse (gl_free's arena, include/ganleaks.h) to the driver"""
        check(self.lib.gl_ctx_trim(self.handle))

    def destroy(self):
        """release the stream and scratch of a context created with Context(device) (arrays and models built on it must be gone)"""
        if getattr(self, "handle", None) is not None:
            for k, v in list(Context._instances.items()):
                if v is self:
                    del Context._instances[k]
            self.lib.gl_ctx_destroy(self.handle)
            self.handle = None

    def h3_saturations(self):
        """workgroups of split-fp16 kernels that clamped a value to the fp16 range since the last call (synchronises)"""
        n = _i64(0)
        check(self.lib.gl_ctx_h3_saturations(self.handle, ctypes.byref(n)))
        return n.value

    @property
    def stream(self):
        s = _p()
        check(self.lib.gl_ctx_get_stream(self.handle, ctypes.byref(s)))
        return s.value or 0

    def set_stream(self, hip_stream):
        check(self.lib.gl_ctx_set_stream(self.handle, _p(hip_stream or 0)))

    # ---- memory
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DeviceArray(self, shape, dtype)
        check(self.lib.gl_memset(self.handle, _p(a.ptr), 0, a.nbytes))
        return a

    def to_device(self, host):
        host = np.ascontiguousarray(host)
        a = DeviceArray(self, host.shape, host.dtype)
        if a.nbytes:
            check(self.lib.gl_memcpy_h2d(self.handle, _p(a.ptr), host.ctypes.data_as(_p), a.nbytes))
        return a

    # ---- events / per-kernel timing
    def event(self):
        return Event(self)

    PROF_TAGS = {"gather_conv": 0, "l2_knn": 1, "convt_rgb": 2, "l2_prepare": 3, "feat_knn": 4}

    def prof_enable(self, on=True):
        check(self.lib.gl_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        check(self.lib.gl_prof_reset(self.handle))

    def prof_read(self):
        """{kernel: (total_ms, launches)} since the last reset (synchronises)."""
        out = {}
        for name, tag in self.PROF_TAGS.items():
            ms, n = ctypes.c_double(0), _i64(0)
            check(self.lib.gl_prof_read(self.handle, tag, ctypes.byref(ms), ctypes.byref(n)))
            out[name] = (ms.value, n.value)
        return out


class Comm:
    """one rank of an RCCL communicator bound to a Context (gl_comm): the cross-GPU minimum of the packed keys.

    Comm.unique_id() on rank 0 -> bytes carried to every rank by the launcher -> Comm(ctx, id, rank, nranks) on every rank (collective),
    or Comm.init_all([ctx0, ctx1, ...]) for one process driving several GPUs."""

    def __init__(self, ctx, unique_id=None, rank=0, nranks=1, _handle=None):
        self.ctx = ctx
        if _handle is not None:
            self.handle = _handle
        else:
            if unique_id is None:
                if nranks != 1:
                    raise ValueError("a communicator of %d ranks needs the unique id rank 0 generated" % nranks)
                unique_id = Comm.unique_id()
            if len(unique_id) != GL_COMM_ID_BYTES:
                raise ValueError("unique id must be %d bytes" % GL_COMM_ID_BYTES)
            buf = ctypes.create_string_buffer(bytes(unique_id), GL_COMM_ID_BYTES)
            h = _p()
            check(ctx.lib.gl_comm_init_rank(ctx.handle, buf, int(rank), int(nranks), ctypes.byref(h)))
            self.handle = h
        r, n = _i(0), _i(0)
        check(ctx.lib.gl_comm_rank(self.handle, ctypes.byref(r), ctypes.byref(n)))
        self.rank, self.nranks = r.value, n.value

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(GL_COMM_ID_BYTES)
        check(load().gl_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_all(cls, contexts):
        n = len(contexts)
        arr = (_p * n)(*[c.handle for c in contexts])
        out = (_p * n)()
        check(load().gl_comm_init_all(arr, n, out))
        return [cls(contexts[i], _handle=_p(out[i])) for i in range(n)]

    def allreduce_min_keys(self, keys):
        """in place on the context's stream (asynchronous): keys[q] = min over ranks"""
        if keys.dtype != np.dtype(np.uint64):
            raise TypeError("keys must be uint64")
        n = int(np.prod(keys.shape, dtype=np.int64))
        check(self.ctx.lib.gl_allreduce_min_keys(self.handle, _p(keys.ptr), n))
        return keys

    def allgather_rows(self, buf, bytes_per_rank):
        """in place on the context's stream (asynchronous): block r of `buf` (bytes_per_rank bytes each, this rank's block already filled)
        = rank r's block, for every rank"""
        total = int(np.prod(buf.shape, dtype=np.int64)) * buf.dtype.itemsize
        if bytes_per_rank * self.nranks > total:
            raise ValueError("buffer of %d bytes cannot hold %d blocks of %d" % (total, self.nranks, bytes_per_rank))
        check(self.ctx.lib.gl_allgather_rows(self.handle, _p(buf.ptr + self.rank * bytes_per_rank), _p(buf.ptr), int(bytes_per_rank)))
        return buf

    def destroy(self):
        if getattr(self, "handle", None) is not None:
            self.ctx.lib.gl_comm_destroy(self.handle)
            self.handle = None

    def abort(self):
        """ncclCommAbort: end the communicator without waiting for its queued collectives (error path; callable from any thread)"""
        h, self.handle = getattr(self, "handle", None), None
        if h is not None:
            self.ctx.lib.gl_comm_abort(h)

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001
            pass


class Event:
    def __init__(self, ctx):
        self.ctx = ctx
        h = _p()
        check(ctx.lib.gl_event_create(ctypes.byref(h)))
        self.handle = h

    def record(self):
        check(self.ctx.lib.gl_event_record(self.ctx.handle, self.handle))
        return self

    def elapsed_ms_until(self, stop):
        ms = ctypes.c_float(0)
        check(self.ctx.lib.gl_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            self.ctx.lib.gl_event_destroy(self.handle)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class DeviceArray:
    """a typed view of device memory owned by the library (gl_malloc / gl_free).

    Exposes __cuda_array_interface__, so `torch.as_tensor(arr, device='cuda')` aliases it without
    a copy when PyTorch-ROCm is used for plumbing (collectives)."""

    def __init__(self, ctx, shape, dtype, ptr=None, owner=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owner = owner
        if ptr is None:
            h = _p()
            check(ctx.lib.gl_malloc(ctx.handle, max(self.nbytes, 16), ctypes.byref(h)))
            self.ptr = h.value
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2, "strides": None}

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(self.ctx.lib.gl_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(_p), _p(self.ptr), self.nbytes))
        return out

    def view(self, shape, dtype=None, offset_bytes=0):
        return DeviceArray(self.ctx, shape, dtype or self.dtype, ptr=self.ptr + offset_bytes, owner=self)

    def __len__(self):
        return self.shape[0]

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            try:
                self.ctx.lib.gl_free(self.ctx.handle, _p(self.ptr))
            except Exception:  # noqa: BLE001
                pass
            self.ptr = 0


def as_device(ctx, x, dtype=None):
    """numpy array / torch tensor (CPU or ROCm) / DeviceArray -> DeviceArray (no copy when already on device)."""
    if isinstance(x, DeviceArray):
        if dtype is not None and np.dtype(dtype) != x.dtype:
            raise TypeError("device array has dtype %s, expected %s" % (x.dtype, np.dtype(dtype)))
        return x
    if type(x).__module__.startswith("torch"):
        t = x.detach()
        if t.is_cuda:
            t = t.contiguous()
            np_dtype = np.dtype(str(t.dtype).replace("torch.", ""))
            if dtype is not None and np.dtype(dtype) != np_dtype:
                raise TypeError("tensor has dtype %s, expected %s" % (np_dtype, np.dtype(dtype)))
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return DeviceArray(ctx, tuple(t.shape), np_dtype, ptr=t.data_ptr(), owner=t)
        x = t.cpu().numpy()
    x = np.asarray(x)
    if dtype is not None and x.dtype != np.dtype(dtype):
        x = x.astype(dtype)
    return ctx.to_device(x)
