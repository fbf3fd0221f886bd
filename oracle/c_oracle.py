"""ctypes wrapper of the C oracle (oracle/fbb_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfbb_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        p = ctypes.c_void_p
        i64 = ctypes.c_int64
        _lib.gl_oracle_knn_l2_u8.argtypes = [p, i64, p, i64, i64, p, p]
        _lib.gl_oracle_ssd_row_u8.argtypes = [p, i64, p, i64, p]
        _lib.gl_oracle_row_norms_u8.argtypes = [p, i64, i64, p]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knn_l2_u8(bank_u8, queries_u8, batch_size):
    """same contract as oracle.knn_l2_u8 (exact integer L2, truncation, first-index ties)."""
    bank = np.ascontiguousarray(bank_u8, np.uint8)
    qs = np.ascontiguousarray(queries_u8, np.uint8)
    n_eff = (len(bank) // batch_size) * batch_size
    if n_eff == 0:
        raise ValueError("bank smaller than BATCH_SIZE (fbb.py:77-83)")
    d = int(np.prod(bank.shape[1:]))
    assert int(np.prod(qs.shape[1:])) == d
    idx = np.empty(len(qs), np.int64)
    ssd = np.empty(len(qs), np.int64)
    rc = lib().gl_oracle_knn_l2_u8(_ptr(bank), n_eff, _ptr(qs), len(qs), d, _ptr(idx), _ptr(ssd))
    assert rc == 0
    dist = (ssd.astype(np.float64) * (4.0 / (65025.0 * d))).astype(np.float32)
    return dist, idx, ssd


def ssd_row_u8(bank_u8, query_u8):
    bank = np.ascontiguousarray(bank_u8, np.uint8)
    q = np.ascontiguousarray(query_u8, np.uint8)
    d = int(np.prod(bank.shape[1:]))
    out = np.empty(len(bank), np.int64)
    lib().gl_oracle_ssd_row_u8(_ptr(bank), len(bank), _ptr(q), d, _ptr(out))
    return out


def row_norms_u8(x_u8):
    x = np.ascontiguousarray(x_u8, np.uint8)
    d = int(np.prod(x.shape[1:]))
    out = np.empty(len(x), np.int32)
    lib().gl_oracle_row_norms_u8(_ptr(x), len(x), d, _ptr(out))
    return out


def knn_l2_f32(bank_f32, queries_f32, batch_size):
    """general fp32 images: the fixed-order fp32 chain shared with the device path (fbb_oracle.c)."""
    bank = np.ascontiguousarray(bank_f32, np.float32)
    qs = np.ascontiguousarray(queries_f32, np.float32)
    n_eff = (len(bank) // batch_size) * batch_size
    if n_eff == 0:
        raise ValueError("bank smaller than BATCH_SIZE (fbb.py:77-83)")
    d = int(np.prod(bank.shape[1:]))
    idx = np.empty(len(qs), np.int64)
    dist = np.empty(len(qs), np.float32)
    fn = lib().gl_oracle_knn_l2_f32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(_ptr(bank), n_eff, _ptr(qs), len(qs), d, _ptr(idx), _ptr(dist)) == 0
    return dist, idx


def l2_pair_f32(y, x):
    y = np.ascontiguousarray(y, np.float32).reshape(-1)
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    fn = lib().gl_oracle_l2_pair_f32
    fn.restype = ctypes.c_float
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    return np.float32(fn(_ptr(y), _ptr(x), y.size))
