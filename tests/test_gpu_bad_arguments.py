"""Error behaviour of the C ABI on a real device: every misuse returns a negative status with a message -- nothing throws across the
boundary, nothing launches on shapes the kernels do not support (a faulting kernel can take the whole GPU down)."""
import ctypes

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu
p = ctypes.c_void_p


@pytest.fixture(scope="module")
def env():
    import ganleaks_amd as gl
    from ganleaks_amd import _lib
    return gl, _lib.load(), gl.Context.get()


def _fails(lib, rc, needle=None):
    assert rc < 0, rc
    msg = lib.gl_last_error().decode()
    assert msg and (needle is None or needle in msg), msg


def test_l2_entry_points_reject_bad_sizes_and_pointers(env):
    gl, lib, ctx = env
    a = ctx.zeros((256, 128), np.int8)
    n = ctx.zeros((256,), np.int32)
    k = ctx.zeros((256,), np.uint64)
    h = ctx.handle
    _fails(lib, lib.gl_l2_knn_i8(h, p(a.ptr), p(n.ptr), 256, 0, p(a.ptr), p(n.ptr), 256, 0, p(k.ptr)), "bad sizes")          # d = 0
    _fails(lib, lib.gl_l2_knn_i8(h, p(a.ptr), p(n.ptr), -1, 0, p(a.ptr), p(n.ptr), 256, 128, p(k.ptr)), "bad sizes")
    _fails(lib, lib.gl_l2_knn_i8(h, p(a.ptr), p(n.ptr), 256, 0, p(a.ptr), p(n.ptr), 256, 1 << 20, p(k.ptr)), "bad sizes")    # d too large
    _fails(lib, lib.gl_l2_knn_i8(h, p(a.ptr + 4), p(n.ptr), 256, 0, p(a.ptr), p(n.ptr), 256, 128, p(k.ptr)), "aligned")
    _fails(lib, lib.gl_l2_knn_i8(h, p(0), p(n.ptr), 256, 0, p(a.ptr), p(n.ptr), 256, 128, p(k.ptr)), "NULL")
    _fails(lib, lib.gl_l2_knn_i8(h, p(a.ptr), p(n.ptr), 256, (1 << 32) - 8, p(a.ptr), p(n.ptr), 256, 128, p(k.ptr)), "global index")
    _fails(lib, lib.gl_l2_knn_i8(p(0), p(a.ptr), p(n.ptr), 256, 0, p(a.ptr), p(n.ptr), 256, 128, p(k.ptr)), "NULL ctx")
    assert lib.gl_l2_knn_i8(h, p(0), p(0), 0, 0, p(0), p(0), 0, 128, p(0)) == 0                                              # empty problem: nothing to do
    u = ctx.zeros((4, 128), np.uint8)
    _fails(lib, lib.gl_l2_prepare(h, p(u.ptr), 4, 300000, p(a.ptr), p(n.ptr)), "exceeds")
    _fails(lib, lib.gl_keys_unpack(h, p(k.ptr), 4, 0, p(n.ptr), p(k.ptr)))
    _fails(lib, lib.gl_l2_rows_u8(h, p(u.ptr), 4, p(u.ptr), 3, 128, p(n.ptr)), "x_gt")                                      # broadcast rule
    _fails(lib, lib.gl_fbb_knn_l2_host(h, p(0), 100, p(0), 4, 128, 64, p(0), p(0)), "NULL")


def test_generators_and_lpips_reject_unsupported_shapes(env):
    gl, lib, ctx = env
    g = p()
    _fails(lib, lib.gl_dcgan_create(ctx.handle, 100, 1, 64, ctypes.byref(g)), "channels_img")
    _fails(lib, lib.gl_dcgan_create(ctx.handle, 100, 3, 24, ctypes.byref(g)), "multiple of 16")
    assert lib.gl_dcgan_create(ctx.handle, 100, 3, 16, ctypes.byref(g)) == 0
    z = ctx.zeros((4, 100), np.float32)
    out = ctx.zeros((4, 3, 64, 64), np.uint8)
    _fails(lib, lib.gl_dcgan_forward(g, p(z.ptr), 4, p(0), p(out.ptr)), "not loaded")                                       # no weights yet
    _fails(lib, lib.gl_dcgan_set_precision(g, 7))
    _fails(lib, lib.gl_dcgan_set_spectral_norm(g, 9, p(z.ptr), p(z.ptr), p(z.ptr), p(z.ptr), p(z.ptr), 1))
    assert lib.gl_dcgan_destroy(g) == 0
    lp = p()
    assert lib.gl_lpips_create(ctx.handle, ctypes.byref(lp)) == 0
    v = ctx.zeros((1024,), np.float32)
    _fails(lib, lib.gl_lpips_features_u8(lp, p(out.ptr), 4, 64, 64, p(v.ptr), p(v.ptr)), "not loaded")
    _fails(lib, lib.gl_lpips_set_conv(lp, 13, p(v.ptr), p(v.ptr)))
    assert lib.gl_lpips_feature_dim(24, 64) == -1 and lib.gl_lpips_search_dim(64, 40) == -1
    _fails(lib, lib.gl_lpips_search_features_u8(lp, p(out.ptr), 4, 64, 64, 2, p(v.ptr), p(v.ptr)), "role")
    assert lib.gl_lpips_destroy(lp) == 0
    k = ctx.zeros((4,), np.uint64)
    _fails(lib, lib.gl_feat_knn_h1(ctx.handle, p(v.ptr), p(v.ptr), 4, 0, p(v.ptr), p(v.ptr), 4, 100, p(k.ptr)), "multiple of 64")
    _fails(lib, lib.gl_feat_knn(ctx.handle, p(v.ptr), p(v.ptr), 4, 0, p(v.ptr), p(v.ptr), 4, 33, p(k.ptr)), "multiple of 32")
    _fails(lib, lib.gl_rows_knn_split(ctx.handle, p(v.ptr), p(v.ptr), p(0), 4, 0, p(v.ptr), p(v.ptr), p(v.ptr), 4, 64, p(k.ptr)), "NULL")
    pg = p()
    _fails(lib, lib.gl_pggan_create(ctx.handle, 64, 48, 3, ctypes.byref(pg)))                                               # channels not a multiple of 32
