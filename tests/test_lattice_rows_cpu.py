"""Host-side facts of the lattice search-row layout (no GPU needed: only the size / scale functions of the C ABI are called)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from ganleaks_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("hw", [(16, 16), (32, 32), (64, 64), (128, 128), (256, 256), (128, 256), (48, 80), (512, 512)])
def test_lattice_dim_and_scale(hw, lib):
    H, W = hw
    K = int(lib.gl_lpips_feature_dim(H, W))              # K_lpips + D, the algorithmic contraction length
    D = 3 * H * W
    K1 = int(lib.gl_lpips_lattice_dim(H, W))
    Dp = -(-D // 64) * 64
    body = K - D + Dp
    assert K1 in (body, body + 64) and K1 % 64 == 0
    assert (K1 == body + 64) == ((body * 2) % 32768 == 0)          # the pad keeps the row stride off multiples of 32 KiB
    hilo = int(lib.gl_lpips_search_dim(H, W))
    assert hilo >= K - D + 3 * Dp and hilo % 64 == 0                # the hi / lo layout carries the image part three times
    u = float(lib.gl_lpips_lattice_scale(H, W))
    assert 8192.0 < u <= 16384.0
    e = np.log2(u / (255.0 * np.sqrt(D)))
    assert abs(e - round(e)) < 1e-5                                 # u = 255 sqrt(D) 2^e with an integer e
    # every pixel value (2 c - 255) * 2^e is exact in fp16, and u * x / sqrt(D) reproduces it
    m = 2.0 * np.arange(256) - 255.0
    v = m * 2.0 ** round(e)
    assert np.array_equal(v.astype(np.float16).astype(np.float64), v)
    x = 2.0 * (np.arange(256) / 255.0) - 1.0                        # attack_models/utils.py:82
    assert np.allclose(u * x / np.sqrt(D), v, rtol=2e-7, atol=0)


def test_sizes_of_the_measured_configurations(lib):
    assert int(lib.gl_lpips_lattice_dim(64, 64)) == 512000 == int(lib.gl_lpips_feature_dim(64, 64))
    assert int(lib.gl_lpips_search_dim(64, 64)) == 536576
    assert int(lib.gl_lpips_lattice_dim(256, 256)) == 8192000 + 64
    assert abs(float(lib.gl_lpips_lattice_scale(64, 64)) - 14133.53) < 0.05 and abs(float(lib.gl_lpips_lattice_scale(256, 256)) - 14133.53) < 0.05
    assert int(lib.gl_lpips_lattice_dim(24, 64)) == -1
