"""Size-independent properties of the L2 search at BASELINE config-2 size (10 000 queries x 100 000 samples of 3x64x64), where
no CPU oracle finishes in seconds:
  * planted queries (exact copies of bank rows) come back with distance 0 and the SMALLEST index holding that image;
  * every reported (dist, idx) is self-consistent: dist equals the exact distance to bank[idx] recomputed on the host, and no row
    of a random host sample of the bank is closer (nor equally close with a smaller index);
  * the truncation bound idx < (N // B) * B holds;
  * min-merging the keys of two bank shards reproduces the unsharded result bit for bit (checksum over all 10 000 results);
  * the one-call host ABI (PCIe included) agrees with the device-pointer path.
The small-size tests pin the same code to the C oracle and to the reference's golden vectors."""
import ctypes
import time

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    return ganleaks_amd


def test_config2_size_properties(gl):
    from ganleaks_amd import _lib
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    N, Q, D, B = 100_000, 10_000, 3 * 64 * 64, 64
    n_eff = (N // B) * B
    rng = np.random.default_rng(2024)
    # low-entropy structure so nearest neighbours are not all at the same distance: 4096 prototypes + noise
    proto = rng.integers(0, 256, size=(4096, D), dtype=np.uint8)
    bank = proto[rng.integers(0, 4096, size=N)]
    noise_rows = rng.integers(0, N, size=N // 2)
    bank[noise_rows, :64] = rng.integers(0, 256, size=(len(noise_rows), 64), dtype=np.uint8)
    q = bank[rng.integers(0, N, size=Q)].copy()
    planted = rng.integers(0, n_eff, size=1000)
    q[:1000] = bank[planted]
    q[1000:, 64:192] = rng.integers(0, 256, size=(Q - 1000, 128), dtype=np.uint8)

    ctx = gl.Context.get()
    t0 = time.perf_counter()
    dist, idx = gl.attack(q, bank, distance="l2", batch_size=B)
    t_attack = time.perf_counter() - t0
    assert idx.max() < n_eff and idx.min() >= 0
    # planted copies: distance 0, smallest index with identical bytes
    assert np.all(dist[:1000] == 0)
    for k in range(0, 1000, 50):
        same = np.flatnonzero((bank[:n_eff] == q[k]).all(axis=1))
        assert idx[k] == same[0]
    # self-consistency on a sample of queries, exact integer arithmetic on the host
    sample_rows = rng.integers(0, n_eff, size=2000)
    sb = bank[sample_rows].astype(np.int32)
    for k in rng.integers(0, Q, size=40):
        qq = q[k].astype(np.int32)
        s_best = int(((bank[idx[k]].astype(np.int32) - qq) ** 2).sum())
        assert dist[k] == np.float32(s_best * (4.0 / (65025.0 * D)))
        s = ((sb - qq) ** 2).sum(axis=1)
        assert (s >= s_best).all()
        assert not ((s == s_best) & (sample_rows < idx[k])).any()
    # two shards, min-merged keys == unsharded, for all Q results
    qb = Bank.from_images(q, ctx)
    keys = None
    half = (n_eff // 2 // B) * B
    for lo, hi in ((half, n_eff), (0, half)):
        keys, _, _ = knn_keys(Bank.from_images(bank[lo:hi], ctx, index_base=lo), qb, keys=keys)
    d2, i2 = unpack_keys(ctx, keys, Q, D)
    assert np.array_equal(i2, idx) and np.array_equal(d2, dist)
    # host one-call ABI: H2D of bank + queries, prepare, search, D2H inside the call
    lib = _lib.load()
    p = ctypes.c_void_p
    d3 = np.empty(Q, np.float32)
    i3 = np.empty(Q, np.int64)
    t0 = time.perf_counter()
    rc = lib.gl_fbb_knn_l2_host(ctx.handle, bank.ctypes.data_as(p), N, q.ctypes.data_as(p), Q, D, B, d3.ctypes.data_as(p), i3.ctypes.data_as(p))
    t_host = time.perf_counter() - t0
    assert rc == 0, lib.gl_last_error()
    assert np.array_equal(i3, idx) and np.array_equal(d3, dist)
    print("\nconfig-2 size: attack() from host arrays %.3f s; gl_fbb_knn_l2_host (PCIe-inclusive) %.3f s = %.0f query-images/s"
          % (t_attack, t_host, Q / t_host))
