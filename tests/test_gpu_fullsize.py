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


def test_config2_size_properties_l2_lpips(gl):
    """The reference's default distance (0.2 * LPIPS + L2) at the same size, seeded VGG16 + the vendored lin weights: planted copies come back
    with the smallest index holding those bytes and a distance that is zero up to the fp32 rounding of the norms (< 1e-5 at |V|^2 ~ 1); reported distances agree with the
    small-problem path (which the fp64 oracle and the reference's goldens pin) on the reported neighbour and on a random sample of other rows,
    none of which is closer; two shards min-merged, and the bank streamed in chunks, reproduce the resident result bit for bit."""
    import os
    from ganleaks_amd import synth
    from ganleaks_amd.lpips import LpipsModel, feat_knn_keys
    from ganleaks_amd.attack import unpack_keys
    N, Q, B = 100_000, 10_000, 64
    n_eff = (N // B) * B
    rng = np.random.default_rng(77)
    proto = rng.integers(0, 256, size=(2048, 3, 64, 64), dtype=np.uint8)
    bank = proto[rng.integers(0, 2048, size=N)]
    rows = rng.integers(0, N, size=N // 2)
    bank[rows, :, :8, :8] = rng.integers(0, 256, size=(len(rows), 3, 8, 8), dtype=np.uint8)
    q = bank[rng.integers(0, N, size=Q)].copy()
    planted = rng.integers(0, n_eff, size=500)
    q[:500] = bank[planted]
    q[500:, :, 8:24, 8:24] = rng.integers(0, 256, size=(Q - 500, 3, 16, 16), dtype=np.uint8)
    lin = np.load(os.path.join(os.path.dirname(__file__), "golden", "lpips_lin_v0.1.npz"))
    model = LpipsModel().load_state_dicts(synth.vgg16_state_dict(7), {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
    ctx = gl.Context.get()

    fq = model.features(q, role="query")
    fb = model.features(bank[:n_eff], role="bank")
    dist, idx = unpack_keys(ctx, feat_knn_keys(fb, fq), Q, fq.K, "f32")
    assert idx.max() < n_eff and idx.min() >= 0
    assert np.all(dist[:500] < 1e-5)          # |V|^2 ~ 1 for these noise images: the fp32 rounding of |q|^2 + |n|^2 - 2 q.n
    flat = bank[:n_eff].reshape(n_eff, -1)
    for k in range(0, 500, 25):
        same = np.flatnonzero((flat == q[k].reshape(-1)).all(axis=1))
        assert idx[k] == same[0]
    # self-consistency through the small-problem path: the reported neighbour + 63 random rows, one query at a time
    for k in rng.integers(0, Q, size=12):
        others = rng.integers(0, n_eff, size=63)
        cand = np.concatenate([[idx[k]], others])
        d_small, i_small = gl.attack(q[k:k + 1], bank[cand], distance="l2-lpips", batch_size=64, lpips=model)
        assert d_small[0] > dist[k] - 1e-5       # nothing in the sample is closer
        d_self, _ = gl.attack(q[k:k + 1], np.repeat(bank[idx[k]][None], 64, axis=0), distance="l2-lpips", batch_size=64, lpips=model)
        assert abs(float(d_self[0]) - float(dist[k])) < 1e-5
    # two shards, min-merged keys == unsharded, for all Q results
    half = (n_eff // 2 // B) * B
    keys = None
    for lo, hi in ((half, n_eff), (0, half)):
        keys = feat_knn_keys(model.features(bank[lo:hi], index_base=lo, role="bank"), fq, keys=keys)
    d2, i2 = unpack_keys(ctx, keys, Q, fq.K, "f32")
    assert np.array_equal(i2, idx) and np.array_equal(d2, dist)
    del fb
    # the bank streamed through HBM in chunks of ~24 GiB of feature rows
    d3, i3 = gl.attack(fq, bank, distance="l2-lpips", batch_size=B, lpips=model, chunk_bytes=24 << 30)
    assert np.array_equal(i3, idx) and np.array_equal(d3, dist)
