"""GPU tests of the streamed form of attack(): the bank passes through HBM chunk by chunk (GeneratedBank, or any bank whose
prepared rows exceed `chunk_bytes`) and must give exactly what the resident form gives -- which the other GPU test files pin
to the oracle and to the reference's golden vectors."""
import os

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    return ganleaks_amd


def test_l2_streamed_equals_resident(gl, synth, oracle):
    case = synth.attack_case(61, 300, 20, 20, 16)
    bank, q = case["bank"], np.concatenate([case["pos"], case["neg"]])
    d0, i0 = gl.attack(q, bank, distance="l2", batch_size=32)
    row = 2 * bank[0].size
    for rows_per_chunk in (1, 37, 288, 10000):
        d, i = gl.attack(q, bank, distance="l2", batch_size=32, chunk_bytes=rows_per_chunk * row)
        assert np.array_equal(i, i0) and np.array_equal(d, d0)
    assert i0.max() < 288                                          # BATCH_SIZE truncation (fbb.py:77) also when streamed
    # float images on the lattice stream through the exact path; one off-lattice pixel in a LATE chunk moves everything to fp32
    bank_f = oracle.dequantize_u8(bank)
    d, i = gl.attack(q, bank_f, distance="l2", batch_size=32, chunk_bytes=50 * row)
    assert np.array_equal(i, i0) and np.array_equal(d, d0)
    bank_f[250, 0, 3, 3] += 1e-3
    dr, ir = gl.attack(q, bank_f, distance="l2", batch_size=32)                               # resident: fp32 path
    ds, is_ = gl.attack(q, bank_f, distance="l2", batch_size=32, chunk_bytes=50 * 4 * bank[0].size)
    assert np.array_equal(is_, ir) and np.array_equal(ds, dr)
    with pytest.raises(ValueError):
        gl.attack(q, bank[:31], distance="l2", batch_size=32, chunk_bytes=row)


def test_generated_bank_never_materialised(gl, synth):
    """attack(queries, GeneratedBank(generator, z)) == attack(queries, generator.generate_u8(z)); shards by index_base"""
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    gen = Generator(100, 3, 16)
    gen.load_state_dict(synth.dcgan_state_dict(1234, features_g=16))
    z = synth.latent(1, 200)
    bank = gen.generate_u8(z)
    q = synth.perturb_u8(5, gen.generate_u8(synth.latent(2, 24)).numpy(), 6.0)
    d0, i0 = gl.attack(q, bank, distance="l2", batch_size=64)
    d, i = gl.attack(q, gl.GeneratedBank(gen, z), distance="l2", batch_size=64, chunk_bytes=50 * 2 * 12288)
    assert np.array_equal(i, i0) and np.array_equal(d, d0) and i.max() < 192
    # two shards of the truncated range [0, 192), merged by the caller's reduction (here: the second call continues the first's keys)
    from ganleaks_amd.attack import Bank, knn_keys, unpack_keys
    ctx = gl.Context.get()
    qb = Bank.from_images(q, ctx, keep_u8=True)
    keys = None
    for lo, hi in ((96, 192), (0, 96)):
        gb = gl.GeneratedBank(gen, z[lo:hi], index_base=lo)
        keys, _, _ = knn_keys(Bank.from_images(gb.rows(0, hi - lo), ctx, index_base=lo), qb, keys=keys)
    ds, is_ = unpack_keys(ctx, keys, qb.n, qb.d, "u8")
    assert np.array_equal(is_, i0) and np.array_equal(ds, d0)
    # a shard with index_base > 0 is not truncated again and reports global indices
    d1, i1 = gl.attack(q, gl.GeneratedBank(gen, z[96:192], index_base=96), distance="l2", batch_size=64, chunk_bytes=40 * 2 * 12288)
    assert i1.min() >= 96 and i1.max() < 192
    with pytest.raises(TypeError):
        gl.GeneratedBank(object(), z)


def test_l2_lpips_streamed_equals_resident(gl, synth, golden_dir):
    from ganleaks_amd.lpips import LpipsModel
    z = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    model = LpipsModel().load_state_dicts(synth.vgg16_state_dict(7), {"lin%d" % i: z["lin%d" % i] for i in range(5)})
    case = synth.attack_case(63, 70, 6, 6, 32, sigma=20.0)
    bank, q = case["bank"], np.concatenate([case["pos"], case["neg"]])
    for rows in ("fp16", "split"):
        model.search_rows = rows
        d0, i0 = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=model)
        row_bytes = (2 * int(gl.Context.get().lib.gl_lpips_search_dim(32, 32))) if rows == "fp16" else 4 * int(gl.Context.get().lib.gl_lpips_feature_dim(32, 32))
        d, i = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=model, chunk_bytes=23 * row_bytes)
        assert np.array_equal(i, i0) and np.array_equal(d, d0) and i.max() < 64
        # a budget smaller than the query rows: the queries go in slices of 5, each against the whole bank stream
        d, i = gl.attack(q, bank, distance="l2-lpips", batch_size=16, lpips=model, chunk_bytes=5 * row_bytes)
        assert np.array_equal(i, i0) and np.array_equal(d, d0)
    model.search_rows = "fp16"


def test_attack_on_devices_in_one_process(gl, synth):
    """shard.attack_on_devices: a host thread and a context per device, keys merged on the host.  One GPU here, so the 'devices' are
    [0, 0, 0]: three contexts (three streams) on the same device; the result must equal the single-context one bit for bit."""
    from ganleaks_amd import shard
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    sd = synth.dcgan_state_dict(1234, features_g=16)

    def make(ctx):
        g = Generator(100, 3, 16, ctx)
        g.load_state_dict(sd)
        return g

    z = synth.latent(1, 500)
    g0 = make(gl.Context.get())
    q = synth.perturb_u8(5, g0.generate_u8(synth.latent(2, 40)).numpy(), 6.0)
    d0, i0 = gl.attack(q, g0.generate_u8(z), distance="l2", batch_size=64)
    d, i = shard.attack_on_devices(q, make, z, devices=[0, 0, 0], distance="l2", batch_size=64)
    assert np.array_equal(i, i0) and np.array_equal(d, d0) and i.max() < 448
    d, i = shard.attack_on_devices(q, make, z, devices=[0, 0], batch_size=64, weights=[1.0, 3.0])
    assert np.array_equal(i, i0) and np.array_equal(d, d0)
    with pytest.raises(ValueError):
        shard.attack_on_devices(q, make, z[:10], devices=[0], batch_size=64)
