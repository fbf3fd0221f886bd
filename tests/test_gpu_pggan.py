"""GPU parity tests of the PGGAN generator (fp32 MFMA convolutions, fused nearest upsampling) against the
reference Generator's outputs (tests/golden/pggan_gen.npz) and the fp64 oracle.  Tolerance 5e-5 on
outputs in (-1,1) (north_star: 1e-4)."""
import os

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu
ATOL = 5e-5


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    return ganleaks_amd


@pytest.mark.parametrize("precision", [1, 0])
def test_generator_matches_reference(precision, gl, synth, golden_dir):
    """precision 1 (default): split-fp16 convolutions; 0: fp32 MFMA"""
    from ganleaks_amd.gan_models.pggan.model_torch import Generator, stackGenerators
    g = np.load(os.path.join(golden_dir, "pggan_gen.npz"))
    gens = {}
    for ci, (z_dim, C, steps, alpha) in enumerate(g["cases"]):
        z_dim, C, steps = int(z_dim), int(C), int(steps)
        if (z_dim, C) not in gens:
            gen = Generator(z_dim, C, 3)
            assert "matched" in gen.load_state_dict(synth.pggan_state_dict(4321 + C, z_dim, C))
            gen.set_precision(precision)
            gens[(z_dim, C)] = gen
        out = gens[(z_dim, C)](synth.latent(3, 4, z_dim), steps, float(alpha))
        ref = g["case%d" % ci]
        assert out.shape == ref.shape and out.dtype == np.float32
        err = np.abs(out - ref).max()
        assert err < ATOL, (ci, err)
    st = stackGenerators(64, 64, 3, 2)
    sd = {}
    for gi in range(2):
        sd.update(synth.pggan_state_dict(100 + gi, 64, 64, prefix="gen.%d." % gi))
    st.load_state_dict(sd)
    assert np.abs(st(synth.latent(3, 4, 64), 2, 1.0, 1) - g["stack_g1"]).max() < ATOL


def test_codes_chunks_and_oracle(gl, synth, oracle):
    import pggan_oracle
    from ganleaks_amd.gan_models.pggan.model_torch import Generator
    sd = synth.pggan_state_dict(77, 128, 128)
    gen = Generator(128, 128, 3)
    gen.load_state_dict(sd)
    z = synth.latent(9, 21, 128)
    f32, u8 = gen.forward_device(z, 3, 1.0, True, True)
    f, u = f32.numpy(), u8.numpy()
    assert f.shape == (21, 3, 32, 32)
    assert np.array_equal(u, oracle.quantize_to_u8(f, "half"))           # pggan/train.py:238 mapping
    ref = pggan_oracle.pggan_forward(sd, z[:5], 3, 1.0)
    assert np.abs(f[:5] - ref).max() < ATOL
    gen2 = Generator(128, 128, 3)
    gen2.load_state_dict(sd)
    gen2.set_chunk(8)                                                     # 21 = 8 + 8 + 5
    assert np.array_equal(gen2(z, 3, 1.0), f)
    # fade-in branch and depth change on the same object
    ref = pggan_oracle.pggan_forward(sd, z[:3], 2, 0.25)
    assert np.abs(gen(z[:3], 2, 0.25) - ref).max() < ATOL
    ref = pggan_oracle.pggan_forward(sd, z[:2], 0, 1.0)
    assert np.abs(gen(z[:2], 0, 1.0) - ref).max() < ATOL
    with pytest.raises(gl.GanLeaksError):
        gen(z[:2], 6, 1.0)              # block 5 of a 128-channel model has 32 -> 16 channels: refused, not wrong
    with pytest.raises(KeyError):
        bad = dict(sd)
        del bad["prog_blocks.1.conv2.bias"]
        Generator(128, 128, 3).load_state_dict(bad)


def test_pggan_bank_attack(gl, synth):
    """generator-direct bank -> L2 1-NN, as the fbb attack would use it"""
    import c_oracle
    from ganleaks_amd.gan_models.pggan.model_torch import Generator
    gen = Generator(64, 64, 3)
    gen.load_state_dict(synth.pggan_state_dict(4321 + 64, 64, 64))
    bank = gen.generate_u8(synth.latent(11, 200, 64), steps=4, alpha=1.0)
    hb = bank.numpy()
    q = synth.perturb_u8(2, hb[[7, 150, 199]], 5.0)
    d, i = gl.attack(q, bank, batch_size=64)
    od, oi, _ = c_oracle.knn_l2_u8(hb, q, 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od) and i[0] == 7 and i[1] == 150 and i[2] != 199


def test_depth_changes_on_one_object(gl, synth):
    """steps=4 then steps=6 then steps=4 on the same generator: the workspace is re-sized per depth (a pass of the shallow depth
    must not be reused as the pass size of the deep one: its largest activation would exceed the 3 GiB addressing limit)"""
    from ganleaks_amd.gan_models.pggan.model_torch import Generator
    sd = synth.pggan_state_dict(11, 64, 256)
    g = Generator(64, 256, 3)
    g.load_state_dict(sd)
    z = synth.latent(12, 300, 64)
    a4 = g.forward_device(z, 4, 1.0, True, False)[0].numpy()
    a6 = g.forward_device(z[:70], 6, 1.0, True, False)[0].numpy()
    b4 = g.forward_device(z, 4, 1.0, True, False)[0].numpy()
    assert a6.shape == (70, 3, 256, 256) and np.array_equal(a4, b4)
    fresh = Generator(64, 256, 3)
    fresh.load_state_dict(sd)
    assert np.array_equal(fresh.forward_device(z[:70], 6, 1.0, True, False)[0].numpy(), a6)


@pytest.mark.parametrize("precision", [1, 0])
def test_generator_matches_reference_at_128_and_256(precision, gl, synth, golden_dir):
    """steps 5 and 6, alpha 1 and 0.4, against the reference's Generator (tests/golden/pggan_gen_big.npz); the last case is BASELINE
    configs[3]'s own channel plan (in_channels 512: 256 channels at 64 x 64, 128 at 128 x 128, 64 at 256 x 256)"""
    from ganleaks_amd.gan_models.pggan.model_torch import Generator
    g = np.load(os.path.join(golden_dir, "pggan_gen_big.npz"))
    gens = {}
    for ci, (z_dim, C, steps, alpha, n) in enumerate(g["cases"]):
        z_dim, C, steps, n = int(z_dim), int(C), int(steps), int(n)
        if (z_dim, C) not in gens:
            gen = Generator(z_dim, C, 3)
            assert "matched" in gen.load_state_dict(synth.pggan_state_dict(4321 + C, z_dim, C))
            gen.set_precision(precision)
            gens[(z_dim, C)] = gen
        out = gens[(z_dim, C)](synth.latent(5, n, z_dim), steps, float(alpha))
        ref = g["case%d" % ci]
        assert out.shape == ref.shape == (n, 3, 4 * 2 ** steps, 4 * 2 ** steps) and out.dtype == np.float32
        err = np.abs(out - ref).max()
        assert err < ATOL, (ci, err)


def test_pggan256_bank_attack_against_c_oracle(gl, synth):
    """configs[3]'s bank: PGGAN steps=6 -> 8-bit 256 x 256 bank -> exact L2 1-NN (D = 196 608: the 64-bit-total kernel), vs the C oracle"""
    import c_oracle
    from ganleaks_amd.gan_models.pggan.model_torch import Generator
    gen = Generator(64, 256, 3)
    gen.load_state_dict(synth.pggan_state_dict(4321 + 256, 64, 256))
    bank = gen.generate_u8(synth.latent(21, 70, 64), steps=6, alpha=1.0)
    hb = bank.numpy()
    assert hb.shape == (70, 3, 256, 256)
    q = np.concatenate([synth.perturb_u8(2, hb[[7, 50, 69]], 5.0), synth.lowpass_u8_images(6, 2, 256)])
    d, i = gl.attack(q, bank, batch_size=32)
    od, oi, _ = c_oracle.knn_l2_u8(hb, q, 32)
    assert np.array_equal(i, oi) and np.array_equal(d, od) and i[0] == 7 and i[1] == 50 and i[2] != 69


def test_configs3_shape_end_to_end_streamed(gl, synth, golden_dir):
    """configs[3] in small: PGGAN steps=6 generated chunk by chunk (GeneratedBank), VGG16 + LPIPS at 256 x 256 on lattice search rows, the bank
    streamed through HBM with the query rows resident; planted queries (copies of generated samples) come back with their index, and the
    result equals the one on the materialised bank bit for bit."""
    import os
    from ganleaks_amd.attack import GeneratedBank
    from ganleaks_amd.lpips import LpipsModel
    from ganleaks_amd.gan_models.pggan.model_torch import Generator
    gen = Generator(64, 256, 3)
    gen.load_state_dict(synth.pggan_state_dict(4321 + 256, 64, 256))
    lin = np.load(os.path.join(golden_dir, "lpips_lin_v0.1.npz"))
    model = LpipsModel().load_state_dicts(synth.vgg16_state_dict(7), {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
    z = synth.latent(31, 96, 64)
    whole = gen.generate_u8(z, steps=6, alpha=1.0).numpy()
    planted = np.array([3, 40, 95, 64])
    q = np.concatenate([whole[planted], synth.perturb_u8(3, whole[[10, 70]], 4.0)])
    row = 2 * int(gl.Context.get().lib.gl_lpips_lattice_dim(256, 256))
    d_res, i_res = gl.attack(q, whole, distance="l2-lpips", batch_size=32, lpips=model)
    assert np.array_equal(i_res[:4], planted) and np.all(d_res[:4] < 1e-5) and i_res[4] == 10 and i_res[5] == 70
    d, i = gl.attack(q, GeneratedBank(gen, z, steps=6, alpha=1.0), distance="l2-lpips", batch_size=32, lpips=model, chunk_bytes=20 * row)
    assert np.array_equal(i, i_res) and np.array_equal(d, d_res)
