#!/usr/bin/env python3
"""One rank's share of the 8-GPU configurations of BASELINE.json, timed on ONE MI355X (the per-GPU work of an 8-way bank shard; the
cross-GPU step that is missing here is the all-reduce(min) of Q packed keys, 80-400 KB):

  configs[3]  PGGAN-256 (in_channels 512, steps 6), 10 000 queries x 1 000 000 samples / 8 = 125 000-sample shard, 0.2*LPIPS+L2 at
              256 x 256: bank generated, featurised, searched and dropped chunk by chunk (GeneratedBank), queries in slices
  configs[4]  image half:   VAEGAN-64, 50 000 queries x 125 000-sample shard, exact L2 on 8-bit codes
              tabular half: medGAN rows (F = 1071 binary columns), 50 000 queries x 125 000-sample shard: the exact int8 path, and
                            float_path='mfma' on continuous rows of the same shape (the fp16-MFMA route the config names)

    python tools/bench_shard_configs.py [--config 3|4|all] [--scale 1.0]       (--scale shrinks queries and shard for a quick look)
One JSON line per measurement.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="all")
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()
    import ganleaks_amd as gl
    from ganleaks_amd.attack import GeneratedBank
    synth = gl.synth
    ctx = gl.Context.get()

    def emit(**kw):
        print(json.dumps(kw), flush=True)

    if args.config in ("3", "all"):
        from ganleaks_amd.gan_models.pggan.model_torch import Generator as PGGAN
        from ganleaks_amd.lpips import LpipsModel
        Q, N = int(10000 * args.scale), int(125000 * args.scale)
        gen = PGGAN(512, 512, 3)
        gen.load_state_dict(synth.pggan_state_dict(1, 512, 512))
        lin = np.load(os.path.join(ROOT, "tests", "golden", "lpips_lin_v0.1.npz"))
        model = LpipsModel(ctx).load_state_dicts(synth.vgg16_state_dict(7), {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
        z = synth.latent(1, N, 512)
        # queries: PGGAN samples of another latent stream, half of them planted copies of bank samples (so the result can be checked)
        n_eff = (N // 64) * 64
        nq_gen = min(Q, 256)
        planted = np.linspace(0, n_eff - 1, nq_gen // 2).astype(np.int64)
        qz = np.concatenate([z[planted], synth.latent(2, nq_gen - len(planted), 512)])
        q_small = gen.generate_u8(qz, steps=6, alpha=1.0).numpy()
        queries = np.tile(q_small, ((Q + nq_gen - 1) // nq_gen, 1, 1, 1))[:Q]         # 10 000 rows (repeats: the cost does not depend on the values)
        ctx.prof_reset()
        ctx.prof_enable(True)
        ctx.sync()
        t0 = time.perf_counter()
        d, i = gl.attack(queries, GeneratedBank(gen, z, steps=6, alpha=1.0), distance="l2-lpips", batch_size=64, lpips=model)
        ctx.sync()
        dt = time.perf_counter() - t0
        ctx.prof_enable(False)
        prof = ctx.prof_read()
        ctx.prof_reset()
        K1 = int(ctx.lib.gl_lpips_lattice_dim(256, 256))          # 8-bit images on both sides: lattice search rows
        K_alg = int(ctx.lib.gl_lpips_feature_dim(256, 256))
        knn_ms, knn_n = prof["feat_knn"]
        conv_ms, conv_n = prof["gather_conv"]
        ok = bool(np.array_equal(i[:len(planted)], np.minimum(planted, n_eff - 1))) if planted.max() < n_eff else None
        row_bytes = 2 * K1
        from ganleaks_amd.attack import _budget_bytes, _query_budget_bytes
        slices = int(np.ceil(Q * row_bytes / _query_budget_bytes(_budget_bytes(), ctx)))
        emit(config="configs[3] one rank of 8: PGGAN-256, %d queries x %d-sample shard, 0.2*LPIPS+L2 at 256x256, streamed" % (Q, N),
             seconds=round(dt, 2), query_images_per_s_this_rank=round(Q / dt, 2), planted_found=ok,
             feat_knn={"launches": int(knn_n), "total_s": round(knn_ms / 1e3, 2), "alg_tflops": round(2.0 * Q * n_eff * K_alg / (knn_ms * 1e-3) / 1e12, 1),
                       "frac_of_fp16_peak": round(2.0 * Q * n_eff * K_alg / (knn_ms * 1e-3) / 2.5e15, 4), "K_search_row": K1},
             convolutions={"launches": int(conv_n), "total_s": round(conv_ms / 1e3, 2),
                           "note": "PGGAN-256 generator (56.3 GFLOP / image) + VGG16 at 256x256 (40.1 GFLOP / image) for the shard, once per query slice"},
             query_slices=slices, note="a search row is %.1f MB; the %d query rows (%.0f GiB) %s; the bank shard is generated and featurised once per "
             "query slice, in chunks" % (row_bytes / 1e6, Q, Q * row_bytes / 2 ** 30, "stay resident" if slices == 1 else "go in %d slices" % slices))
        del gen, model, queries

    if args.config in ("4", "all"):
        from ganleaks_amd.attack import Bank
        from ganleaks_amd.gan_models.vaegan.train import Generator as VAEGAN
        from ganleaks_amd.gan_models.medgan.model import Autoencoder, Generator as MedG, generate_synthetic
        Q, N = int(50000 * args.scale), int(125000 * args.scale)
        gen = VAEGAN(100, 64)
        gen.load_state_dict(synth.vaegan_state_dict(777, 100, 64))
        z = synth.latent(3, N)
        queries = gen.generate_u8(synth.latent(4, Q)).numpy()
        gl.attack(queries[:64], GeneratedBank(gen, z[:4096]), batch_size=64)          # warm-up
        ctx.prof_reset()
        ctx.prof_enable(True)
        ctx.sync()
        t0 = time.perf_counter()
        d, i = gl.attack(queries, GeneratedBank(gen, z), batch_size=64)
        ctx.sync()
        dt = time.perf_counter() - t0
        ctx.prof_enable(False)
        prof = ctx.prof_read()
        ctx.prof_reset()
        n_eff = (N // 64) * 64
        emit(config="configs[4] image half, one rank of 8: VAEGAN-64 generator -> 8-bit bank, %d queries x %d-sample shard, exact L2" % (Q, N),
             seconds=round(dt, 3), query_images_per_s_this_rank=round(Q / dt, 1),
             l2_knn={"total_ms": round(prof["l2_knn"][0], 2), "launches": int(prof["l2_knn"][1]),
                     "tops": round(2.0 * Q * n_eff * 12288 / (prof["l2_knn"][0] * 1e-3) / 1e12, 1),
                     "frac_of_int8_peak": round(2.0 * Q * n_eff * 12288 / (prof["l2_knn"][0] * 1e-3) / 5e15, 4)},
             generator_ms=round(prof["gather_conv"][0] + prof["convt_rgb"][0], 1))
        del gen, queries
        gsd, asd = synth.medgan_state_dicts(555, 1071)
        mg = MedG(128, 128)
        mg.load_state_dict(gsd)
        ae = Autoencoder(1071, 128, binary=True)
        ae.load_state_dict(asd)
        rng = np.random.default_rng(9)
        t0 = time.perf_counter()
        bank = generate_synthetic(mg, ae, rng.standard_normal((N, 128)).astype(np.float32))
        t_gen = time.perf_counter() - t0
        qrows = generate_synthetic(mg, ae, rng.standard_normal((Q, 128)).astype(np.float32))
        for label, b, q, kw in (("binary rows, exact int8 path", bank, qrows, {}),
                                ("continuous rows, float_path='mfma' (split-fp16 on the matrix cores)",
                                 bank + rng.normal(0, 0.01, bank.shape).astype(np.float32), qrows + rng.normal(0, 0.01, qrows.shape).astype(np.float32),
                                 {"float_path": "mfma"})):
            pb, pq = Bank.from_images(b, ctx), Bank.from_images(q, ctx)
            gl.attack(pq, pb, batch_size=64, **kw)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                d, i = gl.attack(pq, pb, batch_size=64, **kw)
            ctx.sync()
            dt = (time.perf_counter() - t0) / 5
            emit(config="configs[4] tabular half, one rank of 8: medGAN rows F=1071, %d queries x %d-sample shard, %s" % (Q, N, label),
                 search_ms=round(dt * 1e3, 2), query_rows_per_s_this_rank=round(Q / dt, 1), bank_kind=pb.kind,
                 rate_tops=round(2.0 * Q * ((N // 64) * 64) * 1071 / dt / 1e12, 1), generate_and_decode_s_host_roundtrip=round(t_gen, 2))


if __name__ == "__main__":
    main()
