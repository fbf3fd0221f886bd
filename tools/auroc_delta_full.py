#!/usr/bin/env python3
"""End-to-end parity at BASELINE config-2 size: the whole path on the GPU against a CPU replay of the reference's arithmetic.

  reference side (host CPU, torch fp32 exactly as gan_models/dcgan/model_torch.py:75-96 runs it in eval mode):
      z -> ConvTranspose2d / BatchNorm2d / ReLU x4 -> ConvTranspose2d -> Tanh -> (x+1)/2 -> ToPILImage bytes  (the bank)
      per query: exact integer L2 over the bank (oracle/fbb_oracle.c; equals the reference's fp32 mean((y-x)^2) to 6e-8), argmin
  device side: gl_dcgan_forward (split-fp16) -> bank -> gl_l2_knn_i8.
Same weights, z and queries (the bench's synthetic ones).  Reports how many bank bytes differ (fp32 summation order at code
boundaries), per-query distance and index differences, and the AUROC of both.  Not part of bench.py: the CPU side takes minutes.

    python tools/auroc_delta_full.py [--bank 100000] [--queries 10000] > profiles/rNN/auroc_delta_full.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ganleaks_amd as gl  # noqa: E402
from ganleaks_amd.attack_models.eval_roc import plot_roc  # noqa: E402
from ganleaks_amd.gan_models.dcgan.model_torch import Generator  # noqa: E402
import c_oracle  # noqa: E402
import oracle as np_oracle  # noqa: E402


def reference_generator(sd, z, threads, batch=2048):
    """the reference's Generator.forward in eval mode, torch fp32 on the CPU"""
    torch.set_num_threads(threads)
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    out = np.empty((len(z), 3, 64, 64), np.float32)
    with torch.no_grad():
        for lo in range(0, len(z), batch):
            x = torch.from_numpy(z[lo:lo + batch])
            for l in range(4):
                x = F.conv_transpose2d(x, t["gen.%d.0.weight" % l], None, 1 if l == 0 else 2, 0 if l == 0 else 1)
                x = F.batch_norm(x, t["gen.%d.1.running_mean" % l], t["gen.%d.1.running_var" % l], t["gen.%d.1.weight" % l], t["gen.%d.1.bias" % l],
                                 False, 0.1, 1e-5)
                x = F.relu(x)
            x = torch.tanh(F.conv_transpose2d(x, t["gen.4.weight"], t["gen.4.bias"], 2, 1))
            out[lo:lo + batch] = x.numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bank", type=int, default=100000)
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    synth = gl.synth
    N, Q, B = a.bank, a.queries, 64
    sd = synth.dcgan_state_dict(1234)
    g = Generator(100, 3, 64)
    g.load_state_dict(sd)
    z = synth.latent(1, N)
    n_pos = Q // 2
    pos = synth.perturb_u8(5, g.generate_u8(synth.latent(2, n_pos)).numpy(), 0.05 * 127.5)
    neg = synth.perturb_u8(6, g.generate_u8(synth.latent(3, Q - n_pos)).numpy(), 0.10 * 127.5)
    q = np.concatenate([pos, neg])

    t0 = time.time()
    bank_dev = g.generate_u8(z)
    dist, idx = gl.attack(q, bank_dev, distance="l2", batch_size=B)
    t_dev = time.time() - t0
    bank = bank_dev.numpy()

    t0 = time.time()
    ref_f32 = reference_generator(sd, z, a.threads)
    t_gen = time.time() - t0
    ref_bank = np_oracle.quantize_to_u8(ref_f32)
    gen_err = float(np.abs(ref_f32[:4096] - g.forward_device(z[:4096], True, False)[0].numpy()).max())
    del ref_f32
    t0 = time.time()
    ref_dist, ref_idx, _ = c_oracle.knn_l2_u8(ref_bank, q, B)
    t_knn = time.time() - t0

    def auc(d):
        return plot_roc(-d[:n_pos].astype(np.float64), -d[n_pos:].astype(np.float64))[3]

    differing = int((bank != ref_bank).sum())
    res = {
        "config": {"bank": N, "queries": Q, "batch_size": B, "generator": "DCGAN-64, synthetic weights seed 1234"},
        "bank_bytes_differing": differing, "bank_bytes_total": int(bank.size), "bank_bytes_differing_frac": differing / bank.size,
        "max_abs_code_diff": int(np.abs(bank.astype(np.int16) - ref_bank.astype(np.int16)).max()),
        "generator_max_abs_err_first_4096": gen_err,
        "dist_max_abs_diff": float(np.abs(dist.astype(np.float64) - ref_dist).max()),
        "idx_mismatches": int((idx != ref_idx).sum()),
        "auroc_device": auc(dist), "auroc_reference_cpu": auc(ref_dist), "auroc_abs_delta": abs(auc(dist) - auc(ref_dist)),
        "seconds": {"device_whole_path_incl_transfers": round(t_dev, 2), "cpu_reference_generator": round(t_gen, 1), "cpu_reference_search": round(t_knn, 1),
                    "cpu_threads": a.threads},
    }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
