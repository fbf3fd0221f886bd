#!/usr/bin/env python3
"""Projected strong scaling of BASELINE configs[1] (L2) and configs[2] (0.2*LPIPS+L2) from ONE MI355X: the one-rank share of an N-way bank
shard -- bank / N samples, ALL 10 000 queries (replicated) -- timed with bench.py itself for N = 1, 2, 4, 8 on the same device.
What is missing against a real N-GPU run is the all-reduce(min) of the 80 KB of packed keys (latency-bound, tens of microseconds) and the
device-to-device clock spread (which bench.py --gpus N balances by sizing the shards).  The projection t(1) / t(N) shows how much of the step
does not shrink with N: the query preparation (L2: 0.1 ms) or the VGG16 features of the replicated queries (l2-lpips: ~65 ms of a 1.6 s step).

    python tools/bench_strong_scaling_shares.py [--steps 5] > profiles/rNN/rNN_strong_scaling_shares.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(distance, bank, steps, queries):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--distance", distance, "--bank", str(bank), "--queries", str(queries), "--steps", str(steps),
           "--warmup", "1", "--cpu-queries", "0", "--check-queries", "0", "--secondary", "off"]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, cwd=ROOT).stdout.decode()
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--bank", type=int, default=100000)
    a = ap.parse_args()
    n_eff = (a.bank // 64) * 64
    res = {"note": __doc__.split("\n\n")[0], "queries": a.queries, "bank_used": n_eff, "configs": {}}
    for name, distance, steps in (("configs[1] DCGAN-64 10k x 100k L2", "l2", a.steps), ("configs[2] DCGAN/WGAN-GP-64 10k x 100k 0.2*LPIPS+L2", "l2-lpips", max(2, a.steps // 2))):
        rows = []
        for n in (1, 2, 4, 8):
            share = n_eff // n
            d = run(distance, share, steps, a.queries)
            rows.append({"n_gpus": n, "shard_rows": share, "ms_per_step_one_rank": d["ms_per_step"], "phases_ms": d["phases_ms_per_step_rank0"]})
            print("%s N=%d share=%d: %.2f ms" % (name, n, share, d["ms_per_step"]), file=sys.stderr, flush=True)
        t1 = rows[0]["ms_per_step_one_rank"]
        for r in rows:
            r["projected_speedup"] = round(t1 / r["ms_per_step_one_rank"], 3)
            r["projected_efficiency"] = round(t1 / r["ms_per_step_one_rank"] / r["n_gpus"], 3)
            r["projected_query_images_per_s"] = round(a.queries / (r["ms_per_step_one_rank"] * 1e-3), 1)
        res["configs"][name] = rows
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
