#!/usr/bin/env python3
"""A/B timing of the pairwise kernels in ONE process (variants alternate round by round, random operands):
    python tools/bench_pairwise.py [--variants 0,1] [--rounds 5] [--l2] [--feat] [--feat-bank 25000]
Prints one JSON line per (kernel, variant) with the median / min launch time and the algorithmic rate."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the GL_PAIR_VARIANT switch and the experiment kernels exist only in the tuning build (make -C gan-leaks_amd/csrc tuning)
os.environ.setdefault("GANLEAKS_LIB", os.path.join(ROOT, "gan-leaks_amd", "libganleaks_hip_tuning.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--l2", action="store_true")
    ap.add_argument("--feat", action="store_true")
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--bank", type=int, default=99968)
    ap.add_argument("--feat-bank", type=int, default=25000)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--k", type=int, default=0, help="halves per search row instead of the LPIPS row of --res (multiple of 64)")
    ap.add_argument("--pad", type=int, default=0, help="extra halves per search row (row stride experiment)")
    args = ap.parse_args()
    if not (args.l2 or args.feat):
        args.l2 = args.feat = True
    variants = [int(v) for v in args.variants.split(",")]
    import torch
    import ganleaks_amd as gl
    from ganleaks_amd._lib import check
    ctx = gl.Context.get()
    lib = ctx.lib
    p = ctypes.c_void_p
    ev = [p(), p()]
    for e in ev:
        check(lib.gl_event_create(ctypes.byref(e)))

    def timed(fn):
        check(lib.gl_event_record(ctx.handle, ev[0]))
        fn()
        check(lib.gl_event_record(ctx.handle, ev[1]))
        ms = ctypes.c_float()
        check(lib.gl_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
        return ms.value

    def run(name, launch, work, unit):
        times = {v: [] for v in variants}
        ref_keys = None
        for r in range(args.rounds + 1):
            for v in variants:
                os.environ["GL_PAIR_VARIANT"] = str(v)
                ms, keys = launch()
                if r:
                    times[v].append(ms)
                if ref_keys is None:
                    ref_keys = keys
                elif not np.array_equal(ref_keys, keys) and r == 0:
                    note = {"kernel": name, "variant": v, "note": "keys differ from variant %d" % variants[0], "n_diff": int((ref_keys != keys).sum())}
                    if unit == "TFLOP/s":       # float keys: distance bits << 32 | index; other accumulation orders move the low bits
                        d0 = (ref_keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
                        d1 = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
                        note["max_rel_dist_diff"] = float(np.max(np.abs(d0 - d1) / np.maximum(d0, 1e-30)))
                        note["idx_diff"] = int(((ref_keys ^ keys) & np.uint64(0xFFFFFFFF) != 0).sum())
                    print(json.dumps(note))
        for v in variants:
            t = np.array(times[v])
            print(json.dumps({"kernel": name, "variant": v, "median_ms": round(float(np.median(t)), 4), "min_ms": round(float(t.min()), 4),
                              "rate": round(work / (float(np.median(t)) * 1e-3) / 1e12, 1), "unit": unit, "rounds": args.rounds}), flush=True)

    if args.l2:
        Q, N, D = args.queries, args.bank, 3 * args.res * args.res
        g = torch.Generator(device="cuda").manual_seed(1)
        bank = torch.randint(0, 256, (N, D), dtype=torch.uint8, device="cuda", generator=g)
        qs = torch.randint(0, 256, (Q, D), dtype=torch.uint8, device="cuda", generator=g)
        torch.cuda.synchronize()
        stride = int(lib.gl_l2_row_stride(D))
        bi8, qi8 = ctx.empty((N, stride), np.int8), ctx.empty((Q, stride), np.int8)
        bn, qn = ctx.empty((N,), np.int32), ctx.empty((Q,), np.int32)
        check(lib.gl_l2_prepare(ctx.handle, p(bank.data_ptr()), N, D, p(bi8.ptr), p(bn.ptr)))
        check(lib.gl_l2_prepare(ctx.handle, p(qs.data_ptr()), Q, D, p(qi8.ptr), p(qn.ptr)))
        keys = ctx.empty((Q,), np.uint64)
        ctx.sync()
        del bank, qs

        def launch():
            check(lib.gl_keys_init(ctx.handle, p(keys.ptr), Q))
            ms = timed(lambda: check(lib.gl_l2_knn_i8(ctx.handle, p(bi8.ptr), p(bn.ptr), N, 0, p(qi8.ptr), p(qn.ptr), Q, D, p(keys.ptr))))
            return ms, keys.numpy().copy()
        run("l2_knn_i8 %dx%dx%d" % (Q, N, D), launch, 2.0 * Q * N * D, "TOP/s")
        del bi8, qi8

    if args.feat:
        Q, N = args.queries, args.feat_bank
        K1 = (args.k or int(lib.gl_lpips_search_dim(args.res, args.res))) + args.pad
        g = torch.Generator(device="cuda").manual_seed(2)
        # search rows hold V * 2^14 with V ~ 1e-3..1e-1: positive halves of order 1..1000
        def rows(n):        # filled in pieces: the fp32 temporaries of a 256 x 256 row set would not fit beside it
            out = torch.empty((n, K1), dtype=torch.float16, device="cuda")
            step = max(1, int(1e9 // K1))
            for a in range(0, n, step):
                out[a:a + step] = (torch.rand((min(step, n - a), K1), dtype=torch.float32, device="cuda", generator=g) * 64.0).to(torch.float16)
            return out
        bank, qs = rows(N), rows(Q)
        def sqn(x):      # |row|^2 in the units of the distance (rows hold V * 2^14)
            step = max(1, int(1e9 // K1))
            return torch.cat([(x[a:a + step].float() ** 2).sum(1) for a in range(0, len(x), step)]) / float(2 ** 28)
        bn, qn = sqn(bank), sqn(qs)
        torch.cuda.synchronize()
        keys = ctx.empty((Q,), np.uint64)

        def launch():
            check(lib.gl_keys_init(ctx.handle, p(keys.ptr), Q))
            ms = timed(lambda: check(lib.gl_feat_knn_h1(ctx.handle, p(bank.data_ptr()), p(bn.data_ptr()), N, 0, p(qs.data_ptr()), p(qn.data_ptr()), Q, K1,
                                                        p(keys.ptr))))
            return ms, keys.numpy().copy()
        run("feat_knn_h1 %dx%dx%d" % (Q, N, K1), launch, 2.0 * Q * N * K1, "TFLOP/s")


if __name__ == "__main__":
    main()
