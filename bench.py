#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fbb attack path on MI355X.

Headline workload (BASELINE.json configs[1]): DCGAN-64 generator, 10 000 queries x 100 000-sample bank, L2.
One "step" = one full pass of the hot path with z, weights and the query codes already resident in HBM:
       z --split-fp16 MFMA generator--> 8-bit bank --prepare--> int8-MFMA pairwise L2 + argmin
       [--RCCL min over ranks-->] (dist[Q], idx[Q]) on the host.
value = queries / step time (whole job, all ranks).  N > 1 shards the bank (strong scaling: the problem is fixed), every rank keeps
all queries, one all-reduce(min) of Q packed keys -- issued by libganleaks_hip.so itself (gl_allreduce_min_keys: RCCL on the
context's stream); torch.distributed is the launcher's rendezvous only (gloo: carries the RCCL unique id, the barrier and the
max-over-ranks of the timing).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (fields: see the project brief); progress goes to stderr.  With the default flags on one GPU the line
also carries two further measurements of the same run:
    "secondary"       BASELINE configs[2]: the same 10k x 100k problem under the reference's own fbb distance 0.2*LPIPS + L2
                      (attack_models/fbb.py:148), with its own roofline, parity block and cpu_baseline
    "secondary_fp32"  the headline workload with strict fp32 MFMA products in the generator (--gen-precision 0)
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_GATHER_PER_IMG = 2.0 * (100 * 1024 * 16 + 16 * 1024 * 512 * 16 + 64 * 512 * 256 * 16 + 256 * 256 * 128 * 16)  # layers 0-3
F_RGB_PER_IMG = 2.0 * (1024 * 128 * 3 * 16)                                                                       # layer 4
F_VGG_PER_IMG = 2.0 * 1252.8e6   # 13 convs at 64x64
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: fp16/bf16 MFMA, dense
PEAK_I8_MFMA_TOPS = 5000.0       # 2 x the ~2.5 PF bf16 dense peak (same cycles at twice the K)
PEAK_HBM_GBS = 8000.0
D = 3 * 64 * 64


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Job:
    """what one process of the job holds: its rank, its GPU context, the launcher's process group and the native communicator"""
    pass


def setup_job(args):
    import torch
    import ganleaks_amd as gl
    from ganleaks_amd import shard
    job = Job()
    job.rank = int(os.environ.get("RANK", "0"))
    job.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    job.world = int(os.environ.get("WORLD_SIZE", "1"))
    if job.world != args.gpus:
        if job.world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed launcher (see module docstring)" % args.gpus)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, job.world))
    ndev = torch.cuda.device_count()
    share = args.collective == "gloo"                 # rehearsal: several ranks may share one GPU, keys are reduced through host memory
    job.dev = job.local_rank % max(ndev, 1) if share else job.local_rank
    if job.dev >= ndev:
        raise SystemExit("rank %d needs GPU %d but only %d visible" % (job.rank, job.dev, ndev))
    torch.cuda.set_device(job.dev)
    job.torch, job.gl, job.shard = torch, gl, shard
    job.ctx = gl.Context.get(job.dev)
    job.lib = job.ctx.lib
    job.dist, job.comm, job.nccl_group, job.collective = None, None, None, "none"
    if job.world > 1:
        import torch.distributed as dist
        job.dist = dist
        dist.init_process_group("gloo")               # rendezvous, barrier, timing max: CPU tensors only
        job.collective = args.collective
        if args.collective == "native":
            ok = 1
            try:
                job.comm = shard.make_comm(job.ctx)
                keys = job.ctx.to_device(np.full(1024, job.rank + 7, np.uint64))
                job.comm.allreduce_min_keys(keys)      # RCCL connects lazily on the first collective: part of setup
                ok = int(keys.numpy()[0] == 7)
            except Exception as e:  # noqa: BLE001
                log("[rank %d] native RCCL communicator failed: %s" % (job.rank, e))
                ok = 0
            t = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                log("[rank %d] falling back to torch.distributed's nccl backend for the key reduction" % job.rank)
                job.comm = None
                job.collective = "torch"
        if job.collective == "torch":
            job.nccl_group = dist.new_group(backend="nccl")
            warm = torch.zeros(1024, dtype=torch.int64, device="cuda:%d" % job.dev)
            dist.all_reduce(warm, op=dist.ReduceOp.MIN, group=job.nccl_group)
            torch.cuda.synchronize()
    return job


def measure(job, args, distance, gen_precision, steps, warmup, cpu_leg, headline):
    """one workload on this job: returns the JSON line (rank 0) or None"""
    torch, gl, shard, ctx, lib, dist = job.torch, job.gl, job.shard, job.ctx, job.lib, job.dist
    from ganleaks_amd._lib import check
    from ganleaks_amd.attack_models.eval_roc import plot_roc
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    p = ctypes.c_void_p
    synth = gl.synth
    rank, world = job.rank, job.world
    Q, N, B = args.queries, args.bank, args.batch_size
    bounds = shard.shard_bounds(N, B, world)
    n_eff = bounds[-1]
    lo, hi = bounds[rank], bounds[rank + 1]
    n_loc = hi - lo
    lpips_mode = distance == "l2-lpips"

    # ---------------------------------------------------------------- setup (untimed)
    t_setup = time.time()
    sd = synth.dcgan_state_dict(1234)
    gen = Generator(100, 3, 64, ctx)
    gen.load_state_dict(sd)
    if args.chunk:
        gen.set_chunk(args.chunk)
    gen.set_precision(gen_precision)
    z_all = synth.latent(1, N)                               # the bank's latents; bank index = z index
    shard_note = "equal shards"
    if world > 1 and not args.no_balance:
        # The GPUs of a node do not run matrix-core loops at the same clock (MI355X_MICROARCH.md: 12 % device-to-device) and the slowest rank
        # sets the step time.  Setup: every rank times the generator on the same 4096 latents, the bank is then cut in proportion to the
        # measured rates.  The result does not depend on the cut (exact keys + min).
        zc = ctx.to_device(z_all[:4096].reshape(-1, 100))
        uc = ctx.empty((len(zc), 3, 64, 64), np.uint8)
        e0, e1 = ctx.event(), ctx.event()
        times = []
        for it in range(7):
            e0.record()
            check(lib.gl_dcgan_forward(gen._handle, p(zc.ptr), len(zc), p(0), p(uc.ptr)))
            e1.record()
            ctx.sync()
            if it >= 2:
                times.append(e0.elapsed_ms_until(e1))
        mine = torch.tensor([float(np.median(times))], dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        ms = [float(t.item()) for t in allt]
        bounds = shard.weighted_bounds(n_eff, [1.0 / t for t in ms], B)
        lo, hi = bounds[rank], bounds[rank + 1]
        n_loc = hi - lo
        shard_note = "shards sized by measured generator speed (%s ms per 4096 images)" % ", ".join("%.2f" % t for t in ms)
        del zc, uc
    z_dev = ctx.to_device(z_all[lo:hi].reshape(n_loc, 100))
    # queries: both classes are fresh generator samples (z streams disjoint from the bank's) with pixel noise; members get the smaller noise,
    # so they sit closer to the bank on average and the AUROC is non-degenerate.  (SURVEY 8(d) proposed low-pass random images as
    # negatives; generator samples keep both classes on the generator's manifold, which is the harder and more realistic case.
    # Throughput does not depend on the choice.)
    n_pos = Q // 2
    pos = synth.perturb_u8(5, gen.generate_u8(synth.latent(2, n_pos)).numpy(), 0.05 * 127.5)
    neg = synth.perturb_u8(6, gen.generate_u8(synth.latent(3, Q - n_pos)).numpy(), 0.10 * 127.5)
    queries_u8 = np.concatenate([pos, neg])
    q_dev = ctx.to_device(queries_u8.reshape(Q, D))

    lp_model = None
    vgg_sd = lin = None
    if lpips_mode:
        from ganleaks_amd.lpips import LpipsModel
        lin = np.load(os.path.join(ROOT, "tests", "golden", "lpips_lin_v0.1.npz"))
        vgg_sd = synth.vgg16_state_dict(7)
        lp_model = LpipsModel(ctx).load_state_dicts(vgg_sd, {"lin%d" % i: lin["lin%d" % i] for i in range(5)})
        lp_model.set_precision(gen_precision)
        KF_ALG = int(lib.gl_lpips_feature_dim(64, 64))          # 512 000: the contraction length the roofline is priced on
        h1 = args.feat_rows == "fp16"
        # 8-bit images on both sides: lattice search rows (image part exact in one fp16 K segment), KF = KF_ALG = 512 000 halves
        KF = int(lib.gl_lpips_lattice_dim(64, 64)) if h1 else KF_ALG
        row_scale = float(lib.gl_lpips_lattice_scale(64, 64))
        row_dtype = np.float16 if h1 else np.float32
        need_gb = (n_loc + Q) * KF * np.dtype(row_dtype).itemsize / 1e9
        log("[rank %d] l2-lpips: feature vectors need %.1f GB of HBM" % (rank, need_gb))
        bank_V = ctx.empty((n_loc, KF), row_dtype)
        bank_Vn = ctx.empty((n_loc,), np.float32)
        # N > 1 with the native communicator: the query features are sharded too (rank r featurises Q / N queries, two all-gathers);
        # everything else keeps all queries replicated
        q_shard = h1 and job.comm is not None and world > 1
        q_per = -(-Q // world) if q_shard else Q
        q_lo, q_hi = (min(rank * q_per, Q), min((rank + 1) * q_per, Q)) if q_shard else (0, Q)
        q_V = ctx.empty((q_per * world if q_shard else Q, KF), row_dtype)
        q_Vn = ctx.empty((q_per * world if q_shard else Q,), np.float32)

    stride = int(lib.gl_l2_row_stride(D))
    bank_u8 = ctx.empty((n_loc, D), np.uint8)
    if not lpips_mode:
        bank_i8 = ctx.empty((n_loc, stride), np.int8)
        bank_nrm = ctx.empty((n_loc,), np.int32)
        q_i8 = ctx.empty((Q, stride), np.int8)
        q_nrm = ctx.empty((Q,), np.int32)
    keys = ctx.empty((Q,), np.uint64)
    dist_dev = ctx.empty((Q,), np.float32)
    idx_dev = ctx.empty((Q,), np.int64)
    out_dist = np.empty(Q, np.float32)
    out_idx = np.empty(Q, np.int64)
    keys_t = None
    if job.collective == "torch":
        keys_t = torch.as_tensor(keys.view((Q,), np.int64), device="cuda:%d" % job.dev)
    log("[rank %d] %s setup %.1fs: bank rows [%d,%d) of n_eff=%d, %d queries" % (rank, distance, time.time() - t_setup, lo, hi, n_eff, Q))

    ev = [ctx.event() for _ in range(4)]

    def step(timed_phases=None):
        if timed_phases is not None:
            ev[0].record()
        check(lib.gl_dcgan_forward(gen._handle, p(z_dev.ptr), n_loc, p(0), p(bank_u8.ptr)))
        if timed_phases is not None:
            ev[1].record()
        if lp_model is None:
            check(lib.gl_l2_prepare(ctx.handle, p(bank_u8.ptr), n_loc, D, p(bank_i8.ptr), p(bank_nrm.ptr)))
            check(lib.gl_l2_prepare(ctx.handle, p(q_dev.ptr), Q, D, p(q_i8.ptr), p(q_nrm.ptr)))
            check(lib.gl_keys_init(ctx.handle, p(keys.ptr), Q))
            check(lib.gl_l2_knn_i8(ctx.handle, p(bank_i8.ptr), p(bank_nrm.ptr), n_loc, lo, p(q_i8.ptr), p(q_nrm.ptr), Q, D, p(keys.ptr)))
        else:
            if h1:
                check(lib.gl_lpips_lattice_features_u8(lp_model._handle, p(bank_u8.ptr), n_loc, 64, 64, p(bank_V.ptr), p(bank_Vn.ptr)))
                if q_shard:
                    if q_hi > q_lo:
                        check(lib.gl_lpips_lattice_features_u8(lp_model._handle, p(q_dev.ptr + q_lo * D), q_hi - q_lo, 64, 64,
                                                               p(q_V.ptr + q_lo * KF * 2), p(q_Vn.ptr + q_lo * 4)))
                    job.comm.allgather_rows(q_V, q_per * KF * 2)
                    job.comm.allgather_rows(q_Vn, q_per * 4)
                else:
                    check(lib.gl_lpips_lattice_features_u8(lp_model._handle, p(q_dev.ptr), Q, 64, 64, p(q_V.ptr), p(q_Vn.ptr)))
            else:
                check(lib.gl_lpips_features_u8(lp_model._handle, p(bank_u8.ptr), n_loc, 64, 64, p(bank_V.ptr), p(bank_Vn.ptr)))
                check(lib.gl_lpips_features_u8(lp_model._handle, p(q_dev.ptr), Q, 64, 64, p(q_V.ptr), p(q_Vn.ptr)))
            check(lib.gl_keys_init(ctx.handle, p(keys.ptr), Q))
            if h1:
                check(lib.gl_feat_knn_h1_scaled(ctx.handle, p(bank_V.ptr), p(bank_Vn.ptr), n_loc, lo, p(q_V.ptr), p(q_Vn.ptr), Q, KF, p(keys.ptr), row_scale))
            else:
                check(lib.gl_feat_knn(ctx.handle, p(bank_V.ptr), p(bank_Vn.ptr), n_loc, lo, p(q_V.ptr), p(q_Vn.ptr), Q, KF, p(keys.ptr)))
        if timed_phases is not None:
            ev[2].record()
        # the path's one exchange step: the minimum over ranks of Q packed keys (80 KB)
        if job.comm is not None:
            job.comm.allreduce_min_keys(keys)                     # RCCL on the library's stream: no host synchronisation
        elif job.collective == "torch":
            ctx.sync()                                            # keys are complete on the library's stream
            dist.all_reduce(keys_t, op=dist.ReduceOp.MIN, group=job.nccl_group)
            torch.cuda.current_stream().synchronize()             # reduced keys visible before the unpack kernel
        elif job.collective == "gloo":
            host = torch.from_numpy(keys.numpy().view(np.int64))
            dist.all_reduce(host, op=dist.ReduceOp.MIN)
            check(lib.gl_memcpy_h2d(ctx.handle, p(keys.ptr), host.numpy().ctypes.data_as(p), Q * 8))
        if lp_model is None:
            check(lib.gl_keys_unpack(ctx.handle, p(keys.ptr), Q, D, p(dist_dev.ptr), p(idx_dev.ptr)))
        else:
            check(lib.gl_keys_unpack_f32(ctx.handle, p(keys.ptr), Q, p(dist_dev.ptr), p(idx_dev.ptr)))
        check(lib.gl_memcpy_d2h(ctx.handle, out_dist.ctypes.data_as(p), p(dist_dev.ptr), Q * 4))
        check(lib.gl_memcpy_d2h(ctx.handle, out_idx.ctypes.data_as(p), p(idx_dev.ptr), Q * 8))
        if timed_phases is not None:
            ev[3].record()
            ctx.sync()
            timed_phases["generator_ms"] += ev[0].elapsed_ms_until(ev[1])
            timed_phases["distance_ms"] += ev[1].elapsed_ms_until(ev[2])
            timed_phases["reduce_unpack_d2h_ms"] += ev[2].elapsed_ms_until(ev[3])

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            ctx.sync()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    phases = {"generator_ms": 0.0, "distance_ms": 0.0, "reduce_unpack_d2h_ms": 0.0}
    ctx.prof_reset()
    ctx.prof_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(phases)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof = ctx.prof_read()
    ctx.prof_reset()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / steps
    value = Q / (elapsed / steps)

    # ---------------------------------------------------------------- per-kernel rooflines (this rank's launches)
    def kernel_entry(name, alg_per_step, bound, peak, unit, scale, executed_factor=1.0, note=None):
        ms, launches = prof[name]
        if launches == 0 or ms <= 0:
            return None
        per_launch = alg_per_step * steps / launches
        avg_ms = ms / launches
        achieved = per_launch / (avg_ms * 1e-3) / scale
        e = {"kernel": name, "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
             "frac": round(achieved / peak, 4), "traffic": None, "launches": int(launches), "avg_ms": round(avg_ms, 4),
             "alg_per_launch": per_launch}
        if executed_factor != 1.0:
            # the split-fp16 kernel issues 3 fp16 MFMAs per algorithmic product: matrix-pipe occupancy = 3 x the algorithmic rate
            e["executed_mfma"] = round(achieved * executed_factor, 2)
            e["executed_frac"] = round(achieved * executed_factor / peak, 4)
            if executed_factor == 3.0:
                # the arithmetic is fp32-class (22-bit operands, fp32 accumulation): against the fp32 matrix path it replaces
                e["frac_of_f32_mfma_peak"] = round(achieved / PEAK_F32_MFMA_TFLOPS, 3)
        if note:
            e["note"] = note
        return e

    split = gen_precision == 1
    conv_peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    kernels = [
        # all five ConvTranspose layers run in gather_conv (the 3-channel tail as a 48-column scatter-form GEMM)
        kernel_entry("gather_conv", n_loc * (F_GATHER_PER_IMG + F_RGB_PER_IMG) + (0 if lp_model is None else (n_loc + Q) * F_VGG_PER_IMG), "mfma",
                     conv_peak, "TFLOP/s", 1e12, 3.0 if split else 1.0,
                     "split-fp16: x = hi + lo, 3 fp16 MFMAs per product; achieved counts each product once, peak is the fp16 dense peak" if split else None),
        kernel_entry("l2_knn", 2.0 * Q * n_loc * D, "mfma", PEAK_I8_MFMA_TOPS, "TOP/s", 1e12),
        # algorithmic length 512 000 (K_lpips + D); lattice search rows are exactly that long, split rows issue 3 MFMA products per algorithmic one
        kernel_entry("feat_knn", 0 if lp_model is None else 2.0 * Q * n_loc * KF_ALG, "mfma", PEAK_F16_MFMA_TFLOPS, "TFLOP/s", 1e12,
                     1.0 if lp_model is None else (KF / KF_ALG if h1 else 3.0),
                     None if lp_model is None else ("fp16 lattice search rows (8-bit images): one MFMA per product, the 12 288 image values exact in fp16" if h1
                                                    else "split-fp16 contraction: 3 fp16 MFMAs per product")),
        # col2im + tanh + quantise: reads P [1024][48] fp32, writes 12288 codes per image
        kernel_entry("convt_rgb", n_loc * (1024 * 48 * 4 + 12288.0), "hbm", PEAK_HBM_GBS, "GB/s", 1e9),
        kernel_entry("l2_prepare", 0 if lp_model is not None else 2.0 * (n_loc + Q) * D, "hbm", PEAK_HBM_GBS, "GB/s", 1e9),
    ]
    kernels = [k for k in kernels if k and k["alg_per_launch"] > 0]
    # HBM-side traffic per launch from the PMC counters: cannot be collected from inside this process; taken from the committed
    # rocprofv3 --pmc passes of this exact workload (tools/pmc_summary.py says how), null for any other workload
    name = ("pmc_traffic_default.json" if split else "pmc_traffic_fp32.json") if lp_model is None else "pmc_traffic_l2lpips.json"
    traffic_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))     # the latest round's passes
    if traffic_files and world == 1 and Q == 10000 and N == 100000 and (lp_model is None or (split and h1)):
        with open(traffic_files[-1]) as f:
            tr = json.load(f)
        for k in kernels:
            if k["kernel"] in tr and "hbm_bytes_per_launch" in tr[k["kernel"]]:
                k["traffic"] = round(tr[k["kernel"]]["hbm_bytes_per_launch"])
                k["traffic_source"] = os.path.relpath(traffic_files[-1], ROOT)
    dominant = max(kernels, key=lambda k: k["avg_ms"] * k["launches"])
    if lp_model is not None:
        dominant = [k for k in kernels if k["kernel"] == "feat_knn"][0]      # the pairwise contraction is this mode's own kernel
    roofline = {k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    for k in ("executed_mfma", "executed_frac", "frac_of_f32_mfma_peak", "note", "traffic_source"):
        if k in dominant:
            roofline[k] = dominant[k]
    roofline["kernel"] = dominant["kernel"]
    roofline["avg_launch_ms"] = dominant["avg_ms"]
    roofline["launches_per_step"] = dominant["launches"] / steps

    # split-fp16 stores clamp at the fp16 range and count it; the library's Python callers redo such a pass with fp32 products, this file
    # calls the C ABI directly: a timed step that clamped anything is not a measurement of the stated arithmetic
    n_sat = ctx.h3_saturations()
    if world > 1:
        t = torch.tensor([n_sat], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)        # every rank raises, or none does
        n_sat = int(t.item())
    if n_sat:
        raise RuntimeError("bench: %d split-fp16 stores saturated during the timed steps; run with --gen-precision 0" % n_sat)

    # ---------------------------------------------------------------- parity check against the oracle (untimed)
    parity = None
    auroc = None
    if args.check_queries > 0 and lp_model is not None and world == 1:
        # l2-lpips: fp64 oracle on a small sub-problem (8 queries x the first 256 bank samples)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import lpips_oracle
        import oracle as np_oracle
        sub_bank = bank_u8.numpy()[:256].reshape(256, 3, 64, 64)
        sel = np.linspace(0, Q - 1, 8).astype(np.int64)
        od, oi, _ = lpips_oracle.knn_l2_lpips(vgg_sd, [lin["lin%d" % i] for i in range(5)], np_oracle.dequantize_u8(sub_bank),
                                              np_oracle.dequantize_u8(queries_u8[sel]), 64)
        gd, gi = gl.attack(queries_u8[sel], sub_bank, distance="l2-lpips", batch_size=64, lpips=lp_model)
        parity = {"queries_checked": 8, "bank_checked": 256, "idx_equal": bool(np.array_equal(gi, oi)),
                  "max_abs_dist_err": float(np.abs(gd.astype(np.float64) - od).max())}
        # the full-size census of this workload (tools/census_l2lpips_full.py, minutes of GPU time: not rerun here): default path against the
        # most exact device path over all 10^4 x 10^5 pairs, and the nearest / second-nearest gaps
        census = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "r*_census_l2lpips_config2.json")))
        if census and Q == 10000 and N == 100000:
            with open(census[-1]) as f:
                c = json.load(f)
            parity["full_size_census"] = {"source": os.path.relpath(census[-1], ROOT), "idx_mismatches": c["idx_mismatches"], "of_queries": c["config"]["queries"],
                                          "dist_max_abs_diff": c["dist_max_abs_diff"], "auroc_abs_delta": c["auroc_abs_delta"],
                                          "runner_up_gap_below_1e-5": c.get("runner_up_gap", {}).get("below_1e-5")}
    elif args.check_queries > 0 and lp_model is None:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        nchk = min(args.check_queries, Q)
        sel = np.linspace(0, Q - 1, nchk).astype(np.int64)
        _, oi, os_ = c_oracle.knn_l2_u8(bank_u8.numpy(), queries_u8[sel].reshape(nchk, D), 1)
        okeys = (os_.astype(np.uint64) << np.uint64(32)) | (oi + lo).astype(np.uint64)
        if world > 1:
            tk = torch.from_numpy(okeys.view(np.int64).copy())
            dist.all_reduce(tk, op=dist.ReduceOp.MIN)
            okeys = tk.numpy().view(np.uint64)
        o_idx = (okeys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        o_dist = ((okeys >> np.uint64(32)).astype(np.float64) * (4.0 / (65025.0 * D))).astype(np.float32)
        parity = {"queries_checked": int(nchk), "idx_equal": bool(np.array_equal(o_idx, out_idx[sel])),
                  "max_abs_dist_err": float(np.abs(o_dist.astype(np.float64) - out_dist[sel]).max())}
    if rank == 0:
        _, _, _, auc, ap_, prec = plot_roc(-out_dist[:n_pos].astype(np.float64), -out_dist[n_pos:].astype(np.float64))
        auroc = {"auc": auc, "ap": ap_, "precision_at_-0.14": prec}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    cpu = None
    cpu_fo = None
    if rank == 0 and world == 1 and cpu_leg and args.cpu_queries > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_port
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(args.cpu_threads or min(16, usable))     # the 1-GPU box's CPU share is 16 cores
        host_bank = bank_u8.numpy().reshape(n_loc, 3, 64, 64)
        if lp_model is None:
            nq_cpu = min(args.cpu_queries, Q)
            sel = np.linspace(0, Q - 1, nq_cpu).astype(np.int64)
            t_prep = time.time()
            bank_f = torch_port.dequantize(host_bank)                 # what fbb.main builds (fbb.py:134-135)
            q_f = torch_port.dequantize(queries_u8[sel])
            log("[cpu] bank dequantised to fp32 in %.1fs; timing %d queries x %d samples on %d threads" %
                (time.time() - t_prep, nq_cpu, n_loc, torch.get_num_threads()))
            torch_port.custom_knn(bank_f[:6400], q_f[0], torch_port.l2_loss, B)      # warm-up
            tc = time.perf_counter()
            cd, ci = [], []
            for k in range(nq_cpu):
                d_, i_ = torch_port.custom_knn(bank_f, q_f[k], torch_port.l2_loss, B)
                cd.append(d_)
                ci.append(i_)
                if k + 1 >= 8 and time.perf_counter() - tc > args.cpu_seconds:      # bounded sample: host speed varies a lot between boxes
                    break
            cpu_s = time.perf_counter() - tc
            nq_cpu = len(cd)
            sel = sel[:nq_cpu]
            cpu = {"value": round(nq_cpu / cpu_s, 4), "unit": "query-images/s", "cores": int(torch.get_num_threads()), "kind": "port",
                   "sample": "%d of the %d queries x the full %d-sample bank, BATCH_SIZE %d, PyTorch-CPU restatement of fbb.custom_knn "
                             "(oracle/torch_port.py); bank search only, the CPU does not run the generator" % (nq_cpu, Q, n_loc, B),
                   "seconds": round(cpu_s, 2),
                   "idx_equal_gpu": bool(np.array_equal(np.array(ci), out_idx[sel])),
                   "max_abs_dist_diff_vs_gpu": float(np.abs(np.array(cd) - out_dist[sel]).max())}
            del bank_f
        else:
            # the reference's literal loop (fbb.py:77-81 with Loss('l2-lpips')): one VGG16 + LPIPS evaluation of 64 bank images + the query per
            # (query, batch); 10^4 x 1 562 of them make the job.  SURVEY 8(d): time >= 3 batches of 64 pairs after a warm-up, scale by Q * N_eff.
            loss = torch_port.make_l2_lpips_loss(vgg_sd, [lin["lin%d" % i] for i in range(5)])
            bank_f = torch_port.dequantize(host_bank[:64 * 8])
            q_f = torch_port.dequantize(queries_u8[:1])
            loss(bank_f[:64], q_f)                                    # warm-up
            tc = time.perf_counter()
            nb, vals = 0, []
            while nb < 8 and (nb < 3 or time.perf_counter() - tc < args.cpu_seconds):
                vals.append(loss(bank_f[64 * nb:64 * (nb + 1)], q_f).detach().numpy())
                nb += 1
            cpu_s = time.perf_counter() - tc
            pairs_s = 64 * nb / cpu_s
            from ganleaks_amd.attack_models.utils import Loss
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):              # Loss announces its distance on stdout like the reference (utils.py:166)
                dev_loss = Loss("l2-lpips", lpips=lp_model)
            dv = np.concatenate([np.asarray(dev_loss(host_bank[64 * b:64 * (b + 1)], queries_u8[:1])) for b in range(nb)])
            cpu = {"value": round(pairs_s / n_eff, 8), "unit": "query-images/s", "cores": int(torch.get_num_threads()), "kind": "port",
                   "sample": "%d batches of 64 (bank image, query) pairs under 0.2*LPIPS+L2, autograd on as in the reference, PyTorch-CPU restatement of "
                             "Loss('l2-lpips').forward + PNetLin (oracle/torch_port.py); scaled by the %d pairs one query costs the reference "
                             "(VGG16 on 65 images per batch of 64)" % (nb, n_eff),
                   "pairs_per_s": round(pairs_s, 2), "seconds": round(cpu_s, 2),
                   "max_abs_loss_diff_vs_gpu": float(np.abs(np.concatenate(vals) - dv).max())}
            # "also report the features-once CPU variant so the algorithmic and the hardware speed-ups are separable":
            # the device path's algorithm on the host cores -- VGG16 once per image, the [Q, N] search as one GEMM over the V rows.
            rows_fn, search_fn = torch_port.make_features_once(vgg_sd, [lin["lin%d" % i] for i in range(5)])
            n_img, n_bq, n_bn = 256, 64, 1024
            img_f = torch_port.dequantize(host_bank[:n_img])
            rows_fn(img_f[:32])                                        # warm-up
            tc = time.perf_counter()
            brow = rows_fn(img_f)
            t_rows = time.perf_counter() - tc
            brow = torch.cat([brow] * (n_bn // n_img))                 # the GEMM sample: 64 query rows x 1024 bank rows x K = 512 000
            qrow = rows_fn(torch_port.dequantize(queries_u8[:n_bq]))
            search_fn(qrow[:8], brow[:64])
            tc = time.perf_counter()
            fo_d, fo_i = search_fn(qrow, brow)
            t_gemm = time.perf_counter() - tc
            img_s = n_img / t_rows
            pair_s = n_bq * n_bn / t_gemm
            t_job = (n_eff + Q) / img_s + Q * float(n_eff) / pair_s
            gd_, gi_ = gl.attack(queries_u8[:n_bq], host_bank[:n_img], distance="l2-lpips", batch_size=64, lpips=lp_model)
            cpu_fo = {"value": round(Q / t_job, 4), "unit": "query-images/s", "cores": int(torch.get_num_threads()), "kind": "port",
                      "sample": "features-once on the host: VGG16 + tap rows of %d images (%.1f images/s), then %d x %d rows of K = %d as one fp32 GEMM + min "
                                "(%.3g pairs/s); scaled to (%d + %d) images + %d x %d pairs" % (n_img, img_s, n_bq, n_bn, int(brow.shape[1]), pair_s, n_eff, Q, Q, n_eff),
                      "seconds": round(t_rows + t_gemm, 2), "projected_job_seconds": round(t_job, 1),
                      "idx_equal_gpu": bool(np.array_equal(fo_i.numpy() % n_img, gi_)),
                      "max_abs_dist_diff_vs_gpu": float(np.abs(fo_d.numpy() - gd_).max())}
            del brow, qrow

    if rank != 0:
        return None
    line = {
        "metric": "attack query-images/sec (10k queries x 100k samples) + AUROC delta vs ref",
        "value": round(value, 2), "unit": "query-images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": (("split-f16 (hi+lo, f32 accumulate; generator)" if split else "f32 (generator)") + " + i8->i32 exact (distance)") if lp_model is None
        else ("split-f16" if split else "f32") + " (generator, VGG16) + " + ("f16 search rows" if args.feat_rows == "fp16" else "split-f16") + " (LPIPS contraction)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: DCGAN-64 generator -> 8-bit bank, L2 1-NN (fbb)" if lp_model is None else
                   "BASELINE configs[2]: DCGAN/WGAN-GP-64 generator -> 8-bit bank, 0.2*LPIPS+L2 1-NN (fbb default distance)",
                   "queries": Q, "bank": N,
                   "bank_used": n_eff, "batch_size": B, "image": "3x64x64", "parallelism": "bank sharded x%d (%s), queries replicated, "
                   "%sall-reduce(min) of %d packed keys (%s)" % (world, shard_note, "query features sharded + all-gathered, " if (lp_model is not None and q_shard) else "", Q, {"native": "RCCL through the C ABI on the library's stream",
                                                                                    "torch": "torch.distributed nccl", "gloo": "gloo through host memory"}.get(job.collective, ""))
                   if world > 1 else "single GPU",
                   "shard_rows": [int(bounds[r + 1] - bounds[r]) for r in range(world)]},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "kernels": kernels,
        "phases_ms_per_step_rank0": {k: round(v / steps, 3) for k, v in phases.items()},
        "parity": parity,
        "auroc": auroc,
        "speedup_vs_cpu_baseline": round(value / cpu["value"], 1) if cpu else None,
    }
    if cpu_fo is not None:
        # algorithm (features once per image instead of once per pair) and hardware, separated: literal -> features-once on the same cores -> this GPU
        line["cpu_baseline_features_once"] = cpu_fo
        line["speedup_vs_cpu_features_once"] = round(value / cpu_fo["value"], 1)
        line["algorithmic_speedup_on_cpu"] = round(cpu_fo["value"] / cpu["value"], 1) if cpu else None
    return line


def live_traffic(args, distance, gen_precision):
    """HBM-side bytes per launch of every kernel family of ONE step of this workload, measured now: two child runs of this file under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes, never together with a trace, as MI355X_MICROARCH.md prescribes), summed per
    family by tools/pmc_summary.py: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (both counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide
    coalesced read stream).  Returns {family: bytes per launch} or None when the profiler is not available / fails (the caller then falls back to
    the committed passes)."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3") or live_traffic.failed:
        return None
    tmp = tempfile.mkdtemp(prefix="gl_traffic_", dir="/tmp")
    try:
        cmd = ["python3", os.path.join(ROOT, "bench.py"), "--secondary", "off", "--steps", "1", "--warmup", "0", "--cpu-queries", "0", "--check-queries", "0",
               "--live-traffic", "off", "--distance", distance, "--gen-precision", str(gen_precision), "--queries", str(args.queries), "--bank", str(args.bank)]
        env = dict(os.environ, TMPDIR="/tmp")
        dirs = []
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter.lower())
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=180)
            if r.returncode != 0:
                live_traffic.failed = True               # one failure (or a pass that hangs: timeout) ends the live measurement for this run
                return None
            dirs.append(d)
        out = os.path.join(tmp, "summary.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), out] + dirs, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        if r.returncode != 0:
            return None
        with open(out) as f:
            tr = json.load(f)
        return {k: v["hbm_bytes_per_launch"] for k, v in tr.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
    except Exception as e:  # noqa: BLE001  (a measurement aid: never fails the bench)
        log("[traffic] live PMC passes failed: %s" % (e,))
        live_traffic.failed = True
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


live_traffic.failed = False


def apply_live_traffic(line, tr):
    """put the measured bytes per launch into a bench line's kernel entries and its roofline block"""
    if not tr:
        return False
    hit = False
    for k in line.get("kernels", []):
        fam = k["kernel"] if k["kernel"] in tr else k["kernel"] + "_f32"       # the fp32-MFMA convolution kernel is its own family in the summary
        if fam in tr:
            k["traffic"] = round(tr[fam])
            k["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (tools/pmc_summary.py)"
            if k["kernel"] == line["roofline"].get("kernel"):
                line["roofline"]["traffic"] = k["traffic"]
                line["roofline"]["traffic_source"] = k["traffic_source"]
                hit = True
    return hit


def measure_config0(job, args):
    """BASELINE configs[0]: DCGAN-64, 256 queries x 1 000 samples, L2 -- the reference's own CPU-runnable case (plumbing).  CPU: the full run of
    the torch restatement of fbb.custom_knn, wall clock, median of 3.  GPU: the same problem through attack() (bank
    generated on the device, queries uploaded from the host), median of 5; indices and distances must agree."""
    import torch
    gl, ctx = job.gl, job.ctx
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_port
    synth = gl.synth
    Q0, N0, B = 256, 1000, args.batch_size
    gen = Generator(100, 3, 64, ctx)
    gen.load_state_dict(synth.dcgan_state_dict(1234))
    z = synth.latent(1, N0)
    pos = synth.perturb_u8(5, gen.generate_u8(synth.latent(2, Q0 // 2)).numpy(), 0.05 * 127.5)
    neg = synth.perturb_u8(6, gen.generate_u8(synth.latent(3, Q0 - Q0 // 2)).numpy(), 0.10 * 127.5)
    q = np.concatenate([pos, neg])
    gpu_t = []
    for _ in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        gd, gi = gl.attack(q, gl.GeneratedBank(gen, z), distance="l2", batch_size=B)
        gpu_t.append(time.perf_counter() - t0)
    gpu_s = float(np.median(gpu_t[1:]))
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(args.cpu_threads or min(16, usable))
    bank_f = torch_port.dequantize(gen.generate_u8(z).numpy())
    q_f = torch_port.dequantize(q)
    cpu_t = []
    for _ in range(3):
        t0 = time.perf_counter()
        cd, ci = zip(*[torch_port.custom_knn(bank_f, q_f[k], torch_port.l2_loss, B) for k in range(Q0)])
        cpu_t.append(time.perf_counter() - t0)
    cpu_s = float(np.median(cpu_t))
    return {"workload": "BASELINE configs[0]: DCGAN-64, %d queries x %d samples (%d used), L2" % (Q0, N0, (N0 // B) * B),
            "value": round(Q0 / gpu_s, 1), "unit": "query-images/s", "ms": round(1e3 * gpu_s, 3),
            "note": "whole attack() call incl. the generator, query upload, kernels and the read-back: launch-latency-bound at this size",
            "cpu_baseline": {"value": round(Q0 / cpu_s, 2), "unit": "query-images/s", "cores": int(torch.get_num_threads()), "kind": "port",
                             "sample": "full run, wall clock, median of 3 (bank search only: the CPU does not run the generator)", "seconds": round(cpu_s, 3)},
            "parity": {"idx_equal": bool(np.array_equal(np.array(ci), gi)), "max_abs_dist_err": float(np.abs(np.array(cd) - gd).max())},
            "speedup_vs_cpu_baseline": round(cpu_s / gpu_s, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--bank", type=int, default=100000)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=0, help="generator images per pass (0 = library default)")
    ap.add_argument("--cpu-queries", type=int, default=128, help="queries timed for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads for the baseline (0 = min(16, usable cores))")
    ap.add_argument("--check-queries", type=int, default=64, help="queries checked against the oracle after the run")
    ap.add_argument("--gen-precision", type=int, default=1, choices=[0, 1],
                    help="generator arithmetic: 1 = split-fp16 (three fp16 MFMAs per product, default), 0 = fp32 MFMA")
    ap.add_argument("--collective", default="native", choices=["native", "torch", "gloo"],
                    help="N > 1, the key reduction: native = RCCL through the C ABI (gl_allreduce_min_keys, default; falls back to torch if the "
                         "communicator cannot be formed), torch = torch.distributed's nccl backend, gloo = through host memory (rehearsal: ranks may share a GPU)")
    ap.add_argument("--backend", default=None, help="deprecated alias: nccl -> --collective torch, gloo -> --collective gloo")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: equal bank shards instead of shards sized by each rank's measured generator speed")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="stop the CPU baseline after this many seconds (at least 8 queries are timed)")
    ap.add_argument("--feat-rows", default="fp16", choices=["fp16", "split"],
                    help="l2-lpips only: rows of the nearest-neighbour search: fp16 = one half per LPIPS value (gl_feat_knn_h1, default), "
                         "split = hi + lo halves of everything (gl_feat_knn)")
    ap.add_argument("--distance", default="l2", choices=["l2", "l2-lpips"],
                    help="l2 = BASELINE configs[1] (default, the headline); l2-lpips = configs[2] (0.2*LPIPS+L2; needs ~2 MB of HBM per image)")
    ap.add_argument("--secondary", default="auto", choices=["auto", "on", "off"],
                    help="the two further measurements (configs[2] under 0.2*LPIPS+L2, and the fp32-MFMA generator): auto = with the default one-GPU "
                         "headline workload only")
    ap.add_argument("--secondary-steps", type=int, default=2)
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic from rocprofv3 --pmc passes of child runs of this workload made now (auto: in the default one-GPU run only); "
                         "otherwise, and when the profiler is unavailable, from the newest committed passes under profiles/")
    args = ap.parse_args()
    if args.backend == "nccl":
        args.collective = "torch"
    elif args.backend == "gloo":
        args.collective = "gloo"

    job = setup_job(args)
    line = measure(job, args, args.distance, args.gen_precision, args.steps, args.warmup, cpu_leg=True, headline=True)
    default_run = job.world == 1 and args.distance == "l2" and args.gen_precision == 1 and (
        args.secondary == "on" or (args.secondary == "auto" and args.queries == 10000 and args.bank == 100000))
    if default_run:
        import gc
        gc.collect()
        sec = measure(job, args, "l2-lpips", 1, args.secondary_steps, 1, cpu_leg=True, headline=False)
        gc.collect()
        f32 = measure(job, args, "l2", 0, 3, 1, cpu_leg=False, headline=False)
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "kernels", "phases_ms_per_step_rank0",
                "parity", "auroc", "speedup_vs_cpu_baseline")
        line["secondary"] = {k: sec[k] for k in keep + ("cpu_baseline_features_once", "speedup_vs_cpu_features_once", "algorithmic_speedup_on_cpu") if k in sec}
        line["secondary_fp32"] = {k: f32[k] for k in keep if k != "cpu_baseline" and k != "speedup_vs_cpu_baseline"}
        line["config0"] = measure_config0(job, args)
    live = job.world == 1 and (args.live_traffic == "on" or (args.live_traffic == "auto" and default_run))
    if live:
        import gc
        gc.collect()
        job.ctx.trim()                                           # the children need the memory this process has released
        t0 = time.time()
        ok = apply_live_traffic(line, live_traffic(args, args.distance, args.gen_precision))
        if default_run:
            ok = apply_live_traffic(line["secondary"], live_traffic(args, "l2-lpips", 1)) and ok
            apply_live_traffic(line["secondary_fp32"], live_traffic(args, "l2", 0))
        log("[traffic] live PMC passes %s in %.0fs" % ("ok" if ok else "unavailable: committed passes kept", time.time() - t0))
    if job.rank == 0:
        print(json.dumps(line), flush=True)
    if job.world > 1:
        job.dist.barrier()
        if job.comm is not None:
            job.comm.destroy()
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
