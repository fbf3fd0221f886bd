"""Sharding of the sample bank across GPUs (one process per GPU) and the single exchange step of the
path: a min-reduce of the packed (distance, index) keys.  The reference is single
device (attack_models/fbb.py:40); this is the [build] multi-GPU extension named by the project brief.

  * the BATCH_SIZE truncation (attack_models/fbb.py:77) is applied to the GLOBAL bank length first,
    then [0, n_eff) is cut into `world` contiguous ranges; rank r owns [bounds[r], bounds[r+1]).
  * every rank holds all queries and produces keys[q] = (S << 32) | global_index for its range.
  * all_reduce(MIN) on the keys viewed as int64 (S < 2^31, so keys are non-negative and the signed
    order equals the unsigned one): smallest distance, then smallest global index -- bit-identical to
    the single-GPU result for any world size.  Q x 8 bytes (80 KB at Q = 10^4): latency-bound, one
    RCCL call over xGMI; `gloo` on CPU tensors in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_total, batch_size, world):
    """global truncation, then `world` near-equal contiguous ranges. returns list of world+1 ints."""
    n_eff = (int(n_total) // int(batch_size)) * int(batch_size)
    return [n_eff * r // world for r in range(world + 1)]


def weighted_bounds(n_eff, weights, multiple=64):
    """contiguous ranges of [0, n_eff) with sizes proportional to `weights` (e.g. measured rows/s of every rank: the GPUs of one node
    differ by up to ~12 % under matrix-core load, and the slowest rank sets the step time), interior boundaries rounded to `multiple`.
    The result of the attack does not depend on the split.  returns list of len(weights)+1 ints."""
    w = np.asarray(weights, np.float64)
    if w.ndim != 1 or len(w) == 0 or not np.all(np.isfinite(w)) or np.any(w <= 0):
        raise ValueError("weights must be positive finite numbers")
    cum = np.concatenate([[0.0], np.cumsum(w)]) / w.sum()
    b = [int(round(c * n_eff / multiple)) * multiple for c in cum]
    b[0], b[-1] = 0, int(n_eff)
    for r in range(1, len(b)):                        # monotone, inside the range
        b[r] = min(max(b[r], b[r - 1]), int(n_eff))
    return b


def merge_keys_host(key_arrays):
    """reference semantics of the reduce, on host arrays (used by tests and the gloo path)."""
    out = np.asarray(key_arrays[0], np.uint64).copy()
    for k in key_arrays[1:]:
        np.minimum(out, np.asarray(k, np.uint64), out=out)
    return out


def allreduce_min_keys(keys, group=None):
    """in-place MIN all-reduce of a keys DeviceArray (uint64 [Q]) over torch.distributed.

    backend nccl (= RCCL on ROCm): the device buffer is aliased as an int64 torch tensor through
    __cuda_array_interface__ and reduced in place over xGMI.
    backend gloo: staged through host memory (CPU tests / rehearsal)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return keys
    ctx = keys.ctx
    if dist.get_backend(group) == "nccl":
        ctx.sync()                                           # keys were produced on the context stream
        t = torch.as_tensor(keys.view(keys.shape, np.int64), device="cuda:%d" % ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        torch.cuda.current_stream(t.device).synchronize()
        return keys
    host = torch.from_numpy(keys.numpy().view(np.int64))
    dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
    return ctx.to_device(host.numpy().view(np.uint64))


def allreduce_min_keys_host(keys_host, group=None):
    """same reduce for a host uint64 array (pure-CPU rehearsal of the merge with gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(keys_host).view(np.int64).copy())
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return t.numpy().view(np.uint64)


def attack_on_devices(queries, make_generator, z, devices=None, distance="l2", batch_size=64, make_lpips=None, weights=None, **generate_kwargs):
    """The sharded attack inside ONE process: a host thread per GPU, each with its own context, generates and searches its range of the
    latents; the packed keys are min-merged on the host (Q x 8 bytes per device) -- no launcher and no RCCL.  The alternative to
    `torch.distributed` + `allreduce_min_keys` for callers that do not want one process per GPU.

    make_generator(ctx) -> a generator bound to that context with its weights loaded (e.g. dcgan.Generator(100, 3, 64, ctx) + load_state_dict)
    make_lpips(ctx)     -> an LpipsModel for 'l2-lpips'
    devices             -> list of device ordinals (default: all visible); weights -> relative speeds for `weighted_bounds`
    returns (dist float32 [Q], idx int64 [Q]), identical to the single-device result."""
    import threading
    from ._lib import Context, device_count
    from .attack import GeneratedBank, attack
    devices = list(range(device_count())) if devices is None else list(devices)
    if not devices:
        raise ValueError("no devices")
    world = len(devices)
    n_eff = (len(z) // int(batch_size)) * int(batch_size)
    if n_eff == 0:
        raise ValueError("bank holds no full batch of %d samples (attack_models/fbb.py:77-83)" % int(batch_size))
    bounds = weighted_bounds(n_eff, weights, int(batch_size)) if weights is not None else [n_eff * r // world for r in range(world + 1)]
    barrier = threading.Barrier(world)
    deposits, results, errors = [None] * world, [None] * world, []

    def reduce_fn_for(rank, ctx):
        def reduce_fn(keys):
            deposits[rank] = keys.numpy()
            barrier.wait()
            return ctx.to_device(merge_keys_host(deposits))
        return reduce_fn

    def work(rank):
        try:
            ctx = Context(devices[rank])
            gen = make_generator(ctx)
            lo, hi = bounds[rank], bounds[rank + 1]
            bank = GeneratedBank(gen, z[lo:hi], index_base=lo, **generate_kwargs)
            model = make_lpips(ctx) if (distance == "l2-lpips" and make_lpips is not None) else None
            results[rank] = attack(queries, bank, distance=distance, batch_size=batch_size, ctx=ctx, reduce_fn=reduce_fn_for(rank, ctx), lpips=model)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results[0]

