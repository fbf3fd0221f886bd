"""Sharding of the sample bank across GPUs (one process per GPU) and the single exchange step of the
path: a min-reduce of the packed (distance, index) keys.  The reference is single
device (attack_models/fbb.py:40); this is the [build] multi-GPU extension named by the project brief.

  * the BATCH_SIZE truncation (attack_models/fbb.py:77) is applied to the GLOBAL bank length first,
    then [0, n_eff) is cut into `world` contiguous ranges; rank r owns [bounds[r], bounds[r+1]).
  * every rank holds all queries and produces keys[q] = (S << 32) | global_index for its range.
  * all-reduce(MIN) of the keys: smallest distance, then smallest global index -- bit-identical to the single-GPU result for
    any world size.  Q x 8 bytes (80 KB at Q = 10^4): latency-bound, one RCCL call over xGMI.  The product route is the C ABI's
    own collective (`gl_comm_*`, `gl_allreduce_min_keys`: ncclAllReduce(ncclMin, ncclUint64) on the context's stream); the
    torch.distributed route (keys viewed as int64 -- they stay below 2^63, so the signed order equals the unsigned one) remains for
    `gloo` on CPU tensors in the CPU tests and as a fallback.
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


class HostMerge:
    """min-merge of per-rank key arrays between the threads of one process (the fallback of attack_on_devices when RCCL cannot form a
    communicator).  merge(rank, keys) blocks until every rank has called it and returns the element-wise minimum; it may be called
    any number of times (once per query slice of a streamed attack): a second rendezvous keeps a fast rank's next deposit from
    overwriting what a slow rank is still merging."""

    def __init__(self, world):
        import threading
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._deposits = [None] * self.world

    def merge(self, rank, keys_host):
        self._deposits[rank] = np.asarray(keys_host, np.uint64)
        self._barrier.wait()                                 # every rank has deposited
        merged = merge_keys_host(self._deposits)
        self._barrier.wait()                                 # every rank has merged: the slots may be reused
        return merged

    def abort(self):
        self._barrier.abort()


def make_comm(ctx, group=None):
    """the native RCCL communicator of a `torch.distributed`-launched job (one process per GPU): rank 0 draws the unique id, the process
    group that the launcher set up carries its 128 bytes to the other ranks (the only thing torch is used for), every rank joins with its
    Context.  Returns None for a world of one.  The data-path collective is then `comm.allreduce_min_keys(keys)` -- queued on the
    context's stream by libganleaks_hip.so itself (gl_allreduce_min_keys), no torch tensor involved."""
    import torch.distributed as dist
    from ._lib import Comm
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return Comm(ctx, box[0], rank, world)


def allreduce_min_keys(keys, group=None, comm=None, _even_alone=False):
    """in-place MIN all-reduce of a keys DeviceArray (uint64 [Q]).

    comm (a `_lib.Comm`, see make_comm): RCCL through the C ABI on the context's stream; asynchronous, returns the same array.
    otherwise over torch.distributed: backend nccl (= RCCL on ROCm) aliases the device buffer as an int64 torch tensor through
    __cuda_array_interface__; backend gloo stages through host memory (CPU tests / rehearsal)."""
    if comm is not None:
        return comm.allreduce_min_keys(keys)
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _even_alone):
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
    latents; the packed keys are min-reduced by RCCL between the contexts (`gl_comm_init_all` + `gl_allreduce_min_keys`, each on its own
    stream).  When RCCL cannot form the communicator -- it refuses two ranks on one device, which is how the single-GPU tests drive
    this -- the keys are merged on the host instead (Q x 8 bytes per device).  The alternative to one process per GPU for callers
    without a launcher.

    make_generator(ctx) -> a generator bound to that context with its weights loaded (e.g. dcgan.Generator(100, 3, 64, ctx) + load_state_dict)
    make_lpips(ctx)     -> an LpipsModel for 'l2-lpips'
    devices             -> list of device ordinals (default: all visible); weights -> relative speeds for `weighted_bounds`
    returns (dist float32 [Q], idx int64 [Q]), identical to the single-device result."""
    import threading
    from ._lib import Comm, Context, GanLeaksError, GL_ERR_RCCL, device_count
    from .attack import GeneratedBank, attack
    devices = list(range(device_count())) if devices is None else list(devices)
    if not devices:
        raise ValueError("no devices")
    world = len(devices)
    n_eff = (len(z) // int(batch_size)) * int(batch_size)
    if n_eff == 0:
        raise ValueError("bank holds no full batch of %d samples (attack_models/fbb.py:77-83)" % int(batch_size))
    bounds = weighted_bounds(n_eff, weights, int(batch_size)) if weights is not None else [n_eff * r // world for r in range(world + 1)]
    contexts = [Context(d) for d in devices]
    comms = None
    if world > 1:
        try:
            comms = Comm.init_all(contexts)
        except GanLeaksError as e:
            if e.code != GL_ERR_RCCL:
                raise
    host = HostMerge(world)
    results, errors = [None] * world, []

    def reduce_fn_for(rank, ctx):
        if comms is not None:
            return comms[rank].allreduce_min_keys
        if world == 1:
            return None
        return lambda keys: ctx.to_device(host.merge(rank, keys.numpy()))

    def work(rank):
        gen = bank = model = None
        try:
            ctx = contexts[rank]
            gen = make_generator(ctx)
            lo, hi = bounds[rank], bounds[rank + 1]
            bank = GeneratedBank(gen, z[lo:hi], index_base=lo, **generate_kwargs)
            model = make_lpips(ctx) if (distance == "l2-lpips" and make_lpips is not None) else None
            results[rank] = attack(queries, bank, distance=distance, batch_size=batch_size, ctx=ctx, reduce_fn=reduce_fn_for(rank, ctx), lpips=model)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            host.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in comms or []:
        c.destroy()
    if errors:
        raise errors[0]             # (the traceback keeps the worker's objects alive: their contexts are left to the process)
    import gc
    gc.collect()                    # generators, banks and models of the workers are gone: their contexts can go too
    for c in contexts:
        c.destroy()
    return results[0]
