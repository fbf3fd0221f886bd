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
    context's stream by libganleaks_hip.so itself (gl_allreduce_min_keys), no torch tensor involved.

    Every rank leaves this function the same way: if rank 0 cannot draw the id (librccl missing, ncclGetUniqueId failing) it broadcasts
    the error text instead and ALL ranks raise GanLeaksError(GL_ERR_RCCL); if some rank cannot join, an all-reduce of a success flag
    makes the others drop their communicator and raise too -- so a caller that falls back to another route (bench.py --collective)
    does so on every rank, with the process group's collectives still matched."""
    import torch
    import torch.distributed as dist
    from ._lib import Comm, GanLeaksError, GL_ERR_RCCL
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        try:
            box = [("id", Comm.unique_id())]
        except Exception as e:  # noqa: BLE001  (whatever it is, the other ranks must hear of it)
            box = [("err", "rank 0: %s" % (e,))]
    dist.broadcast_object_list(box, src=0, group=group)
    tag, payload = box[0]
    if tag != "id":
        raise GanLeaksError(GL_ERR_RCCL, "no RCCL unique id: %s" % payload)
    comm, why = None, ""
    try:
        comm = Comm(ctx, payload, rank, world)
    except Exception as e:  # noqa: BLE001
        why = str(e)
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        ok = ok.to("cuda:%d" % ctx.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) != 1:
        if comm is not None:
            comm.abort()
        raise GanLeaksError(GL_ERR_RCCL, "the RCCL communicator could not be formed on every rank%s" % (": " + why if why else ""))
    return comm


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


class DeviceGroup:
    """The sharded attack inside ONE process: a context per GPU, a host thread per context and call, the packed keys min-reduced by RCCL
    between the contexts (`gl_comm_init_all` + `gl_allreduce_min_keys`, each on its own stream).  When RCCL cannot form the
    communicator -- it refuses two ranks on one device, which is how the single-GPU tests drive this -- the keys are merged on the host
    instead (Q x 8 bytes per device).  The alternative to one process per GPU for callers without a launcher.

    The group outlives a call: the LPIPS model of every context and the prepared (replicated) query rows are kept, so a sweep over many
    banks with the same queries (attack_models/fbb.py:114-123) featurises them once per device.

    Failure handling: everything that can fail without the other ranks (building the generator and the LPIPS model, uploading and
    featurising the queries) runs BEFORE a host rendezvous; if any rank failed there, no rank queues a collective.  A failure after it
    (e.g. out of memory on a bank chunk) aborts every communicator (gl_comm_abort = ncclCommAbort), which ends the reduce kernels the
    other ranks may already have queued, so the call raises instead of hanging the GPUs; the group is unusable afterwards."""

    def __init__(self, devices=None):
        from ._lib import Comm, Context, GanLeaksError, GL_ERR_RCCL, device_count
        devices = list(range(device_count())) if devices is None else [int(d) for d in devices]
        if not devices:
            raise ValueError("no devices")
        self.devices, self.world = devices, len(devices)
        self.contexts = [Context(d) for d in devices]
        self.comms = None
        if self.world > 1:
            try:
                self.comms = Comm.init_all(self.contexts)
            except GanLeaksError as e:
                if e.code != GL_ERR_RCCL:
                    raise
        self._models = [None] * self.world
        self._queries = [None] * self.world          # (key, prepared rows) per rank
        self._broken = False

    @property
    def collective(self):
        return "rccl" if self.comms is not None else ("host-merge" if self.world > 1 else "none")

    def attack(self, queries, make_generator=None, z=None, bank=None, distance="l2", batch_size=64, make_lpips=None, weights=None,
               **generate_kwargs):
        """see attack_on_devices"""
        import threading
        from ._lib import DeviceArray
        from .attack import GeneratedBank, attack, prepare_queries
        if self._broken:
            raise RuntimeError("this DeviceGroup failed in an earlier call; build a new one")
        if (bank is None) == (make_generator is None):
            raise ValueError("needs either make_generator + z or bank=")
        if getattr(queries, "kind", None) in ("feat", "u8", "int", "f32"):
            raise TypeError("queries must be host images: prepared rows live on one context, every context of the group prepares its own")
        if bank is None and z is None:
            raise ValueError("make_generator needs the latents z")
        world, contexts, comms = self.world, self.contexts, self.comms
        if isinstance(bank, DeviceArray):
            bank = bank.numpy()
        n_total = len(bank) if bank is not None else len(z)
        n_eff = (n_total // int(batch_size)) * int(batch_size)
        if n_eff == 0:
            raise ValueError("bank holds no full batch of %d samples (attack_models/fbb.py:77-83)" % int(batch_size))
        bounds = weighted_bounds(n_eff, weights, int(batch_size)) if weights is not None else [n_eff * r // world for r in range(world + 1)]
        host = HostMerge(world)
        ready = threading.Barrier(world)
        results, errors, lock = [None] * world, [], threading.Lock()
        qkey = (id(queries), distance, tuple(getattr(queries, "shape", ())))

        def reduce_fn_for(rank, ctx):
            if comms is not None:
                return comms[rank].allreduce_min_keys
            if world == 1:
                return None
            return lambda keys: ctx.to_device(host.merge(rank, keys.numpy()))

        def fail(e, after_setup):
            with lock:
                errors.append(e)
                ready.abort()
                host.abort()
                if after_setup:
                    self._broken = True
                    for c in comms or []:
                        c.abort()                             # idempotent; ends reduces that can no longer complete

        def work(rank):
            shard = None
            try:
                ctx = contexts[rank]
                lo, hi = bounds[rank], bounds[rank + 1]
                if bank is None:
                    shard = GeneratedBank(make_generator(ctx), z[lo:hi], index_base=lo, **generate_kwargs)
                else:
                    shard = bank[lo:hi]
                model = None
                if distance == "l2-lpips":
                    if self._models[rank] is None:
                        if make_lpips is not None:
                            self._models[rank] = make_lpips(ctx)
                        else:
                            from .lpips import model_for
                            self._models[rank] = model_for(ctx)
                    model = self._models[rank]
                if model is not None and not model._warm:
                    # the split path's calibration pass (first use of a model) is part of the fallible, collective-free setup
                    model.features(np.zeros((1, 3, 32, 32), np.uint8), role="query")
                    model._warm = True
                if comms is None and (self._queries[rank] is None or self._queries[rank][0] != qkey):
                    # no RCCL between the contexts: every context prepares all queries itself, which needs nobody else
                    self._queries[rank] = (qkey, prepare_queries(queries, distance, ctx, model), queries)   # (keeps `queries` alive: id() stays unique)
                ctx.sync()
            except BaseException as e:  # noqa: BLE001
                fail(e, False)
                return
            try:
                ready.wait()                                  # every rank is set up: from here on collectives may be queued
            except threading.BrokenBarrierError:
                return                                        # another rank failed in its setup; nothing was queued
            try:
                if self._queries[rank] is None or self._queries[rank][0] != qkey:
                    # with RCCL between the contexts the VGG16 features of the (replicated) queries are computed Q / world per rank and
                    # all-gathered: a collective, hence behind the rendezvous
                    self._queries[rank] = (qkey, prepare_queries(queries, distance, ctx, model, comm=comms[rank]), queries)
                prepared = self._queries[rank][1]
                results[rank] = attack(prepared, shard, distance=distance, batch_size=batch_size, ctx=ctx, reduce_fn=reduce_fn_for(rank, ctx),
                                       lpips=model, index_base=lo)
            except BaseException as e:  # noqa: BLE001
                fail(e, True)

        threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            first = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)]
            raise (first or errors)[0]
        return results[0]

    def close(self):
        """communicators, cached models and query rows, then the contexts (a failed group leaves its contexts to the process: the
        tracebacks keep the workers' objects alive)"""
        for c in self.comms or []:
            c.destroy()
        self.comms = None
        self._models = [None] * self.world
        self._queries = [None] * self.world
        if not self._broken:
            import gc
            gc.collect()
            for c in self.contexts:
                c.destroy()
        self.contexts = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def attack_on_devices(queries, make_generator=None, z=None, devices=None, distance="l2", batch_size=64, make_lpips=None, weights=None,
                      bank=None, **generate_kwargs):
    """One sharded attack on a DeviceGroup built for the call.  The bank is either generated on the devices or handed over:
      make_generator(ctx), z -> rank r generates rows [bounds[r], bounds[r+1]) from z[lo:hi] with a generator bound to its context
                                (e.g. dcgan.Generator(100, 3, 64, ctx) + load_state_dict); never materialised;
      bank                   -> a host array [N,C,H,W] (u8 codes or floats; numpy / CPU torch; a DeviceArray is read back once): what the
                                reference's fbb.main reads from image_*.png (attack_models/fbb.py:133-135).  Rank r uploads its rows only.
    make_lpips(ctx)     -> an LpipsModel for 'l2-lpips' (default: lpips.model_for(ctx): the registered factory, else the local weight files)
    devices             -> list of device ordinals (default: all visible); weights -> relative speeds for `weighted_bounds`
    returns (dist float32 [Q], idx int64 [Q]), identical to the single-device result."""
    with DeviceGroup(devices) as group:
        return group.attack(queries, make_generator, z, bank, distance, batch_size, make_lpips, weights, **generate_kwargs)
