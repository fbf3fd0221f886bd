"""GPU: the context's arena for large device blocks (gl_malloc / gl_free / gl_ctx_trim / gl_mem_info, include/ganleaks.h): a freed block of
>= 256 MiB is handed out again for the next request of that size instead of going back to the driver (the 153 GiB of query rows of a
256 x 256 attack cost seconds to allocate and free per call), is counted as available memory, and is released when asked or when one of the
library's own allocations would otherwise fail."""
import gc

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


def test_large_blocks_are_kept_and_reused():
    import ganleaks_amd as gl
    ctx = gl.Context(0)
    try:
        avail0, total = ctx.mem_info()
        assert 0 < avail0 <= total and total > 200 * (1 << 30)            # an MI355X
        a = ctx.empty((320 << 20,), np.uint8)
        ptr = a.ptr
        small = ctx.empty((1 << 20,), np.uint8)
        sptr = small.ptr
        del a, small
        gc.collect()
        avail1, _ = ctx.mem_info()
        assert abs(avail1 - avail0) < (64 << 20)                         # the kept block counts as available
        b = ctx.empty((300 << 20,), np.uint8)                            # within an eighth of the kept size: the same block
        assert b.ptr == ptr
        c = ctx.empty((320 << 20,), np.uint8)                            # the block is in use: a new one
        assert c.ptr != ptr
        d = ctx.empty((128 << 20,), np.uint8)                            # far smaller: never served from a 320 MiB block
        assert d.ptr != ptr
        del b, c, d
        gc.collect()
        ctx.trim()
        avail2, _ = ctx.mem_info()
        assert abs(avail2 - avail0) < (64 << 20)
        e = ctx.empty((320 << 20,), np.uint8)
        e.view((16,), np.uint8).numpy()                                   # usable
        del e
        gc.collect()
        assert sptr                                                       # (small blocks go straight back to the driver)
    finally:
        gc.collect()
        ctx.destroy()


def test_query_budget_follows_the_device():
    import ganleaks_amd as gl
    from ganleaks_amd.attack import _query_budget_bytes
    ctx = gl.Context.get()
    avail, _ = ctx.mem_info()
    b = _query_budget_bytes(64 << 30, ctx)
    assert b == max(avail - (64 << 30) - (12 << 30), min(64 << 30, avail // 4)) and b < avail
    assert _query_budget_bytes(1 << 30, ctx) == 1 << 30                   # an explicit chunk budget is also the query budget


def test_library_allocations_reclaim_the_arena_when_memory_runs_out(synth):
    """a kept block must never make one of the library's own allocations fail: gl_device_alloc releases the arena and retries.  Nearly all of the
    device is put into one kept block, then a generator needs its multi-GB workspace."""
    import ganleaks_amd as gl
    from ganleaks_amd.gan_models.dcgan.model_torch import Generator
    ctx = gl.Context(0)
    try:
        avail, _ = ctx.mem_info()
        hog = avail - (3 << 30)                                           # leave 3 GiB: less than the workspace of a 4096-image pass (~5 GiB)
        if hog < (64 << 30):
            pytest.skip("device too full for this test")
        a = ctx.empty((hog,), np.uint8)
        del a
        gc.collect()
        after, _ = ctx.mem_info()
        assert after >= avail - (256 << 20)                              # kept, and counted as available
        g = Generator(100, 3, 64, ctx)
        g.load_state_dict(synth.dcgan_state_dict(1234))
        z = synth.latent(3, 4096)
        u8 = g.generate_u8(z)                                            # needs the workspace: the arena block has to go
        ref = Generator(100, 3, 64)
        ref.load_state_dict(synth.dcgan_state_dict(1234))
        assert np.array_equal(u8.numpy(), ref.generate_u8(z).numpy())
        del u8, g
        gc.collect()
    finally:
        gc.collect()
        ctx.destroy()
