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
