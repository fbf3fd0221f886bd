"""GPU: PyTorch-ROCm as plumbing only -- device tensors passed by pointer, the library running on PyTorch's stream,
device buffers aliased as tensors for the collective -- and the in-library kernel profiler."""
import ctypes

import numpy as np
import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu


def test_torch_device_tensors_and_stream(synth):
    import torch
    import c_oracle
    import ganleaks_amd as gl
    from ganleaks_amd._lib import check
    ctx = gl.Context.get()
    lib = ctx.lib
    p = ctypes.c_void_p
    case = synth.attack_case(101, 640, 30, 30, 32)
    q = np.concatenate([case["pos"], case["neg"]])
    dev = torch.device("cuda", ctx.device)
    # high level: ROCm tensors go in without a host round trip
    d, i = gl.attack(torch.from_numpy(q).to(dev), torch.from_numpy(case["bank"]).to(dev), batch_size=64)
    od, oi, _ = c_oracle.knn_l2_u8(case["bank"], q, 64)
    assert np.array_equal(i, oi) and np.array_equal(d, od)
    # low level: raw data_ptr()s on a PyTorch side stream
    stream = torch.cuda.Stream(device=dev)
    D = 3 * 32 * 32
    stride = int(lib.gl_l2_row_stride(D))
    with torch.cuda.stream(stream):
        bank_t = torch.from_numpy(case["bank"].reshape(640, D)).to(dev, non_blocking=False)
        q_t = torch.from_numpy(q.reshape(60, D)).to(dev)
        bank_i8 = torch.empty((640, stride), dtype=torch.int8, device=dev)
        q_i8 = torch.empty((60, stride), dtype=torch.int8, device=dev)
        bn = torch.empty(640, dtype=torch.int32, device=dev)
        qn = torch.empty(60, dtype=torch.int32, device=dev)
        keys = torch.empty(60, dtype=torch.int64, device=dev)
        dist_t = torch.empty(60, dtype=torch.float32, device=dev)
        idx_t = torch.empty(60, dtype=torch.int64, device=dev)
        ctx.set_stream(stream.cuda_stream)
        try:
            assert ctx.stream == stream.cuda_stream
            check(lib.gl_l2_prepare(ctx.handle, p(bank_t.data_ptr()), 640, D, p(bank_i8.data_ptr()), p(bn.data_ptr())))
            check(lib.gl_l2_prepare(ctx.handle, p(q_t.data_ptr()), 60, D, p(q_i8.data_ptr()), p(qn.data_ptr())))
            check(lib.gl_keys_init(ctx.handle, p(keys.data_ptr()), 60))
            check(lib.gl_l2_knn_i8(ctx.handle, p(bank_i8.data_ptr()), p(bn.data_ptr()), 640, 0, p(q_i8.data_ptr()), p(qn.data_ptr()), 60, D, p(keys.data_ptr())))
            k2 = torch.minimum(keys, keys)                       # a torch op on the same stream sees the kernel's result
            check(lib.gl_keys_unpack(ctx.handle, p(k2.data_ptr()), 60, D, p(dist_t.data_ptr()), p(idx_t.data_ptr())))
        finally:
            ctx.set_stream(0)
        stream.synchronize()
    assert np.array_equal(idx_t.cpu().numpy(), oi) and np.array_equal(dist_t.cpu().numpy(), od)
    assert ctx.stream != stream.cuda_stream
    # library-owned buffer aliased as a tensor (what shard.allreduce_min_keys hands to RCCL)
    arr = ctx.to_device(np.arange(10, dtype=np.int64))
    t = torch.as_tensor(arr, device=dev)
    t += 5
    torch.cuda.synchronize()
    assert arr.numpy().tolist() == list(range(5, 15))


def test_profiler_tags(synth):
    import ganleaks_amd as gl
    ctx = gl.Context.get()
    case = synth.attack_case(102, 256, 4, 4, 16)
    ctx.prof_reset()
    ctx.prof_enable(True)
    gl.attack(case["pos"], case["bank"], batch_size=64)
    ctx.prof_enable(False)
    prof = ctx.prof_read()
    assert prof["l2_knn"][1] == 1 and prof["l2_knn"][0] > 0
    assert prof["l2_prepare"][1] == 2 and prof["gather_conv"][1] == 0
    ctx.prof_reset()
    assert ctx.prof_read()["l2_knn"][1] == 0
