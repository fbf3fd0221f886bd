#!/usr/bin/env python3
"""Measured upper bound of Winograd F(2x2,3x3) on the split-fp16 convolution path (VERDICT r2 "Next 1"; csrc/gl_tune.hip says what is timed):
for every 3x3 layer shape of VGG16 at 64 x 64 (and PGGAN's blocks) the direct convolution against the 16 GEMMs Winograd would run instead --
both on the SAME tuned kernel, random operands, alternating, one process -- plus the bytes its two transforms would move.

    python tools/bench_winograd_bound.py > profiles/rNN/rNN_winograd_bound.json        (needs `make -C gan-leaks_amd/csrc tuning`)
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GANLEAKS_LIB", os.path.join(ROOT, "gan-leaks_amd", "libganleaks_hip_tuning.so"))
import ganleaks_amd as gl  # noqa: E402
from ganleaks_amd._lib import check  # noqa: E402

STREAM_TBS = 5.0      # what the streaming kernels of this library reach (col2im 5.05, l2_prepare 5.57 TB/s): price of the transform passes


def main():
    ctx = gl.Context.get()
    fn = ctx.lib.gl_tune_winograd_bound
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    shapes = [("VGG16 conv1_2", 64, 64, 64, 512), ("VGG16 conv2_1", 64, 128, 32, 1024), ("VGG16 conv2_2", 128, 128, 32, 1024),
              ("VGG16 conv3_1", 128, 256, 16, 2048), ("VGG16 conv3_2/3", 256, 256, 16, 2048), ("VGG16 conv4_1", 256, 512, 8, 2048),
              ("VGG16 conv4_2/3", 512, 512, 8, 2048), ("VGG16 conv5_x", 512, 512, 4, 2048),
              ("PGGAN 512ch 16x16", 512, 512, 16, 256), ("PGGAN 512ch 32x32", 512, 512, 32, 128), ("PGGAN 256ch 64x64", 256, 256, 64, 64)]
    rows = []
    for name, cin, cout, hw, n in shapes:
        ms = (ctypes.c_float * 2)()
        runs = []
        for _ in range(3):
            check(fn(ctx.handle, cin, cout, hw, hw, n, 5, ms))
            runs.append((ms[0], ms[1]))
        direct, gemm = float(np.median([r[0] for r in runs])), float(np.median([r[1] for r in runs]))
        P = n * hw * hw
        flop = 2.0 * 9 * cin * cout * P
        t_in = P * (4 * cin + 16 * cin) / (STREAM_TBS * 1e12) * 1e3            # input transform: read the activation, write the 16 V_j
        t_out = P * (16 * cout + 4 * cout) / (STREAM_TBS * 1e12) * 1e3         # output transform: read the 16 M_j, write the activation
        rows.append({"layer": name, "C_in": cin, "C_out": cout, "HxW": hw, "images": n,
                     "direct_ms": round(direct, 4), "direct_alg_tflops": round(flop / direct / 1e9, 1),
                     "winograd_gemm_stage_ms": round(gemm, 4), "gemm_stage_alg_tflops_of_its_own_flops": round(flop / 2.25 / gemm / 1e9, 1),
                     "bound_if_transforms_were_free": round(direct / gemm, 3),
                     "transform_passes_ms_at_%g_TBs" % STREAM_TBS: round(t_in + t_out, 4),
                     "speedup_unfused": round(direct / (gemm + t_in + t_out), 3)})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    print(json.dumps({"what": __doc__.split("\n\n")[0], "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
