#!/usr/bin/env python3
"""What bounds the halo convolution kernel?  Timing experiments of the tuning build (GL_HALO_DIAG; results are wrong on purpose): the direct 3x3
convolution of csrc/gl_tune.hip on halo-shaped layers with parts of the kernel switched off, alternating in one process."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GANLEAKS_LIB", os.path.join(ROOT, "gan-leaks_amd", "libganleaks_hip_tuning.so"))
import ganleaks_amd as gl
from ganleaks_amd._lib import check
ctx = gl.Context.get()
fn = ctx.lib.gl_tune_winograd_bound
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
shapes = [("VGG conv1_2 64->64 @64x64 x512", 64, 64, 64, 512), ("VGG conv2_2 128->128 @32x32 x1024", 128, 128, 32, 1024),
          ("PGGAN 128->128 @128x128 x64", 128, 128, 128, 64), ("PGGAN 64->64 @256x256 x32", 64, 64, 256, 32)]
names = {0: "full", 1: "no weight DMA in the loop", 3: "no DMA, no barrier", 4: "no epilogue", 7: "MFMAs + fragment reads only"}
for name, cin, cout, hw, n in shapes:
    row = {"layer": name}
    for r in range(2):
        for d in names:
            os.environ["GL_HALO_DIAG"] = str(d)
            ms = (ctypes.c_float * 2)()
            check(fn(ctx.handle, cin, cout, hw, hw, n, 5, ms))
            row.setdefault(names[d], []).append(round(float(ms[0]), 4))
    flop = 2.0 * 9 * cin * cout * n * hw * hw
    row["full_alg_tflops"] = round(flop / min(row["full"]) / 1e9, 1)
    print(json.dumps(row), flush=True)
