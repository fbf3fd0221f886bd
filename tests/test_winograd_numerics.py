"""CPU: the error study behind the decision NOT to build Winograd on the split-fp16 convolution path (DESIGN.md section 5, round 3;
tools/winograd_numerics.py emulates the device arithmetic in numpy).  The review's kill criterion was an error above 2x the direct split path's:
pinned here so that the evidence can be re-run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_winograd_error_ratios(capsys):
    import winograd_numerics as wn
    rng = np.random.default_rng(0)
    # DCGAN layer 1 shape: a 2x2-tap phase of ConvTranspose k4 s2 p1 on a 4 x 4 grid, 1024 input channels
    x = np.maximum(rng.standard_normal((1024, 4, 4)) * 0.5 + 0.1, 0).astype(np.float32)
    x = np.pad(x, ((0, 0), (1, 0), (1, 0)))
    w = (rng.standard_normal((32, 1024, 2, 2)) * 0.02).astype(np.float32)
    r = wn.study("dcgan layer 1", x, w, 2)
    assert r["ratio_rms"] > 2.0 and r["ratio_max"] > 2.0                       # fails the criterion
    assert 0.5 < r["direct_split_rms"] / r["direct_fp32_rms"] < 1.5            # the direct split path is fp32-class
    # VGG16 3 x 3 at 512 channels: at the limit; at 64 channels: fine
    x = np.pad(np.maximum(rng.standard_normal((512, 8, 8)), 0).astype(np.float32), ((0, 0), (1, 1), (1, 1)))
    w = (rng.standard_normal((32, 512, 3, 3)) * np.sqrt(2.0 / (9 * 512))).astype(np.float32)
    r512 = wn.study("vgg 512", x, w, 3)
    assert r512["ratio_rms"] > 1.8
    # the extra error is Winograd's own, not the split's: an all-fp32 Winograd is as far from the fp64 result
    ref = wn.conv_direct_f64(x, w)
    e32 = np.sqrt(((wn.winograd_fp32_only(x, w, wn.BT3, wn.G3, wn.AT3) - ref) ** 2).mean())
    assert 0.8 < r512["winograd_split_rms"] / e32 < 1.25
    x = np.pad(np.maximum(rng.standard_normal((64, 32, 32)), 0).astype(np.float32), ((0, 0), (1, 1), (1, 1)))
    w = (rng.standard_normal((32, 64, 3, 3)) * np.sqrt(2.0 / (9 * 64))).astype(np.float32)
    assert wn.study("vgg 64", x, w, 3)["ratio_rms"] < 1.5
    capsys.readouterr()
