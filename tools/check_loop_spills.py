#!/usr/bin/env python3
"""Compile the matrix-core kernels to gfx950 assembly (device side only, no GPU needed) and report, per hot kernel, how many scratch
(spill) instructions sit between its first and its last MFMA -- i.e. inside the K loop.  One spilled accumulator tile there costs a
scratch reload and an `s_waitcnt vmcnt(0)` per K slice, which drains the LDS-DMA queue: 9 % on every wide convolution the day code
that the 256-channel tile never runs was added to its epilogue (round 2).  tests/test_loop_spills.py asserts zero for the kernels below.

    python tools/check_loop_spills.py            one line per kernel; exit status 1 if any count is non-zero
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gan-leaks_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# file -> substrings of the mangled names of kernels whose MFMAs are all in the K loop (no fused-tail instantiations: their epilogue
# multiplies again) and that are on the measured paths
HOT = {
    "gl_conv_h3.hip": ["gather_conv_h3_kernelILi2ELi4ELi8ELi4ELb0ELb0E", "gather_conv_h3_kernelILi2ELi4ELi8ELi4ELb0ELb1E",
                       "gather_conv_h3_kernelILi1ELi8ELi8ELi4ELb0ELb0E", "gather_conv_h3_kernelILi1ELi8ELi8ELi4ELb0ELb1E",
                       "gather_conv_h3_kernelILi2ELi2ELi4ELi4ELb0ELb0E", "gather_conv_h3_kernelILi1ELi4ELi4ELi4ELb0ELb0E"],
    "gl_conv_halo.hip": ["halo_conv_h3_kernelILi1E", "halo_conv_h3_kernelILi2E"],
    "gl_l2knn.hip": ["l2_knn_i8_256p_kernelILi0ELi8E"],
    "gl_lpips.hip": ["feat_knn_h1c_kernel", "feat_knn_h1p_kernelILi8E"],
}


def loop_spills(asm, needle):
    m = re.search(r"^(_Z\S*%s\S*):" % re.escape(needle), asm, re.M)
    if not m:
        return None
    name = m.group(1)
    body = asm[m.end():asm.index(".amdhsa_kernel " + name)]
    lines = body.split("\n")
    mfma = [i for i, l in enumerate(lines) if "v_mfma" in l]
    if not mfma:
        return None
    return sum(1 for i, l in enumerate(lines) if "scratch_" in l and mfma[0] < i < mfma[-1])


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src, needles in HOT.items():
            out = os.path.join(tmp, src + ".s")
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = open(out).read()
            for n in needles:
                c = loop_spills(asm, n)
                print("%-20s %-52s %s" % (src, n, "not found" if c is None else "%d scratch instructions inside the K loop" % c))
                if c is None or c > 0:
                    bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
