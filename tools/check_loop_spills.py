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
    "gl_conv_h3.hip": ["gather_conv_h3_kernelILi2ELi4ELi8ELi4ELb0ELb0ELb0E", "gather_conv_h3_kernelILi2ELi4ELi8ELi4ELb0ELb1ELb0E",
                       "gather_conv_h3_kernelILi2ELi4ELi8ELi4ELb0ELb0ELb1E",
                       "gather_conv_h3_kernelILi1ELi8ELi8ELi4ELb0ELb0ELb0E", "gather_conv_h3_kernelILi1ELi8ELi8ELi4ELb0ELb1ELb0E",
                       "gather_conv_h3_kernelILi2ELi2ELi4ELi4ELb0ELb0ELb0E", "gather_conv_h3_kernelILi1ELi4ELi4ELi4ELb0ELb0ELb0E"],
    "gl_conv_halo.hip": ["halo_conv_h3_kernelILi1E", "halo_conv_h3_kernelILi2E"],
    "gl_l2knn.hip": ["l2_knn_i8_256p_kernelILi0ELi8E"],
    "gl_lpips.hip": ["feat_knn_h1c_kernel", "feat_knn_h1s_kernel"],
}
# kernels built on gl_pair256.h's main loop: their fragment reads are inline-asm ds_read_b128 whose waits are placed by hand, so the compiler
# does not know a destination register is still in flight -- any instruction it puts between a read and the s_waitcnt that retires it and that
# touches the destination (a copy at a loop phi, a spill) would use stale data.  Checked on the assembly below.
PIPELINED = {"gl_l2knn.hip": ["l2_knn_i8_256p_kernelILi0ELi8E"], "gl_lpips.hip": ["feat_knn_h1c_kernel", "feat_knn_h1s_kernel"]}


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


def _vregs(text):
    """VGPR numbers named in an operand string: v12, v[12:15]"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        regs.add(int(a))
    return regs


def inflight_hazards(asm, needle):
    """instructions that touch the destination of a ds_read_b128 before an `s_waitcnt lgkmcnt(N)` has retired it.  LDS reads return in issue
    order, so lgkmcnt(N) retires all but the N youngest; anything else that names an in-flight register is reported."""
    m = re.search(r"^(_Z\S*%s\S*):" % re.escape(needle), asm, re.M)
    if not m:
        return None
    name = m.group(1)
    body = asm[m.end():asm.index(".amdhsa_kernel " + name)]
    lines = [l.split(";")[0].strip() for l in body.split("\n")]
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    bad, reads = [], 0

    def scan(lo, hi, queue, count):
        nonlocal reads
        i = lo
        while i < hi:
            ins = lines[i]
            i += 1
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op = ins.split()[0]
            if op.startswith("ds_read") or op.startswith("ds_load"):
                queue.append(_vregs(ins[len(op):].split(",")[0]))
                reads += count
                continue
            if op == "s_waitcnt":
                mm = re.search(r"lgkmcnt\((\d+)\)", ins)
                if mm:
                    keep = int(mm.group(1))
                    del queue[:len(queue) - keep if keep else len(queue)]
                continue
            if op.startswith("s_cbranch") or op == "s_branch":
                target = labels.get(ins.split()[-1])
                if count and target is not None and target < i:
                    scan(target, i - 1, list(queue), 0)          # the loop body once more, entered with what is in flight at the back edge
                continue
            if op.startswith("s_") or (op.startswith("buffer_load") and " lds" in ins):
                continue
            flying = set().union(*queue) if queue else set()
            hit = flying & _vregs(ins[len(op):])
            if hit:
                bad.append((ins, sorted(hit)))
        return queue

    scan(0, len(lines), [], 1)
    return reads, bad


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
            for n in PIPELINED.get(src, []):
                r = inflight_hazards(asm, n)
                if r is None:
                    print("%-20s %-52s not found" % (src, n))
                    bad += 1
                    continue
                reads, hazards = r
                print("%-20s %-52s %d LDS reads, %d instructions touch a fragment register still in flight" % (src, n, reads, len(hazards)))
                for ins, regs in hazards[:5]:
                    print("        %s    (v%s)" % (ins, regs))
                if hazards or reads == 0:
                    bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
