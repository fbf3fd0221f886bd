"""No register spills inside the K loops of the matrix-core kernels (hipcc cross-compiles to gfx950 assembly here, no GPU needed).
A spilled accumulator tile in such a loop is invisible to every parity test and costs ~10 % on the convolution path; it happened
once when epilogue code was added for tiles that never run it (tools/check_loop_spills.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_no_spills_inside_the_k_loops():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_loop_spills.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 scratch instructions inside the K loop") >= 12
    # the hand-placed fragment reads of gl_pair256.h: nothing touches a destination register between its ds_read and the wait that retires it
    assert r.stdout.count(", 0 instructions touch a fragment register still in flight") == 3, r.stdout
