"""CPU-only: the C-ABI library builds, loads and exports exactly what include/ganleaks.h declares,
and the product path refuses to run (loudly) without a GPU."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gl():
    import ganleaks_amd
    from ganleaks_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ganleaks_amd


def _declared(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(gl):
    from ganleaks_amd import _lib
    lib = _lib.load()
    declared = _declared(_lib.HEADER_PATH)
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "%s declared in ganleaks.h but not exported" % name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.gl_abi_version() == 1


def test_no_silent_cpu_fallback(gl):
    if gl.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(gl.GanLeaksError) as e:
        gl.Context(0)
    assert "no CPU fallback" in str(e.value)
    import numpy as np
    with pytest.raises(gl.GanLeaksError):
        gl.attack(np.zeros((2, 3, 8, 8), np.uint8), np.zeros((64, 3, 8, 8), np.uint8), batch_size=64)


def test_row_stride(gl):
    from ganleaks_amd import _lib
    lib = _lib.load()
    assert lib.gl_l2_row_stride(12288) == 12288
    assert lib.gl_l2_row_stride(300) == 384
    assert lib.gl_l2_row_stride(1) == 128


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gan-leaks_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "c_oracle" not in src, f


def test_shipped_library_has_one_code_path_per_shape(gl):
    """VERDICT r2 weak 11 / ADVICE: the timing-experiment kernels (gl_pair256.h DIAG: wrong results on purpose) and the GL_* tuning
    switches exist only in libganleaks_hip_tuning.so (-DGL_TUNING); the shipped library contains neither, so no stray environment variable
    can change what the product computes.  The tuning build exports the same ABI."""
    import ctypes
    import subprocess
    from ganleaks_amd import _lib
    shipped = os.path.join(os.path.dirname(_lib.HEADER_PATH), "..", "gan-leaks_amd", "libganleaks_hip.so")
    tuning = _lib.TUNING_LIB_PATH
    if not os.path.exists(tuning):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gan-leaks_amd", "csrc"), "-j8", "tuning"], check=True)
    sym = subprocess.run(["nm", "-C", shipped], check=True, stdout=subprocess.PIPE).stdout.decode()
    for pat in (r"l2_knn_i8_256p_kernel<[1-4]", r"l2_knn_i8_256_kernel", r"feat_knn_h1p_kernel", r"feat_knn_h1_kernel"):
        assert not re.search(pat, sym), pat
    raw = open(shipped, "rb").read()
    for name in (b"GL_PAIR_VARIANT", b"GL_L2_TILE", b"GL_H3_HALO", b"GL_TAP_FUSE", b"GL_PIXNORM_FUSE", b"GL_RGB_FUSE", b"GL_H3_TILE128"):
        assert name not in raw, name
        assert name in open(tuning, "rb").read(), name
    assert re.search(r"l2_knn_i8_256p_kernel<[1-4]", subprocess.run(["nm", "-C", tuning], check=True, stdout=subprocess.PIPE).stdout.decode())
    t = ctypes.CDLL(tuning)
    for name in _declared(_lib.HEADER_PATH):
        assert hasattr(t, name), name
