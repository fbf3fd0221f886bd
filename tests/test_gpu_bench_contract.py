"""The driver's contract with bench.py: one JSON line on stdout with the agreed fields, also at reduced sizes and for both distances."""
import json
import os
import subprocess
import sys

import pytest

import gpu_common  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), check=True, stdout=subprocess.PIPE, cwd=ROOT).stdout.decode()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_line_fields():
    d = _run("--steps", "2", "--warmup", "1", "--queries", "512", "--bank", "4096", "--cpu-queries", "8", "--cpu-seconds", "5")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["parity"]["idx_equal"] is True and d["parity"]["max_abs_dist_err"] == 0.0


def test_bench_line_l2_lpips():
    d = _run("--distance", "l2-lpips", "--steps", "1", "--warmup", "1", "--queries", "256", "--bank", "1024", "--cpu-queries", "0")
    assert d["roofline"]["bound"] == "mfma" and d["cpu_baseline"] is None and d["parity"]["idx_equal"] is True
    assert d["parity"]["max_abs_dist_err"] < 1e-5


def test_bench_line_with_secondary_measurements():
    """what the driver's default run prints, at a reduced size: the headline line plus `secondary` (configs[2]: 0.2*LPIPS+L2, own roofline, parity and
    cpu_baseline) and `secondary_fp32` (generator with fp32 MFMA products)"""
    d = _run("--steps", "2", "--warmup", "1", "--queries", "512", "--bank", "4096", "--cpu-queries", "8", "--cpu-seconds", "4", "--secondary", "on")
    assert d["roofline"]["kernel"] == "gather_conv" and d["cpu_baseline"]["kind"] == "port"
    s = d["secondary"]
    assert "configs[2]" in s["config"]["workload"] and s["roofline"]["kernel"] == "feat_knn" and s["roofline"]["bound"] == "mfma"
    assert abs(s["roofline"]["frac"] - s["roofline"]["achieved"] / s["roofline"]["peak"]) < 1e-3
    assert s["parity"]["idx_equal"] is True and s["parity"]["max_abs_dist_err"] < 1e-5
    c = s["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["max_abs_loss_diff_vs_gpu"] < 1e-5 and "pairs" in c["sample"]
    fo = s["cpu_baseline_features_once"]         # the algorithmic and the hardware speed-up, separable
    assert fo["kind"] == "port" and fo["cores"] >= 1 and fo["value"] > c["value"] and fo["idx_equal_gpu"] is True and fo["max_abs_dist_diff_vs_gpu"] < 1e-5
    assert s["algorithmic_speedup_on_cpu"] > 10 and s["speedup_vs_cpu_features_once"] > 1
    c0 = d["config0"]                            # BASELINE configs[0]: 256 x 1k, L2, CPU full run (median of 3) beside the device
    assert "configs[0]" in c0["workload"] and c0["parity"]["idx_equal"] is True and c0["parity"]["max_abs_dist_err"] < 1e-6
    assert c0["cpu_baseline"]["cores"] >= 1 and c0["cpu_baseline"]["value"] > 0 and "median of 3" in c0["cpu_baseline"]["sample"]
    # roofline.traffic of the default run is measured now (child runs under rocprofv3 --pmc), not copied from a committed file
    for blk in (d, s):
        assert isinstance(blk["roofline"]["traffic"], int) and blk["roofline"]["traffic"] > 0 and blk["roofline"]["traffic_source"].startswith("live"), blk["roofline"]
    f = d["secondary_fp32"]
    assert f["dtype"].startswith("f32") and f["roofline"]["peak"] == 157.3 and f["parity"]["idx_equal"] is True
    assert f["ms_per_step"] > 0 and "cpu_baseline" not in f
