"""bench.py's output contract: exactly one JSON line on stdout with the keys the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line(built_lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"].startswith("1024x1024 images/sec") and d["unit"] == "images/s"
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]      # batch 1, one rank
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "images/s" and c["cores"] >= 1 and c["kind"] == "port" and isinstance(c["sample"], str)
    assert d["value"] > 100 * c["value"]                                                          # three orders of magnitude, in fact
    assert d["parity"]["label_match"] > 0.98 and d["fp32_parity_mode"]["label_match"] > 0.9999
