"""bench.py's output contract: exactly one JSON line on stdout with the keys the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]      # ONE line on stdout, the JSON object
    return json.loads(lines[0])


def check_roofline(r, peak):
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == peak
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic FLOPs per launch / average RAW launch duration
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert isinstance(r["traffic_note"], str)
    # the rocprofv3 table's figure for the same kernel rides along whenever the committed table belongs to these kernel
    # sources (the mechanism `traffic` uses); the raw-event frac reads a few percent under it
    if r["traffic"] is not None:
        assert r["table"].startswith("profiles/per_forward_ops_") and 0 < r["frac_rocprof_table"] < 1
        hc = r["head_conv_rocprof_table"]                 # the longest launch: its in-kernel clock next to the table's frac
        assert hc["op"] == "classifier.0" and 0.5 < hc["clock_ghz_in_pmc_pass"] < 3.0 and 0 < hc["frac_of_peak"] < 1
    else:
        assert "frac_rocprof_table" not in r and "table" not in r and "head_conv_rocprof_table" not in r


def test_default_bench_line(built_lib):
    """The driver's command shape: the headline is configs[1] in the f32-grade f16x2 mode (same parity bounds as the f32
    MFMA mode, which rides along as its own object), configs[2] (bf16, batch 8) as another."""
    d = run_bench("--steps", "8", "--warmup", "2")
    assert d["metric"].startswith("1024x1024 images/sec") and d["unit"] == "images/s"
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["rccl_ranks"] == 1
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]      # batch 1, one rank
    # the line's own spread: the bracketed region five times, `value` the first of them (the contract's)
    assert len(d["value_repeats"]) == 5 and d["value_repeats"][0] == d["value"]
    assert d["value_median"] == sorted(d["value_repeats"])[2] and min(d["value_repeats"]) > 0.5 * d["value_median"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16x2" and "f32-grade" in d["dtype_note"] and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("configs[1]") and "model" not in d["config"]
    assert d["config"]["batch"] == 1 and d["config"]["precision"] == "f16x2"
    assert d["config"]["frames"] == 100             # configs[1]: 100 synthetic frames, resident in HBM
    check_roofline(d["roofline"], 2500.0 / 3.0)
    assert d["roofline"]["mfma_flops_per_algorithmic_flop"] == 3
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "images/s" and c["cores"] >= 1 and c["kind"] == "port" and isinstance(c["sample"], str)
    assert d["value"] > 50 * c["value"]
    # the f32-grade modes: the label mask equals the oracle's but for exact logit ties (at most 4 per megapixel, the bound of
    # tests/test_gpu_parity.py), low-res logits within 5e-6 of the oracle's logit range
    for par, prec in ((d["parity"], "f16x2"), (d["f32_mfma_batch1"]["parity"], "fp32")):
        assert par["precision"] == prec and par["frames_compared"] == 4 and par["pixels"] == 4 * 1024 * 1024
        assert max(par["per_frame_label_mismatches"]) <= 4
        assert par["max_oracle_margin_at_mismatch"] <= 1e-5 * par["oracle_logit_range"]
        assert par["max_lowres_logit_err_over_oracle_range"] <= 5e-6
    f = d["f32_mfma_batch1"]
    assert f["config"]["workload"].startswith("configs[1]") and f["dtype"] == "f32" and f["value"] > 0
    check_roofline(f["roofline"], 157.3)
    assert d["value"] > 1.5 * f["value"]            # measured 2.2-2.3x
    # configs[2] rides along as its own object with its own parity and roofline, never as the headline
    b = d["bf16_batch8"]
    assert b["config"]["workload"].startswith("configs[2]") and b["config"]["batch"] == 8 and b["dtype"] == "bf16"
    assert b["value"] > 0 and abs(b["value"] - 8e3 / b["ms_per_step"]) < 1e-6 * b["value"]
    check_roofline(b["roofline"], 2500.0)
    assert b["parity"]["precision"] == "bf16" and b["parity"]["label_match"] > 0.98
    assert b["parity"]["pixels"] == 8 * 1024 * 1024 and b["parity"]["frames_compared"] == 8


def test_bench_starts_its_own_ranks(built_lib):
    """`bench.py --gpus 2` with no torchrun environment starts two ranks itself (here both on cuda:0 over
    gloo: a one-GPU box) and prints one line for the two-rank job: weight broadcast, sharded frames,
    max-over-ranks timing, the row gather."""
    d = run_bench("--gpus", "2", "--share-gpu", "--steps", "4", "--warmup", "1", "--no-bf16-leg", "--frames", "4")
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo"
    assert d["steps"] == 4 and abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert d["setup"]["weight_broadcast_s"] > 0 and d["setup"]["process_group_init_s"] > 0
    assert d["gather_s"] is not None and d["gather_s"] >= 0
    assert "cpu_baseline" not in d and "parity" not in d          # N = 1 business


def test_gpus_mismatch_is_an_error(built_lib):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr and not p.stdout.strip()
