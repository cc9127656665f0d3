#!/usr/bin/env python3
"""Per-forward table of the hot path from rocprofv3 output of `bench.py` (one configuration, one stream).

usage: per_forward_table.py <prof_dir> <out.json> --precision fp32|bf16 --batch B --steps K --warmup W
  <prof_dir>/trace   rocprofv3 --kernel-trace --stats -- python3 bench.py ... --dump-ops <prof_dir>/ops.json
  <prof_dir>/fetch   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py ... --no-op-events   (optional)
  <prof_dir>/write   rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python3 bench.py ... --no-op-events   (optional)
  <prof_dir>/mfma    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- ...           (optional)

A forward is the run of nbc kernels from an `ingest` kernel to the next `upsample_argmax` with exactly the
plan's number of launches in between; the autotune launches (conv kernels with no ingest in front) are
therefore never counted.  Forward 0 is autotune's own, the next W are warm-up, the next K are THE TIMED
REGION of bench.py: the table holds, per launch of the plan, the median duration over those K forwards,
its algorithmic FLOPs and bytes (from bench.py --dump-ops), and the fabric traffic of the PMC passes
(FETCH_SIZE x 2 per MI355X_MICROARCH.md section HBM, + WRITE_SIZE; KiB -> bytes).  The mfma pass adds the
matrix pipes' utilisation by hardware counters, rocprofiler-sdk's MfmaUtil for gfx950:
SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs: MFMA count x cycles per MFMA) / (GRBM_GUI_ACTIVE per XCD x 1024
SIMDs), i.e. the share of the launch's shader cycles in which a SIMD's matrix pipe is busy, and the shader clock
the launch ran at (GRBM_GUI_ACTIVE per XCD / its duration in that pass: reads high on launches under 0.3 ms).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import statistics

PEAK = {"bf16": 2500.0, "fp32": 157.3, "f16x2": 2500.0 / 3.0}   # f16x2: algorithmic FLOPs, three f16 MFMA FLOPs each
HBM = 6.29e12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_id():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neuralbarkcalculator_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def nbc_rows(path):
    rows = [r for r in csv.DictReader(open(path)) if "nbc::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def forwards(rows, nops):
    out = []
    idx = [i for i, r in enumerate(rows) if "ingest" in r["Kernel_Name"]]
    for a in idx:
        b = a + nops
        if b > len(rows) or "upsample_argmax" not in rows[b - 1]["Kernel_Name"]:
            continue
        if any("ingest" in r["Kernel_Name"] for r in rows[a + 1:b]):
            continue
        out.append(rows[a:b])
    return out


def short(name):
    n = name.replace("void nbc::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prof")
    ap.add_argument("out")
    ap.add_argument("--precision", required=True)
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--no-autotune", action="store_true", help="the bench ran with --no-autotune / installed tiles: no forward 0")
    a = ap.parse_args()
    ops = json.load(open(os.path.join(a.prof, "ops.json")))
    nops = len(ops)
    first = (0 if a.no_autotune else 1) + a.warmup

    trace = glob.glob(os.path.join(a.prof, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
    fw = forwards(nbc_rows(trace), nops)
    timed = fw[first:first + a.steps]
    assert len(timed) == a.steps, (len(fw), first, a.steps)
    dur = [[(int(f[k]["End_Timestamp"]) - int(f[k]["Start_Timestamp"])) / 1e3 for f in timed] for k in range(nops)]
    wall = [(int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in timed]

    def pmc(kind, counter=None, scale=1024.0, with_duration=False):
        g = glob.glob(os.path.join(a.prof, kind, "**", "*counter_collection.csv"), recursive=True)
        if not g:
            return None
        rows = [r for r in csv.DictReader(open(g[0])) if "nbc::" in r["Kernel_Name"] and (counter is None or r["Counter_Name"] == counter)]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        f = forwards(rows, nops)[first:first + a.steps]
        if not f:
            return None
        vals = [statistics.median(float(x[k]["Counter_Value"]) for x in f) * scale for k in range(nops)]
        if with_duration:
            return vals, [statistics.median((int(x[k]["End_Timestamp"]) - int(x[k]["Start_Timestamp"])) / 1e3 for x in f) for k in range(nops)]
        return vals

    fetch, write = pmc("fetch"), pmc("write")
    busy = pmc("mfma", "SQ_VALU_MFMA_BUSY_CYCLES", 1.0)
    gui = pmc("mfma", "GRBM_GUI_ACTIVE", 1.0, with_duration=True)
    SIMDS, XCDS = 1024, 8
    peak = PEAK[a.precision]
    table = []
    for k, o in enumerate(ops):
        med = statistics.median(dur[k])
        row = {"op": o["name"], "kernel": short(timed[0][k]["Kernel_Name"]), "k": o["k"], "cout": o["cout"],
               "grid": int(timed[0][k]["Grid_Size_X"]) // max(1, int(timed[0][k]["Workgroup_Size_X"])),
               "median_us": round(med, 2), "min_us": round(min(dur[k]), 2), "event_ms_in_bench": o["ms"],
               "flops": o["flops"], "algorithmic_bytes": o["bytes"],
               "tflops": round(o["flops"] / med / 1e6, 1) if o["flops"] else 0.0,
               "bound_us": round(max(o["flops"] / (peak * 1e12), o["bytes"] / HBM) * 1e6, 2)}
        if fetch and write:
            row["fetch_x2_bytes"] = fetch[k] * 2.0
            row["write_bytes"] = write[k]
            row["traffic_over_algorithmic"] = round((fetch[k] * 2.0 + write[k]) / o["bytes"], 2) if o["bytes"] else None
        if busy and gui and gui[0][k] > 0:
            cyc = gui[0][k] / XCDS                       # shader cycles of the launch (rocprofv3 sums the 8 XCDs)
            row["mfma_busy_cycles"] = busy[k]
            row["shader_cycles"] = round(cyc, 0)
            row["mfma_util"] = round(busy[k] / (cyc * SIMDS), 4)
            row["clock_ghz_in_pmc_pass"] = round(cyc / gui[1][k] / 1e3, 3)
        table.append(row)

    def agg(rows):
        us = sum(r["median_us"] for r in rows)
        fl = sum(r["flops"] for r in rows)
        d = {"launches": len(rows), "sum_median_us": round(us, 1), "flops": fl,
             "tflops": round(fl / us / 1e6, 1) if us else 0.0, "frac_of_peak": round(fl / us / 1e6 / peak, 4) if us else 0.0,
             "bound_us": round(sum(r["bound_us"] for r in rows), 1)}
        if fetch and write:
            d["traffic_bytes"] = sum(r["fetch_x2_bytes"] + r["write_bytes"] for r in rows)
            d["algorithmic_bytes"] = sum(r["algorithmic_bytes"] for r in rows)
        if busy and gui and all("mfma_util" in r for r in rows) and rows:
            d["mfma_util"] = round(sum(r["mfma_busy_cycles"] for r in rows) / (sum(r["shader_cycles"] for r in rows) * SIMDS), 4)
        return d

    # the convolution kernels: the generic LDS-DMA kernel and (f16x2) the row-resident 3x3 kernel
    conv = [r for r in table if "conv_dma_kernel" in r["kernel"] or "conv3x3_rowstep_kernel" in r["kernel"]]
    dom = [r for r in conv if r["cout"] % 128 == 0 and r["op"] != "backbone.conv1"]
    stages = {}
    for r in table:
        n = r["op"]
        st = n.split(".")[1] if n.startswith("backbone.layer") else ("stem" if n in ("ingest", "backbone.conv1", "backbone.maxpool") else
                                                                    "head" if n.startswith("classifier") else n)
        stages.setdefault(st, []).append(r)
    out = {
        "source": "rocprofv3 --kernel-trace of `python3 bench.py --precision %s --batch %d --streams 1 --steps %d --warmup %d --no-bf16-leg "
                  "--no-cpu-baseline --no-parity`; medians over the %d forwards of bench.py's timed region (autotune launches excluded)"
                  % (a.precision, a.batch, a.steps, a.warmup, a.steps),
        "precision": a.precision, "batch": a.batch, "kernel_source_id": kernel_source_id(),
        "peak_tflops": peak, "hbm_bytes_per_s_for_bound": HBM,
        "forward_wall_us_median": round(statistics.median(wall), 1),
        "whole_forward": agg(table), "dominant_kernel": agg(dom), "conv3x3": agg([r for r in conv if r["k"] == 3]),
        "conv1x1": agg([r for r in conv if r["k"] == 1]),
        "stages": {k: agg(v) for k, v in stages.items()},
        "ops": table,
    }
    d = out["dominant_kernel"]
    d["avg_launch_us"] = round(d["sum_median_us"] / d["launches"], 2)
    if fetch and write:
        d["traffic_bytes_per_launch"] = d["traffic_bytes"] / d["launches"]
        d["algorithmic_bytes_per_launch"] = d["algorithmic_bytes"] / d["launches"]
    stats = glob.glob(os.path.join(a.prof, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        out["rocprofv3_stats_top"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                       "pct": float(r["Percentage"])} for r in list(csv.DictReader(open(stats[0])))[:10]]
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("forward_wall_us_median", "whole_forward", "dominant_kernel", "conv3x3", "stages")}, indent=1))


if __name__ == "__main__":
    main()
