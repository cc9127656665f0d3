#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) + kernel stats of one bench command.

Usage: pmc_summary.py <prof_dir made by profile_pmc.sh> <ops.json from bench --dump-ops> <out.json>
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md
section HBM); both counters are in KiB.  Traffic is L2-miss traffic (Infinity-Cache hits included).
"""
import csv, glob, json, os, sys

prof, ops_path, out_path = sys.argv[1:4]
ops = json.load(open(ops_path))

def seq(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Dispatch_Id']))
    return [(r['Kernel_Name'], float(r['Counter_Value'])) for r in rows]

f = seq(glob.glob(os.path.join(prof, 'fetch', '*', '*counter_collection.csv'))[0])
w = seq(glob.glob(os.path.join(prof, 'write', '*', '*counter_collection.csv'))[0])
def forwards(s):
    idx = [i for i, (n, _) in enumerate(s) if 'ingest' in n]
    return [[x for x in s[a:b] if 'fillBuffer' not in x[0] and 'elementwise' not in x[0]] for a, b in zip(idx[:-1], idx[1:])]
ff, wf = forwards(f)[-1], forwards(w)[-1]
assert len(ff) == len(ops) == len(wf), (len(ff), len(wf), len(ops))
per_op = []
for o, (kn, fv), (_, wv) in zip(ops, ff, wf):
    per_op.append({"name": o["name"], "kernel": kn.split('(')[0][-60:], "fetch_bytes": fv * 1024 * 2, "write_bytes": wv * 1024,
                   "algorithmic_bytes": o["bytes"], "flops": o["flops"]})
dom = [p for o, p in zip(ops, per_op) if o["kernel"] == "conv_igemm" and o["cout"] % 128 == 0 and o["name"] != "backbone.conv1"]
summary = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), last forward of the run",
    "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), KiB -> bytes",
    "dominant_kernel": {"launches_per_step": len(dom),
                        "traffic_bytes_per_launch": sum(p["fetch_bytes"] + p["write_bytes"] for p in dom) / len(dom),
                        "algorithmic_bytes_per_launch": sum(p["algorithmic_bytes"] for p in dom) / len(dom)},
    "per_step": {"fetch_bytes": sum(p["fetch_bytes"] for p in per_op), "write_bytes": sum(p["write_bytes"] for p in per_op),
                 "algorithmic_bytes": sum(p["algorithmic_bytes"] for p in per_op)},
    "per_op": per_op,
}
stats = glob.glob(os.path.join(prof, 'stats', '*', '*kernel_stats.csv'))
if stats:
    summary["kernel_stats"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
                               for r in list(csv.DictReader(open(stats[0])))[:8]]
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("dominant_kernel", "per_step")}, indent=1))
