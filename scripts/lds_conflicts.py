#!/usr/bin/env python3
"""LDS bank conflicts per kernel from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE run (extra cycles over all
LDS-array cycles: MI355X_MICROARCH.md, LDS):  python scripts/lds_conflicts.py <rocprofv3 output dir>"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
    k = (m.group(1) + (m.group(2) or "")) if m else k[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k] += 1
print("kernel | launches | SQ_LDS_BANK_CONFLICT | SQ_LDS_IDX_ACTIVE | conflict share of LDS-array cycles")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
    print("%s | %d | %.3e | %.3e | %.4f" % (k, n[k] // max(1, len(v)), c, a, c / a if a else 0))
