#!/usr/bin/env python3
"""Busy time and inter-kernel gaps of one forward from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ingest' in r['Kernel_Name']]
for a, b in list(zip(idx[:-1], idx[1:]))[-3:]:
    seg = rows[a:b]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    span = int(rows[b]['Start_Timestamp']) - int(seg[0]['Start_Timestamp'])
    gaps = [int(seg[k + 1]['Start_Timestamp']) - int(seg[k]['End_Timestamp']) for k in range(len(seg) - 1)]
    print("kernels", len(seg), "busy us %.1f" % (busy / 1e3), "span us %.1f" % (span / 1e3),
          "gaps us %.1f" % (sum(gaps) / 1e3), "median gap ns", sorted(gaps)[len(gaps) // 2], "max", max(gaps))
