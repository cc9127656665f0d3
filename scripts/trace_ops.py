#!/usr/bin/env python3
"""Per-op kernel durations of overlapped forwards from a rocprofv3 kernel_trace.csv.
usage: trace_ops.py kernel_trace.csv ops.json   (ops.json: bench.py --dump-ops of the same plan, for names/flops)
Kernels are grouped per HIP stream/queue; a forward starts at its ingest kernel.  For every op position
the mean duration over the last forwards is printed next to the single-stream (event-timed) figure."""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ops = json.load(open(sys.argv[2]))
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
per = collections.defaultdict(list)
for r in rows:
    per[r[key]].append(r)
acc = collections.defaultdict(list)
nfw = 0
t_first, t_last = None, None
for q, rs in per.items():
    rs.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rs) if 'ingest' in r['Kernel_Name']]
    fw = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if b - a == len(ops)]
    for a, b in fw[len(fw) // 2:]:
        nfw += 1
        for k in range(a, b):
            acc[k - a].append((int(rs[k]['End_Timestamp']) - int(rs[k]['Start_Timestamp'])) / 1e3)
        t0, t1 = int(rs[a]['Start_Timestamp']), int(rs[b - 1]['End_Timestamp'])
        t_first = t0 if t_first is None else min(t_first, t0)
        t_last = t1 if t_last is None else max(t_last, t1)
print(f"{len(per)} queues, {nfw} forwards averaged; wall per forward {((t_last - t_first) / 1e3 / max(nfw, 1)):.1f} us")
groups = collections.OrderedDict()
tot_c = tot_s = 0.0
for k, o in enumerate(ops):
    if k not in acc: continue
    c = sum(acc[k]) / len(acc[k]); s1 = o['ms'] * 1e3
    n = o['name']
    g = ('conv2(3x3)' if n.endswith('conv2') else 'conv3(1x1+res)' if n.endswith('conv3') else 'conv1(1x1)' if n.endswith('.conv1') and 'layer' in n
         else 'downsample' if 'downsample' in n else n)
    lay = n.split('.')[1] if n.startswith('backbone.layer') else ''
    groups.setdefault((lay, g), [0.0, 0.0, 0.0]); e = groups[(lay, g)]; e[0] += c; e[1] += s1; e[2] += o['flops']
    tot_c += c; tot_s += s1
print(f"sum of kernel durations per forward: overlapped {tot_c:.0f} us, single-stream (event-timed) {tot_s:.0f} us")
for (lay, g), (c, s1, f) in groups.items():
    print(f"  {lay:7s}{g:18s} overlapped {c:7.1f} us   alone {s1:7.1f} us   x{c / max(s1, 1e-9):4.2f}   {f / 1e9:6.1f} GF")
