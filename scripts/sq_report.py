#!/usr/bin/env python3
"""Per-layer SQ counter table from a rocprofv3 counter_collection.csv of one bench run."""
import csv, json, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ops = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
by = defaultdict(dict)
names = {}
for r in rows:
    d = int(r['Dispatch_Id'])
    by[d][r['Counter_Name']] = float(r['Counter_Value'])
    names[d] = r['Kernel_Name']
    by[d]['_dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
ds = sorted(by)
ing = [d for d in ds if 'ingest' in names[d]]
a, b = ing[-2], ing[-1]
seq = [d for d in ds if a <= d < b and 'fillBuffer' not in names[d] and 'elementwise' not in names[d]]
ctrs = sorted(k for k in by[seq[1]] if not k.startswith('_'))
print('%-32s %8s ' % ('op', 'us') + ' '.join('%12s' % c[-12:] for c in ctrs))
for i, d in enumerate(seq):
    nm = ops[i]['name'] if ops and i < len(ops) else names[d][:30]
    if ops and not any(s in nm for s in ('layer4.1', 'layer3.1', 'layer2.1', 'layer1.1', 'classifier.0', 'backbone.conv1')):
        continue
    print('%-32s %8.1f ' % (nm, by[d]['_dur'] / 1e3) + ' '.join('%12.3g' % by[d].get(c, float('nan')) for c in ctrs))
