#!/usr/bin/env python3
"""Group the per-launch records bench.py --dump-ops wrote (scratch analysis helper)."""
import json
import sys

def report(path, detail=False):
    ops = json.load(open(path))
    tot = sum(o['ms'] for o in ops)
    print(path, 'sum of launches %.3f ms' % tot)
    groups = {}
    for o in ops:
        n = o['name']
        key = o['kernel']
        if o['kernel'] == 'conv_igemm':
            if n.endswith('conv3'): key = 'conv3(1x1+res)'
            elif n.endswith('conv1') and 'layer' in n: key = 'conv1(1x1)'
            elif n.endswith('conv2'): key = 'conv2(3x3)'
            elif 'downsample' in n: key = 'downsample'
            elif n == 'classifier.0': key = 'head3x3'
            else: key = 'stem'
        g = groups.setdefault(key, [0.0, 0.0, 0.0])
        g[0] += o['ms']; g[1] += o['flops']; g[2] += o['bytes']
    for k, (ms, fl, by) in sorted(groups.items(), key=lambda t: -t[1][0]):
        print(f"  {k:16s} {ms*1000:8.1f} us {100*ms/tot:5.1f}%  {fl/(ms*1e-3)/1e12 if ms else 0:7.0f} TF {by/(ms*1e-3)/1e9 if ms else 0:7.0f} GB/s")
    if detail:
        for o in ops:
            tf = o['flops'] / (o['ms'] * 1e-3) / 1e12 if o['ms'] > 0 else 0
            print(f"    {o['name']:34s} {o['ms']*1000:8.1f} us  {tf:7.0f} TF  {o['bytes']/(o['ms']*1e-3)/1e9:7.0f} GB/s")

if __name__ == '__main__':
    for p in sys.argv[1:]:
        if p != '-d':
            report(p, '-d' in sys.argv)
