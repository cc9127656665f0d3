#!/bin/bash
# round 5: the streaming 1x1 kernel (tile 20): parity first (short timeout: a barrier mismatch would hang), then timings
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming_1x1" > gpurun_out/r05_stream_tests.log 2>&1
rc=$?
echo "stream tests rc $rc"
tail -12 gpurun_out/r05_stream_tests.log
if [ $rc -eq 0 ]; then
  timeout -k 10 300 python scripts/ab_tiles.py --libs neuralbarkcalculator_amd/libnbc_hip.so --tiles=-1,20 --rounds 3 > gpurun_out/r05_stream_tiles.log 2>&1
  echo "tiles rc $?"
  python - <<'PY'
import re
t={}
cur=None
for l in open('gpurun_out/r05_stream_tiles.log'):
    m=re.match(r'tile (-?\d+):',l)
    if m: cur=int(m.group(1)); t[cur]={}; continue
    m=re.match(r'\s+(\S+)\s+([\d.]+)',l)
    if m and cur is not None: t[cur][m.group(1)]=float(m.group(2))
tot=[0,0]
for k in t[-1]:
    a,b=t[-1][k],t[20][k]
    if abs(a-b)/a>0.015: print("%-34s %7.1f %7.1f %+6.1f %%"%(k,a,b,100*(b/a-1)))
    tot[0]+=a; tot[1]+=b
print("sum %.1f %.1f"%tuple(tot))
PY
fi
