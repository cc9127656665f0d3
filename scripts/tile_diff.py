import re,sys
t={}
cur=None
for l in open('gpurun_out/r05_stream_tiles.log'):
    m=re.match(r'tile (-?\d+):',l)
    if m: cur=int(m.group(1)); t[cur]={}; continue
    m=re.match(r'\s+(\S+)\s+([\d.]+)',l)
    if m and cur is not None: t[cur][m.group(1)]=float(m.group(2))
ks=sorted(t)
for k in t[ks[0]]:
    vals=[t[q][k] for q in ks]
    if max(vals)-min(vals)>0.02*min(vals): print("%-34s "%k+" ".join("%7.1f"%v for v in vals))
print("sum "+" ".join("%.1f"%sum(t[q].values()) for q in ks), "tiles", ks)
