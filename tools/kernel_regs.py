#!/usr/bin/env python3
"""Register / spill / occupancy table of the conv kernel's instantiations (hipcc -Rpass-analysis=kernel-resource-usage).
  python tools/kernel_regs.py [precision]      precision: 0 f32, 1 bf16, 2 f16x2 (default: all)
Columns: template arguments <PREC, WM, WN, MT, NT, S, STEM, VAR, BIGW>, VGPRs, AGPRs, spilled VGPRs, waves per SIMD."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "neuralbarkcalculator_amd", "csrc", "conv_igemm_dma.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-x", "hip", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[2:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
want = sys.argv[1] if len(sys.argv) > 1 else None
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        t = re.search(r"conv_dma_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)E", m.group(1))
        cur = dict(args=tuple(int(v) for v in t.groups())) if t else None
        if cur:
            rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
if "error" in out:
    print(out[-3000:])
print("PREC WM WN MT NT S STEM VAR BIGW | VGPR AGPR spill scratch occ")
for r in sorted(rows, key=lambda r: r["args"]):
    if want is not None and str(r["args"][0]) != want:
        continue
    print("%4d %2d %2d %2d %2d %d %4d %3d %4d | %4d %4d %5d %7d %3d" % (r["args"] + (r.get("vgpr", -1), r.get("agpr", -1), r.get("spill", -1), r.get("scratch", -1), r.get("occ", -1))))
