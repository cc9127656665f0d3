"""Builds libnbc_hip.so (gfx950 only) in-tree with hipcc.  No torch headers are involved:
the library is a plain C-ABI shared object (include/nbc.h) loaded through ctypes."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libnbc_hip.so")
SOURCES = ["nbc_net.cpp", "conv_igemm_dma.hip", "conv3x3_rows.hip", "pointwise.hip", "small_zones.hip", "nbc_api.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libnbc_hip.so")
    return exe


def _deps_newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every source for gfx950 and link the shared library; returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "nbc.h"))
    common = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]

    def compile_one(src: str) -> str:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        if force or _deps_newer(obj, [path] + headers):
            cmd = [hipcc] + common + (["-x", "hip"] if src.endswith(".hip") else ["-ffp-contract=off"]) + ["-c", path, "-o", obj]
            if verbose:
                print("[nbc build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _deps_newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print("[nbc build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
