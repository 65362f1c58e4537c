"""Build libgandanet_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libgandanet_hip.so")
SOURCES = ["api.hip", "gemm_conv.hip", "conv3x3.hip", "gemm_nt.hip", "norm.hip", "resample.hip", "pointwise.hip", "nhwc.hip", "disc_nhwc.hip", "pam.hip", "pam_bwd64.hip", "comm.hip", "histmatch.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "gandanet.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
