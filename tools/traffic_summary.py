"""FETCH_SIZE / WRITE_SIZE passes of tools/pam_bench.py --batch 32 -> profiles/rNN_pam_traffic.json (HBM bytes per
gd_pam_flash_fwd / gd_pam_flash_bwd call at the bench launch shape).  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM:
gfx950 tallies the 128-B requests of wide coalesced reads at 64 B); WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name[:80]


def mean_counter(d, counter):
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


fetch, write = mean_counter(sys.argv[1], "FETCH_SIZE"), mean_counter(sys.argv[2], "WRITE_SIZE")
kern = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("pam_"):
        kern[k] = {"FETCH_SIZE_KiB": fetch.get(k, 0.0), "WRITE_SIZE_KiB": write.get(k, 0.0)}


def total(pred):
    return int(sum((2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024 for k, v in kern.items() if pred(k)))


# the first backward form pam_bench runs is the default one (K64 + atomics): its kernels are pam_bwd_k64_kernel<6, false, 2, true, ORDER>,
# pam_rowconst_kernel, pam_dq_transpose_kernel (+ the hipMemset of the dQ accumulator, not a kernel)
B, N, C, r = 32, 65536, 184, 23
Np, Cp = N, 192
alg_fwd = B * ((Np * 32 * 2) * 2 + Cp * Np * 2 + C * N * 4 * 3 + N * 4)             # qt, kt, v packs + x read + out, o_attn written + lse
alg_bwd = B * ((Np * 32 * 2) * 3 + Np * Cp * 2 * 2 + N * 4 * 2 + 32 * Np * 4 * 2 + Cp * Np * 4)   # qt, kt, kn, vt, dot + lse, delta + dq, dk, dv
out = {
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/pam_bench.py "
           "--batch 32 --iters 1; KiB per dispatch averaged over the run's dispatches; FETCH doubled (gfx950 wide-stream "
           "correction), WRITE as is; Infinity-Cache hits are included in FETCH_SIZE",
    "commit": sys.argv[3] if len(sys.argv) > 3 else None,
    "shape": {"B": B, "N": N, "C": C, "r": r},
    "kernels_per_dispatch": kern,
    "traffic_bytes_per_launch": {
        "pam_flash_fwd": total(lambda k: k.startswith("pam_fwd_dma_kernel") and "false>" in k.replace(" ", "")),
        "pam_flash_bwd": total(lambda k: re.fullmatch(r"pam_bwd_k64_kernel<6,false,2,true,\d+>", k.replace(" ", "")) is not None
                               or k.startswith("pam_rowconst") or k.startswith("pam_dq_transpose")),
    },
    "algorithmic_bytes_per_launch": {"pam_flash_fwd": alg_fwd, "pam_flash_bwd": alg_bwd,
                                     "note": "every operand read once, every result written once"},
}
print(json.dumps(out, indent=1))
