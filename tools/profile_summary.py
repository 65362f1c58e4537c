"""Turn a rocprofv3 output directory into the per-kernel summaries kept under profiles/.

    python tools/profile_summary.py stats <rocprof dir> [--top 40]     kernel-trace CSV -> name, calls, total, average
    python tools/profile_summary.py pmc   <rocprof dir>                counter CSVs    -> per-kernel per-dispatch means

Recipes (on the GPU box; `cd /tmp && export TMPDIR=/tmp` is not needed under gpurun):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 4 --warmup 2
    rocprofv3 -i tools/pmc_r02.txt --kernel-trace --output-format csv -d gpurun_out/prof_pmc -- python3 tools/pam_bench.py --batch 2 --iters 1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 tools/pam_bench.py --batch 2 --iters 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 tools/pam_bench.py --batch 2 --iters 1
Counter notes (MI355X_MICROARCH.md): SQ_* wave counters are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and
SQ_BUSY_CYCLES are cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE (KiB) reads HALF the bytes of a wide
coalesced stream on gfx950 -- the summary prints it doubled next to the raw value."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name[:80]


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def stats(d, top):
    """rows are keyed by (kernel, grid): launches of one kernel at different shapes (bench.py also runs the 16 x 16
    parity fixture through the generator once) stay apart, so "avg ms" is an average over launches of ONE shape"""
    rows = defaultdict(lambda: [0, 0.0])
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            grid = "x".join(str(int(r[k]) // max(1, int(r[k.replace("Grid", "Workgroup")])))
                            for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r)
            n = (short(r["Kernel_Name"]), grid)
            rows[n][0] += 1
            rows[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot = sum(v[1] for v in rows.values())
    print(f"{'kernel':64s} {'workgroups':>14s} {'calls':>6s} {'total ms':>10s} {'avg ms':>9s} {'%':>6s}")
    for (n, g), (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{n[:64]:64s} {g:>14s} {c:6d} {t:10.3f} {t / c:9.4f} {100 * t / tot:6.2f}")
    print(f"{'TOTAL':64s} {'':>14s} {sum(v[0] for v in rows.values()):6d} {tot:10.3f}")


PREFIX = "pam_"          # kernels summarised by the pmc mode (third argument overrides)


def pmc(d):
    # counter_collection.csv: one row per dispatch and counter
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in sorted(vals, key=lambda k: -sum(dur.get(k, [0]))):
        if not k.startswith(PREFIX):
            continue
        c = {n: sum(v) / len(v) for n, v in vals[k].items()}
        ms = sum(dur[k]) / max(1, len(dur[k]))
        print(f"\n{k}   ({len(dur[k])} dispatches under the profiler, avg {ms:.3f} ms)")
        wc = c.get("SQ_WAVE_CYCLES")
        for n in sorted(c):
            extra = ""
            if wc and n.startswith("SQ_") and n not in ("SQ_WAVE_CYCLES", "SQ_WAVES"):
                extra = f"   ({c[n] / wc:.3f} of SQ_WAVE_CYCLES)"
            if n == "FETCH_SIZE":
                extra = f"   KiB raw; x2 (gfx950 wide-stream correction) = {2 * c[n] * 1024 / 1e9:.2f} GB per dispatch"
            if n == "WRITE_SIZE":
                extra = f"   KiB = {c[n] * 1024 / 1e9:.2f} GB per dispatch"
            print(f"    {n:32s} {c[n]:14.4g}{extra}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and wc:
            # waves are resident for the whole kernel: SIMD-cycles = 4 (quad) * SQ_WAVE_CYCLES / (waves sharing a SIMD)
            wps = 1 if "k64" in k else 2
            print(f"    -> matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc / wps):.3f} of the SIMD-cycles "
                  f"({wps} wave(s) per SIMD)")
        if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) > 0 and "SQ_INSTS_VALU" in c:
            mf = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / 64      # a 32x32x16 bf16 MFMA counts 64 MOPS units (512 flop each)
            print(f"    -> MFMA instructions {mf:.4g}; VALU instructions per MFMA {c['SQ_INSTS_VALU'] / mf - 1:.2f}")
        if "GRBM_GUI_ACTIVE" in c and ms > 0:
            print(f"    -> clock under load ~ {c['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3) / 1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / time)")


if __name__ == "__main__":
    mode, d = sys.argv[1], sys.argv[2]
    if mode == "stats":
        stats(d, int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--top" else 40)
    else:
        if len(sys.argv) > 3:
            PREFIX = sys.argv[3]
        pmc(d)
