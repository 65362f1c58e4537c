#!/bin/bash
# Reproduce the profiles/rNN_* summaries from the current head on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/run_profiles.sh r02'
# 1. rocprofv3 --kernel-trace --stats of the default bench command            -> profiles/<tag>_bench_kernel_stats.txt
# 2. SQ / GRBM counters of the PAM kernels (separate --pmc passes)            -> profiles/<tag>_pam_pmc.txt
# 3. FETCH_SIZE and WRITE_SIZE of the PAM kernels at the bench launch shape   -> profiles/<tag>_pam_traffic.json
# Counter passes never combine --pmc with the sys/hip/hsa trace domains (kernel-trace only).
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
COMMIT=$(cat .git_head 2>/dev/null || git rev-parse --short HEAD 2>/dev/null || echo unknown)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_profiler.log 2>&1
python3 tools/profile_summary.py stats $OUT/stats --top 60 > profiles/${TAG}_bench_kernel_stats.txt
grep "^{" $OUT/bench_under_profiler.log | tail -1 > profiles/${TAG}_bench_under_profiler.json.log || true
rocprofv3 -i tools/pmc_r02.txt --kernel-trace --output-format csv -d $OUT/pmc -- python3 tools/pam_bench.py --batch 2 --iters 1 > $OUT/pmc.log 2>&1
{ echo "rocprofv3 -i tools/pmc_r02.txt --kernel-trace --output-format csv -- python3 tools/pam_bench.py --batch 2 --iters 1"
  echo "(MI355X, commit $COMMIT; C=184, N=65536, B=2; per-dispatch means; SQ_* wave counters in quad-cycles summed over waves,"
  echo " SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES in cycles; MFMA busy share of the SIMDs = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES / waves per SIMD))"
  python3 tools/profile_summary.py pmc $OUT/pmc; } > profiles/${TAG}_pam_pmc.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 tools/pam_bench.py --batch 32 --iters 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 tools/pam_bench.py --batch 32 --iters 1 > $OUT/write.log 2>&1
python3 tools/traffic_summary.py $OUT/fetch $OUT/write "$COMMIT" > profiles/${TAG}_pam_traffic.json
echo "profiles written for $TAG at $COMMIT"
