#!/bin/bash
# Capture the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/capture.sh r01'
# One rocprofv3 run per counter group (never --pmc together with API/sys traces); the program follows
# "--" directly.  Outputs land in gpurun_out/prof_<tag>/ and are copied into profiles/ by hand afterwards
# (profiles/make_summary.py turns the counter CSVs into <tag>_pmc_summary.json).
set -u
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --exact-sample 0 --other-paths 0 --extras 0"
ONE="python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --exact-sample 0 --other-paths 0 --extras 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- $BENCH > "$OUT/stats.log" 2>&1
for grp in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$grp" -o "$TAG" -- $ONE > "$OUT/$grp.log" 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$OUT/SQ" -o "$TAG" -- $ONE > "$OUT/SQ.log" 2>&1
# the exact factor policy (reference operation order), 256 problems: per-kernel times of k_qr_exact_lazy / k_lmpar<true>
EX="python3 bench.py --steps 1 --warmup 1 --batch 256 --policy 2 --cpu-sample 0 --exact-sample 0 --other-paths 0 --extras 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_exact" -o "$TAG" -- $EX > "$OUT/stats_exact.log" 2>&1
f=$(find "$OUT/stats_exact" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_exact_kernel_stats.csv"
grep "^{\"metric\"" "$OUT/stats_exact.log" > "$OUT/${TAG}_exact_bench_under_rocprof.json"
# flatten: keep only the small CSVs the judge reads
for grp in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ; do
    f=$(find "$OUT/$grp" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/${TAG}_pmc_${grp}_counter_collection.csv"
done
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_bench_kernel_stats.csv"
grep "^{\"metric\"" "$OUT/stats.log" > "$OUT/${TAG}_bench_under_rocprof.json"
rm -rf "$OUT/stats" "$OUT/stats_exact" "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" "$OUT/GRBM_GUI_ACTIVE" "$OUT/SQ"
ls -la "$OUT"
