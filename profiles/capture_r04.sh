#!/bin/bash
# Round-4 rocprofv3 evidence, captured on the GPU box (from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/capture_r04.sh'
# One rocprofv3 run per counter group (never --pmc together with API / sys traces); the program follows "--" directly.
# Outputs land in gpurun_out/prof_r04/; profiles/summarize_r04.py turns them into the small files committed under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r04
rm -rf "$OUT"; mkdir -p "$OUT"
COMMON="--cpu-sample 0 --extras 0 --other-paths 0"
run_cfg() {   # tag, bench args
    local tag=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag.stats" -o r04 -- python3 bench.py --steps 2 --warmup 1 $COMMON "$@" > "$OUT/$tag.stats.log" 2>&1
    for grp in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag.$grp" -o r04 -- python3 bench.py --steps 1 --warmup 0 $COMMON "$@" > "$OUT/$tag.$grp.log" 2>&1
    done
    grep "^{\"metric\"" "$OUT/$tag.stats.log" > "$OUT/r04_${tag}_bench_under_rocprof.json"
}
run_cfg c2 --batch 2048                                 # headline: 2048 x 4096x256, exact policy
run_cfg c4 --mrows 2048 --ncols 128 --batch 1024        # config 4: 1024 x 2048x128, exact policy
run_cfg c5 --mrows 65536 --ncols 512 --batch 1          # config 5: one 65536x512 problem, exact policy
run_cfg c5auto --mrows 65536 --ncols 512 --batch 1 --policy 0   # config 5, normal-equations policy: FD column + MFMA J^T J kernels
run_cfg c2auto --batch 2048 --policy 0                  # the opt-in fast policy at the headline shape
# config 5 under the normal-equations policy once more WITHOUT the profiler and with the per-kernel breakdown: the line's
# kernel_rooflines (Gram against the fp64 MFMA peak, HIP events) is what DESIGN.md quotes next to the rocprofv3 averages
timeout 600 python3 bench.py --steps 3 --warmup 1 --mrows 65536 --ncols 512 --batch 1 --policy 0 --cpu-sample 0 --other-paths 0 > "$OUT/r04_c5auto_bench_extras.json" 2> "$OUT/c5auto_extras.err"
# FETCH_SIZE calibration for the 8-byte-per-lane streaming pattern of k_qrx_pass (known byte count)
( cd profiles/ubench && hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip > /dev/null 2>&1 )
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib" -o r04 -- profiles/ubench/fetch_calib > "$OUT/calib.log" 2>&1
# memory-system, barrier and serial-chain microbenchmarks quoted in DESIGN.md (store shapes, mixed read/write streams,
# grid barriers, the ordered-sum chain)
( cd profiles/ubench && for f in rw_stream rw_stagger tcp_pattern mfma_f64; do hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o $f $f.hip > /dev/null 2>&1; done
  { echo "== rw_stream 1024"; timeout 120 ./rw_stream 1024 0; echo "== rw_stagger 1024 (round 4: does spreading the flush over the steps pay?)"; timeout 120 ./rw_stagger 1024;
    echo "== tcp_pattern (round 4: what one CU pulls through its vector memory path, lane-per-column against whole sectors per lane quad)"; timeout 120 ./tcp_pattern;
    echo "== mfma_f64 (the fp64 MFMA ceiling the Gram kernels are priced against)"; timeout 60 ./mfma_f64; } > "$OUT/r04_ubench.txt" 2>&1 )
# mid regime (round 4): solve times between a handful and a chipful, and the kernel shares at 32 x 4096x256 / 128 x 2048x128
timeout 600 python3 profiles/sweep_mid.py 4096x256:1,8,16,32,47,64,128,256 2048x128:8,32,64,128,256,512,1024 > "$OUT/r04_sweep_mid.txt" 2>&1
for spec in 4096x256:32 2048x128:128; do
    tag=${spec/:/_}
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mid_$tag.stats" -o r04 -- python3 profiles/sweep_mid.py $spec > "$OUT/mid_$tag.log" 2>&1
    f=$(find "$OUT/mid_$tag.stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r04_mid_${tag}_kernel_stats.csv"
done
python3 profiles/summarize_r04.py "$OUT" > "$OUT/summarize.log" 2>&1
tail -5 "$OUT/summarize.log"
# keep only what is small enough to be merged back
find "$OUT" -name "*.csv" -size +1500k -delete
find "$OUT" -type d -name "*.stats" -prune -o -type d -name "*.FETCH_SIZE" -prune -o -type d -name "*.WRITE_SIZE" -prune
rm -rf "$OUT"/*.stats "$OUT"/*.FETCH_SIZE "$OUT"/*.WRITE_SIZE "$OUT"/calib "$OUT"/mid_*.stats
ls -la "$OUT"
# (added later in round 4) the other solvers' kernels after the latency work on LU, the Householder step, the BFGS pieces and
# the triangular solves: one bench pass with only the other_paths rows, per-kernel stats; the LU of n = 1024 alone
OUT=$PWD/gpurun_out/prof_r04; mkdir -p "$OUT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/other.stats" -o r04 -- python3 bench.py --steps 1 --warmup 0 --batch 16 --m 1024 --n 64 --cpu-sample 0 --extras 0 > "$OUT/other.stats.log" 2>&1
f=$(find "$OUT/other.stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r04_other_paths_kernel_stats.csv"
rm -rf "$OUT/other.stats"
