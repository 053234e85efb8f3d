#!/bin/bash
# Round-3 rocprofv3 evidence, captured on the GPU box (from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/capture_r03.sh'
# One rocprofv3 run per counter group (never --pmc together with API / sys traces); the program follows "--" directly.
# Outputs land in gpurun_out/prof_r03/; profiles/summarize_r03.py turns them into the small files committed under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r03
rm -rf "$OUT"; mkdir -p "$OUT"
COMMON="--cpu-sample 0 --extras 0 --other-paths 0"
run_cfg() {   # tag, bench args
    local tag=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag.stats" -o r03 -- python3 bench.py --steps 2 --warmup 1 $COMMON "$@" > "$OUT/$tag.stats.log" 2>&1
    for grp in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag.$grp" -o r03 -- python3 bench.py --steps 1 --warmup 0 $COMMON "$@" > "$OUT/$tag.$grp.log" 2>&1
    done
    grep "^{\"metric\"" "$OUT/$tag.stats.log" > "$OUT/r03_${tag}_bench_under_rocprof.json"
}
run_cfg c2 --batch 2048                                 # headline: 2048 x 4096x256, exact policy
run_cfg c4 --mrows 2048 --ncols 128 --batch 1024        # config 4: 1024 x 2048x128, exact policy
run_cfg c5 --mrows 65536 --ncols 512 --batch 1          # config 5: one 65536x512 problem, exact policy
run_cfg c5auto --mrows 65536 --ncols 512 --batch 1 --policy 0   # config 5, normal-equations policy: FD column + MFMA J^T J kernels
run_cfg c2auto --batch 2048 --policy 0                  # the opt-in fast policy at the headline shape
# FETCH_SIZE calibration for the 8-byte-per-lane streaming pattern of k_qrx_pass (known byte count)
( cd profiles/ubench && hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip > /dev/null 2>&1 )
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib" -o r03 -- profiles/ubench/fetch_calib > "$OUT/calib.log" 2>&1
# memory-system, barrier and serial-chain microbenchmarks quoted in DESIGN.md (store shapes, mixed read/write streams,
# grid barriers, the ordered-sum chain)
( cd profiles/ubench && for f in rw_stream w_stream grid_sync chain_waves lds_launch; do hipcc -O3 --offload-arch=gfx950 -o $f $f.hip > /dev/null 2>&1; done
  { echo "== rw_stream 1024"; timeout 120 ./rw_stream 1024 0; echo "== w_stream 1024"; timeout 120 ./w_stream 1024;
    echo "== grid_sync"; timeout 60 ./grid_sync 257 256; timeout 60 ./grid_sync 64 256;
    echo "== chain_waves"; timeout 60 ./chain_waves; echo "== lds_launch"; timeout 60 ./lds_launch; } > "$OUT/r03_ubench.txt" 2>&1 )
python3 profiles/summarize_r03.py "$OUT" > "$OUT/summarize.log" 2>&1
tail -5 "$OUT/summarize.log"
# keep only what is small enough to be merged back
find "$OUT" -name "*.csv" -size +1500k -delete
find "$OUT" -type d -name "*.stats" -prune -o -type d -name "*.FETCH_SIZE" -prune -o -type d -name "*.WRITE_SIZE" -prune
rm -rf "$OUT"/*.stats "$OUT"/*.FETCH_SIZE "$OUT"/*.WRITE_SIZE "$OUT"/calib
ls -la "$OUT"
