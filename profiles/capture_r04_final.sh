#!/bin/bash
# end-of-round capture with the final library: default bench line, headline + C5 + lone problem under rocprofv3 --stats, mid sweep, other paths
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
G=$PWD/gpurun_out/final4; rm -rf "$G"; mkdir -p "$G"
python bench.py --steps 20 --warmup 5 2>"$G/bench_default.err" | tail -1 > "$G/r04_bench_default.json"
stats() {  # tag, program args...
  local tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/$tag.stats" -o r04 -- "$@" > "$G/$tag.log" 2>&1
  f=$(find "$G/$tag.stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$G/r04_${tag}_kernel_stats.csv"
  grep "^{\"metric\"" "$G/$tag.log" > "$G/r04_${tag}_bench_under_rocprof.json"
  rm -rf "$G/$tag.stats"
}
stats c2 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --extras 0 --other-paths 0 --batch 2048
stats c5_final python3 bench.py --steps 2 --warmup 1 --mrows 65536 --ncols 512 --batch 1 --cpu-sample 0 --extras 0 --other-paths 0
stats lone_4096x256 python3 profiles/sweep_mid.py 4096x256:1
stats other_paths python3 bench.py --steps 1 --warmup 0 --batch 16 --m 1024 --n 64 --cpu-sample 0 --extras 0
timeout 600 python3 profiles/sweep_mid.py 4096x256:1,4,8,16,32,47,64,128,256 2048x128:1,8,32,64,128,256,512,1024 > "$G/r04_sweep_mid.txt" 2>&1
find "$G" -name "*.csv" -size +1500k -delete
ls -la "$G"
