#!/bin/bash
# Round-6 rocprofv3 evidence, captured on the GPU box (from the repo root):
#   git rev-parse --short HEAD > profiles/.capture_commit        (build container: the GPU box has no git)
#   gpurun --timeout 3000 -- 'bash profiles/capture_r06.sh'
# One rocprofv3 run per counter group (never --pmc together with API / sys traces); the program follows "--" directly.
# Outputs land in gpurun_out/prof_r06/; profiles/summarize_r06.py turns them into the small files committed under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r06
rm -rf "$OUT"; mkdir -p "$OUT"
COMMON="--cpu-sample 0 --extras 0 --other-paths 0"
stats() {   # tag, program + args: kernel stats of one command
    local tag=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag.stats" -o r06 -- "$@" > "$OUT/$tag.stats.log" 2>&1
    local f; f=$(find "$OUT/$tag.stats" -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" "$OUT/r06_${tag}_kernel_stats.csv"; else echo "capture_r06: no kernel stats for $tag (see $OUT/$tag.stats.log)" >&2; fi
    grep "^{\"metric\"" "$OUT/$tag.stats.log" > "$OUT/r06_${tag}_bench_under_rocprof.json" || true
}
pmc() {     # tag, bench args: FETCH_SIZE / WRITE_SIZE, one pass each
    local tag=$1; shift
    for grp in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag.$grp" -o r06 -- python3 bench.py --steps 1 --warmup 0 $COMMON "$@" > "$OUT/$tag.$grp.log" 2>&1
    done
}
# 1. the default bench line, no profiler (what the driver runs)
python3 bench.py > "$OUT/r06_bench_default.json" 2> "$OUT/bench_default.err"
# 2. per-kernel stats + HBM counters of the BASELINE configs
stats c2 python3 bench.py --steps 2 --warmup 1 $COMMON --batch 2048;                       pmc c2 --batch 2048
stats c4 python3 bench.py --steps 2 --warmup 1 $COMMON --mrows 2048 --ncols 128 --batch 1024; pmc c4 --mrows 2048 --ncols 128 --batch 1024
stats c5 python3 bench.py --steps 2 --warmup 1 $COMMON --mrows 65536 --ncols 512 --batch 1
stats c5auto python3 bench.py --steps 2 --warmup 1 $COMMON --mrows 65536 --ncols 512 --batch 1 --policy 0; pmc c5auto --mrows 65536 --ncols 512 --batch 1 --policy 0
stats c2auto python3 bench.py --steps 2 --warmup 1 $COMMON --batch 2048 --policy 0
# config 5 under the normal-equations policy WITHOUT the profiler, with the per-kernel breakdown (kernel_rooflines: Gram vs the fp64 MFMA peak)
timeout 600 python3 bench.py --steps 3 --warmup 1 --mrows 65536 --ncols 512 --batch 1 --policy 0 --cpu-sample 0 --other-paths 0 > "$OUT/r06_c5auto_bench_extras.json" 2> "$OUT/c5auto_extras.err"
# MFMA-busy and effective clock of the Gram kernel (its own pass: counters only)
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d "$OUT/c5auto.MFMA" -o r06 -- python3 bench.py --steps 2 --warmup 1 $COMMON --mrows 65536 --ncols 512 --batch 1 --policy 0 > "$OUT/c5auto.MFMA.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
f = glob.glob(os.path.join(out, "c5auto.MFMA", "**", "*counter_collection.csv"), recursive=True)
t = glob.glob(os.path.join(out, "c5auto.MFMA", "**", "*kernel_trace.csv"), recursive=True)
if f and t:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_gram_512" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t[0])) if "k_gram_512" in r["Kernel_Name"]]
    d = {k: sum(v) / len(v) for k, v in agg.items()}
    us = sum(dur) / len(dur)
    mfmas = 64 * (2 * 8 * 4352 + 2 * 8 * 4096)                       # 64 K-splits x (2 triangle units + 2 square units) x 8 waves
    res = {"kernel": "k_gram_512, one 65536 x 512 problem", "launches": len(dur), "avg_us": us, "counters_avg_per_launch": d,
           "mfma_instructions": mfmas, "cycles_per_mfma": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / mfmas,
           "effective_clock_GHz": d.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3,
           "mfma_busy_frac_of_kernel": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(d.get("GRBM_GUI_ACTIVE", 1) / 8, 1),
           "note": "SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs (MI355X_MICROARCH.md, DVFS)"}
    json.dump(res, open(os.path.join(out, "r06_gram_mfma_busy.json"), "w"), indent=1)
PY
rm -rf "$OUT/c5auto.MFMA"
# 3. the open device-residual path: kernel stats of the device_vecfcn rows (k_fd_jacobian_qrx inside the solves)
stats devfcn python3 profiles/scripts/devfcn_time.py
# 4. mid regime: solve times between a handful and a chipful (46 / 47: the first six-iteration problem), kernel shares at 47 and 128
timeout 900 python3 profiles/sweep_mid.py 4096x256:1,4,8,16,32,46,47,64,128,256 2048x128:1,8,32,64,128,256,512,1024 > "$OUT/r06_sweep_mid.txt" 2>&1
# ... and with the library's automatic sub-batches (what nlh_default_options gives a caller: two halves from 32 problems on)
timeout 900 python3 profiles/sweep_mid.py 4096x256:16,32,46,47,64,96,128,192,256 2048x128:32,64,128,192,256,512,1024 --sub=0 >> "$OUT/r06_sweep_mid.txt" 2>&1
stats mid_4096x256_47 python3 profiles/sweep_mid.py 4096x256:47
stats mid_2048x128_128 python3 profiles/sweep_mid.py 2048x128:128
stats lone_4096x256 python3 profiles/sweep_mid.py 4096x256:1
stats mid_2048x128_256 python3 profiles/sweep_mid.py 2048x128:256 --sub=0
stats mid_2048x128_512 python3 profiles/sweep_mid.py 2048x128:512 --sub=0
# 4b. how much of the default-options headline has a trailing pass resident (3 sub-batches on private streams)
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$OUT/overlap.trace" -o r06 -- python3 profiles/scripts/defaults_one.py 2048 4096 256 1 0 > "$OUT/overlap.log" 2>&1
f=$(find "$OUT/overlap.trace" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 profiles/scripts/overlap_defaults.py "$f" "$OUT/r06_overlap_defaults.json" > /dev/null
rm -rf "$OUT/overlap.trace"
# 4c. N1: the MFMA / Cholesky policy on the zero-residual variant, kernel stats of that solve
stats c2auto_zero_residual python3 profiles/scripts/n1_zero_residual.py --c5 0
# 5. the other solvers' kernels (Newton, LU n = 1024, bounded least squares, BFGS, polynomial fits, mode H)
stats other_paths python3 bench.py --steps 1 --warmup 0 --batch 16 --m 1024 --n 64 --cpu-sample 0 --extras 0
stats lu_n1024 python3 profiles/lu_time.py 1024
# FETCH_SIZE calibration for the 8-byte-per-lane streaming pattern of k_qrx_pass (known byte count)
( cd profiles/ubench && hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip > "$OUT/fetch_calib.build.log" 2>&1 )
if [ -x profiles/ubench/fetch_calib ]; then
    timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib" -o r06 -- profiles/ubench/fetch_calib > "$OUT/calib.log" 2>&1
else
    echo "capture_r06: profiles/ubench/fetch_calib did not build (see $OUT/fetch_calib.build.log): no calibration pass" >&2
fi
python3 profiles/summarize_r06.py "$OUT" > "$OUT/summarize.log" 2>&1
tail -5 "$OUT/summarize.log"
# keep only what is small enough to be merged back
find "$OUT" -name "*.csv" -size +1500k -delete
rm -rf "$OUT"/*.stats "$OUT"/*.FETCH_SIZE "$OUT"/*.WRITE_SIZE "$OUT"/calib
ls -la "$OUT"
