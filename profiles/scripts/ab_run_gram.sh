cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
for tag in base NOLOAD NOLOADG ALL; do
  cp scratch/ab/lib_gram_$tag.so nonlin_amd/libnonlin_hip.so
  OUT=/tmp/gp_$tag; rm -rf $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o g -- python3 bench.py --steps 2 --warmup 1 --mrows 65536 --ncols 512 --batch 1 --policy 0 --cpu-sample 0 --other-paths 0 --extras 0 > /tmp/gp_$tag.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'gram_512' in r['Name']: print('== $tag k_gram_512 avg us', float(r['AverageNs'])/1e3, 'calls', r['Calls'])
"
done
cp scratch/ab/lib_gram_base.so nonlin_amd/libnonlin_hip.so
