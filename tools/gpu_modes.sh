#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for g in 0 12 1 3 4 6 0; do
  rm -rf $R/gpurun_out/pm; HG_RING_GSZ=$g rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  echo "gsz $g: $(python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_ring' in r['Name']: print(r['Name'][10:24], r['Calls'], 'avg=%.0fus'%(float(r['AverageNs'])/1e3), 'min=%.0f'%(float(r['MinNs'])/1e3),'max=%.0f |'%(float(r['MaxNs'])/1e3), end=' ')
PY
) $(tail -1 $R/gpurun_out/pm.log | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")"
done
