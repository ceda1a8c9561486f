#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "0 0" "0 3" "0 0" "0 3"; do
  set -- $cfg
  rm -rf $R/gpurun_out/pm; HG_RING_MODE=$1 HG_RING_BIG=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  echo "mode $1 big $2: $(python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_ring' in r['Name']: print(r['Name'][10:24], r['Calls'], 'avg=%.0fus'%(float(r['AverageNs'])/1e3), 'min=%.0f'%(float(r['MinNs'])/1e3),'max=%.0f |'%(float(r['MaxNs'])/1e3), end=' ')
PY
)"
done
