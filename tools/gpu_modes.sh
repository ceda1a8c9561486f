#!/bin/bash
# A/B sweep of one environment knob inside ONE gpurun call (boxes differ by +-10 %):
#   VAR=HG_RING_GSZ VALS="0 3 6" bash tools/gpu_modes.sh
# (HG_RING_MODE experiment bits need a build with HG_EXTRA_FLAGS=-DHG_EXPERIMENTS)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
VAR=${VAR:-HG_RING_MODE}
for g in ${VALS:-0 4 0}; do
  export $VAR=$g
  rm -rf $R/gpurun_out/pm; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  echo "$VAR=$g: $(python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_ring' in r['Name']: print(r['Name'].replace('void hg::','')[:16], r['Calls'], 'avg=%.0fus |'%(float(r['AverageNs'])/1e3), end=' ')
PY
) $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/pm.log)"
  python3 $R/tools/trace_layer.py
done
