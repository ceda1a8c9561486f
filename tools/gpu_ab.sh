#!/bin/bash
# A/B two builds of the library inside ONE gpurun call: ab/lib_base.so vs the in-tree build (boxes differ by +-10 %).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in base new base new; do
  if [ $v = base ]; then export HG_LIB_PATH=$R/ab/lib_base.so; else unset HG_LIB_PATH; fi
  rm -rf $R/gpurun_out/pm; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  echo "$v: $(python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_ring' in r['Name'] or 'attention' in r['Name']: print(r['Name'].replace('void hg::','')[:16], r['Calls'], 'avg=%.0fus |'%(float(r['AverageNs'])/1e3), end=' ')
PY
) $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/pm.log)"
done
