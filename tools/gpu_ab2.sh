#!/bin/bash
# A/B several builds inside ONE gpurun call: LIBS="base dmam" -> ab/lib_<name>.so ("tree" = the in-tree build).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in $LIBS $LIBS; do
  if [ $v = tree ]; then unset HG_LIB_PATH; else export HG_LIB_PATH=$R/ab/lib_$v.so; fi
  rm -rf $R/gpurun_out/pm; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  echo "$v: $(python3 - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'gemm_ring' in n or 'attention' in n:
        m=re.search(r'(gemm_ring2?|attention\w*)<([^>]*)>',n)
        print((m.group(1)+'<'+m.group(2)+'>') if m else n[:30], r['Calls'], '%.0fus |'%(float(r['AverageNs'])/1e3), end=' ')
PY
) $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/pm.log)"
done
