#!/bin/bash
# attention kernel durations for several builds in one gpurun call: LIBS="base new" -> ab/lib_<name>.so (tree = in-tree)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in $LIBS $LIBS; do
  if [ $v = tree ]; then unset HG_LIB_PATH; else export HG_LIB_PATH=$R/ab/lib_$v.so; fi
  rm -rf $R/gpurun_out/at; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/at -- python3 $R/tools/attn_time.py > $R/gpurun_out/at.log 2>&1
  f=$(find $R/gpurun_out/at -name "*kernel_trace.csv" | head -1)
  echo "$v: $(python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'attention_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(' '.join('%.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows[2:]), 'us')
PY
)"
done
