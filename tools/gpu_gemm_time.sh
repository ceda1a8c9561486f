#!/bin/bash
# kernel durations of tools/gemm_time.py for several settings in one gpurun call: CONFIGS="name:ENV=V,ENV=V ..."
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for kv in ${envs//,/ }; do export $kv; done
  rm -rf $R/gpurun_out/gt; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gt -- python3 $R/tools/gemm_time.py > $R/gpurun_out/gt.log 2>&1
  f=$(find $R/gpurun_out/gt -name "*kernel_trace.csv" | head -1)
  echo "$name: $(python3 - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_ring' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
n=len(d)//2 if len(d)>=2 else len(d)
import re
nm=lambda r: re.sub(r'.*(gemm_ring2?<[^>]*>).*',r'\1',r['Kernel_Name'])
if d: print('%s first-shape %s us | %s second-shape %s us'%(nm(rows[0]), ' '.join('%.0f'%x for x in d[1:n]), nm(rows[-1]), ' '.join('%.0f'%x for x in d[n+1:])))
PY
)"
  for kv in ${envs//,/ }; do unset ${kv%%=*}; done
done
