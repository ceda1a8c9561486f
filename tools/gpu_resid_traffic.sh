#!/bin/bash
# HBM-side traffic (FETCH_SIZE x2, WRITE_SIZE; MB per launch) of the residual GEMMs, out-proj (K=768) and c_proj
# (K=3072) separately, for several launcher settings inside one gpurun call.  CONFIGS="name:ENV=V,ENV=V ..."
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in ${CONFIGS:-"ring2:HG_RING_SK=0" "big:HG_RING_SK=0,HG_RING_BIG=1" "sk:HG_RING_SK=1"}; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for kv in ${envs//,/ }; do export $kv; done
  out="$name:"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_x; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    out="$out $(python3 - "$c" $R/gpurun_out/pmc_x <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[2]+'/**/*counter_collection.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if ('gemm_ring2<10' in r['Kernel_Name'] or 'gemm_ring<4, 10' in r['Kernel_Name'])]
rows.sort(key=lambda r:int(r['Dispatch_Id']))
mul=1024*(2 if sys.argv[1]=='FETCH_SIZE' else 1)/1e6
ev=[float(r['Counter_Value']) for r in rows[0::2]]; od=[float(r['Counter_Value']) for r in rows[1::2]]
print('%s out-proj %.0f MB c_proj %.0f MB (n=%d)'%(sys.argv[1], sum(ev)/max(len(ev),1)*mul, sum(od)/max(len(od),1)*mul, len(rows)), end='')
PY
)"
  done
  echo "$out"
  for kv in ${envs//,/ }; do unset ${kv%%=*}; done
done
