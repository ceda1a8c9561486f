#!/bin/bash
# time and HBM traffic of the c_fc GEMM for several tile-group sizes (HG_RING_GSZ), inside one gpurun call
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for g in ${VALS:-0 6 12}; do
  export HG_RING_GSZ=$g
  rm -rf $R/gpurun_out/pm; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  t=$(python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_ring<4, 9' in r['Name'] or 'gemm_ring<4, 8' in r['Name']: print(r['Name'][15:20], '%.0fus'%(float(r['AverageNs'])/1e3), end=' ')
PY
)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_x; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python3 - "$c" $R/gpurun_out/pmc_x <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[2]+'/**/*counter_collection.csv',recursive=True)[0]
tot={};cnt={}
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    for key in ('gemm_ring<4, 9','gemm_ring<4, 8'):
        if key in k:
            tot[key]=tot.get(key,0)+float(r['Counter_Value']); cnt[key]=cnt.get(key,0)+1
for key in tot: print(sys.argv[1], key[-2:], '%.0f MB'%(tot[key]/cnt[key]*1024*(2 if sys.argv[1]=='FETCH_SIZE' else 1)/1e6), end=' | ')
PY
  done
  echo " gsz=$g $t $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/pm.log)"
done
