#!/bin/bash
# FETCH_SIZE (x2, MB per launch) of the residual / QKV / c_fc GEMMs for several env settings: CONFIGS="name:ENV=V,ENV=V ..."
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export $kv; done; fi
  rm -rf $R/gpurun_out/pmc_x; rocprofv3 --kernel-trace --pmc ${PMC:-FETCH_SIZE} --output-format csv -d $R/gpurun_out/pmc_x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-class-rows > $R/gpurun_out/pmc_x.log 2>&1
  python3 - "$name" $R/gpurun_out/pmc_x <<'PY'
import csv,glob,sys,collections,re
fs=glob.glob(sys.argv[2]+'/**/*counter_collection.csv',recursive=True)
if not fs: print(sys.argv[1],'no csv'); sys.exit()
rows=[r for r in csv.DictReader(open(fs[0])) if 'gemm_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Dispatch_Id']))
agg=collections.defaultdict(list)
prev={}
for r in rows:
    k=re.sub(r'.*(gemm_\w+<[^>]*>).*',r'\1',r['Kernel_Name'])
    # residual kernels alternate out_proj / c_proj: split by grid-independent parity of occurrence
    if '<10, 2>' in r['Kernel_Name']:      # hi / lo stream: the 21 middle launches of a step alternate c_proj, out_proj, c_proj, ...
        n=prev.get(k,0); prev[k]=n+1
        k+= ' c_proj' if (n%21)%2==0 else ' out_proj'
    elif '10>' in k or '<10, 0>' in r['Kernel_Name']:
        n=prev.get(k,0); prev[k]=n+1
        k+= ' out_proj' if (n%23)%2==0 else ' c_proj'   # 12 out_proj + 11 c_proj per step, alternating
    agg[k+' '+r['Counter_Name']].append(float(r['Counter_Value']))
mul=lambda c: 1024*(2 if c=='FETCH_SIZE' else 1)/1e6
print(sys.argv[1], ' | '.join('%s %.0f MB (n=%d)'%(k, sum(v)/len(v)*mul(k.split()[-1]), len(v)) for k,v in sorted(agg.items())))
PY
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do unset ${kv%%=*}; done; fi
done
