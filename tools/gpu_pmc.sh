#!/bin/bash
# PMC counter passes over a short bench run (separate passes; --pmc only with --kernel-trace).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf $R/gpurun_out/pmc_$name; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/pmc_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R
python3 - <<'PY'
import csv,glob,collections
for name in ['sq','fetch','write','tcc']:
    fs=glob.glob(f'gpurun_out/pmc_{name}/**/*counter_collection.csv',recursive=True)
    if not fs: print(name,'no file'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k=row['Kernel_Name'][:60]; agg[k][row['Counter_Name']]+=float(row['Counter_Value']);
        if row['Counter_Name']==list(agg[k].keys())[0]: cnt[k]+=1
    for k in sorted(agg,key=lambda k:-sum(agg[k].values()))[:7]:
        print(name,k,cnt[k],{c:round(v/max(cnt[k],1),1) for c,v in agg[k].items()})
PY
