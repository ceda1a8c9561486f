#!/bin/bash
# SQ counters of the attention kernel (one pass per counter group; --pmc only with --kernel-trace)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf $R/gpurun_out/pmc_$name; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$name.log 2>&1; }
run a1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run a2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run a3 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM
cd $R
python3 - <<'PY'
import csv,glob,collections
for name in ['a1','a2','a3']:
    fs=glob.glob(f'gpurun_out/pmc_{name}/**/*counter_collection.csv',recursive=True)
    if not fs: print(name,'no file'); continue
    agg=collections.defaultdict(float); n=collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        if 'attention_kernelILb0ELb0' in row['Kernel_Name']:
            agg[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
    print(name,{c:round(v/max(n[c],1)) for c,v in agg.items()})
PY
