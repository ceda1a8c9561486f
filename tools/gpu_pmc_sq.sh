#!/bin/bash
# SQ counter pass over the all-rows headline step (own run: --pmc with --kernel-trace only): MFMA busy cycles, wave cycles, waits, LDS
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-class-rows > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
fs=glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv',recursive=True)
tr=glob.glob('gpurun_out/pmc_sq/**/*kernel_trace.csv',recursive=True)
dur=collections.defaultdict(list)
if tr:
    for r in csv.DictReader(open(tr[0])): dur[r['Kernel_Name'][:70]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for row in csv.DictReader(open(fs[0])):
    k=row['Kernel_Name'][:70]; agg[k][row['Counter_Name']]+=float(row['Counter_Value']); disp[k].add(row['Dispatch_Id'])
print("kernel | launches | us (under the profiler) | MFMA busy cycles per launch | = MFMA-busy / (4 SIMD x 256 CU x cycles at the kernel's duration x 2.0 GHz) | LDS bank conflict / LDS active | wait / wave cycles")
for k in sorted(agg,key=lambda k:-agg[k].get('SQ_VALU_MFMA_BUSY_CYCLES',0))[:10]:
    n=len(disp[k]); a={c:v/n for c,v in agg[k].items()}
    us=sum(dur[k])/max(len(dur[k]),1)/1e3
    busy=a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)
    frac=busy/(1024*us*1e-6*2.0e9) if us else 0
    print(f"{k[:64]:64s} | {n:3d} | {us:7.1f} | {busy:14.0f} | {frac:5.3f} | {a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1):6.4f} | {a.get('SQ_WAIT_ANY',0)/max(a.get('SQ_WAVE_CYCLES',1),1):5.3f}")
PY
