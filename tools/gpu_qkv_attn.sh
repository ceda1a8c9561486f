#!/bin/bash
# round 4: fused in_proj + attention kernel: parity tests, timing against the two kernels it replaces (XCD group sizes), bench step
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q -k "fused" > gpurun_out/qa_tests.log 2>&1; tail -15 gpurun_out/qa_tests.log
GSZ="0 3 2 1" timeout 300 python tools/qkv_attn_time.py 2>&1 | tail -4
for v in 1 0; do HG_QKV_ATTN=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-class-rows > gpurun_out/qa_bench_$v.log 2>&1; python - $v <<'PY'
import json,sys
l=[x for x in open(f"gpurun_out/qa_bench_{sys.argv[1]}.log") if x.startswith('{')]
if not l: print('bench FAILED', open(f"gpurun_out/qa_bench_{sys.argv[1]}.log").read()[-1500:]); sys.exit()
d=json.loads(l[-1]); print(f"HG_QKV_ATTN={sys.argv[1]}: {d['ms_per_step']} ms/step e2e {d['roofline']['e2e_frac']} | " + ' | '.join(f"{k['name'][:14]} {k['avg_ms']*1e3:.0f}" for k in d['kernels'][:8]))
PY
done
