#!/bin/bash
# bench line (short) for several environment settings in one gpurun call: CONFIGS="name:ENV=V,ENV=V name2:..."
mkdir -p gpurun_out
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export $kv; done; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs ${BENCH_FLAGS} > gpurun_out/ab_$name.log 2>&1
  python3 - "$name" gpurun_out/ab_$name.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith('{')]
if not l: print(sys.argv[1], 'FAILED', open(sys.argv[2]).read()[-600:]); sys.exit()
d=json.loads(l[-1])
print(sys.argv[1], d['value'], 'crops/s', d['ms_per_step'], 'ms e2e', d['roofline']['e2e_frac'], '|', ' | '.join(f"{k['name'][:14]} {k['avg_ms']*1e3:.0f}" for k in d['kernels'][:6]))
PY
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do unset ${kv%%=*}; done; fi
done
