#!/bin/bash
# step-level A/B of environment settings: CONFIGS="name:ENV=V,ENV=V ..." ("-" = no variables), each run REPS times, alternating
R=$GRAFT_REPO_ROOT; cd $R
for rep in $(seq 1 ${REPS:-2}); do
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export $kv; done; fi
    python bench.py --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-extra-configs --no-class-rows 2>&1 | tail -1 | NAME=$name python3 -c "
import sys,json,os
d=json.loads(sys.stdin.read())
print(os.environ['NAME'], d['value'], d['ms_per_step'], [(k['name'][:8],k['avg_ms']) for k in d['kernels'][:5]])" )
done; done
