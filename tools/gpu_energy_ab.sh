#!/bin/bash
# Round 4: rank variants by bytes and joules per step, not by us per launch.  For each CONFIGS="name:ENV=V,ENV=V name2:-"
# entry, on ONE box: (1) bench line (step ms, kernel table) with package power / clock / joules per step sampled beside the
# headline step, (2) FETCH_SIZE and WRITE_SIZE passes over a short run -> HBM-side bytes per step and per kernel class.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
STEPS=${AB_STEPS:-30}
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export $kv; done; fi
  python3 $R/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extra-configs --no-class-rows --power > $R/gpurun_out/eab_$name.log 2>&1
  python3 - "$name" $R/gpurun_out/eab_$name.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith('{')]
if not l: print(sys.argv[1], 'FAILED', open(sys.argv[2]).read()[-800:]); sys.exit()
d=json.loads(l[-1]); p=d.get('power') or {}
print(f"[bench] {sys.argv[1]}: {d['ms_per_step']} ms/step e2e {d['roofline']['e2e_frac']} | power {p.get('package_watts',{}).get('median')} W sclk {p.get('sclk_mhz',{}).get('median')} MHz J/step {p.get('joules_per_step')} (burst {p.get('burst_ms_per_step')} ms) | "
      + ' | '.join(f"{k['name'][:12]} {k['avg_ms']*1e3:.0f}" for k in d['kernels'][:7]))
PY
  if [ -z "$NO_PMC" ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $R/gpurun_out/pmc_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-class-rows > $R/gpurun_out/pmc_$c.log 2>&1
    done
    python3 $R/tools/step_traffic.py $name 7 $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE | grep -v '^{'
  fi
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do unset ${kv%%=*}; done; fi
done
