#!/bin/bash
# per-kernel means of the headline step for the MLP pair kernel's variants, one box (tools/mlp_pair_kernels.py)
R=$GRAFT_REPO_ROOT; cd $R
run() { tag=$1; shift; env TAG=$tag "$@" python3 tools/mlp_pair_kernels.py 2>&1 | grep "^\[" ; }
for cfg in $CONFIGS; do
  name=${cfg%%:*}; envs=${cfg#*:}
  if [ "$envs" = "-" ]; then run $name; else run $name ${envs//,/ }; fi
done
