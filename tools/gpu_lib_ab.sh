#!/bin/bash
# A/B of library variants (ab/<name>.so built by tools/build_variant.sh) on the headline step, one box, alternating:
# LIBS="r2h0 r2h1 r2h2" [NO_PMC=1] bash tools/gpu_lib_ab.sh  ->  ms per step, power, J per step, HBM-side MB per step and per kernel
R=$GRAFT_REPO_ROOT
for l in $LIBS; do CONFIGS="$l:HG_LIB_PATH=$R/ab/$l.so" bash $R/tools/gpu_energy_ab.sh; done
