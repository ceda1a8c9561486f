#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for m in 0 1 2 3 4 6 0; do
  rm -rf $R/gpurun_out/pm; HG_RING_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pm.log 2>&1
  f=$(find $R/gpurun_out/pm -name "*kernel_stats.csv" | head -1)
  echo "mode $m: $(grep -E 'gemm_ring<4, 1>|gemm_ring<4, 0>|gemm_ring<2, 3>' $f | awk -F, '{printf "%s %s avg=%.0fus min=%.0f max=%.0f | ", $1,$2,$5/1000,$7/1000,$8/1000}')"
done
