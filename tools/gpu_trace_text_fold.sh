#!/bin/bash
# kernel traces of config 3 under the three settings of option text_ln_fold
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for f in 0 1 2; do
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_tf$f
OPTS="text_ln_fold=$f" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tf$f -- python3 $R/tools/text77_time.py > $R/gpurun_out/tf$f.log 2>&1
cd $R; echo "== text_ln_fold=$f"; grep '^{' gpurun_out/tf$f.log
fcsv=$(ls -t gpurun_out/prof_tf$f/*/*kernel_stats.csv | head -1); test -n "$fcsv" && head -9 "$fcsv" | cut -c1-150
done
