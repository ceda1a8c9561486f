#!/bin/bash
# kernel trace of config 3 (600 prompts x 77 tokens through the text tower)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_t77
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t77 -- python3 $R/tools/text77_time.py > $R/gpurun_out/text77_prof.log 2>&1
cd $R; grep '^{' gpurun_out/text77_prof.log
f=$(ls -t gpurun_out/prof_t77/*/*kernel_stats.csv | head -1); test -n "$f" && head -24 "$f" | cut -c1-180
python3 tools/text77_time.py 2>&1 | grep '^{'
