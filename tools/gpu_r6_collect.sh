#!/bin/bash
# round-6 collection on ONE box: GPU test suite, full bench line, kernel trace (--stats) of the all-rows headline alone and of the
# default bench run, FETCH_SIZE / WRITE_SIZE passes of the headline (separate rocprofv3 --pmc runs), power beside the headline.
# Outputs under gpurun_out/r06/ ; tools/r6_profiles.py (run at home) copies the summaries into profiles/ and rewrites profiles/traffic.json.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
python bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_head && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_head -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-class-rows > $O/bench_head.log 2>&1
f=$(ls -t $O/prof_head/*/*kernel_stats.csv | head -1); test -n "$f" && cp $f $O/headline_allrows_kernel_stats.csv && head -8 $f | cut -c1-150
tail -1 $O/bench_head.log > $O/headline_allrows_bench_line.json
rm -rf $O/prof_full && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof_full.log 2>&1
f=$(ls -t $O/prof_full/*/*kernel_stats.csv | head -1); test -n "$f" && cp $f $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-class-rows > $O/pmc_$c.log 2>&1
done
cd $R
python3 tools/step_traffic.py r06 7 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -14 $O/pmc_summary.txt | cut -c1-200
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-class-rows --power > $O/bench_power.log 2>&1; tail -1 $O/bench_power.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('power') or {}
print('[power]', d['ms_per_step'], 'ms/step', p.get('package_watts',{}).get('median'), 'W', p.get('sclk_mhz',{}).get('median'), 'MHz', p.get('joules_per_step'), 'J/step')"
rm -rf $O/prof_head $O/prof_full     # (the kernel traces themselves are large; the summaries above are what is kept)
