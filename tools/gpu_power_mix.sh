#!/bin/bash
# power / clock beside each variant of tools/ubench/power_mix (build it first: hipcc --offload-arch=gfx950 -O3 -o power_mix power_mix.hip)
cd $GRAFT_REPO_ROOT/tools/ubench
for m in ${MASKS:-1 2 3 4 6 7 8 11}; do
  ./power_mix $m 5 > /tmp/pm_$m.log 2>&1 &
  P=$!
  sleep 1.5
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Current Socket|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.3
    kill -0 $P 2>/dev/null || break
  done | sort | awk '{a[NR]=$0} END {print "   power/sclk median sample: " a[int((NR+1)/2)] "   (" NR " samples)"}'
  wait $P
  cat /tmp/pm_$m.log
done
