#!/bin/bash
# DMA-only stream from private regions of different sizes (L2 / Infinity Cache / HBM): rate, power, clock
cd $GRAFT_REPO_ROOT/tools/ubench
for kib in 128 512 768 2048 16384; do
  ./power_mix 8 5 $kib > /tmp/pm_$kib.log 2>&1 &
  P=$!
  sleep 1.5
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Current Socket|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.3
    kill -0 $P 2>/dev/null || break
  done | sort | awk '{a[NR]=$0} END {print "   region '$kib' KiB x 256: power/sclk median sample: " a[int((NR+1)/2)]}'
  wait $P
  cat /tmp/pm_$kib.log
done
