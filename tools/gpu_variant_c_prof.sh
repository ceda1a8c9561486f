#!/bin/bash
# per-kernel statistics of tools/bench_variant_c.py (variant C with trained adapters, batch 256)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/vc; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vc -- python3 $R/tools/bench_variant_c.py 2>&1 | grep -E "^\{"
f=$(find $R/gpurun_out/vc -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print('%-95s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'][:95], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
