#!/bin/bash
# per-kernel statistics of encode_image at a small batch (BATCHES=32 by default)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; export BATCHES=${BATCHES:-32}
rm -rf $R/gpurun_out/sb; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb -- python3 $R/tools/bench_batch_sweep.py 2>&1 | grep -E "^\{"
f=$(find $R/gpurun_out/sb -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print('%-90s calls %6s avg %8.1f us  %5.1f%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
