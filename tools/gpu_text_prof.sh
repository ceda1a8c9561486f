#!/bin/bash
# per-kernel statistics of tools/text_time.py (config 3)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/txt; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/txt -- python3 $R/tools/text_time.py 2>&1 | grep -E "encode_text"
f=$(find $R/gpurun_out/txt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-100s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
