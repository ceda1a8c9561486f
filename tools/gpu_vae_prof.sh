#!/bin/bash
# per-kernel statistics of tools/vae_time.py (config 4)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/vae; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vae -- python3 $R/tools/vae_time.py 2>&1 | grep -v amdgpu.ids | tail -3
f=$(find $R/gpurun_out/vae -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-90s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
