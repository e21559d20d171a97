#!/bin/bash
# average duration of every kernel of the default bench step (rocprofv3 kernel trace): bash tools/kernel_times.sh [tag]
TAG=${1:-kt}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rm -rf gpurun_out/$TAG; mkdir -p gpurun_out/$TAG
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 bench.py --steps 6 --warmup 2 --cpu-baseline off --verify off --boundary off --sweep none > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/err.log
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/%s/*/*kernel_stats.csv' % sys.argv[1])[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:16]:
    print('%-64s calls %4s avg %8.1f us' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
