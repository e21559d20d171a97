#!/bin/bash
# One GPU-box session of round 2: parity tests, the default bench line, rocprofv3 kernel
# stats of the same command.  Usage (from the repo root on the GPU box):
#   bash profiles/run_round2.sh <tag> [tests|notests] [extra bench args]
set -u
TAG=${1:-r02a}
TESTS=${2:-tests}
EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$TESTS" = "tests" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 900 python3 bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 600 $OUT/bench.err
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --cpu-baseline off --verify off --boundary off --sweep "" $EXTRA > $OUT/bench_under_stats.json 2> $OUT/stats.err
python3 profiles/summarize.py $OUT $TAG > $OUT/summarize.log 2>&1
tail -3 $OUT/summarize.log
