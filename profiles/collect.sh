#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object refers to.
# Run on the GPU box from the repo root:   bash profiles/collect.sh r01
# Kernel timing and PMC counters are separate runs (counters perturb timing);
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 10 --warmup 2 --cpu-baseline off"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_under_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_under_write.json 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/bench_under_tcc.json 2> $OUT/tcc.err
python3 profiles/summarize.py $OUT $TAG
