#!/bin/bash
# BASELINE.md §2 in full: 10 warm-up + 200 timed queries at each of three OpenMP thread settings
# (1, nproc-4 = the reference's rule, all cores) on the GPU box's host, probe loop (CS1) and whole
# candidate_selector reported separately, at F = 1 000 (configs[1]) and F = 10 000 (north star).
set -u
TAG=${1:-r02c}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
for F in 1000 10000; do
  timeout 1500 python3 bench.py --frames $F --queries 256 --steps 5 --cpu-baseline on --cpu-protocol full --sweep none --verify off --boundary off \
    > $OUT/cpu_protocol_F$F.json 2> $OUT/cpu_protocol_F$F.err
  echo "F=$F rc=$?"
done
SGTD_BENCH_BACKEND=gloo SGTD_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 5 --cpu-baseline off --sweep none --verify off --boundary off \
  > $OUT/bench_2rank_one_gpu_gloo.json 2> $OUT/bench_2rank_one_gpu_gloo.err
echo "2rank rc=$?"
