cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05i_tests.log
tail -3 gpurun_out/r05i_tests.log
SGTD_BENCH_BACKEND=gloo SGTD_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 6 --warmup 2 --cpu-baseline off > gpurun_out/r05i_2rank_auto.json 2> gpurun_out/r05i_2rank_auto.err
