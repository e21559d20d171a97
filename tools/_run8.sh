cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r05h_tests.log
tail -3 gpurun_out/r05h_tests.log
python bench.py --steps 4 --warmup 1 --cpu-baseline off --verify off --predict-world 0 --cfg1 off --sweep none --skew off > gpurun_out/r05h_bench.json 2> gpurun_out/r05h_bench.err
