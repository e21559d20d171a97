cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frame.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05f_tests.log
tail -5 gpurun_out/r05f_tests.log
python -m pytest tests/test_example_localize.py tests/test_gpu_parity.py -x -q -m gpu -k "localize or manager or shim or verify or build" 2>&1 | tail -8 >> gpurun_out/r05f_tests.log
python bench.py --steps 4 --warmup 1 --cpu-baseline off --verify off --predict-world 0 --cfg1 off --sweep none --skew off > gpurun_out/r05f_bench.json 2> gpurun_out/r05f_bench.err
