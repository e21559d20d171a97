cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_exchange.py tests/test_gpu_configs.py::test_skewed_reference_shaped_workload_sampled_parity -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05c_tests.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
tail -3 gpurun_out/r05c_tests.log; tail -c 1000 gpurun_out/r05c_bench.err
