cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py::test_skewed_reference_shaped_workload_sampled_parity -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05e_tests.log
python bench.py --steps 6 --warmup 2 --cpu-baseline off --verify off --boundary off --predict-world 0 --cfg1 off --sweep "" > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err
