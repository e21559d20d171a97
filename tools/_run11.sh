cd $GRAFT_REPO_ROOT
SGTD_DEBUG=1 python bench.py --steps 6 --warmup 2 --cpu-baseline off --verify off --boundary off --predict-world 0 --cfg1 off --sweep "" --in-flight 1 > gpurun_out/r05k_bench.json 2> gpurun_out/r05k_bench.err
grep "re-run" gpurun_out/r05k_bench.err | tail -30
