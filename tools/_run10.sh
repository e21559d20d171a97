cd $GRAFT_REPO_ROOT
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05j_bench.json 2> gpurun_out/r05j_bench.err
echo "bench wall s: $(( $(date +%s) - T0 ))"
