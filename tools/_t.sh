cd $GRAFT_REPO_ROOT
for seed in 5004 5005 5006; do
STRESS_HEAD=00e5a52 timeout 700 python tools/stress_parity.py 540 $seed gpurun_out/r05_stress_more.jsonl > gpurun_out/r05u_stress_$seed.log 2>&1
tail -1 gpurun_out/r05u_stress_$seed.log
done
