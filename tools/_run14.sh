cd $GRAFT_REPO_ROOT
STRESS_HEAD=37a305e timeout 1000 python tools/stress_parity.py 600 5002 gpurun_out/r05_stress.jsonl > gpurun_out/r05m_stress.log 2>&1
tail -4 gpurun_out/r05m_stress.log
