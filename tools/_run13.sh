cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frame.py -x -q -m gpu 2>&1 | tail -5
STRESS_HEAD=wip timeout 900 python tools/stress_parity.py 420 5001 gpurun_out/r05_stress.jsonl > gpurun_out/r05l_stress.log 2>&1
tail -3 gpurun_out/r05l_stress.log
