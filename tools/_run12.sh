cd $GRAFT_REPO_ROOT
STRESS_VERBOSE=1 SGTD_DEBUG=1 timeout 900 python tools/stress_parity.py 600 5001 > gpurun_out/r05l_stress.log 2>&1
tail -12 gpurun_out/r05l_stress.log
