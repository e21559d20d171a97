cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frame.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05g_tests.log
tail -3 gpurun_out/r05g_tests.log
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
import numpy as np
from sgtd_amd import synth, ingest, evaluate as ev
m = synth.make_map(10000, 200, stream=1)
q = synth.make_queries(m, 32, stream=1)
os.makedirs('/tmp/lz', exist_ok=True)
ingest.write_cache('/tmp/lz/map.cache', m.xyz, m.label, np.stack([ev.pose_row(*p) for p in m.pose]))
ingest.write_cache('/tmp/lz/q.cache', q.xyz, q.label, np.stack([ev.pose_row(*p) for p in q.pose]))
PY
g++ -std=c++17 -O2 -Iinclude examples/localize.cpp -o examples/localize -Lsgtd_amd -lsgtd_accel -Wl,-rpath,$PWD/sgtd_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -pthread
cd /tmp && export TMPDIR=/tmp
LOCALIZE_PER_FRAME=32 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_frame -o frame -- $GRAFT_REPO_ROOT/examples/localize /tmp/lz/map.cache /tmp/lz/q.cache 32 > $GRAFT_REPO_ROOT/gpurun_out/r05g_localize.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_frame | head; find gpurun_out/prof_frame -name "*kernel_trace.csv" -size +6M -delete
