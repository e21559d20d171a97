#!/bin/bash
# per-frame C++ adapter timing with the shim's lap clocks: bash tools/shim_timing.sh [frames=10000] [queries=32]
F=${1:-10000}; Q=${2:-32}
python3 - "$F" "$Q" <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from sgtd_amd import synth, ingest, evaluate as ev
F, Q = int(sys.argv[1]), int(sys.argv[2])
m = synth.make_map(F, 200, stream=1); q = synth.make_queries(m, Q, stream=1)
ingest.write_cache('/tmp/map.cache', m.xyz, m.label, np.stack([ev.pose_row(*p) for p in m.pose]))
ingest.write_cache('/tmp/query.cache', q.xyz, q.label, np.stack([ev.pose_row(*p) for p in q.pose]))
PY
g++ -std=c++17 -O2 -DSGTD_SHIM_TIMING -Iinclude examples/localize.cpp -o /tmp/localize_t -Lsgtd_amd -lsgtd_accel -Wl,-rpath,$PWD/sgtd_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -pthread || exit 1
LOCALIZE_PER_FRAME=$Q /tmp/localize_t /tmp/map.cache /tmp/query.cache $Q 2>&1 | tail -40
