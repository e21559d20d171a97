#!/bin/bash
# per-frame C++ adapter (examples/localize LOCALIZE_PER_FRAME) with glibc's default allocator settings
# and with the thresholds the example sets: bash tools/shim_malloc_exp.sh
F=10000; Q=32
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
for rep in 1 2; do
echo "== glibc defaults"; LOCALIZE_DEFAULT_MALLOC=1 LOCALIZE_PER_FRAME=$Q /tmp/localize_t /tmp/map.cache /tmp/query.cache $Q 2>&1 | tail -10 | grep -v "^map\|^mean\|^time"
echo "== thresholds raised (the example's default)"; LOCALIZE_PER_FRAME=$Q /tmp/localize_t /tmp/map.cache /tmp/query.cache $Q 2>&1 | tail -10 | grep -v "^map\|^mean\|^time"
done
