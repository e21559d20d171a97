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
for T in 2 4 6 8 12; do
g++ -std=c++17 -O2 -DSGTD_SHIM_FILL_THREADS=${T}u -Iinclude examples/localize.cpp -o /tmp/localize_t$T -Lsgtd_amd -lsgtd_accel -Wl,-rpath,$PWD/sgtd_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -pthread || exit 1
done
for rep in 1 2; do for T in 2 4 6 8 12; do echo "threads $T: $(LOCALIZE_PER_FRAME=$Q /tmp/localize_t$T /tmp/map.cache /tmp/query.cache $Q 2>&1 | grep 'per-frame\|by part' | sed 's/per-frame calls through STDescManager (32 frames): //; s/; 32.*//' | tr '\n' ' ')"; done; done
