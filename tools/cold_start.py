"""first batch of a fresh handle, timed step by step (python tools/cold_start.py [frames=10000] [queries=2048])"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from sgtd_amd import synth
from sgtd_amd.manager import STDescManager
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
m = synth.make_map(F, 200, stream=1); q = synth.make_queries(m, Q, stream=1)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
mx, ml = t(m.xyz), t(m.label.astype(np.int64)).to(torch.int32)
qx, ql = t(q.xyz), t(q.label.astype(np.int64)).to(torch.int32)
torch.cuda.synchronize()
g = STDescManager(max_frame_n=max(20000, F + 1))
g.set_stream(torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter(); g.add_frames(mx, ml); g.finalize(); torch.cuda.synchronize(); print("map build %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for i in range(3):
    t0 = time.perf_counter(); g.query_frames(qx, ql, fetch=False); t1 = time.perf_counter(); g.sync(); torch.cuda.synchronize()
    print("batch %d: enqueue %.1f ms, total %.1f ms, overflowed %d" % (i, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0), g.stats()["overflowed"]))
