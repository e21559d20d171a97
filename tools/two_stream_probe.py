"""Experiment: do two handles on two HIP streams overlap each other's HBM-bound assembly stages
with the other's VALU-bound sweep?  (python tools/two_stream_probe.py [frames] [queries])"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sgtd_amd import manager, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
m = synth.make_map(F, 200, stream=1)
qs = synth.make_queries(m, Q, stream=1)


def run(n_handles, steps=6):
    hs, ss = [], []
    for i in range(n_handles):
        h = manager.STDescManager()
        h.add_frames(m.xyz, m.label)
        s = torch.cuda.Stream()
        h.set_stream(s.cuda_stream)
        hs.append(h); ss.append(s)
    per = Q // n_handles
    parts = [(qs.xyz[i * per:(i + 1) * per], qs.label[i * per:(i + 1) * per]) for i in range(n_handles)]

    def step():
        for h, (x, l) in zip(hs, parts):
            h.query_frames(x, l, fetch=False)

    for _ in range(2):
        step()
    for h in hs:
        h.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for h in hs:
        h.sync()
    dt = (time.perf_counter() - t0) / steps
    for h in hs:
        h.close()
    return dt


for n in (1, 2, 1, 2, 4):
    dt = run(n)
    print("handles %d: %.2f ms per %d queries -> %.0f frames/s" % (n, dt * 1e3, Q, Q / dt), flush=True)
