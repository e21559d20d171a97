#!/usr/bin/env python3
"""The skewed, reference-shaped workload's step alone (bench.py's workload_skew leg without the default step around it), for
profiling: Zipf(1.2) labels over the 13 wild classes, 50-400 keypoints per frame, clustered landmarks (synth.make_skewed_map).
    python tools/skew_step.py [F] [queries per step] [steps]       -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sgtd_amd import synth
    from sgtd_amd.manager import STDescManager
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 832
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda", 0)

    def to_dev_flat(fr):
        return (torch.from_numpy(np.ascontiguousarray(fr.xyz.reshape(-1, 3))).to(dev), torch.from_numpy(np.ascontiguousarray(fr.label.reshape(-1))).to(dev))
    smap, world = synth.make_skewed_map(F, stream=31)
    g = STDescManager(device_id=0, max_frame_n=max(20000, F + 1))
    g.add_frames(*to_dev_flat(smap), kp_off=smap.kp_off)
    g.finalize()
    sets = [synth.make_skewed_queries(world, Q, stream=3100 + b) for b in range(2)]
    dsets = [(to_dev_flat(s), s.kp_off) for s in sets]

    def step(i):
        (x, l), off = dsets[i % len(dsets)]
        g.query_frames(x, l, kp_off=off, fetch=False)
    for i in range(3):
        step(i); g.sync()
    s0 = g.stats()
    g.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
        g.sync()
    el = time.perf_counter() - t0
    s1 = g.stats()
    print(json.dumps({"workload": "skewed, %d-frame map" % F, "queries_per_step": Q, "steps": steps, "frames_per_s_each_step_waited_for": Q * steps / el,
                      "ms_per_step": 1000.0 * el / steps, "reruns": int(s1["reruns_total"] - s0["reruns_total"]),
                      "kernel_ms_last_step": {k: s1[k] for k in ("ms_build", "ms_sort", "ms_probe", "ms_votes", "ms_topk", "ms_count", "ms_scan", "ms_write", "ms_total")},
                      "M_matches_per_query": s1["last_M"] / Q, "P_swept_per_query": s1["last_P_swept"] / Q, "candidate_pairs": int(s1["last_cand_pairs"])}))
    g.close()


if __name__ == "__main__":
    main()
