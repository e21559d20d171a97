#!/usr/bin/env python3
"""bench.py — query frames/s of the triangle-descriptor build + hash-match hot path.

A step = one pass of the hot path over one batch of Q synthetic query frames
(keypoints already resident in HBM): BuildSingleScanSTD for every query frame,
then candidate_selector (probe, votes, top-50, ordered match lists) against an
F-frame map table — all on the GPU through the C ABI.  Default workload =
BASELINE.json configs[1]: 200 keypoints/frame, 1k-frame map.

N > 1 (launched by torch.distributed.run, one rank per GPU; total work fixed =
strong scaling), two modes (--shard):
  table  the map's hash table is sharded by frame range over the ranks, every
         rank sweeps its shard with all Q queries, local top-50 tables are
         all-gathered with RCCL and merged (sgtd_amd/dist.py::ShardedMap);
  query  the table is replicated and every rank serves Q/N queries of the batch,
         result tables are all-gathered with RCCL (ReplicatedMap);
  auto   query when the whole table needs < 1/8 of one GPU's HBM, else table.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_BYTES = 288e9
KERNEL_KEYS = ("ms_build", "ms_sort", "ms_probe", "ms_votes", "ms_topk", "ms_count", "ms_scan", "ms_write", "ms_total")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1000, help="map size F")
    ap.add_argument("--keypoints", type=int, default=200, help="keypoints per frame N")
    ap.add_argument("--queries", type=int, default=4096, help="query frames per step (whole job)")
    ap.add_argument("--shard", choices=["auto", "table", "query"], default="auto")
    ap.add_argument("--cpu-baseline", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work budget of the timed sample")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps timed per kernel for the roofline")
    ap.add_argument("--verify", choices=["on", "off"], default="on",
                    help="also time candidate_verify + SearchLoop on the device (reported beside, never inside, value)")
    return ap.parse_args()


def cpu_baseline(smap, queries, gpu_results, mgr, budget_s):
    """the oracle (a port of the reference CPU path, reference data layout) timed on
    this box's host cores on a bounded sample of the same workload"""
    from oracle.oracle import OracleManager
    ncpu = os.cpu_count() or 1
    threads = ncpu - 4 if ncpu > 4 else (2 if ncpu > 3 else 1)  # the reference's MP_PROC_NUM rule, CMakeLists.txt:22-41
    F = smap.xyz.shape[0]
    o = OracleManager(num_threads=threads, max_frame_n=max(20000, F + 1))
    t0 = time.time()
    for f in range(F):
        o.build(smap.xyz[f], smap.label[f], export=False)
        o.add_last()
    t_map = time.time() - t0
    n_done, t_query, ident, P, M = 0, 0.0, 0, 0, 0
    res = gpu_results
    for q in range(queries.xyz.shape[0]):
        t1 = time.time()
        o.build(queries.xyz[q], queries.label[q], export=False)
        r = o.select()
        t_query += time.time() - t1
        n_done += 1
        c = o.counters()
        P += c["P"]; M += c["M"]
        nc = int(res.n_cand[q])
        same = (np.array_equal(res.cand_frame[q, :nc], r["cand_frame"]) and
                np.array_equal(res.cand_votes[q, :nc], r["cand_votes"]))
        if same:
            qi, de = mgr.result_pairs(q, res)
            same = np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
        ident += int(same)
        if t_query > budget_s and n_done >= 3:
            break
    return dict(value=n_done / t_query, unit="frames/s", cores=threads, kind="port",
                sample="%d of the %d query frames of one step, same %d-frame map; oracle map build %.1f s untimed"
                       % (n_done, queries.xyz.shape[0], F, t_map),
                ms_per_query=1000.0 * t_query / n_done,
                host_cpus=ncpu), dict(queries_checked=n_done, identical_candidates_votes_matchlists=ident,
                                      P_per_query=P / n_done, M_per_query=M / n_done)


def recall(smap, queries, top1):
    """top-1 candidate (no geometric verification) against the synthetic ground truth:
    exact frame, and map pose within 5 m of the query's true pose (the reference's
    success radius, semantic_graph_localization.cpp:750)"""
    ok = top1 >= 0
    pose = smap.pose[np.clip(top1, 0, smap.pose.shape[0] - 1), :2]
    dist = np.linalg.norm(pose - queries.pose[:, :2], axis=1)
    return {"top1_is_gt_frame": float(np.mean(ok & (top1 == queries.gt_frame))),
            "top1_pose_within_5m": float(np.mean(ok & (dist < 5.0)))}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from sgtd_amd import synth
    from sgtd_amd.dist import ReplicatedMap, ShardedMap, query_slice, shard_range
    from sgtd_amd.manager import STDescManager

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # SGTD_BENCH_BACKEND=gloo + SGTD_BENCH_SHARE_GPU=1 let the N>1 code path be exercised on a
    # 1-GPU box (all ranks on cuda:0, collectives over gloo); never used for reported numbers
    backend = os.environ.get("SGTD_BENCH_BACKEND", "nccl")
    if os.environ.get("SGTD_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    F, N, Q = args.frames, args.keypoints, args.queries
    smap = synth.make_map(F, N, stream=1)
    queries = synth.make_queries(smap, Q, stream=1)

    def to_dev(xyz, label):
        return (torch.from_numpy(np.ascontiguousarray(xyz)).to(dev).contiguous(),
                torch.from_numpy(np.ascontiguousarray(label).astype(np.int64)).to(dev).to(torch.int32).contiguous())

    # cold + probe layout of the whole table: ~170 B per descriptor, <= 36*N per frame
    table_bytes = 170.0 * 36 * N * F
    mode = "single"
    if world > 1:
        mode = args.shard if args.shard != "auto" else ("query" if table_bytes < HBM_BYTES / 8 else "table")

    stream = torch.cuda.current_stream()
    merged = {}
    if mode == "single":
        mgr = STDescManager(device_id=local_rank, max_frame_n=max(20000, F + 1))
        mgr.set_stream(stream.cuda_stream)
        mgr.add_frames(*to_dev(smap.xyz, smap.label))
        mgr.finalize()
        d_qxyz, d_qlab = to_dev(queries.xyz, queries.label)
        q_lo, q_hi = 0, Q

        def step():
            mgr.query_frames(d_qxyz, d_qlab, fetch=False)
    elif mode == "table":
        sm = ShardedMap(F, rank, world, device_id=local_rank)
        mgr = sm.mgr
        mgr.set_stream(stream.cuda_stream)
        lo, hi = shard_range(F, world, rank)
        sm.add_shard_frames(*to_dev(smap.xyz[lo:hi], smap.label[lo:hi]))
        d_qxyz, d_qlab = to_dev(queries.xyz, queries.label)
        q_lo, q_hi = 0, Q

        def step():
            merged["out"] = sm.query(d_qxyz, d_qlab)[:2]
    else:
        rm = ReplicatedMap(F, rank, world, device_id=local_rank)
        mgr = rm.mgr
        mgr.set_stream(stream.cuda_stream)
        rm.add_frames(*to_dev(smap.xyz, smap.label))
        q_lo, q_hi = query_slice(Q, world, rank)
        d_qxyz, d_qlab = to_dev(queries.xyz[q_lo:q_hi], queries.label[q_lo:q_hi])

        def step():
            merged["out"] = rm.query(d_qxyz, d_qlab, Q)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    mgr.sync()          # grows work buffers if the first batch overflowed them
    step()
    mgr.sync()
    assert mgr.stats()["overflowed"] == 0

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mgr.sync()
    st = mgr.stats()
    assert st["overflowed"] == 0, "a timed step overflowed a work buffer"

    # ---- per-kernel timing for the roofline (HIP events on the handle's stream)
    mgr.set_timing(True)
    acc = {}
    for _ in range(args.profile_steps):
        step()
        mgr.sync()
        s = mgr.stats()
        for k in KERNEL_KEYS:
            acc.setdefault(k, []).append(s[k])
    mgr.set_timing(False)
    kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}
    st = mgr.stats()
    P, M, D = st["last_P"], st["last_M"], st["last_D"]
    probe_bytes = 28 * P + 64 * D + 8 * M     # algorithmic bytes of one sweep launch
    achieved = probe_bytes / (kern_ms["ms_probe"] * 1e-3) / 1e9 if kern_ms["ms_probe"] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "probe_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("frames") == F and tj.get("queries") == Q and tj.get("keypoints") == N and tj.get("gpus", 1) == world:
                traffic = tj.get("bytes_per_launch")
        except Exception:
            traffic = None

    res = mgr.results()
    # ---- next stage of the reference's SearchLoop (STDesc.cpp:105-146), reported separately
    verify = None
    if mode == "single" and args.verify == "on":
        try:   # the extra stage must never cost the headline line
            mgr.verify(); mgr.sync(); torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(3):
                mgr.verify()
            torch.cuda.synchronize()
            tv = (time.perf_counter() - tv) / 3
            bc, bf, bs = mgr.search_loop()
            hit = bf >= 0
            ok = hit & (np.linalg.norm(smap.pose[np.clip(bf, 0, F - 1), :2] - queries.pose[:, :2], axis=1) < 5.0)
            verify = {"ms_per_batch": 1000.0 * tv, "queries": Q, "loops_found": int(hit.sum()),
                      "loop_pose_within_5m": int(ok.sum()), "pairs_verified": int(st["last_cand_pairs"])}
            # the node's own metrics (semantic_graph_localization.cpp:605-745) on the synthetic ground truth
            from sgtd_amd import evaluate as ev
            map_pose4 = np.stack([ev.pose_matrix(*p) for p in smap.pose])
            met = ev.LoopMetrics(mgr.config_setting_["candidate_num"])
            for q in range(min(Q, 256)):
                n_c = int(res.n_cand[q])
                if bf[q] > 0:
                    score, rot, t = mgr.result_verify(q)
                    ev.account(met, ev.pose_matrix(*queries.pose[q]), map_pose4, int(bf[q]), rot[int(bc[q])], t[int(bc[q])],
                               res.cand_frame[q, :n_c], score[:n_c])
                else:
                    ev.account(met, ev.pose_matrix(*queries.pose[q]), map_pose4, int(bf[q]), None, None, (), ())
            loc = met.summary()
            loc["STD_num"] = loc["STD_num"][:10]
            verify["localization_first_256_queries"] = loc
        except Exception as exc:   # reported, not fatal
            verify = {"error": "%s: %s" % (type(exc).__name__, exc)}
    out = None
    if rank == 0:
        value = Q * args.steps / elapsed
        sharding = {"single": "none",
                    "table": "map frames range-sharded over %d GPUs, every rank sweeps all queries, RCCL all_gather + merge of top-50" % world,
                    "query": "map replicated on %d GPUs, queries sharded, RCCL all_gather of the result tables" % world}[mode]
        out = {
            "metric": "query frames/sec vs map size (descriptor build + candidate selection)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic %d keypoints/frame, %d-frame map, descriptor build+match" % (N, F),
                       "map_frames": F, "keypoints_per_frame": N, "queries_per_step": Q,
                       "sharding": sharding, "queries_this_rank": q_hi - q_lo,
                       "table_entries_this_rank": st["n_entries"], "table_buckets_this_rank": st["n_buckets"]},
            "roofline": {"bound": "hbm", "kernel": "probe_sorted_kernel (sweep)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_GBps": (traffic / (kern_ms["ms_probe"] * 1e-3) / 1e9) if traffic and kern_ms["ms_probe"] > 0 else None,
                         "algorithmic_bytes_per_launch": probe_bytes,
                         "P_visited": P, "M_matches": M, "D_query_descs": D, "candidate_pairs": st["last_cand_pairs"],
                         "kernel_ms": kern_ms},
        }
        if verify is not None:
            out["verify"] = verify
        if mode == "single":
            out["recall"] = recall(smap, queries, res.top1())
        else:
            f, v = merged["out"]
            out["recall"] = recall(smap, queries, f[:, 0].cpu().numpy())
        want_cpu = args.cpu_baseline == "on" or (args.cpu_baseline == "auto" and F <= 2000)
        if mode == "single" and want_cpu:
            try:
                cb, par = cpu_baseline(smap, queries, res, mgr, args.cpu_seconds)
                out["cpu_baseline"] = cb
                out["parity"] = par
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
            except Exception as exc:   # a broken checker must not cost the measured line
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = "%s: %s" % (type(exc).__name__, exc)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
