#!/usr/bin/env python3
"""bench.py — query frames/s of the triangle-descriptor build + hash-match hot path.

A step = one pass of the hot path over one batch of Q synthetic query frames
(keypoints already resident in HBM): BuildSingleScanSTD for every query frame,
then candidate_selector (probe, votes, top-50, ordered match lists) against an
F-frame map table — all on the GPU through the C ABI.

Default workload = the north-star point of BASELINE.json: 200 keypoints/frame,
10 000-frame map, 1 GPU (`config.workload` names it).  The same line carries a
`map_size_sweep` with F = 1 000 (configs[1]) and F = 4 541 (configs[2], the
KITTI-00 length).

N > 1: `python bench.py --gpus N` starts N ranks itself (child processes under
torch.distributed.run, before anything touches a GPU); when it is already
running under a launcher (RANK/WORLD_SIZE set) it is one rank.  The N ranks form
a grid of R_t table shards x R_q query groups (sgtd_amd/dist.py::Map2D):
  auto   (default) R_t = the smallest count whose shard fits one GPU's envelope
         (dist.plan_2d), R_q = N / R_t: every query group holds one copy of the
         (sharded) table and serves `--queries` query frames per step, so a step
         serves R_q x `--queries` frames — weak scaling in the query groups.  A
         10 000-frame map fits one GPU: R_t = 1, no data-path collective, one
         all_gather of the result tables per step (on a side stream).
  table  R_t = N: the form BASELINE.json's north_star names — the map's hash table
         sharded by frame range over all ranks, every rank sweeps its shard with the
         same `--queries` frames, ONE all_gather of the packed local top-50 tables and
         the merge kernel (STDesc.cpp:423-433) on a side stream while the match lists
         are written — strong scaling; also measured beside the headline
         (`table_sharded`) whenever the headline is another form, with the ceiling
         its replicated per-query work puts on it (`scaling_parts`).
  query  R_t = 1 whatever the map size.
Beside them, N > 1: `fixed_total_batch` (the headline's grid with `--queries` frames
per step in TOTAL: strong scaling), `multi_device_handle` — ONE process (rank 0)
driving all N devices through sgtd_create_multi, the form the reference's C++
caller uses; and, at N = 8, `cfg4`: BASELINE configs[3]'s 100 000-frame map.
N = 1 also carries `scaling_prediction`: what rank 0 of an 8-rank job does, measured
here part by part, for both forms.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0   # aggregate L2 rate of the 8 XCDs, same guide §L2
HBM_BYTES = 288e9
KERNEL_KEYS = ("ms_build", "ms_sort", "ms_probe", "ms_votes", "ms_topk", "ms_count", "ms_scan", "ms_write", "ms_total")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=10000, help="map size F")
    ap.add_argument("--keypoints", type=int, default=200, help="keypoints per frame N")
    ap.add_argument("--queries", type=int, default=2048,
                    help="query frames per step (per rank in query mode); 512 / 1024 / 2048 / 4096 give 81 / 92 / 98 / 101 k frames/s at F = 10 k")
    ap.add_argument("--shard", choices=["auto", "table", "query"], default="auto",
                    help="N>1: the grid of table shards x query groups `value` is measured on; auto = dist.plan_2d (the smallest number of "
                         "table shards whose shard fits one GPU), table = N shards, query = N replicas")
    ap.add_argument("--rt", type=int, default=None, help="N>1: table shards per query group, explicitly (must divide N)")
    ap.add_argument("--lists", choices=["all", "winners"], default="all",
                    help="N>1, table shards: match lists of every local candidate, written while the exchange runs (all), or of the merge's "
                         "winners only, written behind the exchange (winners)")
    ap.add_argument("--also-table", choices=["on", "off"], default="on",
                    help="N>1: also measure the pure table-sharded form beside a headline of another form (and check that both give the same lists)")
    ap.add_argument("--predict-world", type=int, default=8,
                    help="N=1: ranks of the job whose rank 0 is measured part by part for `scaling_prediction` (0 = off)")
    ap.add_argument("--skew", choices=["on", "off"], default="on", help="N=1: also measure the skewed, reference-shaped workload (workload_skew)")
    ap.add_argument("--cfg1", choices=["on", "off"], default="on", help="N=1: also run BASELINE configs[0] (graph JSON in, 100-frame map) with its CPU timing")
    ap.add_argument("--rccl-one", choices=["on", "off"], default="on",
                    help="N=1: also run the multi-GPU step of ONE rank with its collectives issued through RCCL all the same (a group of one: "
                         "what a one-GPU box can put on hardware of the exchange), in a child process")
    ap.add_argument("--rccl-one-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rccl-one-shape", choices=["headline", "cfg4"], default="headline",
                    help="N=1: the one-rank RCCL leg also at cfg4's shape (one of four 25 000-frame shards, lists = winners)")
    ap.add_argument("--multi-handle", choices=["auto", "on", "off"], default="auto",
                    help="N>1: also drive all N devices from ONE process through sgtd_create_multi (auto: when N real devices exist)")
    ap.add_argument("--cfg4-queries", type=int, default=None, help="query frames per query group of the cfg4 leg (default: --queries)")
    ap.add_argument("--cfg4", choices=["auto", "on", "off"], default="auto",
                    help="N>1: also measure BASELINE configs[3] (100 000-frame map, table-sharded); auto = at N == 8")
    ap.add_argument("--sweep", default="1000,4541",
                    help="other map sizes measured for map_size_sweep ('' = no other sizes, 'none' = also skip the incremental-insert leg)")
    ap.add_argument("--cpu-baseline", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU work budget of the timed sample")
    ap.add_argument("--cpu-protocol", choices=["bounded", "ref_full", "full"], default="ref_full",
                    help="ref_full = BASELINE.md §2 (10 warm-up + 200 queries) at the reference's thread rule, a bounded sample at the "
                         "other settings; full = the protocol at every setting (slow); bounded = a bounded sample everywhere")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="N=1: batches in flight over the one table (handles attached with sgtd_attach_table, one stream each); "
                         "1 = every batch behind the one before (reported beside the headline either way)")
    ap.add_argument("--repeat-region", type=int, default=5, help="how many times the headline's timed region is measured (median, min, max reported; value = the first)")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct query batches the timed region rotates through (all different from the warm-up batch)")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps timed per kernel for the roofline")
    ap.add_argument("--verify", choices=["on", "off"], default="on",
                    help="also time candidate_verify + SearchLoop on the device (reported beside, never inside, value)")
    ap.add_argument("--boundary", choices=["on", "off"], default="on",
                    help="also time the per-frame host-pointer adapter path (reported beside value)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pct(a):
    a = np.asarray(a, dtype=np.float64)
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90)), "n": int(a.size)}


def cpu_baseline(smap, queries, gpu_results, mgr, budget_s, protocol):
    """the oracle (a port of the reference CPU path, reference data layout and flags) timed on
    this box's host cores on a bounded sample of the same workload (three thread
    settings; per query the build, the probe loop = the reference's CS1, the whole selector)"""
    from oracle.oracle import OracleManager
    ncpu = os.cpu_count() or 1
    ref_threads = ncpu - 4 if ncpu > 4 else (2 if ncpu > 3 else 1)  # MP_PROC_NUM rule, CMakeLists.txt:22-41
    F = smap.xyz.shape[0]
    o = OracleManager(num_threads=ref_threads, max_frame_n=max(20000, F + 1))
    t0 = time.time()
    o.add_frames(smap.xyz, smap.label)       # = F x (build, add_last); builds on all host threads, untimed
    t_map = time.time() - t0
    nq_all = queries.xyz.shape[0]
    settings = [("ref_rule_nproc_minus_4", ref_threads), ("all_cores", ncpu), ("one_thread", 1)]
    share = {"ref_rule_nproc_minus_4": 0.5, "all_cores": 0.25, "one_thread": 0.25}
    from sgtd_amd.synth import effective_cpus
    quota = effective_cpus()        # a container may show all hardware threads and be allowed far fewer CPUs' worth of time
    if quota < ncpu:
        settings.insert(2, ("cgroup_quota_cpus", quota))
        share = {"ref_rule_nproc_minus_4": 0.4, "all_cores": 0.2, "cgroup_quota_cpus": 0.2, "one_thread": 0.2}
    out_settings, parity = {}, None
    q_next = 0
    for name, thr in settings:
        o.set_num_threads(thr)
        whole = protocol == "full" or (protocol == "ref_full" and name == "ref_rule_nproc_minus_4")
        warm = 10 if whole else 1
        want = 200 if whole else 10 ** 9
        lim = 1e9 if whole else budget_s * share[name]
        for w in range(warm):
            o.build(queries.xyz[w % nq_all], queries.label[w % nq_all], export=False)
            o.select()
        rows, spent, ident, P, M = [], 0.0, 0, 0, 0
        while len(rows) < want and len(rows) < nq_all:
            q = q_next % nq_all
            q_next += 1
            t1 = time.perf_counter()
            o.build(queries.xyz[q], queries.label[q], export=False)
            r = o.select()
            wall = time.perf_counter() - t1
            c = o.counters()
            rows.append((1000.0 * wall, c["build_ms"], c["probe_ms"], c["select_ms"]))
            spent += wall
            P += c["P"]; M += c["M"]
            if name == "ref_rule_nproc_minus_4":      # parity leg on the primary setting
                res = gpu_results
                nc = int(res.n_cand[q])
                same = (np.array_equal(res.cand_frame[q, :nc], r["cand_frame"]) and
                        np.array_equal(res.cand_votes[q, :nc], r["cand_votes"]))
                if same:
                    qi, de = mgr.result_pairs(q, res)
                    same = np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
                ident += int(same)
            if spent > lim and len(rows) >= 3:
                break
        a = np.array(rows)
        out_settings[name] = {"threads": thr, "queries_timed": len(rows), "frames_per_s": len(rows) / spent,
                              "ms_per_query": pct(a[:, 0]), "ms_build": pct(a[:, 1]),
                              "ms_probe_loop_CS1": pct(a[:, 2]), "ms_candidate_selector": pct(a[:, 3])}
        if name == "ref_rule_nproc_minus_4":
            parity = dict(queries_checked=len(rows), identical_candidates_votes_matchlists=ident,
                          P_per_query=P / len(rows), M_per_query=M / len(rows))
    prim = out_settings["ref_rule_nproc_minus_4"]
    best = max(s["frames_per_s"] for s in out_settings.values())
    cb = dict(value=prim["frames_per_s"], unit="frames/s", cores=ref_threads, kind="port",
              sample="%d (nproc-4 threads) + %d (all cores) + %d (1 thread) of the step's query frames, same %d-frame map; "
                     "oracle map build %.1f s untimed; protocol %s"
                     % (prim["queries_timed"], out_settings["all_cores"]["queries_timed"],
                        out_settings["one_thread"]["queries_timed"], F, t_map, protocol)
                     + ("; %d more on the %d CPUs of the cgroup quota" % (out_settings["cgroup_quota_cpus"]["queries_timed"], quota) if quota < ncpu else "")
                     + ("; queries_timed < 200 per setting (BASELINE.md §2 asks 200: --cpu-protocol full)" if protocol == "bounded" else
                        "; BASELINE.md §2 in full at the reference's thread rule, bounded samples at the other settings" if protocol == "ref_full" else ""),
              ms_per_query=prim["ms_per_query"]["median"], host_cpus=ncpu, host_cpu_quota=quota, best_setting_frames_per_s=best,
              thread_settings=out_settings)
    return cb, parity


def cpp_adapter_leg(smap, queries, n_frames):
    """examples/localize with LOCALIZE_PER_FRAME: the reference's one-frame-per-call pattern through the
    C++ adapter, as a child process (its own handle on the same GPU; started after this process's
    kernels are idle).  Compiled here with g++ -O2."""
    import re
    import tempfile
    from sgtd_amd import evaluate as ev, ingest
    exe = os.path.join(ROOT, "examples", "localize")
    src = os.path.join(ROOT, "examples", "localize.cpp")
    lib_dir = os.path.join(ROOT, "sgtd_amd")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "adapter", "STDesc_shim.hpp"))):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L" + lib_dir, "-lsgtd_accel", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"])
    with tempfile.TemporaryDirectory() as tmp:
        mp, qp = os.path.join(tmp, "map.cache"), os.path.join(tmp, "query.cache")
        ingest.write_cache(mp, smap.xyz, smap.label, np.stack([ev.pose_row(*p) for p in smap.pose]))
        nq = min(n_frames, queries.xyz.shape[0])
        ingest.write_cache(qp, queries.xyz[:nq], queries.label[:nq], np.stack([ev.pose_row(*p) for p in queries.pose[:nq]]))
        out = subprocess.run([exe, mp, qp, str(nq)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, LOCALIZE_PER_FRAME=str(nq)))
    if out.returncode != 0:
        raise RuntimeError(out.stdout[-500:] + out.stderr[-500:])
    m = re.search(r"per-frame calls through STDescManager \((\d+) frames\): ([0-9.]+) ms per frame = BuildSingleScanSTD ([0-9.]+) \+ SearchLoop ([0-9.]+); "
                  r"(\d+)/(\d+) agree with the batched run, ([0-9.]+) inlier pairs per loop", out.stdout)
    if not m:
        raise RuntimeError("unexpected output: " + out.stdout[-500:])
    parts = re.search(r"SearchLoop by part \(ms per frame\): select ([0-9.]+), verify ([0-9.]+), inlier pairs and their entries ([0-9.]+), "
                      r"host fill of loop_std_pair ([0-9.]+)", out.stdout)
    by_part = None
    if parts:
        v = [float(parts.group(i)) for i in range(1, 5)]
        by_part = {"select": v[0], "verify": v[1], "inlier_pairs_and_entries": v[2], "host_fill_of_loop_std_pair": v[3],
                   "device_and_transfers": round(sum(v[:3]), 3)}
    sel = re.search(r"candidate_selector alone: ([0-9.]+) ms per frame for ([0-9.]+) pairs", out.stdout)
    bo = re.search(r"with only the best candidate's loop_std_pair built \(SGTD_SHIM_FILL=best\): ([0-9.]+) ms per frame; (\d+)/(\d+)", out.stdout)
    return {"cpp_adapter_ms_per_frame": float(m.group(2)), "cpp_adapter_search_loop_by_part_ms": by_part,
            "cpp_adapter_ms_candidate_selector_alone": float(sel.group(1)) if sel else None,
            "cpp_adapter_ms_per_frame_best_candidates_list_only": float(bo.group(1)) if bo else None,
            "cpp_adapter_best_only_same_choice_and_list": "%s/%s" % (bo.group(2), bo.group(3)) if bo else None,
            "cpp_adapter_pairs_in_match_lists": float(sel.group(2)) if sel else None, "cpp_adapter_ms_build": float(m.group(3)),
            "cpp_adapter_ms_search_loop": float(m.group(4)), "cpp_adapter_frames": int(m.group(1)),
            "cpp_adapter_agree_with_batched": "%s/%s" % (m.group(5), m.group(6)), "cpp_adapter_inlier_pairs_per_loop": float(m.group(7)),
            "cpp_adapter_note": "examples/localize LOCALIZE_PER_FRAME: BuildSingleScanSTD + SearchLoop (device verification, every "
                                "LOOP_RESULT::loop_std_pair filled) per frame through adapter/STDesc_shim.hpp, g++ -O2; glibc keeps freed "
                                "memory in the heap (mallopt in the example's main; LOCALIZE_DEFAULT_MALLOC=1: about +8 ms per frame)"}


def recall(smap, queries, top1):
    """top-1 candidate (no geometric verification) against the synthetic ground truth:
    exact frame, and map pose within 5 m of the query's true pose (the reference's
    success radius, semantic_graph_localization.cpp:750)"""
    ok = top1 >= 0
    n = top1.shape[0]            # (the first n queries of the set)
    pose = smap.pose[np.clip(top1, 0, smap.pose.shape[0] - 1), :2]
    dist = np.linalg.norm(pose - queries.pose[:n, :2], axis=1)
    return {"top1_is_gt_frame": float(np.mean(ok & (top1 == queries.gt_frame[:n]))),
            "top1_pose_within_5m": float(np.mean(ok & (dist < 5.0)))}


# kernels one step launches, by the form of the passes over the match records (sgtd_stats.select_form) — the
# names the committed PMC row must carry (a row measured on other kernels is refused, loudly)
STEP_KERNELS = {
    0: ("probe_sorted_kernel", "votes_query_kernel", "topk_kernel", "block_count_kernel", "block_write_kernel"),
    1: ("probe_sorted_kernel", "votes_query_kernel", "topk_kernel", "pairs_query_kernel"),
    2: ("probe_sorted_kernel", "votes_topk_kernel", "pairs_query_kernel"),
}
# stage of sgtd_stats' per-kernel times each kernel of a step belongs to
STAGE_OF = (("build_frames", "ms_build"), ("probe_sorted", "ms_probe"), ("resolve_undecided", "ms_probe"), ("votes_", "ms_votes"),
            ("topk_kernel", "ms_topk"), ("block_count", "ms_count"), ("cand_prefix", "ms_count"), ("block_scan", "ms_scan"),
            ("query_base", "ms_scan"), ("block_write", "ms_write"), ("pairs_query", "ms_write"))


def stage_of(kernel):
    for prefix, key in STAGE_OF:
        if kernel.startswith(prefix):
            return key
    return "ms_sort"      # keys, radix passes, scans, group heads, GroupRows, the plan


def check_traffic_row(row, select_form):
    """the committed PMC row describes THIS step only if it was taken on the kernels this run launches"""
    want = STEP_KERNELS.get(select_form)
    if row is None or want is None:
        return
    if row.get("select_form") is not None and row.get("select_form") != select_form:
        raise SystemExit("bench.py: profiles/%s_traffic.json (tag %s, commit %s) was measured with select_form %s, this run launches form %s: "
                         "re-run profiles/collect_r04.sh" % (str(row.get("profile_tag"))[:3], row.get("profile_tag"), row.get("commit"), row.get("select_form"), select_form))
    have = row.get("kernels")
    if have is not None:
        missing = [w for w in want if not any(k.startswith(w) for k in have)]
        if missing:
            raise SystemExit("bench.py: the committed PMC row (tag %s, commit %s) has no kernel named %s — it was measured on other kernels: "
                             "re-run profiles/collect_r04.sh" % (row.get("profile_tag"), row.get("commit"), missing))


def load_traffic(F, N, Q, world):
    """HBM bytes per sweep launch from the committed PMC passes (profiles/r<NN>_traffic.json, written by
    profiles/collect_r<NN>.sh for exactly this configuration; the newest round that has the row), else None"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True):
        try:
            for row in json.load(open(path)):
                if (row.get("frames"), row.get("keypoints"), row.get("queries"), row.get("gpus", 1)) == (F, N, Q, world):
                    return row
        except Exception:
            pass
    return None


FLAT_CONFIG_KEYS = ("delivered_frames_per_s", "one_in_flight_ms_per_step", "overflowed_launches_in_timed_region", "parity_identical_of_200",
                    "verify_ms_per_batch", "skew_frames_per_s", "region_ms_per_step_median", "region_ms_per_step_min", "region_ms_per_step_max",
                    "regions_timed", "cfg1_identical_of_200", "cpp_adapter_ms_per_frame", "R_t", "R_q")
FLAT_CONFIG_KEYS_MULTI = ("table_sharded_frames_per_s", "table_sharded_equals_single_table", "cfg4_frames_per_s", "exchange_exposed_ms")
FLAT_ROOFLINE_KEYS = ("useful_frac", "step_compulsory_frac")


def flatten_evidence(out):
    """The driver's record keeps scalars only (and of nested objects only `config` and `roofline`'s scalar members): what
    the line's nested objects say is repeated here as scalar keys under `config` and `roofline` (VERDICT r5 item 4;
    tests/test_host_cpu.py checks the names)."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d or d[k] is None:
                return None
            d = d[k]
        return d
    c, r = out["config"], out["roofline"]
    tr = out.get("timed_region") or {}
    c["delivered_frames_per_s"] = get(out, "delivered", "frames_per_s")
    c["one_in_flight_ms_per_step"] = get(out, "one_batch_in_flight", "ms_per_step")
    c["overflowed_launches_in_timed_region"] = tr.get("launches_that_overflowed_a_work_buffer")
    for k in ("region_ms_per_step_median", "region_ms_per_step_min", "region_ms_per_step_max", "regions_timed"):
        c[k] = tr.get(k)
    c["parity_identical_of_200"] = get(out, "parity", "identical_candidates_votes_matchlists")
    c["verify_ms_per_batch"] = get(out, "verify", "ms_per_batch")
    c["skew_frames_per_s"] = get(out, "workload_skew", "frames_per_s")
    cfg1 = get(out, "map_size_sweep", "100_cfg1_json_in", "identical_candidates_votes_matchlists")
    c["cfg1_identical_of_200"] = int(str(cfg1).split("/")[0]) if cfg1 else None
    c["cpp_adapter_ms_per_frame"] = get(out, "boundary", "cpp_adapter_ms_per_frame")
    c["R_t"], c["R_q"] = c.get("table_shards_R_t"), c.get("query_groups_R_q")
    if out.get("n_gpus", 1) > 1:
        c["table_sharded_frames_per_s"] = get(out, "table_sharded", "value")
        c["table_sharded_equals_single_table"] = out.get("merged_list_equals_single_table")
        c["cfg4_frames_per_s"] = get(out, "cfg4", "value")
        c["cfg4_leg_wall_s"] = get(out, "cfg4", "leg_wall_s")
        c["exchange_exposed_ms"] = get(out, "scaling_parts", "exchange_exposed_ms")
    # the sweep's USEFUL fraction of the HBM roof: the probe layout once + every 4-byte match record once, over the sweep's
    # live time; and the whole step's compulsory I/O over the step's time
    comp = r.get("compulsory") or {}
    t_probe = (get(r, "kernel_ms", "ms_probe") or 0.0) * 1e-3
    if t_probe > 0 and comp.get("probe_layout_once") is not None and comp.get("match_records_once") is not None:
        r["useful_frac"] = (comp["probe_layout_once"] + comp["match_records_once"]) / t_probe / 1e9 / r["peak"]
    else:
        r["useful_frac"] = None
    r["step_compulsory_frac"] = comp.get("frac")
    return out


def run_steps(step, sync, steps):
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return time.perf_counter() - t0


def claim_stdout():
    """The result line must be the ONLY thing on stdout, and c10d / gloo print from C++ to fd 1 ("[Gloo] Rank 0 is
    connected to ..."): keep the real stdout aside for the result, point fd 1 (and Python's sys.stdout) at stderr."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(os.dup(2), "w")
    return os.fdopen(real, "w")


def last_result_line(text):
    """the last line of `text` that is a bench result (a JSON object with a metric), or None"""
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            try:
                json.loads(line)
                return line
            except ValueError:
                continue
    return None


def modelled_all_gather_ms(bytes_per_rank, ranks):
    """ring all_gather over xGMI, MODELLED (no multi-GPU box here): ranks - 1 steps of bytes_per_rank over one link at
    48 GB/s effective (an xGMI link moves about 64 GB/s one way, MI355X_MICROARCH.md), 8 us per step of latency"""
    if ranks <= 1:
        return 0.0
    return (ranks - 1) * (bytes_per_rank / 48e9 * 1e3 + 0.008)


def skew_leg(args, dev, stream, local_rank, to_dev_flat):
    """the same pipeline on a skewed, reference-shaped workload — Zipf-distributed labels over the 13
    wild classes (get_json_wild.cpp:10-12), 50-400 keypoints per frame, clustered landmarks; F = --frames."""
    import torch
    from sgtd_amd import synth
    from sgtd_amd.manager import STDescManager
    F = args.frames
    smap, world = synth.make_skewed_map(F, stream=31)
    g = STDescManager(device_id=local_rank, max_frame_n=max(20000, F + 1))
    g.set_stream(stream.cuda_stream)
    t0 = time.perf_counter()
    g.add_frames(*to_dev_flat(smap), kp_off=smap.kp_off)
    g.finalize()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    # the largest batch the 32-bit record index and the memory allow (long buckets: more matches per query than the
    # uniform maps give — the estimate from the bucket statistics runs low here, so a small batch is measured first)
    probe = synth.make_skewed_queries(world, 64, stream=3099)
    g.query_frames(*to_dev_flat(probe), kp_off=probe.kp_off, fetch=False)
    g.sync()
    probe_reruns = int(g.stats()["reruns_total"])
    q_safe = int(g.max_batch(int(np.max(np.diff(smap.kp_off)))))
    Q = max(64, min(args.queries, (q_safe * 3 // 4) // 64 * 64 if q_safe >= 128 else 64))
    sets = [synth.make_skewed_queries(world, Q, stream=3100 + b) for b in range(3)]
    dsets = [(to_dev_flat(s), s.kp_off) for s in sets]

    def step(i):
        (x, l), off = dsets[i % len(dsets)]
        g.query_frames(x, l, kp_off=off, fetch=False)
    for i in range(3):          # work buffers reach their size (re-runs happen here, not in the timed steps)
        step(i); g.sync()
    grow_reruns = int(g.stats()["reruns_total"])
    k = max(3, args.steps // 2)
    for attempt in range(3):    # (steps are enqueued back to back: a region in which a batch outgrew a buffer does not count)
        s0 = g.stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        g.sync()
        s1 = g.stats()
        if s1["overflow_launches_total"] == s0["overflow_launches_total"]:
            break
    res = g.results()
    top1 = res.top1()
    last = sets[(k - 1) % len(sets)]
    ok = top1 >= 0
    d = np.linalg.norm(smap.pose[np.clip(top1, 0, F - 1), :2] - last.pose[:, :2], axis=1)
    # frames of 50 .. 400 keypoints: the most votes often go to a neighbouring frame that sees more of the place, so the
    # candidate LIST is what must contain the place (candidate_verify picks from it)
    any_c = []
    for q in range(Q):
        f = res.cand_frame[q, :int(res.n_cand[q])]
        any_c.append(bool((np.linalg.norm(smap.pose[f, :2] - last.pose[q, :2], axis=1) < 5.0).any()) if len(f) else False)
    out = {"workload": "synthetic skewed: Zipf(1.2) labels over 13 classes, 50-400 keypoints/frame, half of the landmarks in clusters; %d-frame map" % F,
           "frames_per_s": Q * k / el, "ms_per_step": 1000.0 * el / k, "queries_per_step": Q, "max_batch_the_record_index_allows": q_safe,
           "keypoints_per_frame_mean": float(np.mean(np.diff(smap.kp_off))), "table_entries": int(s1["n_entries"]), "table_buckets": int(s1["n_buckets"]),
           "bucket_len_sq_over_E": s1["bucket_len_sq_over_E"], "map_build_s": t_build,
           "P_visited_per_query": s1["last_P"] / Q, "P_swept_per_query": s1["last_P_swept"] / Q, "M_matches_per_query": s1["last_M"] / Q,
           "D_descs_per_query": s1["last_D"] / Q, "select_form": int(s1["select_form"]),
           "reruns_in_timed_steps": int(s1["reruns_total"] - s0["reruns_total"]),
           "launches_that_overflowed_in_timed_steps": int(s1["overflow_launches_total"] - s0["overflow_launches_total"]),
           "list_moves_in_timed_steps": int(s1["list_moves_total"] - s0["list_moves_total"]),
           "timed_region_attempts": attempt + 1,
           "reruns_while_the_buffers_grew": grow_reruns, "reruns_of_the_first_64_query_batch": probe_reruns,
           "top1_pose_within_5m": float(np.mean(ok & (d < 5.0))), "some_candidate_within_5m": float(np.mean(any_c))}
    g.close()
    return out


def cfg1_leg(args, dev, stream, local_rank):
    """BASELINE configs[0] / BASELINE.md §2's F = 100 point: graph JSON files in (the producer's format, get_json.cpp:332-341;
    synthetic stand-in, KITTI-00 is not in the image), a 100-frame map, the scans localized one per call on the CPU
    restatement (10 warm-up + 200 timed) and on the GPU (per-frame calls and one batch)."""
    import tempfile
    import torch
    from oracle.oracle import OracleManager
    from sgtd_amd import evaluate as ev, ingest, synth
    from sgtd_amd.manager import STDescManager
    F, NQ, N = 100, 210, args.keypoints
    smap = synth.make_map(F, N, stream=11)
    qs = synth.make_queries(smap, NQ, stream=11)
    with tempfile.TemporaryDirectory() as tmp:
        mp, qp = [], []
        for f in range(F):
            p = os.path.join(tmp, "map_%04d.json" % f)
            ingest.write_graph_json(p, smap.xyz[f], smap.label[f], ev.pose_row(*smap.pose[f]))
            mp.append(p)
        for q in range(NQ):
            p = os.path.join(tmp, "scan_%04d.json" % q)
            ingest.write_graph_json(p, qs.xyz[q], qs.label[q], ev.pose_row(*qs.pose[q]))
            qp.append(p)
        t0 = time.perf_counter()
        gm = ingest.load_graphs(mp)
        gq = ingest.load_graphs(qp)
        t_ingest = time.perf_counter() - t0
    assert gm.n_frames == F and gq.n_frames == NQ and np.array_equal(gm.xyz.reshape(F, N, 3), smap.xyz)
    g = STDescManager(device_id=local_rank)
    g.set_stream(stream.cuda_stream)
    g.add_frames(gm.xyz, gm.label, kp_off=gm.kp_off)
    g.finalize()
    # one scan per call, host pointers in, candidates out (the reference's call pattern)
    qx, ql = gq.xyz.reshape(NQ, N, 3), gq.label.reshape(NQ, N)
    for q in range(10):
        g.query_frames(qx[q:q + 1], ql[q:q + 1])
    t0 = time.perf_counter()
    per = [g.query_frames(qx[q:q + 1], ql[q:q + 1]) for q in range(10, NQ)]
    t_per = (time.perf_counter() - t0) / (NQ - 10)
    g.query_frames(qx[10:], ql[10:])
    t0 = time.perf_counter()
    res = g.query_frames(qx[10:], ql[10:])
    t_batch = time.perf_counter() - t0
    ncpu = os.cpu_count() or 1
    cpu_threads = ncpu - 4 if ncpu > 4 else (2 if ncpu > 3 else 1)       # MP_PROC_NUM rule, CMakeLists.txt:22-41
    o = OracleManager(num_threads=cpu_threads)
    o.add_frames(gm.xyz.reshape(F, N, 3), gm.label.reshape(F, N))
    ms, ident = [], 0
    for q in range(NQ):
        t1 = time.perf_counter()
        o.build(qx[q], ql[q], export=False)
        r = o.select()
        if q >= 10:
            ms.append(1000.0 * (time.perf_counter() - t1))
            k = q - 10
            nc = int(res.n_cand[k])
            same = np.array_equal(res.cand_frame[k, :nc], r["cand_frame"]) and np.array_equal(res.cand_votes[k, :nc], r["cand_votes"])
            same = same and np.array_equal(per[k].cand_frame[0, :nc], r["cand_frame"]) and int(per[k].n_cand[0]) == nc
            if same:
                qi, de = g.result_pairs(k, res)
                same = np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
            ident += int(same)
    g.close()
    return {"workload": "BASELINE configs[0]: one scan per call vs a 100-frame map, graph JSON in (synthetic stand-in in the producer's format)",
            "map_frames": F, "scans": NQ - 10, "ms_ingest_json_%d_files" % (F + NQ): 1000.0 * t_ingest,
            "cpu": {"ms_per_scan": pct(ms), "frames_per_s": 1000.0 / float(np.mean(ms)), "threads": cpu_threads,
                    "kind": "port", "protocol": "BASELINE.md §2: 10 warm-up + 200 timed, build + candidate_selector per scan"},
            "gpu_one_scan_per_call": {"ms_per_scan": 1000.0 * t_per, "frames_per_s": 1.0 / t_per, "note": "host pointers in, candidates out, python ctypes adapter"},
            "gpu_one_batch_of_200": {"ms_per_scan": 1000.0 * t_batch / (NQ - 10), "frames_per_s": (NQ - 10) / t_batch},
            "identical_candidates_votes_matchlists": "%d/%d" % (ident, NQ - 10)}


def rccl_one_child(args):
    """the child process of rccl_one_leg: ONE rank, backend nccl (= RCCL), a Map2D whose collectives are issued although its
    table group and its column have one member (Map2D.force_collective) — communicator set-up, all_gather_into_tensor of the
    packed candidate table on the side stream behind the engine's export event, the merge kernel behind it, the gather of the
    groups' result tables: everything of the exchange but the bytes on the links"""
    result_out = claim_stdout()
    import torch
    import torch.distributed as dist
    from sgtd_amd import synth
    from sgtd_amd.dist import Map2D
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    F, N, Q = args.frames, args.keypoints, args.queries
    smap = synth.make_map(F, N, stream=1)
    sets = [synth.make_queries(smap, Q, stream=1000 + b) for b in range(2)]

    def to_dev(xyz, label):
        return (torch.from_numpy(np.ascontiguousarray(xyz)).to(dev).contiguous(),
                torch.from_numpy(np.ascontiguousarray(label).astype(np.int64)).to(dev).to(torch.int32).contiguous())
    d_map = to_dev(smap.xyz, smap.label)
    d_sets = [to_dev(q.xyz, q.label) for q in sets]
    k = max(4, args.steps // 2)
    out = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "queries_per_step": Q, "map_frames": F, "steps": k,
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "forms": {}}
    for lists in ("all", "winners"):
        m = Map2D(F, 0, 1, r_t=1, device_id=0, lists=lists, max_frame_n=max(20000, F + 1))
        m.force_collective = True
        m.add_shard_frames(*d_map)
        mg = m.mgr

        def with_x(i):
            m.query_async(*d_sets[i % 2])
            with torch.cuda.stream(m.side):
                m.gather_groups()

        def without_x(i):
            mg.query_frames(*d_sets[i % 2], fetch=False)
            if lists == "winners":
                mg.finish_lists(None)

        def clock(step):
            for i in range(2):
                step(i)
            mg.sync(); torch.cuda.synchronize()
            s0 = mg.stats()
            t0 = time.perf_counter()
            for i in range(k):
                step(i)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            mg.sync()
            return 1000.0 * el / k, int(mg.stats()["overflow_launches_total"] - s0["overflow_launches_total"])
        for i in range(4):          # work buffers and the communicator reach their state
            with_x(i); mg.sync()
        torch.cuda.synchronize()
        # alternately, three times each: the smallest of each form (a shared box's noise is one-sided)
        t_w, t_o, ovf = [], [], 0
        for _ in range(3):
            a, o1 = clock(with_x)
            b, o2 = clock(without_x)
            t_w.append(a); t_o.append(b); ovf += o1 + o2
        # the exchange alone, on its stream (the packed table of the last batch is still there)
        with_x(0); mg.sync(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.cuda.stream(m.side):
            ev[0].record()
        for _ in range(10):
            m._exchange()
            with torch.cuda.stream(m.side):
                m.gather_groups()
        with torch.cuda.stream(m.side):
            ev[1].record()
        m.side.synchronize()
        # the merged table of a group of one is the engine's own candidate table
        with_x(0); mg.sync(); torch.cuda.synchronize()
        r = mg.results()
        live = np.arange(m.cand_num)[None, :] < np.asarray(r.n_cand)[:, None]
        same = bool(np.array_equal(m.m_n.cpu().numpy(), r.n_cand) and int(m.m_flags[0].item()) == 0
                    and np.array_equal(m.m_frame.cpu().numpy()[live], np.asarray(r.cand_frame)[live])
                    and np.array_equal(m.m_votes.cpu().numpy()[live], np.asarray(r.cand_votes)[live]))
        out["forms"][lists] = {"ms_per_step_with_exchange": min(t_w), "ms_per_step_without_exchange": min(t_o),
                               "exchange_exposed_ms": max(0.0, min(t_w) - min(t_o)), "all_measurements_ms": {"with": t_w, "without": t_o},
                               "exchange_alone_ms_on_its_stream": ev[0].elapsed_time(ev[1]) / 10,
                               "launches_that_overflowed": ovf, "merged_table_equals_the_engines_own": same,
                               "packed_table_bytes": int(m.packed.numel() * 4)}
        mg.close()
        del m, mg
    if args.rccl_one_shape == "cfg4":
        out["cfg4_shape"] = rccl_one_cfg4_shape(args, dev, N)
    out["note"] = ("one rank over RCCL with every collective of the step issued (group of one: the all-gather copies one table): stream-ordered, "
                   "nothing blocks the host — what the gloo runs on a one-GPU box cannot show; the bytes on the links are not in it "
                   "(modelled under scaling_prediction)")
    dist.destroy_process_group()
    result_out.write(json.dumps({"metric": "rccl_group_of_one", "result": out}) + "\n")
    result_out.flush()


def rccl_one_cfg4_shape(args, dev, N):
    """One rank of cfg4's grid as far as one GPU and one process can stand for it: a quarter of the 100 000-frame
    map (frames 0 .. 24 999: one of R_t = 4 frame-range shards), the whole query group's batch, lists = "winners" — the shard's
    sweep and vote pass, the packed local table out behind votes_topk_kernel, the all-gather through RCCL (group of one: ITS table;
    the three others' arrive as device copies of it here, which costs the local copy engine what three link transfers cost the
    links' DMA — not the links' time), the merge kernel over FOUR tables, the winners' mask, the masked list pass; and the
    verification of the winners with its all-gather.  Times by HIP events on the streams the work runs on."""
    import torch
    import torch.distributed as dist
    from sgtd_amd import synth
    from sgtd_amd.dist import Map2D
    F4, RT, Q = 100000, 4, args.queries
    t0 = time.perf_counter()
    m4 = synth.make_map(F4 // RT, N, stream=4)          # (the shard's frames of a world of its own: the same density and table statistics)
    t_map = time.perf_counter() - t0
    q4 = synth.make_queries(m4, Q, stream=4)

    def to_dev(xyz, label):
        return (torch.from_numpy(np.ascontiguousarray(xyz)).to(dev).contiguous(),
                torch.from_numpy(np.ascontiguousarray(label).astype(np.int64)).to(dev).to(torch.int32).contiguous())

    class OneOfFour(Map2D):
        """a Map2D of a one-rank job that merges FOUR tables: its own through the RCCL all-gather, three copies of it beside"""
        def _buffers(self, nq):
            fresh = nq != self._nq_buf
            super()._buffers(nq)
            if fresh:
                ints = self.packed.numel()
                self.gathered = torch.empty(RT * ints, dtype=torch.int32, device=self.dev)
                self.v_gathered = torch.empty(RT * self.v_local.numel(), dtype=torch.float64, device=self.dev)

        def _exchange(self):
            nq = self.mgr._nq
            ints = self.packed.numel()
            self.mgr.export_wait(self.side.cuda_stream)
            with torch.cuda.stream(self.side):
                dist.all_gather_into_tensor(self.gathered[:ints], self.packed)
                for t in range(1, RT):
                    self.gathered[t * ints:(t + 1) * ints].copy_(self.gathered[:ints], non_blocking=True)
            self.mgr.merge_candidates_dev(self.side.cuda_stream, self.gathered, RT, 0, nq, self.m_frame, self.m_votes, self.m_n,
                                          self.m_src, self.m_keep, self.m_flags)
            self.mgr.export_release(self.side.cuda_stream)
            self.exchanges += 1
    m = OneOfFour(F4 // RT, 0, 1, r_t=1, device_id=0, lists="winners", max_frame_n=F4 + 1)
    m.force_collective = True
    t0 = time.perf_counter()
    m.add_shard_frames(*to_dev(m4.xyz, m4.label))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    x, l = to_dev(q4.xyz, q4.label)
    mg = m.mgr
    for _ in range(3):
        m.query_async(x, l); mg.sync()
    torch.cuda.synchronize()
    k = max(3, args.steps // 2)

    def clock(step):
        step(); mg.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        mg.sync()
        return 1000.0 * el / k

    def without_x():
        mg.query_frames(x, l, fetch=False)
        mg.finish_lists(None)
    t_w = min(clock(lambda: m.query_async(x, l)) for _ in range(2))
    t_o = min(clock(without_x) for _ in range(2))
    # the exchange alone on its stream
    m.query_async(x, l); mg.sync(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(m.side):
        ev[0].record()
    for _ in range(10):
        m._exchange()
    with torch.cuda.stream(m.side):
        ev[1].record()
    m.side.synchronize()
    # the verification of the winners and its all-gather (search_loop's second half), on the engine's stream
    m.query_async(x, l); mg.sync(); torch.cuda.synchronize()
    nq, cn = m.m_frame.shape
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.cuda.stream(m.main):
        ev2[0].record()
        mg.verify_masked(m.m_keep)
        ev2[1].record()
        mg.export_verify(m.v_local[:nq * cn], m.v_local[nq * cn:])
        dist.all_gather_into_tensor(m.v_gathered[:m.v_local.numel()], m.v_local)
        ev2[2].record()
    torch.cuda.synchronize()
    st = mg.stats()
    kept = int(torch.sum(torch.tensor([bin(int(v) & ((1 << 64) - 1)).count("1") for v in m.m_keep.cpu().tolist()])))
    out = {"what": "one rank of R_t = 4 over RCCL (group of one): 25 000 of 100 000 frames, %d query frames per step, lists = winners" % Q,
           "ms_per_step_with_exchange": t_w, "ms_per_step_without_exchange": t_o, "exchange_exposed_ms": max(0.0, t_w - t_o),
           "exchange_alone_ms_on_its_stream": ev[0].elapsed_time(ev[1]) / 10, "packed_table_bytes": int(m.packed.numel() * 4),
           "tables_merged": RT, "winners_kept_of_local_candidates": kept, "verify_masked_ms": ev2[0].elapsed_time(ev2[1]),
           "verify_export_and_all_gather_ms": ev2[1].elapsed_time(ev2[2]), "verified_results_bytes_per_rank": int(m.v_local.numel() * 8),
           "shard_entries": int(st["n_entries"]), "M_matches_per_query_on_the_shard": st["last_M"] / Q,
           "synthetic_map_25000_frames_s": t_map, "shard_build_s": t_build, "frames_per_s_of_the_group_at_this_step": Q / t_w * 1000.0}
    mg.close()
    return out


def rccl_one_leg(args):
    """runs rccl_one_child in a process of its own (a hung communicator set-up must not cost the measured line): a rendezvous
    on 127.0.0.1, WORLD_SIZE = 1"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "8"))      # (as the ranks of an N > 1 run set it: main())
    cmd = [sys.executable, os.path.abspath(__file__), "--rccl-one-child", "--frames", str(args.frames), "--keypoints", str(args.keypoints),
           "--queries", str(args.queries), "--steps", str(args.steps), "--rccl-one-shape", args.rccl_one_shape]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    line = last_result_line(p.stdout)
    if p.returncode != 0 or line is None:
        return {"error": "child exited with %d: %s" % (p.returncode, p.stderr[-600:])}
    return json.loads(line)["result"]


T_PROCESS_START = time.perf_counter()


def main():
    args = parse()
    if args.rccl_one_child:
        return rccl_one_child(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # start the ranks BEFORE anything touches a GPU (no exec after GPU init on this pool):
        # child processes under torch.distributed.run; this process relays the ONE result line and nothing else
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        line = last_result_line(p.stdout)
        if line is not None:
            print(line)
        else:
            sys.stderr.write(p.stdout)
        sys.exit(p.returncode if (p.returncode or line is not None) else 1)
    world = int(env_world or "1")
    if world > 1 and os.environ.get("SGTD_BENCH_SHARE_GPU") != "1":
        # main stream, side stream and RCCL's own stream of every engine in flight: with the runtime's default of four
        # hardware queues some of them share one and the exchange queues up behind the list pass (measured with a group of
        # one: 0.13 ms of the step exposed at 4 queues, 0.04 at 8 — DESIGN.md §4); read by the HIP runtime when it starts.
        # One process per GPU only: two processes with eight queues each on ONE device (the gloo test mode) are
        # time-sliced against each other and run at half the rate (165 k -> 85 k frames/s, gpurun_out/r05an_*).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    result_out = claim_stdout()

    import torch
    import torch.distributed as dist
    from sgtd_amd import synth
    from sgtd_amd.dist import Map2D, plan_2d, shard_range
    from sgtd_amd.manager import STDescManager

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

    # a process group for waits that must not occupy the GPUs (a rank that waits in an RCCL barrier spins a kernel
    # on its device) and for agreeing on success before optional collective stages
    host_group = dist.new_group(backend="gloo") if (world > 1 and backend == "nccl") else None

    def host_barrier():
        if world > 1:
            dist.barrier(group=host_group) if host_group is not None else dist.barrier()

    def all_ok(ok):
        """True on every rank iff `ok` is true on every rank (host-side all-reduce): optional legs run their rank-local
        part under try/except, agree here, and enter the collective part only together"""
        if world == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if host_group is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=host_group)
        else:
            t = t.to(dev) if backend == "nccl" else t
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def make_map_once(n_frames, n_kp, stream_id):
        """the synthetic world of a run with several ranks: rank 0 generates it (and leaves it in synth's cache,
        SGTD_SYNTH_CACHE), the others load it after a barrier — not N copies of the same minute of numpy work
        on one host; without a cache (small maps, cache switched off) every rank generates it itself"""
        if world > 1 and n_frames >= 2000 and os.environ.get("SGTD_SYNTH_CACHE", "x"):
            m, err = None, None
            if rank == 0:
                try:
                    m = synth.make_map(n_frames, n_kp, stream=stream_id)
                except Exception as exc:      # (the barrier below is reached whatever happens here)
                    err = exc
            host_barrier()
            if err is not None:
                raise err
            if rank != 0:
                m = synth.make_map(n_frames, n_kp, stream=stream_id)      # from rank 0's cache file, or generated again without one
            return m
        return synth.make_map(n_frames, n_kp, stream=stream_id)

    F, N, Q = args.frames, args.keypoints, args.queries
    smap = make_map_once(F, N, 1)
    # cold + probe layout of the whole table: ~155 B per descriptor, <= 36*N per frame
    table_bytes = 155.0 * 36 * N * F
    fits_one_gpu = table_bytes < HBM_BYTES / 4
    if world > 1:
        forced = args.rt if args.rt is not None else {"auto": None, "table": world, "query": 1}[args.shard]
        r_t, r_q = plan_2d(world, F, Q, N, r_t=forced)
        mode = "table" if r_t == world else ("query" if r_t == 1 else "2d")
    else:
        r_t, r_q, mode = 1, 1, "single"
    g_idx = rank // r_t
    # query group g serves frames [g Q, (g + 1) Q) of every set; a step serves Q x R_q frames
    n_q_value = Q * r_q
    queries = synth.make_queries(smap, n_q_value, stream=1)
    # The timed region rotates through n_rot batches that differ from each other AND from the warm-up batch
    # (`queries`, which the parity and recall legs use): a node never sees the same frames twice
    # (semantic_graph_localization.cpp:567-604), and the room a match list is given is predicted from the batch BEFORE.
    n_rot = max(1, args.rotate)
    rot_sets = [synth.make_queries(smap, n_q_value, stream=1000 + b) for b in range(n_rot)]
    q_lo, q_hi = g_idx * Q, (g_idx + 1) * Q

    def to_dev(xyz, label):
        return (torch.from_numpy(np.ascontiguousarray(xyz)).to(dev).contiguous(),
                torch.from_numpy(np.ascontiguousarray(label).astype(np.int64)).to(dev).to(torch.int32).contiguous())

    def to_dev_flat(fr):
        return to_dev(fr.xyz, fr.label)

    stream = torch.cuda.current_stream()
    merged = {}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timed_info = {}

    def timed(step, mgr, steps=None, info=None):
        """warm-up on the warm-up batch (incl. work-buffer growth), then K steps between barrier + synchronize, max
        over ranks.  step(i): i < 0 the warm-up batch, i >= 0 the i-th timed step (batch i % n_rot of the rotation).
        Steps are enqueued back to back without a host synchronisation, so a batch that outgrew a work buffer inside
        the region would be overwritten, incomplete, by the next one: such a region does not count — the buffers have
        grown by then, the region is timed again (three attempts, then the run fails)."""
        steps = args.steps if steps is None else steps
        for attempt in range(3):
            for _ in range(max(args.warmup, 1)):
                step(-1)
            mgr.sync()          # grows work buffers if the first batch overflowed them
            step(-1)
            mgr.sync()
            s0 = mgr.stats()
            barrier()
            t0 = time.perf_counter()
            marks = []
            for i in range(steps):
                step(i)
                marks.append(time.perf_counter())
            barrier()
            elapsed = time.perf_counter() - t0
            if os.environ.get("SGTD_BENCH_TRACE"):      # host-side enqueue times of the steps, then the wait for the devices
                sys.stderr.write("[trace rank %d] enqueue ms %s, drain %.2f\n" % (
                    rank, ["%.2f" % (1000.0 * (b - a)) for a, b in zip([t0] + marks[:-1], marks)], 1000.0 * (t0 + elapsed - marks[-1])))
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            mgr.sync()
            s1 = mgr.stats()
            overflowed = int(s1["overflow_launches_total"] - s0["overflow_launches_total"])
            if world > 1:
                t = torch.tensor([overflowed], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                any_overflow = int(t.item())
            else:
                any_overflow = overflowed
            if info is not None:
                info.update(batch_launches_in_timed_region=int(s1["batches_total"] - s0["batches_total"]),
                            launches_that_overflowed_a_work_buffer=overflowed,
                            reruns_in_timed_region=int(s1["reruns_total"] - s0["reruns_total"]),
                            list_pass_reruns_in_timed_region=int(s1["rewrites_total"] - s0["rewrites_total"]),
                            list_moves_in_timed_region=int(s1["list_moves_total"] - s0["list_moves_total"]),
                            device_allocations_in_timed_region=int(s1["device_allocs_total"] - s0["device_allocs_total"]),
                            distinct_batches_in_timed_region=min(n_rot, steps), timed_region_attempts=attempt + 1)
            if any_overflow == 0:
                return elapsed
        sys.exit("bench.py: every attempt at the timed region had a batch that outgrew a work buffer (%d launches in the last): "
                 "its results would have been overwritten incomplete — no valid measurement" % any_overflow)

    class InFlight:
        """`--in-flight` handles over ONE table (sgtd_attach_table: the first owns it, the others borrow it), each with its
        own stream and work buffers; batches go to them in turn, so that the build / sort / plan of one batch runs on the
        device beside the passes over the match records of the batch before.  sync() and the running counters cover all."""
        SUMMED = ("batches_total", "overflow_launches_total", "reruns_total", "rewrites_total", "list_moves_total", "device_allocs_total")

        def __init__(self, owner, n):
            self.h = [owner]
            self.streams = [stream]
            for _ in range(1, n):
                v = STDescManager(device_id=local_rank, max_frame_n=max(20000, F + 1))
                st_ = torch.cuda.Stream(dev)
                v.set_stream(st_.cuda_stream)
                v.attach_table(owner)
                self.h.append(v)
                self.streams.append(st_)
            self.k = 0

        def query_frames(self, x, l):
            self.h[self.k % len(self.h)].query_frames(x, l, fetch=False)
            self.k += 1

        def sync(self):
            for m in self.h:
                m.sync()

        def stats(self):
            out = dict(self.h[0].stats())
            for m in self.h[1:]:
                s_ = m.stats()
                for k in self.SUMMED:
                    out[k] += s_[k]
            return out

    cold = {}
    m2 = None
    flight = None
    if mode == "single":
        mgr = STDescManager(device_id=local_rank, max_frame_n=max(20000, F + 1))
        mgr.set_stream(stream.cuda_stream)
        t0 = time.perf_counter()
        mgr.add_frames(*to_dev(smap.xyz, smap.label))
        mgr.finalize()
        torch.cuda.synchronize()
        cold["map_build_s"] = time.perf_counter() - t0
        d_qxyz, d_qlab = to_dev(queries.xyz, queries.label)
        d_rot = [to_dev(r.xyz, r.label) for r in rot_sets]

        def step(i=-1):
            x, l = (d_qxyz, d_qlab) if i < 0 else d_rot[i % n_rot]
            mgr.query_frames(x, l, fetch=False)
        # cold start: the very first batch of a fresh handle (work buffers sized from the table
        # statistics; re-run if they were too small)
        t0 = time.perf_counter()
        step(); mgr.sync(); torch.cuda.synchronize()
        cold["first_batch_ms"] = 1000.0 * (time.perf_counter() - t0)
        cold["first_batch_overflowed"] = int(mgr.stats()["overflowed"])
        cold["bucket_len_sq_over_E"] = mgr.stats()["bucket_len_sq_over_E"]
    else:
        m2 = Map2D(F, rank, world, r_t=r_t, device_id=local_rank, lists=args.lists, max_frame_n=max(20000, F + 1))
        mgr = m2.mgr
        m2.add_shard_frames(*to_dev(smap.xyz[m2.lo:m2.hi], smap.label[m2.lo:m2.hi]))
        d_qxyz, d_qlab = to_dev(queries.xyz[q_lo:q_hi], queries.label[q_lo:q_hi])
        d_rot = [to_dev(r.xyz[q_lo:q_hi], r.label[q_lo:q_hi]) for r in rot_sets]

        # `--in-flight` engines per rank over the rank's one table (the others attached to the first: sgtd_attach_table),
        # each with its own streams and exchange buffers; steps go to them in turn
        m2s = [m2] + [Map2D(F, rank, world, r_t=r_t, device_id=local_rank, lists=args.lists, attach_to=m2, max_frame_n=max(20000, F + 1))
                      for _ in range(1, max(1, args.in_flight))]
        turn = {"k": 0}

        class Engines:
            def sync(self):
                for m_ in m2s:
                    m_.mgr.sync()

            def stats(self):
                out_ = dict(m2s[0].mgr.stats())
                for m_ in m2s[1:]:
                    s_ = m_.mgr.stats()
                    for k_ in InFlight.SUMMED:
                        out_[k_] += s_[k_]
                return out_
        mgr_all = Engines()

        def step(i=-1):
            """the shard's pipeline on the engine's main stream; exchange (all_gather + merge kernel inside the table group) and
            the gather of the groups' result tables on its side stream: nothing waits on the host"""
            x, l = (d_qxyz, d_qlab) if i < 0 else d_rot[i % n_rot]
            mm = m2s[0 if turn.get("pin") else turn["k"] % len(m2s)]
            turn["k"] += 1
            mm.query_async(x, l)
            with torch.cuda.stream(mm.side):
                merged["out"] = mm.gather_groups()

    if m2 is not None:
        for _ in range(2 * len(m2s)):       # every engine's work buffers reach their size
            step(-1)
        mgr_all.sync(); torch.cuda.synchronize()
    elapsed = timed(step, mgr if m2 is None else mgr_all, info=timed_info)
    if m2 is not None:
        timed_info["batches_in_flight"] = len(m2s)
    one_in_flight = None
    if mode == "single" and args.in_flight > 1:
        # The headline: `--in-flight` batches in flight over the one table.  The measurement above (ONE batch in flight,
        # every kernel of a batch behind the one before) stays beside it.
        one_in_flight = {"frames_per_s": n_q_value * args.steps / elapsed, "ms_per_step": 1000.0 * elapsed / args.steps,
                         "timed_region": dict(timed_info)}
        flight = InFlight(mgr, args.in_flight)

        def fstep_(i=-1):
            x, l = (d_qxyz, d_qlab) if i < 0 else d_rot[i % n_rot]
            flight.query_frames(x, l)
        for _ in range(2 * args.in_flight):      # every handle's work buffers reach their size
            fstep_(-1)
        flight.sync()
        timed_info = {}
        elapsed = timed(fstep_, flight, info=timed_info)
        timed_info["batches_in_flight"] = args.in_flight
    cur_step, cur_mgr = (fstep_, flight) if flight is not None else (step, mgr if m2 is None else mgr_all)
    # the headline region four more times (each with its own warm-up, barrier and overflow check): `value` stays the FIRST
    # region's — the contract's K steps — the spread of the five is reported beside it
    region_ms = [1000.0 * elapsed / args.steps]
    for _ in range(max(0, args.repeat_region - 1)):
        region_ms.append(1000.0 * timed(cur_step, cur_mgr) / args.steps)
    timed_info.update(regions_timed=len(region_ms), region_ms_per_step_all=region_ms, region_ms_per_step_median=float(np.median(region_ms)),
                      region_ms_per_step_min=float(min(region_ms)), region_ms_per_step_max=float(max(region_ms)))
    # the same measurement the way rounds 1-3 took it — ONE batch over and over (the room prediction is then exact, no
    # list ever moves) — beside the headline, which rotates fresh batches
    k_same = max(3, args.steps // 2)
    same_elapsed = timed(lambda i: cur_step(-1), cur_mgr, steps=k_same)
    timed_info.update(same_batch_ms_per_step=1000.0 * same_elapsed / k_same, same_batch_steps=k_same,
                      rotating_over_same_batch=(elapsed / args.steps) / (same_elapsed / k_same))
    # ... and steps that DELIVER: before a handle takes its next batch the host waits for the one it holds (a batch that
    # outgrew a work buffer is re-run there) and takes its candidate tables and list offsets through page-locked arrays
    delivered = None
    if mode == "single":
        hs = flight.h if flight is not None else [mgr]
        held = [False] * len(hs)
        got = {"n": 0, "bytes": 0}

        def take(j):
            if held[j]:
                rd = hs[j].results()
                got["n"] += 1
                got["bytes"] = int(rd.n_cand.nbytes + rd.cand_frame.nbytes + rd.cand_votes.nbytes + rd.pair_off.nbytes)
                held[j] = False

        def dstep(i):
            j = i % len(hs)
            take(j)
            x, l = d_rot[i % n_rot]
            hs[j].query_frames(x, l, fetch=False)
            held[j] = True
        for i in range(len(hs)):
            dstep(i)
        for j in range(len(hs)):
            take(j)
        torch.cuda.synchronize()
        kd = max(4, args.steps // 2)
        got["n"] = 0
        t0 = time.perf_counter()
        for i in range(kd):
            dstep(i)
        for j in range(len(hs)):
            take(j)
        td = time.perf_counter() - t0
        assert got["n"] == kd
        delivered = {"frames_per_s": Q * kd / td, "ms_per_step": 1000.0 * td / kd, "steps": kd, "batches_in_flight": len(hs),
                     "bytes_to_host_per_step": got["bytes"],
                     "note": "every batch's n_cand, candidate frames, votes and list offsets reach the host (sgtd_sync + sgtd_result_candidates through "
                             "page-locked arrays) before its handle takes the next batch; a batch that outgrew a work buffer would be re-run inside"}
        if flight is not None:
            flight.sync()
    if m2 is not None:
        mgr_all.sync()
        turn["pin"] = True          # (everything below works on the rank's first engine)
    step(-1); mgr.sync(); torch.cuda.synchronize()          # leave the handle on the warm-up batch: the parity, recall and profile legs read its results
    backend_ran = dist.get_backend() if world > 1 else None
    collective = {"nccl": "RCCL (torch.distributed backend nccl)", "gloo": "gloo (NOT RCCL: test fallback)"}.get(backend_ran, backend_ran)
    st = mgr.stats()

    # ---- per-kernel timing for the roofline (HIP events on the handle's stream), on the rotating batches
    mgr.set_timing(True)
    acc = {}
    for j in range(args.profile_steps):
        step(j)
        mgr.sync()
        s = mgr.stats()
        for k in KERNEL_KEYS:
            acc.setdefault(k, []).append(s[k])
    mgr.set_timing(False)
    torch.cuda.synchronize()
    kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}
    st = mgr.stats()
    P, M, D = st["last_P"], st["last_M"], st["last_D"]
    select_form = int(st["select_form"])
    t_probe = kern_ms["ms_probe"] * 1e-3
    t_step = elapsed / args.steps
    entry_bytes = st["hbm_bytes_table"] // max(st["n_entries"], 1)   # probe-layout bytes per loaded entry
    P_swept = st.get("last_P_swept") or P
    # (1) The dominant kernel (the sweep) against the HBM roof: bytes per launch from the committed PMC passes of this
    # very command (profiles/collect_r05.sh: separate --pmc passes, 2 x FETCH_SIZE + WRITE_SIZE; both counters sit on
    # the L2's fabric side and count Infinity-Cache hits, so they bound the HBM bytes from above) over the kernel's
    # average launch duration measured live (HIP events on the launch stream).  A row measured on other kernels than
    # the ones this run launches is refused.
    tr = load_traffic(F, N, Q, world)
    check_traffic_row(tr, select_form)
    traffic = tr.get("bytes_per_launch") if tr else None
    hbm_gbs = traffic / t_probe / 1e9 if (traffic and t_probe > 0) else None
    l2_gbs = (entry_bytes * P_swept + 4 * M) / t_probe / 1e9 if t_probe > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "probe_sorted_kernel (sweep)", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs is not None else None, "traffic": traffic,
                "traffic_profile": tr.get("profile_tag") if tr else None, "traffic_profile_commit": tr.get("commit") if tr else None,
                "traffic_note": "PMC bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md §HBM), committed under profiles/ by "
                                "profiles/collect_r05.sh for this command; both counters include Infinity-Cache hits: an upper bound on HBM bytes",
                "avg_launch_ms": kern_ms["ms_probe"], "select_form": select_form,
                "l2_frac": l2_gbs / L2_PEAK_GBS, "l2_GBps_moved_by_the_waves": l2_gbs,
                "valu_issue_frac": tr.get("valu_issue_frac") if tr else None,
                "probe_layout_bytes_per_loaded_entry": entry_bytes,
                "P_visited": P, "P_swept_after_slice_pruning": st.get("last_P_swept"), "M_matches": M,
                "D_query_descs": D, "candidate_pairs": st["last_cand_pairs"], "kernel_ms": kern_ms}
    # (2) The whole step, stage by stage: PMC bytes of the committed profile per stage of sgtd_stats' kernel times, the
    # times measured live in this run
    if tr and tr.get("kernels"):
        stages = {}
        for kname, kv in tr["kernels"].items():
            if kname.startswith("__amd") or kname.startswith("at::"):
                continue        # (memsets and torch's copies: not the library's kernels)
            sg = stages.setdefault(stage_of(kname), {"read_bytes": 0.0, "write_bytes": 0.0, "kernels": []})
            sg["read_bytes"] += kv.get("read_bytes") or 0.0
            sg["write_bytes"] += kv.get("write_bytes") or 0.0
            sg["kernels"].append(kname)
        for key, sg in stages.items():
            ms = kern_ms.get(key, 0.0)
            sg["ms_live"] = ms
            sg["GBps"] = (sg["read_bytes"] + sg["write_bytes"]) / (ms * 1e-3) / 1e9 if ms > 0 else None
            sg["frac_of_hbm_peak"] = sg["GBps"] / HBM_PEAK_GBS if sg["GBps"] else None
        tot = sum(sg["read_bytes"] + sg["write_bytes"] for sg in stages.values())
        roofline["step"] = {"stages": stages, "bytes": tot, "read_bytes": sum(sg["read_bytes"] for sg in stages.values()),
                            "write_bytes": sum(sg["write_bytes"] for sg in stages.values()),
                            "ms_per_step_timed_region": 1000.0 * t_step, "kernel_ms_sum_live": kern_ms["ms_total"],
                            "GBps": tot / t_step / 1e9, "frac": tot / t_step / 1e9 / HBM_PEAK_GBS}
    # (3) What the step cannot avoid moving, against the same roof: the probe layout read once, every match record
    # written once (the reference's M, 4 B each: votes need every one of them), the candidates' match lists out,
    # the keypoints in.  (Without the records — they never leave the device — the compulsory I/O is the first, third
    # and fourth term.)
    comp = {"probe_layout_once": int(st["hbm_bytes_table"]), "match_records_once": int(4 * M), "candidate_pairs_out": int(8 * st["last_cand_pairs"]),
            "keypoints_in": int(16 * N * st["last_queries"])}
    comp_bytes = sum(comp.values())
    roofline["compulsory_bytes"] = comp_bytes
    roofline["compulsory"] = dict(comp, bytes=comp_bytes, io_only_bytes=comp_bytes - comp["match_records_once"],
                                  frac=comp_bytes / t_step / 1e9 / HBM_PEAK_GBS, ms_at_hbm_peak=comp_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
                                  note="frac = compulsory bytes / measured step time / 8 TB/s: how far the whole step is from streaming "
                                       "only what it must")
    # (4) SURVEY §8d's model prices 28 B for every entry the REFERENCE's loop visits; this kernel never loads 86 % of them
    # (pruned sub-cells, four descriptors per visit list), so the figure is not a bound and is kept out of `frac`
    algo_bytes = 28 * P + 64 * D + 8 * M
    roofline["survey_8d_model"] = {"bytes_per_launch": algo_bytes, "GBps": algo_bytes / t_probe / 1e9 if t_probe > 0 else None,
                                   "over_hbm_peak": algo_bytes / t_probe / 1e9 / HBM_PEAK_GBS if t_probe > 0 else None,
                                   "reference_visits_per_s": P / t_probe if t_probe > 0 else 0.0,
                                   "note": "28 B x reference visits + 64 B x descriptors + 8 B x matches: exceeds the HBM peak because most "
                                           "visits are never loaded — not a roofline fraction"}

    def parts_of(km):
        """a rank's step by what scales how: the part every rank of a table group repeats for all of the group's queries
        (descriptor build, home-cell sort, GroupRows, the plan) and the part that shrinks with the shard (the sweep and
        the passes over its records)"""
        return (km["ms_build"] + km["ms_sort"],
                km["ms_probe"] + km["ms_votes"] + km["ms_topk"] + km["ms_count"] + km["ms_scan"] + km["ms_write"])

    def measure_parts(map2d, xl_sets, steps):
        """scaling_parts of a Map2D form: per-rank kernel parts (HIP events on each rank's stream), the exchange timed
        alone (side-stream work of one step), and how much of it the step really waits for: step with exchange minus the
        same step without"""
        mg = map2d.mgr
        mg.set_timing(True)
        a = {}
        for j in range(max(2, args.profile_steps)):
            map2d.query_async(*xl_sets[j % len(xl_sets)])
            mg.sync()
            s_ = mg.stats()
            for k in KERNEL_KEYS:
                a.setdefault(k, []).append(s_[k])
        mg.set_timing(False)
        km = {k: float(np.mean(v)) for k, v in a.items()}
        pre_ms, shard_ms = parts_of(km)
        barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            map2d._exchange()
            with torch.cuda.stream(map2d.side):
                map2d.gather_groups()
        map2d.side.synchronize()
        barrier()
        ex_ms = 1000.0 * (time.perf_counter() - t0) / 10

        def with_x(i):
            map2d.query_async(*xl_sets[i % len(xl_sets)])
            with torch.cuda.stream(map2d.side):
                map2d.gather_groups()

        def without_x(i):
            mg.query_frames(*xl_sets[i % len(xl_sets)], fetch=False)
            if map2d.lists == "winners":
                mg.finish_lists(None)
        t_with = timed(with_x, mg, steps=steps) / steps
        t_without = timed(without_x, mg, steps=steps) / steps
        # a transport that blocks the host inside the all-gather (gloo: the test fallback) keeps the host from feeding the
        # device; the same step with that all-gather issued asynchronously, its merge one step late (Map2D.defer): what is
        # left is the device side of the exchange
        t_defer = None
        if backend_ran == "gloo" and map2d.r_t > 1 and map2d.lists == "all":
            map2d.defer = True
            t_defer = timed(with_x, mg, steps=steps) / steps
            map2d.flush()
            map2d.defer = False
            torch.cuda.synchronize()
        mine = torch.tensor([pre_ms, shard_ms, ex_ms, 1000.0 * t_with, 1000.0 * t_without, 1000.0 * (t_defer if t_defer is not None else t_with)],
                            dtype=torch.float64, device=dev)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        pr = np.array([p_.cpu().numpy() for p_ in allp])
        rt, rq = map2d.r_t, map2d.r_q
        # the same work on ONE GPU: R_q batches, each the replicated part once and every shard's part of group 0
        one_gpu_ms = float(rq * (pr[:rt, 0].mean() + pr[:rt, 1].sum()))
        exposed = float(max(0.0, (pr[:, 3] - pr[:, 4]).max()))
        pred_ms = float((pr[:, 0] + pr[:, 1]).max() + exposed)
        return {"table_shards_R_t": rt, "query_groups_R_q": rq, "lists": map2d.lists,
                "per_rank_ms": {"replicated_build_sort_plan": pr[:, 0].tolist(), "sharded_sweep_and_record_passes": pr[:, 1].tolist(),
                                "exchange_alone_all_gather_merge_result_gather": pr[:, 2].tolist(),
                                "step_with_exchange": pr[:, 3].tolist(), "step_without_exchange": pr[:, 4].tolist()},
                "exchange_exposed_ms": exposed,
                "exchange_exposed_ms_host_not_blocked": float(max(0.0, (pr[:, 5] - pr[:, 4]).max())) if t_defer is not None else None,
                "predicted_ms_per_step": pred_ms, "predicted_one_gpu_ms_for_the_same_frames": one_gpu_ms,
                "predicted_speedup_over_one_gpu": one_gpu_ms / pred_ms,
                "speedup_ceiling_if_the_sharded_part_vanished": one_gpu_ms / float(pr[:, 0].max() + exposed) if rt > 1 else float(rq),
                "note": "kernel times by HIP events on each rank's stream; the build, sort and plan of a group's query frames are repeated on "
                        "every rank of its table group, only the sweep and the passes over its match records shrink with the shard; "
                        "exchange_exposed = step with exchange - step without (the exchange runs on a side stream beside the list pass)"}

    # (5) N > 1: what a rank's step consists of and what that predicts
    scaling_parts = None
    if m2 is not None:
        scaling_parts = measure_parts(m2, [d_rot[b] for b in range(n_rot)], max(3, args.steps // 2))
    step(-1); mgr.sync(); torch.cuda.synchronize()          # back on the warm-up batch

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

    # ---- the drop-in boundary as the reference's caller uses it: one frame per call, host
    # pointers in, descriptors and match lists out (semantic_graph_localization.cpp:590-601)
    boundary = None
    if mode == "single" and args.boundary == "on":
        try:
            nb = min(Q, 48)
            mgr.BuildSingleScanSTD(queries.xyz[0], queries.label[0])   # warm
            t0 = time.perf_counter()
            pairs = 0
            for q in range(nb):
                d = mgr.BuildSingleScanSTD(queries.xyz[q], queries.label[q])
                for c in mgr.candidate_selector(d):
                    pairs += len(c)
            t_frame = (time.perf_counter() - t0) / nb
            # ... and with the table entries of every match fetched too, as the adapter does to rebuild
            # the reference's pair<STDesc, STDesc> lists for candidate_verify on the host
            nb_e = min(nb, 8)
            t0 = time.perf_counter()
            for q in range(nb_e):
                d = mgr.BuildSingleScanSTD(queries.xyz[q], queries.label[q])
                for c in mgr.candidate_selector(d):
                    mgr.fetch_entries(c.db_entry)
            t_frame_e = (time.perf_counter() - t0) / nb_e
            nb2 = min(Q, 128)
            t0 = time.perf_counter()
            r2 = mgr.query_frames(queries.xyz[:nb2], queries.label[:nb2])
            pairs2 = 0
            for q in range(nb2):
                qi, de = mgr.result_pairs(q, r2)
                pairs2 += len(qi)
            t_batch = (time.perf_counter() - t0) / nb2
            boundary = {"per_frame_calls_host_pointers_frames_per_s": 1.0 / t_frame, "ms_per_frame": 1000.0 * t_frame,
                        "frames": nb, "pairs_fetched_per_frame": pairs / nb,
                        "ms_per_frame_with_entries_fetched": 1000.0 * t_frame_e,
                        "batched_host_pointers_all_lists_fetched_frames_per_s": 1.0 / t_batch,
                        "batch": nb2, "pairs_fetched_per_frame_batched": pairs2 / nb2,
                        "note": "python ctypes adapter (sgtd_amd/manager.py); value above excludes PCIe and list fetches"}
        except Exception as exc:
            boundary = {"error": "%s: %s" % (type(exc).__name__, exc)}
        # ... and the same call pattern through the reference-typed C++ adapter (adapter/STDesc_shim.hpp
        # behind include/sgtd/STDescManager.hpp): examples/localize replays BuildSingleScanSTD + SearchLoop
        # frame by frame (semantic_graph_localization.cpp:590-603) as a child process of its own
        try:
            boundary = boundary or {}
            boundary.update(cpp_adapter_leg(smap, queries, min(Q, 64)))
        except Exception as exc:
            boundary["cpp_adapter_error"] = "%s: %s" % (type(exc).__name__, exc)
        step(); mgr.sync()   # leave the handle on the headline batch
        res = mgr.results()

    def entries_of(m):
        ent = torch.tensor([m.stats()["n_entries"]], dtype=torch.int64, device=dev)
        ents = [torch.zeros_like(ent) for _ in range(world)]
        dist.all_gather(ents, ent)
        return [int(e.item()) for e in ents]

    # ---- N > 1: the pure table-sharded form of the same map beside a headline of another form (north_star's wording), with
    # its own scaling parts, and the check that its merged list IS the list of a rank that holds the whole table
    table_sharded, fixed_total, merged_equal = None, None, None
    if m2 is not None:
        step(-1); mgr.sync(); torch.cuda.synchronize()
        head_f, head_v = merged["out"][0][:Q].clone(), merged["out"][1][:Q].clone()      # query group 0's lists of queries[:Q]
        if mode != "table" and args.also_table == "on" and fits_one_gpu:
            ts = Map2D(F, rank, world, r_t=world, device_id=local_rank, lists=args.lists, max_frame_n=max(20000, F + 1))
            ts.add_shard_frames(*to_dev(smap.xyz[ts.lo:ts.hi], smap.label[ts.lo:ts.hi]))
            tq = to_dev(queries.xyz[:Q], queries.label[:Q])
            trot = [to_dev(r.xyz[:Q], r.label[:Q]) for r in rot_sets]

            def tstep(i=-1):
                ts.query_async(*(tq if i < 0 else trot[i % n_rot]))
            t_el = timed(tstep, ts.mgr)
            tparts = measure_parts(ts, trot, max(3, args.steps // 2))
            tf, tv_, _ = ts.query(*tq)
            torch.cuda.synchronize()
            same = bool(torch.equal(tf, head_f) and torch.equal(tv_, head_v))
            table_sharded = {"value": Q * args.steps / t_el, "unit": "frames/s", "ms_per_step": 1000.0 * t_el / args.steps,
                             "queries_per_step": Q, "scaling": "strong", "ranks_in_collective": dist.get_world_size(),
                             "collective": collective, "table_entries_per_rank": entries_of(ts.mgr), "scaling_parts": tparts,
                             "merged_list_equals_headline_list": same}
            if mode == "query":
                merged_equal = same          # (the headline's ranks hold the whole table)
            ts.mgr.close()
            del ts
        elif mode == "table" and fits_one_gpu and args.also_table == "on":
            # explicit --shard table: the replica-per-rank form beside it, the same check the other way round
            rp = Map2D(F, rank, world, r_t=1, device_id=local_rank, max_frame_n=max(20000, F + 1))
            rp.add_shard_frames(*to_dev(smap.xyz, smap.label))
            rq_ = to_dev(queries.xyz[:Q], queries.label[:Q])
            rf, rv, _ = rp.query(*rq_)
            torch.cuda.synchronize()
            merged_equal = bool(torch.equal(rf, head_f) and torch.equal(rv, head_v))
            rp.mgr.close()
            del rp
        # ---- the headline's grid with --queries frames per step in TOTAL (strong scaling of a fixed batch)
        if r_q > 1 and Q // r_q >= 1:
            qs_ = Q // r_q
            fq = [to_dev(r.xyz[g_idx * qs_:(g_idx + 1) * qs_], r.label[g_idx * qs_:(g_idx + 1) * qs_]) for r in [queries] + rot_sets]

            def fstep(i=-1):
                m2.query_async(*fq[0 if i < 0 else 1 + i % n_rot])
                with torch.cuda.stream(m2.side):
                    m2.gather_groups()
            f_el = timed(fstep, mgr)
            fixed_total = {"value": qs_ * r_q * args.steps / f_el, "unit": "frames/s", "ms_per_step": 1000.0 * f_el / args.steps,
                           "queries_per_step_total": qs_ * r_q, "queries_per_group": qs_, "scaling": "strong"}
            step(-1); mgr.sync(); torch.cuda.synchronize()

    # ---- N > 1: BASELINE configs[3], the 100 000-frame map on the grid dist.plan_2d gives it (the reference itself cannot
    # hold it: MAX_FRAME_N = 20 000, STDesc.h:33)
    cfg4 = None
    if world > 1 and (args.cfg4 == "on" or (args.cfg4 == "auto" and world == 8)) and F != 100000:
        # Only rank-local work sits under try/except; the ranks agree on its success (host-side) before every stage
        # that contains a collective, so that one rank's failure (an allocation, a damaged cache file) skips the leg
        # on EVERY rank instead of leaving the others inside an all_gather.
        # (2048 query frames per group: with the match lists named by granules a step's records may pass 2^32 — on a 100 000-frame
        # map that is where batches pay, 29.0 k against 22.9 k frames/s at 768 on one GPU)
        F4, Q4 = 100000, (args.queries if args.cfg4_queries is None else args.cfg4_queries)
        rt4, rq4 = plan_2d(world, F4, Q4, N)
        m4 = q4 = s4 = x4 = l4 = None
        err4 = None
        t_leg4 = time.perf_counter()
        try:
            m4 = make_map_once(F4, N, 4)                  # every rank holds the world, keeps its frames (its barrier is always reached)
        except Exception as exc:
            err4 = exc
        t_map4 = time.perf_counter() - t_leg4
        if all_ok(err4 is None):
            try:
                q4 = synth.make_queries(m4, Q4 * rq4, stream=4)
                s4 = Map2D(F4, rank, world, r_t=rt4, device_id=local_rank, lists=args.lists, max_frame_n=F4 + 1)
                s4.add_shard_frames(*to_dev(m4.xyz[s4.lo:s4.hi], m4.label[s4.lo:s4.hi]))
                x4, l4 = to_dev(q4.xyz[s4.g * Q4:(s4.g + 1) * Q4], q4.label[s4.g * Q4:(s4.g + 1) * Q4])
                for _ in range(2):                       # the shard's own sweep without the exchange: work buffers reach their size here
                    s4.mgr.query_frames(x4, l4, fetch=False)
                    s4.mgr.sync()
            except Exception as exc:
                err4 = exc
        if all_ok(err4 is None):
            out4 = {}

            def step4(i=-1):
                s4.query_async(x4, l4)
                with torch.cuda.stream(s4.side):
                    out4["out"] = s4.gather_groups()
            e4 = timed(step4, s4.mgr)
            top1 = out4["out"][0][:, 0].cpu().numpy()
            cfg4 = {"workload": "synthetic %d keypoints/frame, %d-frame map: %d table shards (frame ranges) x %d query groups over %d GPUs" % (N, F4, rt4, rq4, world),
                    "value": Q4 * rq4 * args.steps / e4, "unit": "frames/s", "ms_per_step": 1000.0 * e4 / args.steps,
                    "queries_per_step": Q4 * rq4, "table_shards_R_t": rt4, "query_groups_R_q": rq4,
                    "table_entries_per_rank": entries_of(s4.mgr), "collective": collective,
                    "recall": recall(m4, q4, top1),
                    "map_generation_s_rank0_then_cache": t_map4, "leg_wall_s": time.perf_counter() - t_leg4}
        else:
            cfg4 = {"error": "skipped on every rank: %s" % ("%s: %s" % (type(err4).__name__, err4) if err4 is not None else "another rank failed")}
        del s4, m4, q4, x4, l4

    # ---- N > 1: ONE process over all N devices (sgtd_create_multi: frame blocks dealt to the devices,
    # concurrent sweeps, host-side merge with the same rule) — the form the reference's C++ caller uses.
    # Rank 0 drives it while the other ranks wait at the barrier behind it.
    multi_handle = None
    want_multi = world > 1 and (args.multi_handle == "on" or (args.multi_handle == "auto" and torch.cuda.device_count() >= world))
    if want_multi:
        if rank == 0:
            try:
                gm = STDescManager(devices=list(range(world)), max_frame_n=max(20000, F + 1))
                gm.add_frames(smap.xyz, smap.label)            # host pointers (the multi-device handle's contract)
                gm.finalize()
                hq_xyz, hq_lab = np.ascontiguousarray(queries.xyz[:Q]), np.ascontiguousarray(queries.label[:Q])
                for _ in range(2):
                    gm.query_frames(hq_xyz, hq_lab, fetch=False); gm.sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    gm.query_frames(hq_xyz, hq_lab, fetch=False)
                    gm.sync()
                tm = time.perf_counter() - t0
                rg = gm.results()
                ref_f = merged["out"][0][:Q].cpu().numpy()
                same = bool(np.array_equal(rg.cand_frame[:, :ref_f.shape[1]], ref_f))
                multi_handle = {"value": Q * args.steps / tm, "unit": "frames/s", "ms_per_step": 1000.0 * tm / args.steps,
                                "devices": gm.device_count, "queries_per_step": Q,
                                "note": "one process, host pointers in (PCIe inside the timed region), per-device sweeps concurrent, host merge; "
                                        "the other ranks' processes idle on the host meanwhile, their tables still resident",
                                "candidates_equal_headline_list": same}
                gm.close()
            except Exception as exc:
                multi_handle = {"error": "%s: %s" % (type(exc).__name__, exc)}
        host_barrier()      # (the other ranks wait on the host: no barrier kernel spins on the devices rank 0 is timing)

    # ---- another batch size on the same map (beside the headline, which stays at --queries): 4096 query frames per step
    # amortise the per-batch work (sort, plan, passes of four descriptors per home cell) over twice the frames
    batch_sweep = None
    if mode == "single" and args.sweep not in ("", "none") and Q == 2048:
        try:
            qb = [synth.make_queries(smap, 4096, stream=2000 + b) for b in range(2)]
            db = [to_dev(x.xyz, x.label) for x in qb]

            def sb(i=-1):
                mgr.query_frames(*db[i % 2], fetch=False)
            eb = timed(sb, mgr, steps=max(3, args.steps // 2))
            batch_sweep = {"4096": {"frames_per_s": 4096 * max(3, args.steps // 2) / eb, "ms_per_step": 1000.0 * eb / max(3, args.steps // 2)},
                           str(Q): {"frames_per_s": Q * args.steps / elapsed, "ms_per_step": 1000.0 * elapsed / args.steps}}
            del db, qb
            step(-1); mgr.sync()
        except Exception as exc:
            batch_sweep = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- N = 1: what rank 0 of a --predict-world job does, measured here part by part (no multi-GPU box in this round's
    # builder runs: the per-rank parts are measured, the all_gather's time is modelled and says so)
    prediction = None
    if mode == "single" and args.predict_world > 1 and args.sweep != "none":
        try:
            W = args.predict_world
            cn = mgr.config_setting_["candidate_num"]
            pre1, shard1 = parts_of(kern_ms)
            forms = {}
            for name, rt in (("auto_2d", plan_2d(W, F, Q, N)[0]), ("pure_table", W)):
                rq = W // rt
                if rt == 1:
                    pre_r, shard_r, entries_r = pre1, shard1, int(st["n_entries"])
                    mm = None
                else:
                    lo, hi = shard_range(F, rt, 0)
                    mm = STDescManager(device_id=local_rank, max_frame_n=max(20000, F + 1))     # rank 0's shard as a table of its own
                    mm.set_stream(stream.cuda_stream)
                    mm.add_frames(*to_dev(smap.xyz[lo:hi], smap.label[lo:hi]))
                    mm.finalize()
                    for j in range(3):
                        mm.query_frames(*d_rot[j % n_rot], fetch=False); mm.sync()
                    mm.set_timing(True)
                    a = {}
                    for j in range(3):
                        mm.query_frames(*d_rot[j % n_rot], fetch=False); mm.sync()
                        s_ = mm.stats()
                        for k in KERNEL_KEYS:
                            a.setdefault(k, []).append(s_[k])
                    mm.set_timing(False)
                    pre_r, shard_r = parts_of({k: float(np.mean(v)) for k, v in a.items()})
                    entries_r = int(mm.stats()["n_entries"])
                # the merge kernel over rt packed tables of Q queries, timed alone on this GPU
                ints = 2 * Q * cn + 4
                # (tables as full as real ones: cn qualifying candidates per query and table, distinct frames, interleaved vote
                # counts — the kernel's rounds are what it costs; round 5's first figures were taken on empty tables: no rounds)
                fake = torch.zeros(rt * ints, dtype=torch.int32, device=dev)
                kk = torch.arange(cn, dtype=torch.int32, device=dev)
                for t_ in range(rt):
                    b_ = t_ * ints
                    fake[b_:b_ + Q * cn] = (kk + t_ * 100000).repeat(Q)
                    fake[b_ + Q * cn:b_ + 2 * Q * cn] = (5 + (cn - kk) * rt + t_).repeat(Q)
                    fake[b_ + 2 * Q * cn + 1] = Q
                    fake[b_ + 2 * Q * cn + 2] = cn
                outs = [torch.empty((Q, cn), dtype=torch.int32, device=dev) for _ in range(3)] + \
                       [torch.empty(Q, dtype=torch.int32, device=dev), torch.empty(Q, dtype=torch.int64, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    mgr.merge_candidates_dev(stream.cuda_stream, fake, rt, 0, Q, outs[0], outs[1], outs[3], outs[2], outs[4], outs[5])
                torch.cuda.synchronize()
                merge_ms = 1000.0 * (time.perf_counter() - t0) / 20
                assert int(outs[3].min().item()) == cn, "the timed merge must pick candidate_num candidates per query"
                ag_ms = modelled_all_gather_ms(ints * 4, rt) + modelled_all_gather_ms(2 * Q * cn * 4, rq)
                one_gpu_ms = rq * (pre1 + shard1)
                # the exchange runs beside the list pass (lists "all"): only what outlasts that pass would be waited for
                exposed = max(0.0, merge_ms + ag_ms - kern_ms["ms_write"] * (shard_r / shard1 if rt > 1 else 1.0)) if rt > 1 else 0.0
                pred_ms = pre_r + shard_r + exposed
                forms[name] = {"table_shards_R_t": rt, "query_groups_R_q": rq, "frames_per_step": Q * rq, "table_entries_rank0": entries_r,
                               "rank0_ms": {"replicated_build_sort_plan": pre_r, "sharded_sweep_and_record_passes": shard_r,
                                            "merge_kernel_alone": merge_ms, "all_gathers_MODELLED": ag_ms, "exchange_exposed": exposed},
                               "predicted_ms_per_step": pred_ms, "predicted_frames_per_s": Q * rq / pred_ms * 1e3,
                               "one_gpu_ms_for_the_same_frames": one_gpu_ms, "predicted_speedup_over_one_gpu": one_gpu_ms / pred_ms,
                               "scaling": "weak" if rq > 1 else "strong"}
                if mm is not None:
                    mm.close()
                    del mm
            prediction = {"world": W, "forms": forms,
                          "note": "rank 0's kernel parts measured on THIS GPU (its shard, the group's %d query frames per step; the other ranks' shards "
                                  "are the same size); the merge kernel timed alone; the all_gathers' times are a MODEL of a ring over xGMI "
                                  "(48 GB/s per link, 8 us per step), not a measurement — no step of this path has run on two physical GPUs "
                                  "from the builder's side; the driver's SCALE run measures it" % Q}
            step(-1); mgr.sync()
        except Exception as exc:
            prediction = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- other map sizes (same batch, same pipeline), N = 1 only
    sweep = None
    if mode == "single" and args.sweep not in ("", "none"):
        sweep = {str(F): {"frames_per_s": Q * args.steps / elapsed, "ms_per_step": 1000.0 * elapsed / args.steps}}
        for f2 in [int(x) for x in args.sweep.split(",") if x]:
            if f2 == F:
                continue
            try:
                m2_ = synth.make_map(f2, N, stream=1)
                q2 = synth.make_queries(m2_, Q, stream=1)
                g2 = STDescManager(device_id=local_rank, max_frame_n=max(20000, f2 + 1))
                g2.set_stream(stream.cuda_stream)
                g2.add_frames(*to_dev(m2_.xyz, m2_.label))
                g2.finalize()
                x2, l2 = to_dev(q2.xyz, q2.label)

                def s2():
                    g2.query_frames(x2, l2, fetch=False)
                s2(); g2.sync(); s2(); g2.sync()
                assert g2.stats()["overflowed"] == 0
                torch.cuda.synchronize()
                k2 = max(3, args.steps // 2)
                t2 = run_steps(s2, torch.cuda.synchronize, k2)
                r2 = g2.results()
                sweep[str(f2)] = {"frames_per_s": Q * k2 / t2, "ms_per_step": 1000.0 * t2 / k2,
                                  "top1_pose_within_5m": recall(m2_, q2, r2.top1())["top1_pose_within_5m"],
                                  "table_entries": g2.stats()["n_entries"]}
                g2.close()
                del g2, x2, l2
            except Exception as exc:
                sweep[str(f2)] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if args.cfg1 == "on":
            try:
                sweep["100_cfg1_json_in"] = cfg1_leg(args, dev, stream, local_rank)
            except Exception as exc:
                sweep["100_cfg1_json_in"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- the skewed, reference-shaped workload (N = 1)
    skew = None
    if mode == "single" and args.skew == "on" and args.sweep != "none":
        try:
            skew = skew_leg(args, dev, stream, local_rank, to_dev_flat)
        except Exception as exc:
            skew = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- incremental insert: 100 frames appended to the finalized map are
    # sorted into a tail segment (cost proportional to the appended entries), the batch then
    # sweeps main + tail
    incremental = None
    if mode == "single" and args.sweep != "none":
        try:
            # (ms_finalize is wall time and includes the device allocations of a handle's first build:
            # after the host-pointer legs above those have been seen to stall for seconds once, so
            # a throw-away handle takes that hit and the second one is timed)
            full_ms = None
            for attempt in range(2):
                g3 = STDescManager(device_id=local_rank, max_frame_n=max(20000, F + 200))
                g3.set_stream(stream.cuda_stream)
                g3.add_frames(*to_dev(smap.xyz, smap.label))
                g3.sync()
                g3.finalize()
                ms = g3.stats()["ms_finalize"]
                full_ms = ms if full_ms is None else min(full_ms, ms)
                if attempt == 0:
                    g3.close()

            def s3():
                g3.query_frames(d_qxyz, d_qlab, fetch=False)
            s3(); g3.sync(); s3(); g3.sync(); torch.cuda.synchronize()
            t_main = run_steps(s3, torch.cuda.synchronize, 3) / 3
            g3.add_frames(*to_dev(smap.xyz[:100], smap.label[:100]))     # 100 more frames (ids F .. F+99)
            g3.finalize()
            s3_st = g3.stats()
            s3(); g3.sync(); torch.cuda.synchronize()
            t_tail = run_steps(s3, torch.cuda.synchronize, 2) / 2     # (a tail that stays unchanged for 4 batches is merged)
            tail_after = g3.stats()["tail_entries"]
            for _ in range(3):
                s3()
            g3.sync()
            merged_after = g3.stats()["tail_entries"]
            incremental = {"appended_frames": 100, "tail_entries": s3_st["tail_entries"], "table_entries": s3_st["n_entries"],
                           "ms_finalize_full_table": full_ms, "ms_finalize_after_append": s3_st["ms_finalize"],
                           "ms_per_step_one_segment": 1000.0 * t_main, "ms_per_step_main_plus_tail": 1000.0 * t_tail,
                           "tail_entries_while_timed": tail_after, "tail_entries_after_4_more_batches": merged_after}
            g3.close()
            del g3
        except Exception as exc:
            incremental = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- the exchange through RCCL itself, as far as one GPU goes (a child process: one rank, a group of one)
    rccl_one = None
    if mode == "single" and args.rccl_one == "on" and args.sweep != "none":
        try:
            rccl_one = rccl_one_leg(args)
        except Exception as exc:
            rccl_one = {"error": "%s: %s" % (type(exc).__name__, exc)}

    ranks_seen = 1
    entries_per_rank = [int(st["n_entries"])]
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        entries_per_rank = entries_of(mgr)

    out = None
    if rank == 0:
        value = n_q_value * args.steps / elapsed
        if mode == "single":
            sharding = "none"
        else:
            sharding = ("%d table shards (map frames by range) x %d query groups over %d GPUs: every rank sweeps its shard with its group's %d query frames; "
                        "per group one all_gather of the packed top-50 tables + the merge kernel on a side stream, one all_gather of the groups' "
                        "result tables; collectives over %s" % (r_t, r_q, world, Q, collective))
        out = {
            "metric": "query frames/sec vs map size (descriptor build + candidate selection)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if (world == 1 or r_q > 1) else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "arithmetic": "every decision equals the reference's f64 result: the sweep pre-tests in f32 against two "
                          "conservative thresholds and decides the few entries between them on the exact f64 sides",
            "config": {"workload": "synthetic %d keypoints/frame, %d-frame map, descriptor build+match (BASELINE north-star point: 10k-frame map, 1 GPU)" % (N, F)
                       if F == 10000 else "synthetic %d keypoints/frame, %d-frame map, descriptor build+match" % (N, F),
                       "map_frames": F, "keypoints_per_frame": N, "queries_per_step": n_q_value, "queries_per_query_group": Q,
                       "table_shards_R_t": r_t, "query_groups_R_q": r_q, "lists": args.lists if world > 1 else None,
                       "sharding": sharding, "mode": mode, "collective_backend": backend_ran, "queries_this_rank": q_hi - q_lo,
                       "ranks_in_collective": ranks_seen, "table_entries_per_rank": entries_per_rank,
                       "table_entries_this_rank": st["n_entries"], "table_buckets_this_rank": st["n_buckets"],
                       # (under `config` so that a parser that keeps only the contract's keys still carries them)
                       "timed_region": dict(timed_info, warmup_batch="a batch of its own: no timed step repeats it",
                                            note="the timed steps rotate through %d distinct synthetic batches; a region in which a batch outgrew a work "
                                                 "buffer is discarded and timed again (timed_region_attempts); moved match lists are counted; "
                                                 "same_batch_ms_per_step = rounds 1-3's measurement (one batch over and over) in the same run" % n_rot),
                       "delivered": delivered},
            "roofline": roofline,
        }
        out["timed_region"] = out["config"]["timed_region"]
        if scaling_parts is not None:
            out["scaling_parts"] = scaling_parts
        if prediction is not None:
            out["scaling_prediction"] = prediction
        if delivered is not None:
            out["delivered"] = delivered
        if one_in_flight is not None:
            out["one_batch_in_flight"] = one_in_flight
            out["config"]["batches_in_flight"] = args.in_flight
        if cold:
            out["cold_start"] = cold
        if sweep is not None:
            out["map_size_sweep"] = sweep
        if batch_sweep is not None:
            out["batch_size_sweep"] = batch_sweep
        if skew is not None:
            out["workload_skew"] = skew
        if verify is not None:
            out["verify"] = verify
        if boundary is not None:
            out["boundary"] = boundary
        if incremental is not None:
            out["incremental_insert"] = incremental
        if rccl_one is not None:
            out["rccl_group_of_one"] = rccl_one
        if table_sharded is not None:
            out["table_sharded"] = table_sharded
        if fixed_total is not None:
            out["fixed_total_batch"] = fixed_total
        if merged_equal is not None:
            out["merged_list_equals_single_table"] = merged_equal
        if cfg4 is not None:
            out["cfg4"] = cfg4
        if multi_handle is not None:
            out["multi_device_handle"] = multi_handle
        if mode == "single":
            out["recall"] = recall(smap, queries, res.top1())
        else:
            out["recall"] = recall(smap, queries, merged["out"][0][:, 0].cpu().numpy())
        want_cpu = args.cpu_baseline == "on" or (args.cpu_baseline == "auto" and F <= 12000)
        if mode == "single" and want_cpu:
            try:
                cb, par = cpu_baseline(smap, queries, res, mgr, args.cpu_seconds, args.cpu_protocol)
                out["cpu_baseline"] = cb
                out["parity"] = par
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
                out["speedup_vs_best_cpu_setting"] = value / cb["best_setting_frames_per_s"]
            except Exception as exc:   # a broken checker must not cost the measured line
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = "%s: %s" % (type(exc).__name__, exc)
        else:
            out["cpu_baseline"] = None
    if rank == 0:
        out["config"]["wall_s_of_this_rank_until_the_line"] = time.perf_counter() - T_PROCESS_START
        flatten_evidence(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()


if __name__ == "__main__":
    main()
