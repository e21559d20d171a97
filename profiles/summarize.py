#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/collect_r02.sh into the small files that
are committed under profiles/: per-kernel duration stats, per-kernel PMC means,
<tag>_probe_traffic.json and a row of <round>_traffic.json (HBM bytes per sweep launch, read by bench.py).

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md §HBM:
FETCH_SIZE (KB) counts 128-B requests as 64 B for wide coalesced reads (the
sweep reads 16 B per lane), so the read side is doubled; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def pmc_means(pattern, reduce="mean"):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if reduce == "median":      # steady-state launches: the first, aborted (work-buffer overflow) ones do not count
        return {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in out.items()}
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    if tag[:3] >= "r04":      # round 4 on: per kernel and per step (profiles/collect_r04.sh), summarised by summarize_step.py
        import summarize_step
        return summarize_step.main()
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(here, "%s_kernel_stats.csv" % tag))
    b = os.path.join(src, "bench_under_stats.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(here, "%s_bench_under_rocprof.json" % tag))
    pmc = {}
    for name in ("fetch", "write", "tcc"):
        for k, cs in pmc_means(os.path.join(src, name, "*", "*_counter_collection.csv")).items():
            pmc.setdefault(k, {}).update(cs)
    keep = {k: v for k, v in pmc.items() if any(s in k for s in
            ("probe", "plan_passes", "resolve", "block_", "votes", "topk", "build_frames", "locality", "radix", "query_base"))}
    json.dump(keep, open(os.path.join(here, "%s_pmc_means.json" % tag), "w"), indent=1, sort_keys=True)
    med = {}
    for name in ("fetch", "write", "tcc"):
        for k, cs in pmc_means(os.path.join(src, name, "*", "*_counter_collection.csv"), "median").items():
            med.setdefault(k, {}).update(cs)
    names = [k for k in med if k.startswith("probe_kernel") or k.startswith("probe_sorted_kernel")]
    # the variant the steady-state launches use = the one that moves the most bytes
    names.sort(key=lambda k: -(med[k].get("FETCH_SIZE", 0.0) + med[k].get("WRITE_SIZE", 0.0)))
    if names and os.path.exists(b):
        cfg = json.loads(open(b).read().strip().splitlines()[-1])["config"]
        p = med[names[0]]
        fetch_kb, write_kb = p.get("FETCH_SIZE", 0.0), p.get("WRITE_SIZE", 0.0)
        traffic = {"frames": cfg["map_frames"], "queries": cfg["queries_per_step"], "keypoints": cfg["keypoints_per_frame"],
                   "gpus": 1, "kernel": names[0], "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                   "reduction": "median over the launches of the run",
                   "correction": "read bytes = 2 * FETCH_SIZE (gfx950 wide-load undercount), write bytes = WRITE_SIZE",
                   "bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
                   "TCC_HIT_sum": p.get("TCC_HIT_sum"), "TCC_MISS_sum": p.get("TCC_MISS_sum")}
        json.dump(traffic, open(os.path.join(here, "%s_probe_traffic.json" % tag), "w"), indent=1, sort_keys=True)
        # bench.py reads the rows of <round>_traffic.json (one per measured configuration), newest round first
        rows_path = os.path.join(here, "%s_traffic.json" % tag[:3])
        rows = []
        if os.path.exists(rows_path):
            try:
                rows = json.load(open(rows_path))
            except Exception:
                rows = []
        key = (traffic["frames"], traffic["keypoints"], traffic["queries"], traffic["gpus"])
        rows = [r for r in rows if (r.get("frames"), r.get("keypoints"), r.get("queries"), r.get("gpus", 1)) != key]
        traffic["profile_tag"] = tag
        # steady-state duration of the sweep in the kernel trace of the --stats pass (the stats
        # file's average also counts the short launches of other table segments / re-runs)
        tr = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
        if tr:
            dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr[0]))
                         if r["Kernel_Name"].startswith("void probe_sorted_kernel"))
            if dur:
                traffic["sweep_median_ms_kernel_trace"] = dur[len(dur) // 2]
                traffic["sweep_launches_in_trace"] = len(dur)
        rows.append(traffic)
        json.dump(rows, open(rows_path, "w"), indent=1, sort_keys=True)
        print(json.dumps(traffic))


if __name__ == "__main__":
    main()
