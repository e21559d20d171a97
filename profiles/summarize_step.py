#!/usr/bin/env python3
"""Per-kernel and whole-step traffic of bench.py's default step from the rocprofv3 passes of
profiles/collect_r05.sh:  python3 profiles/summarize_step.py <out dir> <tag> <commit>

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats summary), <tag>_step.json (per kernel
and step: launches, median duration in the kernel trace, HBM-side bytes from the FETCH_SIZE and WRITE_SIZE
passes) and the row of profiles/<round>_traffic.json that bench.py reads for `roofline`.

A STEP is cut out of each pass's dispatch sequence: everything from the first build_frames_kernel after the
previous step's last kernel up to the step's last kernel (pairs_query_kernel or block_write_kernel); a
kernel's bytes of a step are summed over its launches in that step (the radix passes, the scans), and the
step's figure is the median over the steady-state steps (the run's last ones: no map construction, no
first-batch re-run among them).

Bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: read = 2 x FETCH_SIZE (gfx950 tallies the 128-B
requests of wide loads at 64 B), write = WRITE_SIZE; both counters sit on the L2's fabric side, so
Infinity-Cache hits are counted: upper bounds on HBM bytes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

LAST = ("pairs_query_kernel", "block_write_kernel")
FIRST = "build_frames_kernel"
STEADY = 4          # steady-state steps taken from the end of a run


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def steps_of(rows, value_of):
    """rows: dicts in dispatch order with 'name'; returns a list of steps, each {kernel: [values]}"""
    steps, cur, open_ = [], None, False
    for r in rows:
        n = r["name"]
        if n.startswith(FIRST) and not open_:
            cur, open_ = collections.defaultdict(list), True
        if open_:
            cur[n].append(value_of(r))
            if n.startswith(LAST):
                steps.append(cur)
                open_ = False
    return steps


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def per_kernel(steps):
    steps = steps[-STEADY:]
    names = sorted({k for s in steps for k in s})
    return {k: {"launches_per_step": median([len(s.get(k, [])) for s in steps]),
                "per_step": median([sum(s.get(k, [])) for s in steps])} for k in names}


def counter_steps(src, sub, counter):
    out = []
    for path in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        rows = [dict(name=short(r["Kernel_Name"]), id=int(r["Dispatch_Id"]), v=float(r["Counter_Value"]))
                for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: r["id"])
        out = steps_of(rows, lambda r: r["v"])
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else None
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(here, "%s_kernel_stats.csv" % tag))
    b = os.path.join(src, "bench_under_stats.json")
    line = None
    if os.path.exists(b):
        shutil.copy(b, os.path.join(here, "%s_bench_under_rocprof.json" % tag))
        line = json.loads(open(b).read().strip().splitlines()[-1])
    # durations from the kernel trace
    dur = {}
    tr = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
    if tr:
        rows = [dict(name=short(r["Kernel_Name"]), t0=int(r["Start_Timestamp"]), ms=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                for r in csv.DictReader(open(tr[0]))]
        rows.sort(key=lambda r: r["t0"])
        dur = per_kernel(steps_of(rows, lambda r: r["ms"]))
    fetch = per_kernel(counter_steps(src, "fetch", "FETCH_SIZE"))
    write = per_kernel(counter_steps(src, "write", "WRITE_SIZE"))
    hit = per_kernel(counter_steps(src, "tcc", "TCC_HIT_sum"))
    miss = per_kernel(counter_steps(src, "tcc", "TCC_MISS_sum"))
    kernels = {}
    for k in sorted(set(dur) | set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * fetch[k]["per_step"] if k in fetch else None
        wr = 1024.0 * write[k]["per_step"] if k in write else None
        kernels[k] = {"launches_per_step": (dur.get(k) or fetch.get(k) or write.get(k))["launches_per_step"],
                      "ms_per_step_kernel_trace": dur[k]["per_step"] if k in dur else None,
                      "read_bytes": rd, "write_bytes": wr,
                      "l2_hit_rate": (hit[k]["per_step"] / max(hit[k]["per_step"] + miss[k]["per_step"], 1.0)) if k in hit and k in miss else None}
    tot_rd = sum(v["read_bytes"] or 0.0 for v in kernels.values())
    tot_wr = sum(v["write_bytes"] or 0.0 for v in kernels.values())
    tot_ms = sum(v["ms_per_step_kernel_trace"] or 0.0 for v in kernels.values())
    step = {"profile_tag": tag, "commit": commit, "kernels": kernels,
            "step": {"read_bytes": tot_rd, "write_bytes": tot_wr, "bytes": tot_rd + tot_wr, "kernel_ms_sum": tot_ms},
            "correction": "read bytes = 2 x FETCH_SIZE (gfx950 wide-load undercount), write bytes = WRITE_SIZE; both count Infinity-Cache "
                          "hits (fabric-side counters): upper bounds on HBM bytes",
            "reduction": "per kernel: sum over its launches inside a step, median over the last %d steps of the run" % STEADY}
    if line:
        cfg = line["config"]
        step.update(frames=cfg["map_frames"], keypoints=cfg["keypoints_per_frame"], queries=cfg["queries_per_step"], gpus=line.get("n_gpus", 1),
                    select_form=(line.get("roofline") or {}).get("select_form"))
    json.dump(step, open(os.path.join(here, "%s_step.json" % tag), "w"), indent=1, sort_keys=True)
    # the row bench.py reads
    sweep = [k for k in kernels if k.startswith("probe_sorted_kernel")]
    sweep.sort(key=lambda k: -((kernels[k]["read_bytes"] or 0) + (kernels[k]["write_bytes"] or 0)))
    if sweep and line:
        k = sweep[0]
        row = {key: step[key] for key in ("frames", "keypoints", "queries", "gpus", "profile_tag", "commit", "select_form", "correction", "reduction")}
        row.update(kernel=k, bytes_per_launch=int((kernels[k]["read_bytes"] or 0) + (kernels[k]["write_bytes"] or 0)),
                   sweep_ms_kernel_trace=kernels[k]["ms_per_step_kernel_trace"], kernels=kernels, step=step["step"])
        sq = os.path.join(src, "sq.json")
        if os.path.exists(sq):
            s = json.load(open(sq)).get(k) or {}
            if s.get("GRBM_GUI_ACTIVE") and s.get("SQ_INSTS_VALU"):
                cyc = s["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
                row.update(SQ_INSTS_VALU=s["SQ_INSTS_VALU"], SQ_INSTS_SALU=s.get("SQ_INSTS_SALU"), kernel_cycles=cyc,
                           valu_issue_frac=s["SQ_INSTS_VALU"] * 4.0 / (cyc * 1024.0))
        rows_path = os.path.join(here, "%s_traffic.json" % tag[:3])
        rows = []
        if os.path.exists(rows_path):
            try:
                rows = json.load(open(rows_path))
            except Exception:
                rows = []
        key = (row["frames"], row["keypoints"], row["queries"], row["gpus"])
        rows = [r for r in rows if (r.get("frames"), r.get("keypoints"), r.get("queries"), r.get("gpus", 1)) != key]
        rows.append(row)
        json.dump(rows, open(rows_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in step.items() if k != "kernels"}))
    for k, v in sorted(kernels.items(), key=lambda kv: -(kv[1]["ms_per_step_kernel_trace"] or 0)):
        if (v["ms_per_step_kernel_trace"] or 0) > 0.01:
            print("%-44s x%-2s %7.3f ms  read %6.2f GB  write %6.2f GB  L2 hit %s" % (k[:44], v["launches_per_step"], v["ms_per_step_kernel_trace"] or 0,
                  (v["read_bytes"] or 0) / 1e9, (v["write_bytes"] or 0) / 1e9, "%.2f" % v["l2_hit_rate"] if v["l2_hit_rate"] is not None else "-"))


if __name__ == "__main__":
    main()
