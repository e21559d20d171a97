#!/bin/bash
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of a bench command other than the default step (the
# F = 100 000 step and the skewed workload had no counter set).  Run on the GPU box from the repo root:
#   bash profiles/collect_more.sh <tag> <commit> "<bench args>"
# Separate passes for timing and each counter; the program itself after `--`.
set -u
TAG=$1; COMMIT=$2; ARGS=$3
PROG=${4:-bench.py}          # the program under the profiler (default bench.py; tools/skew_step.py for the skewed step alone)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $PROG $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $PROG $ARGS > $OUT/bench_under_fetch.json 2> $OUT/fetch.err
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $PROG $ARGS > $OUT/bench_under_write.json 2> $OUT/write.err
python3 - "$OUT" "$TAG" "$COMMIT" "$ARGS" <<'PY'
import sys, glob, csv, json, collections, os
src, tag, commit, args = sys.argv[1:5]
def name_of(s): return s.split("(")[0].replace("void ", "").strip()
stats = {}
for path in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(path)):
        stats[name_of(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for path in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            ctr[name_of(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = {}
for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
    c = ctr.get(k, {})
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    f, w = med(c.get("FETCH_SIZE", [])), med(c.get("WRITE_SIZE", []))
    rows[k] = dict(st, read_bytes_per_launch=2048.0 * f if f is not None else None, write_bytes_per_launch=1024.0 * w if w is not None else None)
line = None
try:
    line = json.loads(open(os.path.join(src, "bench_under_stats.json")).read().strip().splitlines()[-1])
except Exception as exc:
    line = {"error": str(exc)}
out = {"tag": tag, "commit": commit, "args": args, "kernels": rows,
       "correction": "read bytes = 2 x FETCH_SIZE KiB (gfx950 wide-load undercount), write bytes = WRITE_SIZE KiB, medians per launch; both count Infinity-Cache hits",
       "line_under_the_kernel_trace": {k: v for k, v in line.items() if k in ("value", "ms_per_step", "config", "workload", "queries_per_step", "frames_per_s_each_step_waited_for", "reruns", "kernel_ms_last_step", "M_matches_per_query", "P_swept_per_query", "candidate_pairs")} if isinstance(line, dict) else None}
if isinstance(line, dict) and isinstance(line.get("roofline"), dict):
    out["kernel_ms_live"] = line["roofline"].get("kernel_ms")
json.dump(out, open(os.path.join(src, "%s_profile.json" % tag), "w"), indent=1, sort_keys=True)
for k, v in rows.items():
    print("%-44s x%-4d avg %9.1f us  read %s  write %s" % (k[:44], v["calls"], v["avg_us"], "%.2f GB" % (v["read_bytes_per_launch"] / 1e9) if v["read_bytes_per_launch"] else "-", "%.2f GB" % (v["write_bytes_per_launch"] / 1e9) if v["write_bytes_per_launch"] else "-"))
PY
mkdir -p $OUT/for_profiles
cp $OUT/${TAG}_profile.json $OUT/for_profiles/
for f in $OUT/stats/*/*_kernel_stats.csv; do cp $f $OUT/for_profiles/${TAG}_kernel_stats.csv; done
