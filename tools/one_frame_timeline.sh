#!/bin/bash
# The kernels of ONE BuildSingleScanSTD + SearchLoop call through adapter/STDesc_shim.hpp (sgtd_build + sgtd_search_frame) on the
# device's timeline: rocprofv3 --kernel-trace of examples/localize LOCALIZE_PER_FRAME=32 on the 10 000-frame map; the last
# call's kernels are cut out of the trace.      bash tools/one_frame_timeline.sh <tag> <commit>   -> gpurun_out/<tag>/
set -u
TAG=${1:-r06_frame}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG; mkdir -p $OUT
python3 - "$OUT" <<'PY'
import os, subprocess, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from sgtd_amd import evaluate as ev, ingest, synth
out = sys.argv[1]
smap = synth.make_map(10000, 200, stream=1)
qs = synth.make_queries(smap, 32, stream=1)
ingest.write_cache(os.path.join(out, "map.cache"), smap.xyz, smap.label, np.stack([ev.pose_row(*p) for p in smap.pose]))
ingest.write_cache(os.path.join(out, "query.cache"), qs.xyz, qs.label, np.stack([ev.pose_row(*p) for p in qs.pose]))
lib = os.path.join(ROOT, "sgtd_amd")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "localize.cpp"), "-o", os.path.join(ROOT, "examples", "localize"),
                       "-L" + lib, "-lsgtd_accel", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"])
PY
LOCALIZE_PER_FRAME=32 ./examples/localize $OUT/map.cache $OUT/query.cache 32 > $OUT/localize.out 2> $OUT/localize.err
grep -h "per-frame calls\|SearchLoop by part" $OUT/localize.out | cut -c1-400
LOCALIZE_PER_FRAME=32 timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- ./examples/localize $OUT/map.cache $OUT/query.cache 32 > $OUT/localize_traced.out 2> $OUT/trace.err
python3 - "$OUT" "$TAG" "$COMMIT" <<'PY'
import csv, glob, json, os, sys
out, tag, commit = sys.argv[1:4]
rows = []
for path in glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
builds = [i for i, r in enumerate(rows) if "build_frames_kernel" in r[2]]
# a per-frame call through the adapter ends in pack_frame_kernel (sgtd_search_frame) and the gather of the entries: the last
# such call but one (the example's later phases time candidate_selector alone and take the separate calls)
packs = [i for i, r in enumerate(rows) if "pack_frame_kernel" in r[2]]
at = packs[-2]
a = max(i for i in builds if i < at)
b = min([i for i in builds if i > at] + [len(rows)])
call = rows[a:b]
t0 = call[0][0]
kern = [{"us_from_build": (s - t0) / 1e3, "us": (e - s) / 1e3, "grid": g, "kernel": k[:70]} for s, e, k, g in call]
search = [k for k in kern if "build_frames_kernel" not in k["kernel"]]
# the search chain: from the copy of the query descriptors' count to pack_frame_kernel, the call's last kernel (what follows belongs to the next
# call: the host is filling loop_std_pair in between)
chain = []
for k in search:
    chain.append(k)
    if "pack_frame_kernel" in k["kernel"]:
        break
entries = [k for k in chain if "gather_pair_entries_kernel" in k["kernel"]]
res = {"what": "rocprofv3 --kernel-trace of examples/localize LOCALIZE_PER_FRAME=32 on the 10 000-frame map: the kernels of ONE BuildSingleScanSTD + SearchLoop call through adapter/STDesc_shim.hpp (sgtd_build + sgtd_search_frame)",
       "tag": tag, "commit": commit, "kernels_in_the_call": len(kern), "search_kernels": len(chain),
       "search_chain_us_first_to_last_kernel": (chain[-1]["us_from_build"] + chain[-1]["us"] - chain[0]["us_from_build"]) if chain else None,
       "search_kernel_us_sum": sum(k["us"] for k in chain),
       "of_it_gather_pair_entries_us": entries[0]["us"] if entries else None,
       "gather_note": "gather_pair_entries_kernel writes the inlier pairs' table entries (136 B + 4 B per pair, some 160 000 pairs of this map's frames) over the link into the caller's page-locked arrays: it runs at the link's rate and replaces eight device -> host copies and the call's second wait",
       "one_call": kern}
try:
    res["localize_output"] = [l.strip() for l in open(os.path.join(out, "localize.out")) if "per-frame calls" in l or "SearchLoop by part" in l]
except Exception:
    pass
json.dump(res, open(os.path.join(out, "%s_one_frame_kernel_timeline.json" % tag), "w"), indent=1)
print("kernels in the call", len(kern), "search kernels", len(chain), "chain us", res["search_chain_us_first_to_last_kernel"], "sum of kernel us", res["search_kernel_us_sum"], "of it the entries over the link", res["of_it_gather_pair_entries_us"])
for k in chain:
    print("%8.1f %7.1f %8d %s" % (k["us_from_build"], k["us"], k["grid"], k["kernel"][:60]))
PY
