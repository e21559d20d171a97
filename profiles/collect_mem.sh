#!/bin/bash
# vL1D / TA / TLB counters of the sweep kernel (separate passes: the blocks have two counter slots;
# a set the hardware cannot collect makes rocprofv3 abort and then hang: every pass has its own timeout).
#   bash profiles/collect_mem.sh r02g
set -u
TAG=${1:-r02g}
ARGS=${2:-"--steps 3 --warmup 1 --cpu-baseline off --verify off --boundary off --sweep none"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for set in \
  "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum" \
  "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" ; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --pmc $set --kernel-include-regex "probe_sorted" --output-format csv -d $OUT/m$i -- python3 bench.py $ARGS > $OUT/m$i.json 2> $OUT/m$i.err
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, json, collections, os
src = sys.argv[1]
out = collections.defaultdict(list)
for path in glob.glob(os.path.join(src, "m*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "probe_sorted" in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: {"median": sorted(v)[len(v) // 2], "max": max(v), "n": len(v)} for c, v in out.items()}
json.dump(res, open(os.path.join(src, "mem_counters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
