#!/bin/bash
# SQ instruction / wait counters of one kernel (the sweep, or KREGEX=<name>) in two passes.   bash profiles/collect_sq.sh r02h
set -u
TAG=${1:-r02h}
ARGS=${2:-"--steps 3 --warmup 1 --cpu-baseline off --verify off --boundary off --sweep none"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" ; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --pmc $set --kernel-include-regex "${KREGEX:-probe_sorted}" --output-format csv -d $OUT/s$i -- python3 bench.py $ARGS > $OUT/s$i.json 2> $OUT/s$i.err
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, json, collections, os
src = sys.argv[1]
out = collections.defaultdict(list)
for path in glob.glob(os.path.join(src, "s*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if os.environ.get("KREGEX", "probe_sorted") in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: sorted(v)[len(v) // 2] for c, v in out.items()}
json.dump(res, open(os.path.join(src, "sq_sweep.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
