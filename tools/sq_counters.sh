#!/bin/bash
# SQ instruction / wait / LDS counters of the kernels whose name matches a regex, medians per kernel, three PMC passes
# of a short bench run.   bash tools/sq_counters.sh <tag> <kernel-regex> ["bench args"]   -> gpurun_out/<tag>/sq.json
set -u
TAG=${1:-sq}; KRE=${2:-pairs_query}
ARGS=${3:-"--steps 3 --warmup 1 --cpu-baseline off --verify off --boundary off --sweep none --profile-steps 1"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES_EQ_64" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-include-regex "$KRE" --output-format csv -d $OUT/s$i -- python3 bench.py $ARGS > $OUT/s$i.json 2> $OUT/s$i.err
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, json, collections, os
src = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "s*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, open(os.path.join(src, "sq.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
