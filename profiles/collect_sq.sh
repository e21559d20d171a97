#!/bin/bash
# Instruction-mix / stall counters per kernel (separate passes, SQ has 8 slots).
# Run on the GPU box from the repo root:   bash profiles/collect_sq.sh r01c
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 4 --warmup 1 --cpu-baseline off"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/inst -- python3 bench.py $ARGS > $OUT/b1.json 2> $OUT/inst.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/wait -- python3 bench.py $ARGS > $OUT/b2.json 2> $OUT/wait.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/b3.json 2> $OUT/tcc.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/lds -- python3 bench.py $ARGS > $OUT/b4.json 2> $OUT/lds.err
python3 - "$OUT" "$TAG" <<'PY'
import sys, glob, csv, json, collections, os
src, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()
       if not k.startswith("at::") and "elementwise" not in k}
json.dump(res, open("profiles/%s_sq_means.json" % tag, "w"), indent=1, sort_keys=True)
json.dump(res, open(os.path.join(src, "sq_means.json"), "w"), indent=1, sort_keys=True)
PY
tail -3 $OUT/*.err
