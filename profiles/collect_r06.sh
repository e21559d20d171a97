#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object at the default (north-star) workload, per kernel and for the whole step.
# Run on the GPU box from the repo root:   bash profiles/collect_r06.sh r06a <commit> ["bench args"]
# (--in-flight 1: one batch in flight, so that a step is a contiguous run of dispatches the summary can cut out.)
# Every pass has its own timeout; kernel timing and PMC counters are separate runs (counters perturb timing);
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots); no trace domains together with --pmc.
set -u
TAG=${1:-r06a}
COMMIT=${2:-unknown}
ARGS=${3:-"--steps 6 --warmup 2 --cpu-baseline off --verify off --boundary off --sweep none --in-flight 1 --predict-world 0"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_under_fetch.json 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_under_write.json 2> $OUT/write.err
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/bench_under_tcc.json 2> $OUT/tcc.err
# SQ counters of every kernel of the step (medians per kernel): $OUT/sq.json
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" ; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $set --output-format csv -d $OUT/sq$i -- python3 bench.py $ARGS > $OUT/bench_under_sq$i.json 2> $OUT/sq$i.err
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, json, collections, os
src = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "sq*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in out.items()
       if not k.startswith("at::") and "elementwise" not in k and "rocclr" not in k}
json.dump(res, open(os.path.join(src, "sq.json"), "w"), indent=1, sort_keys=True)
PY
python3 profiles/summarize_step.py $OUT $TAG $COMMIT > $OUT/summarize.log 2>&1
# everything to be committed under profiles/ also goes to gpurun_out (only that travels back)
mkdir -p $OUT/for_profiles
cp profiles/${TAG}_* profiles/${TAG:0:3}_traffic.json $OUT/for_profiles/ 2>/dev/null
cp $OUT/sq.json $OUT/for_profiles/${TAG}_sq_medians.json
cat $OUT/summarize.log | cut -c1-300
