#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object at the default (north-star) workload.
# Run on the GPU box from the repo root:   bash profiles/collect_r03.sh r03a
# Every pass has its own timeout (a counter set the hardware refuses makes rocprofv3 abort and hang).
# Kernel timing and PMC counters are separate runs (counters perturb timing); FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC slots); no trace domains together with --pmc.
set -u
TAG=${1:-r03a}
ARGS=${2:-"--steps 6 --warmup 2 --cpu-baseline off --verify off --boundary off --sweep none"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
timeout -s KILL 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_under_fetch.json 2> $OUT/fetch.err
timeout -s KILL 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_under_write.json 2> $OUT/write.err
timeout -s KILL 420 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/bench_under_tcc.json 2> $OUT/tcc.err
timeout -s KILL 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/inst -- python3 bench.py $ARGS > $OUT/b1.json 2> $OUT/inst.err
timeout -s KILL 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/wait -- python3 bench.py $ARGS > $OUT/b2.json 2> $OUT/wait.err
timeout -s KILL 420 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/lds -- python3 bench.py $ARGS > $OUT/b4.json 2> $OUT/lds.err
python3 profiles/summarize.py $OUT $TAG > $OUT/summarize.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import sys, glob, csv, json, collections, os
src, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("inst", "wait", "lds"):
    for path in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in out.items()
       if not k.startswith("at::") and "elementwise" not in k and "rocclr" not in k}
json.dump(res, open(os.path.join(src, "sq_medians.json"), "w"), indent=1, sort_keys=True)
# the sweep's VALU issue utilisation into its row of ${TAG:0:3}_traffic.json: wave64 VALU instructions
# take 4 cycles of one of the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
rows_path = os.path.join("profiles", tag[:3] + "_traffic.json")
sw = [v for k, v in res.items() if k.startswith("probe_sorted_kernel") and v.get("SQ_INSTS_VALU")]
if sw and os.path.exists(rows_path):
    sw = max(sw, key=lambda v: v["SQ_INSTS_VALU"])
    rows = json.load(open(rows_path))
    for r in rows:
        if r.get("profile_tag") == tag and sw.get("GRBM_GUI_ACTIVE"):
            cyc = sw["GRBM_GUI_ACTIVE"] / 8.0
            r["SQ_INSTS_VALU"] = sw["SQ_INSTS_VALU"]
            r["SQ_INSTS_SALU"] = sw.get("SQ_INSTS_SALU")
            r["SQ_INSTS_VMEM_RD"] = sw.get("SQ_INSTS_VMEM_RD")
            r["SQ_INSTS_VMEM_WR"] = sw.get("SQ_INSTS_VMEM_WR")
            r["kernel_cycles"] = cyc
            r["valu_issue_frac"] = sw["SQ_INSTS_VALU"] * 4.0 / (cyc * 1024.0)
    json.dump(rows, open(rows_path, "w"), indent=1, sort_keys=True)
PY
# everything to be committed under profiles/ also goes to gpurun_out (only that travels back)
mkdir -p $OUT/for_profiles
cp profiles/${TAG}_* profiles/${TAG:0:3}_traffic.json $OUT/for_profiles/ 2>/dev/null
cp $OUT/sq_medians.json $OUT/for_profiles/${TAG}_sq_medians.json
tail -2 $OUT/summarize.log
