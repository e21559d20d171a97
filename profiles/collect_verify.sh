#!/bin/bash
# rocprofv3 evidence for candidate_verify (verify_solve_kernel + verify_kernel) on a full benchmark batch
# (north-star map, 2048 query frames, 102 400 candidates).  Run on the GPU box from the repo root:
#   bash profiles/collect_verify.sh r06a <commit> [variants/lib_vstat.so]
# Kernel timing and PMC counters are separate runs; FETCH_SIZE and WRITE_SIZE separate passes; the program itself after `--`;
# no trace domains together with --pmc.  The third argument names a -DSGTD_EXP_VSTAT build whose fall-out counts are kept.
set -u
TAG=${1:-r06a}
COMMIT=${2:-unknown}
VSTAT_LIB=${3:-}
ARGS=${VERIFY_ARGS:-"--steps 2 --warmup 1 --cpu-baseline off --verify on --boundary off --sweep none --in-flight 1 --predict-world 0 --skew off --cfg1 off --rccl-one off --profile-steps 1"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" ; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $set --kernel-include-regex "verify" --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/bench_under_p$i.json 2> $OUT/p$i.err
done
if [ -n "$VSTAT_LIB" ] && [ -f "$VSTAT_LIB" ]; then
  SGTD_ACCEL_LIB=$PWD/$VSTAT_LIB python3 bench.py $ARGS > $OUT/bench_vstat.json 2> $OUT/vstat.err
fi
python3 - "$OUT" "$TAG" "$COMMIT" <<'PY'
import sys, glob, csv, json, collections, os, re
src, tag, commit = sys.argv[1:4]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        ctr[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in ctr.items()}
stats = {}
for path in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "").strip()
        if "verify" in name or "search_loop" in name:
            stats[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
def line(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as exc:
        return {"error": str(exc)}
b = line(os.path.join(src, "bench.json"))
out = {"tag": tag, "commit": commit, "what": "candidate_verify of a full benchmark batch (STDesc.cpp:462-571): kernel trace and PMC passes (medians per launch) restricted to the verify kernels",
       "bench_verify": b.get("verify"), "kernel_stats": stats, "pmc_medians": med}
vk = med.get("verify_kernel", {})
if "FETCH_SIZE" in vk:
    # FETCH_SIZE / WRITE_SIZE are in KiB.  The guide's x 2 on gfx950 holds for wide (16 B per lane) streaming reads; this kernel's reads
    # are 4-byte gathers of 36-byte vertex records, for which the counter is uncalibrated: both readings are kept
    out["verify_kernel_hbm_bytes"] = {"fetch_as_counted": vk["FETCH_SIZE"] * 1024, "fetch_times_2": vk["FETCH_SIZE"] * 2048, "write": vk.get("WRITE_SIZE", 0) * 1024}
vs = os.path.join(src, "vstat.err")
if os.path.exists(vs):
    out["vstat"] = [l.strip() for l in open(vs) if l.startswith("[vstat]") or l.startswith("[vmstat]")]
json.dump(out, open(os.path.join(src, "%s_verify_profile.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:6000])
PY
mkdir -p $OUT/for_profiles
cp $OUT/${TAG}_verify_profile.json $OUT/for_profiles/
for f in $OUT/stats/*/*_kernel_stats.csv; do cp $f $OUT/for_profiles/${TAG}_verify_kernel_stats.csv; done
