#!/bin/bash
# times the sweep of kernel variants under test (variants/lib_*.so, built with -DSGTD_EXP_*) and of launch
# knobs with a short bench run each; prints ms per kernel stage.   bash tools/exp_variants.sh [knob=value ...]
ARGS="--steps 4 --warmup 1 --cpu-baseline off --verify off --boundary off --sweep none --profile-steps 2"
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py $ARGS 2>gpurun_out/exp_err_$label.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('%-28s step %.2f ms  build %.2f sort %.2f probe %.2f votes %.2f count %.2f write %.2f  swept %.3e' % ('$label', d['ms_per_step'], k['ms_build'], k['ms_sort'], k['ms_probe'], k['ms_votes'], k['ms_count'], k['ms_write'], d['roofline'].get('P_swept_after_slice_pruning') or 0))"
}
run base X=1
for f in variants/lib_*.so; do [ -f "$f" ] && run $(basename $f .so) SGTD_ACCEL_LIB=$PWD/$f; done
for kv in "$@"; do run "$kv" "$kv"; done
