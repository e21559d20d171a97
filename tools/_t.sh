for hq in default 8 16 default 8; do
  if [ $hq != default ]; then export GPU_MAX_HW_QUEUES=$hq; else unset GPU_MAX_HW_QUEUES; fi
  python3 bench.py --steps 20 --warmup 3 --cpu-baseline off --verify off --boundary off --skew off --cfg1 off --sweep none --predict-world 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$hq','value %.0f ms %.3f one-in-flight %.0f delivered %.0f'%(d['value'],d['ms_per_step'],d['one_batch_in_flight']['frames_per_s'],d['delivered']['frames_per_s']))
"
done
