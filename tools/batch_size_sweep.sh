for Q in ${QS:-48 96 192 512 2048}; do
python3 bench.py --queries $Q --steps 6 --warmup 2 --cpu-baseline off --verify off --boundary off --sweep none --profile-steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; Q=$Q
print('Q=%4d step %.3f ms  per query (us): probe %.2f votes %.2f topk %.2f count %.2f write %.2f | records/query %.0f' % (Q, d['ms_per_step'], 1e3*k['ms_probe']/Q, 1e3*k['ms_votes']/Q, 1e3*k['ms_topk']/Q, 1e3*k['ms_count']/Q, 1e3*k['ms_write']/Q, d['roofline']['M_matches']/Q))"
done
