cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sed -i "s/-fopenmp -DSGTD_SHIM_TIMING/-DSGTD_SHIM_TIMING/" tools/shim_timing.sh
bash tools/shim_timing.sh 10000 16 > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/r03n_hip -- /tmp/localize_t /tmp/map.cache /tmp/query.cache 16 > gpurun_out/r03n_hip.log 2>&1
LOCALIZE_PER_FRAME=16 timeout -s KILL 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/r03n_hip_pf -- /tmp/localize_t /tmp/map.cache /tmp/query.cache 16 > gpurun_out/r03n_hip_pf.log 2>&1
ls gpurun_out/r03n_hip_pf/*/
python3 - <<'PY'
import csv,glob
for tag in ('r03n_hip','r03n_hip_pf'):
    f=glob.glob('gpurun_out/%s/*/*_hip_api_stats.csv'%tag)
    if not f: print('no hip stats', tag); continue
    print(tag)
    for r in list(csv.DictReader(open(f[0])))[:14]: print('  ', r['Name'][:40].ljust(40), r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
