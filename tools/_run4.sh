cd $GRAFT_REPO_ROOT
python - > gpurun_out/r05d_skew.log 2>&1 <<'PY'
import sys, time, json, traceback, os
sys.path.insert(0, '.')
os.environ['SGTD_DEBUG'] = '1'
import numpy as np, torch
from sgtd_amd import synth
from sgtd_amd.manager import STDescManager
for F, Q in ((2500, 48), (10000, 64)):
    smap, world = synth.make_skewed_map(F, stream=31)
    qs = synth.make_skewed_queries(world, Q, stream=3100)
    g = STDescManager()
    t0=time.time()
    g.add_frames(smap.xyz, smap.label, kp_off=smap.kp_off); g.finalize()
    print('F', F, 'entries', g.stats()['n_entries'], 'buckets', g.stats()['n_buckets'], 'L', g.stats()['bucket_len_sq_over_E'], 'build s', time.time()-t0, flush=True)
    try:
        res = g.query_frames(qs.xyz, qs.label, kp_off=qs.kp_off)
    except Exception:
        traceback.print_exc(); continue
    st = g.stats()
    print({k: st[k] for k in ('last_D','last_P','last_P_swept','last_M','last_cand_pairs','reruns_total','select_form')}, flush=True)
    top1 = res.top1()
    d = np.linalg.norm(smap.pose[np.clip(top1, 0, F - 1), :2] - qs.pose[:, :2], axis=1)
    print('top1 within 5m', np.mean(d < 5.0))
    anyc = []
    for q in range(Q):
        nc = int(res.n_cand[q]); f = res.cand_frame[q,:nc]
        dd = np.linalg.norm(smap.pose[f, :2] - qs.pose[q, :2], axis=1)
        anyc.append(bool((dd < 5.0).any()))
    print('any candidate within 5m', np.mean(anyc))
    g.verify(); bc, bf, bs = g.search_loop()
    dl = np.linalg.norm(smap.pose[np.clip(bf, 0, F - 1), :2] - qs.pose[:, :2], axis=1)
    print('loops', int((bf >= 0).sum()), 'loop within 5m', np.mean((bf >= 0) & (dl < 5.0)))
    print('max_batch', g.max_batch(225))
    t0=time.time(); g.query_frames(qs.xyz, qs.label, kp_off=qs.kp_off, fetch=False); g.sync(); print('batch ms', 1000*(time.time()-t0))
    g.close()
PY
