import time, numpy as np, sys
sys.path.insert(0, '.')
from sgtd_amd import manager, synth
t=time.time()
def lap(s):
    global t; n=time.time(); print("%-40s %.1f s" % (s, n-t), flush=True); t=n
F,N,Q=100000,200,48
m=synth.make_map(F,N,stream=4); lap("make_map")
qs=synth.make_queries(m,Q,stream=4); lap("make_queries")
single=manager.STDescManager(max_frame_n=F+1); single.add_frames(m.xyz,m.label); lap("single add_frames")
sres=single.query_frames(qs.xyz,qs.label); lap("single query"); print(single.stats()["overflowed"], single.stats()["n_entries"])
sres=single.query_frames(qs.xyz,qs.label); lap("single query again")
pairs=[single.result_pairs(q,sres) for q in range(0,Q,12)]; lap("result_pairs")
single.close(); del single; lap("close")
multi=manager.STDescManager(max_frame_n=F+1, devices=[0]*8); multi.add_frames(m.xyz,m.label); lap("multi add_frames")
mres=multi.query_frames(qs.xyz,qs.label); lap("multi query")
mres=multi.query_frames(qs.xyz,qs.label); lap("multi query again")
for k,q in enumerate(range(0,Q,12)):
    mq,me=multi.result_pairs(q,mres)
lap("multi result_pairs")
multi.close(); lap("multi close")
