import sys; sys.path.insert(0,'.')
import numpy as np
from sgtd_amd import manager, synth
m=synth.make_map(400,200,stream=1); qs=synth.make_queries(m,24,stream=1)
g=manager.STDescManager(); g.add_frames(m.xyz,m.label)
for i in range(5):
    r=g.query_frames(qs.xyz,qs.label); st=g.stats()
    print("batch",i,"overflowed",st["overflowed"],"M",st["last_M"],"P_swept",st["last_P_swept"],flush=True)
