cd $GRAFT_REPO_ROOT
python bench.py --frames 100000 --queries 256 --steps 5 --warmup 2 --sweep none --cpu-baseline off --verify off --boundary off --predict-world 0 > gpurun_out/r05b_f100k.json 2> gpurun_out/r05b_f100k.err
SGTD_BENCH_TRACE=1 SGTD_BENCH_BACKEND=gloo SGTD_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 6 --warmup 2 --shard table --cpu-baseline off --also-table off > gpurun_out/r05b_2rank.json 2> gpurun_out/r05b_2rank.err
python - > gpurun_out/r05b_skew.log 2>&1 <<'PY'
import sys, time, json, traceback
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from types import SimpleNamespace
dev = torch.device('cuda', 0)
def to_dev_flat(fr):
    return (torch.from_numpy(np.ascontiguousarray(fr.xyz)).to(dev).contiguous(), torch.from_numpy(np.ascontiguousarray(fr.label).astype(np.int64)).to(dev).to(torch.int32).contiguous())
for F in (2500, 10000):
    args = SimpleNamespace(frames=F, queries=2048, steps=6)
    try:
        import os
        os.environ['SGTD_DEBUG'] = '1'
        print(json.dumps(bench.skew_leg(args, dev, torch.cuda.current_stream(), 0, to_dev_flat), indent=1))
    except Exception:
        traceback.print_exc()
PY
