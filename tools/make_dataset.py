"""Writes a synthetic map/query dataset in the reference's graph-JSON format
(one file per frame, pose row included) for examples/localize:
python tools/make_dataset.py <out_dir> [n_map_frames=1000] [n_queries=1000] [keypoints=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgtd_amd import evaluate as ev, ingest, synth  # noqa: E402


def main():
    out = sys.argv[1]
    f = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    q = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    smap = synth.make_map(f, n, stream=1)
    qs = synth.make_queries(smap, q, stream=1)
    for sub in ("map", "query"):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
    for i in range(f):
        ingest.write_graph_json(os.path.join(out, "map", "%06d.json" % i), smap.xyz[i], smap.label[i], ev.pose_row(*smap.pose[i]))
    for i in range(q):
        ingest.write_graph_json(os.path.join(out, "query", "%06d.json" % i), qs.xyz[i], qs.label[i], ev.pose_row(*qs.pose[i]))
    print("wrote %d map and %d query graphs under %s" % (f, q, out))


if __name__ == "__main__":
    main()
