"""Times the native graph-JSON loader (sgtd_amd/ingest.py) on synthetic files:
python tools/ingest_bench.py [n_files] — prints ms per thread count and the Python json time."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgtd_amd import ingest, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    m = synth.make_map(min(n, 500), 200, stream=1)
    d = tempfile.mkdtemp()
    paths = []
    for f in range(n):
        p = os.path.join(d, "%06d.json" % f)
        ingest.write_graph_json(p, m.xyz[f % len(m.xyz)], m.label[f % len(m.xyz)], np.zeros(12))
        paths.append(p)
    size = sum(os.path.getsize(p) for p in paths) / 1e6
    ingest.load_graphs(paths, threads=4)
    for th in (1, 2, 4, 8, 16, 32, 64):
        if th > 2 * (os.cpu_count() or 1):
            break
        t = time.perf_counter()
        ingest.load_graphs(paths, threads=th)
        dt = time.perf_counter() - t
        print("native, %2d threads: %7.1f ms for %d files (%.1f MB) = %.0f MB/s" % (th, dt * 1e3, n, size, size / dt))
    t = time.perf_counter()
    for p in paths[:200]:
        j = json.load(open(p))
        np.array(j["centers"], np.float32), np.array(j["nodes"])
    print("python json + numpy: %7.1f ms for %d files" % ((time.perf_counter() - t) * n / 200 * 1e3, n))
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
