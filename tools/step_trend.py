"""Experiment: per-step wall time and matched count over a long run (is the early slowness clocks or workload drift?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from __graft_entry__ import load_package
pkg = load_package()
x, P, zs = bench.make_workload(10000, 64, 600, bench.SEED)
st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=10000)
st.set_async(True)
rows = []
for i, z in enumerate(zs):
    t0 = time.perf_counter()
    a = st.observe(z, bench.R, 4.0, 25.0)
    st.sync()
    rows.append(((time.perf_counter() - t0) * 1e3, int((a > 0).sum())))
for i in range(0, 600, 50):
    blk = rows[i:i + 50]
    print("steps %3d-%3d  ms/step %.4f  matched/step %.1f" % (i, i + 49, np.mean([b[0] for b in blk]), np.mean([b[1] for b in blk])))
worst = sorted(range(len(rows)), key=lambda i: -rows[i][0])[:6]
print("slowest steps:", [(i, round(rows[i][0], 2)) for i in sorted(worst)])
