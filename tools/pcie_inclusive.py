#!/usr/bin/env python3
"""What the headline would be if the caller handed the state over per call (it does not: the state is device resident):
slam_ekf_set_state + one observation step + slam_ekf_get_state at C3, wall clock."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from __graft_entry__ import load_package
pkg = load_package()
N, nz = 10000, 64
x, P, zs = B.make_workload(N, nz, 8, B.SEED)
st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
for rep in range(3):
    t0 = time.perf_counter(); st.set_state(x, P); st.sync(); t1 = time.perf_counter()
    a = st.observe(zs[rep], B.R, B.GATE1, B.GATE2); st.sync(); t2 = time.perf_counter()
    xd, Pd = st.download(); t3 = time.perf_counter()
    m = int((a > 0).sum())
    print(f"rep {rep}: upload {1e3 * (t1 - t0):.1f} ms, step {1e3 * (t2 - t1):.2f} ms, download {1e3 * (t3 - t2):.1f} ms "
          f"=> {m / (t3 - t0):.0f} obs-updates/s with the state crossing PCIe both ways (1.6 GB each way)", flush=True)
st.close()
