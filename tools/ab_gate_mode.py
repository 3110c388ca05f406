#!/usr/bin/env python3
"""One-box A/B of the observation step (slam_ekf_observe: gating + update) with the gating in the sweep form and in the
grid form: C3 (10k landmarks, 64 observations) and C2 (1k, 16).  Alternating blocks of steps on the SAME filter, wall clock
with the queue kept full (the step is enqueued; the host waits for the decisions only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from __graft_entry__ import load_package
pkg = load_package()
for N, nz in ((10000, 64), (1000, 16)):
    x, P, zs = B.make_workload(N, nz, 64, B.SEED)
    st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
    st.set_async(True) if hasattr(st, "set_async") else None
    res = {"sweep": [], "grid": []}
    for rep in range(6):
        for mode in ("sweep", "grid"):
            st.set_gate_mode(mode)
            for k in range(30):
                st.observe(zs[k % 64], B.R, B.GATE1, B.GATE2)
            st.sync()
            t0 = time.perf_counter()
            for k in range(200):
                st.observe(zs[k % 64], B.R, B.GATE1, B.GATE2)
            st.sync()
            res[mode].append((time.perf_counter() - t0) / 200 * 1e6)
    for mode in ("sweep", "grid"):
        v = res[mode]
        print(f"N={N} nz={nz} {mode:5s}: {np.median(v):.1f} us/step (median of {len(v)} blocks of 200; min {min(v):.1f}, max {max(v):.1f})", flush=True)
    print("   ", st.gate_info(), flush=True)
    st.close()
