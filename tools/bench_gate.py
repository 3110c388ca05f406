#!/usr/bin/env python3
"""Gating sweep alone (slam_ekf_associate) at several map sizes, with the N2 pre-gate and without (SLAMHIP_X=32):
device time of the sweep kernel from the library's event timers, per call."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from __graft_entry__ import load_package
pkg = load_package()
for N, nz, dtype in ((1000, 16, "f32"), (10000, 64, "f32"), (50000, 64, "f32")):
    st, zs = B.make_workload_on_device(pkg, N, nz, 40, B.SEED, dtype, 0) if N > 14000 else (None, None)
    if st is None:
        x, P, zs = B.make_workload(N, nz, 40, B.SEED)
        st = pkg.EKFSlamState(x, P, dtype=dtype, max_landmarks=N)
    for z in zs[:10]:
        st.associate_vector(z, B.R, B.GATE1, B.GATE2)
    st.timing(True); st.timing_reset()
    for z in zs[10:]:
        a = st.associate_vector(z, B.R, B.GATE1, B.GATE2)
    t = st.timing_read()
    print(f"X={os.environ.get('SLAMHIP_X','0')} N={N} nz={nz} {dtype}: gate {1e3*t['gate'][0]/t['gate'][1]:.2f} us  gate_final {1e3*t['gate_final'][0]/t['gate_final'][1]:.2f} us  matched {int((a>0).sum())}")
    st.close()
