#!/usr/bin/env python3
"""The front half of the update at the headline workload (C3: 10000 landmarks, 64 observations): where the time of
factor_w1_kernel goes.  100 MHz wall-clock stamps of the workgroup that factors S and of the first panel workgroup
(slam_ekf_debug_stamps), printed as microseconds from the factorising workgroup's start; median over the steps.
FRONT_N / FRONT_NZ change the workload."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from __graft_entry__ import load_package
pkg = load_package()
N, nz = int(os.environ.get("FRONT_N", 10000)), int(os.environ.get("FRONT_NZ", 64))
x, P, zs = B.make_workload(N, nz, 40, B.SEED)
st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
st.debug_stamps(True)
rows = []
for z in zs:
    B.gpu_step(st, z)
    s = np.array(st.debug_stamps(True), dtype=np.int64)
    rows.append((s - s[0]) / 100.0)
rows = np.array(rows[8:])
med = np.median(rows, axis=0)
if int(os.environ.get("SLAMHIP_FW1", "0")) & 16:
    names = ["F start", "F jacobians", "F S in LDS", "F (S out)", "F eliminated", "F g", "F end", "-",
             "E step 3 starts", "E wave 0: trailing done", "E wave 0: diagonal block done", "E after barrier A", "E wave 0 at barrier B", "E after barrier B", "E loop starts", "E loop ends"]
    print("(experiments build, SLAMHIP_FW1 bit 16: step 3 of the elimination, wave 0)")
else:
  names = ["F start", "F jacobians", "F S in LDS", "F (S out)", "F eliminated", "F g", "F end", "-",
           "P start", "P operands", "P block col 0 in", "P last block col in", "P last W1 out", "P g in", "P end", "-"]
print(f"N={N} nz={nz} SLAMHIP_X={os.environ.get('SLAMHIP_X', '0')}: median over {len(rows)} steps, us from the factorising workgroup's start")
for n, v in zip(names, med):
    if n != "-":
        print(f"  {n:22s} {v:8.2f}")
st.close()
