#!/usr/bin/env python3
"""Gating alone (slam_ekf_associate) at several map sizes, in the sweep form and in the grid form (slam_ekf_set_gate_mode):
device time of the gating kernels from the library's event timers, per call, and what the grid queries visited.
SLAMHIP_X=32 / 64 switch the sweep's threshold pre-gate off / on at every size."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from __graft_entry__ import load_package
pkg = load_package()
SIZES = [(100, 8, "f32"), (300, 16, "f32"), (1000, 16, "f32"), (10000, 64, "f32"), (50000, 64, "f32")]
if os.environ.get("GATE_BENCH_BIG"):
    SIZES.append((100000, 64, "f32"))
for N, nz, dtype in SIZES:
    st, zs = B.make_workload_on_device(pkg, N, nz, 40, B.SEED, dtype, 0) if N > 14000 else (None, None)
    if st is None:
        x, P, zs = B.make_workload(N, nz, 40, B.SEED)
        st = pkg.EKFSlamState(x, P, dtype=dtype, max_landmarks=N)
    ref = None
    for mode in ("sweep", "grid"):
        st.set_gate_mode(mode)
        for z in zs[:10]:
            st.associate_vector(z, B.R, B.GATE1, B.GATE2)
        i0 = st.gate_info()
        st.timing(True); st.timing_reset()
        out = [st.associate_vector(z, B.R, B.GATE1, B.GATE2) for z in zs[10:]]
        t = st.timing_read()
        st.timing(False)
        i1 = st.gate_info()
        if ref is None:
            ref = out
        same = all(np.array_equal(a, b) for a, b in zip(out, ref))
        q = max(i1["queries"] - i0["queries"], 1)
        extra = (f"  cells {i1['cells_per_axis']}^2, visited {(i1['visited'] - i0['visited']) / q / nz:.1f} and evaluated "
                 f"{(i1['evaluated'] - i0['evaluated']) / q / nz:.1f} landmarks per observation, rebuilds {i1['rebuilds']}") if mode == "grid" else ""
        print(f"X={os.environ.get('SLAMHIP_X','0')} N={N} nz={nz} {dtype} {mode:5s}: gate {1e3*t['gate'][0]/t['gate'][1]:.2f} us  "
              f"gate_final {1e3*t['gate_final'][0]/max(t['gate_final'][1], 1):.2f} us  matched {int((out[-1]>0).sum())}  same decisions {same}{extra}", flush=True)
    st.close()
