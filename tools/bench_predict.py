#!/usr/bin/env python3
"""Time of slam_ekf_predict (E1, src/ekf.jl:8-43) and slam_ekf_augment (E3, :84-122) on the C3 state: the reference's
sim! calls predict nine times per observation step, so its cost belongs next to the step's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from __graft_entry__ import load_package                      # noqa: E402

pkg = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
x, P, zs = bench.make_workload(N, 64, 1, bench.SEED)
st = pkg.EKFSlamState(x[:3], np.zeros((3, 3), np.float32), dtype="f32", max_landmarks=N + 64)
st.set_state(x, P)
Q = np.array([[0.25, 0.0], [0.0, (3 * np.pi / 180) ** 2]])
for _ in range(200):
    st.predict(8.0, 0.01, 4.0, Q, 0.025)
st.sync()
t0 = time.perf_counter()
K = 2000
for _ in range(K):
    st.predict(8.0, 0.01, 4.0, Q, 0.025)
st.sync()
print(f"predict at N = {N}: {(time.perf_counter() - t0) / K * 1e6:.2f} us per call (enqueued back to back)")
st.timing(True, ["predict"])
st.timing_reset()
for _ in range(200):
    st.predict(8.0, 0.01, 4.0, Q, 0.025)
st.sync()
ms, cnt = st.timing_read()["predict"]
print(f"  device time of the bracketed launches: {ms / max(cnt, 1) * 1e3:.2f} us per call ({cnt} calls)")
# where the host time of a call goes
import ctypes as C
from importlib import import_module
lib = sys.modules[pkg.__name__ + "._lib"].lib if (pkg.__name__ + "._lib") in sys.modules else None
if lib is not None:
    nn = C.c_int()
    t0 = time.perf_counter()
    for _ in range(20000):
        lib.slam_ekf_num_landmarks(st._h, C.byref(nn))
    print(f"  a trivial ctypes call: {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us")
    q = np.ascontiguousarray(Q.T.reshape(-1))
    qp = q.ctypes.data_as(C.POINTER(C.c_double))
    st.sync()
    t0 = time.perf_counter()
    for _ in range(2000):
        lib.slam_ekf_predict(st._h, 8.0, 0.01, 4.0, qp, 0.025)
    t1 = time.perf_counter()
    st.sync()
    t2 = time.perf_counter()
    print(f"  raw slam_ekf_predict through ctypes: {(t1 - t0) / 2000 * 1e6:.2f} us per call to enqueue, {(t2 - t0) / 2000 * 1e6:.2f} us per call until done")
