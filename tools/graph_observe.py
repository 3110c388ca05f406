#!/usr/bin/env python3
"""Does a hipGraph (or just not waiting) shorten the observation step of a SMALL map?  
C2 (N = 1000, 16 observations, fp32), per step:
  (a) slam_ekf_observe            -- the product call: the host waits for the decisions of every step
  (b) slam_exp_observe_enqueue    -- the same six launches, enqueued back to back, no host wait (experiments build)
  (c) the launches of (b) captured ONCE into a hipGraph and replayed (the observations are baked into the kernel arguments)
Needs SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so.  Results are valid states only while no new feature arises (true here)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
lib = pkg._lib.lib
hip = None
for name in ("libamdhip64.so.7", "libamdhip64.so"):
    try:
        hip = C.CDLL(name)
        break
    except OSError:
        pass
N, NZ, STEPS = int(os.environ.get("GO_N", "1000")), int(os.environ.get("GO_NZ", "16")), 400
x, P, zs = B.make_workload(N, NZ, 64, B.SEED)
st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
Rv = np.ascontiguousarray(B.R.T.reshape(-1).astype(np.float64))
Rp = Rv.ctypes.data_as(C.POINTER(C.c_double))
zlist = [np.ascontiguousarray(z.T.reshape(-1)) for z in zs]          # (range, bearing) pairs
lib.slam_exp_observe_enqueue.restype = C.c_int
lib.slam_exp_observe_enqueue.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double]


def enqueue(k):
    z = zlist[k % len(zlist)]
    rc = lib.slam_exp_observe_enqueue(st._h, z.ctypes.data_as(C.POINTER(C.c_double)), NZ, Rp, B.GATE1, B.GATE2)
    assert rc == 0, pkg._lib.last_error()


for k in range(50):                                   # warm-up: workspaces allocated, clocks up
    st.observe(zs[k % len(zs)], B.R, B.GATE1, B.GATE2)
st.sync()
t0 = time.perf_counter()
for k in range(STEPS):
    st.observe(zs[k % len(zs)], B.R, B.GATE1, B.GATE2)
st.sync()
ta = (time.perf_counter() - t0) / STEPS
for k in range(50):
    enqueue(k)
st.sync()
t0 = time.perf_counter()
for k in range(STEPS):
    enqueue(k)
st.sync()
tb = (time.perf_counter() - t0) / STEPS
print(f"N={N} nz={NZ}: (a) observe, host waits for the decisions {ta * 1e6:.1f} us/step   (b) enqueue only {tb * 1e6:.1f} us/step")

# (c) capture one step into a graph and replay it
_, _, _, stream = st.device_ptrs()
stream = C.c_void_p(stream)
graph, gexec = C.c_void_p(), C.c_void_p()
rc = hip.hipStreamBeginCapture(stream, C.c_int(0))    # hipStreamCaptureModeGlobal
assert rc == 0, f"hipStreamBeginCapture {rc}"
enqueue(0)
rc = hip.hipStreamEndCapture(stream, C.byref(graph))
assert rc == 0 and graph.value, f"hipStreamEndCapture {rc}"
nnodes = C.c_size_t(0)
hip.hipGraphGetNodes(graph, None, C.byref(nnodes))
rc = hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0))
assert rc == 0, f"hipGraphInstantiate {rc}"
for _ in range(50):
    assert hip.hipGraphLaunch(gexec, stream) == 0
st.sync()
t0 = time.perf_counter()
for _ in range(STEPS):
    hip.hipGraphLaunch(gexec, stream)
st.sync()
tc = (time.perf_counter() - t0) / STEPS
print(f"            (c) hipGraph of one step ({nnodes.value} nodes), replayed {tc * 1e6:.1f} us/step")
# ten steps per graph: fewer graph launches for the same kernels
graph10, gexec10 = C.c_void_p(), C.c_void_p()
assert hip.hipStreamBeginCapture(stream, C.c_int(0)) == 0
for k in range(10):
    enqueue(k)
assert hip.hipStreamEndCapture(stream, C.byref(graph10)) == 0
assert hip.hipGraphInstantiate(C.byref(gexec10), graph10, None, None, C.c_size_t(0)) == 0
for _ in range(5):
    hip.hipGraphLaunch(gexec10, stream)
st.sync()
t0 = time.perf_counter()
for _ in range(STEPS // 10):
    hip.hipGraphLaunch(gexec10, stream)
st.sync()
td = (time.perf_counter() - t0) / (STEPS // 10 * 10)
print(f"            (d) hipGraph of TEN steps, replayed {td * 1e6:.1f} us/step")
hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph); hip.hipGraphExecDestroy(gexec10); hip.hipGraphDestroy(graph10)
st.close()
