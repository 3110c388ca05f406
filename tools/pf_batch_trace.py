#!/usr/bin/env python3
"""Experiments build: the per-step timeline of a persistent launch (g_pb_tr): for every step the start / sweep-done / folded /
line-stored times of workgroup 0 and the spread over all workgroups."""
import ctypes as C, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from slam_jl_amd._lib import lib
NL, M, K = 512, 16, int(os.environ.get("PF_PROBE_K", "16"))
NP = int(os.environ.get("PF_PROBE_NP", "262144"))
force = {"0": False, "1": True, "-1": None}[os.environ.get("PF_PROBE_FORCE", "0")]
Q = np.array([[0.25, 0.0], [0.0, (3 * math.pi / 180) ** 2]]); R = np.array([[0.01, 0.0], [0.0, (math.pi / 180) ** 2]])
rng = np.random.default_rng(1)
lm = rng.uniform(-200, 200, (NL, 2))
Qs, Rs = pkg.small(Q), pkg.small(R)
pf = pkg.PFSlamState(NP, NL, seed=7, dtype="f32", distributed=False)
pf.shard.set_pose([0.0, 0.0, 0.3]); pf.shard.init_landmarks(lm, 0.01, 0.1)
obs = []
for t in range(64):
    ids = (np.arange(M) + M * t) % NL + 1
    obs.append((np.vstack([np.hypot(lm[ids - 1, 0], lm[ids - 1, 1]), np.arctan2(lm[ids - 1, 1], lm[ids - 1, 0]) - 0.3]), ids))
batches = [pkg.PFShard.prepare_batch([(0.0, 0.0)] * K, [obs[(k0 + j) % 64] for j in range(K)], force) for k0 in range(0, 64, K)]
NWARM, NTIMED = int(os.environ.get('PF_PROBE_WARM', '40')), int(os.environ.get('PF_PROBE_TIMED', '20'))
for rep in range(NWARM):
    pf.step_async_batch(batches[rep % len(batches)], 4.0, Qs, 0.025, Rs, persistent=True)
pf.flush(); pf.shard.sync()
t0 = time.perf_counter()
for rep in range(NTIMED):
    pf.step_async_batch(batches[rep % len(batches)], 4.0, Qs, 0.025, Rs, persistent=True)
pf.flush(); pf.shard.sync()
print(f"n {NP} K {K} force {force}: {1e6 * (time.perf_counter() - t0) / (NTIMED * K):.1f} us per step over {NTIMED} launches after {NWARM}")
tr = np.zeros(256 * 16 * 4, dtype=np.uint64)
fn = lib.slam_pf_debug_batch_trace
fn.restype = C.c_int
assert fn(pf.shard._h, tr.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
tr = tr.reshape(256, 16, 4)
nwg = int(np.count_nonzero(tr[:, 0, 0]))
t = (tr[:nwg, :K].astype(np.int64) - int(tr[:nwg, 0, 0].min())) / 100.0
print(f"{nwg} workgroups; us since the first workgroup's start; per step: WG0 start/sweep/folded/line | all WGs: start min-max, sweep-done min-max, line min-max")
for s in range(K):
    print(f" step {s:2d}: WG0 {t[0, s, 0]:7.1f} {t[0, s, 1]:7.1f} {t[0, s, 2]:7.1f} {t[0, s, 3]:7.1f} | start {t[:, s, 0].min():7.1f}-{t[:, s, 0].max():7.1f}  "
          f"sweep {t[:, s, 1].min():7.1f}-{t[:, s, 1].max():7.1f}  line {t[:, s, 3].min():7.1f}-{t[:, s, 3].max():7.1f}  slowest line WG {int(t[:, s, 3].argmax())}")
pf.close()
