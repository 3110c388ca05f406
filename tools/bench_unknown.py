"""Experiment: cost of the unknown-correspondence update at the size of BASELINE config 4 (262144 particles, 512 slots)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
NP, NL, M = 262144, 512, 16
R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
rng = np.random.default_rng(1)
lm = rng.uniform(-200, 200, (NL, 2))
sh = pkg.PFShard(NP, NL, 7, dtype="f32")
sh.set_pose([0.0, 0.0, 0.3])
sh.init_landmarks(lm, 0.01, 0.1)                  # all 512 slots in use: the sweep is at its most expensive
pose = np.array([0.0, 0.0, 0.3])
def obs(t):
    ids = (np.arange(M) + M * t) % NL
    dx, dy = lm[ids, 0] - pose[0], lm[ids, 1] - pose[1]
    return np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, M))
for t in range(300):
    sh.update_unknown(obs(t), R, 4.0, 25.0)
sh.sync()
t0 = time.perf_counter()
K = 200
for t in range(K):
    sh.update_unknown(obs(t), R, 4.0, 25.0)
sh.sync()
dt = (time.perf_counter() - t0) / K
print("unknown-correspondence update, %d particles x %d slots x %d obs: %.3f ms per step = %.2f G pair evaluations/s, %.0f M particle-steps/s"
      % (NP, NL, M, dt * 1e3, NP * NL * M / dt / 1e9, NP / dt / 1e6))
