#!/usr/bin/env python3
"""Where the time of a resampling step goes in the MULTI-RANK flow of FastSLAM.resample, measured on one GPU: the
general path is forced with a single-process communicator (the collectives are identities, everything else -- the
whole-filter ancestor table, the torch index work, the host read-back of the split sizes, pack, apply -- runs)."""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package                      # noqa: E402

pkg = load_package()
import torch                                                   # noqa: E402

NP, NL, M = 262144, 512, 16
R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
rng = np.random.default_rng(1)
lm = rng.uniform(-200, 200, (NL, 2))


def run(force_general, steps=150, warm=60):
    pf = pkg.PFSlamState(NP, NL, seed=3, dtype="f32", distributed=False)
    pf.shard.set_pose([0.0, 0.0, 0.3])
    pf.shard.init_landmarks(lm, 0.01, 0.1)
    pf.force_exchange = force_general
    pose = np.array([0.0, 0.0, 0.3])
    obs = []
    for t in range(steps + warm):
        pose = np.array([pose[0] + 0.2 * math.cos(pose[2]), pose[1] + 0.2 * math.sin(pose[2]), pose[2]])
        ids = (np.arange(M) + M * t) % NL + 1
        dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
        obs.append((np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, M)), ids))
    import gc
    gc.collect()
    gc.disable()
    for z, ids in obs[:warm]:
        pf.step(8.0, 0.0, 4.0, Q, 0.025, z, ids, R, force_resample=True)
    pf.shard.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for z, ids in obs[warm:]:
        pf.step(8.0, 0.0, 4.0, Q, 0.025, z, ids, R, force_resample=True)
    pf.shard.sync()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    gc.enable()
    pf.close()
    return el * 1e3


for k in range(2):
    print("shortcut (world == 1): %.3f ms/step   general path forced: %.3f ms/step" % (run(False), run(True)))
