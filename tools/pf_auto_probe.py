#!/usr/bin/env python3
"""Where does an auto step spend its time?  C4-sized filter: per-step wall time with the queue kept full, and the
in-kernel stamps of the last step (kernel start -> statistics collected -> folded -> decided -> bookkeeping -> published;
[6] the collecting workgroup done with its own share, [7] its number of polls)."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
NP, NL, M = int(os.environ.get("PF_PROBE_NP", "262144")), 512, 16      # PF_PROBE_NP: particles (a shard of an 8-rank filter has 32768)
Q = np.array([[0.25, 0.0], [0.0, (3 * math.pi / 180) ** 2]]); R = np.array([[0.01, 0.0], [0.0, (math.pi / 180) ** 2]])
rng = np.random.default_rng(1)
lm = rng.uniform(-200, 200, (NL, 2))
pf = pkg.PFSlamState(NP, NL, seed=7, dtype="f32", distributed=False)
pf.shard.set_pose([0.0, 0.0, 0.3]); pf.shard.init_landmarks(lm, 0.01, 0.1)
obs = []
for t in range(64):
    ids = (np.arange(M) + M * t) % NL + 1
    z = np.vstack([np.hypot(lm[ids - 1, 0], lm[ids - 1, 1]), np.arctan2(lm[ids - 1, 1], lm[ids - 1, 0]) - 0.3])
    obs.append(pkg.PFShard.prepare_obs(z, ids))
Qs, Rs = pkg.small(Q), pkg.small(R)
PROPOSAL = bool(int(os.environ.get("PF_PROBE_PROPOSAL", "0")))       # the FastSLAM-2.0 step instead of the 1.0 step
for force in (False, True):
    for k in range(2000):
        pf.step_async(8.0, 0.0, 4.0, Qs, 0.025, None, None, Rs, force_resample=force, proposal=PROPOSAL, prepared=obs[k % 64])
    pf.flush()
    t0 = time.perf_counter()
    for k in range(2000):
        pf.step_async(8.0, 0.0, 4.0, Qs, 0.025, None, None, Rs, force_resample=force, proposal=PROPOSAL, prepared=obs[k % 64])
    t1 = time.perf_counter()
    pf.flush()
    t2 = time.perf_counter()
    print(f"force={force}: enqueue {1e6 * (t1 - t0) / 2000:.1f} us/step, total {1e6 * (t2 - t0) / 2000:.1f} us/step, stamps (us) {pf.shard.debug_stamps()}")
