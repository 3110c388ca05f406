#!/usr/bin/env python3
"""Per-step time of the persistent K-step launch (slam_pf_step_auto_batch) against the step-by-step auto mode, C4 workload
(512 landmarks, 16 observations per step), at the particle counts in PF_PROBE_NPS (default: the shard sizes of 1 / 2 / 4 / 8
GPUs), in the three regimes: no resampling (force 0), every step (force 1), the Neff rule.  Prints one line per size."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
NL, M = 512, int(os.environ.get("PF_PROBE_M", "16"))
K = int(os.environ.get("PF_PROBE_K", "16"))
STEPS = int(os.environ.get("PF_PROBE_STEPS", "1920"))
NPS = [int(v) for v in os.environ.get("PF_PROBE_NPS", "262144,131072,65536,32768").split(",")]
Q = np.array([[0.25, 0.0], [0.0, (3 * math.pi / 180) ** 2]]); R = np.array([[0.01, 0.0], [0.0, (math.pi / 180) ** 2]])
rng = np.random.default_rng(1)
lm = rng.uniform(-200, 200, (NL, 2))
Qs, Rs = pkg.small(Q), pkg.small(R)
for NP in NPS:
    pf = pkg.PFSlamState(NP, NL, seed=7, dtype="f32", distributed=False)
    obs, prep = [], []
    for t in range(64):
        ids = (np.arange(M) + M * t) % NL + 1
        z = np.vstack([np.hypot(lm[ids - 1, 0], lm[ids - 1, 1]), np.arctan2(lm[ids - 1, 1], lm[ids - 1, 0]) - 0.3])
        obs.append((z, ids))
        prep.append(pkg.PFShard.prepare_obs(z, ids))
    out = {}
    for regime, force in (("no_resample", False), ("every_step", True), ("neff", None)):
        batches = [pkg.PFShard.prepare_batch([(0.0, 0.0)] * K, [obs[(k0 + j) % 64] for j in range(K)], force) for k0 in range(0, 64, K)]
        for mode in ("batch", "single"):
            pf.shard.set_pose([0.0, 0.0, 0.3]); pf.shard.init_landmarks(lm, 0.01, 0.1)

            def run(nsteps):
                if mode == "batch":
                    for b in range(nsteps // K):
                        pf.step_async_batch(batches[b % len(batches)], 4.0, Qs, 0.025, Rs, persistent=True)
                else:
                    for k in range(nsteps):
                        pf.step_async(0.0, 0.0, 4.0, Qs, 0.025, None, None, Rs, force_resample=force, prepared=prep[k % 64])
            run(STEPS // 2)
            pf.flush()
            n0 = pf.resamples
            t0 = time.perf_counter()
            run(STEPS)
            pf.flush()
            pf.shard.sync()
            el = time.perf_counter() - t0
            out[(regime, mode)] = (1e6 * el / STEPS, pf.resamples - n0)
            if os.environ.get("PF_PROBE_STAMPS") == "1":
                print(f"   stamps {regime} {mode}: {[round(v, 1) for v in pf.shard.debug_stamps()]}")
    print(f"n {NP:7d} K {K:2d}: " + "  ".join(f"{r}: batch {out[(r, 'batch')][0]:6.1f} us ({out[(r, 'batch')][1]} res) single {out[(r, 'single')][0]:6.1f} us"
                                             for r in ("no_resample", "every_step", "neff")), flush=True)
    pf.close()
