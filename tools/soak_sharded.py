#!/usr/bin/env python3
"""Soak of the sharded device-side resampling: random configurations (ranks, particles per rank, landmarks, observations per
step, dtype, FastSLAM-1.0 / 2.0, resampling schedule), the shards driven by one host thread each on ONE card, against the
one-rank synchronous filter: particles bit-identical, log-weights within 8 ulp, zero halts; and, where the ranks' slices are
multiples of 1024 particles, against the one-rank AUTO filter: log-weights EQUAL.  usage: soak_sharded.py [configs]"""
import math
import os
import sys
import threading

# several shards of ONE process spin on each other's records from their own streams: every stream needs a hardware queue of
# its own (two streams sharing one queue = the kernel that waits sits in front of the kernel it waits for)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])


class Rank:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world


def one(cfg_seed):
    rng = np.random.default_rng(cfg_seed)
    world = int(rng.choice([2, 3, 4]))
    # below and above the step kernels' thresholds; multiples of 1024 (the statistics tree is then the one-rank filter's:
    # log-weights must be EQUAL to the one-rank AUTO filter's) and ragged sizes (a few ulp)
    per = int(rng.choice([257, 1000, 4096, 20 * 1024, 20000, 49 * 1024, 70000, 69 * 1024]))
    nl = int(rng.choice([6, 20, 70]))
    m = int(rng.integers(1, min(nl, 20) + 1))
    dtype = str(rng.choice(["f32", "f64"]))
    proposal = bool(rng.integers(0, 2))
    repeats = bool(rng.integers(0, 2))
    steps = int(rng.integers(8, 30))
    n = per * world
    lm = rng.uniform(-40, 40, (nl, 2))
    known = int(rng.integers(1, nl + 1))
    ref_sh = pkg.PFShard(n, nl, cfg_seed, dtype=dtype)
    shards = [pkg.PFShard(per, nl, cfg_seed, dtype=dtype, first=r * per, n_global=n) for r in range(world)]
    for sh in [ref_sh] + shards:
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm[:known], 0.01, 0.1)
    pkg.attach_local_peers(shards)
    ref = pkg.FastSLAM(ref_sh, None, neff_frac=0.75)
    ranks = [pkg.FastSLAM(sh, Rank(r, world), neff_frac=0.75) for r, sh in enumerate(shards)]
    pose = np.array([0.5, 1.5, -0.2])
    plan = []
    for t in range(steps):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = rng.choice(np.arange(1, nl + 1), size=m, replace=repeats)
        dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, m))
        force = [None, None, True, True, False][int(rng.integers(0, 5))]
        plan.append((0.01 * (t % 5), z, ids, force))
    trace = os.environ.get("SOAK_TRACE") == "1"
    ref_hist = []
    for g, z, ids, force in plan:
        info = ref.step(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal)
        if trace:
            p_, w_, _ = ref_sh.download(landmarks=False)
            ref_hist.append((p_.copy(), w_.copy(), info))
    want = ref_sh.download()
    ref_resamples = ref.resamples
    want_auto = None
    if per % 1024 == 0:                     # the one-rank AUTO filter: the sharded one must equal it bit for bit, weights included
        a_sh = pkg.PFShard(n, nl, cfg_seed, dtype=dtype)
        a_sh.set_pose([0.5, 1.5, -0.2])
        a_sh.init_landmarks(lm[:known], 0.01, 0.1)
        fa = pkg.FastSLAM(a_sh, None, neff_frac=0.75)
        for g, z, ids, force in plan:
            fa.step_async(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal)
        fa.flush()
        want_auto = a_sh.download()
        a_sh.close()
    if os.environ.get("SOAK_KEEP_REF") != "1":
        ref_sh.close()                      # (its stream goes before the shards' kernels start waiting for each other)
    got, errs = [None] * world, []
    step_pw = [[None] * world for _ in plan]
    bar = threading.Barrier(world)

    def drive(r):
        try:
            f = ranks[r]
            for t, (g, z, ids, force) in enumerate(plan):
                f.step_async(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal)
                if trace:                 # poses and weights after every step (no landmark download: the lazy state is left alone)
                    info = f.flush()
                    p_, w_, _ = f.shard.download(landmarks=False)
                    step_pw[t][r] = (p_.copy(), w_.copy(), info, f.resamples)
                    bar.wait()
            f.flush()
            assert f.resamples == ref_resamples, (f.resamples, ref_resamples)
            got[r] = f.shard.download()
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    ok = not errs and all(g is not None for g in got)
    if trace and ok:
        for t in range(len(plan)):
            pa = np.hstack([step_pw[t][r][0] for r in range(world)])
            wa = np.concatenate([step_pw[t][r][1] for r in range(world)])
            pr, wr, info = ref_hist[t]
            dw = float(np.abs(wa.astype(np.float64) - wr.astype(np.float64)).max())
            print(f"   step {t}: force {plan[t][3]} ids {plan[t][2].tolist()} ref (neff, did) {info} shard {step_pw[t][0][2]} resamples {step_pw[t][0][3]} "
                  f"poses equal {np.array_equal(pa, pr)} ({int((pa != pr).any(axis=0).sum())} particles differ) max |dlogw| {dw:.3e}", flush=True)
    if ok:
        halts = [sh.comm_info()["halts"] for sh in shards]
        pa = np.hstack([g[0] for g in got])
        la = np.concatenate([g[2] for g in got], axis=2)
        wa = np.concatenate([g[1] for g in got])
        # log-weights: the ranks' partial sums are folded in rank order, the one-rank filter folds its workgroups' -- every
        # normalisation's shift may differ by an ulp, and the steps between two resamplings add up (seen: 4.2 ulp after 18
        # FastSLAM-2.0 steps with 13 resamplings, poses and maps bit-identical)
        tol = 8 * np.finfo(wa.dtype).eps * max(1.0, float(np.abs(want[1]).max()))
        checks = dict(halts=halts == [0] * world, poses=bool(np.array_equal(pa, want[0])), landmarks=bool(np.array_equal(la, want[2])),
                      logw=bool(np.allclose(wa, want[1], rtol=0, atol=tol)))
        if want_auto is not None:           # aligned slices: EQUAL to the one-rank auto filter (round 4: the canonical statistics tree)
            checks["logw_equal_to_one_rank_auto"] = bool(np.array_equal(wa, want_auto[1]))
            checks["poses_equal_to_one_rank_auto"] = bool(np.array_equal(pa, want_auto[0]))
        ok = all(checks.values())
        if not ok:
            print(f"   checks {checks} halts {halts}; particles with different poses {int((pa != want[0]).any(axis=0).sum())}, "
                  f"landmark records differing {int((la != want[2]).any(axis=1).sum())} of {la.shape[0] * la.shape[2]}, "
                  f"max |dlogw| {float(np.abs(wa.astype(np.float64) - want[1].astype(np.float64)).max()):.3e} (tol {tol:.3e})", flush=True)
    th = [threading.Thread(target=sh.detach_peers) for sh in shards]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=60)
    for sh in shards:
        sh.close()
    if os.environ.get("SOAK_KEEP_REF") == "1":
        ref_sh.close()
    print(f"cfg {cfg_seed}: world {world} per {per} nl {nl} m {m} {dtype} proposal {proposal} repeats {repeats} steps {steps} "
          f"resamples {ref_resamples}: {'ok' if ok else 'FAILED ' + str(errs)[:900]}", flush=True)
    return ok


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    results = [one(first + i) for i in range(count)]
    print(f"{sum(results)} of {count} configurations ok")
    sys.exit(0 if all(results) else 1)
