"""GPU tests of slam_pf_step_auto_batch (csrc/pf_batch.hip): K filter steps per call; runs of at least four consecutive steps that
cannot resample (force = False) go as ONE persistent launch -- every workgroup reduces the step's statistics itself, the bookkeeping
of the lazy resampling is replayed in LDS, the statistics tail of a step runs under the next sweep -- and every other step as
slam_pf_step_auto enqueues it.

The bar is EQUALITY with the step-by-step auto mode (slam_pf_step_auto, itself checked against the synchronous driver and the
oracle in tests/test_gpu_pf.py): poses, maps and log-weights bit for bit, Neff bit for bit, the same resampling steps -- for
K in {1, 4, 16} steps per call and for every kernel form (1 / 2 / 4 / 8 observation ways), with repeats and first sightings
inside a step, empty observation lists, Neff-triggered / forced resamplings between the runs, ragged particle counts, an
exhausted table pool (a queued launch is skipped on the device, the host resamples eagerly and replays its steps one by one) and
legacy calls in between.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])


def scene(nl, seed):
    return np.random.default_rng(seed).uniform(-40, 40, (nl, 2))


def observe(lm, pose, ids, rng):
    dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
    return np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, len(ids)))


def same_filter(a, b, what):
    pa, wa, la = a.download()
    pb, wb, lb = b.download()
    assert np.array_equal(pa, pb), f"{what}: poses differ"
    assert np.array_equal(la, lb), f"{what}: landmarks differ"
    assert np.array_equal(wa, wb), f"{what}: log-weights differ"


def pair(pkg, n, nl, seed, lm_known, dtype="f32"):
    f = {}
    for name in ("batch", "single"):
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm_known, 0.01, 0.1)
        f[name] = pkg.FastSLAM(sh, None, neff_frac=0.75)
    return f


def drive(pkg, f, steps, K, checks, what, proposal=False):
    """``steps``: list of (V, G, z, ids, force).  The batch filter takes them K per call, the single one step by step."""
    for k0 in range(0, len(steps), K):
        chunk = steps[k0:k0 + K]
        batch = pkg.PFShard.prepare_batch([(s[0], s[1]) for s in chunk], [(s[2], s[3]) for s in chunk], [s[4] for s in chunk])
        # (persistent launches are opt-in: the caller vouches that nothing else keeps the device busy -- so the other filter's steps
        #  are enqueued only when this one's are through)
        f["batch"].step_async_batch(batch, 4.0, Q, 0.1, R, proposal=proposal, persistent=True)
        f["batch"].shard.sync()
        for V, G, z, ids, force in chunk:
            f["single"].step_async(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal)
        f["single"].shard.sync()
        if any(k0 <= c < k0 + K for c in checks):
            ra, rb = f["batch"].flush(), f["single"].flush()
            assert ra == rb, f"{what}: Neff / resampled after step {k0 + len(chunk) - 1}: {ra} vs {rb}"
            assert f["batch"].resamples == f["single"].resamples
            same_filter(f["batch"].shard, f["single"].shard, f"{what} after step {k0 + len(chunk) - 1}")


@pytest.mark.parametrize("K", [1, 4, 16])
def test_batch_equals_the_steps_one_by_one_with_repeats_and_first_sightings(pkg, K):
    """The sequential form (a repeated landmark in a step rules the observation ways out): three workgroups of 1024 particles,
    the last one ragged; repeats and first sightings inside a step, an empty observation list, all three resampling rules."""
    n, nl, seed = 3000 + 37, 14, 91
    lm = scene(nl, 17)
    f = pair(pkg, n, nl, seed, lm[:9])                            # 10..14 are first seen later
    rng = np.random.default_rng(6)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(48):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = np.zeros(0, dtype=np.int32) if t == 17 else np.array([1 + t % 9, 1 + (t + 4) % 9, 1 + t % 9, 10 + t % 5, 10 + t % 5, 3])
        z = observe(lm, pose, ids, rng) if len(ids) else np.zeros((2, 0))
        force = None if t % 8 == 7 else (True if t % 24 == 11 else False)      # runs of seven (and three) steps that cannot resample
        steps.append((6.0, 0.01 * (t % 5), z, ids, force))
    drive(pkg, f, steps, K, (4, 19, 33, 47), f"K {K}")
    assert f["single"].resamples >= 8
    assert f["batch"].shard.resample_count() == f["single"].resamples
    for g in f.values():
        g.shard.close()


@pytest.mark.parametrize("n,m,K", [(5000 + 13, 9, 4), (5000 + 13, 2, 16), (40000 + 5, 5, 16), (40000 + 5, 23, 4), (70000, 16, 16),
                                    (131072, 16, 4), (200000 + 3, 16, 16), (262144, 16, 16), (262144, 31, 1)])
def test_batch_equals_the_steps_one_by_one_in_every_kernel_form(pkg, n, m, K):
    """Distinct landmarks per step: 8 ways up to 32 768 particles, 4 up to 65 536, 2 up to 131 072, the sequential form up to
    262 144 (one workgroup of 1024 threads per compute unit).  First sightings, ragged counts, m below / at / above the ring
    depth and the number of ways, one step with a repeated landmark in between (that call takes the sequential form, or the
    step-by-step path where the sequential form does not fit the chip)."""
    nl, seed = 40, 58
    lm = scene(nl, 29)
    f = pair(pkg, n, nl, seed, lm[:30])                           # 31..40 are first seen later
    rng = np.random.default_rng(160 + m)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(2 * K + 8):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = rng.choice(np.arange(1, nl + 1), size=m, replace=False)
        if t == K + 1 and m >= 2:
            ids[-1] = ids[0]
        z = observe(lm, pose, ids, rng)
        force = None if t % 9 == 8 else (True if t % 27 == 13 else False)
        steps.append((6.0, 0.01 * (t % 5), z, ids, force))
    drive(pkg, f, steps, K, (K - 1, len(steps) - 1), f"n {n} m {m} K {K}")
    assert f["single"].resamples >= 1
    for g in f.values():
        g.shard.close()


def test_batch_with_an_exhausted_table_pool_and_legacy_calls_in_between(pkg):
    """One observation per step over 100 landmarks with a resampling at nearly every step needs more live ancestor tables than the
    pool holds: a step HALTS, what is queued behind it -- single steps and, in the last third, persistent launches of six steps -- is
    skipped on the device, the host resamples eagerly and replays the steps from its log.  Legacy entry points in between wait
    for the queue."""
    n, nl, seed = 1500 + 7, 100, 29
    lm = scene(nl, 41)
    f = pair(pkg, n, nl, seed, lm[:90])
    rng = np.random.default_rng(42)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(112):
        pose = np.array([pose[0] + 0.3 * math.cos(0.02 + pose[2]), pose[1] + 0.3 * math.sin(0.02 + pose[2]), pose[2] + 0.3 * math.sin(0.02) / 4.0])
        ids = np.array([(7 * t) % 90 + 1]) if t < 80 else np.array([(3 * t) % 90 + 1, 91 + t % 10, (3 * t) % 90 + 1])
        steps.append((3.0, 0.02, observe(lm, pose, ids, rng), ids, (t % 4 != 3) if t < 80 else (t % 7 == 6)))
    for k0 in range(0, 112, 16):
        drive(pkg, f, steps[k0:k0 + 16], 16, (), "pool")
        if k0 == 32:                                              # a separate legacy update in the middle of the queue
            for g in f.values():
                g.update_known(steps[k0][2], steps[k0][3], R)
        if k0 % 32 == 16:
            same_filter(f["batch"].shard, f["single"].shard, f"pool after step {k0 + 15}")
    ra, rb = f["batch"].flush(), f["single"].flush()
    assert ra == rb and f["batch"].resamples == f["single"].resamples >= 55
    same_filter(f["batch"].shard, f["single"].shard, "pool end")
    for g in f.values():
        g.shard.close()


def test_batch_falls_back_step_by_step_where_the_persistent_launch_does_not_apply(pkg):
    """fp64, the FastSLAM-2.0 proposal and a call that does not allow persistent launches take slam_pf_step_auto K times: still the same filter."""
    n, nl, seed = 2000 + 3, 12, 7
    lm = scene(nl, 19)
    rng = np.random.default_rng(3)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(12):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = rng.choice(np.arange(1, nl + 1), size=4, replace=False)
        steps.append((6.0, 0.01, observe(lm, pose, ids, rng), ids, True if t % 6 == 5 else False))
    for dtype, proposal in (("f64", False), ("f32", True)):
        f = pair(pkg, n, nl, seed, lm, dtype=dtype)
        drive(pkg, f, steps, 4, (3, 11), f"{dtype} proposal {proposal}", proposal=proposal)
        for g in f.values():
            g.shard.close()
    f = pair(pkg, n, nl, seed, lm)
    batch = pkg.PFShard.prepare_batch([(s[0], s[1]) for s in steps], [(s[2], s[3]) for s in steps], [s[4] for s in steps])
    f["batch"].step_async_batch(batch, 4.0, Q, 0.1, R)                  # (persistent launches not allowed: every step on its own)
    for V, G, z, ids, force in steps:
        f["single"].step_async(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=force)
    assert f["batch"].flush() == f["single"].flush()
    same_filter(f["batch"].shard, f["single"].shard, "one by one")
    for g in f.values():
        g.shard.close()


def test_two_filters_batching_side_by_side(pkg):
    """Two independent filters of one process enqueue persistent launches on their own streams: the launches are chained by an
    event (two co-resident grids could starve each other's unstarted workgroups), both filters come out as alone."""
    nl = 20
    lm = scene(nl, 5)
    rng = np.random.default_rng(11)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(32):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = rng.choice(np.arange(1, nl + 1), size=6, replace=False)
        steps.append((6.0, 0.01, observe(lm, pose, ids, rng), ids, None if t % 8 == 7 else False))
    fa = pair(pkg, 150000, nl, 21, lm)
    fb = pair(pkg, 90000 + 11, nl, 22, lm)
    for k0 in range(0, 32, 16):
        chunk = steps[k0:k0 + 16]
        batch = pkg.PFShard.prepare_batch([(s[0], s[1]) for s in chunk], [(s[2], s[3]) for s in chunk], [s[4] for s in chunk])
        fa["batch"].step_async_batch(batch, 4.0, Q, 0.1, R, persistent=True)
        fb["batch"].step_async_batch(batch, 4.0, Q, 0.1, R, persistent=True)
    for f in (fa, fb):
        f["batch"].shard.sync()
    for V, G, z, ids, force in steps:
        fa["single"].step_async(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=force)
        fb["single"].step_async(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=force)
    for f, what in ((fa, "a"), (fb, "b")):
        assert f["batch"].flush() == f["single"].flush()
        same_filter(f["batch"].shard, f["single"].shard, what)
        for g in f.values():
            g.shard.close()
