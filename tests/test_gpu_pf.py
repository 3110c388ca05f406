"""GPU parity tests of the FastSLAM-1.0 particle path (slam_pf_* C ABI via slam.jl_amd/pf.py)
against the float64 oracle (oracle/pf_ref.py) on the same seeds.

The reference has no particle filter (parity unpinned, see oracle/pf_ref.py); the oracle is pinned by
the Random123 known-answer vectors and by the EKF oracle on the 2x2 feature block.

Tolerances: fp64 1e-9 (values) -- the device and NumPy evaluate the same formulas on the same Philox
words, only libm differs; fp32 2e-4 relative to the quantity's scale.  Index work (ancestor tables,
gather sources, record exchange) must be exact.
"""
import math

import numpy as np
import pytest

from oracle import pf_ref as F

pytestmark = pytest.mark.gpu

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
TOL = {"f64": 1e-9, "f32": 2e-4}


def close(a, b, tol, scale=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    s = scale if scale is not None else max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) <= tol * s


def scene(nl, seed):
    return np.random.default_rng(seed).uniform(-40, 40, (nl, 2))


def observe(lm, pose, ids, rng):
    dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
    return np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, len(ids)))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_predict_update_weights_against_oracle(pkg, dtype):
    n, nl, seed = 3000, 10, 77
    lm = scene(nl, 1)
    sh = pkg.PFShard(n, nl, seed, dtype=dtype)
    orc = F.OraclePF(n, nl, seed)
    for f in (sh, orc):
        f.set_pose([1.0, -2.0, 0.4])
        f.init_landmarks(lm[:7], 0.01, 0.1)              # landmarks 8..10 are first seen later
    rng = np.random.default_rng(2)
    pose = np.array([1.0, -2.0, 0.4])
    tol = TOL[dtype]
    for t in range(6):
        for f in (sh, orc):
            f.predict(6.0, 0.05 * t, 4.0, Q, 0.1)
        pose = np.array([pose[0] + 0.6 * math.cos(0.05 * t + pose[2]), pose[1] + 0.6 * math.sin(0.05 * t + pose[2]),
                         pose[2] + 0.6 * math.sin(0.05 * t) / 4.0])
        ids = np.array([(2 * t) % nl + 1, (2 * t + 1) % nl + 1, 8 + t % 3, (2 * t) % nl + 1])   # repeats + first sightings
        z = observe(lm, pose, ids, rng)
        for f in (sh, orc):
            f.update_known(z, ids, R)
        p, lw, l = sh.download()
        assert close(p, orc.pose, tol), f"pose step {t}"
        assert close(l[:, 0:2], orc.lm[:, 0:2], tol), f"landmark means step {t}"
        assert close(l[:, 2:5], orc.lm[:, 2:5], tol * 10, scale=float(np.max(np.abs(orc.lm[:, 2:5])))), f"landmark cov {t}"
        assert close(lw, orc.logw, tol * 10, scale=max(1.0, float(np.max(np.abs(orc.logw))))), f"log-weights step {t}"
    # statistics, normalisation, mean pose
    gm, s1, s2 = sh.weight_stats()
    om, o1, o2 = orc.weight_stats()
    assert gm == pytest.approx(om, abs=tol * 50) and s1 == pytest.approx(o1, rel=tol * 50) and s2 == pytest.approx(o2, rel=tol * 50)
    sh.normalize(gm, s1)
    orc.normalize(om, o1)
    assert np.exp(sh.download()[1].astype(np.float64)).sum() == pytest.approx(1.0, rel=1e-5)
    assert close(sh.mean_pose_sums(), orc.mean_pose_sums(), tol * 50, scale=1.0)
    sh.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_fused_step_equals_the_separate_calls(pkg, dtype):
    """slam_pf_step = predict + update_known + weight_stats in one sweep: the particles must be BIT-identical
    to the three calls (same Philox words, same arithmetic), the statistics equal up to summation order.
    n is not a multiple of the block size and the id list has adjacent repeats, first sightings and a
    repeat two apart (the record prefetch must not read a stale row)."""
    n, nl, seed = 3000 + 37, 12, 5
    lm = scene(nl, 3)
    a = pkg.PFShard(n, nl, seed, dtype=dtype)
    b = pkg.PFShard(n, nl, seed, dtype=dtype)
    for f in (a, b):
        f.set_pose([0.5, 1.5, -0.2])
        f.init_landmarks(lm[:8], 0.01, 0.1)
    rng = np.random.default_rng(4)
    pose = np.array([0.5, 1.5, -0.2])
    for t in range(5):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = np.array([1 + t % 8, 1 + t % 8, 2 + t % 6, 1 + t % 8, 9 + t % 4, 3, 9 + t % 4])
        z = observe(lm, pose, ids, rng)
        sa = a.step_fused(6.0, 0.01 * t, 4.0, Q, 0.1, z, ids, R)
        b.predict(6.0, 0.01 * t, 4.0, Q, 0.1)
        b.update_known(z, ids, R)
        sb = b.weight_stats()
        pa, wa, la = a.download()
        pb, wb, lb = b.download()
        assert np.array_equal(pa, pb) and np.array_equal(wa, wb) and np.array_equal(la, lb), f"step {t}"
        assert sa[0] == sb[0] and abs(sa[1] - sb[1]) <= 1e-12 * sb[1] and abs(sa[2] - sb[2]) <= 1e-12 * sb[2]
        # and against a plain NumPy evaluation of the same statistics
        w = np.exp(wa.astype(np.float64) - float(wa.max()))
        assert sa[0] == float(wa.max()) and close(sa[1], w.sum(), 1e-12) and close(sa[2], (w * w).sum(), 1e-12)
        if t == 2:                                    # an empty observation list is a pure predict + statistics
            sa = a.step_fused(6.0, 0.0, 4.0, Q, 0.1, np.zeros((2, 0)), np.zeros(0, dtype=np.int32), R)
            b.predict(6.0, 0.0, 4.0, Q, 0.1)
            assert np.array_equal(a.download()[0], b.download()[0]) and sa[0] == b.weight_stats()[0]
    a.close()
    b.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_ancestors_and_resampling_are_exact(pkg, dtype):
    import torch
    n, nl, seed = 5000, 4, 5
    rng = np.random.default_rng(3)
    sh = pkg.PFShard(n, nl, seed, dtype=dtype)
    sh.set_pose([0.0, 0.0, 0.0])
    sh.init_landmarks(scene(nl, 4), 0.02, 0.3)
    sh.predict(5.0, 0.0, 4.0, Q, 0.1)                       # make every particle distinct
    pose0, _, lm0 = sh.download()
    logw = rng.normal(0, 2.0, n).astype(sh.np_dtype)
    t = torch.from_numpy(logw).to(sh.device)
    for u0 in (0.0, 0.37, 0.999):
        anc = sh.ancestors(t, float(logw.max()), u0).cpu().numpy()
        want = F.OraclePF.ancestors(logw.astype(np.float64), u0)
        assert np.all(np.diff(anc) >= 0)
        bad = np.flatnonzero(anc != want)
        assert len(bad) <= 2 and np.all(np.abs(anc[bad] - want[bad]) <= 1)    # cdf rounding at a bin edge at most
    anc_t = sh.ancestors(t, float(logw.max()), 0.37)
    anc = anc_t.cpu().numpy()
    sh.resample_apply(anc_t, None, None)
    pose1, lw1, lm1 = sh.download()
    assert np.array_equal(pose1, pose0[:, anc]) and np.array_equal(lm1, lm0[:, :, anc])
    assert np.allclose(lw1, -math.log(n), rtol=1e-6)
    # uniform weights: the identity table, nothing moves
    z = torch.zeros(n, dtype=t.dtype, device=sh.device)
    ident = sh.ancestors(z, 0.0, 0.5)
    assert np.array_equal(ident.cpu().numpy(), np.arange(n))
    sh.close()


def test_two_shards_with_record_exchange_equal_one_shard(pkg):
    """What two ranks do on a resampling step, played out by hand on one GPU: all-gather of the
    log-weights, the same ancestor table, pack / exchange / apply.  Must equal the unsplit filter."""
    import torch
    n, nl, seed = 4096, 3, 9
    lm = scene(nl, 6)
    full = pkg.PFShard(n, nl, seed, dtype="f32")
    halves = [pkg.PFShard(n // 2, nl, seed, dtype="f32", first=g * (n // 2), n_global=n) for g in range(2)]
    rng = np.random.default_rng(8)
    z = observe(lm, np.array([0.5, 0.0, 0.1]), np.array([1, 2, 3]), rng)
    for f in [full] + halves:
        f.set_pose([0.0, 0.0, 0.1])
        f.init_landmarks(lm, 0.01, 0.2)
        f.predict(5.0, 0.1, 4.0, Q, 0.1)
        f.update_known(z, [1, 2, 3], R)
    pf, lwf, lmf = full.download()
    parts = [h.download() for h in halves]
    assert np.array_equal(np.hstack([p[0] for p in parts]), pf)                 # split-independent RNG
    assert np.array_equal(np.concatenate([p[1] for p in parts]), lwf)
    logw_all = torch.cat([h.logw_tensor() for h in halves])
    gmax = float(logw_all.max().item())
    u0 = pkg.philox_uniform(0, 2, seed)
    full_anc = full.ancestors(full.logw_tensor(), gmax, u0)
    full.resample_apply(full_anc, None, None)
    ancs = [h.ancestors(logw_all, gmax, u0) for h in halves]
    for h in halves:                                   # the whole-filter table every rank computes for itself
        assert torch.equal(h.ancestors_all(logw_all, gmax, u0), torch.cat(ancs))
    assert np.array_equal(torch.cat(ancs).cpu().numpy(), full_anc.cpu().numpy())
    moved = 0
    packs = []
    for g, h in enumerate(halves):
        a = ancs[g].to(torch.int64)
        need = torch.unique(a[(a < h.first) | (a >= h.first + h.n)])
        other = halves[1 - g]
        packs.append((need, other.pack((need - other.first).to(torch.int32))))
        moved += int(need.numel())
    for g, h in enumerate(halves):
        need, rec = packs[g]
        h.resample_apply(ancs[g], need.to(torch.int32) if need.numel() else None, rec if need.numel() else None)
    pf2, lwf2, lmf2 = full.download()
    parts = [h.download() for h in halves]
    assert moved > 0, "the test should exercise particle migration"
    assert np.array_equal(np.hstack([p[0] for p in parts]), pf2)
    assert np.array_equal(np.concatenate([p[2] for p in parts], axis=2), lmf2)
    for f in [full] + halves:
        f.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_driver_end_to_end_against_oracle(pkg, dtype):
    """FastSLAM.step on one GPU vs the oracle driven through the same host logic."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from pf_numpy_shard import NumpyShard
    n, nl, seed = 2048, 8, 33
    lm = scene(nl, 10)
    gpu = pkg.FastSLAM(pkg.PFShard(n, nl, seed, dtype=dtype), None)
    cpu = pkg.FastSLAM(NumpyShard(n, nl, seed), None)
    for f in (gpu, cpu):
        f.shard.set_pose([0.0, 0.0, 0.2])
        f.shard.init_landmarks(lm, 0.01, 0.1)
    rng = np.random.default_rng(12)
    pose = np.array([0.0, 0.0, 0.2])
    tol = TOL[dtype]
    for t in range(8):
        pose = np.array([pose[0] + 0.5 * math.cos(0.05 + pose[2]), pose[1] + 0.5 * math.sin(0.05 + pose[2]),
                         pose[2] + 0.5 * math.sin(0.05) / 4.0])
        ids = (np.arange(3) + 3 * t) % nl + 1
        z = observe(lm, pose, ids, rng)
        ng, dg = gpu.step(5.0, 0.05, 4.0, Q, 0.1, z, ids, R, force_resample=(t == 5))
        nc, dc = cpu.step(5.0, 0.05, 4.0, Q, 0.1, z, ids, R, force_resample=(t == 5))
        assert dg == dc
        assert ng == pytest.approx(nc, rel=max(tol * 100, 1e-7))
        if dg and dtype == "f32":
            break          # after an fp32 resample individual ancestors may differ at bin edges: stop the element-wise comparison
    gpu.normalize(); cpu.normalize()
    # the analytically derived maximum of the normalised log-weights (spares resample() a reduction) is EXACT
    assert gpu._gmax_norm == float(gpu.shard.download(landmarks=False)[1].max())
    assert close(gpu.mean_pose(), cpu.mean_pose(), tol * 100, scale=1.0)
    assert np.hypot(*(gpu.mean_pose()[:2] - pose[:2])) < 0.6
    gpu.shard.close()


def test_unknown_correspondence_driver_against_oracle(pkg):
    """FastSLAM.step_unknown (predict + per-particle association + updates + normalise + resample) on one GPU vs
    the oracle driven through the same host logic; fp64 so that the decisions are identical."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from pf_numpy_shard import NumpyShard
    n, nslots, seed = 2048, 8, 41
    lm = np.array([[12.0, 3.0], [6.0, -9.0], [-10.0, 4.0], [15.0, -2.0], [-4.0, -12.0], [9.0, 11.0]])
    gpu = pkg.FastSLAM(pkg.PFShard(n, nslots, seed, dtype="f64"), None)
    cpu = pkg.FastSLAM(NumpyShard(n, nslots, seed), None)
    for f in (gpu, cpu):
        f.shard.set_pose([0.0, 0.0, 0.2])
        f.shard.clear_landmarks()
    rng = np.random.default_rng(3)
    pose = np.array([0.0, 0.0, 0.2])
    for t in range(7):
        pose = np.array([pose[0] + 0.3 * math.cos(0.02 + pose[2]), pose[1] + 0.3 * math.sin(0.02 + pose[2]),
                         pose[2] + 0.3 * math.sin(0.02) / 4.0])
        ids = np.array([1 + t % 6, 1 + (t + 2) % 6, 1 + (t + 4) % 6])
        z = observe(lm, pose, ids, rng)
        ng, dg = gpu.step_unknown(3.0, 0.02, 4.0, Q, 0.1, z, R, 4.0, 25.0, force_resample=(t == 4))
        nc, dc = cpu.step_unknown(3.0, 0.02, 4.0, Q, 0.1, z, R, 4.0, 25.0, force_resample=(t == 4))
        assert dg == dc and ng == pytest.approx(nc, rel=1e-7)
        pose_g, logw_g, lm_g = gpu.shard.download()
        assert np.array_equal(lm_g[:, 2, :] >= 0, cpu.shard.o.lm[:, 2, :] >= 0), f"step {t}: slot usage differs"
        assert close(pose_g, cpu.shard.o.pose, 1e-9, scale=20.0) and close(logw_g, cpu.shard.o.logw, 1e-8, scale=50.0)
    gpu.normalize(); cpu.normalize()
    assert close(gpu.mean_pose(), cpu.mean_pose(), 1e-7, scale=1.0)
    assert np.hypot(*(gpu.mean_pose()[:2] - pose[:2])) < 1.0
    gpu.shard.close()


def test_full_size_config4_properties(pkg):
    """BASELINE.json config 4 at full size on one GPU: 262144 particles x 512 landmarks, 16 known-id
    observations per step, fp32.  Size-independent properties only."""
    n, nl, seed = 262144, 512, 20240602
    rng = np.random.default_rng(seed)
    lm = rng.uniform(-200, 200, (nl, 2))
    pf = pkg.PFSlamState(n, nl, seed=seed, dtype="f32", distributed=False)
    pf.shard.set_pose([0.0, 0.0, 0.3])
    pf.shard.init_landmarks(lm, 0.01, 0.1)
    pose = np.array([0.0, 0.0, 0.3])
    for t in range(4):
        pose = np.array([pose[0] + 0.2 * math.cos(pose[2]), pose[1] + 0.2 * math.sin(pose[2]), pose[2]])
        ids = (np.arange(16) + 16 * t) % nl + 1
        z = observe(lm, pose, ids, rng)
        neff, did = pf.step(8.0, 0.0, 4.0, Q, 0.025, z, ids, R, force_resample=(t == 2))
        assert 1.0 <= neff <= n * (1 + 1e-6)
    pf.normalize()
    gm, s1, s2 = pf.shard.weight_stats()
    assert math.exp(gm) * s1 == pytest.approx(1.0, rel=1e-4)              # weights sum to one
    mp = pf.mean_pose()
    assert np.hypot(*(mp[:2] - pose[:2])) < 0.5
    # idempotence: resampling uniform weights is the identity permutation
    import torch
    before = pf.shard.download(landmarks=False)[0].copy()
    ident = pf.shard.ancestors(torch.zeros(n, dtype=torch.float32, device=pf.shard.device), 0.0, 0.25)
    pf.shard.resample_apply(ident, None, None)
    assert np.array_equal(pf.shard.download(landmarks=False)[0], before)
    pf.close()


def test_full_size_config4_auto_mode_against_the_synchronous_driver_and_the_oracle(pkg):
    """The entry point bench.py TIMES at C4 -- FastSLAM.step_async (slam_pf_step_auto: tagged-line hand-over of 1024
    workgroups, decision and lazy resampling on the device, ~31 live ancestor tables) -- at the benchmarked shape
    262144 x 512 x 16 observations, fp32, against the host-driven FastSLAM.step on a second filter: poses and every
    sampled landmark record bit-identical, log-weights within 4 ulp, the same resampling steps and count.  The steps
    before the first resampling are also checked against the fp64 oracle on global ids [0, 4096) (the random numbers
    are keyed by global id, particles are independent until a resampling)."""
    import torch
    n, nl, seed, m = 262144, 512, 20240602, 16
    rng = np.random.default_rng(seed)
    lm = rng.uniform(-200, 200, (nl, 2))
    f = {}
    for name in ("auto", "sync"):
        f[name] = pkg.PFSlamState(n, nl, seed=seed, dtype="f32", distributed=False)
        f[name].shard.set_pose([0.0, 0.0, 0.3])
        f[name].shard.init_landmarks(lm, 0.01, 0.1)
    no = 4096
    orc = F.OraclePF(no, nl, seed, first_id=0, n_global=n)
    orc.set_pose([0.0, 0.0, 0.3])
    orc.init_landmarks(lm, 0.01, 0.1)
    sample = torch.tensor(np.r_[0:8, 1000:1008, no - 8:no, 131072 - 4:131072 + 4, n - 8:n], dtype=torch.int32, device="cuda")
    osample = np.r_[0:8, 1000:1008, no - 8:no]
    pose = np.array([0.0, 0.0, 0.3])
    # never / never / never, then the Neff rule, forced, and one forced "no" in between
    schedule = [False, False, False, None, None, True, None, False, None, True, None, None, True, None]
    hist = []
    for t, force in enumerate(schedule):
        pose = np.array([pose[0] + 0.2 * math.cos(pose[2]), pose[1] + 0.2 * math.sin(pose[2]), pose[2]])
        ids = (np.arange(m) + m * t) % nl + 1
        z = observe(lm, pose, ids, rng)
        f["auto"].step_async(8.0, 0.0, 4.0, Q, 0.025, z, ids, R, force_resample=force)
        hist.append(f["sync"].step(8.0, 0.0, 4.0, Q, 0.025, z, ids, R, force_resample=force))
        if t < 3:
            orc.predict(8.0, 0.0, 4.0, Q, 0.025)
            orc.update_known(z, ids, R)
        if t == 2:                                  # no resampling so far: the oracle's particles are the filter's
            f["auto"].flush()
            p, lw, _ = f["auto"].shard.download(landmarks=False)
            assert close(p[:, :no], orc.pose, 2e-5, scale=1.0), "poses against the oracle"
            d_gpu = lw[:no].astype(np.float64) - float(lw[0])
            d_orc = orc.logw - orc.logw[0]
            assert close(d_gpu, d_orc, 2e-4, scale=max(1.0, float(np.abs(d_orc).max()))), "log-weights against the oracle"
            rec = f["auto"].shard.pack(sample[:len(osample)]).cpu().numpy().astype(np.float64)      # [3 + 5 nl, samples]
            got, want = rec[3:].reshape(nl, 5, -1), orc.lm[:, :, osample]
            assert close(got[:, 0:2], want[:, 0:2], 2e-6, scale=200.0), "landmark means against the oracle"
            assert close(got[:, 2:5], want[:, 2:5], 2e-3, scale=float(np.abs(want[:, 2:5]).max())), "landmark covariances"
        if t in (2, 6, 9, len(schedule) - 1):
            neff, did = f["auto"].flush()
            assert did == hist[-1][1], f"step {t}"
            assert neff == pytest.approx(hist[-1][0], rel=1e-6), f"step {t}"
            assert f["auto"].resamples == f["sync"].resamples, f"step {t}"
            pa, wa, _ = f["auto"].shard.download(landmarks=False)
            pb, wb, _ = f["sync"].shard.download(landmarks=False)
            assert np.array_equal(pa, pb), f"step {t}: poses differ"
            assert np.allclose(wa, wb, rtol=0, atol=4 * np.finfo(np.float32).eps * max(1.0, float(np.abs(wb).max()))), f"step {t}"
            if t in (6, len(schedule) - 1):          # (pack materialises the lazily resampled maps: also a legacy call mid-queue)
                assert torch.equal(f["auto"].shard.pack(sample), f["sync"].shard.pack(sample)), f"step {t}: landmark records differ"
    dids = [d for _, d in hist]
    assert f["sync"].resamples == sum(dids) >= 6 and not dids[7] and all(dids[i] for i in (5, 9, 12))
    assert f["auto"].shard.resample_count() == f["sync"].resamples
    for g in f.values():
        g.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_unknown_correspondences_against_oracle(pkg, dtype):
    """per-particle gated nearest-neighbour association over each particle's own landmark slots,
    then updates / new landmarks (slam_pf_update_unknown), against the oracle.  The run starts from an empty map
    (every observation is new), revisits landmarks (matched), adds more later, overflows the slot capacity
    (dropped) and includes an observation inside the outer but outside the inner gate (dropped).  Decisions must be
    identical in fp64; in fp32 the geometry is evaluated in fp32, so a vanishing fraction may flip at a gate."""
    n, nslots, seed = 1500 + 13, 6, 11
    lm = np.array([[12.0, 3.0], [6.0, -9.0], [-10.0, 4.0], [15.0, -2.0], [-4.0, -12.0], [9.0, 11.0], [-13.0, -6.0]])
    sh = pkg.PFShard(n, nslots, seed, dtype=dtype)
    orc = F.OraclePF(n, nslots, seed)
    for f in (sh, orc):
        f.set_pose([0.5, -0.5, 0.3])
        f.clear_landmarks()
    rng = np.random.default_rng(5)
    pose = np.array([0.5, -0.5, 0.3])
    tol = TOL[dtype]
    plan = [[1, 2], [2, 1, 3], [1, 3, 4, 2], [5, 1], [6, 2, 3], [7, 4, 6]]          # 7 > 6 slots: the last new one is dropped
    agree = total = 0
    for t, ids in enumerate(plan):
        for f in (sh, orc):
            f.predict(3.0, 0.02 * t, 4.0, Q, 0.1)
        pose = np.array([pose[0] + 0.3 * math.cos(0.02 * t + pose[2]), pose[1] + 0.3 * math.sin(0.02 * t + pose[2]),
                         pose[2] + 0.3 * math.sin(0.02 * t) / 4.0])
        z = observe(lm, pose, np.array(ids), rng)
        if t == 3:
            z = np.hstack([z, z[:, 1:2] + np.array([[0.35], [0.0]])])      # 3.5 sigma off in range: inside gate2 only
        a = sh.update_unknown(z, R, 4.0, 25.0, want_assoc=True).cpu().numpy()
        ao = orc.update_unknown(z, R, 4.0, 25.0)
        agree += int(np.sum(a == ao))
        total += a.size
        if dtype == "f64":
            assert np.array_equal(a, ao), f"step {t}"
        same = np.all(a == ao, axis=0)                    # compare the state where the decisions agree
        pose_g, logw_g, lm_g = sh.download()
        assert close(pose_g[:, same], orc.pose[:, same], tol, scale=20.0)
        used_o = orc.lm[:, 2, :] >= 0
        assert np.array_equal((lm_g[:, 2, :] >= 0)[:, same], used_o[:, same])
        mask = used_o[:, None, :] & same[None, None, :]
        mask = np.broadcast_to(mask, orc.lm.shape)
        assert close(np.where(mask, lm_g, 0.0), np.where(mask, orc.lm, 0.0), 10 * tol, scale=20.0)
        assert close(logw_g[same], orc.logw[same], 10 * tol, scale=max(1.0, float(np.max(np.abs(orc.logw)))))
    assert agree >= 0.999 * total
    assert int((orc.lm[:, 2, :] >= 0).sum(axis=0).max()) == nslots          # the capacity was reached
    sh.close()


# ---- N4: FastSLAM-2.0 proposal ------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_proposal_step_against_oracle(pkg, dtype):
    """slam_pf_step_proposal vs oracle/pf_ref.py::step_proposal over several steps: a full (non-diagonal) Q, first
    sightings, a landmark observed twice in one call, a landmark first seen AND re-observed in the same call (it must
    not enter the proposal), n not a multiple of the block size."""
    n, nl, seed = 3000 + 11, 10, 91
    lm = scene(nl, 21)
    Qf = np.array([[0.3, 0.004], [0.004, 0.003]])
    sh = pkg.PFShard(n, nl, seed, dtype=dtype)
    orc = F.OraclePF(n, nl, seed)
    for f in (sh, orc):
        f.set_pose([1.0, -2.0, 0.4])
        f.init_landmarks(lm[:6], 0.01, 0.1)
    rng = np.random.default_rng(22)
    pose = np.array([1.0, -2.0, 0.4])
    tol = TOL[dtype]
    for t in range(6):
        g = 0.04 * t - 0.1
        pose = np.array([pose[0] + 0.6 * math.cos(g + pose[2]), pose[1] + 0.6 * math.sin(g + pose[2]),
                         pose[2] + 0.6 * math.sin(g) / 4.0])
        ids = np.array([1 + t % 6, 1 + (t + 3) % 6, 7 + t % 4, 1 + t % 6, 7 + t % 4])
        z = observe(lm, pose, ids, rng)
        stats = sh.step_proposal(6.0, g, 4.0, Qf if t % 2 else Q, 0.1, z, ids, R)
        orc.step_proposal(6.0, g, 4.0, Qf if t % 2 else Q, 0.1, z, ids, R)
        p, lw, l = sh.download()
        assert close(p, orc.pose, tol), f"pose step {t}"
        assert close(l[:, 0:2], orc.lm[:, 0:2], tol), f"landmark means step {t}"
        assert close(l[:, 2:5], orc.lm[:, 2:5], tol * 10, scale=float(np.max(np.abs(orc.lm[:, 2:5])))), f"landmark cov {t}"
        assert close(lw, orc.logw, tol * 10, scale=max(1.0, float(np.max(np.abs(orc.logw))))), f"log-weights step {t}"
        om, o1, o2 = orc.weight_stats()
        assert stats[0] == float(lw.max())
        assert stats[0] == pytest.approx(om, abs=tol * 50) and stats[1] == pytest.approx(o1, rel=tol * 200)
        gm, s1, _ = stats
        sh.normalize(gm, s1)
        orc.normalize(om, o1)
    sh.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_proposal_without_information_is_the_fused_step(pkg, dtype):
    """No observation, or first sightings only: the proposal is the motion model, and the step must be
    slam_pf_step BIT for bit (same Philox words, same arithmetic)."""
    n, nl, seed = 2000 + 3, 6, 17
    lm = scene(nl, 5)
    a = pkg.PFShard(n, nl, seed, dtype=dtype)
    b = pkg.PFShard(n, nl, seed, dtype=dtype)
    for f in (a, b):
        f.set_pose([0.5, 1.5, -0.2])
        f.init_landmarks(lm[:3], 0.01, 0.1)
    none = (np.zeros((2, 0)), np.zeros(0, dtype=np.int32))
    sa = a.step_proposal(6.0, 0.03, 4.0, Q, 0.1, *none, R)
    sb = b.step_fused(6.0, 0.03, 4.0, Q, 0.1, *none, R)
    assert sa == sb
    z = observe(lm, np.array([1.1, 1.4, -0.2]), np.array([4, 5]), np.random.default_rng(0))
    sa = a.step_proposal(6.0, -0.02, 4.0, Q, 0.1, z, np.array([4, 5]), R)
    sb = b.step_fused(6.0, -0.02, 4.0, Q, 0.1, z, np.array([4, 5]), R)
    assert sa == sb
    for x, y in zip(a.download(), b.download()):
        assert np.array_equal(x, y)
    a.close()
    b.close()


def test_proposal_driver_against_oracle_and_neff_gain(pkg):
    """FastSLAM.step(proposal=True) on one GPU vs the oracle driven through the same host logic (fp64), with a
    forced resampling in between; and the reason for FastSLAM 2.0: on the same data the effective sample size
    after an informative observation is several times that of the FastSLAM-1.0 step."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from pf_numpy_shard import NumpyShard
    n, nl, seed = 4096, 8, 63
    lm = scene(nl, 30)
    gpu = pkg.FastSLAM(pkg.PFShard(n, nl, seed, dtype="f64"), None)
    cpu = pkg.FastSLAM(NumpyShard(n, nl, seed), None)
    one = pkg.FastSLAM(pkg.PFShard(n, nl, seed, dtype="f64"), None)           # FastSLAM 1.0 on the same data
    for f in (gpu, cpu, one):
        f.shard.set_pose([0.0, 0.0, 0.2])
        f.shard.init_landmarks(lm, 0.01, 0.1)
    rng = np.random.default_rng(31)
    pose = np.array([0.0, 0.0, 0.2])
    gains = []
    Q = 9.0 * globals()["Q"]                 # three times the control noise: the observations carry real information
    for t in range(8):
        pose = np.array([pose[0] + 0.5 * math.cos(0.05 + pose[2]), pose[1] + 0.5 * math.sin(0.05 + pose[2]),
                         pose[2] + 0.5 * math.sin(0.05) / 4.0])
        ids = (np.arange(3) + 3 * t) % nl + 1
        z = observe(lm, pose, ids, rng)
        ng, dg = gpu.step(5.0, 0.05, 4.0, Q, 0.1, z, ids, R, force_resample=True, proposal=True)
        nc, dc = cpu.step(5.0, 0.05, 4.0, Q, 0.1, z, ids, R, force_resample=True, proposal=True)
        n1, _ = one.step(5.0, 0.05, 4.0, Q, 0.1, z, ids, R, force_resample=True)
        assert dg == dc and ng == pytest.approx(nc, rel=1e-7)
        gains.append(ng / n1)
        pose_g, logw_g, lm_g = gpu.shard.download()
        assert close(pose_g, cpu.shard.o.pose, 1e-9, scale=20.0) and close(lm_g, cpu.shard.o.lm, 1e-8, scale=40.0), f"step {t}"
    assert np.median(gains) > 2.0, gains
    assert close(gpu.mean_pose(), cpu.mean_pose(), 1e-7, scale=1.0)
    assert np.hypot(*(gpu.mean_pose()[:2] - pose[:2])) < 0.6
    for f in (gpu, one):
        f.shard.close()


# ---- lazy resampling ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_lazy_resampling_equals_the_eager_gather(pkg, monkeypatch, dtype):
    """A filter that lives on one shard resamples lazily: poses are permuted, the maps stay where they are behind
    composed ancestor tables and move landmark by landmark when next updated (csrc/pf_legacy.hip, "lazy resampling").
    SLAMHIP_PF_EAGER=1 keeps the eager gather of whole particle records.  Both must give the SAME bits, whatever
    comes in between: landmarks untouched across many resamplings, repeats and first sightings in a call, the
    FastSLAM-2.0 step, the unknown-correspondence sweep and a record pack (which materialise the maps), and more live
    tables than the pool holds (one observation per step over 100 landmarks: the eager fallback)."""
    import torch
    n, nl, seed = 1500 + 7, 100, 29
    lm = scene(nl, 41)
    shards = {}
    for name, flag in (("lazy", "0"), ("eager", "1")):
        monkeypatch.setenv("SLAMHIP_PF_EAGER", flag)
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.0, 0.0, 0.2])
        sh.init_landmarks(lm[:90], 0.01, 0.1)                 # 91..100 are first seen later
        shards[name] = pkg.FastSLAM(sh, None)
    monkeypatch.delenv("SLAMHIP_PF_EAGER", raising=False)
    rng = np.random.default_rng(42)
    pose = np.array([0.0, 0.0, 0.2])

    def same(what):
        a, b = shards["lazy"].shard.download(), shards["eager"].shard.download()
        for x, y, part in zip(a, b, ("pose", "logw", "landmarks")):
            assert np.array_equal(x, y), f"{what}: {part} differ"

    for t in range(90):
        pose = np.array([pose[0] + 0.3 * math.cos(0.02 + pose[2]), pose[1] + 0.3 * math.sin(0.02 + pose[2]),
                         pose[2] + 0.3 * math.sin(0.02) / 4.0])
        if t < 70:                                             # one observation per step: > 64 live tables
            ids = np.array([(7 * t) % 90 + 1])
        else:                                                  # several per step, a repeat, first sightings
            ids = np.array([(3 * t) % 90 + 1, (3 * t + 1) % 90 + 1, (3 * t) % 90 + 1, 91 + t % 10, 91 + t % 10])
        z = observe(lm, pose, ids, rng)
        out = []
        for f in shards.values():
            out.append(f.step(3.0, 0.02, 4.0, Q, 0.1, z, ids, R, force_resample=(t % 3 != 2), proposal=(t % 5 == 4)))
        assert out[0] == out[1], f"step {t}"
        if t % 10 == 9:
            same(f"step {t}")
        if t == 75:                                            # a record pack in the lazy state
            idx = torch.arange(0, 64, dtype=torch.int32, device="cuda")
            assert torch.equal(shards["lazy"].shard.pack(idx), shards["eager"].shard.pack(idx))
    # the unknown-correspondence sweep reads every slot: it must see the materialised maps
    z = observe(lm, pose, np.array([3, 40, 77]), rng)
    assoc = [f.shard.update_unknown(z, R, 4.0, 25.0, want_assoc=True) for f in shards.values()]
    assert torch.equal(assoc[0], assoc[1])
    same("after the unknown-correspondence sweep")
    for f in shards.values():
        f.shard.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_lazy_resampling_random_operation_sequences(pkg, monkeypatch, seed):
    """Random programs over the particle-filter entry points -- known-id steps with random (repeating, sometimes first
    seen) landmark lists, FastSLAM-2.0 steps, separate predict / update_known calls, forced and skipped resamplings,
    unknown-correspondence steps, packs, downloads, landmark re-initialisation -- run on a lazy and on an eager shard
    (SLAMHIP_PF_EAGER=1): the states must stay bit-identical."""
    import torch
    rs = np.random.default_rng(1000 + seed)
    n, nl = 700 + 13 * seed, 40
    lm = scene(nl, 50 + seed)
    f = {}
    for name, flag in (("lazy", "0"), ("eager", "1")):
        monkeypatch.setenv("SLAMHIP_PF_EAGER", flag)
        sh = pkg.PFShard(n, nl, 77 + seed, dtype="f64" if seed % 2 else "f32")
        sh.set_pose([0.0, 0.0, 0.2])
        sh.init_landmarks(lm[:30], 0.01, 0.1)
        f[name] = pkg.FastSLAM(sh, None)
    monkeypatch.delenv("SLAMHIP_PF_EAGER", raising=False)
    pose = np.array([0.0, 0.0, 0.2])
    for t in range(120):
        pose = np.array([pose[0] + 0.3 * math.cos(0.02 + pose[2]), pose[1] + 0.3 * math.sin(0.02 + pose[2]),
                         pose[2] + 0.3 * math.sin(0.02) / 4.0])
        op = rs.integers(0, 10)
        m = int(rs.integers(1, 7))
        ids = rs.integers(1, (36 if t > 40 else 30) + 1, m)            # after step 40 also landmarks never seen before
        z = observe(lm, pose, ids, rs)
        res = rs.random() < 0.6
        pick = rs.integers(0, n, 50).astype(np.int32)
        outs = []
        for g in f.values():
            if op <= 5:
                outs.append(g.step(3.0, 0.02, 4.0, Q, 0.1, z, ids, R, force_resample=res, proposal=(op == 5)))
            elif op == 6:                                                # the separate entry points
                g.predict(3.0, 0.02, 4.0, Q, 0.1)
                g.update_known(z, ids, R)
                outs.append(g.normalize())
                if res:
                    g.resample()
            elif op == 7:                                                # unknown correspondences (materialises the maps)
                outs.append(g.step_unknown(3.0, 0.02, 4.0, Q, 0.1, z[:, :3], R, 4.0, 25.0, force_resample=res))
            elif op == 8:
                outs.append(g.shard.pack(torch.from_numpy(pick).cuda()).cpu().numpy().tobytes())
            else:
                outs.append(tuple(a.tobytes() for a in g.shard.download()))
        assert outs[0] == outs[1], f"seed {seed} step {t} op {op}"
    a, b = f["lazy"].shard.download(), f["eager"].shard.download()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for g in f.values():
        g.shard.close()


# ---- auto mode: the step without the host in the loop -----------------------------------------------------------------
def _compare(a, b, what, exact_logw):
    pa, wa, la = a.download()
    pb, wb, lb = b.download()
    assert np.array_equal(pa, pb), f"{what}: poses differ"
    assert np.array_equal(la, lb), f"{what}: landmarks differ"
    if exact_logw:
        assert np.array_equal(wa, wb), f"{what}: log-weights differ"
    else:                      # the shift is gmax + log(sum): the device's log and the host's may differ in the last bit
        assert np.allclose(wa, wb, rtol=0, atol=4 * np.finfo(wa.dtype).eps * max(1.0, float(np.abs(wb).max()))), what


@pytest.mark.parametrize("proposal", [False, True])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_auto_mode_equals_the_synchronous_driver(pkg, dtype, proposal):
    """FastSLAM.step_async (slam_pf_step_auto: statistics, normalisation, Neff, the decision and the lazy resampling all on
    the device, nothing read back per step) against FastSLAM.step (the host decides after every step): the same
    particles, bit for bit, the same Neff, the same resampling steps -- with Neff-triggered and forced resamplings,
    repeats and first sightings inside a call, and an empty observation list."""
    n, nl, seed = 3000 + 37, 14, 91
    lm = scene(nl, 17)
    f = {}
    for name in ("auto", "sync"):
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm[:9], 0.01, 0.1)                  # 10..14 are first seen later
        f[name] = pkg.FastSLAM(sh, None, neff_frac=0.75)
    rng = np.random.default_rng(6)
    pose = np.array([0.5, 1.5, -0.2])
    hist = []
    for t in range(40):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        if t == 17:
            ids = np.zeros(0, dtype=np.int32)
        else:
            ids = np.array([1 + t % 9, 1 + (t + 4) % 9, 1 + t % 9, 10 + t % 5, 10 + t % 5, 3])
        z = observe(lm, pose, ids, rng) if len(ids) else np.zeros((2, 0))
        force = True if t % 7 == 3 else (False if t % 7 == 5 else None)
        f["auto"].step_async(6.0, 0.01 * (t % 5), 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal)
        hist.append(f["sync"].step(6.0, 0.01 * (t % 5), 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal))
        if t in (4, 19, 39):                                  # read-backs at arbitrary places
            neff, did = f["auto"].flush()
            assert did == hist[-1][1], f"step {t}"
            assert neff == pytest.approx(hist[-1][0], rel=1e-12 if dtype == "f64" else 1e-6)
            assert f["auto"].resamples == f["sync"].resamples
            _compare(f["auto"].shard, f["sync"].shard, f"step {t}", exact_logw=False)
    assert f["sync"].resamples >= 8 and any(d for _, d in hist) and not all(d for _, d in hist)
    assert f["auto"].shard.resample_count() == f["sync"].resamples
    for g in f.values():
        g.shard.close()


@pytest.mark.parametrize("m", [2, 8, 9, 16, 23])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_auto_mode_observation_parallel_kernel(pkg, dtype, m):
    """Small filters (and the shards of a sharded one) take pf_auto_step_par_kernel when no landmark occurs twice in the call:
    a workgroup owns 64 particles, its eight waves take the observations w, w + 8, ..., the log-weight terms are added in
    observation order afterwards.  Against the host-driven FastSLAM.step (the sequential kernel): the same particles bit for
    bit, log-weights within 4 ulp, the same resampling steps -- with first sightings, a particle count that is no multiple
    of 64, fewer / exactly / more observations than waves, Neff-triggered and forced resamplings and a step whose
    repeated landmark sends it back to the sequential kernel."""
    n, nl, seed = 5000 + 13, 40, 57
    lm = scene(nl, 29)
    f = {}
    for name in ("auto", "sync"):
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm[:30], 0.01, 0.1)                 # 31..40 are first seen later
        f[name] = pkg.FastSLAM(sh, None, neff_frac=0.75)
    rng = np.random.default_rng(60 + m)
    pose = np.array([0.5, 1.5, -0.2])
    hist = []
    for t in range(24):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = rng.choice(np.arange(1, nl + 1), size=m, replace=False)
        if t == 11:
            ids[-1] = ids[0]                                  # a repeat: this step takes the sequential kernel
        z = observe(lm, pose, ids, rng)
        force = True if t % 6 == 2 else (False if t % 6 == 4 else None)
        f["auto"].step_async(6.0, 0.01 * (t % 5), 4.0, Q, 0.1, z, ids, R, force_resample=force)
        hist.append(f["sync"].step(6.0, 0.01 * (t % 5), 4.0, Q, 0.1, z, ids, R, force_resample=force))
        if t in (3, 12, 23):
            neff, did = f["auto"].flush()
            assert did == hist[-1][1], f"step {t}"
            assert neff == pytest.approx(hist[-1][0], rel=1e-12 if dtype == "f64" else 1e-6)
            assert f["auto"].resamples == f["sync"].resamples
            _compare(f["auto"].shard, f["sync"].shard, f"m {m} step {t}", exact_logw=False)
    assert f["sync"].resamples >= 4
    for g in f.values():
        g.shard.close()


@pytest.mark.parametrize("dtype,n,m", [("f32", 65536 + 77, 16), ("f32", 131072, 16), ("f32", 70000, 5), ("f32", 60000, 31),
                                       ("f64", 65536, 16), ("f64", 150001, 7)])
def test_auto_mode_observation_ways_on_256_particle_workgroups(pkg, dtype, n, m):
    """Round 4: between the 8-way observation-parallel kernel (up to 49152 particles) and the size at which the sequential sweep
    fills the chip, a step takes pf_auto_step_way_kernel -- a workgroup keeps the sweep's 256 particles and splits the call's
    observations over 4 ways (up to 98304 particles, fp32) or 2 ways (up to 196608; fp64 always 2), each way with the sweep's
    record ring; the log-weight terms are added in observation order.  Against the synchronous driver (FastSLAM.step, which
    only has the sequential kernel): the same particles bit for bit, log-weights within 4 ulp, the same resampling steps -- ragged particle counts,
    first sightings, m below / at / above the ring depth, a step with a repeated landmark (sequential kernel) in between."""
    nl, seed = 40, 58
    lm = scene(nl, 29)
    f = {}
    for name in ("auto", "sync"):
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm[:30], 0.01, 0.1)                 # 31..40 are first seen later
        f[name] = pkg.FastSLAM(sh, None, neff_frac=0.75)
    rng = np.random.default_rng(160 + m)
    pose = np.array([0.5, 1.5, -0.2])
    hist = []
    for t in range(14):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = rng.choice(np.arange(1, nl + 1), size=m, replace=False)
        if t == 7:
            ids[-1] = ids[0]                                  # a repeat: this step takes the sequential kernel
        z = observe(lm, pose, ids, rng)
        force = True if t % 6 == 2 else (False if t % 6 == 4 else None)
        f["auto"].step_async(6.0, 0.01 * (t % 5), 4.0, Q, 0.1, z, ids, R, force_resample=force)
        hist.append(f["sync"].step(6.0, 0.01 * (t % 5), 4.0, Q, 0.1, z, ids, R, force_resample=force))
        if t in (3, 8, 13):
            neff, did = f["auto"].flush()
            assert did == hist[-1][1], f"step {t}"
            assert neff == pytest.approx(hist[-1][0], rel=1e-12 if dtype == "f64" else 1e-6)
            assert f["auto"].resamples == f["sync"].resamples
            _compare(f["auto"].shard, f["sync"].shard, f"n {n} m {m} step {t}", exact_logw=False)
    assert f["sync"].resamples >= 2
    for g in f.values():
        g.shard.close()


@pytest.mark.parametrize("n", [200, 1024 * 256 + 700, 16 * 1024 * 256 + 300])
def test_auto_mode_grid_sizes_of_the_statistics_hand_over(pkg, n):
    """The step kernel's last workgroup collects one tagged statistics line per workgroup, 1024 lines per pass: a grid of ONE
    workgroup (it collects its own line), one of 1027 workgroups (a second pass, a ragged last workgroup) and one of 16386 (a
    SEVENTEENTH pass: round 4's tail kept sixteen pass nodes and dropped the rest without a word; now 64, and the host refuses
    what lies beyond) against the synchronous driver -- same particles, same Neff, same resampling steps; legacy calls in between
    reuse the partials buffer (their lines carry no tag of the next auto step)."""
    nl, seed, dtype = 6, 13, "f32"
    lm = scene(nl, 23)
    f = {}
    for name in ("auto", "sync"):
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm, 0.01, 0.1)
        f[name] = pkg.FastSLAM(sh, None, neff_frac=0.75)
    rng = np.random.default_rng(8)
    pose = np.array([0.5, 1.5, -0.2])
    for t in range(9):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = np.array([1 + t % 6, 1 + (t + 2) % 6, 1 + t % 6])
        z = observe(lm, pose, ids, rng)
        force = True if t in (2, 6) else None
        f["auto"].step_async(6.0, 0.02, 4.0, Q, 0.1, z, ids, R, force_resample=force)
        want = f["sync"].step(6.0, 0.02, 4.0, Q, 0.1, z, ids, R, force_resample=force)
        if t in (3, 8):
            neff, did = f["auto"].flush()
            assert did == want[1] and neff == pytest.approx(want[0], rel=1e-6), f"step {t}"
            _compare(f["auto"].shard, f["sync"].shard, f"n {n} step {t}", exact_logw=False)
        if t == 4:                                            # a legacy call on both: statistics through the same buffer
            sa, sb = f["auto"].shard.weight_stats(), f["sync"].shard.weight_stats()
            assert sa[0] == sb[0] and sa[1] == pytest.approx(sb[1], rel=1e-12)
    assert f["auto"].resamples == f["sync"].resamples >= 2
    for g in f.values():
        g.shard.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_auto_mode_with_an_exhausted_table_pool_and_legacy_calls_in_between(pkg, dtype):
    """One observation per step over 100 landmarks with a resampling at every step needs more live ancestor tables than
    the pool holds: the device then HALTS the step, the host resamples eagerly and re-enqueues what was queued behind
    it (slam_pf_flush / the next slam_pf_step_auto do that by themselves on one shard).  Legacy entry points mixed in
    (downloads, a separate update_known, a pack) wait for the queue and see the same filter."""
    import torch
    n, nl, seed = 1500 + 7, 100, 29
    lm = scene(nl, 41)
    f = {}
    for name in ("auto", "sync"):
        sh = pkg.PFShard(n, nl, seed, dtype=dtype)
        sh.set_pose([0.0, 0.0, 0.2])
        sh.init_landmarks(lm[:90], 0.01, 0.1)
        f[name] = pkg.FastSLAM(sh, None)
    rng = np.random.default_rng(42)
    pose = np.array([0.0, 0.0, 0.2])
    for t in range(110):
        pose = np.array([pose[0] + 0.3 * math.cos(0.02 + pose[2]), pose[1] + 0.3 * math.sin(0.02 + pose[2]),
                         pose[2] + 0.3 * math.sin(0.02) / 4.0])
        ids = np.array([(7 * t) % 90 + 1]) if t < 80 else np.array([(3 * t) % 90 + 1, 91 + t % 10, (3 * t) % 90 + 1])
        z = observe(lm, pose, ids, rng)
        f["auto"].step_async(3.0, 0.02, 4.0, Q, 0.1, z, ids, R, force_resample=(t % 4 != 3))
        f["sync"].step(3.0, 0.02, 4.0, Q, 0.1, z, ids, R, force_resample=(t % 4 != 3))
        if t == 30:                                           # a separate legacy update in the middle of the queue
            for g in f.values():
                g.update_known(z, ids, R)
        if t == 55:
            idx = torch.arange(0, 64, dtype=torch.int32, device="cuda")
            assert torch.equal(f["auto"].shard.pack(idx), f["sync"].shard.pack(idx))
        if t % 20 == 19:
            _compare(f["auto"].shard, f["sync"].shard, f"step {t}", exact_logw=False)
    f["auto"].flush()
    assert f["auto"].resamples == f["sync"].resamples
    _compare(f["auto"].shard, f["sync"].shard, "end", exact_logw=False)
    for g in f.values():
        g.shard.close()


def test_whole_filter_calls(pkg):
    """SURVEY 8b's slam_pf_resample / slam_pf_get_mean_pose / slam_pf_get_weights against the driver's pieces."""
    n, nl, seed = 4096, 6, 3
    lm = scene(nl, 5)
    a, b = pkg.PFShard(n, nl, seed, dtype="f64"), pkg.PFShard(n, nl, seed, dtype="f64")
    drv = pkg.FastSLAM(b, None, neff_frac=0.6)
    rng = np.random.default_rng(8)
    pose = np.array([0.0, 0.0, 0.1])
    for sh in (a, b):
        sh.set_pose(pose)
        sh.init_landmarks(lm, 0.01, 0.1)
    did_any = False
    for t in range(12):
        pose = np.array([pose[0] + 0.5 * math.cos(pose[2]), pose[1] + 0.5 * math.sin(pose[2]), pose[2]])
        ids = (np.arange(3) + 3 * t) % nl + 1
        z = observe(lm, pose, ids, rng)
        a.predict(5.0, 0.0, 4.0, Q, 0.1)
        a.update_known(z, ids, R)
        w = a.weights()
        lw = a.download(landmarks=False)[1]
        assert np.allclose(w, np.exp(lw), rtol=1e-14)
        mp = a.mean_pose()
        did = a.resample_if_needed(0.6)
        _neff, did_b = drv.step(5.0, 0.0, 4.0, Q, 0.1, z, ids, R)
        assert did == did_b
        did_any = did_any or did
        if not did:                                            # (after a resampling the weights are uniform again)
            assert np.allclose(mp, drv.mean_pose(), rtol=1e-10, atol=1e-12)
        assert a.weights().sum() == pytest.approx(1.0, rel=1e-12)
    assert did_any
    pa, wa, la = a.download()
    pb, wb, lb = b.download()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb) and np.allclose(wa, wb, rtol=0, atol=1e-13)
    a.close()
    b.close()


class _Rank:
    """The world of one in-process rank (FastSLAM only checks the slice layout against it)."""
    def __init__(self, rank, world):
        self.rank, self.world = rank, world


@pytest.mark.parametrize("proposal", [False, True])
@pytest.mark.parametrize("dtype,world", [("f32", 2), ("f64", 3)])
def test_sharded_filter_resamples_on_the_device(pkg, dtype, world, proposal):
    """A filter sharded over `world` ranks with PEERS attached (slam_pf_attach_peers; here the ranks are shards of this
    process on one card, one host thread each -- the same kernels and the same peer table as IPC-mapped ranks): the
    per-step scalars travel through the ranks' inboxes and EVERY resampling happens on the device -- cdf over the
    gathered weights, global ancestors, remote poses / table entries read from their owners, remote landmark records
    read lazily by the sweep.  Regime: a resampling on (almost) every step, forced and Neff-triggered, with repeats
    and first sightings.  The shards together must be the one-rank synchronous filter bit for bit, with ZERO
    SLAM_PF_HALTED returns."""
    import threading
    per, nl, seed = 1365, 14, 77
    n = per * world
    lm = scene(nl, 19)
    ref_shard = pkg.PFShard(n, nl, seed, dtype=dtype)
    ref_shard.set_pose([0.5, 1.5, -0.2])
    ref_shard.init_landmarks(lm[:9], 0.01, 0.1)
    ref = pkg.FastSLAM(ref_shard, None, neff_frac=0.75)
    shards = [pkg.PFShard(per, nl, seed, dtype=dtype, first=r * per, n_global=n) for r in range(world)]
    for sh in shards:
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm[:9], 0.01, 0.1)
    pkg.attach_local_peers(shards)
    ranks = [pkg.FastSLAM(sh, _Rank(r, world), neff_frac=0.75) for r, sh in enumerate(shards)]
    rng = np.random.default_rng(6)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(36):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        ids = np.zeros(0, dtype=np.int32) if t == 17 else np.array([1 + t % 9, 1 + (t + 4) % 9, 1 + t % 9, 10 + t % 5, 10 + t % 5, 3])
        z = observe(lm, pose, ids, rng) if len(ids) else np.zeros((2, 0))
        force = False if t % 9 == 5 else (None if t % 3 == 2 else True)
        steps.append((0.01 * (t % 5), z, ids, force))
    hist = [ref.step(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal) for g, z, ids, force in steps]
    want = ref_shard.download()
    got, errs = [None] * world, []

    def drive(r):
        try:
            f = ranks[r]
            assert f.shard.peer_selftest(10000)      # (collective) every peer's inbox write arrives
            for t, (g, z, ids, force) in enumerate(steps):
                f.step_async(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force, proposal=proposal)
                if t in (7, 20):
                    neff, did = f.flush()
                    assert did == hist[t][1] and neff == pytest.approx(hist[t][0], rel=1e-12 if dtype == "f64" else 1e-6), f"step {t}"
            neff, did = f.flush()
            assert did == hist[-1][1] and neff == pytest.approx(hist[-1][0], rel=1e-12 if dtype == "f64" else 1e-6)
            assert f.resamples == ref.resamples
            got[r] = f.shard.download()              # collective: remote records come home first
        except BaseException as e:                    # noqa: BLE001 -- reported by the main thread
            errs.append((r, e))

    th = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errs, errs
    assert all(g is not None for g in got)
    assert ref.resamples >= 20
    info = [sh.comm_info() for sh in shards]
    assert all(i["halts"] == 0 and i["peers"] and i["world"] == world and i["resamples"] == ref.resamples for i in info), info
    assert np.array_equal(np.hstack([g[0] for g in got]), want[0]), "poses differ"
    assert np.array_equal(np.concatenate([g[2] for g in got], axis=2), want[2]), "landmarks differ"
    wa = np.concatenate([g[1] for g in got])
    assert np.allclose(wa, want[1], rtol=0, atol=4 * np.finfo(wa.dtype).eps * max(1.0, float(np.abs(want[1]).max())))
    th = [threading.Thread(target=sh.detach_peers) for sh in shards]          # collective, too
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=60)
    for sh in shards + [ref_shard]:
        sh.close()


@pytest.mark.parametrize("dtype,world,per", [("f32", 4, 16384), ("f64", 2, 2048), ("f32", 2, 3 * 1024), ("f32", 2, 65536), ("f32", 3, 66 * 1024),
                                             ("f32", 4, 320 * 1024)])
def test_sharded_normalisation_is_invariant_in_the_number_of_ranks(pkg, dtype, world, per):
    """identical results for any number of ranks -- INCLUDING the normalisation (round 4).  The weight statistics
    are the root of ONE fixed radix-4 tree over the global particle index (csrc/pf_device.h: WRec): every rank writes its
    1024-particle records into every rank's inbox and all ranks reduce the same sequence, so a sharded filter whose slices
    are multiples of 1024 particles has the ONE-RANK auto filter's log-weights BIT FOR BIT, not within ulps -- also where
    the two use different step kernels (65536 particles: the one-rank filter takes the 4-way kernel and its 256-particle lines, the
    four 16384-particle shards the 8-way kernel and 64-particle leaves; 131072: 2 ways against 4; 202752: the sequential sweep
    against three 2-way shards; 1 310 720 particles: more than 1024 statistics lines per rank AND more than 1024 records in the
    inbox -- the tail's second pass on both sides, the shape of the weak-scaling filter on 5+ GPUs)."""
    import threading
    nl, seed = 14, 91
    n = per * world
    lm = scene(nl, 19)

    def fresh(sh):
        sh.set_pose([0.5, 1.5, -0.2])
        sh.init_landmarks(lm[:9], 0.01, 0.1)
    one = pkg.PFShard(n, nl, seed, dtype=dtype)
    fresh(one)
    ref = pkg.FastSLAM(one, None, neff_frac=0.75)
    shards = [pkg.PFShard(per, nl, seed, dtype=dtype, first=r * per, n_global=n) for r in range(world)]
    for sh in shards:
        fresh(sh)
    pkg.attach_local_peers(shards)
    ranks = [pkg.FastSLAM(sh, _Rank(r, world), neff_frac=0.75) for r, sh in enumerate(shards)]
    rng = np.random.default_rng(8)
    pose = np.array([0.5, 1.5, -0.2])
    steps = []
    for t in range(24):
        pose = np.array([pose[0] + 0.6 * math.cos(pose[2]), pose[1] + 0.6 * math.sin(pose[2]), pose[2]])
        # distinct landmarks per call (the observation-parallel kernel), a repeat every fifth step (the sequential one), first sightings
        ids = np.array([1 + t % 9, 1 + (t + 4) % 9, 10 + t % 5, 1 + (t + 2) % 9]) if t % 5 else np.array([1 + t % 9, 3 + t % 5, 1 + t % 9])
        z = observe(lm, pose, ids, rng)
        force = False if t % 4 == 1 else (None if t % 3 == 2 else True)      # never / Neff rule / always
        steps.append((0.01 * (t % 5), z, ids, force))
    snaps = {}
    for t, (g, z, ids, force) in enumerate(steps):
        ref.step_async(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force)
        if t in (1, 9, 23):                       # (step 1 and 9 do not resample: the weights there are NOT uniform)
            ref.flush()
            snaps[t] = one.download()
    got, errs = {}, []

    def drive(r):
        try:
            f = ranks[r]
            assert f.shard.peer_selftest(10000)
            for t, (g, z, ids, force) in enumerate(steps):
                f.step_async(6.0, g, 4.0, Q, 0.1, z, ids, R, force_resample=force)
                if t in (1, 9, 23):
                    f.flush()
                    got[(r, t)] = f.shard.download()          # collective
            assert f.resamples == ref.resamples
        except BaseException as e:                    # noqa: BLE001 -- reported by the main thread
            errs.append((r, e))

    th = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errs, errs
    assert ref.resamples >= 8
    for t in (1, 9, 23):
        want = snaps[t]
        parts = [got[(r, t)] for r in range(world)]
        assert np.array_equal(np.hstack([g[0] for g in parts]), want[0]), f"poses differ at step {t}"
        assert np.array_equal(np.concatenate([g[2] for g in parts], axis=2), want[2]), f"landmarks differ at step {t}"
        wa = np.concatenate([g[1] for g in parts])
        assert np.array_equal(wa, want[1]), f"log-weights differ at step {t}: max |d| {np.max(np.abs(wa - want[1])):.3e}"
        if t != 23:
            assert np.ptp(want[1]) > 0                # (these are weighted particle sets, not the uniform weights after a resampling)
    assert all(sh.comm_info()["halts"] == 0 for sh in shards)
    th = [threading.Thread(target=sh.detach_peers) for sh in shards]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=60)
    for sh in shards + [one]:
        sh.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_workers(tmp_path, tag, world, mode, regime, extra_env=None):
    """`world` processes of tests/pf_auto_worker.py, all on card 0; returns the concatenated filter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / tag)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "pf_auto_worker.py"), out, mode, regime],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    parts = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    return dict(pose=np.hstack([p["pose"] for p in parts]), lm=np.concatenate([p["lm"] for p in parts], axis=2),
                logw=np.concatenate([p["logw"] for p in parts]), resamples=int(parts[0]["resamples"]),
                neff=parts[0]["neff"], halts=[int(p["halts"]) for p in parts], peers=[int(p["peers"]) for p in parts])


@pytest.mark.parametrize("regime", ["mixed", "every_step"])
@pytest.mark.parametrize("nranks", [2, 4])
def test_auto_mode_two_ranks_on_one_card(pkg, tmp_path, nranks, regime):
    """The sharded auto mode rehearsed with two (and four) PROCESSES on ONE card: PFSlamState under torch.distributed
    (gloo, for the set-up's object all-gather of the peer blobs only) attaches the ranks as peers through IPC handles
    (slam_pf_attach_peers), the ranks' GPUs -- here the same card -- exchange their per-step scalars through each other's
    inboxes and resample ON THE DEVICE: no step may halt (the count of SLAM_PF_HALTED returns must be ZERO), in the
    `every_step` regime every step resamples.  The shards together must equal the one-rank synchronous filter."""
    s = _run_workers(tmp_path, "sync", 1, "sync", regime)
    a = _run_workers(tmp_path, "auto", nranks, "auto", regime)
    assert a["peers"] == [1] * nranks and a["halts"] == [0] * nranks, (a["peers"], a["halts"])
    assert a["resamples"] == s["resamples"] >= (13 if regime == "every_step" else 3)
    assert np.array_equal(a["pose"], s["pose"]) and np.array_equal(a["lm"], s["lm"])
    assert np.allclose(a["logw"], s["logw"], rtol=0, atol=1e-12)
    assert np.allclose(a["neff"], s["neff"], rtol=1e-10)
    # ... and the ONE-RANK AUTO filter bit for bit, log-weights and Neff included (round 4: the statistics are the root of one
    # fixed tree over the global particle index; the slices here are 2048 / 1024 particles)
    a1 = _run_workers(tmp_path, "auto1", 1, "auto", regime)
    assert np.array_equal(a["pose"], a1["pose"]) and np.array_equal(a["lm"], a1["lm"])
    assert np.array_equal(a["logw"], a1["logw"]), float(np.max(np.abs(a["logw"] - a1["logw"])))
    assert np.array_equal(a["neff"], a1["neff"])


def test_auto_mode_two_ranks_without_peers_halts_and_resumes(pkg, tmp_path):
    """The fallback (SLAMHIP_PF_PEERS=0; also what several nodes get): per-step scalars through the shared pinned page, a
    resampling step HALTS, the hosts resample through the collectives (gloo here), resume, and the skipped steps are
    replayed.  Same filter; the halts are counted."""
    s = _run_workers(tmp_path, "sync", 1, "sync", "mixed")
    a = _run_workers(tmp_path, "halt", 2, "auto", "mixed", extra_env={"SLAMHIP_PF_PEERS": "0"})
    assert a["peers"] == [0, 0] and all(h >= 3 for h in a["halts"]), (a["peers"], a["halts"])
    assert a["resamples"] == s["resamples"] >= 3
    assert np.array_equal(a["pose"], s["pose"]) and np.array_equal(a["lm"], s["lm"])
    assert np.allclose(a["logw"], s["logw"], rtol=0, atol=1e-12)
    assert np.allclose(a["neff"], s["neff"], rtol=1e-10)


def test_sharded_filter_generations_and_the_weak_scaling_shape_attach_their_peers(pkg):
    """The IPC mappings of the sharded filter (csrc/pf_peers.hip: PF_IPC_MAX_BYTES), two processes on one card (tools/ipc_gen_test.py):
    `gen`  attach -> steps with resamplings -> detach -> close, three GENERATIONS of filters of changing buffer sizes in the
           same pair of processes (every generation re-exports and re-imports every buffer);
    `big`  BASELINE config 4 PER GPU, the weak-scaling shape: 262144 particles x 512 landmarks per rank = 2.5 GiB per landmark
           buffer.  hipIpcOpenMemHandle of an allocation above 2 GiB never returns on this runtime, so round 3 refused this
           peer; the records now live in chunks of at most 1 GiB and the filter ATTACHES (and resamples on the device)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, npeers in (("gen", 3), ("big:524288", 1)):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "ipc_gen_test.py"), mode], capture_output=True, text=True,
                           timeout=240)
        out = r.stdout + r.stderr
        assert "exit codes [0, 0]" in out and "HUNG" not in out, out[-3000:]
        assert out.count("peers True") == 2 * npeers and "peers False" not in out, out[-3000:]
        assert out.count("halts 0") == 2 * npeers, out[-3000:]


def test_rccl_collectives_of_the_sharded_flow_on_a_one_rank_group(pkg, tmp_path):
    """The `nccl` (= RCCL) branch of the sharded driver has no multi-GPU box to run on here; a ONE-rank RCCL process group
    does exist on a one-GPU box: the filter is driven through FastSLAM with the multi-rank resampling flow forced
    (all-gather of the log-weights, the ancestor table of the whole filter, pack, all-to-all of records, apply), so every
    collective call of slam.jl_amd/pf.py:TorchComm executes on RCCL with device tensors.  Same particles as the plain
    single-process filter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("sync", "nccl1"):
        out = str(tmp_path / f"r_{mode}")
        env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "pf_auto_worker.py"), out, mode, "mixed"], env=env,
                           capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = np.load(out + ".rank0.npz")
    a, s = outs["nccl1"], outs["sync"]
    assert int(a["resamples"]) == int(s["resamples"]) >= 3
    assert np.array_equal(a["pose"], s["pose"]) and np.array_equal(a["lm"], s["lm"])
    assert np.allclose(a["logw"], s["logw"], rtol=0, atol=1e-12)


def test_bench_fastslam_leg_with_two_ranks_on_one_card():
    """exactly the path `bench.py --gpus N` takes on a multi-GPU node, rehearsed with two processes on ONE card
    (SLAM_BENCH_REHEARSE: gloo as the control plane, both ranks on device 0, a smaller filter): the sharded filter is checked
    against a one-rank filter before anything is timed (comm.parity_vs_one_rank, for the device-side exchange bit for bit and for
    the halting flow over the collectives), the self-test of the peers is reported, and BOTH exchange paths are timed in the
    same run (regimes_peers, regimes_rccl)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SLAM_BENCH_REHEARSE="1", SLAM_BENCH_NP="65536", HSA_ENABLE_IPC_MODE_LEGACY="0", SLAM_BENCH_PF_BUDGET_S="400")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pmc",
                        "--no-configs", "--landmarks", "1000", "--obs", "16", "--prewarm-ms", "10"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    f = line["fastslam"]
    assert "error" not in f, f
    c = f["comm"]
    par = c["parity_vs_one_rank"]
    assert par["peers"]["available"] and par["peers"]["ok"] and par["peers"]["poses_equal"] and par["peers"]["logw_max_ulp"] == 0.0, par
    assert par["rccl"]["ok"] and par["rccl"]["poses_equal"] and par["rccl"]["logw_max_ulp"] <= 4.0, par
    assert par["peers"]["resamples"][0] == par["peers"]["resamples"][1] >= 1
    assert c["selftest_ok"] is True and c["peers_attached"] is True and c["timed_paths"] == ["peers", "rccl"]
    for key in ("regimes_peers", "regimes_rccl"):
        assert {"no_resample", "every_step", "neff_triggered"} <= set(f[key]), f[key]
        assert all(v["ms_per_step"] > 0 for v in f[key].values())
    assert f["comm_rccl"]["peers_attached"] is False and f["comm_rccl"]["halts"] >= 1      # (the halting flow halted: it ran)
    assert f["regimes"] == f["regimes_peers"] and f["weak_scaling"]["peers_attached"] is True
