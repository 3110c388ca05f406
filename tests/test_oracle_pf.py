"""CPU: the FastSLAM oracle (oracle/pf_ref.py) -- the counter-based RNG against the published
Random123 known-answer vectors, filter invariants, and independence from the particle split."""
import math

import numpy as np
import pytest

from oracle import pf_ref as F
from oracle import ekf_ref as O

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    def run(c, k):
        r = F.philox4x32(*[np.array([x], dtype=np.uint32) for x in c], k[0], k[1])
        return [int(x[0]) for x in r]
    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_host_philox_matches_oracle(pkg):
    for step, seed in [(0, 0), (3, 12345), (1000, 0xDEADBEEFCAFE)]:
        assert pkg.philox_uniform(step, 2, seed) == F.uniform1(step, 2, seed)


def test_normals_are_standard():
    e1, e2 = F.normals2(np.arange(200000), 5, 0, 42)
    for e in (e1, e2):
        assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01
    assert abs(np.corrcoef(e1, e2)[0, 1]) < 0.01


def make_scene(rng, nl):
    return rng.uniform(-40, 40, (nl, 2))


def run_filter(shards, steps, lm, rng_obs, resample_every=None):
    """Drive a list of OraclePF shards as one filter (manual collectives)."""
    n_global = shards[0].n_global
    pose_true = np.array([0.0, 0.0, 0.2])
    for s in shards:
        s.set_pose(pose_true)
        s.init_landmarks(lm, 0.01, 0.1)
    nres = 0
    for t in range(steps):
        V, G = 6.0, 0.05
        for s in shards:
            s.predict(V, G, 4.0, Q, 0.1)
        pose_true = np.array([pose_true[0] + V * 0.1 * math.cos(G + pose_true[2]),
                              pose_true[1] + V * 0.1 * math.sin(G + pose_true[2]),
                              pose_true[2] + V * 0.1 * math.sin(G) / 4.0])
        ids = (np.arange(4) + 4 * t) % len(lm) + 1
        dx, dy = lm[ids - 1, 0] - pose_true[0], lm[ids - 1, 1] - pose_true[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose_true[2]]) + rng_obs.normal(0, [[0.1], [math.pi / 180]], (2, 4))
        for s in shards:
            s.update_known(z, ids, R)
        gm, gs, gs2 = F.OraclePF.combine_stats([s.weight_stats() for s in shards])
        for s in shards:
            s.normalize(gm, gs)
        neff = F.OraclePF.neff(gs, gs2)
        if (resample_every and (t + 1) % resample_every == 0) or (resample_every is None and neff < 0.75 * n_global):
            logw_all = np.concatenate([s.logw for s in shards])
            u0 = F.uniform1(nres, F.STREAM_RESAMPLE, shards[0].seed)
            anc = F.OraclePF.ancestors(logw_all, u0)
            recs = {}
            for s in shards:
                mine = anc[s.first:s.first + s.n]
                rem = np.unique(mine[(mine < s.first) | (mine >= s.first + s.n)])
                cols = []
                for g in rem:
                    owner = next(o for o in shards if o.first <= g < o.first + o.n)
                    cols.append(owner.record_of([g - owner.first]))
                recs[s.first] = (mine, rem, np.hstack(cols) if cols else None)
            for s in shards:
                mine, rem, rec = recs[s.first]
                s.resample_apply(mine, rem, rec)
            nres += 1
    return nres, pose_true


def test_filter_tracks_and_weights_are_normalised():
    rng = np.random.default_rng(1)
    lm = make_scene(rng, 12)
    pf = F.OraclePF(4096, 12, seed=7)
    nres, pose_true = run_filter([pf], 25, lm, np.random.default_rng(2))
    assert nres >= 1                                   # Neff-triggered resampling happened
    gm, gs, gs2 = F.OraclePF.combine_stats([pf.weight_stats()])
    pf.normalize(gm, gs)
    assert np.exp(pf.logw).sum() == pytest.approx(1.0, rel=1e-12)
    s = pf.mean_pose_sums()
    est = np.array([s[0], s[1], math.atan2(s[2], s[3])])
    assert np.hypot(*(est[:2] - pose_true[:2])) < 0.5 and abs(est[2] - pose_true[2]) < 0.05
    assert np.all(pf.lm[:, 2] > 0) and np.all(pf.lm[:, 4] > 0)
    assert np.all(pf.lm[:, 2] * pf.lm[:, 4] - pf.lm[:, 3] ** 2 > 0)       # every 2x2 block stays SPD


def test_results_do_not_depend_on_the_split():
    rng = np.random.default_rng(3)
    lm = make_scene(rng, 8)
    one = F.OraclePF(1024, 8, seed=11)
    run_filter([one], 12, lm, np.random.default_rng(4), resample_every=3)
    for G in (2, 4):
        per = 1024 // G
        shards = [F.OraclePF(per, 8, seed=11, first_id=g * per, n_global=1024) for g in range(G)]
        run_filter(shards, 12, lm, np.random.default_rng(4), resample_every=3)
        pose = np.hstack([s.pose for s in shards])
        lms = np.concatenate([s.lm for s in shards], axis=2)
        logw = np.concatenate([s.logw for s in shards])
        assert np.array_equal(pose, one.pose) and np.array_equal(lms, one.lm) and np.array_equal(logw, one.logw)


def test_landmark_update_matches_the_ekf_feature_block():
    """F2 is the reference's Cholesky-form update restricted to the 2x2 feature block: compare with
    the EKF oracle on a state whose pose is known exactly (P_vv = 0, P_vf = 0)."""
    pf = F.OraclePF(1, 1, seed=0)
    pf.set_pose([1.0, -2.0, 0.3])
    pf.lm[0, :, 0] = [9.0, 4.0, 0.3, 0.05, 0.2]
    pf.seen[0] = True
    z = np.array([[10.1], [0.31]])
    x = np.array([1.0, -2.0, 0.3, 9.0, 4.0])
    P = np.zeros((5, 5))
    P[3:, 3:] = [[0.3, 0.05], [0.05, 0.2]]
    xe, Pe = O.update(x, P, z, R, np.array([[1]]))
    pf.update_known(z, [1], R)
    assert np.allclose(pf.lm[0, 0:2, 0], xe[3:5], rtol=1e-12)
    assert np.allclose([pf.lm[0, 2, 0], pf.lm[0, 3, 0], pf.lm[0, 4, 0]], [Pe[3, 3], Pe[3, 4], Pe[4, 4]], rtol=1e-11)
    # weight = N(v; 0, S)
    zp, H = O.predict_observation(x, 1)
    S = H @ P @ H.T + R
    v = np.array([z[0, 0] - zp[0], z[1, 0] - zp[1]])
    want = -0.5 * v @ np.linalg.solve(S, v) - 0.5 * math.log(np.linalg.det(S)) - math.log(2 * math.pi)
    assert pf.logw[0] - (-math.log(1)) == pytest.approx(want, rel=1e-11)


def test_new_landmark_matches_add_features_without_pose_term():
    pf = F.OraclePF(1, 1, seed=0)
    pf.set_pose([1.0, -2.0, 0.3])
    z = np.array([[12.0], [-0.4]])
    pf.update_known(z, [1], R)
    xe, Pe = O.add_features(np.array([1.0, -2.0, 0.3]), np.zeros((3, 3)), z, R)
    assert np.allclose(pf.lm[0, 0:2, 0], xe[3:5], rtol=1e-13)
    assert np.allclose([pf.lm[0, 2, 0], pf.lm[0, 3, 0], pf.lm[0, 4, 0]], [Pe[3, 3], Pe[3, 4], Pe[4, 4]], rtol=1e-12)
    assert pf.logw[0] == 0.0                            # a first sighting carries no likelihood


def test_systematic_resampling_properties():
    rng = np.random.default_rng(9)
    logw = rng.normal(0, 2, 5000)
    anc = F.OraclePF.ancestors(logw, 0.37)
    assert np.all(np.diff(anc) >= 0) and anc.min() >= 0 and anc.max() < 5000
    w = np.exp(logw - logw.max())
    w /= w.sum()
    counts = np.bincount(anc, minlength=5000)
    assert np.all(np.abs(counts - 5000 * w) < 1.0 + 1e-9)             # systematic: floor or ceil of N*w
    assert np.array_equal(F.OraclePF.ancestors(np.zeros(64), 0.5), np.arange(64))   # uniform weights: identity


# ---- N4: FastSLAM-2.0 proposal ------------------------------------------------------------------------
def _proposal_case(n=2048, seed=5):
    lm = np.array([[20.0, 5.0], [15.0, -8.0], [25.0, 12.0], [10.0, 10.0]])
    pf = F.OraclePF(n, 6, seed=seed)
    pf.set_pose([1.0, 2.0, 0.3])
    pf.init_landmarks(lm, 0.01, 0.1)
    V, G, wb, dt = 8.0, 0.05, 4.0, 0.225
    Vt, Gt = V + 0.4, G - 0.03                                    # the controls actually applied
    tp = np.array([1 + Vt * dt * math.cos(Gt + 0.3), 2 + Vt * dt * math.sin(Gt + 0.3), 0.3 + Vt * dt * math.sin(Gt) / wb])
    z = np.array([[math.hypot(*(l - tp[:2])), math.atan2(l[1] - tp[1], l[0] - tp[0]) - tp[2]] for l in lm]).T
    return pf, lm, (V, G, wb, dt), z, tp


def test_proposal_without_observations_is_the_motion_model():
    a, _, (V, G, wb, dt), _, _ = _proposal_case()
    b, _, _, _, _ = _proposal_case()
    a.step_proposal(V, G, wb, Q, dt, np.zeros((2, 0)), [], R)
    b.predict(V, G, wb, Q, dt)
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.logw, b.logw) and a.step == b.step == 1
    # ... and so is a step that only has first sightings: they carry no information about the pose
    a, _, (V, G, wb, dt), z, _ = _proposal_case()
    b, _, _, _, _ = _proposal_case()
    a.step_proposal(V, G, wb, Q, dt, z[:, :2], [5, 6], R)
    b.predict(V, G, wb, Q, dt)
    b.update_known(z[:, :2], [5, 6], R)
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.lm, b.lm) and np.array_equal(a.logw, b.logw)


def test_proposal_raises_neff_and_keeps_the_posterior():
    a, _, (V, G, wb, dt), z, tp = _proposal_case(n=8192)
    b, _, _, _, _ = _proposal_case(n=8192)
    ids = [1, 2, 3, 4]
    a.predict(V, G, wb, Q, dt)
    a.update_known(z, ids, R)
    b.step_proposal(V, G, wb, Q, dt, z, ids, R)

    def summary(pf):
        m, s1, s2 = pf.weight_stats()
        w = np.exp(pf.logw - m) / s1
        return s1 * s1 / s2, (w[None] * pf.pose).sum(1)
    neff1, mean1 = summary(a)
    neff2, mean2 = summary(b)
    assert neff2 > 3 * neff1                                        # the point of FastSLAM 2.0
    assert np.allclose(mean1, mean2, atol=0.02)                     # same target distribution
    assert np.hypot(*(mean2[:2] - tp[:2])) < 0.1
    assert np.all(b.lm[:4, 2] > 0) and np.all(b.lm[:4, 2] * b.lm[:4, 4] - b.lm[:4, 3] ** 2 > 0)


def test_proposal_sequential_form_equals_the_batch_solution():
    """The sequential 2 x 2 Cholesky-form assimilation is checked against the textbook batch form on the stacked
    linear model v = B w + noise:  Sig = (I + B' Sf^-1 B)^-1,  mu = Sig B' Sf^-1 v,  log weight = log N(v; 0, B B' +
    Sf), with the Jacobians taken from the EKF oracle's predict_observation (src/common.jl:139-165)."""
    pf, lm, (V, G, wb, dt), z, _ = _proposal_case(n=4)
    Qf = np.array([[0.3, 0.004], [0.004, 0.003]])                   # a full Q
    ids = [1, 2, 3, 4, 2]                                           # landmark 2 observed twice
    z = np.hstack([z, z[:, 1:2] + [[0.05], [0.002]]])
    before = pf.pose.copy(), pf.lm.copy(), pf.logw.copy()
    pf.step_proposal(V, G, wb, Qf, dt, z, ids, R)
    e1, e2 = F.normals2(pf.gids, 0, F.STREAM_PREDICT, pf.seed)
    Lq = np.linalg.cholesky(Qf)
    for p in range(4):
        x, y, phi = before[0][:, p]
        s, c = math.sin(G + phi), math.cos(G + phi)
        Gu = np.array([[dt * c, -V * dt * s], [dt * s, V * dt * c], [dt * math.sin(G) / wb, V * dt * math.cos(G) / wb]])
        GL = Gu @ Lq
        pm = np.array([x + V * dt * c, y + V * dt * s, phi + V * dt * math.sin(G) / wb])
        Bs, vs, Sfs = [], [], []
        for i, l1 in enumerate(ids):
            rec = before[1][l1 - 1, :, p]
            xs = np.array([pm[0], pm[1], pm[2], rec[0], rec[1]])
            zp, H = O.predict_observation(xs, 1)
            Pf = np.array([[rec[2], rec[3]], [rec[3], rec[4]]])
            Bs.append(H[:, :3] @ GL)
            Sfs.append(H[:, 3:5] @ Pf @ H[:, 3:5].T + R)
            vs.append([z[0, i] - zp[0], O.mpi_to_pi(z[1, i] - zp[1])])
        B = np.vstack(Bs)
        v = np.concatenate(vs)
        Sf = np.zeros((2 * len(ids), 2 * len(ids)))
        for i, blk in enumerate(Sfs):
            Sf[2 * i:2 * i + 2, 2 * i:2 * i + 2] = blk
        Sig = np.linalg.inv(np.eye(2) + B.T @ np.linalg.solve(Sf, B))
        mu = Sig @ B.T @ np.linalg.solve(Sf, v)
        Sv = B @ B.T + Sf
        lw = -0.5 * v @ np.linalg.solve(Sv, v) - 0.5 * np.linalg.slogdet(Sv)[1] - len(ids) * math.log(2 * math.pi)
        assert pf.logw[p] - before[2][p] == pytest.approx(lw, rel=1e-9)
        w = mu + np.linalg.cholesky(Sig) @ np.array([e1[p], e2[p]])
        u = Lq @ w
        Vn, Gn = V + u[0], G + u[1]
        want = [x + Vn * dt * math.cos(Gn + phi), y + Vn * dt * math.sin(Gn + phi), phi + Vn * dt * math.sin(Gn) / wb]
        assert np.allclose(pf.pose[:, p], want, rtol=1e-10, atol=1e-10)


def test_proposal_does_not_depend_on_the_split():
    one, _, (V, G, wb, dt), z, _ = _proposal_case(n=512)
    one.step_proposal(V, G, wb, Q, dt, z, [1, 2, 3, 4], R)
    shards = []
    for g in range(4):
        full, _, _, _, _ = _proposal_case(n=512)
        s = F.OraclePF(128, 6, seed=5, first_id=128 * g, n_global=512)
        s.pose, s.lm, s.seen = full.pose[:, 128 * g:128 * (g + 1)], full.lm[:, :, 128 * g:128 * (g + 1)], full.seen
        s.step_proposal(V, G, wb, Q, dt, z, [1, 2, 3, 4], R)
        shards.append(s)
    assert np.array_equal(np.hstack([s.pose for s in shards]), one.pose)
    assert np.array_equal(np.concatenate([s.lm for s in shards], axis=2), one.lm)
