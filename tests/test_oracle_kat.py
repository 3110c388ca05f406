"""Known-answer tests that pin the CPU oracle (KAT-8..KAT-10: tests/kat_vectors.py).

The reference ships no tests or golden vectors and cannot be executed here, so
these hand-derived values (from the formulas at the cited reference lines) are
the only external pins; everything else is oracle self-consistency.
"""
import math

import numpy as np
import pytest

from oracle import ekf_ref as O
from tests import kat_vectors as KV

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])


def test_kat1_predict_observation():
    # src/common.jl:139-165
    x = np.array([0.0, 0.0, 0.0, 10.0, 0.0])
    z, H = O.predict_observation(x, 1)
    assert np.array_equal(z, [10.0, 0.0])
    assert np.allclose(H, [[-1, 0, 0, 1, 0], [0, -0.1, -1, 0, 0.1]], rtol=0, atol=1e-16)


def test_kat2_compute_association():
    # src/data-association.jl:53-63 with P = I5
    x = np.array([0.0, 0.0, 0.0, 10.0, 0.0])
    P = np.eye(5)
    _, H = O.predict_observation(x, 1)
    S = H @ P @ H.T + R
    assert np.allclose(S, np.diag([2.01, 1.02 + (math.pi / 180) ** 2]), atol=1e-15)
    nis, nd = O.compute_association(x, P, np.array([10.5, 0.02]), R, 1)
    assert nis == pytest.approx(0.12477014923494524, rel=1e-13)
    assert nd == pytest.approx(0.843006098545911, rel=1e-13)
    nis_s, nd_s = O.compute_association_sparse(x, P, np.array([10.5, 0.02]), R, 1)
    assert nis_s == pytest.approx(nis, rel=1e-13)
    assert nd_s == pytest.approx(nd, rel=1e-13)


def test_kat13_non_symmetric_innovation_covariance():
    # src/data-association.jl:59-60: S = H*P*H' + R is inverted WITHOUT being symmetrised; hand-derived in tests/kat_vectors.py
    from tests import kat_vectors as KV
    x, P, R13, z, nis, nd, nis_s, gate1, gate2 = KV.kat13()
    _, H = O.predict_observation(x, 1)
    S = H @ P @ H.T + R13
    assert S[0, 1] == pytest.approx(0.5, abs=1e-15) and S[1, 0] == pytest.approx(-0.3, abs=1e-15)
    for fn in (O.compute_association, O.compute_association_sparse):
        got = fn(x, P, z, R13, 1)
        assert got[0] == pytest.approx(nis, rel=1e-13) and got[1] == pytest.approx(nd, rel=1e-13)
    assert nis < gate1 < nis_s
    for fn in (O.associate, O.associate_sparse):
        zf, idf, zn = fn(x, P, z.reshape(2, 1), R13, gate1, gate2)
        assert idf.tolist() == [[1]] and zn.shape[1] == 0          # matched (a symmetrised S would give nis_s > gate1: dropped)
        zf, idf, zn = fn(x, P, z.reshape(2, 1), R13, 0.5 * nis, gate2)
        assert idf.shape[1] == 0 and zn.shape[1] == 0              # dead band: gate1 <= nis <= gate2


def test_kat3_predict():
    # src/ekf.jl:8-43 from x = 0, P = 0
    x = np.zeros(3)
    P = np.zeros((3, 3))
    x, P = O.predict(x, P, 8.0, 0.0, 4.0, Q, 0.025)
    assert np.allclose(x, [0.2, 0.0, 0.0], atol=1e-16)
    s3 = (3 * math.pi / 180) ** 2
    expect = np.array([[0.025 ** 2 * 0.25, 0, 0],
                       [0, 0.2 ** 2 * s3, 0.2 * 0.05 * s3],
                       [0, 0.2 * 0.05 * s3, 0.05 ** 2 * s3]])
    assert np.allclose(P, expect, rtol=1e-13, atol=0)
    assert P[0, 0] == pytest.approx(1.5625e-4, rel=1e-12)
    assert P[1, 1] == pytest.approx(1.09662271e-4, rel=1e-8)
    assert P[1, 2] == pytest.approx(2.74155678e-5, rel=1e-8)
    assert P[2, 2] == pytest.approx(6.85389195e-6, rel=1e-8)


def test_kat4_add_features():
    # src/ekf.jl:84-122 from x = 0, P = 0, z = (10, 0)
    x, P = O.add_features(np.zeros(3), np.zeros((3, 3)), np.array([[10.0], [0.0]]), R)
    assert np.allclose(x, [0, 0, 0, 10, 0], atol=1e-16)
    assert np.allclose(P[3:, 3:], np.diag([R[0, 0], 100 * R[1, 1]]), rtol=1e-14, atol=1e-20)
    assert np.all(P[0:3, :] == 0) and np.all(P[:, 0:3] == 0)


def _table_fn(table):
    def fn(x, P, z, Rm, j):
        return table[j - 1]
    return fn


def test_kat5_gating_rules():
    # src/data-association.jl:21-50.  Scripted (nis, nd) tables isolate the scan logic.
    z = np.array([[1.0], [0.1]])
    x3 = np.zeros(3 + 2 * 3)
    P3 = np.eye(9)
    # Nf = 0: outer stays Inf > gate2 -> new feature
    zf, idf, zn = O.associate(np.zeros(3), np.zeros((3, 3)), z, R, 4.0, 25.0)
    assert zf.shape == (2, 0) and idf.shape == (1, 0) and zn.shape == (2, 1)
    # two equal-nd landmarks in gate -> lowest index (strict <)
    zf, idf, zn = O.associate(x3, P3, z, R, 4.0, 25.0, pair_fn=_table_fn([(1.0, 2.0), (1.0, 2.0), (9.0, 1.0)]))
    assert idf.tolist() == [[1]] and zn.shape[1] == 0
    # best nd wins among in-gate ones, not best nis
    zf, idf, zn = O.associate(x3, P3, z, R, 4.0, 25.0, pair_fn=_table_fn([(1.0, 5.0), (3.9, 0.5), (4.0, -9.0)]))
    assert idf.tolist() == [[2]]
    # gate1 <= min nis <= gate2 -> dropped
    zf, idf, zn = O.associate(x3, P3, z, R, 4.0, 25.0, pair_fn=_table_fn([(4.0, 0.0), (25.0, 0.0), (30.0, 0.0)]))
    assert zf.shape[1] == 0 and zn.shape[1] == 0
    # min nis > gate2 -> new
    zf, idf, zn = O.associate(x3, P3, z, R, 4.0, 25.0, pair_fn=_table_fn([(25.1, 0.0), (26.0, 0.0), (30.0, 0.0)]))
    assert zf.shape[1] == 0 and zn.shape[1] == 1
    # the same cases through the order-independent form used by the kernels
    for table, want in [([(1.0, 2.0), (1.0, 2.0), (9.0, 1.0)], 1),
                        ([(1.0, 5.0), (3.9, 0.5), (4.0, -9.0)], 2),
                        ([(4.0, 0.0), (25.0, 0.0), (30.0, 0.0)], 0),
                        ([(25.1, 0.0), (26.0, 0.0), (30.0, 0.0)], -1)]:
        nis = np.array([[t[0] for t in table]])
        nd = np.array([[t[1] for t in table]])
        assert O.assoc_vector(nis, nd, 4.0, 25.0).tolist() == [want]
    assert O.assoc_vector(np.zeros((2, 0)), np.zeros((2, 0)), 4.0, 25.0).tolist() == [-1, -1]


def test_kat6_mpi_to_pi_single_wrap():
    # src/common.jl:102-110
    assert O.mpi_to_pi(3.5 * math.pi) == pytest.approx(1.5 * math.pi, rel=1e-15)
    assert O.mpi_to_pi(-3.5 * math.pi) == pytest.approx(-1.5 * math.pi, rel=1e-15)
    assert O.mpi_to_pi(math.pi) == math.pi
    assert O.mpi_to_pi(-math.pi) == -math.pi
    assert O.mpi_to_pi(0.3) == 0.3


def test_kat7_update_cadence():
    # sim/ekfslam-sim.jl:75-76,102-105: dtsum > 8*dt first holds after 9 additions
    dt = 0.025
    dt_obs = 8 * dt
    dtsum, fired = 0.0, []
    for step in range(1, 28):
        dtsum += dt
        if dtsum > dt_obs:
            dtsum = 0.0
            fired.append(step)
    assert fired == [9, 18, 27]


def _random_state(rng, N, spread=60.0):
    n = 3 + 2 * N
    x = np.concatenate([[50.0, 50.0, 0.4], rng.uniform(50 - spread / 2, 50 + spread / 2, 2 * N)])
    A = rng.normal(0, 0.3, (n, n))
    P = A @ A.T / n + 0.01 * np.eye(n)
    return x, (P + P.T) / 2


@pytest.mark.parametrize("N", [1, 2, 7, 35])
def test_dense_vs_sparse_pairs_and_associate(N):
    rng = np.random.default_rng(100 + N)
    x, P = _random_state(rng, N)
    zs = []
    for j in rng.choice(np.arange(1, N + 1), size=min(N, 5), replace=False):
        zp, _ = O.predict_observation(x, j)
        zs.append(zp + rng.normal(0, [0.1, math.pi / 180]))
    zs.append(np.array([200.0, 0.3]))           # far away -> new feature
    z = np.array(zs).T
    nis_t, nd_t = O.association_table_sparse(x, P, z, R)
    for i in range(z.shape[1]):
        for j in range(1, N + 1):
            nis, nd = O.compute_association(x, P, z[:, i], R, j)
            assert nis_t[i, j - 1] == pytest.approx(nis, rel=1e-10, abs=1e-12)
            assert nd_t[i, j - 1] == pytest.approx(nd, rel=1e-10, abs=1e-10)
    zf, idf, zn = O.associate(x, P, z, R, 4.0, 25.0)
    zf2, idf2, zn2 = O.associate_sparse(x, P, z, R, 4.0, 25.0)
    assert np.array_equal(idf, idf2) and np.array_equal(zf, zf2) and np.array_equal(zn, zn2)
    assert idf.shape[0] == 1 and zf.shape[0] == 2 and zn.shape[0] == 2


@pytest.mark.parametrize("N,m", [(1, 1), (2, 2), (35, 6), (60, 0)])
def test_dense_vs_sparse_update(N, m):
    rng = np.random.default_rng(7 * N + m)
    x, P = _random_state(rng, N)
    idf = rng.choice(np.arange(1, N + 1), size=m, replace=False) if m else np.zeros(0, int)
    z = np.zeros((2, m))
    for i, j in enumerate(idf):
        zp, _ = O.predict_observation(x, j)
        z[:, i] = zp + rng.normal(0, [0.1, math.pi / 180])
    xd, Pd = O.update(x, P, z, R, idf.reshape(1, -1))
    xs, Ps = O.update_sparse(x, P, z, R, idf)
    assert np.allclose(xd, xs, rtol=1e-12, atol=1e-12)
    assert np.allclose(Pd, Ps, rtol=1e-11, atol=1e-14)
    xj, Pj = O.update_joseph_sparse(x, P, z, R, idf)
    assert np.allclose(xd, xj, rtol=1e-10, atol=1e-11)
    assert np.allclose(Pd, Pj, rtol=1e-9, atol=1e-12)
    if m:
        assert np.all(np.diag(Pd) <= np.diag(P) + 1e-15)           # information only shrinks variance
        assert np.allclose(Pd, Pd.T, atol=1e-15)


def test_update_duplicate_landmark_rows_stack():
    # two observations may pick the same landmark; update stacks both rows
    rng = np.random.default_rng(5)
    x, P = _random_state(rng, 4)
    zp, _ = O.predict_observation(x, 2)
    z = np.stack([zp + [0.05, 0.001], zp - [0.03, 0.002]], axis=1)
    xd, Pd = O.update(x, P, z, R, np.array([[2, 2]]))
    xs, Ps = O.update_sparse(x, P, z, R, [2, 2])
    assert np.allclose(xd, xs, rtol=1e-12) and np.allclose(Pd, Ps, rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("N,nn", [(0, 1), (0, 3), (5, 2), (35, 4)])
def test_dense_vs_sparse_add_features(N, nn):
    rng = np.random.default_rng(31 * N + nn)
    x, P = _random_state(rng, N)
    z = np.vstack([rng.uniform(5, 30, nn), rng.uniform(-1.5, 1.5, nn)])
    xd, Pd = O.add_features(x, P, z, R)
    xs, Ps = O.add_features_sparse(x, P, z, R)
    assert xd.shape == (3 + 2 * (N + nn),) and Pd.shape == (len(xd), len(xd))
    assert np.allclose(xd, xs, rtol=0, atol=1e-13)
    assert np.allclose(Pd, Ps, rtol=1e-13, atol=1e-16)


@pytest.mark.parametrize("N", [0, 3, 35])
def test_dense_vs_sparse_predict(N):
    rng = np.random.default_rng(N + 11)
    x, P = _random_state(rng, N)
    x1, P1 = O.predict(x.copy(), P.copy(), 7.6, 0.13, 4.0, Q, 0.025)
    x2, P2 = O.predict_sparse(x.copy(), P.copy(), 7.6, 0.13, 4.0, Q, 0.025)
    assert np.allclose(x1, x2, rtol=0, atol=1e-15)
    assert np.allclose(P1, P2, rtol=1e-14, atol=1e-18)
    assert np.array_equal(P1[3:, 3:], P[3:, 3:])                  # map block untouched (ekf.jl:32-36)


def test_predict_wraps_heading_once():
    x = np.array([0.0, 0.0, math.pi - 1e-3])
    P = np.zeros((3, 3))
    O.predict(x, P, 8.0, 0.5, 4.0, Q, 0.025)
    assert -math.pi <= x[2] < -math.pi + 0.03


# ---- KAT-8 .. KAT-10: closed forms worked on paper in tests/kat_vectors.py ---------------------------------
@pytest.mark.parametrize("kat", [KV.kat8, KV.kat9])
def test_kat8_kat9_update_closed_forms(kat):
    # src/ekf.jl:46-77: one observation (KAT-8); two stacked observations of the same landmark (KAT-9)
    x, P, z, idf, xp, Pp = kat()
    for fn in (O.update, O.update_sparse, O.update_joseph_sparse):
        xn, Pn = fn(x.copy(), P.copy(), z, KV.R, idf if fn is O.update else idf.reshape(-1))
        assert np.allclose(xn, xp, rtol=1e-13, atol=1e-15), fn.__name__
        assert np.allclose(Pn, Pp, rtol=1e-12, atol=1e-15), fn.__name__
    # the two measurement rows decouple: entries between the state groups {0, 3} and {1, 2, 4} stay exactly zero
    _, Pn = O.update(x.copy(), P.copy(), z, KV.R, idf)
    assert np.all(Pp[np.ix_([0, 3], [1, 2, 4])] == 0) and np.allclose(Pn[np.ix_([0, 3], [1, 2, 4])], 0, atol=1e-16)
    # spot values of KAT-8 (p = (0.5, 0.4, 0.02, 1, 2)): s1 = 1.51, P+[0,0] = 0.5 - 0.25/1.51, P+[0,3] = +0.5/1.51
    if kat is KV.kat8:
        assert Pp[0, 0] == pytest.approx(0.5 - 0.25 / 1.51, rel=1e-14) and Pp[0, 3] == pytest.approx(0.5 / 1.51, rel=1e-14)
        assert xp[3] == pytest.approx(10.0 + 0.5 / 1.51, rel=1e-15)


def test_kat11_update_rotated_heading_off_axis_landmark_coupled_covariance():
    # src/ekf.jl:46-77 with src/common.jl:139-165 at phi = pi/6, (dx, dy) = (3, 4) and a P with cross terms, against
    # the information form of the same update (tests/kat_vectors.py): the Jacobian is the hand-written one
    x, P, z, idf, xp, Pp = KV.kat11()
    zhat, H = O.predict_observation(x, 1)
    assert np.allclose(zhat, [5.0, math.atan2(4.0, 3.0) - math.pi / 6], rtol=0, atol=1e-15)
    assert np.allclose(H, [[-0.6, -0.8, 0.0, 0.6, 0.8], [0.16, -0.12, -1.0, -0.16, 0.12]], rtol=0, atol=1e-15)
    assert np.all(np.linalg.eigvalsh(P) > 0)
    for fn in (O.update, O.update_sparse, O.update_joseph_sparse):
        xn, Pn = fn(x.copy(), P.copy(), z, KV.R, idf if fn is O.update else idf.reshape(-1))
        assert np.allclose(xn, xp, rtol=1e-11, atol=1e-13), fn.__name__
        assert np.allclose(Pn, Pp, rtol=1e-9, atol=1e-13), fn.__name__
    # the posterior is tighter than the prior in the measured directions, and really coupled
    assert np.all(np.diag(Pp) < np.diag(P)) and abs(Pp[2, 4]) > 1e-4 and abs(Pp[0, 4]) > 1e-4
    # the gate sees the same innovation: nis = v' S^-1 v with S = H P H' + R
    v = np.array([0.3, -0.015])
    S = H @ P @ H.T + KV.R
    nis, nd = O.compute_association(x, P, z[:, 0], KV.R, 1)
    assert nis == pytest.approx(float(v @ np.linalg.solve(S, v)), rel=1e-12)
    assert nd == pytest.approx(nis + math.log(np.linalg.det(S)), rel=1e-12)


def test_kat10_add_features_with_vehicle_covariance_and_existing_landmark():
    # src/ekf.jl:84-122 incl. the cross block with the existing map (rnm, :115-118)
    x, P, zn, xp, Pp = KV.kat10()
    for fn in (O.add_features, O.add_features_sparse):
        xn, Pn = fn(x.copy(), P.copy(), zn, KV.R)
        assert np.allclose(xn, xp, rtol=0, atol=1e-14), fn.__name__
        assert np.allclose(Pn, Pp, rtol=1e-13, atol=1e-16), fn.__name__
    assert Pp[5, 3] == pytest.approx(0.03 - 2 * 0.005) and Pp[5, 5] == pytest.approx(0.30 - 0.08 + 0.04 + 4 * KV.R[1, 1])


def test_kat12_predict_with_heading_steering_and_coupled_covariance():
    # src/ekf.jl:8-43 at phi = pi/3, g = pi/6 (s = 1, c = 0), v dt = 1, with KAT-10's coupled P and a landmark
    x, P, (v, g, w, Q, dt), xp, Pp = KV.kat12()
    for fn in (O.predict, O.predict_sparse):
        xn, Pn = fn(x.copy(), P.copy(), v, g, w, Q, dt)
        assert np.allclose(xn, xp, rtol=0, atol=1e-15), fn.__name__
        assert np.allclose(Pn, Pp, rtol=1e-13, atol=1e-17), fn.__name__
        assert np.array_equal(Pn[3:, 3:], P[3:, 3:])
    # spot values: P+[0,0] = a - 2e + g0 + q2 = 0.30 - 0.04 + 0.01 + (3 deg)^2, P+[0,3] = h1 - j1 = 0.025
    assert Pp[0, 0] == pytest.approx(0.27 + (3 * math.pi / 180) ** 2, rel=1e-15) and Pp[0, 3] == pytest.approx(0.025, rel=1e-15)
    assert Pp[1, 1] == pytest.approx(0.20 + 0.25 / 16, rel=1e-15)


def test_low_rank_covariance_view_equals_the_dense_matrix():
    """LowRankCov (test infrastructure for the N = 50k configuration: P = A A' + d I is never materialised) must index
    exactly like the dense matrix in every pattern the sparse oracle uses, and update_joseph_factors must reproduce
    update_joseph_sparse block by block."""
    rng = np.random.default_rng(3)
    N = 40
    n = 3 + 2 * N
    A = rng.normal(0, 0.2, (n, 5))
    Pd = A @ A.T + 0.01 * np.eye(n)
    Pv = O.LowRankCov(A, 0.01)
    f = np.array([3, 9, 21])
    for got, want in ((Pv[0:3, 0:3], Pd[0:3, 0:3]), (Pv[0:3, f], Pd[0:3, f]), (Pv[f, 0:3], Pd[f, 0:3]), (Pv[f, f + 1], Pd[f, f + 1]),
                      (Pv[:, 0:3], Pd[:, 0:3]), (Pv[:, 7:9], Pd[:, 7:9]), (Pv[5:9, 60:70], Pd[5:9, 60:70])):
        assert got.shape == want.shape and np.allclose(got, want, rtol=1e-14, atol=1e-17)
    x = np.concatenate([[50.0, 50.0, 0.3], rng.uniform(10, 90, 2 * N)])
    ids = np.array([4, 17, 30])
    z = np.zeros((2, 3))
    for i, j in enumerate(ids):
        zp, _ = O.predict_observation(x, j)
        z[:, i] = zp + rng.normal(0, [0.1, math.pi / 180])
    nis_d, nd_d = O.association_table_sparse(x, Pd, z, R)
    nis_v, nd_v = O.association_table_sparse(x, Pv, z, R)
    assert np.allclose(nis_d, nis_v, rtol=1e-11) and np.allclose(nd_d, nd_v, rtol=1e-11, atol=1e-12)
    xj, Pj = O.update_joseph_sparse(x, Pd, z, R, ids)
    xf, K, T = O.update_joseph_factors(x, Pv, z, R, ids)
    assert np.allclose(xj, xf, rtol=1e-13)
    rows, cols = slice(10, 30), slice(0, 83)
    assert np.allclose(O.joseph_block(Pv, K, T, rows, cols), Pj[rows, cols], rtol=1e-11, atol=1e-15)
