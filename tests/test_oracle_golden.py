"""CPU: the sparse oracle (what the kernels compute) against the committed golden
fixtures produced by the LITERAL oracle (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import pytest

from oracle import ekf_ref as O

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])


@pytest.fixture(scope="module")
def single(golden_dir):
    return np.load(os.path.join(golden_dir, "single_calls.npz"))


@pytest.fixture(scope="module")
def config1(golden_dir):
    return np.load(os.path.join(golden_dir, "config1.npz"))


@pytest.mark.parametrize("N", [0, 1, 2, 35, 100])
def test_single_calls_sparse_vs_golden(single, N):
    t = f"N{N}"
    x, P = single[f"{t}_x"], single[f"{t}_P"]
    v, g, w, dt = single[f"{t}_predict_vg"]
    xp, Pp = O.predict_sparse(x.copy(), P.copy(), v, g, w, Q, dt)
    assert np.allclose(xp, single[f"{t}_predict_x"], rtol=0, atol=1e-14)
    assert np.allclose(Pp, single[f"{t}_predict_P"], rtol=1e-13, atol=1e-18)
    z = single[f"{t}_z"]
    nis, nd = O.association_table_sparse(x, P, z, R)
    assert np.allclose(nis, single[f"{t}_nis"], rtol=1e-12) and np.allclose(nd, single[f"{t}_nd"], rtol=1e-12)
    assoc = O.assoc_vector(nis, nd, 4.0, 25.0)
    assert np.array_equal(assoc, single[f"{t}_assoc"])
    if N:
        assert (assoc > 0).sum() >= min(N, 6)
    if N in (1, 2):
        assert (assoc == 0).sum() == 1            # the nis = 10 observation falls between the gates
    assert (assoc == -1).sum() == 2
    zf, idf, zn = O.split_assoc(z, assoc)
    xu, Pu = O.update_sparse(x, P, zf, R, idf)
    assert np.allclose(xu, single[f"{t}_update_x"], rtol=1e-12, atol=1e-12)
    assert np.allclose(Pu, single[f"{t}_update_P"], rtol=1e-10, atol=1e-14)
    xa, Pa = O.add_features_sparse(xu, Pu, zn, R)
    # new features at 350-400 m amplify the 1e-13 heading difference of the two update forms
    assert np.allclose(xa, single[f"{t}_augment_x"], rtol=1e-10, atol=1e-10)
    assert np.allclose(Pa, single[f"{t}_augment_P"], rtol=1e-9, atol=1e-12)


def test_config1_fixture_shape_and_cadence(config1):
    steps = config1["obs_steps"]
    assert np.all(np.diff(steps) == 9) and steps[0] == 8          # every 9th predict (0-based step 8)
    assert config1["landmarks"].shape == (2, 35)
    lm = config1["landmarks"]
    assert np.all(lm == np.round(lm)) and lm.min() >= 5 and lm.max() <= 95     # make_landmarks pool
    assert config1["waypoints"].shape == (2, 19)
    assert config1["z_offsets"][-1] == config1["z"].shape[1] == config1["assoc"].shape[0]
    err = np.linalg.norm(config1["true_track"][:, :2] - config1["slam_track"][:, :2], axis=1)
    assert err.max() < 2.0                                        # the only system-level signal the reference offers
    assert (len(config1["final_x"]) - 3) // 2 == 35


def test_config1_replay_sparse_oracle(config1):
    """Replay the recorded call sequence through the sparse oracle: same decisions, same states."""
    c = config1
    f = O.OracleEKF(c["slam_track"][0] * 0 + np.r_[c["waypoints"][:, 0],
                                                  math.atan2(c["waypoints"][1, 1] - c["waypoints"][1, 0],
                                                             c["waypoints"][0, 1] - c["waypoints"][0, 0])],
                    np.zeros((3, 3)), sparse=True)
    zoff, xoff = c["z_offsets"], c["x_offsets"]
    obs_steps = c["obs_steps"]
    oi = 0
    ck = set(c["ckpt_ids"].tolist())
    for step, (v, g) in enumerate(c["controls"]):
        f.predict(v, g, 4.0, Q, 0.025)
        if oi < len(obs_steps) and obs_steps[oi] == step:
            z = c["z"][:, zoff[oi]:zoff[oi + 1]]
            nis, nd = O.association_table_sparse(f.x, f.cov, z, R)
            a = O.assoc_vector(nis, nd, 4.0, 25.0)
            assert np.array_equal(a, c["assoc"][zoff[oi]:zoff[oi + 1]])
            zf, idf, zn = O.split_assoc(z, a)
            f.update(zf, R, idf)
            f.add_features(zn, R)
            assert np.allclose(f.x, c["x_after"][xoff[oi]:xoff[oi + 1]], rtol=1e-9, atol=1e-9)
            if oi in ck:
                assert np.allclose(f.cov, c[f"ckpt_P_{oi}"], rtol=1e-7, atol=1e-12)
            oi += 1
        assert np.allclose(f.x[0:3], c["slam_track"][step], rtol=1e-9, atol=1e-9)
    assert np.allclose(f.x, c["final_x"], rtol=1e-9, atol=1e-9)
    assert np.allclose(f.cov, c["final_P"], rtol=1e-7, atol=1e-12)
