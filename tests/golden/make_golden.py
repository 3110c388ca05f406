"""Generates the committed golden fixtures in tests/golden/ from the fp64 LITERAL oracle
(oracle/ekf_ref.py).  Run from the repo root:  python tests/golden/make_golden.py

The reference itself cannot run here (Julia 0.6 source, no julia binary) and ships no
vectors, so these are oracle outputs, not reference outputs ("parity unpinned", see
oracle/ekf_ref.py).  Fixtures are data only: inputs (including every noise draw the
filter saw) and expected outputs.

  single_calls.npz   one predict / associate / update / add_features call each on seeded
                     random states with N in {0, 1, 2, 35, 100}
  config1.npz        BASELINE.json config 1: sim/course1.txt waypoints (tests/golden/
                     course1.txt), 35 seeded make_landmarks landmarks, 2 laps, the
                     reference's constants; the full call sequence with its inputs, the
                     association decisions, both tracks and covariance checkpoints
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ekf_ref as O  # noqa: E402


def load_sim():
    """slam.jl_amd/sim.py is pure host code; load it without importing the package
    (whose __init__ needs the built HIP library)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("slam_sim_only", os.path.join(ROOT, "slam.jl_amd", "sim.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])


def random_state(rng, N):
    n = 3 + 2 * N
    x = np.concatenate([[50.0, 50.0, rng.uniform(-3, 3)], rng.uniform(5, 95, 2 * N)])
    A = rng.normal(0, 0.2, (n, 6))
    P = A @ A.T + 0.01 * np.eye(n)
    return x, (P + P.T) / 2


def single_calls():
    out = {}
    for N in (0, 1, 2, 35, 100):
        rng = np.random.default_rng(4200 + N)
        x, P = random_state(rng, N)
        tag = f"N{N}"
        out[f"{tag}_x"] = x
        out[f"{tag}_P"] = P
        # predict
        v, g = 7.7, -0.21
        xp, Pp = O.predict(x.copy(), P.copy(), v, g, 4.0, Q, 0.025)
        out[f"{tag}_predict_vg"] = np.array([v, g, 4.0, 0.025])
        out[f"{tag}_predict_x"] = xp
        out[f"{tag}_predict_P"] = Pp
        # observations: a few true landmarks (noisy), one duplicate, one far away, one in the dead band
        zs = []
        if N:
            pick = rng.choice(np.arange(1, N + 1), size=min(N, 6), replace=False)
            for j in pick:
                zp, _ = O.predict_observation(x, j)
                zs.append(zp + rng.normal(0, [0.1, math.pi / 180]))
            zp, _ = O.predict_observation(x, pick[0])
            zs.append(zp + np.array([0.05, -0.002]))                 # second hit on the same landmark
            _, Hd = O.predict_observation(x, pick[0])
            Sinv = np.linalg.inv(Hd @ P @ Hd.T + R)
            zs.append(zp + np.array([math.sqrt(10.0 / Sinv[0, 0]), 0.0]))   # nis = 10: gate1 < nis < gate2, dropped
        zs.append(np.array([400.0, 0.4]))                            # new feature
        zs.append(np.array([350.0, -2.0]))                           # new feature
        z = np.array(zs).T
        zf, idf, zn = O.associate(x, P, z, R, 4.0, 25.0)
        nis, nd = O.association_table_sparse(x, P, z, R)
        assoc = O.assoc_vector(nis, nd, 4.0, 25.0)
        zf2, idf2, zn2 = O.split_assoc(z, assoc)
        assert np.array_equal(idf, idf2) and np.array_equal(zf, zf2) and np.array_equal(zn, zn2)
        out[f"{tag}_z"] = z
        out[f"{tag}_assoc"] = assoc
        out[f"{tag}_nis"] = nis
        out[f"{tag}_nd"] = nd
        xu, Pu = O.update(x, P, zf, R, idf)
        out[f"{tag}_update_x"] = xu
        out[f"{tag}_update_P"] = Pu
        xa, Pa = O.add_features(xu, Pu, zn, R)
        out[f"{tag}_augment_x"] = xa
        out[f"{tag}_augment_P"] = Pa
    np.savez_compressed(os.path.join(HERE, "single_calls.npz"), **out)
    print("single_calls.npz:", len(out), "arrays")


def config1():
    S = load_sim()
    wp = S.get_waypoints(os.path.join(HERE, "course1.txt"))
    lm = S.make_landmarks(35, [0.0, 100.0, 0.0, 100.0], 0.05, np.random.default_rng(20240601))
    filt = O.OracleEKF(S.initial_pose(wp), np.zeros((3, 3)), sparse=False)
    cov_ckpt = {}
    x_after = []

    def monitor(vehicle, f, nsteps):
        pass

    # wrap update/add_features to record the state after each observation step
    log = S.sim(filt, wp, lm, seed=20240602, nlaps=2, monitor=monitor)
    # replay to collect per-observation-step states (cheap at n <= 73)
    f2 = O.OracleEKF(S.initial_pose(wp), np.zeros((3, 3)), sparse=False)
    obs_i = 0
    ckpt_every = 60
    for step, (v, g) in enumerate(log.controls):
        f2.predict(v, g, 4.0, Q, S.DT)
        if obs_i < len(log.obs_steps) and log.obs_steps[obs_i] == step:
            z = log.observations[obs_i]
            zf, idf, zn = f2.associate(z, R, S.GATE1, S.GATE2)
            assert idf.reshape(-1).tolist() == log.assoc[obs_i][0] and zn.shape[1] == log.assoc[obs_i][1]
            f2.update(zf, R, idf)
            f2.add_features(zn, R)
            x_after.append(f2.x.copy())
            if obs_i % ckpt_every == 0 or obs_i == len(log.obs_steps) - 1:
                cov_ckpt[obs_i] = f2.cov.copy()
            obs_i += 1
    assert np.allclose(f2.x, filt.x, rtol=0, atol=0)

    nobs = len(log.observations)
    zoff = np.zeros(nobs + 1, dtype=np.int64)
    for i, z in enumerate(log.observations):
        zoff[i + 1] = zoff[i] + z.shape[1]
    zcat = np.concatenate([z for z in log.observations], axis=1) if nobs else np.zeros((2, 0))
    assoc_cat = np.zeros(zoff[-1], dtype=np.int32)
    # association vector per observation step, recomputed in vector form for the C ABI comparison
    f3 = O.OracleEKF(S.initial_pose(wp), np.zeros((3, 3)), sparse=True)
    obs_i = 0
    for step, (v, g) in enumerate(log.controls):
        f3.predict(v, g, 4.0, Q, S.DT)
        if obs_i < nobs and log.obs_steps[obs_i] == step:
            z = log.observations[obs_i]
            nis, nd = O.association_table_sparse(f3.x, f3.cov, z, R)
            a = O.assoc_vector(nis, nd, S.GATE1, S.GATE2)
            assoc_cat[zoff[obs_i]:zoff[obs_i + 1]] = a
            zf, idf, zn = O.split_assoc(z, a)
            assert idf.reshape(-1).tolist() == log.assoc[obs_i][0]
            f3.update(zf, R, idf)
            f3.add_features(zn, R)
            obs_i += 1
    xoff = np.zeros(nobs + 1, dtype=np.int64)
    for i, xa in enumerate(x_after):
        xoff[i + 1] = xoff[i] + len(xa)
    out = dict(
        waypoints=wp, landmarks=lm, seed=np.array([20240601, 20240602]),
        controls=np.array(log.controls), obs_steps=np.array(log.obs_steps, dtype=np.int64),
        z=zcat, z_offsets=zoff, assoc=assoc_cat,
        x_after=np.concatenate(x_after), x_offsets=xoff,
        true_track=np.array(log.true_track), slam_track=np.array(log.slam_track),
        final_x=filt.x, final_P=filt.cov,
        ckpt_ids=np.array(sorted(cov_ckpt), dtype=np.int64),
    )
    for kx in sorted(cov_ckpt):
        out[f"ckpt_P_{kx}"] = cov_ckpt[kx]
    np.savez_compressed(os.path.join(HERE, "config1.npz"), **out)
    err = np.linalg.norm(np.array(log.true_track)[:, :2] - np.array(log.slam_track)[:, :2], axis=1)
    print(f"config1.npz: {len(log.controls)} predict steps, {nobs} observation steps, "
          f"{(len(filt.x) - 3) // 2} landmarks mapped, track error mean {err.mean():.3f} m max {err.max():.3f} m")


if __name__ == "__main__":
    single_calls()
    config1()
