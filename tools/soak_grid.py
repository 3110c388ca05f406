#!/usr/bin/env python3
"""Fuzz of the grid form of the gating against the sweep form (slam_ekf_set_gate_mode): random and degenerate maps
(clustered, collinear, coincident landmarks, coordinates of any sign and scale), poses inside and far outside the map,
tight and loose covariances, tight and loose gates, observations of real landmarks, near misses and nonsense; interleaved
with updates (means move), add_features (tail) and uploads.  Every association vector must be identical."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
ROUNDS = int(os.environ.get("SOAK_ROUNDS", "60"))
R0 = np.diag([0.1 ** 2, (math.pi / 180) ** 2])
t0 = time.time(); queries = 0; bad = 0

def make_map(N, kind, scale):
    if kind == "uniform":
        lm = rng.uniform(-scale, scale, (N, 2))
    elif kind == "clustered":
        c = rng.uniform(-scale, scale, (max(N // 40, 1), 2))
        lm = c[rng.integers(0, len(c), N)] + rng.normal(0, scale * 0.01, (N, 2))
    elif kind == "line":
        t = rng.uniform(-scale, scale, N); lm = np.stack([t, 0.3 * t + 5.0], axis=1)
    elif kind == "vertical":
        lm = np.stack([np.full(N, 7.0), rng.uniform(-scale, scale, N)], axis=1)
    elif kind == "coincident":
        lm = np.tile(rng.uniform(-scale, scale, (1, 2)), (N, 1)) + rng.normal(0, 1e-3, (N, 2))
    else:                                    # offset: far from the origin (cell arithmetic at large magnitudes)
        lm = rng.uniform(-scale, scale, (N, 2)) + np.array([3e5, -7e5])
    return lm

for rnd in range(ROUNDS):
    N = int(rng.choice([1, 2, 5, 37, 300, 1500, 4000]))
    kind = str(rng.choice(["uniform", "clustered", "line", "vertical", "coincident", "offset"]))
    scale = float(rng.choice([3.0, 60.0, 2000.0]))
    dtype = "f64" if kind == "offset" or rng.random() < 0.3 else "f32"
    lm = make_map(N, kind, scale)
    n = 3 + 2 * N
    centre = lm.mean(axis=0)
    pose = np.array([*(centre + rng.normal(0, scale * float(rng.choice([0.1, 1.0, 5.0])), 2)), rng.uniform(-3.1, 3.1)])
    x = np.concatenate([pose, lm.reshape(-1)])
    rank = int(rng.choice([1, 4]))
    A = rng.normal(0, float(rng.choice([1e-3, 0.05, 1.0])), (n, rank))
    P = A @ A.T + float(rng.choice([1e-4, 0.01, 0.5])) * np.eye(n)
    g1 = float(rng.choice([0.5, 4.0, 9.0])); g2 = g1 * float(rng.choice([1.0, 6.25, 50.0]))
    sts = {}
    for mode in ("grid", "sweep"):
        sts[mode] = pkg.EKFSlamState(x, P, dtype=dtype, max_landmarks=N + 200)
        sts[mode].set_gate_mode(mode)
    ops = []
    for it in range(8):
        xs = sts["sweep"].download("x").astype(np.float64)
        nz = int(rng.choice([1, 7, 40, 150]))
        ids = rng.integers(0, sts["sweep"].N, nz)
        dx = xs[3 + 2 * ids] - xs[0]; dy = xs[4 + 2 * ids] - xs[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - xs[2]])
        z += rng.normal(0, 1, z.shape) * np.array([[0.1], [math.pi / 180]]) * float(rng.choice([0.0, 1.0, 8.0]))
        k = rng.random(nz) < 0.2                                  # nonsense: any range, any bearing, unwrapped
        z[0, k] = rng.uniform(-5, 3 * scale + 50, k.sum()); z[1, k] = rng.uniform(-10, 10, k.sum())
        a = {m: sts[m].associate_vector(z, R0, g1, g2) for m in sts}
        queries += nz
        if not np.array_equal(a["grid"], a["sweep"]):
            bad += 1
            i = int(np.flatnonzero(a["grid"] != a["sweep"])[0])
            xg, xs2 = sts["grid"].download("x"), sts["sweep"].download("x")
            print("   states equal before the query:", np.array_equal(xg, xs2, equal_nan=True), "ops so far:", ops, flush=True)
            for m in sts:
                for j in {int(a["grid"][i]), int(a["sweep"][i])}:
                    if j > 0:
                        print(f"   filter {m}: landmark {j}: nis, nd = {sts[m].compute_association(z[:, i], R0, j)}  mean {sts[m].download('x')[1 + 2 * j: 3 + 2 * j]} pose {sts[m].download('x')[:3]}", flush=True)
            print(f"MISMATCH round {rnd} it {it}: N={sts['sweep'].N} {kind} scale {scale} {dtype} gates {g1}/{g2} obs {i} z={z[:, i]} grid {a['grid'][i]} sweep {a['sweep'][i]} info {sts['grid'].gate_info()}", flush=True)
            break
        op = rng.integers(0, 4)
        ops.append(int(op))
        if op == 0 and sts["sweep"].N >= 3:                       # an update with known correspondences: the means move
            m = min(int(rng.integers(1, 12)), sts["sweep"].N)
            uid = rng.choice(sts["sweep"].N, m, replace=False) + 1
            dxu = xs[1 + 2 * uid] - xs[0]; dyu = xs[2 + 2 * uid] - xs[1]
            zu = np.vstack([np.hypot(dxu, dyu), np.arctan2(dyu, dxu) - xs[2]]) + rng.normal(0, 1, (2, m)) * np.array([[0.3], [0.02]])
            try:
                for st in sts.values():
                    st.update(zu, R0, uid)
            except pkg.NotPositiveDefinite:
                pass
        elif op == 1:                                             # the fused step: update + add_features
            for st in sts.values():
                try:
                    st.observe(z[:, :min(nz, 60)], R0, g1, g2)
                except (pkg.NotPositiveDefinite, pkg.SlamHipError):
                    pass
        elif op == 2:
            v, gma = float(rng.uniform(0, 20)), float(rng.uniform(-0.5, 0.5))
            for st in sts.values():
                st.predict(v, gma, 4.0, np.diag([0.25, 0.003]), 0.1)
    xa, xb = sts["grid"].download("x"), sts["sweep"].download("x")
    if not np.array_equal(xa, xb, equal_nan=True):
        bad += 1
        print(f"STATE MISMATCH round {rnd}: {kind} N={N}", flush=True)
    for st in sts.values():
        st.close()
    if rnd % 10 == 9:
        print(f"[{time.time() - t0:6.1f} s] {rnd + 1} rounds, {queries} observations gated in both forms, {bad} mismatches", flush=True)
print(f"done: {ROUNDS} rounds, {queries} observations, {bad} mismatches")
sys.exit(1 if bad else 0)
