import math, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import ekf_ref as O
pkg = load_package()
R = np.array([[0.01, 0.0], [0.0, (math.pi / 180) ** 2]])
for N in (40, 100):
    rng = np.random.default_rng(7)
    n = 3 + 2 * N
    x = np.concatenate([[50.0, 50.0, 0.3], rng.uniform(10, 90, 2 * N)])
    A = rng.normal(0, 0.2, (n, 8)); P = A @ A.T + 0.01 * np.eye(n)
    st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
    xo, Po = st.download(); xo = xo.astype(np.float64); Po = Po.astype(np.float64)
    ids = np.array([3, 7, 11])
    z = np.zeros((2, 3))
    for i, j in enumerate(ids):
        zp, _ = O.predict_observation(xo, j); z[:, i] = zp + [0.05, 0.002]
    st.update(z, R, ids)
    xn, Pn = O.update_sparse(xo, Po, z, R, ids)
    xg, Pg = st.download()
    E = np.abs(Pg - Pn)
    print("N", N, "n", n, "max err", E.max(), "x err", np.abs(xg - xn).max())
    nb = (n + 31) // 32
    for bi in range(nb):
        print(" ".join(f"{E[32*bi:32*bi+32, 32*bj:32*bj+32].max():8.1e}" for bj in range(nb)))
    st.close()
