#!/usr/bin/env python3
"""Soak of the down-date's LDS-DMA chunk pipeline (hand-counted waits): the same filter run with it and with the
register-staged pipeline (SLAMHIP_X=512 at create), several steps, many repetitions and sizes; the covariances must agree
bit for bit every time.  A chunk read before it had landed would show up as a difference that comes and goes.
    python tools/soak_dma.py [repetitions]"""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
from oracle import ekf_ref as O        # (test infrastructure: only the observation model, to make plausible observations)

R = np.diag([0.1 ** 2, (math.pi / 180) ** 2])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
total = 0
for rep in range(reps):
    for N, m in ((6000, 64), (6000, 40), (10000, 64), (3000, 48)):
        rng = np.random.default_rng(1000 * rep + N + m)
        n = 3 + 2 * N
        x = np.concatenate([[50.0, 50.0, rng.uniform(-3, 3)], rng.uniform(50 - 1500, 50 + 1500, 2 * N)])
        A = rng.normal(0, 0.2, (n, 6)).astype(np.float32)
        P = (A @ A.T).astype(np.float64) + 0.01 * np.eye(n)
        got = {}
        for name, flag in (("dma", None), ("staged", "512")):
            if flag is None:
                os.environ.pop("SLAMHIP_X", None)
            else:
                os.environ["SLAMHIP_X"] = flag
            st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
            r2 = np.random.default_rng(7 + rep)
            for step in range(3):
                xo = st.download("x").astype(np.float64)
                ids = r2.permutation(N)[:m] + 1
                z = np.zeros((2, m))
                for i, j in enumerate(ids):
                    zp, _ = O.predict_observation(xo, j)
                    z[:, i] = zp + r2.normal(0, [0.1, math.pi / 180])
                st.update(z, R, ids)
            got[name] = st.download("cov")
            st.close()
        os.environ.pop("SLAMHIP_X", None)
        same = np.array_equal(got["dma"], got["staged"])
        total += 1
        bad += 0 if same else 1
        if not same:
            d = np.abs(got["dma"].astype(np.float64) - got["staged"]).max()
            print(f"rep {rep} N {N} m {m}: DIFFERENT, max |d| {d:.3e}", flush=True)
    print(f"rep {rep}: {total - bad}/{total} identical so far", flush=True)
print(f"soak_dma: {total - bad}/{total} identical")
sys.exit(1 if bad else 0)
