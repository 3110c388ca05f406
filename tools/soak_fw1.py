#!/usr/bin/env python3
"""Soak of the streamed front half (factor_w1_kernel: C read by the panel waves while the elimination still runs): the same
filter run with it and with round 4's two launches (SLAMHIP_X=128 at create), many steps with a changing number of matched
observations (1 ... 64: one to eight block columns, every padding shape), fp32 and fp64; mean and covariance must agree bit for bit
at every checkpoint.  A block column read before it was complete would show up as a difference that comes and goes.
    python tools/soak_fw1.py [steps]"""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
from oracle import ekf_ref as O        # (test infrastructure: only the observation model, to make plausible observations)

R = np.diag([0.1 ** 2, (math.pi / 180) ** 2])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
bad = total = 0
for N, dtype in ((2500, "f32"), (6000, "f32"), (1200, "f64"), (10000, "f32")):
    rng = np.random.default_rng(N)
    n = 3 + 2 * N
    x = np.concatenate([[50.0, 50.0, rng.uniform(-3, 3)], rng.uniform(50 - 1500, 50 + 1500, 2 * N)])
    A = rng.normal(0, 0.2, (n, 6)).astype(np.float32)
    P = (A @ A.T).astype(np.float64) + 0.01 * np.eye(n)
    sts = {}
    for name, flag in (("fused", None), ("two", "128")):
        if flag is None:
            os.environ.pop("SLAMHIP_X", None)
        else:
            os.environ["SLAMHIP_X"] = flag
        sts[name] = pkg.EKFSlamState(x, P, dtype=dtype, max_landmarks=N)
    os.environ.pop("SLAMHIP_X", None)
    del P, A
    r2 = np.random.default_rng(7 + N)
    for step in range(steps):
        m = int(r2.integers(1, 65))
        xo = sts["fused"].download("x").astype(np.float64)
        ids = r2.permutation(N)[:m] + 1
        z = np.zeros((2, m))
        for i, j in enumerate(ids):
            zp, _ = O.predict_observation(xo, j)
            z[:, i] = zp + r2.normal(0, [0.1, math.pi / 180])
        for st in sts.values():
            st.update(z, R, ids)
        if step % 10 == 9 or step == steps - 1:
            total += 1
            same = np.array_equal(sts["fused"].download("x"), sts["two"].download("x")) and np.array_equal(sts["fused"].diag(), sts["two"].diag())
            if step == steps - 1 or step % 40 == 39:
                same = same and np.array_equal(sts["fused"].download("cov"), sts["two"].download("cov"))
            bad += 0 if same else 1
            if not same:
                print(f"N={N} {dtype} step {step} m={m}: DIFFERENT", flush=True)
    print(f"N={N} {dtype}: {steps} steps done, mismatching checkpoints so far {bad} of {total}", flush=True)
    for st in sts.values():
        st.close()
print(f"soak_fw1: {total - bad} of {total} checkpoints bit-identical")
sys.exit(1 if bad else 0)
