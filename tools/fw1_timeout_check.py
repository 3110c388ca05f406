#!/usr/bin/env python3
"""The panel waves' bounded wait (factor_w1_kernel), provoked: experiments build, SLAMHIP_FW1=32 makes the factorising workgroup
publish nothing.  Expected: the update returns after about 2 s with SLAM_E_HIP ("did not publish"), the covariance is unchanged,
and with the switch off again the next update on the same handle works and matches a handle that never saw the fault.
    SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so python tools/fw1_timeout_check.py"""
import os, sys, time, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
from oracle import ekf_ref as O        # (test infrastructure: the observation model, to make plausible observations)
assert os.environ.get("SLAMHIP_LIBRARY"), "experiments build only"
R = np.diag([0.1 ** 2, (math.pi / 180) ** 2])
N, m = 3000, 40
rng = np.random.default_rng(5)
n = 3 + 2 * N
x = np.concatenate([[50.0, 50.0, 0.3], rng.uniform(-1450, 1550, 2 * N)])
A = rng.normal(0, 0.2, (n, 6)).astype(np.float32)
P = (A @ A.T).astype(np.float64) + 0.01 * np.eye(n)
a, b = (pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N) for _ in range(2))
ids = rng.permutation(N)[:m] + 1
z = np.zeros((2, m))
for i, j in enumerate(ids):
    zp, _ = O.predict_observation(x, j)
    z[:, i] = zp + rng.normal(0, [0.1, math.pi / 180])
before = a.download()
os.environ["SLAMHIP_FW1"] = "32"
t0 = time.time()
try:
    a.update(z, R, ids)
    print("NO ERROR: the fault was not provoked")
    sys.exit(1)
except pkg.SlamHipError as e:
    print(f"update failed after {time.time() - t0:.2f} s with code {e.code}: {e}")
    assert e.code == pkg._lib.SLAM_E_HIP
os.environ["SLAMHIP_FW1"] = "0"
after = a.download()
print("covariance unchanged:", np.array_equal(before[1], after[1]), "| mean unchanged:", np.array_equal(before[0], after[0]))
a.update(z, R, ids)
b.update(z, R, ids)
ga, gb = a.download(), b.download()
print("next update on the same handle equals a clean handle's:", np.array_equal(ga[0], gb[0]) and np.array_equal(ga[1], gb[1]))
