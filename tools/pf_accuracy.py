#!/usr/bin/env python3
"""How far is the fp32 particle sweep from the float64 oracle?  The scenario of
tests/test_gpu_pf.py::test_predict_update_weights_against_oracle, longer (40 steps, 20 000 particles); prints the
largest error of each quantity relative to its scale.  Run once per build (SLAMHIP_LIBRARY) to compare the hardware
transcendentals of the fp32 instantiation (PF_FAST_MATH=1, the default) with the exact library forms."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import pf_ref as F
pkg = load_package()
R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
n, nl, seed = 20000, 10, 77
lm = np.random.default_rng(1).uniform(-40, 40, (nl, 2))
for dtype in ("f32", "f64"):
    sh = pkg.PFShard(n, nl, seed, dtype=dtype)
    orc = F.OraclePF(n, nl, seed)
    for f in (sh, orc):
        f.set_pose([1.0, -2.0, 0.4]); f.init_landmarks(lm[:7], 0.01, 0.1)
    rng = np.random.default_rng(2)
    pose = np.array([1.0, -2.0, 0.4])
    worst = {"pose": 0.0, "lm mean": 0.0, "lm cov": 0.0, "logw": 0.0}
    def rel(a, b):
        b = np.asarray(b, dtype=np.float64)
        return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / max(float(np.max(np.abs(b))), 1e-30))
    for t in range(40):
        for f in (sh, orc):
            f.predict(6.0, 0.05 * (t % 8), 4.0, Q, 0.1)
        pose = np.array([pose[0] + 0.6 * math.cos(0.05 * (t % 8) + pose[2]), pose[1] + 0.6 * math.sin(0.05 * (t % 8) + pose[2]),
                         pose[2] + 0.6 * math.sin(0.05 * (t % 8)) / 4.0])
        ids = np.array([(2 * t) % nl + 1, (2 * t + 1) % nl + 1, 8 + t % 3, (2 * t) % nl + 1])
        dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, len(ids)))
        for f in (sh, orc):
            f.update_known(z, ids, R)
        p, lw, l = sh.download()
        worst["pose"] = max(worst["pose"], rel(p, orc.pose))
        worst["lm mean"] = max(worst["lm mean"], rel(l[:, 0:2], orc.lm[:, 0:2]))
        worst["lm cov"] = max(worst["lm cov"], rel(l[:, 2:5], orc.lm[:, 2:5]))
        worst["logw"] = max(worst["logw"], float(np.max(np.abs(lw.astype(np.float64) - orc.logw)) / max(1.0, float(np.max(np.abs(orc.logw))))))
    print(dtype, os.environ.get("SLAMHIP_LIBRARY", "default"), {k: f"{v:.2e}" for k, v in worst.items()})
    sh.close()
