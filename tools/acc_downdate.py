#!/usr/bin/env python3
"""Accuracy of the fp32 covariance down-date variants against the fp64 oracle (oracle/ekf_ref.py::update) on the
bench workload at N = 2000, nz = 64: native fp32 MFMA (SLAMHIP_X=8), split-bf16 with six terms (default) and with
all nine (SLAMHIP_X=64).  Prints max and rms error of P after one and after several updates, relative to max|P|.
TEST/MEASUREMENT TOOL: the oracle is the checker here, as in tests/."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from __graft_entry__ import load_package                      # noqa: E402
from oracle import ekf_ref as O                               # noqa: E402

N, NZ, STEPS = 1500, 64, 6
x0, P0, zs = bench.make_workload(N, NZ, STEPS, bench.SEED)
pkg = load_package()

# fp64 oracle trajectory from the SAME fp32 inputs; associations are taken from the oracle so that every variant
# assimilates the same observations
xo, Po = x0.astype(np.float64), P0.astype(np.float64)
traj, assoc_list = [], []
for z in zs:
    zf, idf, zn = O.associate_sparse(xo, Po, z, bench.R, bench.GATE1, bench.GATE2)
    xo, Po = O.update(xo, Po, zf, bench.R, idf)
    traj.append((xo.copy(), Po.copy()))
    assoc_list.append((np.asarray(zf), np.asarray(idf).reshape(-1)))

for name, xf in (("native fp32 MFMA", "8"), ("split-bf16, 6 terms", "0"), ("split-bf16, 9 terms", "64")):
    os.environ["SLAMHIP_X"] = xf
    st = pkg.EKFSlamState(x0[:3], np.zeros((3, 3), np.float32), dtype="f32", max_landmarks=N)
    st.set_state(x0, P0)
    out = []
    for t, (zf, idf) in enumerate(assoc_list):
        st.update(zf, bench.R, idf)
        if t in (0, STEPS - 1):
            xg, Pg = st.x.numpy(), st.cov.numpy()
            xo, Po = traj[t]
            d = Pg.astype(np.float64) - Po
            out.append((t + 1, np.abs(d).max() / np.abs(Po).max(), math.sqrt((d * d).mean()) / np.abs(Po).max(),
                        np.abs(xg - xo).max()))
    for t, emax, erms, ex in out:
        print(f"{name:22s} after {t} update(s): max|dP|/max|P| {emax:.3e}  rms {erms:.3e}  max|dx| {ex:.3e}")
    st.close()
