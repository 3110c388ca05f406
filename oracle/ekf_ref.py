"""oracle/ekf_ref.py -- CPU restatement of SLAM.jl's EKF-SLAM filter core.

TEST INFRASTRUCTURE ONLY.  Nothing under ``slam.jl_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED (by reference-run outputs): the reference is Julia 0.5/0.6
source that no toolchain in this image can execute, and it ships no tests,
golden vectors or fixtures.  This restatement is pinned
only by the hand-derived known-answer tests KAT-1..KAT-7 in
``tests/test_oracle_kat.py`` (derived from the reference's formulas, not from
running it) and by the dense-vs-sparse self-consistency tests.

Two restatements live here, all float64 / NumPy:

* the *literal dense* functions (``predict``, ``update``, ``add_features``,
  ``associate``, ``compute_association``, ``predict_observation``,
  ``mpi_to_pi``) follow the reference op for op -- dense 2 x n Jacobians,
  dense ``H*P*H'`` -- and cite the reference lines they follow
  (paths relative to /root/reference);
* the ``*_sparse`` twins compute the same quantities from the 5 non-zero
  Jacobian columns only (what the HIP kernels compute) and are vectorised so
  that they finish in seconds at N = 10k.  Their agreement with the literal
  functions is itself a test.

Conventions (reference): x = [xv, yv, phi, l1x, l1y, ...]; landmark index
``idf`` is 1-based; landmark j occupies x[2j+1 : 2j+3] in 0-based NumPy terms
(``fpos = 3 + 2*idf - 1`` 1-based, src/common.jl:142-143); z is 2 x nz with
rows (range, bearing); R, Q are 2 x 2.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = [
    "mpi_to_pi", "predict_observation", "compute_association", "associate",
    "predict", "update", "add_features",
    "obs_blocks", "compute_association_sparse", "association_table_sparse",
    "associate_sparse", "assoc_vector", "split_assoc", "update_sparse",
    "update_joseph_sparse", "add_features_sparse", "predict_sparse",
]


# --------------------------------------------------------------------------
# literal restatement
# --------------------------------------------------------------------------

def mpi_to_pi(phi):
    """src/common.jl:102-110 -- ONE conditional wrap, not a modulo."""
    if phi > math.pi:
        return phi - 2 * math.pi
    if phi < -math.pi:
        return phi + 2 * math.pi
    return phi


def predict_observation(x, idf):
    """src/common.jl:139-165.  Returns (z (2,), H (2, n) dense)."""
    x = np.asarray(x, dtype=np.float64)
    fpos = 3 + int(idf) * 2 - 1          # 1-based position      (:142-143)
    f = fpos - 1                         # 0-based
    dx = x[f] - x[0]                     # :146
    dy = x[f + 1] - x[1]                 # :147
    d2 = dx * dx + dy * dy               # :148
    d = math.sqrt(d2)                    # :149
    z = np.array([d, math.atan2(dy, dx) - x[2]])   # :152 (bearing NOT wrapped)
    xd = dx / d                          # :155-158
    yd = dy / d
    xd2 = dx / d2
    yd2 = dy / d2
    H = np.zeros((2, len(x)))            # :160
    H[:, 0:3] = [[-xd, -yd, 0.0], [yd2, -xd2, -1.0]]      # :161
    H[:, f:f + 2] = [[xd, yd], [-yd2, xd2]]               # :162
    return z, H


def compute_association(x, P, z, R, idf):
    """src/data-association.jl:53-63.  Returns (nis, nd)."""
    zp, H = predict_observation(x, idf)
    v = np.asarray(z, dtype=np.float64) - zp
    v[1] = mpi_to_pi(v[1])
    S = H @ P @ H.T + R
    nis = float(np.dot(v, np.linalg.inv(S) @ v))
    nd = nis + math.log(np.linalg.det(S))
    return nis, nd


def associate(x, P, z, R, gate1, gate2, pair_fn=compute_association):
    """src/data-association.jl:1-51, sequential scan exactly as written.

    Returns (zf (2, nf), idf (1, nf) int, zn (2, nn)).
    """
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    zf = np.zeros((2, 0))
    zn = np.zeros((2, 0))
    idf = np.zeros((1, 0), dtype=np.int64)
    Nxv = 3
    Nf = int(round((len(x) - Nxv) / 2))              # :16
    for i in range(z.shape[1]):                      # :21
        jbest = 0
        nbest = math.inf
        outer = math.inf
        for j in range(1, Nf + 1):                   # :27
            nis, nd = pair_fn(x, P, z[:, i], R, j)
            ingate = 0
            if nis < gate1:                          # :30
                if nd < nbest:                       # :31
                    ingate = 1
            if ingate == 1:                          # :35
                nbest = nd
                jbest = j
            elif nis < outer:                        # :38
                outer = nis
        if jbest != 0:                               # :43
            zf = np.hstack([zf, z[:, i:i + 1]])
            idf = np.hstack([idf, [[jbest]]])
        elif outer > gate2:                          # :46
            zn = np.hstack([zn, z[:, i:i + 1]])
    return zf, idf, zn


def predict(x, P, v, g, w, Q, dt):
    """src/ekf.jl:8-43.  (v, g, w) = vehicle.measured_speed / measured_gamma /
    wheelbase (:14-16).  Mutates x, P in place like the reference and returns them.
    """
    phi = x[2]
    s = math.sin(g + phi)
    c = math.cos(g + phi)
    vts = v * dt * s
    vtc = v * dt * c
    Gv = np.array([[1.0, 0.0, -vts], [0.0, 1.0, vtc], [0.0, 0.0, 1.0]])       # :24-26
    Gu = np.array([[dt * c, -vts], [dt * s, vtc],
                   [dt * math.sin(g) / w, v * dt * math.cos(g) / w]])        # :27-29
    P[0:3, 0:3] = Gv @ P[0:3, 0:3] @ Gv.T + Gu @ Q @ Gu.T                    # :32
    if P.shape[0] > 3:
        P[0:3, 3:] = Gv @ P[0:3, 3:]                                         # :34
        P[3:, 0:3] = P[0:3, 3:].T                                            # :35
    x[0:3] = [x[0] + vtc, x[1] + vts, mpi_to_pi(phi + v * dt * math.sin(g) / w)]  # :39-41
    return x, P


def update(x, P, z, R, idf):
    """src/ekf.jl:46-77, dense H and dense products.  Returns NEW (x, P)."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    idf = np.asarray(idf).reshape(-1)
    lenz = z.shape[1]
    lenx = len(x)
    H = np.zeros((2 * lenz, lenx))
    v = np.zeros(2 * lenz)
    RR = np.zeros((2 * lenz, 2 * lenz))
    for i in range(lenz):                                   # :55-61
        zp, Hi = predict_observation(x, idf[i])
        H[2 * i:2 * i + 2, :] = Hi
        v[2 * i] = z[0, i] - zp[0]
        v[2 * i + 1] = mpi_to_pi(z[1, i] - zp[1])
        RR[2 * i:2 * i + 2, 2 * i:2 * i + 2] = R
    PHt = P @ H.T                                           # :67
    S = H @ PHt + RR                                        # :68
    S = (S + S.T) * 0.5                                     # :69
    if lenz:
        U = np.linalg.cholesky(S).T                         # chol(S): upper, S = U'U
        C = np.linalg.inv(U)                                # :70
    else:
        C = np.zeros((0, 0))
    W1 = PHt @ C                                            # :71
    W = W1 @ C.T                                            # :72
    xn = x + W @ v                                          # :74
    Pn = P - W1 @ W1.T                                      # :75
    return xn, Pn


def add_features(x, P, z, R):
    """src/ekf.jl:84-122.  Sequential over the new observations.  Returns NEW (x, P)."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    x = np.array(x, dtype=np.float64)
    P = np.array(P, dtype=np.float64)
    phi = x[2]                                              # :88 (fixed for the call)
    for i in range(z.shape[1]):
        ln = len(x)
        r, b = z[0, i], z[1, i]
        s, c = math.sin(phi + b), math.cos(phi + b)
        x = np.concatenate([x, [x[0] + r * c, x[1] + r * s]])          # :99
        Gv = np.array([[1.0, 0.0, -r * s], [0.0, 1.0, r * c]])          # :102
        Gz = np.array([[c, -r * s], [s, r * c]])                        # :103
        Pn = np.zeros((ln + 2, ln + 2))                                 # :108-109
        Pn[:ln, :ln] = P
        P = Pn
        rng = slice(ln, ln + 2)
        P[rng, rng] = Gv @ P[0:3, 0:3] @ Gv.T + Gz @ R @ Gz.T           # :112
        P[rng, 0:3] = Gv @ P[0:3, 0:3]                                  # :113
        P[0:3, rng] = P[rng, 0:3].T                                     # :114
        if ln > 3:
            rnm = slice(3, ln)
            P[rng, rnm] = Gv @ P[0:3, rnm]                              # :117
            P[rnm, rng] = P[rng, rnm].T                                 # :118
    return x, P


# --------------------------------------------------------------------------
# sparse twins (what the kernels compute), vectorised
# --------------------------------------------------------------------------

def _wrap_vec(a):
    """Vectorised single-step wrap of src/common.jl:102-110."""
    a = np.asarray(a, dtype=np.float64)
    return np.where(a > math.pi, a - 2 * math.pi, np.where(a < -math.pi, a + 2 * math.pi, a))


def obs_blocks(x, idf):
    """Non-zero blocks of predict_observation for landmark indices idf (1-based,
    array).  Returns zp (m,2), Hv (m,2,3), Hf (m,2,2).  src/common.jl:146-162."""
    x = np.asarray(x, dtype=np.float64)
    idf = np.asarray(idf, dtype=np.int64).reshape(-1)
    f = 3 + 2 * (idf - 1)
    dx = x[f] - x[0]
    dy = x[f + 1] - x[1]
    d2 = dx * dx + dy * dy
    d = np.sqrt(d2)
    zp = np.stack([d, np.arctan2(dy, dx) - x[2]], axis=1)
    xd, yd, xd2, yd2 = dx / d, dy / d, dx / d2, dy / d2
    m = len(idf)
    Hv = np.zeros((m, 2, 3))
    Hv[:, 0, 0] = -xd
    Hv[:, 0, 1] = -yd
    Hv[:, 1, 0] = yd2
    Hv[:, 1, 1] = -xd2
    Hv[:, 1, 2] = -1.0
    Hf = np.zeros((m, 2, 2))
    Hf[:, 0, 0] = xd
    Hf[:, 0, 1] = yd
    Hf[:, 1, 0] = -yd2
    Hf[:, 1, 1] = xd2
    return zp, Hv, Hf


def _landmark_S(x, P, R, idf):
    """S = H P H' + R from the 5 x 5 sub-block, for an array of landmarks.
    Returns zp (m,2), S (m,2,2)."""
    idf = np.asarray(idf, dtype=np.int64).reshape(-1)
    zp, Hv, Hf = obs_blocks(x, idf)
    f = 3 + 2 * (idf - 1)
    Pvv = P[0:3, 0:3]
    # the literal product uses the row strip P[0:3, f] on the right and the
    # column strip P[f, 0:3] on the left; keep both so an asymmetric P is
    # handled like the reference does.
    Pvf = np.stack([P[0:3, f], P[0:3, f + 1]], axis=2)           # (3, m, 2)
    Pvf = np.transpose(Pvf, (1, 0, 2))                           # (m, 3, 2)
    Pfv = np.stack([P[f, 0:3], P[f + 1, 0:3]], axis=1)           # (m, 2, 3)
    Pff = np.empty((len(idf), 2, 2))
    Pff[:, 0, 0] = P[f, f]
    Pff[:, 0, 1] = P[f, f + 1]
    Pff[:, 1, 0] = P[f + 1, f]
    Pff[:, 1, 1] = P[f + 1, f + 1]
    HvT = np.transpose(Hv, (0, 2, 1))
    HfT = np.transpose(Hf, (0, 2, 1))
    S = (Hv @ Pvv @ HvT + Hv @ Pvf @ HfT + Hf @ Pfv @ HvT + Hf @ Pff @ HfT) + R
    return zp, S


def compute_association_sparse(x, P, z, R, idf):
    """Sparse twin of compute_association for ONE landmark.  Returns (nis, nd)."""
    zp, S = _landmark_S(np.asarray(x, float), np.asarray(P, float), np.asarray(R, float), [idf])
    v0 = z[0] - zp[0, 0]
    v1 = mpi_to_pi(z[1] - zp[0, 1])
    S = S[0]
    det = S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]
    nis = (v0 * (S[1, 1] * v0 - S[0, 1] * v1) + v1 * (-S[1, 0] * v0 + S[0, 0] * v1)) / det
    return float(nis), float(nis + math.log(det))


def association_table_sparse(x, P, z, R):
    """(nis, nd) for every (observation, landmark) pair: arrays (nz, Nf)."""
    x = np.asarray(x, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    Nf = (len(x) - 3) // 2
    nz = z.shape[1]
    if Nf == 0:
        return np.zeros((nz, 0)), np.zeros((nz, 0))
    zp, S = _landmark_S(x, P if isinstance(P, LowRankCov) else np.asarray(P), np.asarray(R, dtype=np.float64),
                        np.arange(1, Nf + 1))
    det = S[:, 0, 0] * S[:, 1, 1] - S[:, 0, 1] * S[:, 1, 0]
    logdet = np.log(det)
    v0 = z[0][:, None] - zp[None, :, 0]
    v1 = _wrap_vec(z[1][:, None] - zp[None, :, 1])
    nis = (v0 * (S[:, 1, 1] * v0 - S[:, 0, 1] * v1) + v1 * (-S[:, 1, 0] * v0 + S[:, 0, 0] * v1)) / det
    return nis, nis + logdet


def assoc_vector(nis, nd, gate1, gate2):
    """Order-independent form of the scan in src/data-association.jl:21-50
    (SURVEY.md section 3.2).  Returns int32 assoc[nz]: j >= 1 matched, 0 dropped, -1 new."""
    nz, Nf = nis.shape
    out = np.zeros(nz, dtype=np.int32)
    for i in range(nz):
        if Nf == 0:
            out[i] = -1                   # outer = Inf > gate2
            continue
        with np.errstate(invalid="ignore"):
            cand = (nis[i] < gate1) & (nd[i] < math.inf)
        if cand.any():
            ndc = np.where(cand, nd[i], math.inf)
            out[i] = int(np.argmin(ndc)) + 1          # argmin: lowest index on ties
        else:
            fin = nis[i][~np.isnan(nis[i])]
            outer = fin.min() if fin.size else math.inf
            out[i] = -1 if outer > gate2 else 0
    return out


def split_assoc(z, assoc):
    """(zf, idf, zn) in the reference's shapes from the assoc vector, preserving
    observation order (src/data-association.jl:43-47)."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    assoc = np.asarray(assoc)
    zf = z[:, assoc > 0]
    idf = assoc[assoc > 0].astype(np.int64).reshape(1, -1)
    zn = z[:, assoc < 0]
    return zf, idf, zn


def associate_sparse(x, P, z, R, gate1, gate2):
    """Sparse, vectorised twin of associate().  Same return shapes."""
    nis, nd = association_table_sparse(x, P, z, R)
    return split_assoc(z, assoc_vector(nis, nd, gate1, gate2))


def _update_front(x, P, z, R, idf):
    """Innovation, PHt, S, C for the sparse forms (src/ekf.jl:55-70)."""
    x = np.asarray(x, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    idf = np.asarray(idf, dtype=np.int64).reshape(-1)
    m = z.shape[1]
    n = len(x)
    zp, Hv, Hf = obs_blocks(x, idf)
    v = np.empty(2 * m)
    v[0::2] = z[0] - zp[:, 0]
    v[1::2] = _wrap_vec(z[1] - zp[:, 1])
    f = 3 + 2 * (idf - 1)
    PHt = np.empty((n, 2 * m))
    for i in range(m):
        PHt[:, 2 * i:2 * i + 2] = P[:, 0:3] @ Hv[i].T + P[:, f[i]:f[i] + 2] @ Hf[i].T
    S = np.empty((2 * m, 2 * m))
    for i in range(m):
        S[2 * i:2 * i + 2, :] = Hv[i] @ PHt[0:3, :] + Hf[i] @ PHt[f[i]:f[i] + 2, :]
    S += np.kron(np.eye(m), R)
    S = (S + S.T) * 0.5
    return v, PHt, S


def update_sparse(x, P, z, R, idf, inplace=False):
    """Sparse twin of update(): sparse P*H', BLAS-level rank-k down-date."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    if z.shape[1] == 0:
        return (x, P) if inplace else (np.array(x, float), np.array(P, float))
    v, PHt, S = _update_front(x, P, z, R, idf)
    U = np.linalg.cholesky(S).T
    C = np.linalg.inv(U)
    W1 = PHt @ C
    xn = np.asarray(x, dtype=np.float64) + W1 @ (C.T @ v)
    if inplace:
        P -= W1 @ W1.T
        return xn, P
    return xn, P - W1 @ W1.T


class LowRankCov:
    """P = A A' + d I as an indexable VIEW that never forms the n x n matrix: test infrastructure for BASELINE.json's
    N = 50k configuration (80 GB in fp64), where the GPU state is built from the same factor.  Supports exactly the
    index patterns of the sparse oracle: (slice | index array) x (slice | index array); two index arrays pair up
    element by element like NumPy's, anything with a slice is an outer block."""

    def __init__(self, A, d):
        self.A = np.asarray(A, dtype=np.float64)
        self.d = float(d)
        self.shape = (self.A.shape[0], self.A.shape[0])

    def _idx(self, k):
        n = self.shape[0]
        if isinstance(k, slice):
            return np.arange(*k.indices(n)), True
        return np.asarray(k, dtype=np.int64).reshape(-1), False

    def __getitem__(self, key):
        r, rs = self._idx(key[0])
        c, cs = self._idx(key[1])
        if not rs and not cs:                                   # P[f, g]: element by element
            return np.einsum("ij,ij->i", self.A[r], self.A[c]) + self.d * (r == c)
        return self.A[r] @ self.A[c].T + self.d * (r[:, None] == c[None, :])

    def diagonal(self):
        return np.einsum("ij,ij->i", self.A, self.A) + self.d


def update_joseph_factors(x, P, z, R, idf):
    """The Joseph-form update of update_joseph_sparse WITHOUT forming P+:  returns (x+, K, T) with
    P+ = P - K T' - T K'.  P may be a LowRankCov (only the pose columns and the observed landmarks' columns
    of P are read)."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    v, A, S = _update_front(x, P, z, R, idf)
    K = np.linalg.solve(S, A.T).T
    T = A - 0.5 * K @ S
    return np.asarray(x, dtype=np.float64) + K @ v, K, T


def joseph_block(P, K, T, rows, cols):
    """P+[rows, cols] from the factors of update_joseph_factors."""
    return P[rows, cols] - K[rows] @ T[cols].T - T[rows] @ K[cols].T


def update_joseph_sparse(x, P, z, R, idf):
    """Joseph-form covariance update in its one-pass rank-2k shape (SURVEY.md
    section 8d, NOT in the reference):  A = P H', S = H A + RR, K = A S^-1,
    T = A - K S / 2,  P+ = P - K T' - T K'  (= (I-KH) P (I-KH)' + K RR K')."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    if z.shape[1] == 0:
        return np.array(x, float), np.array(P, float)
    v, A, S = _update_front(x, P, z, R, idf)
    K = np.linalg.solve(S, A.T).T
    T = A - 0.5 * K @ S
    xn = np.asarray(x, dtype=np.float64) + K @ v
    return xn, P - K @ T.T - T @ K.T


def add_features_sparse(x, P, z, R):
    """Strip form of add_features(): every new row/column only needs the pose
    rows of P, so all new features can be written independently."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    x = np.array(x, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    nn = z.shape[1]
    n0 = len(x)
    if nn == 0:
        return x, np.array(P)
    phi = x[2]
    xn = np.concatenate([x, np.zeros(2 * nn)])
    Pn = np.zeros((n0 + 2 * nn, n0 + 2 * nn))
    Pn[:n0, :n0] = P
    Pvv = P[0:3, 0:3]
    Gvs = []
    for i in range(nn):
        r, b = z[0, i], z[1, i]
        s, c = math.sin(phi + b), math.cos(phi + b)
        xn[n0 + 2 * i] = x[0] + r * c
        xn[n0 + 2 * i + 1] = x[1] + r * s
        Gv = np.array([[1.0, 0.0, -r * s], [0.0, 1.0, r * c]])
        Gz = np.array([[c, -r * s], [s, r * c]])
        Gvs.append(Gv)
        rng = slice(n0 + 2 * i, n0 + 2 * i + 2)
        Pn[rng, rng] = Gv @ Pvv @ Gv.T + Gz @ R @ Gz.T
        Pn[rng, 0:3] = Gv @ Pvv
        Pn[0:3, rng] = Pn[rng, 0:3].T
        if n0 > 3:
            Pn[rng, 3:n0] = Gv @ P[0:3, 3:n0]
            Pn[3:n0, rng] = Pn[rng, 3:n0].T
        for k in range(i):                      # earlier new features (ekf.jl:116, rnm grows)
            rk = slice(n0 + 2 * k, n0 + 2 * k + 2)
            Pn[rng, rk] = Gv @ (Gvs[k] @ Pvv).T
            Pn[rk, rng] = Pn[rng, rk].T
    return xn, Pn


def predict_sparse(x, P, v, g, w, Q, dt):
    """Strip form of predict(): reads the COLUMN strip P[3:, 0:3] (contiguous in
    column-major storage) instead of the row strip.  In place, like predict()."""
    phi = x[2]
    s, c = math.sin(g + phi), math.cos(g + phi)
    vts, vtc = v * dt * s, v * dt * c
    Gv = np.array([[1.0, 0.0, -vts], [0.0, 1.0, vtc], [0.0, 0.0, 1.0]])
    Gu = np.array([[dt * c, -vts], [dt * s, vtc],
                   [dt * math.sin(g) / w, v * dt * math.cos(g) / w]])
    P[0:3, 0:3] = Gv @ P[0:3, 0:3] @ Gv.T + Gu @ Q @ Gu.T
    if P.shape[0] > 3:
        col = P[3:, 0:3].copy()                 # (2N, 3): col[c, :] = P_vm[:, c]
        new = np.empty_like(col)
        new[:, 0] = col[:, 0] - vts * col[:, 2]
        new[:, 1] = col[:, 1] + vtc * col[:, 2]
        new[:, 2] = col[:, 2]
        P[3:, 0:3] = new
        P[0:3, 3:] = new.T
    x[0:3] = [x[0] + vtc, x[1] + vts, mpi_to_pi(phi + v * dt * math.sin(g) / w)]
    return x, P


# --------------------------------------------------------------------------
# filter object with the call surface the headless sim driver expects
# --------------------------------------------------------------------------

class OracleEKF:
    """Holds (x, P) and forwards to the literal (default) or sparse functions.
    Mirrors how ``sim!`` rebinds ``state.x, state.cov`` after every call
    (sim/ekfslam-sim.jl:100,117,120)."""

    def __init__(self, x, P, sparse=False):
        self.x = np.array(x, dtype=np.float64)
        self.cov = np.array(P, dtype=np.float64)
        self.sparse = sparse

    def predict(self, v, g, wheelbase, Q, dt):
        fn = predict_sparse if self.sparse else predict
        self.x, self.cov = fn(self.x, self.cov, v, g, wheelbase, np.asarray(Q, float), dt)

    def associate(self, z, R, gate1, gate2):
        fn = associate_sparse if self.sparse else associate
        return fn(self.x, self.cov, z, np.asarray(R, float), gate1, gate2)

    def update(self, zf, R, idf):
        fn = update_sparse if self.sparse else update
        self.x, self.cov = fn(self.x, self.cov, zf, np.asarray(R, float), idf)

    def add_features(self, zn, R):
        fn = add_features_sparse if self.sparse else add_features
        self.x, self.cov = fn(self.x, self.cov, zn, np.asarray(R, float))

    def pose(self):
        return self.x[0:3].copy()


# ---- telemetry (sim/browser/wsserver.jl:60-65,72-85) -----------------------------------------------

def feature_ellipses(x, cov):
    """``feature_ellipses`` (sim/browser/wsserver.jl:72-85): per landmark ``[x_j; sqrt(l); atan2(u[2,1], u[1,1])]``
    with ``l, u = eig(cov[j, j])`` (ascending eigenvalues, LAPACK eigenvectors -- their sign is arbitrary, so phi
    is defined modulo pi)."""
    nf = (len(x) - 3) // 2
    out = np.empty((5, nf))
    for i in range(nf):
        j = slice(3 + 2 * i, 5 + 2 * i)
        l, u = np.linalg.eigh(np.asarray(cov, dtype=np.float64)[j, j])
        out[:, i] = [x[3 + 2 * i], x[4 + 2 * i], math.sqrt(max(l[0], 0.0)), math.sqrt(max(l[1], 0.0)),
                     math.atan2(u[1, 0], u[0, 0])]
    return out


def vehicle_ellipse(x, cov):
    """The ``vehicle-ellipse`` record of ``monitor`` (sim/browser/wsserver.jl:60-65)."""
    l, u = np.linalg.eigh(np.asarray(cov, dtype=np.float64)[0:2, 0:2])
    return np.array([x[0], x[1], x[2], math.sqrt(max(l[0], 0.0)), math.sqrt(max(l[1], 0.0)), math.atan2(u[1, 0], u[0, 0])])
