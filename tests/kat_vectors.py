"""Hand-derived known answers KAT-8 .. KAT-11 (closed forms worked on paper from the reference's formulas,
never by running the oracle).  Shared by the CPU tests of the oracle (tests/test_oracle_kat.py) and the GPU tests
through the C ABI (tests/test_gpu_ekf.py).

KAT-8  ``update`` (src/ekf.jl:46-77), one observation.
    x = [0, 0, 0, 10, 0] (pose at the origin, heading 0, landmark 1 at (10, 0)), P = diag(p1..p5), R = diag(r1, r2),
    z = (10.5, 0.02).  From KAT-1, H = [[-1, 0, 0, 1, 0], [0, -0.1, -1, 0, 0.1]]: row 1 touches the state indices
    {0, 3}, row 2 touches {1, 2, 4}, and P is diagonal, so the two measurement rows decouple:
        S = diag(s1, s2),   s1 = p1 + p4 + r1,   s2 = 0.01 p2 + p3 + 0.01 p5 + r2
        c1 = P h1' = [-p1, 0, 0, p4, 0]',        c2 = P h2' = [0, -0.1 p2, -p3, 0, 0.1 p5]'
        v = (0.5, 0.02)
        K = [c1 / s1, c2 / s2],   x+ = x + c1 v1 / s1 + c2 v2 / s2
        P+ = P - c1 c1' / s1 - c2 c2' / s2           (= P - W1 W1' with W1 = P H' inv(chol(S)), ekf.jl:67-75)

KAT-9  ``update`` with TWO stacked observations of the same landmark (rows are simply stacked).
    Same x, P, R; z_a = (10.5, 0.02), z_b = (9.8, -0.01).  The stacked Jacobian is [H; H]; the range rows (1, 3)
    and the bearing rows (2, 4) still decouple.  For one pair of identical rows h with prior variance a = h P h' and
    noise r:   S = a 11' + r I,   S^-1 = (I - a/(r + 2a) 11') / r,   so
        [h; h]' S^-1 [h; h] = h'h * 2 / (r + 2a)      and      P [h; h]' S^-1 = c 1' / (r + 2a),   c = P h'
    giving   P+ = P - 2 c1 c1' / (r1 + 2 a1) - 2 c2 c2' / (r2 + 2 a2)
             x+ = x + c1 (v1a + v1b) / (r1 + 2 a1) + c2 (v2a + v2b) / (r2 + 2 a2)
    with a1 = p1 + p4, a2 = 0.01 p2 + p3 + 0.01 p5 (two measurements of noise r are one of noise r/2 at the mean).

KAT-10 ``add_features`` (src/ekf.jl:84-122) with a NON-ZERO vehicle covariance and an EXISTING landmark (the cross
    block with the rest of the map, ``rnm``, :115-118).  x = [1, 2, 0.5, 4, 6]; new observation (r, b) = (2, pi/2 - 0.5),
    so phi + b = pi/2, s = 1, c = 0:
        new landmark at (xv + r c, yv + r s) = (1, 4)
        Gv = [[1, 0, -r s], [0, 1, r c]] = [[1, 0, -2], [0, 1, 0]]      Gz = [[c, -r s], [s, r c]] = [[0, -2], [1, 0]]
    With Pvv = [[a, d, e], [d, b, f], [e, f, g]], Pvm = [[h1, h2], [i1, i2], [j1, j2]] and R = diag(r1, r2):
        P_fv = Gv Pvv       = [[a - 2e, d - 2f, e - 2g], [d, b, f]]
        P_ff = Gv Pvv Gv' + Gz R Gz' = [[a - 4e + 4g + 4 r2, d - 2f], [d - 2f, b + r1]]
        P_fm = Gv Pvm       = [[h1 - 2 j1, h2 - 2 j2], [i1, i2]]
    and the old 5 x 5 block is unchanged.

KAT-11 ``update`` with a ROTATED heading, an OFF-AXIS landmark and a COUPLED covariance (KAT-8/9 have the landmark on
    the x axis, phi = 0 and a diagonal P: nothing there exercises the bearing's -phi term, the sign pattern of the
    off-axis Jacobian or a P with cross terms).  A 3-4-5 triangle keeps the Jacobian rational:
        x = [1, 2, pi/6, 4, 6]:  dx = 3, dy = 4, d = 5, d^2 = 25
        zhat = (5, atan2(4, 3) - pi/6)                                        (src/common.jl:148-152)
        H = [[-3/5, -4/5, 0, 3/5, 4/5], [4/25, -3/25, -1, -4/25, 3/25]]       (:161-162)
    z = zhat + (0.3, -0.015) so v = (0.3, -0.015) exactly as data; P = the coupled 5 x 5 matrix of KAT-10 (positive
    definite).  The expected posterior is NOT formed the reference's way (P H' inv(chol(S)), W1 W1') but through the
    INFORMATION form of the same linear-Gaussian update, whose only inverses are a 2 x 2 diagonal one (R) and 5 x 5
    ones:        Lambda+ = P^-1 + H' R^-1 H,     P+ = (Lambda+)^-1,     x+ = x + P+ H' R^-1 v
    (Woodbury: (P^-1 + H' R^-1 H)^-1 = P - P H' (H P H' + R)^-1 H P, and P+ H' R^-1 = P H' S^-1 = K.)

KAT-12 ``predict`` (src/ekf.jl:8-43) with a ROTATED heading, a STEERING angle, a COUPLED covariance and a landmark
    (KAT-3 starts from x = 0, P = 0: nothing there exercises Gv P_vv Gv', the map strip Gv P_vm or sin g / cos g).
    phi = pi/3, g = pi/6, so g + phi = pi/2: s = 1, c = 0; v = 4, dt = 0.25: v dt = 1, vts = 1, vtc = 0; w = 2:
        x+ = [x + vtc, y + vts, mpi_to_pi(phi + v dt sin(g) / w)] = [1, 3, pi/3 + 0.25]          (:39-41, pre-update phi)
        Gv = [[1, 0, -1], [0, 1, 0], [0, 0, 1]]                                                  (:24-26)
        Gu = [[dt c, -vts], [dt s, vtc], [dt sin(g)/w, v dt cos(g)/w]] = [[0, -1], [1/4, 0], [1/16, sqrt(3)/4]]   (:27-29)
    With Pvv = [[a, d, e], [d, b, f], [e, f, g0]], Pvm = [[h1, h2], [i1, i2], [j1, j2]] (KAT-10's matrix), Q = diag(q1, q2):
        Gv Pvv Gv' = [[a - 2e + g0, d - f, e - g0], [d - f, b, f], [e - g0, f, g0]]
        Gu Q Gu'   = q1 [[0, 0, 0], [0, 1/16, 1/64], [0, 1/64, 1/256]] + q2 [[1, 0, -r3], [0, 0, 0], [-r3, 0, 3/16]],  r3 = sqrt(3)/4
        P_vm+ = Gv Pvm = [[h1 - j1, h2 - j2], [i1, i2], [j1, j2]],     P_mm unchanged                (:32-36)

KAT-13 ``compute_association`` / ``associate`` (src/data-association.jl:53-63, :21-50) with a NON-SYMMETRIC innovation
    covariance.  :59-60 form ``S = H*P*H' + R`` and ``nis = dot(v, inv(S)*v)`` WITHOUT symmetrising S (``update`` does
    symmetrise, src/ekf.jl:69; the association does not), so a P_vv or an R whose two off-diagonal entries differ gives
    S[1,2] != S[2,1] and  inv(S) = [[s22, -s12], [-s21, s11]] / (s11 s22 - s12 s21):
        nis = (s22 v1^2 - (s12 + s21) v1 v2 + s11 v2^2) / (s11 s22 - s12 s21)            (1-based, the reference's indices)
    -- NOT the value a symmetrised S' = (S + S')/2 gives (its determinant is s11 s22 - ((s12 + s21)/2)^2).
    KAT-2's geometry: x = [0, 0, 0, 10, 0], H = [[-1, 0, 0, 1, 0], [0, -0.1, -1, 0, 0.1]], H H' = diag(2, 1.02).
    P = I + a e1 e2' + b e2 e1' (only P[1,2] = a and P[2,1] = b differ from the identity): H e1 = (-1, 0)', H e2 = (0, -0.1)',
        H P H' = diag(2, 1.02) + a (H e1)(H e2)' + b (H e2)(H e1)' = [[2, 0.1 a], [0.1 b, 1.02]]
    and R = [[r11, r12], [r21, r22]] is added entry by entry:  S = [[2 + r11, 0.1 a + r12], [0.1 b + r21, 1.02 + r22]].
    With a = 4, b = -2, r12 = 0.1, r21 = -0.1:  s12 = 0.5, s21 = -0.3: det S = s11 s22 + 0.15, the symmetrised
    determinant is s11 s22 - 0.01 -- the two nis differ by 8 %, and gate1 is put between them: the reference's rule
    matches the observation, a symmetrising implementation would drop it.
"""
import math

import numpy as np

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])

# ---- KAT-8 / KAT-9 ------------------------------------------------------------------------------------------
X8 = np.array([0.0, 0.0, 0.0, 10.0, 0.0])
P8 = np.array([0.5, 0.4, 0.02, 1.0, 2.0])           # diag(P)
Z8A = np.array([10.5, 0.02])
Z8B = np.array([9.8, -0.01])


def _c12():
    p1, p2, p3, p4, p5 = P8
    c1 = np.array([-p1, 0.0, 0.0, p4, 0.0])
    c2 = np.array([0.0, -0.1 * p2, -p3, 0.0, 0.1 * p5])
    a1 = p1 + p4
    a2 = 0.01 * p2 + p3 + 0.01 * p5
    return c1, c2, a1, a2


def kat8():
    """(x, P, z (2 x 1), idf, x_plus, P_plus)"""
    c1, c2, a1, a2 = _c12()
    s1, s2 = a1 + R[0, 0], a2 + R[1, 1]
    v1, v2 = Z8A[0] - 10.0, Z8A[1] - 0.0
    xp = X8 + c1 * v1 / s1 + c2 * v2 / s2
    Pp = np.diag(P8) - np.outer(c1, c1) / s1 - np.outer(c2, c2) / s2
    return X8.copy(), np.diag(P8), Z8A.reshape(2, 1), np.array([[1]]), xp, Pp


def kat9():
    """(x, P, z (2 x 2), idf, x_plus, P_plus): both observations belong to landmark 1"""
    c1, c2, a1, a2 = _c12()
    g1, g2 = R[0, 0] + 2.0 * a1, R[1, 1] + 2.0 * a2
    v1 = (Z8A[0] - 10.0) + (Z8B[0] - 10.0)
    v2 = Z8A[1] + Z8B[1]
    xp = X8 + c1 * v1 / g1 + c2 * v2 / g2
    Pp = np.diag(P8) - 2.0 * np.outer(c1, c1) / g1 - 2.0 * np.outer(c2, c2) / g2
    return X8.copy(), np.diag(P8), np.stack([Z8A, Z8B], axis=1), np.array([[1, 1]]), xp, Pp


# ---- KAT-10 -------------------------------------------------------------------------------------------------
def kat10():
    """(x, P, zn (2 x 1), x_plus, P_plus)"""
    a, b, g, d, e, f = 0.30, 0.20, 0.01, 0.05, 0.02, -0.01
    h1, h2, i1, i2, j1, j2 = 0.03, -0.02, 0.01, 0.04, 0.005, -0.003
    x = np.array([1.0, 2.0, 0.5, 4.0, 6.0])
    P = np.array([[a, d, e, h1, h2],
                  [d, b, f, i1, i2],
                  [e, f, g, j1, j2],
                  [h1, i1, j1, 0.5, 0.1],
                  [h2, i2, j2, 0.1, 0.4]])
    zn = np.array([[2.0], [math.pi / 2 - 0.5]])
    r1, r2 = R[0, 0], R[1, 1]
    xp = np.concatenate([x, [1.0, 4.0]])
    Pfv = np.array([[a - 2 * e, d - 2 * f, e - 2 * g], [d, b, f]])
    Pff = np.array([[a - 4 * e + 4 * g + 4 * r2, d - 2 * f], [d - 2 * f, b + r1]])
    Pfm = np.array([[h1 - 2 * j1, h2 - 2 * j2], [i1, i2]])
    Pp = np.zeros((7, 7))
    Pp[:5, :5] = P
    Pp[5:, 0:3] = Pfv
    Pp[0:3, 5:] = Pfv.T
    Pp[5:, 3:5] = Pfm
    Pp[3:5, 5:] = Pfm.T
    Pp[5:, 5:] = Pff
    return x, P, zn, xp, Pp


# ---- KAT-11 -------------------------------------------------------------------------------------------------
def kat11():
    """(x, P, z (2 x 1), idf, x_plus, P_plus): rotated heading, off-axis landmark, coupled P; information form"""
    _, P, _, _, _ = kat10()
    x = np.array([1.0, 2.0, math.pi / 6, 4.0, 6.0])
    H = np.array([[-3 / 5, -4 / 5, 0.0, 3 / 5, 4 / 5],
                  [4 / 25, -3 / 25, -1.0, -4 / 25, 3 / 25]])
    v = np.array([0.3, -0.015])
    z = np.array([5.0 + v[0], math.atan2(4.0, 3.0) - math.pi / 6 + v[1]]).reshape(2, 1)
    Ri = np.diag([1.0 / R[0, 0], 1.0 / R[1, 1]])
    Pp = np.linalg.inv(np.linalg.inv(P) + H.T @ Ri @ H)
    Pp = 0.5 * (Pp + Pp.T)
    xp = x + Pp @ H.T @ Ri @ v
    return x, P, z, np.array([[1]]), xp, Pp


def kat12():
    """(x, P, (v, g, w, Q, dt), x_plus, P_plus): predict with heading, steering, coupled P and a landmark"""
    a, b, g0, d, e, f = 0.30, 0.20, 0.01, 0.05, 0.02, -0.01
    h1, h2, i1, i2, j1, j2 = 0.03, -0.02, 0.01, 0.04, 0.005, -0.003
    _, P, _, _, _ = kat10()
    x = np.array([1.0, 2.0, math.pi / 3, 4.0, 6.0])
    q1, q2 = 0.5 ** 2, (3 * math.pi / 180) ** 2
    Q = np.diag([q1, q2])
    r3 = math.sqrt(3.0) / 4
    xp = np.array([1.0, 3.0, math.pi / 3 + 0.25, 4.0, 6.0])
    Pp = P.copy()
    Pp[0:3, 0:3] = (np.array([[a - 2 * e + g0, d - f, e - g0], [d - f, b, f], [e - g0, f, g0]])
                    + q1 * np.array([[0, 0, 0], [0, 1 / 16, 1 / 64], [0, 1 / 64, 1 / 256]])
                    + q2 * np.array([[1, 0, -r3], [0, 0, 0], [-r3, 0, 3 / 16]]))
    Pp[0:3, 3:5] = np.array([[h1 - j1, h2 - j2], [i1, i2], [j1, j2]])
    Pp[3:5, 0:3] = Pp[0:3, 3:5].T
    return x, P, (4.0, math.pi / 6, 2.0, Q, 0.25), xp, Pp


# ---- KAT-13 -------------------------------------------------------------------------------------------------
def kat13():
    """(x, P, R13, z (2,), nis, nd, nis_if_symmetrised, gate1, gate2): one landmark, non-symmetric P_vv and R"""
    a, b = 4.0, -2.0
    r11, r22, r12, r21 = R[0, 0], R[1, 1], 0.1, -0.1
    x = np.array([0.0, 0.0, 0.0, 10.0, 0.0])
    P = np.eye(5)
    P[0, 1] = a
    P[1, 0] = b
    R13 = np.array([[r11, r12], [r21, r22]])
    s11, s12, s21, s22 = 2.0 + r11, 0.1 * a + r12, 0.1 * b + r21, 1.02 + r22
    v1, v2 = 1.5, 1.0                                    # z = zhat + v, zhat = (10, 0)
    det = s11 * s22 - s12 * s21
    nis = (s22 * v1 * v1 - (s12 + s21) * v1 * v2 + s11 * v2 * v2) / det
    nd = nis + math.log(det)
    sm = 0.5 * (s12 + s21)
    det_s = s11 * s22 - sm * sm
    nis_s = (s22 * v1 * v1 - 2.0 * sm * v1 * v2 + s11 * v2 * v2) / det_s
    gate1 = 0.5 * (nis + nis_s)                          # nis < gate1 < nis_s
    return x, P, R13, np.array([10.0 + v1, v2]), nis, nd, nis_s, gate1, 25.0
