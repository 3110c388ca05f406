"""oracle/pf_ref.py -- CPU restatement of the FastSLAM particle path (1.0, unknown correspondences, 2.0 proposal).

TEST INFRASTRUCTURE ONLY (see oracle/ekf_ref.py for the rules).

PARITY UNPINNED: the reference implements NO particle filter -- only the data types ``Particle``
(src/common.jl:14-20) and ``PFSlamState`` (src/common.jl:31-34) exist and README.md:6 says
"FastSLAM is ongoing".  The algorithm below is specified in SURVEY.md section 8a rows F1-F4 from the
reference's own EKF building blocks, and this float64 NumPy restatement is the only oracle:

  F1  per-particle control noise (sim/sim-utils.jl:35-38) + the pose update of src/ekf.jl:39-41
  F2  per-landmark 2x2 EKF: the feature block of predict_observation (src/common.jl:162) and the
      Cholesky-form update of src/ekf.jl:67-75 restricted to the 2x2 feature block; weight
      w *= N(v; 0, S)
  F3  new landmark: src/ekf.jl:94-103,112 without the vehicle-covariance term
  F4  weight normalisation, Neff = 1 / sum(w^2), systematic resampling

Random numbers are counter based (Philox4x32-10) and keyed by (seed, step, GLOBAL particle id), so a
run does not depend on how the particles are split over GPUs.
"""
from __future__ import annotations

import math

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-10 (Salmon et al. 2011), vectorised over uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.asarray(c1, dtype=np.uint32).copy()
    c2 = np.asarray(c2, dtype=np.uint32).copy()
    c3 = np.asarray(c3, dtype=np.uint32).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(x):
    """uint32 -> (0, 1): 24 random bits, offset by half a step (never 0 or 1)."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) + 0.5) * (1.0 / 16777216.0)


def normals2(gid, step, stream, seed):
    """Two standard normals per global particle id (Box-Muller on Philox output words 0, 1)."""
    gid = np.asarray(gid, dtype=np.uint64)
    r = philox4x32((gid & MASK32).astype(np.uint32), (gid >> np.uint64(32)).astype(np.uint32),
                   np.full(gid.shape, step, dtype=np.uint32), np.full(gid.shape, stream, dtype=np.uint32),
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1, u2 = _u01(r[0]), _u01(r[1])
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.cos(2 * math.pi * u2), rad * np.sin(2 * math.pi * u2)


def uniform1(step, stream, seed):
    """One U(0,1) shared by all ranks (the systematic-resampling offset)."""
    r = philox4x32(np.zeros(1, np.uint32), np.zeros(1, np.uint32), np.full(1, step, np.uint32),
                   np.full(1, stream, np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return float(_u01(r[0])[0])


STREAM_PREDICT, STREAM_INIT, STREAM_RESAMPLE = 0, 1, 2


def _wrap(a):
    return np.where(a > math.pi, a - 2 * math.pi, np.where(a < -math.pi, a + 2 * math.pi, a))


class OraclePF:
    """FastSLAM-1.0 with known correspondences; state in SoA float64 arrays.

    lm[l, c, p]: landmark l of particle p, c = (x, y, Pxx, Pxy, Pyy);  seen[l] is global.
    """

    def __init__(self, n_particles, max_landmarks, seed, first_id=0, n_global=None):
        self.n = int(n_particles)
        self.first = int(first_id)
        self.n_global = int(n_global if n_global is not None else n_particles)
        self.nl = int(max_landmarks)
        self.seed = int(seed)
        self.step = 0
        self.resamples = 0
        self.pose = np.zeros((3, self.n))
        self.logw = np.full(self.n, -math.log(self.n_global))
        self.lm = np.zeros((self.nl, 5, self.n))
        self.seen = np.zeros(self.nl, dtype=bool)

    @property
    def gids(self):
        return np.arange(self.first, self.first + self.n, dtype=np.uint64)

    def set_pose(self, pose):
        self.pose[:] = np.asarray(pose, dtype=np.float64).reshape(3, 1)

    def init_landmarks(self, lm_xy, var, jitter_sigma):
        """Every particle gets landmark l at truth + N(0, jitter^2) with Pf = diag(var, var)."""
        lm_xy = np.asarray(lm_xy, dtype=np.float64).reshape(-1, 2)
        for l in range(lm_xy.shape[0]):
            e1, e2 = normals2(self.gids, l, STREAM_INIT, self.seed)
            self.lm[l, 0] = lm_xy[l, 0] + jitter_sigma * e1
            self.lm[l, 1] = lm_xy[l, 1] + jitter_sigma * e2
            self.lm[l, 2] = var
            self.lm[l, 3] = 0.0
            self.lm[l, 4] = var
            self.seen[l] = True

    # F1 ------------------------------------------------------------------------------------------
    def predict(self, V, G, wheelbase, Q, dt):
        Q = np.asarray(Q, dtype=np.float64)
        e1, e2 = normals2(self.gids, self.step, STREAM_PREDICT, self.seed)
        Vn = V + math.sqrt(Q[0, 0]) * e1                      # sim/sim-utils.jl:36
        Gn = G + math.sqrt(Q[1, 1]) * e2                      # :37
        x, y, phi = self.pose
        self.pose = np.stack([x + Vn * dt * np.cos(Gn + phi),          # src/ekf.jl:39-41
                              y + Vn * dt * np.sin(Gn + phi),
                              _wrap(phi + Vn * dt * np.sin(Gn) / wheelbase)])
        self.step += 1

    # F2 / F3 --------------------------------------------------------------------------------------
    def update_known(self, z, ids, R):
        z = np.asarray(z, dtype=np.float64).reshape(2, -1)
        R = np.asarray(R, dtype=np.float64)
        x, y, phi = self.pose
        for i, l1 in enumerate(np.asarray(ids).reshape(-1)):
            l = int(l1) - 1                                   # 1-based like idf
            r, b = z[0, i], z[1, i]
            if not self.seen[l]:
                s, c = np.sin(phi + b), np.cos(phi + b)       # src/ekf.jl:94-103
                self.lm[l, 0] = x + r * c
                self.lm[l, 1] = y + r * s
                # Gz = [c -r*s; s r*c];  Pf = Gz R Gz'        (:103,:112 without the pose term)
                g00, g01, g10, g11 = c, -r * s, s, r * c
                a00 = g00 * R[0, 0] + g01 * R[1, 0]
                a01 = g00 * R[0, 1] + g01 * R[1, 1]
                a10 = g10 * R[0, 0] + g11 * R[1, 0]
                a11 = g10 * R[0, 1] + g11 * R[1, 1]
                self.lm[l, 2] = a00 * g00 + a01 * g01
                self.lm[l, 3] = a00 * g10 + a01 * g11
                self.lm[l, 4] = a10 * g10 + a11 * g11
                self.seen[l] = True
                continue
            lx, ly, pxx, pxy, pyy = self.lm[l]
            dx, dy = lx - x, ly - y
            d2 = dx * dx + dy * dy
            d = np.sqrt(d2)
            v0 = r - d                                        # src/ekf.jl:58
            v1 = _wrap(b - (np.arctan2(dy, dx) - phi))
            h00, h01, h10, h11 = dx / d, dy / d, -dy / d2, dx / d2   # src/common.jl:162
            # PHt = Pf Hf'
            t00 = pxx * h00 + pxy * h01
            t01 = pxx * h10 + pxy * h11
            t10 = pxy * h00 + pyy * h01
            t11 = pxy * h10 + pyy * h11
            # S = Hf PHt + R, symmetrised (src/ekf.jl:68-69)
            s00 = h00 * t00 + h01 * t10 + R[0, 0]
            s01 = h00 * t01 + h01 * t11 + R[0, 1]
            s10 = h10 * t00 + h11 * t10 + R[1, 0]
            s11 = h10 * t01 + h11 * t11 + R[1, 1]
            s01 = 0.5 * (s01 + s10)
            # chol(S) = U upper: u00, u01, u11;  C = inv(U)  (:70)
            u00 = np.sqrt(s00)
            u01 = s01 / u00
            u11 = np.sqrt(s11 - u01 * u01)
            c00, c01, c11 = 1.0 / u00, -u01 / (u00 * u11), 1.0 / u11
            # W1 = PHt C (:71)
            w00 = t00 * c00
            w01 = t00 * c01 + t01 * c11
            w10 = t10 * c00
            w11 = t10 * c01 + t11 * c11
            # y = C' v ; x += W1 y (= W v, :72,:74) ; P -= W1 W1' (:75)
            y0 = c00 * v0
            y1 = c01 * v0 + c11 * v1
            self.lm[l, 0] = lx + w00 * y0 + w01 * y1
            self.lm[l, 1] = ly + w10 * y0 + w11 * y1
            self.lm[l, 2] = pxx - (w00 * w00 + w01 * w01)
            self.lm[l, 3] = pxy - (w00 * w10 + w01 * w11)
            self.lm[l, 4] = pyy - (w10 * w10 + w11 * w11)
            # w *= exp(-nis/2) / (2 pi sqrt(det S));  nis = y'y,  sqrt(det S) = u00*u11
            self.logw = self.logw - 0.5 * (y0 * y0 + y1 * y1) - np.log(u00 * u11) - math.log(2 * math.pi)

    # N4: FastSLAM-2.0 proposal -------------------------------------------------------------------
    def step_proposal(self, V, G, wheelbase, Q, dt, z, ids, R):
        """One FastSLAM-2.0 step (Montemerlo et al. 2003; no reference code): the pose is drawn from
        the proposal that already knows this step's observations of landmarks the particle holds.

        The proposal lives in CONTROL space: pose = f(pose, V + u0, G + u1) with the motion model of
        src/ekf.jl:39-41 and u = Lq w, Lq = chol(Q) (lower), w ~ N(0, I) a priori (sim/sim-utils.jl:35-38).
        Around w = 0 the pose moves by GL w with GL = Gu Lq (Gu: src/ekf.jl:27-29), so observation i of a landmark
        with 2 x 2 block (Hf, Pf) (src/common.jl:161-162) is a linear measurement of w,
            v_i = z_i - h(f(pose, V, G)) = B_i w + noise,  B_i = Hv_i GL,  noise ~ N(0, Sf_i),  Sf_i = Hf Pf Hf' + R,
        assimilated one after the other in the Cholesky form of src/ekf.jl:67-75 (2 x 2 throughout).  The weight is
        the product of the predictive densities N(v_i - B_i mu; 0, B_i Sig B_i' + Sf_i); then w = mu + chol(Sig) e
        with the SAME two normals FastSLAM-1.0's predict would use, the pose follows from the exact motion model,
        and the landmarks are updated from that pose as in update_known (weights untouched).  First sightings
        say nothing about the pose; they are initialised from the sampled pose.  Without observations the step
        is FastSLAM-1.0's predict, bit for bit."""
        Q = np.asarray(Q, dtype=np.float64)
        R = np.asarray(R, dtype=np.float64)
        z = np.asarray(z, dtype=np.float64).reshape(2, -1)
        ids = np.asarray(ids).reshape(-1)
        x, y, phi = self.pose
        lq00 = math.sqrt(Q[0, 0])
        lq10 = 0.5 * (Q[0, 1] + Q[1, 0]) / lq00
        lq11 = math.sqrt(Q[1, 1] - lq10 * lq10)
        # motion mean (u = 0) and GL = Gu Lq
        s, c = np.sin(G + phi), np.cos(G + phi)
        vts, vtc = V * dt * s, V * dt * c
        xm, ym = x + vtc, y + vts
        pm = _wrap(phi + V * dt * math.sin(G) / wheelbase)
        gu20, gu21 = dt * math.sin(G) / wheelbase, V * dt * math.cos(G) / wheelbase
        gl00, gl01 = dt * c * lq00 + (-vts) * lq10, (-vts) * lq11
        gl10, gl11 = dt * s * lq00 + vtc * lq10, vtc * lq11
        gl20, gl21 = gu20 * lq00 + gu21 * lq10, gu21 * lq11
        mu0 = np.zeros(self.n)
        mu1 = np.zeros(self.n)
        g00 = np.ones(self.n)
        g01 = np.zeros(self.n)
        g11 = np.ones(self.n)
        logw = self.logw
        for i, l1 in enumerate(ids):
            l = int(l1) - 1
            if not self.seen[l]:
                continue
            r, b = z[0, i], z[1, i]
            lx, ly, pxx, pxy, pyy = self.lm[l]
            dx, dy = lx - xm, ly - ym
            d2 = dx * dx + dy * dy
            d = np.sqrt(d2)
            h00, h01, h10, h11 = dx / d, dy / d, -dy / d2, dx / d2
            b00 = -(h00 * gl00 + h01 * gl10)
            b01 = -(h00 * gl01 + h01 * gl11)
            b10 = -(h10 * gl00 + h11 * gl10) - gl20
            b11 = -(h10 * gl01 + h11 * gl11) - gl21
            v0 = (r - d) - (b00 * mu0 + b01 * mu1)
            v1 = _wrap(b - (np.arctan2(dy, dx) - pm)) - (b10 * mu0 + b11 * mu1)
            # Sf = Hf Pf Hf' + R, symmetrised
            t00 = pxx * h00 + pxy * h01
            t01 = pxx * h10 + pxy * h11
            t10 = pxy * h00 + pyy * h01
            t11 = pxy * h10 + pyy * h11
            f00 = h00 * t00 + h01 * t10 + R[0, 0]
            f01 = 0.5 * ((h00 * t01 + h01 * t11 + R[0, 1]) + (h10 * t00 + h11 * t10 + R[1, 0]))
            f11 = h10 * t01 + h11 * t11 + R[1, 1]
            # T = Sig B', S = B T + Sf
            q00 = g00 * b00 + g01 * b01
            q01 = g00 * b10 + g01 * b11
            q10 = g01 * b00 + g11 * b01
            q11 = g01 * b10 + g11 * b11
            s00 = b00 * q00 + b01 * q10 + f00
            s01 = 0.5 * ((b00 * q01 + b01 * q11 + f01) + (b10 * q00 + b11 * q10 + f01))
            s11 = b10 * q01 + b11 * q11 + f11
            u00 = np.sqrt(s00)
            u01 = s01 / u00
            u11 = np.sqrt(s11 - u01 * u01)
            c00, c01, c11 = 1.0 / u00, -u01 / (u00 * u11), 1.0 / u11
            w00 = q00 * c00
            w01 = q00 * c01 + q01 * c11
            w10 = q10 * c00
            w11 = q10 * c01 + q11 * c11
            y0 = c00 * v0
            y1 = c01 * v0 + c11 * v1
            mu0 = mu0 + (w00 * y0 + w01 * y1)
            mu1 = mu1 + (w10 * y0 + w11 * y1)
            g00 = g00 - (w00 * w00 + w01 * w01)
            g01 = g01 - (w00 * w10 + w01 * w11)
            g11 = g11 - (w10 * w10 + w11 * w11)
            logw = logw - 0.5 * (y0 * y0 + y1 * y1) - np.log(u00 * u11) - math.log(2 * math.pi)
        # sample w ~ N(mu, Sig), the control, the pose
        e1, e2 = normals2(self.gids, self.step, STREAM_PREDICT, self.seed)
        l00 = np.sqrt(g00)
        l10 = g01 / l00
        l11 = np.sqrt(g11 - l10 * l10)
        w0 = mu0 + l00 * e1
        w1 = mu1 + l10 * e1 + l11 * e2
        Vn = V + lq00 * w0
        Gn = G + (lq10 * w0 + lq11 * w1)
        self.pose = np.stack([x + Vn * dt * np.cos(Gn + phi), y + Vn * dt * np.sin(Gn + phi),
                              _wrap(phi + Vn * dt * np.sin(Gn) / wheelbase)])
        self.step += 1
        # landmark updates / first sightings from the sampled pose; the weights stay as computed above
        self.update_known(z, ids, R)
        self.logw = logw

    # N4: unknown correspondences -----------------------------------------------------------------
    def clear_landmarks(self):
        """Every slot of every particle unused: Pxx = -1 is the "no landmark here" mark."""
        self.lm[:] = 0.0
        self.lm[:, 2, :] = -1.0
        self.seen[:] = False

    def associate_unknown(self, z, R, gate1, gate2):
        """Per-particle gated nearest neighbour over the particle's OWN landmarks: the rule of ``associate``
        (src/data-association.jl:1-51, order-independent form of SURVEY 3.2) with ``compute_association``
        (:53-63) restricted to the landmark's 2 x 2 block.  Returns assoc[m, n]: slot >= 0 matched, -1 new,
        -2 dropped.  All observations are associated against the map as it is BEFORE this step's updates."""
        z = np.asarray(z, dtype=np.float64).reshape(2, -1)
        R = np.asarray(R, dtype=np.float64)
        m = z.shape[1]
        x, y, phi = self.pose
        best_nd = np.full((m, self.n), np.inf)
        best_l = np.full((m, self.n), -1, dtype=np.int64)
        near = np.zeros((m, self.n), dtype=bool)
        for l in range(self.nl):
            lx, ly, pxx, pxy, pyy = self.lm[l]
            ok = pxx >= 0.0
            with np.errstate(all="ignore"):
                dx, dy = lx - x, ly - y
                d2 = dx * dx + dy * dy
                d = np.sqrt(d2)
                zp1 = np.arctan2(dy, dx) - phi
                h00, h01, h10, h11 = dx / d, dy / d, -dy / d2, dx / d2          # src/common.jl:162
                t00 = pxx * h00 + pxy * h01
                t01 = pxx * h10 + pxy * h11
                t10 = pxy * h00 + pyy * h01
                t11 = pxy * h10 + pyy * h11
                s00 = h00 * t00 + h01 * t10 + R[0, 0]                           # S = Hf Pf Hf' + R (:59), not symmetrised
                s01 = h00 * t01 + h01 * t11 + R[0, 1]
                s10 = h10 * t00 + h11 * t10 + R[1, 0]
                s11 = h10 * t01 + h11 * t11 + R[1, 1]
                det = s00 * s11 - s01 * s10
                rdet = 1.0 / det
                qa, qb, qc = s11 * rdet, -(s01 + s10) * rdet, s00 * rdet
                logdet = np.log(det)
                for i in range(m):
                    v0 = z[0, i] - d
                    v1 = _wrap(z[1, i] - zp1)                                   # :57
                    nis = qa * v0 * v0 + qb * v0 * v1 + qc * v1 * v1            # :60
                    nd = nis + logdet                                           # :61
                    better = ok & (nis < gate1) & (nd < best_nd[i])             # strict: the lowest slot wins a tie
                    best_nd[i] = np.where(better, nd, best_nd[i])
                    best_l[i] = np.where(better, l, best_l[i])
                    near[i] |= ok & (nis <= gate2)
        return np.where(best_l >= 0, best_l, np.where(near, -2, -1))

    def update_unknown(self, z, R, gate1, gate2):
        """associate_unknown, then in observation order: matched -> the 2 x 2 update of update_known on that slot
        (a second observation of the same landmark sees the first one's update); new -> the initialisation of
        update_known in the particle's lowest unused slot (none left: the observation is dropped)."""
        z = np.asarray(z, dtype=np.float64).reshape(2, -1)
        assoc = self.associate_unknown(z, R, gate1, gate2)
        saved_seen = self.seen.copy()
        pidx = np.arange(self.n)
        for i in range(z.shape[1]):
            a = assoc[i].copy()
            newp = a == -1
            if newp.any():
                free = self.lm[:, 2, :] < 0.0                                    # [L, n]
                first_free = np.argmax(free, axis=0)
                has_free = free[first_free, pidx]
                a = np.where(newp, np.where(has_free, first_free, -2), a)
            for l in range(self.nl):
                sel = a == l
                if not sel.any():
                    continue
                is_new = sel & newp
                for mask, seen_flag in ((sel & ~newp, True), (is_new, False)):
                    if not mask.any():
                        continue
                    sub = OraclePF.__new__(OraclePF)                             # the 2 x 2 arithmetic of update_known on a view
                    sub.pose = self.pose[:, mask]
                    sub.lm = self.lm[:, :, mask].copy()
                    sub.logw = self.logw[mask].copy()
                    sub.seen = np.zeros(self.nl, dtype=bool)
                    sub.seen[l] = seen_flag
                    OraclePF.update_known(sub, z[:, i:i + 1], [l + 1], R)
                    self.lm[:, :, mask] = sub.lm
                    self.logw[mask] = sub.logw
        self.seen = saved_seen
        return assoc

    # F4 ------------------------------------------------------------------------------------------
    def weight_stats(self):
        """Local (max logw, sum exp(logw-max), sum exp(2(logw-max)))."""
        m = float(self.logw.max())
        e = np.exp(self.logw - m)
        return m, float(e.sum()), float((e * e).sum())

    @staticmethod
    def combine_stats(stats):
        """Fold per-rank stats into global (max, sum, sum2) -- what the all-reduce computes."""
        gm = max(s[0] for s in stats)
        gs = sum(s[1] * math.exp(s[0] - gm) for s in stats)
        gs2 = sum(s[2] * math.exp(2 * (s[0] - gm)) for s in stats)
        return gm, gs, gs2

    def normalize(self, gmax, gsum):
        self.logw = self.logw - (gmax + math.log(gsum))

    @staticmethod
    def neff(gsum, gsum2):
        return gsum * gsum / gsum2

    @staticmethod
    def ancestors(logw_all, u0):
        """Systematic resampling over the GLOBAL normalised weights: ancestor of slot i is the first j
        with cdf[j] >= (i + u0) / Np."""
        w = np.exp(np.asarray(logw_all, dtype=np.float64) - np.max(logw_all))
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        n = len(w)
        targets = (np.arange(n) + u0) / n
        return np.minimum(np.searchsorted(cdf, targets, side="left"), n - 1).astype(np.int64)

    def record_of(self, local_idx):
        """(3 + 5*nl, cnt) block of particle records (pose rows, then landmark rows l*5+c)."""
        local_idx = np.asarray(local_idx, dtype=np.int64)
        return np.vstack([self.pose[:, local_idx], self.lm[:, :, local_idx].reshape(self.nl * 5, -1)])

    def resample_apply(self, anc_local, remote_ids=None, remote_records=None):
        """anc_local[i] = GLOBAL ancestor id of local slot i; records of ancestors that live on other
        ranks arrive in remote_records (columns in the order of the sorted remote_ids)."""
        anc_local = np.asarray(anc_local, dtype=np.int64)
        is_local = (anc_local >= self.first) & (anc_local < self.first + self.n)
        rec = np.empty((3 + 5 * self.nl, self.n))
        rec[:, is_local] = self.record_of(anc_local[is_local] - self.first)
        if (~is_local).any():
            pos = np.searchsorted(np.asarray(remote_ids), anc_local[~is_local])
            rec[:, ~is_local] = np.asarray(remote_records)[:, pos]
        self.pose = rec[0:3].copy()
        self.lm = rec[3:].reshape(self.nl, 5, self.n).copy()
        self.logw = np.full(self.n, -math.log(self.n_global))
        self.resamples += 1

    def mean_pose_sums(self):
        """Local weighted sums (sum w x, sum w y, sum w sin phi, sum w cos phi) with w = exp(logw)."""
        w = np.exp(self.logw)
        return np.array([np.sum(w * self.pose[0]), np.sum(w * self.pose[1]),
                         np.sum(w * np.sin(self.pose[2])), np.sum(w * np.cos(self.pose[2]))])
