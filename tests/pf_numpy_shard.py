"""TEST-ONLY shard: the FastSLAM driver's shard protocol implemented with the CPU oracle, so the
distributed host logic (slam.jl_amd/pf.py) can be exercised over gloo without a GPU.  It lives in
tests/ on purpose: the product package ships only the HIP shard."""
import math

import numpy as np
import torch

from oracle.pf_ref import OraclePF


class NumpyShard:
    def __init__(self, n_local, max_landmarks, seed, first=0, n_global=None):
        self.o = OraclePF(n_local, max_landmarks, seed, first_id=first, n_global=n_global)
        self.n, self.first, self.n_global = self.o.n, self.o.first, self.o.n_global
        self.nl, self.seed = self.o.nl, self.o.seed
        self.rows = 3 + 5 * self.nl

    def set_pose(self, pose): self.o.set_pose(pose)
    def init_landmarks(self, xy, var, jit): self.o.init_landmarks(xy, var, jit)
    def predict(self, V, G, w, Q, dt): self.o.predict(V, G, w, Q, dt)
    def update_known(self, z, ids, R): self.o.update_known(z, ids, R)
    def clear_landmarks(self): self.o.clear_landmarks()
    def update_unknown(self, z, R, gate1, gate2, want_assoc=False): return self.o.update_unknown(z, R, gate1, gate2)
    def weight_stats(self): return self.o.weight_stats()

    def step_proposal(self, V, G, w, Q, dt, z, ids, R):
        self.o.step_proposal(V, G, w, Q, dt, z, ids, R)
        return self.o.weight_stats()

    def normalize(self, gmax, gsum): self.o.normalize(gmax, gsum)
    def mean_pose_sums(self): return self.o.mean_pose_sums()
    def logw_tensor(self): return torch.from_numpy(self.o.logw.copy())

    def ancestors(self, logw_all, gmax, u0):
        anc = OraclePF.ancestors(logw_all.numpy(), u0)
        return torch.from_numpy(anc[self.first:self.first + self.n].astype(np.int32))

    def ancestors_all(self, logw_all, gmax, u0):
        return torch.from_numpy(OraclePF.ancestors(logw_all.numpy(), u0).astype(np.int32))

    def pack(self, local_idx):
        return torch.from_numpy(self.o.record_of(local_idx.numpy().astype(np.int64)).copy())

    def resample_apply(self, anc, remote_ids, remote_records):
        rid = None if remote_ids is None else remote_ids.numpy().astype(np.int64)
        rec = None if remote_records is None else remote_records.numpy()
        self.o.resample_apply(anc.numpy().astype(np.int64), rid, rec)

    # ---- the auto-mode protocol of the library (slam_pf_step_auto / flush / halt_info / resume), restated on the CPU so
    # that FastSLAM.step_async / flush / _resolve_halt run under gloo without a GPU.  "Enqueued" steps run at once (so the
    # host notices a halt immediately and nothing needs replaying); the ranks' scalars travel through the same shared
    # page the GPUs use (two parities, sequence number last). ----
    def attach_exchange(self, rank, world, page):
        self._x = (int(rank), int(world), page)

    def set_resample_count(self, count):
        self._count = int(count)

    def _auto_init(self):
        if not hasattr(self, "_seq"):
            self._seq, self._halted, self._last = 0, False, (float(self.n_global), False)
            self._count = getattr(self, "_count", 0)
            self._x = getattr(self, "_x", (0, 1, None))

    def step_auto(self, V, G, w, Q, dt, z, ids, R, neff_frac=0.75, force=None, proposal=False, prepared=None):
        from oracle.pf_ref import uniform1
        self._auto_init()
        if self._halted:
            return True                                      # SLAM_PF_HALTED: nothing enqueued by this call
        if proposal:
            self.o.step_proposal(V, G, w, Q, dt, z, ids, R)
        else:
            self.o.predict(V, G, w, Q, dt)
            self.o.update_known(z, ids, R)
        self._seq += 1
        m, s1, s2 = self.o.weight_stats()
        rank, world, page = self._x
        if world > 1:
            pg = page.reshape(2, world, 8)[self._seq & 1]
            pg[rank, 0:3] = (m, s1, s2)
            pg[rank, 3] = self._seq
            spins = 0
            while not (pg[:, 3] == self._seq).all():
                spins += 1
                assert spins < 200_000_000, "a rank did not arrive"
            table = pg[:, 0:3].copy()
            gm = float(table[:, 0].max())
            gs1 = sum(float(r[1]) * math.exp(float(r[0]) - gm) for r in table)
            gs2 = sum(float(r[2]) * math.exp(float(r[0]) - gm) ** 2 for r in table)
        else:
            gm, gs1, gs2 = m, s1, s2
        neff = gs1 * gs1 / gs2
        want = (neff < neff_frac * self.n_global) if force is None else bool(force)
        self.o.normalize(gm, gs1)                            # (the library defers the shift; the weights are the same)
        self._last = (neff, want)
        if want and world == 1:                              # the device resamples by itself
            self.o.resample_apply(OraclePF.ancestors(self.o.logw, uniform1(self._count, 2, self.seed)))
            self._count += 1
        elif want:
            self._halted = True
            self._halt_gmax = float(gm - (gm + math.log(gs1)))
        return False

    def flush(self):
        self._auto_init()
        if self._halted:
            return None
        return self._last[0], self._last[1], self._count, self._seq

    def halt_info(self):
        return self._halt_gmax, self._count

    def resume(self, resamplings):
        self._count = int(resamplings)
        self._halted = False
