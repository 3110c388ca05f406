"""TEST-ONLY shard: the FastSLAM driver's shard protocol implemented with the CPU oracle, so the
distributed host logic (slam.jl_amd/pf.py) can be exercised over gloo without a GPU.  It lives in
tests/ on purpose: the product package ships only the HIP shard."""
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
