"""FastSLAM-1.0 (known correspondences) on the GPU: ``PFSlamState`` and its sharded driver.

The reference only declares the types ``Particle`` / ``PFSlamState`` (src/common.jl:14-20,31-34)
and no particle-filter code (README.md:6); the algorithm is SURVEY.md 8a rows F1-F4.

Two layers:

* :class:`PFShard` -- one process's slice of the particles, a thin wrapper of the ``slam_pf_*``
  C ABI (HIP kernels, SoA state resident in HBM).  No CPU fallback.
* :class:`FastSLAM` -- the host logic that is the same for 1 and for G GPUs: per step an
  all-reduce of three scalars (max log-weight, sum w, sum w^2) for normalisation and Neff; on a
  resampling step an all-gather of the log-weights (the RCCL collective BASELINE.json names),
  the same global systematic-resampling table on every rank, and an exchange of the particle
  records whose ancestor lives on another rank.  It talks to the shard through a small protocol
  (the methods of :class:`PFShard`) and to the other ranks through ``torch.distributed``
  (``nccl`` = RCCL on the GPUs; the multi-process CPU tests drive the same logic over ``gloo``
  with a NumPy shard that lives in the test-suite, never here).

Random numbers are Philox4x32-10 keyed by (seed, step, global particle id): a run gives the same
particles whatever the number of ranks.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys

import numpy as np

from ._lib import SLAM_F32, SLAM_F64, SLAM_PF_HALTED, SLAM_PF_PEER_BLOB_BYTES, check, lib
from .ekf import _obs, _small, _ptr

__all__ = ["PFShard", "PFSlamState", "FastSLAM", "philox_uniform", "small", "shared_page", "attach_local_peers"]

_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
STREAM_RESAMPLE = 2


def philox_uniform(step: int, stream: int, seed: int) -> float:
    """One U(0,1) from Philox4x32-10 with counter (0, 0, step, stream): the systematic-resampling
    offset every rank derives for itself (host scalar; same generator as the kernels)."""
    c = [0, 0, step & 0xFFFFFFFF, stream & 0xFFFFFFFF]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return ((c[0] >> 8) + 0.5) / 16777216.0


class _Small:
    """A 2 x 2 matrix already in the library's column-major double[4] form, with its pointer (see :func:`small`)."""
    __slots__ = ("a", "ptr")

    def __init__(self, M):
        self.a = _small(M)
        self.ptr = _ptr(self.a)


def small(M):
    """Convert a 2 x 2 matrix once for repeated ``step_auto`` calls."""
    return M if isinstance(M, _Small) else _Small(M)


class PFShard:
    """Global particle ids [first, first + n) of an n_global-particle filter on one GPU."""

    def __init__(self, n_local, max_landmarks, seed, dtype="f32", first=0, n_global=None, device=0):
        import torch
        self.n = int(n_local)
        self.first = int(first)
        self.n_global = int(n_global if n_global is not None else n_local)
        self.nl = int(max_landmarks)
        self.seed = int(seed)
        self.np_dtype = np.float32 if dtype == "f32" else np.float64
        self.torch_dtype = torch.float32 if dtype == "f32" else torch.float64
        self.device = torch.device("cuda", int(device))
        self._h = C.c_void_p()
        check(lib.slam_pf_create(C.byref(self._h), SLAM_F32 if dtype == "f32" else SLAM_F64, self.n, self.n_global,
                                 self.first, self.nl, int(device), C.c_uint64(self.seed)))
        rows = C.c_int()
        check(lib.slam_pf_record_rows(self._h, C.byref(rows)))
        self.rows = rows.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.slam_pf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- filter operations ---------------------------------------------------------------------
    def set_pose(self, pose):
        p = np.ascontiguousarray(np.asarray(pose, dtype=np.float64).reshape(3))
        check(lib.slam_pf_set_pose(self._h, _ptr(p)))

    def init_landmarks(self, lm_xy, var, jitter_sigma):
        xy = np.ascontiguousarray(np.asarray(lm_xy, dtype=np.float64).reshape(-1, 2))
        check(lib.slam_pf_init_landmarks(self._h, _ptr(xy), xy.shape[0], float(var), float(jitter_sigma)))

    def predict(self, V, G, wheelbase, Q, dt):
        q = _small(Q)
        check(lib.slam_pf_predict(self._h, float(V), float(G), float(wheelbase), _ptr(q), float(dt)))

    def update_known(self, z, ids, R):
        zp = _obs(z)
        idv = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).reshape(-1))
        if idv.shape[0] != zp.shape[0]:
            raise ValueError("ids and z disagree on the number of observations")
        if zp.shape[0] == 0:
            return
        r = _small(R)
        check(lib.slam_pf_update_known(self._h, _ptr(zp), _ptr(idv, C.c_int32), zp.shape[0], _ptr(r)))

    def clear_landmarks(self):
        """Every landmark slot of every particle unused (the start of an unknown-correspondence run)."""
        check(lib.slam_pf_clear_landmarks(self._h))

    def update_unknown(self, z, R, gate1, gate2, want_assoc=False):
        """Unknown correspondences: per-particle gated nearest-neighbour association against the
        particle's own landmarks, then updates / new landmarks.  ``want_assoc``: return the decisions as an
        int32 torch tensor [m, n] on the device (slot >= 0 matched, -1 new, -2 dropped)."""
        zp = _obs(z)
        m = zp.shape[0]
        if m == 0:
            return None
        r = _small(R)
        assoc = None
        ptr = None
        if want_assoc:
            import torch
            assoc = torch.empty((m, self.n), dtype=torch.int32, device=self.device)
            ptr = C.c_void_p(assoc.data_ptr())
        check(lib.slam_pf_update_unknown(self._h, _ptr(zp), m, _ptr(r), float(gate1), float(gate2), ptr))
        if want_assoc:
            self.sync()
        return assoc

    def weight_stats(self):
        out = np.empty(3)
        check(lib.slam_pf_weight_stats(self._h, _ptr(out)))
        return float(out[0]), float(out[1]), float(out[2])

    def step_fused(self, V, G, wheelbase, Q, dt, z, ids, R):
        """predict + update_known + weight_stats as ONE sweep over the particles (slam_pf_step): the same
        particles bit for bit, the same three statistics."""
        zp = _obs(z)
        idv = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).reshape(-1))
        if idv.shape[0] != zp.shape[0]:
            raise ValueError("ids and z disagree on the number of observations")
        q, r = _small(Q), _small(R)
        out = np.empty(3)
        check(lib.slam_pf_step(self._h, float(V), float(G), float(wheelbase), _ptr(q), float(dt), _ptr(zp),
                               _ptr(idv, C.c_int32), zp.shape[0], _ptr(r), _ptr(out)))
        return float(out[0]), float(out[1]), float(out[2])

    def step_proposal(self, V, G, wheelbase, Q, dt, z, ids, R):
        """The FastSLAM-2.0 step (slam_pf_step_proposal): as step_fused, but the pose is drawn from the
        proposal that already knows this step's observations.  Returns the same three statistics."""
        zp = _obs(z)
        idv = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).reshape(-1))
        if idv.shape[0] != zp.shape[0]:
            raise ValueError("ids and z disagree on the number of observations")
        q, r = _small(Q), _small(R)
        out = np.empty(3)
        check(lib.slam_pf_step_proposal(self._h, float(V), float(G), float(wheelbase), _ptr(q), float(dt), _ptr(zp),
                                        _ptr(idv, C.c_int32), zp.shape[0], _ptr(r), _ptr(out)))
        return float(out[0]), float(out[1]), float(out[2])

    def step_fused_normalized(self, V, G, wheelbase, Q, dt, z, ids, R):
        """step_fused + normalize with this shard's own statistics (the whole filter lives here): returns
        (Neff, max normalised log-weight)."""
        zp = _obs(z)
        idv = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).reshape(-1))
        if idv.shape[0] != zp.shape[0]:
            raise ValueError("ids and z disagree on the number of observations")
        q, r = _small(Q), _small(R)
        out = np.empty(4)
        check(lib.slam_pf_step_normalized(self._h, float(V), float(G), float(wheelbase), _ptr(q), float(dt), _ptr(zp),
                                          _ptr(idv, C.c_int32), zp.shape[0], _ptr(r), _ptr(out)))
        T = self.np_dtype                          # (neff, largest log-weight after the shift, exactly as stored)
        return float(out[3]), float(T(out[0]) - T(out[0] + math.log(out[1])))

    def normalize(self, gmax, gsum):
        check(lib.slam_pf_normalize(self._h, float(gmax), float(gsum)))

    def mean_pose_sums(self):
        out = np.empty(4)
        check(lib.slam_pf_mean_pose_sums(self._h, _ptr(out)))
        return out

    # -- the step without the host in the loop (slam_pf_step_auto) -----------------------------------
    @staticmethod
    def prepare_obs(z, ids):
        """Observations in the layout the library takes, converted once, with their ctypes pointers (a timed loop calls
        step_auto with these: a step is then one foreign call)."""
        zp = _obs(z)
        idv = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).reshape(-1))
        if idv.shape[0] != zp.shape[0]:
            raise ValueError("ids and z disagree on the number of observations")
        return zp, idv, _ptr(zp), _ptr(idv, C.c_int32)

    def step_auto(self, V, G, wheelbase, Q, dt, z, ids, R, neff_frac=0.75, force=None, proposal=False, prepared=None):
        """One whole filter step ENQUEUED (slam_pf_step_auto): statistics, normalisation, Neff, the decision to resample
        and -- filter wholly on this shard -- the resampling itself stay on the device.  ``force``: None = Neff rule,
        False / True = never / always.  Returns False, or True when the library reports SLAM_PF_HALTED (a queued step
        of a sharded filter wants a resampling; nothing was enqueued by this call)."""
        zp, _idv, pz, pi = prepared if prepared is not None else self.prepare_obs(z, ids)
        pq = Q.ptr if isinstance(Q, _Small) else _ptr(_small(Q))
        pr = R.ptr if isinstance(R, _Small) else _ptr(_small(R))
        rc = lib.slam_pf_step_auto(self._h, V, G, wheelbase, pq, dt, pz, pi, zp.shape[0], pr, neff_frac,
                                   -1 if force is None else int(bool(force)), 1 if proposal else 0)
        if rc == SLAM_PF_HALTED:
            return True
        if rc:
            check(rc)
        return False

    @staticmethod
    def prepare_batch(controls, obs, force=None):
        """K steps in the layout slam_pf_step_auto_batch takes, converted once: ``controls`` K x (V, G); ``obs`` K pairs
        (z 2 x m_k, ids m_k); ``force`` None (the Neff rule at every step), one value for all steps or K values (None / False /
        True each).  Returns an opaque tuple for ``step_auto_batch``."""
        K = len(obs)
        vg = np.ascontiguousarray(np.asarray(controls, dtype=np.float64).reshape(K, 2))
        ms = np.array([np.asarray(i).reshape(-1).shape[0] for _, i in obs], dtype=np.int32)
        stride = max(1, int(ms.max()) if K else 1)
        zz = np.zeros((K, stride, 2))
        ii = np.zeros((K, stride), dtype=np.int32)
        for k, (z, ids) in enumerate(obs):
            if ms[k]:
                zz[k, :ms[k]] = _obs(z)
                ii[k, :ms[k]] = np.asarray(ids, dtype=np.int32).reshape(-1)
        if force is None or isinstance(force, (bool, np.bool_)):
            force = [force] * K
        ff = np.array([-1 if f is None else int(bool(f)) for f in force], dtype=np.int32)
        if ff.shape[0] != K:
            raise ValueError("force: one value per step")
        return K, vg, zz, ii, ms, stride, ff

    def step_auto_batch(self, batch, wheelbase, Q, dt, R, neff_frac=0.75, proposal=False, persistent=False, start=0):
        """K filter steps ENQUEUED by one call (slam_pf_step_auto_batch; ``batch`` from prepare_batch): the same filter as
        K step_auto calls, bit for bit.  ``persistent=True``: runs of steps that cannot resample may go as persistent launches of
        up to 16 steps -- the caller vouches that nothing else keeps the device's compute units busy meanwhile (the grid takes them
        whole and its workgroups wait for each other).  Returns the number of steps taken from ``start`` on: all of them, or fewer when the library reports SLAM_PF_HALTED (sharded
        halting flow: resolve the halt and call again with ``start`` advanced)."""
        K, vg, zz, ii, ms, stride, ff = batch
        pq = Q.ptr if isinstance(Q, _Small) else _ptr(_small(Q))
        pr = R.ptr if isinstance(R, _Small) else _ptr(_small(R))
        took = C.c_int(0)
        rc = lib.slam_pf_step_auto_batch(self._h, K - start, _ptr(vg[start:]), wheelbase, pq, dt, _ptr(zz[start:]),
                                         _ptr(ii[start:], C.c_int32), _ptr(ms[start:], C.c_int32), stride, pr, neff_frac,
                                         _ptr(ff[start:], C.c_int32), 1 if proposal else 0, 2 if persistent else 0, C.byref(took))
        if rc and rc != SLAM_PF_HALTED:
            check(rc)
        return int(took.value)

    def flush(self):
        """Wait for the queued steps: (Neff of the last step, it resampled?, resamplings so far, steps so far), or None
        when the library reports SLAM_PF_HALTED."""
        out = np.empty(4)
        rc = lib.slam_pf_flush(self._h, _ptr(out))
        if rc == SLAM_PF_HALTED:
            return None
        check(rc)
        return float(out[0]), bool(out[1]), int(out[2]), int(out[3])

    def debug_stamps(self):
        """100 MHz stamps of the last auto step (slam_pf_debug_stamps), as microseconds since the kernel's start."""
        out = (C.c_uint64 * 8)()
        check(lib.slam_pf_debug_stamps(self._h, out))
        return [(int(v) - int(out[0])) / 100.0 for v in out[:8]]

    def halt_info(self):
        out = np.empty(2)
        check(lib.slam_pf_halt_info(self._h, _ptr(out)))
        return float(out[0]), int(out[1])

    def resume(self, resamplings):
        check(lib.slam_pf_resume(self._h, int(resamplings)))

    def resample_count(self):
        out = C.c_int64()
        check(lib.slam_pf_resample_count(self._h, C.byref(out)))
        return int(out.value)

    def set_resample_count(self, count):
        check(lib.slam_pf_set_resample_count(self._h, int(count)))

    def attach_exchange(self, rank, world, page):
        """``page``: a float64 NumPy array over host memory that every rank has mapped (>= 2 * world * 8 values)."""
        assert page.dtype == np.float64 and page.flags.c_contiguous and page.size >= 2 * world * 8
        self._xchg_page = page                       # keeps the mapping alive
        check(lib.slam_pf_attach_exchange(self._h, int(rank), int(world), C.c_void_p(page.ctypes.data), page.nbytes))

    # -- sharding behind the C ABI: peers (slam_pf_export_peer / slam_pf_attach_peers) ------------------------
    def export_peer(self):
        """This shard's peer blob (bytes): IPC handles of its buffers and inbox; every rank attaches all ranks' blobs."""
        buf = C.create_string_buffer(SLAM_PF_PEER_BLOB_BYTES)
        check(lib.slam_pf_export_peer(self._h, buf))
        return bytes(buf.raw)

    def attach_peers(self, rank, world, blobs):
        """``blobs``: the ``world`` peer blobs in rank order.  From here on step_auto resamples the sharded filter on
        the device (no SLAM_PF_HALTED); calls that need plain maps (download with landmarks, pack, legacy sweeps) are
        collective."""
        assert len(blobs) == world and all(len(b) == SLAM_PF_PEER_BLOB_BYTES for b in blobs)
        check(lib.slam_pf_attach_peers(self._h, int(rank), int(world), C.c_char_p(b"".join(blobs))))

    def detach_peers(self):
        check(lib.slam_pf_detach_peers(self._h))

    def peer_selftest(self, timeout_ms=3000):
        """Collective: True when every attached peer's inbox write arrived here within the time-out."""
        return lib.slam_pf_peer_selftest(self._h, int(timeout_ms)) == 0

    def comm_info(self):
        """{world, peers attached?, SLAM_PF_HALTED returns so far, resamplings so far}."""
        out = (C.c_int64 * 4)()
        check(lib.slam_pf_comm_info(self._h, out))
        return dict(world=int(out[0]), peers=bool(out[1]), halts=int(out[2]), resamples=int(out[3]))

    def resample_if_needed(self, neff_frac=0.75):
        """slam_pf_resample: normalise and resample if Neff < neff_frac * n (filter wholly on this shard)."""
        out = C.c_int()
        check(lib.slam_pf_resample(self._h, float(neff_frac), C.byref(out)))
        return bool(out.value)

    def mean_pose(self):
        out = np.empty(3)
        check(lib.slam_pf_get_mean_pose(self._h, _ptr(out)))
        return out

    def weights(self):
        out = np.empty(self.n)
        check(lib.slam_pf_get_weights(self._h, _ptr(out)))
        return out

    # -- resampling pieces (torch tensors on this shard's device) ---------------------------------
    def logw_tensor(self):
        import torch
        t = torch.empty(self.n, dtype=self.torch_dtype, device=self.device)
        check(lib.slam_pf_copy_logw(self._h, C.c_void_p(t.data_ptr())))
        return t

    def ancestors(self, logw_all, gmax, u0):
        import torch
        assert logw_all.is_cuda and logw_all.dtype == self.torch_dtype and logw_all.numel() == self.n_global
        torch.cuda.synchronize(self.device)            # logw_all was produced on torch's stream
        anc = torch.empty(self.n, dtype=torch.int32, device=self.device)
        check(lib.slam_pf_ancestors(self._h, C.c_void_p(logw_all.data_ptr()), float(gmax), float(u0),
                                    C.c_void_p(anc.data_ptr())))
        return anc

    def resample_local(self, gmax, u0):
        """Resampling of a filter that lives wholly on this shard, one library call (slam_pf_resample_local)."""
        check(lib.slam_pf_resample_local(self._h, float(gmax), float(u0)))

    def ancestors_all(self, logw_all, gmax, u0):
        """Global ancestor id of EVERY slot of the filter (n_global int32): identical on every rank."""
        import torch
        assert logw_all.is_cuda and logw_all.dtype == self.torch_dtype and logw_all.numel() == self.n_global
        torch.cuda.synchronize(self.device)            # logw_all was produced on torch's stream
        anc = torch.empty(self.n_global, dtype=torch.int32, device=self.device)
        check(lib.slam_pf_ancestors_all(self._h, C.c_void_p(logw_all.data_ptr()), float(gmax), float(u0),
                                        C.c_void_p(anc.data_ptr())))
        return anc

    def pack(self, local_idx):
        import torch
        idx = local_idx.to(device=self.device, dtype=torch.int32).contiguous()
        rec = torch.empty((self.rows, idx.numel()), dtype=self.torch_dtype, device=self.device)
        if idx.numel():
            torch.cuda.synchronize(self.device)
            check(lib.slam_pf_pack(self._h, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(rec.data_ptr())))
        return rec

    def resample_apply(self, anc, remote_ids, remote_records):
        import torch
        torch.cuda.synchronize(self.device)
        nrem = 0 if remote_ids is None else int(remote_ids.numel())
        if nrem:
            ids = remote_ids.to(device=self.device, dtype=torch.int32).contiguous()
            rec = remote_records.to(device=self.device, dtype=self.torch_dtype).contiguous()
            assert rec.shape == (self.rows, nrem)
            check(lib.slam_pf_resample_apply(self._h, C.c_void_p(anc.data_ptr()), C.c_void_p(ids.data_ptr()), nrem,
                                             C.c_void_p(rec.data_ptr())))
        else:
            check(lib.slam_pf_resample_apply(self._h, C.c_void_p(anc.data_ptr()), None, 0, None))

    # -- inspection ---------------------------------------------------------------------------------
    def download(self, landmarks=True):
        """(pose [3, n], logw [n], lm [nl, 5, n] or None) as NumPy arrays in the shard's dtype."""
        pose = np.empty((3, self.n), dtype=self.np_dtype)
        logw = np.empty(self.n, dtype=self.np_dtype)
        lm = np.empty((self.nl, 5, self.n), dtype=self.np_dtype) if landmarks else None
        check(lib.slam_pf_download(self._h, pose.ctypes.data, logw.ctypes.data, lm.ctypes.data if landmarks else None))
        return pose, logw, lm

    def sync(self):
        check(lib.slam_pf_sync(self._h))


class _SingleProcess:
    """The world of one rank."""
    rank, world = 0, 1

    def allreduce_max(self, v):
        return v

    def allreduce_sum(self, vec):
        return vec

    def all_gather(self, t, n_global):
        return t

    def all_gather_scalars(self, vec):
        return [list(vec)]

    def all_to_all_v(self, send, send_counts, recv_counts):
        return send


class ShmScalars:
    """All-gather of a few float64 per rank through a shared-memory page, for ranks of ONE node.

    The per-step exchange of a sharded filter is three scalars per rank that the HOST needs (Neff decides about
    resampling): through a device collective that is a host-to-device copy, a collective launch and a read-back,
    ~50 us, against a few microseconds here.  Layout [parity][rank][8 float64], entry 7 = the sequence number of the
    call, written last; a call spins until every rank's entry shows its number.  Two parities: a rank can be at most
    one call ahead of the slowest one (it needs everybody's entry of call s before it can write call s + 1), so
    the page of call s is not overwritten while anybody still reads it."""
    WIDTH = 8

    def __init__(self, dist, rank, world):
        import socket
        import uuid
        hosts = [None] * world
        dist.all_gather_object(hosts, socket.gethostname())
        if len(set(hosts)) != 1 or not os.path.isdir("/dev/shm"):
            raise RuntimeError("ranks are not on one node")
        name = [f"/dev/shm/slamhip-{uuid.uuid4().hex}" if rank == 0 else None]
        dist.broadcast_object_list(name, src=0)
        self.rank, self.world, self.seq = rank, world, 0
        shape = (2, world, self.WIDTH)
        if rank == 0:
            np.zeros(shape, dtype=np.float64).tofile(name[0])
        dist.barrier()
        self.mem = np.memmap(name[0], dtype=np.float64, mode="r+", shape=shape)
        dist.barrier()
        if rank == 0:
            os.unlink(name[0])                       # the mappings keep the page alive; nothing is left behind in /dev/shm

    def all_gather(self, vec):
        assert len(vec) < self.WIDTH
        self.seq += 1
        page = self.mem[self.seq & 1]
        mine = page[self.rank]
        mine[:len(vec)] = vec
        mine[self.WIDTH - 1] = self.seq              # published last (x86 keeps the order of the stores)
        flags = page[:, self.WIDTH - 1]
        spins = 0
        while not (flags == self.seq).all():
            spins += 1
            if spins > 50_000_000:
                raise RuntimeError("shared-memory exchange: a rank did not arrive")
        return page[:, :len(vec)].tolist()


class TorchComm:
    """torch.distributed (nccl = RCCL over xGMI on the GPUs, gloo in the CPU tests).

    With the gloo backend device tensors are staged through the host (gloo's CUDA support is partial): that
    is how several ranks can rehearse the real GPU shards on ONE card; RCCL runs never take that path.
    The per-step scalar exchange goes through shared memory when all ranks share a node (ShmScalars)."""

    def __init__(self, device, shm_scalars=True):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.stage = dist.get_backend() == "gloo"
        self.shm = None
        if shm_scalars and self.world > 1 and os.environ.get("SLAMHIP_SHM_SCALARS", "1") != "0":
            try:
                self.shm = ShmScalars(dist, self.rank, self.world)
            except RuntimeError:
                self.shm = None                      # several nodes: the collective below

    def _in(self, t):
        return t.cpu() if self.stage and t.is_cuda else t

    def _out(self, t, like):
        return t.to(like.device) if self.stage and like.is_cuda else t

    def allreduce_max(self, v):
        import torch
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if self.stage else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_sum(self, vec):
        import torch
        t = torch.tensor(list(vec), dtype=torch.float64, device="cpu" if self.stage else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def all_gather_scalars(self, vec):
        """[world][len(vec)] float64 table, rows in rank order."""
        if self.shm is not None:
            return self.shm.all_gather([float(v) for v in vec])
        import torch
        t = torch.tensor(list(vec), dtype=torch.float64, device="cpu" if self.stage else self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.float64, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.reshape(self.world, -1).tolist()

    def all_gather(self, t, n_global):
        import torch
        src = self._in(t.contiguous())
        out = torch.empty(n_global, dtype=src.dtype, device=src.device)
        self.dist.all_gather_into_tensor(out, src)       # equal slices: rank r owns [r*n, (r+1)*n)
        return self._out(out, t)

    def all_to_all_v(self, send, send_counts, recv_counts):
        """Rows of `send` (dim 0) are grouped by destination rank; returns the rows received, grouped by source."""
        import torch
        src = self._in(send.contiguous())
        out = torch.empty((int(sum(recv_counts)),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        self.dist.all_to_all_single(out, src, [int(c) for c in recv_counts], [int(c) for c in send_counts])
        return self._out(out, send)


class FastSLAM:
    """Host logic of the particle filter, identical for one and for many ranks.

    ``shard`` follows the :class:`PFShard` protocol; ``comm`` is ``None`` (single process) or a
    :class:`TorchComm`.  Equal slices are assumed: rank r owns [r * n, (r + 1) * n).
    """

    def __init__(self, shard, comm=None, neff_frac=0.75):
        self.shard = shard
        self.comm = comm if comm is not None else _SingleProcess()
        self.neff_frac = float(neff_frac)
        self.fused = True                           # use shard.step_fused when the shard offers it
        self.force_exchange = False                 # measurement only: take the multi-rank resampling flow on one rank
        self._gmax_norm = None                      # max log-weight after the last normalize() (None: unknown)
        self.resamples = 0
        self.last_neff = float(shard.n_global)
        assert shard.n * self.comm.world == shard.n_global and shard.first == self.comm.rank * shard.n, (
            "ranks must own equal, contiguous slices in rank order")

    def predict(self, V, G, wheelbase, Q, dt):
        self.shard.predict(V, G, wheelbase, Q, dt)

    def update_known(self, z, ids, R):
        self._gmax_norm = None
        self.shard.update_known(z, ids, R)

    def global_stats(self, local=None):
        """(gmax, sum w, sum w^2) with w = exp(logw - gmax): one all-gather of three scalars per rank.
        ``local``: this shard's (max, sum, sum2) if a fused step has already produced them."""
        lmax, s1, s2 = self.shard.weight_stats() if local is None else local
        # ONE small all-gather of (max, sum, sum2) per rank; every rank then folds the same table in rank order
        # (two dependent all-reduces -- MAX, then SUM of the rescaled sums -- cost two collective latencies per step)
        table = self.comm.all_gather_scalars([lmax, s1, s2])
        gmax = max(row[0] for row in table)
        gs1 = gs2 = 0.0
        for m_r, s1_r, s2_r in table:
            f = math.exp(m_r - gmax)
            gs1 += s1_r * f
            gs2 += s2_r * f * f
        return gmax, gs1, gs2

    def normalize(self, local=None):
        """Normalise the weights, return Neff = 1 / sum(w_normalised^2)."""
        gmax, gs1, gs2 = self.global_stats(local)
        self.shard.normalize(gmax, gs1)
        self.last_neff = gs1 * gs1 / gs2
        # the largest log-weight AFTER the shift, exactly as stored (subtracting one constant in the storage type is
        # monotone, so it is the image of the old maximum): spares resample() a reduction over all weights + a sync
        T = getattr(self.shard, "np_dtype", np.float64)
        self._gmax_norm = float(T(gmax) - T(gmax + math.log(gs1)))
        return self.last_neff

    def resample(self):
        """Systematic resampling over the global weights (call after normalize())."""
        import torch
        sh, comm = self.shard, self.comm
        u0 = philox_uniform(self.resamples, STREAM_RESAMPLE, sh.seed)
        local = getattr(sh, "resample_local", None)
        if comm.world == 1 and not self.force_exchange and local is not None and self._gmax_norm is not None:
            local(self._gmax_norm, u0)                                  # the whole filter on one GPU: one library call
            self._gmax_norm = None
            self._count_resampling()
            return 0
        logw_all = comm.all_gather(sh.logw_tensor(), sh.n_global)      # the all-gather of log-weights
        gmax = self._gmax_norm if self._gmax_norm is not None else float(logw_all.max().item())
        self._gmax_norm = None
        if comm.world == 1 and not self.force_exchange:                 # every ancestor is local: nothing to exchange
            sh.resample_apply(sh.ancestors(logw_all, gmax, u0), None, None)
            self._count_resampling()
            return 0
        # Every rank computes the ancestor of EVERY slot from the same all-gathered weights, so each knows which of its
        # particles every other rank needs: no request round, ONE all-to-all of records.  The table is ascending
        # (systematic resampling), and so is the owner of each slot: the (destination rank, ancestor) pairs come out
        # sorted, unique_consecutive removes the duplicates, and both the send list (my particles, grouped by
        # destination) and the receive list (ascending ids = grouped by source) are slices of that one list.
        n, me, world = sh.n, comm.rank, comm.world
        anc_all = sh.ancestors_all(logw_all, gmax, u0).to(torch.int64)
        slot_rank = torch.arange(sh.n_global, device=anc_all.device, dtype=torch.int64) // n
        # (boolean-mask indexing synchronises: each mask is turned into an index list once)
        remote = torch.nonzero(torch.div(anc_all, n, rounding_mode="floor") != slot_rank).reshape(-1)
        pairs = torch.unique_consecutive(slot_rank[remote] * sh.n_global + anc_all[remote])
        dst, aid = torch.div(pairs, sh.n_global, rounding_mode="floor"), pairs % sh.n_global
        src = torch.div(aid, n, rounding_mode="floor")
        out_idx, in_idx = torch.nonzero(src == me).reshape(-1), torch.nonzero(dst == me).reshape(-1)
        send_ids, need_ids = aid[out_idx], aid[in_idx]
        counts = torch.stack([torch.bincount(dst[out_idx], minlength=world),
                              torch.bincount(src[in_idx], minlength=world)]).tolist()      # split sizes of the all-to-all
        rec = sh.pack((send_ids - sh.first).to(torch.int32))                                # [rows, n_send]
        got = comm.all_to_all_v(rec.t().contiguous(), counts[0], counts[1])                 # [n_need, rows], ascending ids
        anc = anc_all[sh.first:sh.first + n].to(torch.int32).contiguous()
        sh.resample_apply(anc, need_ids.to(torch.int32), got.t().contiguous())
        self._count_resampling()
        return int(need_ids.numel())

    def _count_resampling(self):
        # the count is part of the filter state (it keys the systematic-resampling offset): the library's copy follows
        self.resamples += 1
        setter = getattr(self.shard, "set_resample_count", None)
        if setter is not None:
            setter(self.resamples)

    def step(self, V, G, wheelbase, Q, dt, z, ids, R, force_resample=None, proposal=False):
        """predict + known-id updates + normalise + (Neff-triggered) resample.  Returns (Neff, resampled?).
        ``proposal``: the FastSLAM-2.0 step -- the pose is drawn from the proposal that knows this step's
        observations (shard.step_proposal) instead of from the motion model alone."""
        fused = getattr(self.shard, "step_fused", None) if self.fused else None
        local = getattr(self.shard, "step_fused_normalized", None) if (self.fused and self.comm.world == 1) else None
        self._gmax_norm = None
        if proposal:
            neff = self.normalize(self.shard.step_proposal(V, G, wheelbase, Q, dt, z, ids, R))
        elif local is not None:                       # the whole filter on one GPU: one library call per step
            neff, self._gmax_norm = local(V, G, wheelbase, Q, dt, z, ids, R)
            self.last_neff = neff
        elif fused is not None:                     # one sweep over the particles instead of five launches
            neff = self.normalize(fused(V, G, wheelbase, Q, dt, z, ids, R))
        else:
            self.predict(V, G, wheelbase, Q, dt)
            self.update_known(z, ids, R)
            neff = self.normalize()
        do = force_resample if force_resample is not None else (neff < self.neff_frac * self.shard.n_global)
        if do:
            self.resample()
        return neff, bool(do)

    # -- the same step without the host in the loop --------------------------------------------------------------------
    def step_async(self, V, G, wheelbase, Q, dt, z, ids, R, force_resample=None, proposal=False, prepared=None):
        """``step`` ENQUEUED (shard.step_auto): returns nothing; Neff and the resampling decision stay on the device,
        steps queue back to back, ``flush()`` waits and reports.  On one rank the resampling itself happens on the
        device too.  A sharded filter resamples through the host (all-gather of the log-weights + record exchange): the
        library halts at such a step, skips what was queued behind it, and this method resolves the halt the moment
        the library reports it -- resample(), resume (the skipped steps are enqueued again), carry on."""
        self._gmax_norm = None
        sh = self.shard
        while sh.step_auto(V, G, wheelbase, Q, dt, z, ids, R, neff_frac=self.neff_frac, force=force_resample,
                           proposal=proposal, prepared=prepared):
            self._resolve_halt()

    def step_async_batch(self, batch, wheelbase, Q, dt, R, proposal=False, persistent=False):
        """K ``step_async`` calls as one (shard.step_auto_batch, ``batch`` from PFShard.prepare_batch); ``persistent=True`` allows
        persistent launches of up to 16 steps (see there).  Halts of the sharded halting flow are resolved on the way."""
        self._gmax_norm = None
        sh = self.shard
        k, K = 0, batch[0]
        while k < K:
            took = sh.step_auto_batch(batch, wheelbase, Q, dt, R, neff_frac=self.neff_frac, proposal=proposal,
                                      persistent=persistent, start=k)
            k += took
            if k < K:
                self._resolve_halt()

    def flush(self):
        """Wait for the steps queued by step_async.  Returns (Neff of the last step, it resampled?)."""
        while True:
            r = self.shard.flush()
            if r is None:
                self._resolve_halt()
                continue
            self.last_neff, did, self.resamples = r[0], r[1], r[2]
            return self.last_neff, did

    def _resolve_halt(self):
        sh = self.shard
        self._gmax_norm, self.resamples = sh.halt_info()
        self.resample()                                   # the legacy path: collectives issued from here
        sh.resume(self.resamples)

    def step_unknown(self, V, G, wheelbase, Q, dt, z, R, gate1, gate2, force_resample=None):
        """The filter step with UNKNOWN correspondences: predict, per-particle gated nearest-neighbour
        association + updates / new landmarks, normalise, (Neff-triggered) resample.  Returns (Neff, resampled?).
        Resampling copies whole particle records, unused landmark slots included, so it needs no change."""
        self._gmax_norm = None
        self.predict(V, G, wheelbase, Q, dt)
        self.shard.update_unknown(z, R, gate1, gate2)
        neff = self.normalize()
        do = force_resample if force_resample is not None else (neff < self.neff_frac * self.shard.n_global)
        if do:
            self.resample()
        return neff, bool(do)

    def mean_pose(self):
        s = self.comm.allreduce_sum(list(self.shard.mean_pose_sums()))
        return np.array([s[0], s[1], math.atan2(s[2], s[3])])       # weights are normalised: sums are means


def socket_host():
    import socket
    return socket.gethostname()


def shared_page(dist, rank, world, doubles):
    """A zero-filled float64 page in /dev/shm that every rank of ONE node maps (the ranks' scalar page of a sharded
    filter: slam_pf_attach_exchange).  The file is unlinked once every rank has mapped it."""
    import uuid
    name = [f"/dev/shm/slamhip-x-{uuid.uuid4().hex}" if rank == 0 else None]
    dist.broadcast_object_list(name, src=0)
    if rank == 0:
        fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(np.zeros(doubles, dtype=np.float64).tobytes())
    dist.barrier()
    mem = np.memmap(name[0], dtype=np.float64, mode="r+", shape=(doubles,))
    dist.barrier()
    if rank == 0:
        os.unlink(name[0])
    return mem


class PFSlamState(FastSLAM):
    """``PFSlamState{T}`` (src/common.jl:31-34) as a device-resident, optionally sharded filter.

    ``n`` particles in total; under ``torch.distributed`` every rank constructs it with the same
    arguments and owns n / world of them on its own GPU.
    """

    def __init__(self, n, max_landmarks, seed=0, dtype="f32", device=0, neff_frac=0.75, distributed=None, peers=None):
        """``peers`` (sharded filter): None = the device-side exchange where it can be had (one node, at most 8 ranks, the
        self-test passes; SLAMHIP_PF_PEERS=0 says no), False = the halting flow through ``torch.distributed`` (RCCL on the
        GPUs) whatever the node could do.  ``self.peers`` says which one runs, ``self.selftest_ok`` what the self-test of
        the GPUs' view of each other's inboxes said (None: not tried)."""
        import torch.distributed as dist
        use_dist = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        if use_dist:
            import torch
            rank, world = dist.get_rank(), dist.get_world_size()
            trace = os.environ.get("SLAMHIP_TRACE_CLOSE") == "1"

            def say(msg):
                if trace:
                    print(f"[create rank {rank}] {msg}", file=sys.stderr, flush=True)
            if n % world:
                raise ValueError("n must be divisible by the number of ranks")
            per = n // world
            shard = PFShard(per, max_landmarks, seed, dtype=dtype, first=rank * per, n_global=n, device=device)
            comm = TorchComm(torch.device("cuda", int(device)))
            self.peers = False
            self.selftest_ok = None
            want_peers = (os.environ.get("SLAMHIP_PF_PEERS", "1") != "0") if peers is None else bool(peers)
            if world > 1:
                # the ranks' GPUs address each other's buffers (IPC handles, moved here by an object all-gather): the
                # per-step scalars travel GPU to GPU and a resampling step stays on the device.  SLAMHIP_PF_PEERS=0, more
                # than 8 ranks or ranks on several nodes: the legacy flow -- scalars through a shared pinned page, a
                # resampling step halts and the host resamples through the collectives
                blobs = [None] * world
                say("shard created; exporting")
                mine = shard.export_peer()
                say("exported; all-gather of the blobs")
                dist.all_gather_object(blobs, (socket_host(), mine))
                say("blobs gathered")
                one_node = len({h for h, _ in blobs}) == 1
                if one_node and world <= 8 and want_peers:
                    # attach, then prove that the GPUs see each other's inbox writes; any rank failing either step sends
                    # every rank to the fallback (the decision must be the same everywhere)
                    # one rank at a time (a precaution: with dmabuf IPC the importer asks the EXPORTER's process for the
                    # buffer, so two processes inside an open of each other's memory depend on each other's runtime).
                    # A landmark buffer above 2 GiB makes attach_peers refuse (hipIpcOpenMemHandle hangs above that size
                    # on ROCm 7.2: tools/ipc_gen_test.py) and every rank takes the fallback below
                    ok = True
                    for turn in range(world):
                        if turn == rank:
                            try:
                                shard.attach_peers(rank, world, [b for _, b in blobs])
                            except Exception:  # noqa: BLE001 -- e.g. hipIpcOpenMemHandle refused
                                ok = False
                        dist.barrier()
                    say(f"attached ({ok}); self-test")
                    try:
                        ok = ok and shard.peer_selftest()
                    except Exception:  # noqa: BLE001
                        ok = False
                    say(f"self-test {ok}")
                    oks = [None] * world
                    dist.all_gather_object(oks, bool(ok))
                    self.selftest_ok = bool(all(oks))
                    if all(oks):
                        self.peers = True
                    else:
                        try:
                            shard.detach_peers()
                        except Exception:  # noqa: BLE001
                            pass
                if one_node and not self.peers:
                    shard.attach_exchange(rank, world, shared_page(dist, rank, world, 2 * world * 8))
        else:
            shard = PFShard(n, max_landmarks, seed, dtype=dtype, device=device)
            comm = None
            self.peers = False
            self.selftest_ok = None
        super().__init__(shard, comm, neff_frac)
        self.n = n

    def close(self):
        """Collective for a sharded filter with peers: remote records are brought home, the peers are detached, then a
        barrier -- nobody frees buffers a peer may still read."""
        if getattr(self, "peers", False):
            import torch.distributed as dist
            trace = os.environ.get("SLAMHIP_TRACE_CLOSE") == "1"
            if trace:
                print(f"[close rank {dist.get_rank()}] detach", file=sys.stderr, flush=True)
            self.shard.detach_peers()
            self.shard.sync()
            if trace:
                print(f"[close rank {dist.get_rank()}] barrier", file=sys.stderr, flush=True)
            dist.barrier()
            if trace:
                print(f"[close rank {dist.get_rank()}] destroy", file=sys.stderr, flush=True)
            self.peers = False
        self.shard.close()


def attach_local_peers(shards):
    """Several shards of ONE process (one host thread per shard; several GPUs, or -- the tests -- several shards on one
    card) become the ranks of one sharded filter: every shard attaches every shard's blob (raw pointers, no IPC).
    Shards on ONE card wait for each other inside their kernels, so each shard's stream needs a hardware queue of its own:
    set ``GPU_MAX_HW_QUEUES`` (default 4 per device) to at least ``len(shards) + 2`` before the first HIP call."""
    blobs = [sh.export_peer() for sh in shards]
    for r, sh in enumerate(shards):
        sh.attach_peers(r, len(shards), blobs)
