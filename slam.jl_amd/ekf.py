"""Host-side mirror of SLAM.jl's EKF surface (src/SLAM.jl:5-30) over libslamhip.

Same names, argument meaning and error behaviour as the reference functions --
``predict``, ``update``, ``add_features``, ``associate``, ``compute_association``,
``predict_observation``, ``mpi_to_pi`` and the types ``SlamState`` /
``EKFSlamState`` -- so a caller written like ``sim!`` (sim/ekfslam-sim.jl:100-120)

    state.x, state.cov = predict(state, vehicle, Q, dt)
    zf, idf, zn = associate(state, z, R, 4.0, 25.0)
    state.x, state.cov = update(state, zf, R, idf)
    state.x, state.cov = add_features(state, zn, R)

runs unchanged.  The state lives on the GPU: ``state.x`` / ``state.cov`` are lazy
references (:class:`DeviceRef`) that download only when turned into an array, and
re-assigning a state's own references back to it is free.  The in-place names of
BASELINE.json's north star are provided as ``ekf_predict_`` / ``ekf_update_`` /
``augment_`` (Julia: ``ekf_predict!`` / ``ekf_update!`` / ``augment!``).

Every numeric operation is a HIP kernel behind the C ABI; nothing here computes
filter quantities on the CPU, and nothing imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import SLAM_F32, SLAM_F64, SLAM_FORM_CHOLESKY, SLAM_FORM_JOSEPH, check, lib

__all__ = [
    "observe",
    "SlamState", "EKFSlamState", "DeviceRef", "predict", "update", "add_features", "associate",
    "compute_association", "predict_observation", "mpi_to_pi", "ekf_predict_", "ekf_update_", "augment_",
]

_DTYPES = {"f32": (SLAM_F32, np.float32), "f64": (SLAM_F64, np.float64),
           np.float32: (SLAM_F32, np.float32), np.float64: (SLAM_F64, np.float64),
           "float32": (SLAM_F32, np.float32), "float64": (SLAM_F64, np.float64)}


def _dbl(a, n=None):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if n is not None and arr.size != n:
        raise ValueError(f"expected {n} values, got {arr.size}")
    return arr


def _small(M):
    """2 x 2 matrix -> column-major double[4]."""
    M = np.asarray(M, dtype=np.float64)
    if M.shape != (2, 2):
        raise ValueError("expected a 2 x 2 matrix")
    return np.ascontiguousarray(M.T).reshape(4)          # [m11, m21, m12, m22]


def _ptr(a, ctype=C.c_double):
    return a.ctypes.data_as(C.POINTER(ctype))


def _obs(z):
    """Reference layout z: 2 x nz (rows range, bearing) -> contiguous (range, bearing) pairs."""
    z = np.asarray(z, dtype=np.float64)
    if z.size == 0:
        return np.zeros((0, 2))
    if z.ndim == 1:
        z = z.reshape(2, 1)
    if z.shape[0] != 2:
        raise ValueError("z must be 2 x nz")
    return np.ascontiguousarray(z.T)


def mpi_to_pi(phi):
    """src/common.jl:102-110 (host scalar helper: one conditional wrap)."""
    if phi > math.pi:
        return phi - 2 * math.pi
    if phi < -math.pi:
        return phi + 2 * math.pi
    return phi


class SlamState:
    """``abstract SlamState`` (src/common.jl:22)."""


class DeviceRef:
    """Lazy reference to ``x`` or ``cov`` of a device-resident state."""

    __slots__ = ("state", "which")

    def __init__(self, state, which):
        self.state = state
        self.which = which

    def __array__(self, dtype=None, copy=None):
        a = self.state.download(self.which)
        return a.astype(dtype) if dtype is not None else a

    def numpy(self):
        return self.state.download(self.which)

    def __len__(self):
        return self.state.n

    @property
    def shape(self):
        n = self.state.n
        return (n,) if self.which == "x" else (n, n)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __repr__(self):
        return f"<DeviceRef {self.which} of {self.state!r}>"


class EKFSlamState(SlamState):
    """``EKFSlamState{T}(x, cov)`` (src/common.jl:25-28), device resident.

    ``max_landmarks`` fixes the capacity (the reference grows P by reallocating,
    src/ekf.jl:108-109); ``dtype`` is "f32" or "f64" (reference: Float64).
    """

    def __init__(self, x, cov, dtype="f64", max_landmarks=None, device=0):
        code, npdt = _DTYPES[dtype]
        x = np.asarray(x)
        n = int(x.shape[0])
        if n < 3 or (n - 3) % 2:
            raise ValueError("length(x) must be 3 + 2*N")
        N = (n - 3) // 2
        if max_landmarks is None:
            max_landmarks = max(64, 2 * N)
        self.np_dtype = npdt
        self.max_landmarks = int(max_landmarks)
        self._h = C.c_void_p()
        check(lib.slam_ekf_create(C.byref(self._h), code, self.max_landmarks, int(device)))
        try:
            self.set_state(x, cov)
        except Exception:
            self.close()
            raise

    # -- lifetime ----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.slam_ekf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return f"EKFSlamState(N={self.N}, dtype={np.dtype(self.np_dtype).name}, max_landmarks={self.max_landmarks})"

    # -- sizes ----------------------------------------------------------------------
    @property
    def N(self):
        out = C.c_int()
        check(lib.slam_ekf_num_landmarks(self._h, C.byref(out)))
        return out.value

    @property
    def n(self):
        return 3 + 2 * self.N

    # -- state I/O --------------------------------------------------------------------
    def set_state(self, x, cov):
        x = np.ascontiguousarray(np.asarray(x, dtype=self.np_dtype))
        n = x.shape[0]
        cov = np.asarray(cov, dtype=self.np_dtype)
        if cov.shape != (n, n):
            raise ValueError("cov must be n x n")
        covf = np.asfortranarray(cov)
        check(lib.slam_ekf_set_state(self._h, x.ctypes.data, covf.ctypes.data, n, n))

    def set_state_device(self, d_x_ptr, d_P_ptr, n, ldP):
        """Upload from device pointers (e.g. ``tensor.data_ptr()``); column-major P."""
        check(lib.slam_ekf_set_state_device(self._h, C.c_void_p(d_x_ptr), C.c_void_p(d_P_ptr), int(n), int(ldP)))

    def download(self, which="both"):
        n = self.n
        x = np.empty(n, dtype=self.np_dtype) if which in ("x", "both") else None
        P = np.empty((n, n), dtype=self.np_dtype, order="F") if which in ("cov", "both") else None
        check(lib.slam_ekf_get_state(self._h, x.ctypes.data if x is not None else None,
                                     P.ctypes.data if P is not None else None, n, n))
        if which == "x":
            return x
        if which == "cov":
            return P
        return x, P

    def get_block(self, r0, c0, nr, nc):
        """cov[r0:r0+nr, c0:c0+nc] (0-based) without downloading the matrix (slam_ekf_get_block)."""
        out = np.empty((int(nr), int(nc)), dtype=self.np_dtype, order="F")
        check(lib.slam_ekf_get_block(self._h, int(r0), int(c0), int(nr), int(nc), out.ctypes.data, max(int(nr), 1)))
        return out

    def diag(self):
        """diag(cov) (slam_ekf_get_diag)."""
        out = np.empty(self.n, dtype=self.np_dtype)
        check(lib.slam_ekf_get_diag(self._h, out.ctypes.data))
        return out

    def landmark_blocks(self):
        """[3, N]: P[f, f], P[f+1, f], P[f+1, f+1] of every landmark (slam_ekf_get_landmark_blocks: the packed side array
        the gating sweep streams)."""
        out = np.empty((3, self.N), dtype=self.np_dtype)
        check(lib.slam_ekf_get_landmark_blocks(self._h, out.ctypes.data))
        return out

    GATE_MODES = {"auto": 0, "sweep": 1, "grid": 2}

    def set_gate_mode(self, mode):
        """How associate / observe search the map (slam_ekf_set_gate_mode; the reference's TODO, src/data-association.jl:18-20):
        "sweep" = every landmark, "grid" = the uniform grid over the landmark means (O(candidates)), "auto" = the grid
        from 16384 landmarks on.  Decisions are identical in every mode."""
        check(lib.slam_ekf_set_gate_mode(self._h, self.GATE_MODES[mode]))

    def gate_info(self):
        """slam_ekf_gate_info as a dict: which form the last gating used, the grid's size, and what the grid queries have
        visited / fully evaluated so far."""
        out = (C.c_int64 * 8)()
        check(lib.slam_ekf_gate_info(self._h, out))
        keys = ("form", "cells_per_axis", "in_grid", "tail", "rebuilds", "queries", "visited", "evaluated")
        d = dict(zip(keys, (int(v) for v in out)))
        d["form"] = {0: None, 1: "sweep", 2: "grid"}[d["form"]]
        return d

    def device_ptrs(self):
        """(x_ptr, P_ptr, ld, stream_ptr) raw device addresses for zero-copy interop."""
        dx, dP, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ld = C.c_int()
        check(lib.slam_ekf_device_ptrs(self._h, C.byref(dx), C.byref(dP), C.byref(ld), C.byref(st)))
        return dx.value, dP.value, ld.value, st.value

    @property
    def x(self):
        return DeviceRef(self, "x")

    @x.setter
    def x(self, value):
        if isinstance(value, DeviceRef) and value.state is self:
            return                                   # state.x, state.cov = predict(state, ...)
        self._pending_x = np.asarray(value, dtype=self.np_dtype)
        self._flush_pending()

    @property
    def cov(self):
        return DeviceRef(self, "cov")

    @cov.setter
    def cov(self, value):
        if isinstance(value, DeviceRef) and value.state is self:
            return
        self._pending_cov = np.asarray(value, dtype=self.np_dtype)
        self._flush_pending()

    def _flush_pending(self):
        # x and cov change size together (reset: sim/browser/wsserver.jl:161-174); upload once both agree
        px = getattr(self, "_pending_x", None)
        pc = getattr(self, "_pending_cov", None)
        if px is None:
            px = self.download("x")
        if pc is None:
            pc = self.download("cov")
        if pc.shape == (px.shape[0], px.shape[0]):
            self.set_state(px, pc)
            self._pending_x = None
            self._pending_cov = None

    def pose(self):
        out = np.empty(3)
        check(lib.slam_ekf_get_pose(self._h, _ptr(out)))
        return out

    def feature_ellipses(self):
        """feature_ellipses(x, cov) (sim/browser/wsserver.jl:72-85): 5 x N array [cx; cy; rx; ry; phi], computed
        on the device from the 2 x 2 diagonal blocks -- P is not downloaded."""
        N = self.N
        out = np.empty((5, N), dtype=np.float64, order="F")
        check(lib.slam_ekf_ellipses(self._h, _ptr(out) if N else None, None))
        return out

    def vehicle_ellipse(self):
        """[cx, cy, vehicle_phi, rx, ry, phi] of monitor() (sim/browser/wsserver.jl:60-65)."""
        out = np.empty(6)
        check(lib.slam_ekf_ellipses(self._h, None, _ptr(out)))
        return out

    # -- in-place operations (ekf_predict!, ekf_update!, augment!) ---------------------
    def predict(self, v, g, wheelbase, Q, dt):
        q = _small(Q)
        check(lib.slam_ekf_predict(self._h, float(v), float(g), float(wheelbase), _ptr(q), float(dt)))

    def associate_vector(self, z, R, gate1, gate2):
        """int32 assoc[nz]: j >= 1 matched landmark, 0 dropped, -1 new feature."""
        zp = _obs(z)
        nz = zp.shape[0]
        assoc = np.zeros(nz, dtype=np.int32)
        if nz:
            r = _small(R)
            check(lib.slam_ekf_associate(self._h, _ptr(zp), nz, _ptr(r), float(gate1), float(gate2),
                                         _ptr(assoc, C.c_int32)))
        return assoc

    def associate(self, z, R, gate1, gate2):
        """(zf 2 x nf, idf 1 x nf Int, zn 2 x nn), observation order preserved
        (src/data-association.jl:11-13,43-47)."""
        z = np.asarray(z, dtype=np.float64).reshape(2, -1)
        assoc = self.associate_vector(z, R, gate1, gate2)
        zf = z[:, assoc > 0]
        idf = assoc[assoc > 0].astype(np.int64).reshape(1, -1)
        zn = z[:, assoc < 0]
        return zf, idf, zn

    def update(self, zf, R, idf, form="cholesky"):
        zp = _obs(zf)
        ids = np.ascontiguousarray(np.asarray(idf, dtype=np.int32).reshape(-1))
        if ids.shape[0] != zp.shape[0]:
            raise ValueError("idf and z disagree on the number of observations")
        if zp.shape[0] == 0:
            return
        r = _small(R)
        code = SLAM_FORM_JOSEPH if form == "joseph" else SLAM_FORM_CHOLESKY
        check(lib.slam_ekf_update(self._h, _ptr(zp), _ptr(ids, C.c_int32), zp.shape[0], _ptr(r), code))

    def add_features(self, zn, R):
        zp = _obs(zn)
        if zp.shape[0] == 0:
            return
        r = _small(R)
        check(lib.slam_ekf_augment(self._h, _ptr(zp), zp.shape[0], _ptr(r)))

    def observe(self, z, R, gate1, gate2, form="cholesky"):
        """associate -> update -> add_features (sim/ekfslam-sim.jl:114-120) in one library call with no host
        round trip between the gating and the update.  Returns the association vector
        (see associate_vector); the state afterwards equals the three calls in sequence."""
        zp = _obs(z)
        nz = zp.shape[0]
        assoc = np.zeros(nz, dtype=np.int32)
        if nz:
            r = _small(R)
            code = SLAM_FORM_JOSEPH if form == "joseph" else SLAM_FORM_CHOLESKY
            check(lib.slam_ekf_observe(self._h, _ptr(zp), nz, _ptr(r), float(gate1), float(gate2), code,
                                       _ptr(assoc, C.c_int32)))
        return assoc

    def compute_association(self, z, R, idf):
        zz = _dbl(z, 2)
        r = _small(R)
        out = np.empty(2)
        check(lib.slam_ekf_nis(self._h, _ptr(zz), int(idf), _ptr(r), _ptr(out)))
        return float(out[0]), float(out[1])

    def predict_observation(self, idf):
        """(z (2,), H (2, n) dense with 5 non-zero columns) -- src/common.jl:139-165."""
        zp, Hv, Hf = np.empty(2), np.empty(6), np.empty(4)
        check(lib.slam_ekf_predict_observation(self._h, int(idf), _ptr(zp), _ptr(Hv), _ptr(Hf)))
        n = self.n
        H = np.zeros((2, n))
        H[:, 0:3] = Hv.reshape(3, 2).T
        f = 3 + 2 * (int(idf) - 1)
        H[:, f:f + 2] = Hf.reshape(2, 2).T
        return zp, H

    # -- stream / timing ----------------------------------------------------------------
    def set_async(self, flag=True):
        check(lib.slam_ekf_set_async(self._h, 1 if flag else 0))

    def sync(self):
        check(lib.slam_ekf_sync(self._h))

    def timing(self, enable=True, kernels=None):
        """Bracket kernel launches with HIP events; ``kernels``: names from ``_lib.KERNEL_IDS`` to restrict
        it to (each event pair costs ~10 us of stream time)."""
        mask = 1 if enable else 0
        if enable and kernels:
            mask = 0
            for name in kernels:
                mask |= 2 << _lib.KERNEL_IDS[name]
        check(lib.slam_ekf_timing(self._h, mask))

    def state_written(self):
        """Tell the library that landmark entries of x / P were written through the raw device views
        (:meth:`device_ptrs`): the gating's side array, variance bound and grid are refreshed on the device."""
        check(lib.slam_ekf_state_written(self._h))

    def copy_floor(self, reps=10):
        """Bare read + rewrite of the stored covariance tiles (the memory side of the down-date, nothing else), timed on
        this handle's own matrix: ``(milliseconds per pass, launch form)``.  The state is unchanged."""
        out = (C.c_double * 2)()
        check(lib.slam_ekf_copy_floor(self._h, int(reps), out))
        return float(out[0]), ("one workgroup per tile", "persistent grid")[int(out[1])]

    def debug_stamps(self, enable=True):
        """Diagnostics: 100 MHz wall-clock stamps of the factorisation kernel's phases (last update): [0..7] the workgroup that
        factors S, [8..15] the first of the workgroups that form W1 in the same launch."""
        out = (C.c_uint64 * 16)()
        check(lib.slam_ekf_debug_stamps(self._h, 1 if enable else 0, out))
        return [int(v) for v in out]

    def timing_reset(self):
        check(lib.slam_ekf_timing_reset(self._h))

    def timing_read(self):
        """{kernel: (total_ms, launches)} accumulated since the last reset."""
        out = {}
        for name, kid in _lib.KERNEL_IDS.items():
            ms, cnt = C.c_double(), C.c_int64()
            check(lib.slam_ekf_timing_read(self._h, kid, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def timing_min(self, kernel):
        """Milliseconds of the fastest bracketed launch of `kernel` since the last reset (0.0: none)."""
        ms = C.c_double()
        check(lib.slam_ekf_timing_min(self._h, _lib.KERNEL_IDS[kernel], C.byref(ms)))
        return ms.value

    def timing_stats(self, kernel):
        """(launches, mean_ms, stdev_ms, min_ms) of the bracketed launches of `kernel` since the last reset."""
        out = np.zeros(4)
        check(lib.slam_ekf_timing_stats(self._h, _lib.KERNEL_IDS[kernel], _ptr(out)))
        return int(out[0]), float(out[1]), float(out[2]), float(out[3])


# ---- the reference's function surface ---------------------------------------------------

def _state_of(ref_or_state):
    if isinstance(ref_or_state, EKFSlamState):
        return ref_or_state
    if isinstance(ref_or_state, DeviceRef):
        return ref_or_state.state
    return None


def predict(state: EKFSlamState, vehicle, Q, dt):
    """predict(state, vehicle, Q, dt) -> (x, P)   src/ekf.jl:8-43.
    ``vehicle`` needs ``measured_speed``, ``measured_gamma``, ``wheelbase`` (:14-16)."""
    state.predict(vehicle.measured_speed, vehicle.measured_gamma, vehicle.wheelbase, Q, dt)
    return state.x, state.cov


def update(state: EKFSlamState, z, R, idf):
    """update(state, z, R, idf) -> (x, P)   src/ekf.jl:46-77 (here: in place on the device)."""
    state.update(z, R, idf)
    return state.x, state.cov


def add_features(state: EKFSlamState, z, R):
    """add_features(state, z, R) -> (x, P)   src/ekf.jl:84-122."""
    state.add_features(z, R)
    return state.x, state.cov


def associate(state: SlamState, z, R, gate1, gate2):
    """associate(state, z, R, gate1, gate2) -> (zf, idf, zn)   src/data-association.jl:1-51."""
    return state.associate(z, R, gate1, gate2)


def observe(state: EKFSlamState, z, R, gate1, gate2, form="cholesky"):
    """associate + update + add_features (sim/ekfslam-sim.jl:114-120) in one call -> association vector."""
    return _state_of(state).observe(z, R, gate1, gate2, form=form)


def _temp_state(x, P=None):
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if P is None:
        P = np.zeros((n, n))
    return EKFSlamState(x, P, dtype="f64", max_landmarks=max(1, (n - 3) // 2))


def compute_association(x, P, z, R, idf):
    """compute_association(x, P, z, R, idf) -> (nis, nd)   src/data-association.jl:53-63.
    ``x`` / ``P`` may be a state's references (no transfer) or host arrays (uploaded to a
    temporary device state)."""
    st = _state_of(x)
    if st is not None:
        return st.compute_association(z, R, idf)
    tmp = _temp_state(x, P)
    try:
        return tmp.compute_association(z, R, idf)
    finally:
        tmp.close()


def predict_observation(x, idf):
    """predict_observation(x, idf) -> (z, H)   src/common.jl:139-165."""
    st = _state_of(x)
    if st is not None:
        return st.predict_observation(idf)
    tmp = _temp_state(x)
    try:
        return tmp.predict_observation(idf)
    finally:
        tmp.close()


# in-place names of BASELINE.json's north star (Julia: ekf_predict!, ekf_update!, augment!)
def ekf_predict_(state, v, g, wheelbase, Q, dt):
    state.predict(v, g, wheelbase, Q, dt)
    return state


def ekf_update_(state, z, R, idf, form="cholesky"):
    state.update(z, R, idf, form=form)
    return state


def augment_(state, z, R):
    state.add_features(z, R)
    return state
