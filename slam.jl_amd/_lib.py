"""ctypes binding of libslamhip.so (include/slamhip.h).

There is NO fallback: if the shared library is missing or cannot be loaded the
import fails loudly, and if no HIP device is usable ``slam_ekf_create`` returns
SLAM_E_HIP which is raised as :class:`SlamHipError`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslamhip.so")
# measurement tooling only (tools/gpu_exp.sh): the experiments build of the SAME sources, `make -C csrc exp`
if os.environ.get("SLAMHIP_LIBRARY"):
    LIB_PATH = os.path.abspath(os.environ["SLAMHIP_LIBRARY"])

SLAM_OK = 0
SLAM_E_BADARG = -1
SLAM_E_CAPACITY = -2
SLAM_E_NOTPD = -3
SLAM_E_HIP = -4
SLAM_E_NOMEM = -5
SLAM_PF_HALTED = 1
SLAM_PF_PEER_BLOB_BYTES = 4096
SLAM_F32, SLAM_F64 = 0, 1
SLAM_FORM_CHOLESKY, SLAM_FORM_JOSEPH = 0, 1
KERNEL_IDS = {"gate": 0, "gate_final": 1, "predict": 2, "augment": 3, "pht": 4, "factor": 5, "w1": 6, "syrk": 7}

_ERRNAMES = {SLAM_E_BADARG: "SLAM_E_BADARG", SLAM_E_CAPACITY: "SLAM_E_CAPACITY", SLAM_E_NOTPD: "SLAM_E_NOTPD",
             SLAM_E_HIP: "SLAM_E_HIP", SLAM_E_NOMEM: "SLAM_E_NOMEM"}


class SlamHipError(RuntimeError):
    """A non-zero status from libslamhip (Julia wrapper: ``error(...)``)."""

    def __init__(self, code, message):
        super().__init__(f"{_ERRNAMES.get(code, code)}: {message}")
        self.code = code


class NotPositiveDefinite(SlamHipError):
    """SLAM_E_NOTPD -- the reference's ``chol`` would throw PosDefException (src/ekf.jl:70)."""


def _load():
    # libslamhip.so itself does not depend on torch.  But when torch is used in the same process (the
    # FastSLAM collectives, bench.py) both must share ONE HIP runtime: torch's wheel bundles its own
    # libamdhip64, and if the system runtime gets loaded first torch later reports "No HIP GPUs are
    # available".  So torch, if installed, is imported before the library is opened.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C slam.jl_amd/csrc`.  slam.jl_amd has no CPU fallback.")
    try:
        return C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise ImportError(f"cannot load {LIB_PATH}: {e}.  slam.jl_amd has no CPU fallback.") from e


lib = _load()

_h = C.c_void_p
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

#: every symbol include/slamhip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "slam_last_error": (C.c_char_p, []),
    "slam_device_count": (C.c_int, []),
    "slam_ekf_create": (C.c_int, [C.POINTER(_h), C.c_int, C.c_int, C.c_int]),
    "slam_ekf_destroy": (C.c_int, [_h]),
    "slam_ekf_set_state": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "slam_ekf_set_state_device": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "slam_ekf_get_state": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "slam_ekf_get_block": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "slam_ekf_get_diag": (C.c_int, [_h, C.c_void_p]),
    "slam_ekf_get_landmark_blocks": (C.c_int, [_h, C.c_void_p]),
    "slam_ekf_set_gate_mode": (C.c_int, [_h, C.c_int]),
    "slam_ekf_gate_info": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "slam_ekf_get_pose": (C.c_int, [_h, _dp]),
    "slam_ekf_num_landmarks": (C.c_int, [_h, C.POINTER(C.c_int)]),
    "slam_ekf_dtype": (C.c_int, [_h, C.POINTER(C.c_int)]),
    "slam_ekf_device_ptrs": (C.c_int, [_h, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                       C.POINTER(C.c_void_p)]),
    "slam_ekf_predict": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_double]),
    "slam_ekf_associate": (C.c_int, [_h, _dp, C.c_int, _dp, C.c_double, C.c_double, _ip]),
    "slam_ekf_nis": (C.c_int, [_h, _dp, C.c_int, _dp, _dp]),
    "slam_ekf_predict_observation": (C.c_int, [_h, C.c_int, _dp, _dp, _dp]),
    "slam_ekf_update": (C.c_int, [_h, _dp, _ip, C.c_int, _dp, C.c_int]),
    "slam_ekf_augment": (C.c_int, [_h, _dp, C.c_int, _dp]),
    "slam_ekf_ellipses": (C.c_int, [_h, _dp, _dp]),
    "slam_ekf_observe": (C.c_int, [_h, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_int, _ip]),
    "slam_ekf_set_async": (C.c_int, [_h, C.c_int]),
    "slam_ekf_sync": (C.c_int, [_h]),
    "slam_ekf_timing": (C.c_int, [_h, C.c_int]),
    "slam_ekf_timing_read": (C.c_int, [_h, C.c_int, _dp, C.POINTER(C.c_int64)]),
    "slam_ekf_timing_reset": (C.c_int, [_h]),
    "slam_ekf_timing_min": (C.c_int, [_h, C.c_int, _dp]),
    "slam_ekf_timing_stats": (C.c_int, [_h, C.c_int, _dp]),
    "slam_ekf_debug_stamps": (C.c_int, [_h, C.c_int, C.POINTER(C.c_uint64)]),
    "slam_ekf_state_written": (C.c_int, [_h]),
    "slam_ekf_copy_floor": (C.c_int, [_h, C.c_int, _dp]),
    "slam_pf_create": (C.c_int, [C.POINTER(_h), C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint64]),
    "slam_pf_destroy": (C.c_int, [_h]),
    "slam_pf_set_pose": (C.c_int, [_h, _dp]),
    "slam_pf_init_landmarks": (C.c_int, [_h, _dp, C.c_int, C.c_double, C.c_double]),
    "slam_pf_predict": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_double]),
    "slam_pf_update_known": (C.c_int, [_h, _dp, _ip, C.c_int, _dp]),
    "slam_pf_clear_landmarks": (C.c_int, [_h]),
    "slam_pf_update_unknown": (C.c_int, [_h, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_void_p]),
    "slam_pf_step": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_double, _dp, _ip, C.c_int, _dp, _dp]),
    "slam_pf_step_proposal": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_double, _dp, _ip, C.c_int, _dp, _dp]),
    "slam_pf_step_normalized": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_double, _dp, _ip, C.c_int, _dp,
                                          _dp]),
    "slam_pf_weight_stats": (C.c_int, [_h, _dp]),
    "slam_pf_normalize": (C.c_int, [_h, C.c_double, C.c_double]),
    "slam_pf_copy_logw": (C.c_int, [_h, C.c_void_p]),
    "slam_pf_resample_local": (C.c_int, [_h, C.c_double, C.c_double]),
    "slam_pf_ancestors_all": (C.c_int, [_h, C.c_void_p, C.c_double, C.c_double, C.c_void_p]),
    "slam_pf_ancestors": (C.c_int, [_h, C.c_void_p, C.c_double, C.c_double, C.c_void_p]),
    "slam_pf_record_rows": (C.c_int, [_h, C.POINTER(C.c_int)]),
    "slam_pf_pack": (C.c_int, [_h, C.c_void_p, C.c_int, C.c_void_p]),
    "slam_pf_resample_apply": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "slam_pf_mean_pose_sums": (C.c_int, [_h, _dp]),
    "slam_pf_download": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_void_p]),
    "slam_pf_sync": (C.c_int, [_h]),
    "slam_pf_stream": (C.c_int, [_h, C.POINTER(C.c_void_p)]),
    "slam_pf_step_auto": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_double, _dp, _ip, C.c_int, _dp, C.c_double,
                                    C.c_int, C.c_int]),
    "slam_pf_step_auto_batch": (C.c_int, [_h, C.c_int, _dp, C.c_double, _dp, C.c_double, _dp, _ip, _ip, C.c_int, _dp, C.c_double,
                                          _ip, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "slam_pf_flush": (C.c_int, [_h, _dp]),
    "slam_pf_halt_info": (C.c_int, [_h, _dp]),
    "slam_pf_resume": (C.c_int, [_h, C.c_int64]),
    "slam_pf_resample_count": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "slam_pf_set_resample_count": (C.c_int, [_h, C.c_int64]),
    "slam_pf_attach_exchange": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "slam_pf_export_peer": (C.c_int, [_h, C.c_void_p]),
    "slam_pf_attach_peers": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "slam_pf_detach_peers": (C.c_int, [_h]),
    "slam_pf_peer_selftest": (C.c_int, [_h, C.c_int]),
    "slam_pf_comm_info": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "slam_pf_debug_stamps": (C.c_int, [_h, C.POINTER(C.c_uint64)]),
    "slam_pf_resample": (C.c_int, [_h, C.c_double, C.POINTER(C.c_int)]),
    "slam_pf_get_mean_pose": (C.c_int, [_h, _dp]),
    "slam_pf_get_weights": (C.c_int, [_h, _dp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = the library lacks a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    msg = lib.slam_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    if rc == SLAM_OK:
        return
    if rc == SLAM_E_NOTPD:
        raise NotPositiveDefinite(rc, last_error())
    raise SlamHipError(rc, last_error())


def device_count() -> int:
    return int(lib.slam_device_count())
