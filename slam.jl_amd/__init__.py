"""slam.jl_amd -- MI355X-native EKF-SLAM filter core behind SLAM.jl's function surface.

The directory name is not a Python identifier; load it with
``__graft_entry__.load_package()`` (registers it as ``slam_jl_amd``).

Contents: ``csrc/`` (HIP kernels + C ABI -> ``libslamhip.so``), ``_lib`` (ctypes
binding), ``ekf`` (host mirror of src/SLAM.jl's exports), ``sim`` (headless
``sim!`` driver), ``telemetry`` (the browser monitor's message schema), ``SLAMHip.jl`` (the Julia 1.x binding).  Importing this package
fails loudly if the HIP library has not been built: there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when libslamhip.so is missing)
from ._lib import NotPositiveDefinite, SlamHipError, device_count  # noqa: F401
from .ekf import (DeviceRef, EKFSlamState, SlamState, add_features, associate, augment_,  # noqa: F401
                  compute_association, ekf_predict_, ekf_update_, mpi_to_pi, predict, predict_observation,
                  observe, update)
from .pf import (FastSLAM, PFShard, PFSlamState, TorchComm, attach_local_peers, philox_uniform, shared_page,  # noqa: F401
                 small)
from . import sim  # noqa: F401
from . import telemetry  # noqa: F401
