"""Telemetry messages with the schema of the reference's browser monitor
(``sim/browser/wsserver.jl:20-69,92-98``), SURVEY.md section 8(f) row N3.

The reference sends, per simulation step, JSON messages ``{"type", "data", "timestamp"}`` of type
``tracks``, ``state``, ``lidar``, ``feature-ellipses`` and ``vehicle-ellipse`` to its D3 client
(``sim/browser/wsclient.js``).  Its ``state`` message carries the WHOLE covariance (``:36``) -- 1.6 GB at
10k landmarks; here the ellipses are computed on the device from the 2 x 2 diagonal blocks
(``slam_ekf_ellipses``) and ``state`` carries the covariance only on request.

This module is HOST code (message assembly, a few scalars per landmark); the transport (HTTP/WebSocket
server) is out of scope.  Nothing here imports the oracle.
"""
from __future__ import annotations

import json
import math
import time

import numpy as np


def local_to_global(l, g):
    """``local_to_global`` (src/common.jl:118-132) for 2 x k points: rotate by g[2], translate by g[0:2]."""
    l = np.asarray(l, dtype=np.float64).reshape(2, -1)
    c, s = math.cos(g[2]), math.sin(g[2])
    R = np.array([[c, -s], [s, c]])
    return R @ l + np.asarray(g[:2], dtype=np.float64).reshape(2, 1)


def laser_lines(z, vehicle_pose):
    """``laser_lines`` (src/common.jl:269-283): 4 x nz matrix of [vx, vy, fx, fy] beam end points."""
    z = np.asarray(z, dtype=np.float64).reshape(2, -1)
    lines = np.empty((4, z.shape[1]))
    lines[0, :] = vehicle_pose[0]
    lines[1, :] = vehicle_pose[1]
    r, b = z[0], z[1]
    lines[2:4, :] = local_to_global(np.vstack([r * np.cos(b), r * np.sin(b)]), vehicle_pose)
    return lines


def dict_array(a, keys):
    """``dict_array`` (sim/browser/wsserver.jl:120-131): one dict per COLUMN of a, keyed by ``keys``."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    if len(keys) > a.shape[0]:
        raise ValueError("more keys than rows")
    return [{k: float(a[i, j]) for i, k in enumerate(keys)} for j in range(a.shape[1])]


def message(name, data, timestamp=None):
    """``send_json``'s envelope (sim/browser/wsserver.jl:92-98)."""
    return {"type": name, "data": data, "timestamp": time.time() if timestamp is None else timestamp}


def monitor_messages(state, true_pose, slam_pose, z=None, state_updated=False, include_cov=False, timestamp=None):
    """The messages of one ``monitor`` call (sim/browser/wsserver.jl:20-69), in its order.

    ``state``: an ``EKFSlamState``; ``true_pose`` / ``slam_pose``: the latest track points (``tt[:, n]``,
    ``st[:, n]``); ``z``: the 2 x nz observations of this step (``simdata.z[:, 1:nz]``) if the state was
    updated.  ``include_cov`` reproduces the reference's ``"cov"`` field by downloading P.
    """
    out = []
    out.append(message("tracks", {"ideal": {"x": float(true_pose[0]), "y": float(true_pose[1]), "phi": float(true_pose[2])},
                                  "slam": {"x": float(slam_pose[0]), "y": float(slam_pose[1]), "phi": float(slam_pose[2])}},
                       timestamp))
    pose = [float(v) for v in state.pose()]
    d = {"pose": pose}
    if include_cov:
        d["cov"] = np.asarray(state.cov).tolist()
    out.append(message("state", d, timestamp))
    if state_updated and z is not None and np.asarray(z).size > 0:
        out.append(message("lidar", dict_array(laser_lines(z, pose), ["x1", "y1", "x2", "y2"]), timestamp))
        if state.N > 0:
            out.append(message("feature-ellipses", dict_array(state.feature_ellipses(), ["cx", "cy", "rx", "ry", "phi"]),
                               timestamp))
    out.append(message("vehicle-ellipse", dict_array(state.vehicle_ellipse(), ["cx", "cy", "vehicle_phi", "rx", "ry", "phi"]),
                       timestamp))
    return out


def to_json(msg):
    """What ``write(client, JSON.json(msg))`` puts on the wire."""
    return json.dumps(msg)
