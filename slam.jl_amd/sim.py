"""Headless, seeded, sleep-free driver with the call pattern of the reference's
``sim!`` (sim/ekfslam-sim.jl:54-143).  SURVEY.md section 8(f) row N1.

This is HOST logic around the hot path, not the hot path: vehicle kinematics,
the waypoint follower and the simulated sensor are a few scalar operations per
step and stay on the CPU exactly as in the reference.  The filter is an object
passed in by the caller with the reference's four entry points

    filt.predict(v, g, wheelbase, Q, dt)
    zf, idf, zn = filt.associate(z, R, gate1, gate2)
    filt.update(zf, R, idf)
    filt.add_features(zn, R)
    filt.pose() -> (x, y, phi)

(``slam.jl_amd.ekf.EKFSlamState`` provides them on the GPU; the tests wrap the
oracle the same way).  Nothing here imports the oracle.

Differences from the reference, all deliberate and test-visible:
* the real-time frame limiter ``sleep`` (sim/ekfslam-sim.jl:132-136) and the
  pause loop (:138-140) are dropped -- they are UI pacing, not work;
* the unseeded global RNG (sim/sim-utils.jl:5,36-37,68) is replaced by a
  seeded ``numpy.random.Generator``; draw order follows the reference:
  speed noise, steering noise per step; then per observation step one row of
  range draws followed by one row of bearing draws;
* every filter input is recorded (``SimLog``) so a run can be replayed through
  another filter without the RNG.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


def mpi_to_pi(phi: float) -> float:
    """Single conditional wrap, src/common.jl:102-110."""
    if phi > math.pi:
        return phi - 2 * math.pi
    if phi < -math.pi:
        return phi + 2 * math.pi
    return phi


def get_waypoints(txtfile) -> np.ndarray:
    """2 x N waypoint array from a header + 2-column text file (src/common.jl:84-87)."""
    return np.loadtxt(txtfile, skiprows=1).T


def make_landmarks(n: int, boundaries, margin: float, rng: np.random.Generator) -> np.ndarray:
    """sim/sim-utils.jl:1-6 -- every coordinate drawn uniformly from the pool
    ``[xmin+bx : xmax-bx ; ymin+by : ymax-by]`` (unit-step ranges, so whole
    numbers for the reference's 0..100 scene)."""
    xmin, xmax, ymin, ymax = boundaries
    bx = margin * (xmax - xmin)
    by = margin * (ymax - ymin)
    pool = np.concatenate([np.arange(xmin + bx, xmax - bx + 1e-9, 1.0),
                           np.arange(ymin + by, ymax - by + 1e-9, 1.0)])
    return pool[rng.integers(0, len(pool), size=(2, n))]


@dataclass
class Vehicle:
    """The fields of src/common.jl:36-57 that the filter path touches."""
    wheelbase: float = 4.0                      # sim/ekfslam-sim.jl:30
    max_gamma: float = 60 * math.pi / 180       # :31
    steer_rate: float = 60 * math.pi / 180      # :32
    sensor_range: float = 30.0                  # :33
    pose: np.ndarray = field(default_factory=lambda: np.zeros(3))
    target_speed: float = 8.0                   # :36
    measured_speed: float = 0.0
    target_gamma: float = 0.0
    measured_gamma: float = 0.0
    waypoint_id: int = 1                        # 1-based, 0 = finished


def initial_pose(waypoints: np.ndarray) -> np.ndarray:
    """src/common.jl:93-96."""
    return np.array([waypoints[0, 0], waypoints[1, 0],
                     math.atan2(waypoints[1, 1] - waypoints[1, 0],
                                waypoints[0, 1] - waypoints[0, 0])])


def step_vehicle(vehicle: Vehicle, dt: float) -> None:
    """src/common.jl:172-181 (ideal speed / steering angle)."""
    x, y, phi = vehicle.pose
    v, g = vehicle.target_speed, vehicle.target_gamma
    vehicle.pose = np.array([x + v * dt * math.cos(g + phi),
                             y + v * dt * math.sin(g + phi),
                             mpi_to_pi(phi + v * dt * math.sin(g) / vehicle.wheelbase)])


def steer(vehicle: Vehicle, waypoints: np.ndarray, d_min: float, dt: float) -> None:
    """src/common.jl:189-230."""
    g = vehicle.target_gamma
    iwp = vehicle.waypoint_id
    x, y, phi = vehicle.pose
    cwp = waypoints[:, iwp - 1]
    d2 = (cwp[0] - x) ** 2 + (cwp[1] - y) ** 2
    if d2 < d_min ** 2:
        iwp += 1
        if iwp > waypoints.shape[1]:
            vehicle.waypoint_id = 0
            return
        cwp = waypoints[:, iwp - 1]
    dg = mpi_to_pi(math.atan2(cwp[1] - y, cwp[0] - x) - phi - g)
    dgmax = vehicle.steer_rate * dt
    if abs(dg) > dgmax:
        dg = math.copysign(dgmax, dg)
    g += dg
    if abs(g) > vehicle.max_gamma:
        g = math.copysign(vehicle.max_gamma, g)
    vehicle.target_gamma = g
    vehicle.waypoint_id = iwp


def get_observations(vehicle: Vehicle, landmarks: np.ndarray, R, rng):
    """sim/sim-utils.jl:12-28,53-75.  Returns (z 2 x nz, tags 1-based)."""
    x, y, phi = vehicle.pose
    dx = landmarks[0] - x
    dy = landmarks[1] - y
    near = [i for i in range(landmarks.shape[1])
            if (dx[i] * math.cos(phi) + dy[i] * math.sin(phi)) > 0
            and (dx[i] ** 2 + dy[i] ** 2) < vehicle.sensor_range ** 2]
    near = np.asarray(near, dtype=np.int64)
    z = np.vstack([np.sqrt(dx[near] ** 2 + dy[near] ** 2),
                   np.arctan2(dy[near], dx[near]) - phi])
    if z.shape[1] > 0:
        z = z + np.vstack([rng.standard_normal(z.shape[1]) * math.sqrt(R[0, 0]),
                           rng.standard_normal(z.shape[1]) * math.sqrt(R[1, 1])])
    return z, near + 1


@dataclass
class SimLog:
    """Everything the filter was fed, in call order, plus both tracks."""
    controls: list = field(default_factory=list)        # (v, g) per predict
    obs_steps: list = field(default_factory=list)       # predict-step index of each observation step
    observations: list = field(default_factory=list)    # z (2 x nz) per observation step
    assoc: list = field(default_factory=list)           # (idf list, n_new) per observation step
    true_track: list = field(default_factory=list)
    slam_track: list = field(default_factory=list)


#: constants of sim/ekfslam-sim.jl:62-76,114
SIGMA_SPEED = 0.5
SIGMA_STEER = 3.0 * math.pi / 180
SIGMA_R = 0.1
SIGMA_B = 1.0 * math.pi / 180
DT = 0.025
DT_OBS = 8 * DT
GATE1 = 4.0
GATE2 = 25.0
D_MIN = 1.0


def default_QR():
    Q = np.array([[SIGMA_SPEED ** 2, 0.0], [0.0, SIGMA_STEER ** 2]])
    R = np.array([[SIGMA_R ** 2, 0.0], [0.0, SIGMA_B ** 2]])
    return Q, R


def sim(filt, waypoints: np.ndarray, landmarks: np.ndarray, seed: int, nlaps: int = 2,
        max_steps: int = 100000, monitor=None, fused: bool = False) -> SimLog:
    """The loop of sim/ekfslam-sim.jl:80-141 without sleep/pause.

    ``filt`` must already hold the initial state (x = initial pose, P = 0).
    ``fused``: use ``filt.observe`` (associate + update + add_features in one library call,
    same results) instead of the three calls of :114-120.
    """
    rng = np.random.default_rng(seed)
    Q, R = default_QR()
    vehicle = Vehicle(pose=initial_pose(waypoints))
    log = SimLog()
    dtsum = 0.0
    nsteps = 0
    while vehicle.waypoint_id != 0 and nsteps < max_steps:
        steer(vehicle, waypoints, D_MIN, DT)                              # :85
        if vehicle.waypoint_id == 0 and nlaps > 1:                        # :88-91
            vehicle.waypoint_id = 1
            nlaps -= 1
        step_vehicle(vehicle, DT)                                         # :94
        vehicle.measured_speed = vehicle.target_speed + rng.standard_normal() * math.sqrt(Q[0, 0])
        vehicle.measured_gamma = vehicle.target_gamma + rng.standard_normal() * math.sqrt(Q[1, 1])
        filt.predict(vehicle.measured_speed, vehicle.measured_gamma, vehicle.wheelbase, Q, DT)  # :100
        log.controls.append((vehicle.measured_speed, vehicle.measured_gamma))
        dtsum += DT                                                       # :102
        if dtsum > DT_OBS:                                                # :105 (fires every 9th step)
            dtsum = 0.0
            z, _tags = get_observations(vehicle, landmarks, R, rng)       # :108
            if fused:
                assoc = np.asarray(filt.observe(z, R, GATE1, GATE2))      # :114-120 in one call
                idf = assoc[assoc > 0]
                zn = np.asarray(z, dtype=float).reshape(2, -1)[:, assoc < 0]
            else:
                zf, idf, zn = filt.associate(z, R, GATE1, GATE2)          # :114
                filt.update(zf, R, idf)                                   # :117
                filt.add_features(zn, R)                                  # :120
            log.obs_steps.append(nsteps)
            log.observations.append(np.array(z))
            log.assoc.append((np.asarray(idf).reshape(-1).tolist(), int(np.asarray(zn).reshape(2, -1).shape[1])))
        nsteps += 1
        log.true_track.append(np.array(vehicle.pose))
        log.slam_track.append(np.array(filt.pose()))
        if monitor is not None:
            monitor(vehicle, filt, nsteps)
    return log
