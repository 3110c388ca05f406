#!/usr/bin/env python3
"""bench.py -- EKF-SLAM hot path on MI355X: associate + update steps at N = 10k landmarks.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric "EKF updates/sec @ 10k
landmarks" is quoted on): N = 10 000 landmarks at config-1 density, fp32 state
resident in HBM, nz = 64 range-bearing observations per step (the 64 nearest landmarks in the
forward half-plane, fresh noise every step), gates 4.0 / 25.0.  One step = one pass of the hot
path: associate (gating sweep over all landmarks, host gets the decisions like the
reference's caller) + update (P*H', S, C = inv(chol(S)), W1, P -= W1*W1') through the C ABI.

Unit: an "EKF update" is ONE observation assimilated; a step with m matched observations
counts m.  steps/s is reported beside it.

The EKF path does not shard (one dense coupled covariance): with --gpus N > 1 every rank runs
an independent replica on its own GPU ("replicas only", weak scaling, no data-path collective).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R = np.array([[0.1 ** 2, 0.0], [0.0, (math.pi / 180) ** 2]])
GATE1, GATE2 = 4.0, 25.0
SEED = 20240601
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md, Peak FP32 (matrix)
RED_DEVICE = "cpu" if os.environ.get("SLAM_BENCH_REHEARSE") == "1" else "cuda"     # where timing scalars are reduced
NROOF = 48                          # individually bracketed down-date launches behind the timed region (roofline.frac)
HBM_PEAK_GBPS = 8000.0                # MI355X_MICROARCH.md, HBM3E peak (6290 measured copy rate)


def make_workload(N, nz, nsteps, seed):
    """Synthetic map, state and per-step observations."""
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    L = 100.0 * math.sqrt(N / 35.0)
    lm = rng.uniform(0, L, (2, N))
    pose = np.array([L / 2, L / 2, 0.3])
    x = np.concatenate([pose, (lm + rng.normal(0, 0.1, lm.shape)).T.reshape(-1)]).astype(np.float32)
    A = rng.normal(0, 0.05, (n, 16)).astype(np.float32)
    P = A @ A.T
    P[np.diag_indices(n)] += np.float32(0.01)
    P = np.maximum(P, P.T)
    dx, dy = lm[0] - pose[0], lm[1] - pose[1]
    fwd = np.flatnonzero(dx * math.cos(pose[2]) + dy * math.sin(pose[2]) > 0)
    ids = fwd[np.argsort(dx[fwd] ** 2 + dy[fwd] ** 2)[:nz]]
    ztrue = np.vstack([np.hypot(dx[ids], dy[ids]), np.arctan2(dy[ids], dx[ids]) - pose[2]])
    zs = [ztrue + rng.normal(0, 1, ztrue.shape) * np.array([[0.1], [math.pi / 180]]) for _ in range(nsteps)]
    return x, P, zs


def make_workload_on_device(pkg, N, nz, nsteps, seed, dtype, device):
    """The same kind of workload for maps whose covariance does not fit the host comfortably (N = 50k, fp64:
    80 GB): x and the observations come from NumPy, P = A A' + 0.01 I is formed on the GPU by torch and handed to
    the library device-to-device (slam_ekf_set_state_device)."""
    import torch
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    L = 100.0 * math.sqrt(N / 35.0)
    lm = rng.uniform(0, L, (2, N))
    pose = np.array([L / 2, L / 2, 0.3])
    npdt = np.float32 if dtype == "f32" else np.float64
    x = np.concatenate([pose, (lm + rng.normal(0, 0.1, lm.shape)).T.reshape(-1)]).astype(npdt)
    tdt = torch.float32 if dtype == "f32" else torch.float64
    dev = torch.device("cuda", device)
    A = torch.from_numpy(rng.normal(0, 0.05, (n, 16)).astype(npdt)).to(dev)
    Pd = torch.empty((n, n), dtype=tdt, device=dev)
    torch.mm(A, A.t(), out=Pd)                     # symmetric by construction (same products both ways)
    Pd.diagonal().add_(0.01)
    Pd = torch.maximum(Pd, Pd.t()) if n <= 30000 else Pd
    xd = torch.from_numpy(x).to(dev)
    st = pkg.EKFSlamState(np.asarray(x[:3]), np.zeros((3, 3), dtype=npdt), dtype=dtype, max_landmarks=N, device=device)
    torch.cuda.synchronize(dev)
    st.set_state_device(xd.data_ptr(), Pd.data_ptr(), n, n)     # row-major == column-major for a symmetric matrix
    st.sync()
    del Pd, A
    torch.cuda.empty_cache()
    dx, dy = lm[0] - pose[0], lm[1] - pose[1]
    fwd = np.flatnonzero(dx * math.cos(pose[2]) + dy * math.sin(pose[2]) > 0)
    ids = fwd[np.argsort(dx[fwd] ** 2 + dy[fwd] ** 2)[:nz]]
    ztrue = np.vstack([np.hypot(dx[ids], dy[ids]), np.arctan2(dy[ids], dx[ids]) - pose[2]])
    zs = [ztrue + rng.normal(0, 1, ztrue.shape) * np.array([[0.1], [math.pi / 180]]) for _ in range(nsteps)]
    return st, zs


def gpu_step(st, z):
    """associate -> (host splits the decisions, as sim! does) -> update.  Returns matched count."""
    a = st.associate_vector(z, R, GATE1, GATE2)
    sel = a > 0
    m = int(sel.sum())
    if m:
        st.update(z[:, sel], R, a[sel])
    return m


def cpu_baseline(x, P, zs, budget_s=20.0):
    """The oracle's sparse restatement (vectorised NumPy / BLAS, fp64) on the host cores, same
    workload, bounded sample: whole steps until ~budget_s of CPU time is spent."""
    from oracle import ekf_ref as O
    threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)     # a 1-GPU box's CPU share is 16
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=threads):
            return _cpu_baseline(O, x, P, zs, budget_s, threads)
    except ImportError:
        return _cpu_baseline(O, x, P, zs, budget_s, os.cpu_count())


def _cpu_baseline(O, x, P, zs, budget_s, threads):
    xo = x.astype(np.float64)
    Po = P.astype(np.float64)
    done = matched = 0
    t0 = time.perf_counter()
    for z in zs:
        zf, idf, _zn = O.associate_sparse(xo, Po, z, R, GATE1, GATE2)
        xo, Po = O.update_sparse(xo, Po, zf, R, idf, inplace=True)
        matched += idf.shape[1]
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": matched / dt, "unit": "obs-updates/s", "cores": threads, "kind": "port",
            "steps_per_s": done / dt,
            "sample": f"{done} full associate+update steps of the same N={len(x) // 2 - 1} nz={zs[0].shape[1]} "
                      f"workload in fp64 (oracle sparse restatement: vectorised 5x5 gating + BLAS rank-k "
                      f"down-date, {threads} BLAS threads), {dt:.1f} s"}


class FastslamParityError(RuntimeError):
    """The sharded filter disagrees with the one-rank filter on this node: the FastSLAM leg reports it and the process exits 3."""


EXIT_CODE = [0]


def _ulps(a, b):
    """Largest distance of two float32 arrays in units in the last place (of the larger magnitude)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    ok = np.isfinite(a) & np.isfinite(b)
    if not ok.all() and not np.array_equal(a[~ok], b[~ok], equal_nan=True):
        return float("inf")
    return float(np.abs(ia[ok] - ib[ok]).max()) if ok.any() else 0.0


def fastslam_parity(pkg, world, rank, local_rank, NP, NL, lm, Q, obs, peers):
    """before anything is timed on several ranks, the SHARDED filter is checked against a ONE-RANK filter on
    the node it runs on: 8 steps (Neff rule, one forced resampling, one forbidden one) on both, then rank 0 compares the first
    particles of its slice -- poses, log-weights -- and Neff / the resampling count.  Device-side exchange: bit for bit (the
    canonical statistics tree).  Halting flow over the collectives: poses bit for bit, log-weights within 4 ulp (the ranks' roots
    are combined in rank order there).  Returns the record for comm.parity_vs_one_rank; raises FastslamParityError on a mismatch
    (on every rank)."""
    import torch.distributed as dist
    pf = pkg.PFSlamState(NP, NL, seed=20240602, dtype="f32", device=local_rank, distributed=True, peers=peers)
    one = None
    try:
        got_peers, selftest = bool(pf.peers), pf.selftest_ok
        if peers and not got_peers:
            pf.close()
            return {"available": False, "selftest_ok": selftest, "note": "the peers could not be attached on this node (or the self-test failed): no device-side exchange"}
        if rank == 0:
            one = pkg.PFSlamState(NP, NL, seed=20240602, dtype="f32", device=local_rank, distributed=False)
        forces = [None, None, True, None, False, None, None, None]
        Qs, Rs = pkg.small(Q), pkg.small(R)
        for f in ([pf] + ([one] if one is not None else [])):
            f.shard.set_pose([0.0, 0.0, 0.3])
            f.shard.init_landmarks(lm, 0.01, 0.1)
        for t, force in enumerate(forces):
            z, ids = obs[t]
            pf.step_async(8.0, 0.0, 4.0, Qs, 0.025, z, ids, Rs, force_resample=force)
            if one is not None:
                one.step_async(8.0, 0.0, 4.0, Qs, 0.025, z, ids, Rs, force_resample=force)
        neff_sh, _ = pf.flush()
        rec = None
        if rank == 0:
            neff_one, _ = one.flush()
            nw = min(4096, NP // world)
            ps, ws, _ = pf.shard.download(landmarks=False)
            po, wo, _ = one.shard.download(landmarks=False)
            poses_equal = bool(np.array_equal(ps[:, :nw], po[:, :nw]))
            ulp = _ulps(ws[:nw], wo[:nw])
            rec = {"available": True, "selftest_ok": selftest, "steps": len(forces), "particles_compared": int(nw), "poses_equal": poses_equal,
                   "logw_max_ulp": ulp, "neff": [neff_sh, neff_one], "neff_rel_diff": abs(neff_sh - neff_one) / max(abs(neff_one), 1e-300),
                   "resamples": [int(pf.resamples), int(one.resamples)]}
            tol_ulp, tol_neff = (0.0, 0.0) if got_peers else (4.0, 1e-12)
            rec["ok"] = bool(poses_equal and ulp <= tol_ulp and rec["neff_rel_diff"] <= tol_neff and pf.resamples == one.resamples and pf.resamples >= 1)
        box = [rec]
        dist.broadcast_object_list(box, src=0)
        rec = box[0]
    finally:
        if one is not None:
            one.close()
    pf.close()
    if not rec["ok"]:
        raise FastslamParityError(f"the sharded filter ({'device-side exchange' if peers else 'halting flow over the collectives'}) disagrees with "
                                  f"the one-rank filter on this node: {json.dumps(rec)}")
    return rec


def bench_fastslam(pkg, world, rank, local_rank, steps, warmup, fence):
    """BASELINE.json config 4: FastSLAM-1.0, 262144 particles x 512 landmarks, 16 known-id observations per
    step, fp32, particles sharded over the ranks (weak scaling is NOT used here: the particle count is
    fixed, so this sub-metric is strong scaling).  Returns the sub-object for the JSON line.
    Several ranks: the sharded filter is first CHECKED against a one-rank filter on this node
    (fastslam_parity: comm.parity_vs_one_rank, comm.selftest_ok; a mismatch is an error and exit code 3), then BOTH exchange
    paths are timed in the same run -- regimes_peers (device-side: inboxes and peer reads over IPC mappings) and regimes_rccl
    (the halting flow: all_reduce of the three scalars, all_gather of the log-weight slices, the record exchange through
    torch.distributed = RCCL on the GPUs) -- so that the first run on a multi-GPU node yields a comparison and survives a
    failure of either."""
    import torch
    import torch.distributed as dist
    NP, NL, M = 262144, 512, 16
    if os.environ.get("SLAM_BENCH_REHEARSE") == "1" and os.environ.get("SLAM_BENCH_NP"):
        NP = int(os.environ["SLAM_BENCH_NP"])        # one-card rehearsal only (e.g. 6 ranks x 32768: the shard size of 8 GPUs)
    Q = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
    rng = np.random.default_rng(20240602)                      # same scene and observations on every rank
    lm = rng.uniform(-200, 200, (NL, 2))
    pose = np.array([0.0, 0.0, 0.3])
    obs, poses = [], [pose.copy()]
    for t in range(5 * (steps + warmup) + 8):
        pose = np.array([pose[0] + 0.2 * math.cos(pose[2]), pose[1] + 0.2 * math.sin(pose[2]), pose[2]])
        ids = (np.arange(M) + M * t) % NL + 1
        dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, M))
        obs.append((z, ids))
        poses.append(pose.copy())                # poses[t]: where the vehicle is BEFORE step t
    ctx = dict(pkg=pkg, world=world, rank=rank, local_rank=local_rank, steps=steps, warmup=warmup, fence=fence, NP=NP, NL=NL, M=M,
               Q=Q, lm=lm, obs=obs, poses=poses)
    parity, paths = None, {}
    if world > 1:
        parity = {}
        for name, peers in (("peers", True), ("rccl", False)):
            parity[name] = fastslam_parity(pkg, world, rank, local_rank, NP, NL, lm, Q, obs, peers)
        peers_ok = bool(parity["peers"].get("available"))
        if peers_ok:
            paths["peers"] = _fastslam_regimes(ctx, peers=True, full=True)
        paths["rccl"] = _fastslam_regimes(ctx, peers=False, full=not peers_ok)
        main = paths["peers"] if peers_ok else paths["rccl"]
    else:
        main = _fastslam_regimes(ctx, peers=None, full=True)
    res, comm = main["regimes"], main["comm"]
    if world > 1:
        comm["parity_vs_one_rank"] = parity
        comm["selftest_ok"] = parity["peers"].get("selftest_ok")
        comm["timed_paths"] = sorted(paths)
    weak = _fastslam_weak(ctx) if world > 1 else None
    bytes_per = 24 + 8 + M * 40             # pose r/w + log-weight r/w + 5 floats read and written per observed landmark
    t_step = res["no_resample"]["ms_per_step"] * 1e-3
    out = {"metric": "FastSLAM particle-steps/sec", "value": res["neff_triggered"]["particle_steps_per_s"],
           "unit": "particle-steps/s", "n_gpus": world, "scaling": "strong",
           "config": {"workload": f"FastSLAM-1.0 known correspondences, {NP} particles x {NL} landmarks, {M} obs/step, fp32, "
                                  f"predict + {M} 2x2 EKF updates + weights + Neff all-reduce (+ resample when Neff < 0.75 Np)"},
           "regimes": res, "weak_scaling": weak, "comm": comm,
           "resampling": ("decided and done on the device, lazily (poses permuted, ancestor tables composed, maps moved on "
                          "their next update)" if world == 1 else
                          "decided and done on the device on every rank: cdf over all ranks' weights (read from the peers' "
                          "buffers), global ancestors, remote poses / table entries read from their owners, maps stay put (an "
                          "ancestor-table entry is a global particle id; a remote record is read when its landmark is next updated)"
                          if comm["peers_attached"] else
                          "decided on the device (scalars exchanged GPU to GPU through a pinned page); a resampling step halts "
                          "the queue, the hosts all-gather the log-weights and exchange records, then resume"),
           "roofline": {"bound": "hbm", "achieved": NP * bytes_per / t_step / 1e9, "peak": HBM_PEAK_GBPS * world,
                        "unit": "GB/s", "frac": NP * bytes_per / t_step / 1e9 / (HBM_PEAK_GBPS * world), "traffic": None,
                        "algorithmic_bytes_per_particle_step": bytes_per, "regime": "no_resample: per step ONE sweep kernel "
                        "(statistics folded, Neff and the resampling decision taken by its last workgroup) + two "
                        "conditional no-op launches, nothing read back by the host"}}
    if world > 1:
        out["regimes_peers"] = paths["peers"]["regimes"] if "peers" in paths else {"unavailable": parity["peers"].get("note")}
        out["regimes_rccl"] = paths["rccl"]["regimes"]
        out["comm_rccl"] = paths["rccl"]["comm"]
    return out


def _fastslam_regimes(ctx, peers, full):
    """The timed regimes on ONE filter (created here, closed here).  peers: None (one rank), True (device-side exchange), False (the
    halting flow through torch.distributed).  full: all regimes; else the three that BASELINE.json's metric is about."""
    import gc
    import torch
    import torch.distributed as dist
    pkg, world, rank, local_rank = ctx["pkg"], ctx["world"], ctx["rank"], ctx["local_rank"]
    steps, warmup, fence, NP, NL, M = ctx["steps"], ctx["warmup"], ctx["fence"], ctx["NP"], ctx["NL"], ctx["M"]
    Q, lm, obs, poses = ctx["Q"], ctx["lm"], ctx["obs"], ctx["poses"]
    pf = pkg.PFSlamState(NP, NL, seed=20240602, dtype="f32", device=local_rank, distributed=world > 1, peers=peers)
    try:
        pf.shard.set_pose([0.0, 0.0, 0.3])
        pf.shard.init_landmarks(lm, 0.01, 0.1)
        res = {}
        trace = os.environ.get("SLAM_BENCH_TRACE") == "1" and rank == 0       # progress lines on stderr (diagnosing a slow rehearsal)
        t_trace = time.perf_counter()

        def say(msg):
            if trace:
                print(f"[fastslam +{time.perf_counter() - t_trace:7.2f} s] {msg}", file=sys.stderr, flush=True)
        n_align = 300 if os.environ.get("SLAM_BENCH_REHEARSE") != "1" else 20    # (a one-card rehearsal only checks the plumbing)
        if peers is False:
            n_align = min(n_align, 40)         # (the halting flow: a resampling step costs milliseconds of host work)
        say(f"filter created, world {world}, peers {pf.shard.comm_info()}")
        fence()                                # the ranks start their (device-side) scalar exchange together
        # observations converted once, outside the timed regions: a timed step is one library call
        prep = [pkg.PFShard.prepare_obs(z, ids) for z, ids in obs]
        Qs, Rs = pkg.small(Q), pkg.small(R)
        KB = 16                                # steps per slam_pf_step_auto_batch call of the batched regime

        def run(k, force, prop, mode, V=8.0):
            z, ids = obs[k]
            if mode == "async":                # slam_pf_step_auto: enqueued; statistics, Neff, decision, resampling on the device
                pf.step_async(V, 0.0, 4.0, Qs, 0.025, z, ids, Rs, force_resample=force, proposal=prop, prepared=prep[k])
            else:                              # the host decides after every step (slam_pf_step + read-back)
                pf.step(V, 0.0, 4.0, Q, 0.025, z, ids, R, force_resample=force, proposal=prop)

        def run_many(k0, count, force, prop, mode, V=8.0):
            if mode != "batch":
                for k in range(k0, k0 + count):
                    run(k, force, prop, mode, V)
                return
            for b0 in range(k0, k0 + count, KB):       # slam_pf_step_auto_batch: runs of steps that cannot resample as one launch
                kk = range(b0, min(b0 + KB, k0 + count))
                batch = pkg.PFShard.prepare_batch([(V, 0.0)] * len(kk), [obs[k] for k in kk], force)
                pf.step_async_batch(batch, 4.0, Qs, 0.025, Rs, persistent=True)     # (the bench has the device to itself)

        def fresh(k):
            """Every regime starts from a CLEAN filter at the vehicle's pose before step k: every particle there, the map at
            truth + jitter, uniform weights (a regime that never resamples must not inherit -- or hand on -- a degenerate
            particle set; with 200+ steps per regime the fifth regime used to start from non-finite weights)."""
            pf.shard.set_pose(poses[k])
            pf.shard.init_landmarks(lm, 0.01, 0.1)

        # (the fourth regime is the FastSLAM-2.0 step of SURVEY 8f N4: the pose drawn from the observation-aware proposal; the fifth
        #  is the first one with the host back in the loop, for comparison; the sixth is the first one through
        #  slam_pf_step_auto_batch: up to 16 steps per persistent launch -- one rank only, the sharded filter takes them one by one)
        regimes = [("no_resample", False, False, "async"), ("every_step", True, False, "async"), ("neff_triggered", None, False, "async")]
        if full:
            regimes += [("proposal_no_resample", False, True, "async"), ("no_resample_host_in_loop", False, False, "sync")]
            if world == 1:
                regimes += [("no_resample_batched", False, False, "batch")]
        for idx, (regime, force, prop, mode) in enumerate(regimes):
            gc.collect()                       # parked until the end of the timed region (see main)
            gc.disable()
            # untimed device warm-up in the regime's own mode (a GPU out of idle needs ~40 ms of load to reach its
            # sustained clocks): the vehicle stands still at the regime's first observation (V = 0: same kernels, and the
            # particles stay where the observations are); a fixed count keeps the ranks' exchanges aligned
            k0 = (steps + warmup) * min(idx, 4)
            fresh(k0)
            t_pw = time.perf_counter()
            while world == 1 and time.perf_counter() - t_pw < 0.1:
                run_many(k0, KB if mode == "batch" else 1, force, prop, mode, V=0.0)
            say(f"regime {regime}: warm-up")
            for j in range(n_align if world > 1 else 0):
                run(k0, force, prop, mode, V=0.0)
            if mode != "sync":
                pf.flush()
            fresh(k0)
            run_many(k0, warmup, force, prop, mode)
            if mode != "sync":
                pf.flush()
            say(f"regime {regime}: timed region")
            pf.shard.sync()
            fence()
            n0 = pf.resamples
            t0 = time.perf_counter()
            run_many(k0 + warmup, steps, force, prop, mode)
            if mode != "sync":
                pf.flush()
            pf.shard.sync()
            fence()
            el = time.perf_counter() - t0
            gc.enable()
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device=RED_DEVICE)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            res[regime] = {"particle_steps_per_s": NP * steps / el, "ms_per_step": el / steps * 1e3, "resamples": pf.resamples - n0}
            say(f"regime {regime}: {el / steps * 1e3:.3f} ms per step")
        say("regimes done; comm_info")
        info = pf.shard.comm_info()            # what the exchange between the ranks saw: peers attached?  steps that halted for the host?
        halts = info["halts"]
        if world > 1:
            th = torch.tensor([float(halts)], dtype=torch.float64, device=RED_DEVICE)
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            halts = int(th.item())
        comm = {"world": world, "halts": halts, "peers_attached": bool(info["peers"]),
                "backend": ("single GPU: no exchange" if world == 1 else
                            "device-side: per-step scalars written into the peers' inboxes and the resampling's reads of the peers' "
                            "weights / poses / ancestor tables / records go over IPC-mapped buffers (xGMI between GPUs), no collective "
                            "launch, no host" if info["peers"] else
                            "halting flow: scalars through a pinned host page, a resampling step halts and the hosts resample through "
                            f"torch.distributed ({dist.get_backend()}): all_reduce of (max, sum w, sum w^2), all_gather of the log-weight slices, "
                            "all_to_all of the migrating records"),
                "control_plane": None if world == 1 else f"torch.distributed ({dist.get_backend()}): set-up (object all-gather of the peer blobs) and timing only"}
        say(f"closing the filter ({comm['halts']} halts)")
        pf.close()
        say("filter closed")
        return {"regimes": res, "comm": comm}
    except BaseException:
        # never leave an attached shard to the garbage collector: the orderly close is a collective (detach, barrier) and the
        # other ranks may be anywhere -- destroy THIS shard alone; slam_pf_destroy tells the peers first (their queued steps
        # then fail with "a peer is gone" instead of reading freed memory)
        try:
            pf.shard.close()
        except Exception:  # noqa: BLE001
            pass
        raise


def _fastslam_weak(ctx):
    """The same filter with the per-GPU particle count held at 262144 (weak scaling): the strong-scaling figure divides ~45 us of
    sweep per step by N and leaves the per-step exchange latency."""
    import torch
    import torch.distributed as dist
    pkg, world, local_rank = ctx["pkg"], ctx["world"], ctx["local_rank"]
    steps, fence, NP, NL = ctx["steps"], ctx["fence"], ctx["NP"], ctx["NL"]
    Q, lm, obs, poses = ctx["Q"], ctx["lm"], ctx["obs"], ctx["poses"]
    n_align = 300 if os.environ.get("SLAM_BENCH_REHEARSE") != "1" else 20
    prep = [pkg.PFShard.prepare_obs(z, ids) for z, ids in obs[:max(steps, 1)]]
    Qs, Rs = pkg.small(Q), pkg.small(R)
    pfw = pkg.PFSlamState(NP * world, NL, seed=20240602, dtype="f32", device=local_rank, distributed=True)
    try:
        pfw.shard.set_pose([0.0, 0.0, 0.3])
        pfw.shard.init_landmarks(lm, 0.01, 0.1)
        for j in range(n_align):
            pfw.step_async(0.0, 0.0, 4.0, Qs, 0.025, *obs[0], Rs, force_resample=False, prepared=prep[0])
        pfw.flush()
        pfw.shard.set_pose(poses[0])
        pfw.shard.init_landmarks(lm, 0.01, 0.1)
        pfw.shard.sync()
        fence()
        t0 = time.perf_counter()
        for j in range(steps):
            pfw.step_async(8.0, 0.0, 4.0, Qs, 0.025, *obs[j], Rs, force_resample=False, prepared=prep[j])
        pfw.flush()
        pfw.shard.sync()
        fence()
        el = time.perf_counter() - t0
        tt = torch.tensor([el], dtype=torch.float64, device=RED_DEVICE)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        weak = {"particles": NP * world, "particle_steps_per_s": NP * world * steps / el, "ms_per_step": el / steps * 1e3,
                "regime": "no_resample, 262144 particles per GPU", "peers_attached": bool(pfw.shard.comm_info()["peers"])}
        pfw.close()
        return weak
    except BaseException:                      # (as above: destroy this shard alone, then report)
        try:
            pfw.shard.close()
        except Exception:  # noqa: BLE001
            pass
        raise


def fastslam_guarded(out_partial, rank, world, *a):
    """bench_fastslam under a deadline and an exception guard.  The EKF headline never depends on the FastSLAM leg: a
    failure comes back as {"error": ...}; a leg that does not come back at all (SLAM_BENCH_PF_BUDGET_S, default 600 s
    on one rank, 420 s on several) makes rank 0 print the line without it and every rank leave."""
    budget = float(os.environ.get("SLAM_BENCH_PF_BUDGET_S", "600" if world == 1 else "420"))
    lock = threading.Lock()
    state = {"done": False}

    def expire():
        with lock:
            if state["done"]:
                return
            if rank == 0 and out_partial is not None:
                out_partial["fastslam"] = {"error": f"the FastSLAM leg did not finish within {budget:.0f} s on {world} rank(s): abandoned"}
                print(json.dumps(out_partial), flush=True)
            os._exit(3)                    # a hang is a failure: the line (with fastslam.error) is out, the exit code says so

    timer = threading.Timer(budget, expire)
    timer.daemon = True
    timer.start()
    try:
        fast = bench_fastslam(*a)
    except FastslamParityError as e:       # a WRONG sharded filter: the line says so and the process leaves with exit code 3
        fast = {"error": f"{type(e).__name__}: {e}"}
        EXIT_CODE[0] = 3
    except Exception as e:  # noqa: BLE001 -- reported in the line, never fatal for the headline
        fast = {"error": f"{type(e).__name__}: {e}"}
    with lock:
        state["done"] = True
    timer.cancel()
    return fast


def measure_traffic(args, want_fastslam, landmarks=None, obs=None, dtype=None, form=None):
    """HBM-side traffic of the dominant kernels, MEASURED in this run: two child passes of this script under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots), started
    BEFORE this process touches the GPU (children are plain subprocesses).  Returns {kernel: {"fetch_kb": ..,
    "write_kb": .., "launches": ..}} with the per-launch means, or {"error": ...}.  gfx950 corrections as the guide's
    HBM section prescribes: FETCH_SIZE counts 64 B per 128-B request (x2), WRITE_SIZE is exact; unit KB = 1024 B."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not found"}
    names = {"downdate_f32_mfma": "downdate", "downdate_f64_mfma": "downdate", "pf_auto_step_kernel": "pf_step"}
    out = {}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pmc", "--no-configs",
             "--prewarm-ms", "20", "--landmarks", str(landmarks or args.landmarks), "--obs", str(obs or args.obs),
             "--dtype", dtype or args.dtype, "--form", form or args.form] + ([] if want_fastslam else ["--no-fastslam"])
    for counter, key in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb")):
        d = tempfile.mkdtemp(prefix="slam_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", SLAM_BENCH_CHILD="1"), capture_output=True,
                               text=True, timeout=400)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter} failed (exit {r.returncode}): {r.stderr[-300:]}"}
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != counter:
                    continue
                short = next((v for k, v in names.items() if k in row["Kernel_Name"]), None)
                if short:
                    acc.setdefault(short, []).append(float(row["Counter_Value"]))
            for short, vals in acc.items():
                out.setdefault(short, {})[key] = sum(vals) / len(vals)
                out[short]["launches"] = len(vals)
        except Exception as e:                                  # the bench line must still come out
            return {"error": f"{type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def hbm_bytes(rec):
    """FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, KB of 1024 B -> bytes per launch."""
    if not rec or "fetch_kb" not in rec or "write_kb" not in rec:
        return None
    return (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0


def literal_cpu_legs(budget_s=12.0, landmarks=10000, nobs=64):
    """the LITERAL dense restatement (oracle/ekf_ref.py: dense 2 x n Jacobians, dense H*P*H' per pair, as
    the reference computes) timed at C2 (N = 1000, 16 observations, fp64) on a bounded sample -- one observation against
    all 1000 landmarks for the association, one dense batched update -- and the probe for a `julia` binary."""
    import shutil
    from oracle import ekf_ref as O
    x, P, zs = make_workload(1000, 16, 1, SEED)
    xo, Po = x.astype(np.float64), P.astype(np.float64)
    z = zs[0]
    t0 = time.perf_counter()
    pairs = 0
    for j in range(1, 1001):
        O.compute_association(xo, Po, z[:, 0], R, j)
        pairs += 1
        if time.perf_counter() - t0 > budget_s:
            break
    t_pair = (time.perf_counter() - t0) / pairs
    zf, idf, _zn = O.associate_sparse(xo, Po, z, R, GATE1, GATE2)
    t1 = time.perf_counter()
    O.update(xo, Po, zf, R, idf)
    t_upd = time.perf_counter() - t1
    step = 16 * 1000 * t_pair + t_upd
    legs = [{"kind": "literal", "value": idf.shape[1] / step, "unit": "obs-updates/s",
             "cores": min(os.cpu_count() or 1, len(os.sched_getaffinity(0))),
             "sample": f"C2 (N=1000, 16 obs, fp64): {pairs} dense compute_association pairs timed ({t_pair * 1e3:.2f} ms each, "
                       f"extrapolated to the 16000 of a step) + one dense batched update ({t_upd * 1e3:.0f} ms); the literal form is "
                       f"O(nz N n^2) and is not timed at N = 10k"}]
    jl = shutil.which("julia")
    leg = {"kind": "julia", "available": bool(jl), "path": jl}
    if not jl:
        leg["note"] = ("no julia binary on this node.  oracle/ekf_ref.jl (a Julia-1.x restatement of src/ekf.jl, "
                       "src/data-association.jl and the observation model; the reference's own 0.5/0.6 sources do not parse on "
                       "Julia >= 1.0) would be timed here: `julia oracle/ekf_ref.jl --bench N nz seconds`")
    else:
        # the Julia restatement on this node's host cores, same workload shape, bounded to ~15 s
        import subprocess
        env = dict(os.environ, JULIA_NUM_THREADS=str(min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))))
        try:
            r = subprocess.run([jl, os.path.join(ROOT, "oracle", "ekf_ref.jl"), "--bench", str(landmarks), str(nobs), "15"],
                               capture_output=True, text=True, timeout=600, env=env)
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
            if r.returncode == 0 and line:
                leg.update(json.loads(line))
            else:
                leg["note"] = "oracle/ekf_ref.jl failed: " + (r.stderr or r.stdout)[-400:]
        except Exception as e:  # noqa: BLE001 -- a reported baseline, never fatal
            leg["note"] = f"oracle/ekf_ref.jl could not be run: {e}"
    legs.append(leg)
    return legs

OTHER_CONFIGS = (
    # BASELINE.json configs[1] and configs[4]: driver-timed beside the headline (configs[2]) in every default run
    {"name": "C2", "landmarks": 1000, "obs": 16, "dtype": "f32", "form": "cholesky", "cpu_budget": 6.0},
    {"name": "C5", "landmarks": 50000, "obs": 8, "dtype": "f64", "form": "joseph", "cpu_budget": 0.0},
)


def config_leg(pkg, cfg, steps, warmup, local_rank, pmc_rec):
    """One of BASELINE.json's other single-GPU EKF configurations, timed like the headline: W warm-up steps, K timed
    slam_ekf_observe steps between synchronisations, the down-date bracketed by HIP events on every 4th of them.  Returns
    the sub-object for `configs` (ms_per_step, value, roofline with the in-run traffic of this configuration's own
    rocprofv3 --pmc child passes, and -- where the host can hold the covariance -- cpu_baseline)."""
    import gc
    N, nz, dtype, form = cfg["landmarks"], cfg["obs"], cfg["dtype"], cfg["form"]
    n = 3 + 2 * N
    total = warmup + steps
    big = n > 30000
    if big:
        st, zs = make_workload_on_device(pkg, N, nz, total, SEED, dtype, local_rank)
        x = P = None
    else:
        x, P, zs = make_workload(N, nz, total, SEED)
        st = pkg.EKFSlamState(x, P, dtype=dtype, max_landmarks=N, device=local_rank)
    try:
        return _config_leg_body(st, x, P, zs, cfg, N, nz, n, dtype, form, steps, warmup, total, big, pmc_rec)
    finally:
        st.close()                         # (also on an exception: C5's 40 GB must not wait for the garbage collector -- ADVICE r4)


def _config_leg_body(st, x, P, zs, cfg, N, nz, n, dtype, form, steps, warmup, total, big, pmc_rec):
    import gc
    st.set_async(True)
    gc.collect()
    gc.disable()
    try:
        rng_pw = np.random.default_rng(SEED + 7919)
        t_pw = time.perf_counter()
        k = 0
        while (time.perf_counter() - t_pw) < 0.05:              # device warm-up, as the headline's (shorter: the GPU is warm)
            st.observe(zs[k % len(zs)] + rng_pw.normal(0, 1, zs[0].shape) * np.array([[0.02], [0.2 * math.pi / 180]]), R, GATE1, GATE2, form=form)
            k += 1
            if k % 16 == 0:
                st.sync()
        st.sync()
        for i in range(warmup):
            st.observe(zs[i], R, GATE1, GATE2, form=form)
        st.sync()
        st.timing_reset()
        stride = 4
        matched = matched_timed = 0
        t0 = time.perf_counter()
        for i in range(warmup, total):
            timed = (i - warmup) % stride == 0
            st.timing(timed, kernels=["syrk"])
            a = st.observe(zs[i], R, GATE1, GATE2, form=form)
            mi = int((a > 0).sum())
            matched += mi
            if timed:
                matched_timed += mi
        st.sync()
        el = time.perf_counter() - t0
    finally:
        gc.enable()
    tim = st.timing_read()
    st.timing(True)
    st.timing_reset()
    nd = min(5, total)
    for i in range(nd):
        st.observe(zs[total - 1 - i], R, GATE1, GATE2, form=form)
    st.sync()
    tim_all = st.timing_read()
    st.timing(False)
    # (as the headline: NROOF more steps with the down-date bracketed on every one; roofline.frac from their mean)
    st.timing(True, kernels=["syrk"])
    st.timing_reset()
    rng_rf = np.random.default_rng(SEED + 104729)
    for i in range(NROOF):
        st.observe(zs[(warmup + i) % total] + rng_rf.normal(0, 1, zs[0].shape) * np.array([[0.02], [0.2 * math.pi / 180]]), R, GATE1, GATE2, form=form)
    st.sync()
    roof_n, roof_mean_ms, roof_sd_ms, _roof_min = st.timing_stats("syrk")
    st.timing(False)
    try:
        floor_ms, _form = st.copy_floor(5)
    except Exception:  # noqa: BLE001
        floor_ms = None
    esz = 4 if dtype == "f32" else 8
    syrk_ms, syrk_n = tim["syrk"]
    t_region = syrk_ms / max(syrk_n, 1) * 1e-3
    t_dd = roof_mean_ms * 1e-3 if roof_n > 0 and roof_mean_ms > 0 else t_region
    k_avg = 2.0 * matched_timed / max(syrk_n, 1) * (2.0 if form == "joseph" else 1.0)
    alg_bytes = 1.0 * n * n * esz
    alg_flops = 1.0 * n * n * k_avg
    gbps = alg_bytes / t_dd / 1e9 if t_dd > 0 else 0.0
    traffic = hbm_bytes(pmc_rec)
    out = {"workload": f"EKF-SLAM observation step, N={N} landmarks (n={n}), nz={nz} obs/step, {dtype}, {form} form, state resident in HBM",
           "value": matched / el, "unit": "obs-updates/s", "ms_per_step": el / steps * 1e3, "steps_per_s": steps / el,
           "steps": steps, "warmup": warmup, "matched_per_step": matched / steps, "dtype": dtype,
           "roofline": {"kernel": "downdate (P -= X*Y')", "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": t_dd * 1e3, "launches": roof_n if roof_n > 0 else syrk_n,
                        "avg_launch_ms_48": roof_mean_ms, "stdev_ms": roof_sd_ms, "launches_48": roof_n,
                        "avg_launch_ms_timed_region": t_region * 1e3, "launches_timed_region": syrk_n,
                        "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": alg_flops,
                        "copy_floor_ms": floor_ms, "kernel_over_floor": (t_dd * 1e3 / floor_ms) if floor_ms else None,
                        "traffic_note": ("FETCH_SIZE x 2 + WRITE_SIZE per launch from this configuration's own two rocprofv3 --pmc child "
                                         f"passes ({(pmc_rec or {}).get('launches', 0)} launches)") if traffic is not None else "not measured"},
           "kernel_ms_per_step": {kk: v[0] / max(nd, 1) for kk, v in tim_all.items()}}
    if N <= 2000:
        out["roofline"]["note"] = ("a launch of ~11 us: the covariance (16 MB) fits the Infinity Cache and the kernel sits at its latency "
                                   "floor; the step is six dependent kernels of 5-19 us")
    if cfg.get("cpu_budget", 0) > 0 and not big:
        out["cpu_baseline"] = cpu_baseline(x, P, zs, cfg["cpu_budget"])
    return out


def configs_guarded(pkg, steps, warmup, local_rank, pmc_by_cfg, budget_s=150.0):
    """C2 and C5 after the headline, each under an exception guard and a shared wall-clock budget: whatever happens here,
    the headline line still comes out."""
    res = {}
    t0 = time.perf_counter()
    for cfg in OTHER_CONFIGS:
        if time.perf_counter() - t0 > budget_s:
            res[cfg["name"]] = {"error": f"skipped: the configs leg's budget of {budget_s:.0f} s was spent"}
            continue
        try:
            res[cfg["name"]] = config_leg(pkg, cfg, steps, warmup, local_rank, (pmc_by_cfg or {}).get(cfg["name"]))
        except Exception as e:  # noqa: BLE001 -- reported in the line
            res[cfg["name"]] = {"error": f"{type(e).__name__}: {e}"}
    return res


def spawn_ranks(n):
    """One child process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the same command line; children
    are plain subprocesses (never an exec of this process).  Returns the exit code for the parent."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # a rank that dies must not leave the others waiting in a collective: end them (exact PIDs) as soon as one fails
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    out = procs[0].stdout.read() if procs[0].stdout else ""       # (one JSON line: far below the pipe's capacity)
    codes = [p.wait() for p in procs]
    if out:
        sys.stdout.write(out)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 3 if any(c == 3 for _r, c in bad) else 1      # 3: a watchdog fired (the FastSLAM leg or the final barrier hung)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--landmarks", type=int, default=10000)
    ap.add_argument("--obs", type=int, default=64)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--form", default="cholesky", choices=["cholesky", "joseph"])
    ap.add_argument("--unfused", action="store_true", help="three library calls per step instead of slam_ekf_observe")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fastslam", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C5 sub-benchmarks (`configs` in the line)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed device warm-up before the W warm-up steps: a GPU coming out of idle needs ~40 ms of load "
                         "to reach its sustained clocks (tools/step_trend.py: 0.63 -> 0.54 ms per step over the first 60 steps)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, fresh children, nothing in
        # this process has touched the GPU), relay rank 0's JSON line, fail if any rank fails.  Under
        # torch.distributed.run the ranks already exist (WORLD_SIZE is set) and this is skipped.
        sys.exit(spawn_ranks(args.gpus))

    # roofline.traffic is measured in this run: two child passes under rocprofv3 --pmc, before this process touches the GPU
    pmc = None
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if under_profiler:                 # (the profiler's library has initialised the GPU already: no child processes from here)
        pmc = {"error": "this run is itself under rocprofv3"}
    elif not args.no_pmc and "WORLD_SIZE" not in os.environ and args.gpus == 1:
        pmc = measure_traffic(args, not args.no_fastslam)
    headline_cfg = args.landmarks == 10000 and args.obs == 64 and args.dtype == "f32" and args.form == "cholesky"
    want_configs = headline_cfg and not args.no_configs and "WORLD_SIZE" not in os.environ and args.gpus == 1
    pmc_cfg = {}
    if want_configs and pmc is not None and "error" not in pmc and not args.no_pmc:
        for cfg in OTHER_CONFIGS:                                 # (each configuration's own two --pmc child passes, still before this process touches the GPU)
            rec = measure_traffic(args, False, cfg["landmarks"], cfg["obs"], cfg["dtype"], cfg["form"])
            pmc_cfg[cfg["name"]] = rec.get("downdate") if isinstance(rec, dict) else None

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("SLAM_BENCH_REHEARSE") == "1"    # several ranks on ONE card over gloo: plumbing check only
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    from __graft_entry__ import load_package
    pkg = load_package()

    N, nz = args.landmarks, args.obs
    n = 3 + 2 * N
    total_steps = args.warmup + args.steps
    big = n > 30000                                                # covariance built on the device (see above)
    if big:
        st, zs = make_workload_on_device(pkg, N, nz, total_steps, SEED + rank, args.dtype, local_rank)
        x = P = None
        args.no_cpu_baseline = True
    else:
        x, P, zs = make_workload(N, nz, total_steps, SEED + rank)  # every replica gets its own noise
        st = pkg.EKFSlamState(x, P, dtype=args.dtype, max_landmarks=N, device=local_rank)

    def step(z):
        if args.unfused:                   # the reference's three calls, decisions split on the host (sim! :114-120)
            a = st.associate_vector(z, R, GATE1, GATE2)
            sel = a > 0
            if sel.any():
                st.update(z[:, sel], R, a[sel], form=args.form)
            if (a < 0).any():
                st.add_features(z[:, a < 0], R)
        else:                              # the same step as ONE library call (slam_ekf_observe)
            a = st.observe(z, R, GATE1, GATE2, form=args.form)
        return int((a > 0).sum())

    st.set_async(True)                     # the update's status is collected at the final sync
    # The cyclic garbage collector is parked from here to the end of the timed region: a full collection of the
    # interpreter's ~10^5 objects (torch is imported) takes 40 ms, lands deterministically on one step
    # (tools/step_trend.py: step 351) and, placed between warm-up and timing, would let the GPU fall idle again.
    import gc
    gc.collect()
    gc.disable()
    # device warm-up (untimed, before the W warm-up steps): the same kind of step on extra observation sets
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        rng_pw = np.random.default_rng(SEED + 7919 + rank)
        base = zs[0]
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            step(zs[prewarm_steps % len(zs)] + rng_pw.normal(0, 1, base.shape) * np.array([[0.02], [0.2 * math.pi / 180]]))
            prewarm_steps += 1
            if prewarm_steps % 16 == 0:
                st.sync()
        st.sync()
    for i in range(args.warmup):
        step(zs[i])
    st.sync()
    # HIP events around the dominant kernel only, and on every TIMING_STRIDE-th step of the timed region only: an
    # event pair costs ~10 us of stream time (1.5 % of a step); the host toggles it while the GPU is busy with the
    # previous update, so the toggling itself costs nothing
    TIMING_STRIDE = 4
    st.timing_reset()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    matched = 0
    matched_timed = 0
    for i in range(args.warmup, total_steps):
        timed = (i - args.warmup) % TIMING_STRIDE == 0
        st.timing(timed, kernels=["syrk"])
        mi = step(zs[i])
        matched += mi
        if timed:
            matched_timed += mi
    st.sync()
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()

    tim = st.timing_read()
    syrk_min_ms = st.timing_min("syrk")
    # diagnostics (outside the timed region): every kernel bracketed, and the factorisation kernel's phase stamps
    ndiag = min(5, total_steps)
    st.timing(True)
    st.timing_reset()
    for i in range(ndiag):
        step(zs[total_steps - 1 - i])
    st.sync()
    tim_all = st.timing_read()
    st.timing(False)
    # the timed region brackets every 4th of its steps (5 launches at the driver's 20 steps: a mean whose +-2 % is
    # the size of a round's gain).  NROOF more steps of the same kind with the down-date bracketed on every one: roofline.frac is
    # formed from THIS mean; the timed region's stays beside it as avg_launch_ms.
    st.timing(True, kernels=["syrk"])
    st.timing_reset()
    rng_rf = np.random.default_rng(SEED + 104729 + rank)
    matched_roof = 0
    for i in range(NROOF):
        matched_roof += step(zs[(args.warmup + i) % total_steps] + rng_rf.normal(0, 1, zs[0].shape) * np.array([[0.02], [0.2 * math.pi / 180]]))
    st.sync()
    roof_n, roof_mean_ms, roof_sd_ms, roof_min_ms = st.timing_stats("syrk")
    st.timing(False)
    st.debug_stamps(True)
    step(zs[-1])
    st.sync()
    stamps = st.debug_stamps(False)
    phases = [(b - a) / 100.0 for a, b in zip(stamps[:6], stamps[1:7])]     # 100 MHz ticks -> us
    # predict (src/ekf.jl:8-43: it runs nine times as often as the update in sim!) and add_features (:84-122) are timed
    # SEPARATELY, after the timed region: device time of their kernels by HIP events
    Qp = np.array([[0.5 ** 2, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
    st.timing(True, kernels=["predict"])
    st.timing_reset()
    for _ in range(20):
        st.predict(8.0, 0.05, 4.0, Qp, 0.025)
    st.sync()
    pr_ms, pr_n = st.timing_read()["predict"]
    st.timing(False)
    other = {"predict_us": 1e3 * pr_ms / max(pr_n, 1), "predict_calls": pr_n}
    # the copy floor: the bare read + rewrite of the covariance tiles the down-date touches, on THIS
    # box, in THIS run, on the SAME buffer -- the boxes of the pool differ by +-6 %, kernel time / floor does not
    try:
        floor_ms, floor_form = st.copy_floor(10)
    except Exception as e:  # noqa: BLE001 -- a diagnostic, never fatal
        floor_ms, floor_form = None, f"{type(e).__name__}: {e}"
    gate_info = st.gate_info()             # which form of the gating the steps used (SLAM_GATE_AUTO: the grid from 16384 landmarks on)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RED_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mt = torch.tensor([matched], dtype=torch.float64, device=RED_DEVICE)
        dist.all_reduce(mt, op=dist.ReduceOp.SUM)
        matched_all = float(mt.item())
    else:
        matched_all = float(matched)

    st.close()
    if not big and rank == 0:
        # add_features needs head-room in the pre-allocated capacity: a second handle with 64 spare landmarks, same state
        st2 = pkg.EKFSlamState(x, P, dtype=args.dtype, max_landmarks=N + 64, device=local_rank)
        rng_af = np.random.default_rng(SEED + 31)
        st2.add_features(np.vstack([rng_af.uniform(20, 60, 8), rng_af.uniform(-1, 1, 8)]), R)      # (warm)
        st2.timing(True, kernels=["augment"])
        st2.timing_reset()
        for _ in range(6):
            st2.add_features(np.vstack([rng_af.uniform(20, 60, 8), rng_af.uniform(-1, 1, 8)]), R)
        st2.sync()
        af_ms, af_n = st2.timing_read()["augment"]
        other.update({"add_features_us": 1e3 * af_ms / max(af_n, 1), "add_features_new_per_call": 8, "add_features_calls": af_n})
        st2.close()

    out = None
    if rank == 0:
        esz = 4 if args.dtype == "f32" else 8
        syrk_ms, syrk_n = tim["syrk"]
        syrk_avg_s = (syrk_ms / max(syrk_n, 1)) * 1e-3
        k_avg = 2.0 * matched_timed / max(syrk_n, 1)                 # actual k = 2m per (event-bracketed) launch
        if args.form == "joseph":
            k_avg *= 2.0
        # the down-date updates ONE triangle (the tiles on/below the diagonal), like BLAS syrk:
        alg_flops = 1.0 * n * n * k_avg                               # n^2*k (one triangle)
        alg_bytes = 1.0 * n * n * esz                                 # lower triangle read + lower triangle written
        tflops = alg_flops / syrk_avg_s / 1e12 if syrk_avg_s > 0 else 0.0
        gbps = alg_bytes / syrk_avg_s / 1e9 if syrk_avg_s > 0 else 0.0
        traffic = hbm_bytes((pmc or {}).get("downdate"))           # measured in this run (child rocprofv3 --pmc passes)
        traffic_note = ("FETCH_SIZE x 2 + WRITE_SIZE per launch from two rocprofv3 --pmc child passes of this run "
                        f"({(pmc or {}).get('downdate', {}).get('launches', 0)} launches)" if traffic is not None else
                        ("not measured: " + ((pmc or {}).get("error") or "--no-pmc / multi-rank run")))
        split_bf16 = (args.dtype == "f32" and args.form == "cholesky" and 64 < k_avg <= 128
                      and not (int(os.environ.get("SLAMHIP_X", "0")) & 8))
        if split_bf16:
            # the down-date runs on the bf16 matrix cores with every fp32 operand split into three bf16 terms (six exact
            # products per fp32 product, csrc/ekf_syrk.hip): 6x the algorithmic flops at a 16x higher peak, so HBM bounds it
            roof = {"kernel": "downdate (P -= W1*W1'), fp32 via split-bf16 on the bf16 matrix cores", "bound": "hbm",
                    "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                    "traffic": traffic, "avg_launch_ms": syrk_avg_s * 1e3, "launches": syrk_n,
                    "launches_note": f"every {TIMING_STRIDE}th step of the timed region is bracketed by HIP events",
                    "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "fp32_equivalent_TFLOPs": tflops, "frac_of_fp32_mfma_peak_157TF": tflops / MFMA_F32_PEAK_TFLOPS,
                    "bf16_mfma_TFLOPs_executed": 6.0 * tflops,        # (k padded to a multiple of 16 adds a few per cent)
                    "bf16_mfma_peak_TFLOPs": 2500.0}
        elif args.dtype == "f32" and alg_bytes / (HBM_PEAK_GBPS * 1e9) >= alg_flops / (MFMA_F32_PEAK_TFLOPS * 1e12):
            # small k on the fp32 matrix cores (C2: k = 32): the P traffic's floor is above the matrix work's (SURVEY 8d:
            # "memory / launch-bound"), so HBM is the roofline that bounds the kernel
            roof = {"kernel": "downdate (P -= W1*W1'), fp32 matrix cores, small k", "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": syrk_avg_s * 1e3, "launches": syrk_n,
                    "launches_note": f"every {TIMING_STRIDE}th step of the timed region is bracketed by HIP events",
                    "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "fp32_mfma_TFLOPs": tflops, "frac_of_fp32_mfma_peak_157TF": tflops / MFMA_F32_PEAK_TFLOPS,
                    "note": "a launch of ~11 us: the covariance (16 MB at C2) fits the Infinity Cache and the kernel sits at its latency floor"}
        elif args.dtype == "f32":
            roof = {"kernel": "downdate (P -= W1*W1')", "bound": "mfma", "achieved": tflops,
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_F32_PEAK_TFLOPS,
                    "traffic": traffic, "avg_launch_ms": syrk_avg_s * 1e3, "launches": syrk_n, "launches_note": f"every {TIMING_STRIDE}th step of the timed region is bracketed by HIP events",
                    "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "hbm_achieved_GBps": gbps, "hbm_frac_of_8TBps": gbps / HBM_PEAK_GBPS}
        else:
            roof = {"kernel": "downdate (P -= X*Y')", "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic,
                    "avg_launch_ms": syrk_avg_s * 1e3, "launches": syrk_n,
                    "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes}
        # the roofline figures from the NROOF individually bracketed launches behind the timed region
        if roof_n > 0 and roof_mean_ms > 0:
            k48 = 2.0 * matched_roof / roof_n * (2.0 if args.form == "joseph" else 1.0)
            roof["avg_launch_ms_timed_region"] = roof["avg_launch_ms"]
            roof["launches_timed_region"] = roof.pop("launches")
            roof["avg_launch_ms_48"] = roof_mean_ms
            roof["stdev_ms"] = roof_sd_ms
            roof["launches_48"] = roof_n
            roof["avg_launch_ms"] = roof_mean_ms
            roof["launches"] = roof_n
            if roof["bound"] == "hbm":
                roof["achieved"] = alg_bytes / (roof_mean_ms * 1e-3) / 1e9
                roof["frac"] = roof["achieved"] / HBM_PEAK_GBPS
            else:
                roof["achieved"] = 1.0 * n * n * k48 / (roof_mean_ms * 1e-3) / 1e12
                roof["frac"] = roof["achieved"] / MFMA_F32_PEAK_TFLOPS
            roof["frac_note"] = (f"achieved / frac from the mean of {roof_n} individually bracketed launches run behind the timed region "
                                 f"(stdev {roof_sd_ms * 1e3:.1f} us, standard error {roof_sd_ms * 1e3 / math.sqrt(roof_n):.1f} us); the timed region's own "
                                 f"{roof['launches_timed_region']} bracketed launches: avg_launch_ms_timed_region")
            syrk_avg_s = roof_mean_ms * 1e-3
            if roof_min_ms and (not syrk_min_ms or roof_min_ms < syrk_min_ms):
                syrk_min_ms = roof_min_ms
        roof["copy_floor_ms"] = floor_ms
        roof["min_launch_ms"] = syrk_min_ms
        roof["kernel_over_floor"] = (syrk_avg_s * 1e3 / floor_ms) if floor_ms else None
        roof["kernel_over_floor_min"] = (syrk_min_ms / floor_ms) if (floor_ms and syrk_min_ms) else None
        roof["copy_floor_note"] = (f"slam_ekf_copy_floor: every stored tile read once and written back unchanged (non-temporal, 16 B per lane, "
                                   f"the down-date's band-major order, no panels, no matrix cores), 10 individually timed passes of each launch form on "
                                   f"this run's own matrix, the FASTEST pass ({floor_form}); kernel_over_floor = average launch / floor, "
                                   f"kernel_over_floor_min = fastest bracketed launch / floor (the box-independent figure)")
        out = {
            "metric": "EKF updates/sec @ 10k landmarks" if N == 10000 else f"EKF updates/sec @ {N} landmarks",
            "value": matched_all / elapsed,
            "unit": "obs-updates/s (one observation assimilated = one update)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "steps_per_s": world * args.steps / elapsed, "prewarm_steps": prewarm_steps,
            "matched_per_step": matched / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EKF-SLAM observation step (associate+update; no new features arise), N={N} landmarks (n={n}), nz={nz} obs/step, "
                                   f"gates {GATE1}/{GATE2}, {args.form} form, state resident in HBM",
                       "landmarks": N, "obs_per_step": nz, "form": args.form,
                       "parallelism": "single GPU" if world == 1 else f"{world} independent replicas (EKF does not shard)"},
            "roofline": roof,
            "kernel_ms_per_step": {k: v[0] / max(ndiag, 1) for k, v in tim_all.items()},
            "traffic_note": traffic_note,
            "kernel_ms_per_step_note": f"{ndiag} extra steps after the timed region with every kernel bracketed by events",
            "other_kernels": other,
            "factor_phases_us": dict(zip(["innovation", "build_S", "symmetrise", "eliminate", "y_g", "emit_C"], phases)),
        }
        # the gating sweep (K1): 12 state values per landmark read once per sweep, nz * N pairs evaluated
        g_ms = tim_all["gate"][0] / max(ndiag, 1)
        if g_ms > 0:
            out["gating_sweep"] = {"form": gate_info["form"], "ms": g_ms, "algorithmic_bytes": 12 * esz * N, "achieved_GBps": 12 * esz * N / (g_ms * 1e-3) / 1e9,
                                   "pairs_per_s": nz * N / (g_ms * 1e-3), "bound": "latency (one launch; HBM floor %.2f us)" %
                                   (12 * esz * N / (HBM_PEAK_GBPS * 1e9) * 1e6)}
            if gate_info["form"] == "grid":         # the O(candidates) form: not nz * N pairs but the landmarks of the gates' cells
                q = max(gate_info["queries"], 1) * nz
                out["gating_sweep"].pop("pairs_per_s")
                out["gating_sweep"].pop("achieved_GBps")        # (the sweep's 12 values per landmark are not what the grid reads)
                out["gating_sweep"].pop("algorithmic_bytes")
                out["gating_sweep"].update({"landmarks_visited_per_observation": gate_info["visited"] / q,
                                            "landmarks_evaluated_per_observation": gate_info["evaluated"] / q,
                                            "bound": "latency (one launch, four dependent round trips; O(candidates), independent of N)"})
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(x, P, zs, args.cpu_budget)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["other_legs"] = literal_cpu_legs(landmarks=N, nobs=nz)
    if want_configs and rank == 0:
        out["configs"] = configs_guarded(pkg, max(args.steps, 20), args.warmup, local_rank, pmc_cfg)
    # the FastSLAM leg comes AFTER the headline's numbers are complete, under a deadline: whatever happens to it on a node
    # this code has never seen (ranks on several physical GPUs, IPC mappings, a peer that dies), the line still comes out
    fast = None if args.no_fastslam else fastslam_guarded(out, rank, world, pkg, world, rank, local_rank, max(args.steps, 10),
                                                          args.warmup, fence)
    if rank == 0:
        if fast is not None:
            if "roofline" in fast:
                fast["roofline"]["traffic"] = hbm_bytes((pmc or {}).get("pf_step"))
            out["fastslam"] = fast
        print(json.dumps(out), flush=True)
    if world > 1:
        # (a rank that failed above may have left the others' collectives out of step: do not wait for ever)
        t = threading.Timer(90.0, lambda: os._exit(3))      # (the line is out; a barrier that hangs is still a failure)
        t.daemon = True
        t.start()
        dist.barrier()
        dist.destroy_process_group()
        t.cancel()
    if EXIT_CODE[0]:
        sys.exit(EXIT_CODE[0])


if __name__ == "__main__":
    main()
