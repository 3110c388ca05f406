"""Worker of tests/test_pf_gloo.py: one rank of the FastSLAM driver over gloo with the NumPy shard."""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from __graft_entry__ import load_package  # noqa: E402
from pf_numpy_shard import NumpyShard  # noqa: E402


def main():
    out_path = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "known"
    proposal = mode in ("proposal", "proposal-async")               # the FastSLAM-2.0 step
    use_async = mode.endswith("async")                              # FastSLAM.step_async / flush (halt + resume protocol)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = load_package()
    N, NL, SEED = 512, 6, 21
    per = N // world
    shard = NumpyShard(per, NL, SEED, first=rank * per, n_global=N)
    comm = pkg.TorchComm(torch.device("cpu"))
    pf = pkg.FastSLAM(shard, comm, neff_frac=0.75)
    if use_async and world > 1:
        shard.attach_exchange(rank, world, pkg.shared_page(dist, rank, world, 2 * world * 8))
    rng = np.random.default_rng(5)                      # same scene / observations on every rank
    lm = rng.uniform(-30, 30, (NL, 2))
    R = np.array([[0.01, 0.0], [0.0, (math.pi / 180) ** 2]])
    Q = np.array([[0.25, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
    shard.set_pose([0.0, 0.0, 0.1])
    shard.init_landmarks(lm, 0.01, 0.1)
    pose = np.array([0.0, 0.0, 0.1])
    info = []
    for t in range(10):
        V, G = 5.0, 0.1
        pose = np.array([pose[0] + V * 0.1 * math.cos(G + pose[2]), pose[1] + V * 0.1 * math.sin(G + pose[2]),
                         pose[2] + V * 0.1 * math.sin(G) / 4.0])
        ids = (np.arange(3) + 3 * t) % NL + 1
        dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, 3))
        if use_async:                   # enqueue only; Neff and the decision are read back every third step
            pf.step_async(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=True if t % 2 == 1 else None, proposal=proposal)
            if t % 3 == 2 or t == 9:
                info.append(pf.flush())
        else:
            neff, did = pf.step(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=True if t % 2 == 1 else None, proposal=proposal)
            info.append((neff, did))
    pf.normalize()
    mp = pf.mean_pose()
    np.savez(out_path + f".rank{rank}", pose=shard.o.pose, lm=shard.o.lm, logw=shard.o.logw, mean_pose=mp,
             info=np.array(info, dtype=np.float64), resamples=pf.resamples)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
