"""Worker of tests/test_gpu_pf.py::test_auto_mode_two_ranks_on_one_card: one rank of the GPU FastSLAM driver, either
the synchronous single-rank filter ("sync") or the sharded auto mode with all ranks on card 0 ("auto": peers attached
through IPC handles, resampling on the device; with SLAMHIP_PF_PEERS=0 the halting flow over gloo).  argv[3]: the
resampling regime ("mixed" or "every_step")."""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    regime = sys.argv[3] if len(sys.argv) > 3 else "mixed"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "nccl1":
        # a ONE-rank RCCL group: the collectives of the sharded flow (all-gather of the log-weights, all-to-all of the
        # records, object collectives, barriers) execute through RCCL on the one GPU a test box has
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    elif world > 1:
        dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    pkg = load_package()
    N, NL, SEED = 4096, 8, 21
    pf = pkg.PFSlamState(N, NL, seed=SEED, dtype="f64", device=0, distributed=(world > 1 or mode == "nccl1"))
    if mode == "nccl1":
        assert pf.comm.dist.get_backend() == "nccl" and not pf.comm.stage
        pf.force_exchange = True                 # take the multi-rank resampling flow (collectives) although every ancestor is local
        # the object collectives and barriers the sharded set-up uses, over RCCL; and a /dev/shm page registered with HIP
        shm = pkg.pf.ShmScalars(dist, 0, 1)
        assert shm.all_gather([1.0, 2.0, 3.0]) == [[1.0, 2.0, 3.0]]
        pf.shard.attach_exchange(0, 1, pkg.shared_page(dist, 0, 1, 16))
    rng = np.random.default_rng(5)                      # same scene / observations on every rank
    lm = rng.uniform(-30, 30, (NL, 2))
    R = np.array([[0.01, 0.0], [0.0, (math.pi / 180) ** 2]])
    Q = np.array([[0.25, 0.0], [0.0, (3 * math.pi / 180) ** 2]])
    pf.shard.set_pose([0.0, 0.0, 0.1])
    pf.shard.init_landmarks(lm, 0.01, 0.1)
    pose = np.array([0.0, 0.0, 0.1])
    neffs = []
    for t in range(14):
        V, G = 5.0, 0.1
        pose = np.array([pose[0] + V * 0.1 * math.cos(G + pose[2]), pose[1] + V * 0.1 * math.sin(G + pose[2]),
                         pose[2] + V * 0.1 * math.sin(G) / 4.0])
        ids = (np.arange(3) + 3 * t) % NL + 1
        dx, dy = lm[ids - 1, 0] - pose[0], lm[ids - 1, 1] - pose[1]
        z = np.vstack([np.hypot(dx, dy), np.arctan2(dy, dx) - pose[2]]) + rng.normal(0, [[0.1], [math.pi / 180]], (2, 3))
        if regime == "every_step":                      # (what BASELINE config 4 does at full size: every step resamples)
            force = None if t == 6 else True
        else:
            force = True if t % 4 == 1 else None
        if mode == "auto":
            pf.step_async(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=force)
            if t % 5 == 4 or t == 13:
                neffs.append(pf.flush()[0])
        else:
            neff, _ = pf.step(V, G, 4.0, Q, 0.1, z, ids, R, force_resample=force)
            if t % 5 == 4 or t == 13:
                neffs.append(neff)
    p, lw, l = pf.shard.download()                      # (collective when peers are attached)
    info = pf.shard.comm_info()
    np.savez(out_path + f".rank{rank}", pose=p, lm=l, logw=lw, resamples=pf.resamples, neff=np.array(neffs),
             halts=info["halts"], peers=int(info["peers"]))
    pf.close()
    if world > 1 or mode == "nccl1":
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
