"""Worker of tests/test_pf_protocol_gloo.py: one rank of the MODEL of the sharded filter's device-side exchange
(tests/pf_protocol_model.py), gloo all-gathers standing in for the posted stores into the peers' inboxes."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pf_protocol_model as M  # noqa: E402


def all_gather_rows(rows, world):
    t = torch.from_numpy(rows.view(np.int64).copy())
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [o.numpy().view(np.uint64) for o in outs]


def weights(step, n_global):
    """The filter's log-weights after step `step`: the same vector whatever the number of ranks (keyed by the global particle id)."""
    rng = np.random.default_rng(1000 + step)
    lw = rng.normal(-8.0, 2.0 + 0.7 * step, n_global)
    lw[rng.integers(0, n_global, 5)] += 30.0 * (step % 3)         # a few dominant particles: the Neff rule fires on some steps
    if step == 4:
        lw[: n_global // 2] = -np.inf                              # half of the filter without weight: empty leaves and nodes
    return lw.astype(np.float32).astype(np.float64)               # (fp32 storage, as the filter keeps them)


def main():
    out_path, n_global = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    me = M.RankModel(rank, world, n_global)
    roots, neffs, wants = [], [], []
    for step in range(1, 8):
        seq = step
        lw = weights(step, n_global)
        mine = lw[rank * me.n:(rank + 1) * me.n]
        # nothing of this step is in yet: the parity still holds the records of step seq - 2 (or zeros): the reader must not take them
        assert me.decide(seq) is None, "a stale record was accepted"
        rows = all_gather_rows(me.outgoing(seq, mine), world)
        # the posted stores arrive in pieces: everything but the last rank's records, one of them TORN (new values, old tag)
        for r in range(world - 1):
            me.deliver(seq, rows[r])
        torn = rows[world - 1].copy()
        torn[0, 4] = M.record_tag((1.0, 2.0, 3.0), seq, int(torn[0, 0]))
        me.deliver(seq, torn)
        assert me.decide(seq) is None, "a torn record was accepted"
        me.deliver(seq, rows[world - 1])
        force = True if step == 3 else (False if step == 5 else None)
        got = me.decide(seq, force)
        assert got is not None
        root, shift, neff, want = got
        # every rank has the same bits
        mine_bits = np.array([M._bits(root[0]), M._bits(root[1]), M._bits(root[2]), M._bits(shift), M._bits(neff), int(want)], dtype=np.uint64)
        for other in all_gather_rows(mine_bits.reshape(1, -1), world):
            assert np.array_equal(other.reshape(-1), mine_bits), f"step {step}: the ranks disagree"
        # ... and they are the one-rank filter's (formed level by level over all leaves, not through records and passes)
        one = M.root_one_rank(lw)
        assert tuple(M._bits(v) for v in one) == tuple(M._bits(v) for v in root), f"step {step}: sharded root != one-rank root"
        s1, n1, w1 = M.statistics(one, n_global, me.neff_frac, force)
        assert (M._bits(s1), M._bits(n1), w1) == (M._bits(shift), M._bits(neff), want)
        roots.append([M._bits(v) for v in root]); neffs.append(neff); wants.append(int(want))
        if want:
            # the gate of the resampling: this rank's step kernel has completed; the slowest rank's word comes late
            ready = all_gather_rows(np.array([[seq if rank != world - 1 or world == 1 else seq - 1]], dtype=np.uint64), world)
            me.inbox.ready[:] = [int(v.reshape(-1)[0]) for v in ready]
            assert me.gate_open(seq) == (world == 1), "the gate opened before every rank's step kernel had completed"
            me.inbox.ready[world - 1] = seq
            assert me.gate_open(seq)
    # a rank goes away: every rank's next step stops with PF_ERR_PEER instead of touching its memory
    gone = all_gather_rows(np.array([[1 if rank == world - 1 else 0]], dtype=np.uint64), world)
    me.inbox.gone[:] = [int(v.reshape(-1)[0]) for v in gone]
    assert me.decide(8) is None and me.error == M.PF_ERR_PEER
    assert not me.gate_open(8)
    if rank == 0:
        np.savez(out_path, roots=np.array(roots, dtype=np.uint64), neffs=np.array(neffs), wants=np.array(wants))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
