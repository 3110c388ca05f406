"""The DEFAULT exchange of the sharded filter -- the device-side protocol: tagged 1024-particle records into every rank's inbox, the
radix-4 statistics tree, the gate word of a resampling step, the `gone` word -- as a host-side model (tests/pf_protocol_model.py)
driven by 1, 2 and 4 gloo ranks on the CPU (tests/test_pf_gloo.py covers the halting flow).  Every rank count must
give the SAME root (max, sum w, sum w^2), shift, Neff and decisions bit for bit, equal to the one-rank tree formed level by level;
stale and torn records must be polled again, the gate must wait for the slowest rank, a peer's `gone` word must stop every rank.
The model's constants are checked against the sources, so a protocol change in csrc/ fails here, in the build container."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_world(world, tmp_path, port, n_global):
    out = str(tmp_path / f"proto{world}_{n_global}.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "pf_protocol_worker.py"), out, str(n_global)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1"), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_global,port", [(8192, 29661), (4 * 1024 * 5, 29671)])
def test_device_protocol_model_is_invariant_in_the_number_of_ranks(tmp_path, n_global, port):
    one = run_world(1, tmp_path, port, n_global)
    assert one["wants"].sum() >= 2 and one["wants"].sum() < len(one["wants"])          # both decisions occur
    for world, p in ((2, port + 1), (4, port + 2)):
        got = run_world(world, tmp_path, p, n_global)
        assert np.array_equal(got["roots"], one["roots"]), f"{world} ranks: the root differs from one rank's"
        assert np.array_equal(got["neffs"].view(np.uint64), one["neffs"].view(np.uint64))
        assert np.array_equal(got["wants"], one["wants"])


def test_the_model_follows_the_sources():
    """The constants and formulas of the model are those of csrc/pf_device.h and csrc/pf_auto.hip (as text): the key of a step, the
    exchange's extra key, the index multiplier of a record's tag, the rotations of the hash, the record capacity of an inbox."""
    import pf_protocol_model as M
    dev = open(os.path.join(ROOT, "slam.jl_amd", "csrc", "pf_device.h")).read()
    auto = open(os.path.join(ROOT, "slam.jl_amd", "csrc", "pf_auto.hip")).read()
    internal = open(os.path.join(ROOT, "slam.jl_amd", "csrc", "pf_internal.h")).read()
    assert f"0x{M.KEY_MUL:016X}ull + 0x{M.KEY_ADD:016X}ull".lower() in dev.lower()                     # part_key
    assert "((b0 << 7) | (b0 >> 57)) ^ ((b1 << 23) | (b1 >> 41)) ^ ((b2 << 41) | (b2 >> 23))" in dev         # wrec_hash
    assert f"key ^ 0x{M.XKEY:016X}ull".lower() in auto.lower()                                           # xkey
    assert auto.lower().count(f"0x{M.INDEX_MUL:016X}ull".lower()) >= 2                                   # writer and reader
    assert "wrec_hash(r) ^ xkey ^ ((unsigned long long)gi * 0xD6E8FEB86659FD93ull)" in auto                 # the tag a rank writes
    assert re.search(r"const int gi = a\.rank \* nc_local \+ cj;", auto)                                     # a record's index
    assert "(n_global + 1023) / 1024 + PF_MAX_WORLD" in internal and "constexpr int PF_MAX_WORLD = 8;" in internal      # Inbox.rec_cap
    assert "r.s1 = ((a1 + b1) + c1) + d1;" in dev and "const int dk = (int)fmax(wrec_k(x.m) - K, -4000.0);" in dev       # combine4
    assert "constexpr int PF_ERR_PEER = 3;" in internal
    # and the tree itself: a node over leaves is the same whatever the grouping
    rng = np.random.default_rng(3)
    lw = rng.normal(-5, 4, 64 * 16 * 5 + 17)
    recs = M.records_of_slice(lw)
    assert tuple(map(M._bits, M.root_from_records(recs))) == tuple(map(M._bits, M.root_one_rank(lw)))
