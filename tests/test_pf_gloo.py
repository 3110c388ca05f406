"""The N > 1 path on CPU: the FastSLAM driver (slam.jl_amd/pf.py: scalar all-reduces, all-gather of
log-weights, global systematic resampling, exchange of migrating particle records) run with 2 and
4 gloo ranks must reproduce the 1-rank run bit for bit.  The local compute is the NumPy shard from
tests/ (the product ships only the HIP shard)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, tmp_path, port, mode="known", shm="1"):
    out = str(tmp_path / f"w{world}{mode}{shm}")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "pf_gloo_worker.py"), out, mode]
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", SLAMHIP_SHM_SCALARS=shm)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    parts = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    return dict(pose=np.hstack([p["pose"] for p in parts]), lm=np.concatenate([p["lm"] for p in parts], axis=2),
                logw=np.concatenate([p["logw"] for p in parts]), mean_pose=parts[0]["mean_pose"],
                info=parts[0]["info"], resamples=int(parts[0]["resamples"]),
                all_mean=[p["mean_pose"] for p in parts])


@pytest.mark.timeout(600)
def test_sharded_driver_matches_single_rank(tmp_path):
    one = run_world(1, tmp_path, 29631)
    assert one["resamples"] >= 5
    for world, port in ((2, 29632), (4, 29633)):
        got = run_world(world, tmp_path, port)
        assert got["resamples"] == one["resamples"]
        assert np.array_equal(got["pose"], one["pose"])
        assert np.array_equal(got["lm"], one["lm"])
        assert np.allclose(got["logw"], one["logw"], rtol=0, atol=1e-12)
        assert np.allclose(got["info"], one["info"], rtol=1e-12)
        assert np.allclose(got["mean_pose"], one["mean_pose"], rtol=1e-12, atol=1e-12)
        for mp in got["all_mean"]:                               # every rank holds the same global estimate
            assert np.allclose(mp, got["mean_pose"], rtol=0, atol=1e-14)
    # the per-step scalars travel through shared memory by default; the torch.distributed collective must agree
    coll = run_world(2, tmp_path, 29634, shm="0")
    shm = run_world(2, tmp_path, 29635, shm="1")
    for key in ("pose", "lm", "logw", "info"):
        assert np.array_equal(coll[key], shm[key]), key
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("slamhip-")]       # nothing left behind


@pytest.mark.timeout(600)
def test_sharded_proposal_step_matches_single_rank(tmp_path):
    """The same for the FastSLAM-2.0 step (proposal=True): the proposal is per particle, so sharding changes nothing."""
    one = run_world(1, tmp_path, 29641, "proposal")
    got = run_world(2, tmp_path, 29642, "proposal")
    assert got["resamples"] == one["resamples"] >= 5
    assert np.array_equal(got["pose"], one["pose"]) and np.array_equal(got["lm"], one["lm"])
    assert np.allclose(got["logw"], one["logw"], rtol=0, atol=1e-12)
    assert np.allclose(got["info"], one["info"], rtol=1e-12)
    known = run_world(1, tmp_path, 29643)
    assert not np.array_equal(known["pose"], one["pose"])          # it is a different sampler


@pytest.mark.timeout(600)
def test_async_driver_with_halt_and_resume_matches_the_synchronous_one(tmp_path):
    """FastSLAM.step_async / flush (steps enqueued; a sharded filter halts at a resampling step, the host resamples through
    the collectives and resumes) must give the particles of FastSLAM.step, on 1, 2 and 4 ranks.  The NumPy shard restates
    the library's protocol (tests/pf_numpy_shard.py), including the ranks' shared scalar page."""
    sync = run_world(1, tmp_path, 29651)
    for world, port in ((1, 29652), (2, 29653), (4, 29654)):
        got = run_world(world, tmp_path, port, "known-async")
        assert got["resamples"] == sync["resamples"] >= 5
        assert np.array_equal(got["pose"], sync["pose"]) and np.array_equal(got["lm"], sync["lm"])
        assert np.allclose(got["logw"], sync["logw"], rtol=0, atol=1e-12)
        assert np.allclose(got["info"][-1], sync["info"][-1], rtol=1e-12)       # (Neff, resampled?) of the last step
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("slamhip-")]
