"""Shared pytest configuration: markers, repo-root imports, package loader."""
import os
import sys

import pytest

# Several shards of ONE process (tests/test_gpu_pf.py::test_sharded_filter_resamples_on_the_device: one host thread and one
# stream per shard) wait for each other's records inside their kernels: every such stream needs a hardware queue of its
# own -- two streams on one queue put the kernel that waits in front of the kernel it waits for.  The runtime's default
# is four queues per device; it is read at the first HIP call, hence here.  (One process per GPU -- the deployment -- has
# no such limit to mind.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The host package ``slam.jl_amd`` (directory name is not an identifier, so
    it is loaded by path and registered as ``slam_jl_amd``)."""
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
