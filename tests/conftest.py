"""Shared pytest configuration: markers, repo-root imports, package loader."""
import os
import sys

import pytest

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
