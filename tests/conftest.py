import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """The CPU oracle (test infrastructure only), built on demand with gcc."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "strmatch_oracle"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(ROOT, "oracle", "strmatch_oracle")


@pytest.fixture(scope="session")
def fixtures_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("fx"))
