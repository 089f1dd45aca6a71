"""pytest configuration: registers the `gpu` marker and builds the CPU oracle.

`-m "not gpu"` runs everything that needs no GPU (oracle vs golden vectors,
host logic, C-ABI symbol checks); `-m gpu` runs the parity tests proper on
an MI355X through the C-ABI library.
"""
import os

# the encoder's frame pipelining wants more hardware queues than the HIP runtime's default 4;
# read when the runtime starts, so it has to be in the environment before the first HIP call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """Build oracle/build/libh264o.so (gcc, seconds). Test infrastructure only."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
