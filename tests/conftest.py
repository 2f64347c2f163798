import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup quota (a 1-GPU box grants 16 of a host with many
    more hardware threads).  The oracle's OpenMP runtime would otherwise start one spinning thread per HARDWARE thread of the host and run
    10x slower under the quota (seen on some boxes: the same GPU suite took 4 to 10 minutes instead of one)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except OSError:
        pass
    return n


# before libocn_oracle.so (and with it libgomp) is loaded
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, _usable_cores())))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libocn_oracle.so on first use."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ocn():
    """The product package; GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import oceananigans_jl_amd as pkg
    pkg._lib.lib()
    return pkg
