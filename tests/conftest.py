import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """Device context for -m gpu tests.  Fails (does not skip) when the HIP library or device is missing."""
    from microaligner_amd.device import get_context
    return get_context()


def oracle_threads():
    """OpenMP threads for the oracle side of the big GPU tests: the CPUs this process can actually run on at once -- the smaller
    of the affinity mask and the container's CPU-time quota (cgroup v2 cpu.max / v1 cfs_quota), as bench.effective_cpus().  The
    GPU boxes of this pool show 256 hardware threads and grant 16 CPUs of time: 256 threads there are 16 cores' worth of work
    plus the switching.  (Whether a host is big enough for the full-size cases is still judged by its hardware threads: the
    cases pass on the quota, they just take minutes.)"""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n = max(1, int(quota))
    return n
