import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The checkers (oracle/: torch CPU ops, C + OpenMP loops, library sgemm) take most of the suite's time.  Importing bench (stdlib-only
# at import) sets the passive-waiting defaults for OpenBLAS / OpenMP workers before either runtime loads; and where the container's CFS
# quota is below the CPUs it shows (the GPU boxes of this pool: 16 of 256 - profiles/r06_host_probe.txt) the thread teams are sized to
# the quota instead of being throttled as a group.  The engine under test is not involved: nothing on the GPU path uses these runtimes.
import bench  # noqa: E402

_budget = bench.host_cpu_budget()
if _budget["cpu_quota"] is not None and _budget["usable_cpus"] < _budget["affinity"]:
    for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_k, str(_budget["usable_cpus"]))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _checker_threads_fit_the_cpu_quota():
    """Runtime twin of the environment defaults above, for a process whose OpenMP / BLAS runtimes were loaded before this file ran
    (a sitecustomize hook, a plugin): limit the already-loaded pools to the CPUs the quota allows."""
    limits = None
    if _budget["cpu_quota"] is not None and _budget["usable_cpus"] < _budget["affinity"]:
        try:
            import numpy  # noqa: F401  (loads the BLAS so that it can be limited)
            import threadpoolctl
            import torch
            torch.set_num_threads(_budget["usable_cpus"])
            limits = threadpoolctl.threadpool_limits(limits=_budget["usable_cpus"])
        except Exception:      # a missing helper only costs time
            limits = None
    yield
    if limits is not None:
        limits.restore_original_limits()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
