"""Host-side facts the benchmark / tests report (no GPU needed)."""
from __future__ import annotations

import os


def usable_cores() -> int:
    """CPU cores this process may actually use: min(sched affinity, cgroup-v2/v1 CPU quota).  os.cpu_count() reports the
    machine's cores even inside a CPU-limited container, and OpenMP spinning on 100+ threads with a quota of a few cores
    is orders of magnitude slower than using the quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)
