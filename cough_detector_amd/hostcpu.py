"""Host CPU share of this process (cgroup quota), so host-side thread pools are sized to what the box
actually grants: an MI355X box shows 256 logical CPUs but a 16-core cgroup quota, and torch's default
128 spinning intra-op threads get the whole process throttled (~88 ms stalls every 100 ms)."""
from __future__ import annotations

import os


def cpu_share() -> int:
    n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


def bound_torch_threads(limit: int = 16) -> int:
    import torch
    n = max(1, min(limit, cpu_share(), torch.get_num_threads()))
    torch.set_num_threads(n)
    return n
