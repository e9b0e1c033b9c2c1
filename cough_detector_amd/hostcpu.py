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


def visible_gpu_count(kfd_root: str = "/sys/class/kfd/kfd/topology/nodes", environ=None, dri_root: str = "/dev/dri"):
    """GPUs this process would see, WITHOUT initialising HIP/HSA (no /dev/kfd open): the KFD topology nodes with
    SIMDs (CPU nodes have ``simd_count 0``), cut down by ``ROCR_VISIBLE_DEVICES`` / ``HIP_VISIBLE_DEVICES`` /
    ``CUDA_VISIBLE_DEVICES`` when one is set and by the number of render nodes under ``/dev/dri`` this process may
    open (a container is usually handed only its own GPUs' nodes while sysfs shows the whole host).  0 when the
    host has no KFD driver at all; ``None`` when the topology exists but cannot be read (the ranks then report a
    shortage themselves).  A multi-rank launcher uses this instead of ``torch.cuda.device_count()``, which falls back
    to ``hipGetDeviceCount`` (runtime initialisation) when the amdsmi python package is absent."""
    environ = os.environ if environ is None else environ
    if not os.path.exists("/sys/class/kfd") and kfd_root.startswith("/sys/class/kfd"):
        return 0
    try:
        nodes = sorted(os.listdir(kfd_root))
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(kfd_root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue                      # a node this cgroup may not read is a device it may not use
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    try:
        render = [d for d in os.listdir(dri_root) if d.startswith("renderD")]
        n = min(n, sum(os.access(os.path.join(dri_root, d), os.R_OK | os.W_OK) for d in render))
    except OSError:
        pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = environ.get(var)
        if val is not None:
            n = min(n, len([v for v in val.split(",") if v.strip() != ""]))
    return n
