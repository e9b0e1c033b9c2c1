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


def explicit_device_limit(environ=None):
    """The device count a ``*_VISIBLE_DEVICES`` variable EXPLICITLY grants (the smallest, when several are set), or
    ``None`` when none is set.  This is the only evidence on which a multi-rank launcher may refuse to start: the sysfs
    heuristics of ``visible_gpu_count`` can under-count in an unfamiliar container, an operator's own variable cannot."""
    environ = os.environ if environ is None else environ
    limit = None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = environ.get(var)
        if val is not None:
            n = len([v for v in val.split(",") if v.strip() != ""])
            limit = n if limit is None else min(limit, n)
    return limit


def _parse_cpulist(text: str) -> set:
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(local_rank: int, kfd_root: str = "/sys/class/kfd/kfd/topology/nodes", drm_root: str = "/sys/class/drm",
                  environ=None, dri_root: str = "/dev/dri"):
    """``(numa_node, pci_address)`` of the GPU that HIP device ``local_rank`` will be, read from sysfs WITHOUT touching
    the runtime: the KFD topology lists GPU nodes in the runtime's enumeration order, each with the minor number of its
    DRM render node, and ``/sys/class/drm/renderD<minor>/device`` is the PCI function, whose ``numa_node`` names the host
    memory node it hangs off.  Nodes whose render node this process may not open are skipped (a container is handed only
    its own), and an index list in ``ROCR_VISIBLE_DEVICES`` / ``HIP_VISIBLE_DEVICES`` is applied.  ``(None, None)``
    whenever any step cannot be read -- the caller then leaves the affinity alone."""
    environ = os.environ if environ is None else environ
    try:
        gpus = []
        for node in sorted(os.listdir(kfd_root), key=lambda s: int(s) if s.isdigit() else 1 << 30):
            try:
                with open(os.path.join(kfd_root, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0 and "drm_render_minor" in props:
                gpus.append(int(props["drm_render_minor"]))
        if os.path.isdir(dri_root):
            gpus = [m for m in gpus if os.access(os.path.join(dri_root, f"renderD{m}"), os.R_OK | os.W_OK)]
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            val = environ.get(var)
            if val:
                idx = [v.strip() for v in val.split(",") if v.strip() != ""]
                if not all(v.isdigit() for v in idx):
                    return None, None                       # UUID form: the order cannot be resolved from here
                gpus = [gpus[int(v)] for v in idx if int(v) < len(gpus)]
        if not 0 <= local_rank < len(gpus):
            return None, None
        dev = os.path.join(drm_root, f"renderD{gpus[local_rank]}", "device")
        with open(os.path.join(dev, "numa_node")) as f:
            node = int(f.read().strip())
        pci = os.path.basename(os.path.realpath(dev))
        return (node if node >= 0 else None), pci
    except (OSError, ValueError, IndexError):
        return None, None


def bind_to_gpu_numa(local_rank: int, node_root: str = "/sys/devices/system/node", **sysfs) -> dict:
    """Pin this process (the calling thread; thread pools created afterwards inherit it) to the host CPUs of the NUMA
    node its GPU hangs off, before anything touches the GPU: a rank's launch thread, its RCCL proxy thread and its
    pinned staging buffers then sit next to its own device on a two-socket, 8-GPU node.  A no-op (reported as such)
    when the node cannot be determined, has no CPUs this process may run on, or the platform has no affinity call.
    Returns ``{"numa_node", "pci", "cpus"}`` for the bench line."""
    info = {"numa_node": None, "pci": None, "cpus": None}
    node, pci = gpu_numa_node(local_rank, **sysfs)
    info["pci"] = pci
    if node is None or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        with open(os.path.join(node_root, f"node{node}", "cpulist")) as f:
            want = _parse_cpulist(f.read())
        allowed = want & _initial_affinity()
        if not allowed:
            return info
        os.sched_setaffinity(0, allowed)
        info["numa_node"], info["cpus"] = node, len(allowed)
    except (OSError, ValueError):
        pass
    return info


def rebind_to_pci_numa(pci: str, info: dict, node_root: str = "/sys/devices/system/node",
                       pci_root: str = "/sys/bus/pci/devices") -> dict:
    """After the runtime is up: ``pci`` is the address of the device this rank REALLY got (from the device properties).
    When it differs from what ``bind_to_gpu_numa`` derived from sysfs order (an enumeration the runtime re-ordered), bind
    again to the NUMA node of the real device.  Threads created in between keep the first binding; the launch thread and
    everything created later move.  Returns the updated record (``rebound`` says whether anything changed)."""
    info = dict(info)
    info["rebound"] = False
    if not pci or pci == info.get("pci") or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        with open(os.path.join(pci_root, pci, "numa_node")) as f:
            node = int(f.read().strip())
        info["pci"] = pci
        if node < 0:
            return info
        with open(os.path.join(node_root, f"node{node}", "cpulist")) as f:
            want = _parse_cpulist(f.read())
        # the first binding may have narrowed the affinity to another node: widen from the process-wide mask of pid 1's view
        allowed = want & _initial_affinity()
        if allowed:
            os.sched_setaffinity(0, allowed)
            info.update(numa_node=node, cpus=len(allowed), rebound=True)
    except (OSError, ValueError):
        pass
    return info


_AFFINITY0 = None


def _initial_affinity() -> set:
    """The affinity mask this process started with (remembered at import, before any binding narrows it)."""
    return set(_AFFINITY0) if _AFFINITY0 is not None else set(os.sched_getaffinity(0))


try:
    _AFFINITY0 = frozenset(os.sched_getaffinity(0))
except (AttributeError, OSError):
    _AFFINITY0 = None
