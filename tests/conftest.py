import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the GPU boxes show 256 logical CPUs under a 16-core cgroup quota: torch's default intra-op pool gets the process
    # throttled and the CPU oracle legs of the parity tests run 5x slower (same rule as bench.py)
    from cough_detector_amd.hostcpu import bound_torch_threads
    bound_torch_threads()


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when collected on a box without a GPU."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def resnet_golden():
    import numpy as np
    import torch
    g = np.load(os.path.join(GOLDEN, "resnet_golden.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    vec = {k: torch.from_numpy(g[k]) for k in g.files if not k.startswith("sd.")}
    return sd, vec


@pytest.fixture(scope="session")
def resnet_heights_golden(resnet_golden):
    """{"h64" | "h68" | "h92" | "h95" | "h103" | "h110": (state_dict, vectors)}: the reference module on images of those heights
    (oracle/make_golden_heights.py); the conv weights of resnet_golden.npz, the head re-calibrated per height."""
    import numpy as np
    import torch
    base_sd, _ = resnet_golden
    g = np.load(os.path.join(GOLDEN, "resnet_heights_golden.npz"))
    out = {}
    for name in ("h64", "h68", "h92", "h95", "h103", "h110"):
        sd = dict(base_sd)
        sd["fc.2.weight"] = torch.from_numpy(g[name + ".fc.2.weight"])
        sd["fc.2.bias"] = torch.from_numpy(g[name + ".fc.2.bias"])
        vec = {k[len(name) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + ".") and ".fc.2." not in k}
        out[name] = (sd, vec)
    return out


@pytest.fixture(scope="session")
def resnet_channels_golden():
    """{case: (channels, state_dict, vectors)} for non-default ``channels`` tuples (oracle/make_golden_channels.py);
    the inputs are the first clips of features_golden.npz."""
    import numpy as np
    import torch
    g = np.load(os.path.join(GOLDEN, "resnet_channels_golden.npz"))
    n = int(g["n_clips"])
    x = torch.from_numpy(np.load(os.path.join(GOLDEN, "features_golden.npz"))["features"][:n]).unsqueeze(1).contiguous()
    cases = {}
    for name in sorted({k.split(".")[0] for k in g.files if "." in k}):
        pre = name + ".sd."
        sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
        vec = {k[len(name) + 1:]: torch.from_numpy(g[k]) for k in g.files
               if k.startswith(name + ".") and not k.startswith(pre)}
        vec["x"] = x
        cases[name] = (tuple(int(c) for c in vec.pop("channels")), sd, vec)
    return cases


@pytest.fixture(scope="session")
def resblock_golden():
    """{case: ((in_ch, out_ch, stride), state_dict, x, y)}: the reference ResidualBlock called on its own
    (oracle/make_golden_channels.py), "identity" being the nn.Identity skip."""
    import numpy as np
    import torch
    g = np.load(os.path.join(GOLDEN, "resblock_golden.npz"))
    cases = {}
    for name in sorted({k.split(".")[0] for k in g.files}):
        pre = name + ".sd."
        sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
        cases[name] = (tuple(int(v) for v in g[name + ".cfg"]), sd, torch.from_numpy(g[name + ".x"]),
                       torch.from_numpy(g[name + ".y"]))
    return cases


@pytest.fixture(scope="session")
def features_golden():
    import numpy as np
    g = np.load(os.path.join(GOLDEN, "features_golden.npz"))
    return {k: g[k] for k in g.files}


@pytest.fixture(scope="session")
def cnn_golden():
    """{"x": tensor, "standard": (state_dict, vectors), "small": (state_dict, vectors)} from the reference modules."""
    import numpy as np
    import torch
    g = np.load(os.path.join(GOLDEN, "cnn_golden.npz"))
    out = {"x": torch.from_numpy(g["x"])}
    for kind in ("standard", "small"):
        sd = {k[len(kind) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(kind + ".sd.")}
        vec = {k[len(kind) + 1:]: torch.from_numpy(g[k]) for k in g.files
               if k.startswith(kind + ".") and not k.startswith(kind + ".sd.")}
        out[kind] = (sd, vec)
    return out
