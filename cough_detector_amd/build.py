"""Build libcough_amd.so (HIP, gfx950 only) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the gpurun snapshot.
Usage: ``python -m cough_detector_amd.build [--force]``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcough_amd.so")
SOURCES = ("api.hip", "featurize.hip", "spectrogram.hip", "resnet.hip", "cnn.hip", "stream.hip", "synth.hip")
FLAGS = ["-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "cough_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), *FLAGS, "-o", LIB, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
