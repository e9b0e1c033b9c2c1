"""Build libcough_amd.so (HIP, gfx950 only) in-tree with hipcc.

Every translation unit is compiled to an object file of its own (in parallel, cached under ``build/`` by the newest
source / header time) and the objects are linked into the shared library.  The library is git-ignored but travels to
the GPU box with the gpurun snapshot.  Usage: ``python -m cough_detector_amd.build [--force]``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcough_amd.so")
SOURCES = ("api.hip", "featurize.hip", "featurize_generic.hip", "spectrogram.hip", "resnet.hip", "cnn.hip", "stream.hip", "synth.hip")
# -fno-slp-vectorize: left alone, -O3 packs adjacent f32 adds / multiplies of the FFT butterflies into v_pk_*_f32, which issue
# slower than the two scalar operations they replace on gfx950 (same-box A/B: K1 -2.4 %, STFT stage -3.2 %, classifier unchanged;
# profiles/r04_flag_ab.txt)
# -fvisibility=hidden: the shared object exports the entry points include/cough_amd.h declares (its `#pragma GCC visibility
# push(default)`) and nothing else -- no C++-mangled internals, no std::vector instantiations
CFLAGS = ["-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
          "-fvisibility=hidden", "-fvisibility-inlines-hidden"]
FLAGS = CFLAGS + ["-shared"]          # one-shot command line (the diagnostic tools build variants with it)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _headers_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "cough_amd.h"))
    deps.append(os.path.abspath(__file__))   # the flags live here
    deps.append(os.path.join(CSRC, "exports.map"))
    return max(os.path.getmtime(d) for d in deps)


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return _headers_mtime() > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build_library(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB) -> str:
    """Compile (only what changed, unless ``force`` / ``extra_flags``) and link.  ``extra_flags`` (e.g. ``-DCOUGH_K1_STAMPS``)
    build a variant library at ``out`` with objects of its own."""
    if not force and not extra_flags and not is_stale():
        return LIB
    hipcc = _hipcc()
    objdir = OBJ if not extra_flags else OBJ + "_" + str(abs(hash(tuple(extra_flags))) % 10**8)
    os.makedirs(objdir, exist_ok=True)
    hdr = _headers_mtime()

    def compile_one(src: str) -> str:
        s, o = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(hdr, os.path.getmtime(s)):
            cmd = [hipcc, *CFLAGS, *extra_flags, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
