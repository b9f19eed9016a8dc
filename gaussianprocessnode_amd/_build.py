"""Build the gfx950 shared library (C ABI of include/sgp_hip.h) in-tree with hipcc.

hipcc cross-compiles without a GPU; the resulting csrc/libsgp_hip.so travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsgp_hip.so")
SOURCES = [os.path.join(CSRC, "sgp_api.hip")]
HEADERS = [os.path.join(CSRC, "sgp_kernels.hip.h"), os.path.join(CSRC, "sgp_chain.hip.h"), os.path.join(os.path.dirname(HERE), "include", "sgp_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built (no CPU fallback exists)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libsgp_hip.so; returns the library path."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
