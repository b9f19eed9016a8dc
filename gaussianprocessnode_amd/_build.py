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
# variant libraries (same ABI, extra compile-time options): "chain" = with the round-2 persistent factorisation launch
# (csrc/sgp_chain.hip.h; correct, measured slower, kept out of the default library)
VARIANTS = {"chain": ("libsgp_hip_chain.so", ["-DSGP_WITH_PERSISTENT_CHAIN"]),
            # diagnostics: in-kernel begin / end stamps of every kernel of a sweep (tools/sweep_trace.py)
            "trace": ("libsgp_hip_trace.so", ["-DSGP_SWEEP_TRACE", "-DSGP_STEP_TRACE"])}
SOURCES = [os.path.join(CSRC, "sgp_api.hip")]
HEADERS = [os.path.join(CSRC, "sgp_kernels.hip.h"), os.path.join(CSRC, "sgp_chain.hip.h"), os.path.join(os.path.dirname(HERE), "include", "sgp_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built (no CPU fallback exists)")


def lib_path(variant: str | None = None) -> str:
    return LIB if not variant else os.path.join(CSRC, VARIANTS[variant][0])


def needs_build(variant: str | None = None) -> bool:
    lib = lib_path(variant)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, variant: str | None = None) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libsgp_hip.so (or a variant library); returns the library path."""
    lib = lib_path(variant)
    if not force and not needs_build(variant):
        return lib
    extra = VARIANTS[variant][1] if variant else []
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + extra + [
           "-o", lib] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    for v in VARIANTS:
        print(build(force=True, verbose=True, variant=v))
