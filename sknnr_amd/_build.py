"""Build the HIP shared library in-tree with hipcc for gfx950.

``python -m sknnr_amd._build`` (or ``__graft_entry__.build()``) produces
``sknnr_amd/csrc/libsknnr_hip.so``.  The library is git-ignored but travels with the
repository snapshot to the GPU box.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "libsknnr_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ["sknnr_hip.hip"]
HEADERS = ["coarse.hip.h", "coarse2.hip.h", "bucket.hip.h", "hamming.hip.h", "exact.hip.h", "../../include/sknnr_hip.h"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    # the float64 stages rely on explicit fma() only (see exact.hip.h)
    "-ffp-contract=off",
    "-fvisibility=hidden",
    "-Wall",
    "-Wno-unused-function",
]


def hipcc_path() -> str:
    cand = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cand):
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return cand


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(p) > lib_m for p in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out_path: str | None = None) -> str:
    """Compile the library if it is missing or older than its sources (always, for a variant
    written to ``out_path``)."""
    if out_path is None and not force and not is_stale():
        return LIB_PATH
    target = out_path or LIB_PATH
    cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    if verbose and proc.stderr:
        print(proc.stderr, file=sys.stderr)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
