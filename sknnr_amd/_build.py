"""Build the HIP shared library in-tree with hipcc for gfx950.

``python -m sknnr_amd._build`` (or ``__graft_entry__.build()``) produces
``sknnr_amd/csrc/libsknnr_hip.so``.  The library is git-ignored but travels with the
repository snapshot to the GPU box.

The sources are one host translation unit (``sknnr_hip.hip``: index build, workspace, host
pipeline, C ABI) and the kernel translation units ``k_*.hip`` behind ``launch.hip.h``; the units
are compiled in parallel into ``csrc/_obj/`` (each one only when it or a header is newer than its
object) and linked by hipcc.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "libsknnr_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
OBJ_DIR = os.path.join(CSRC, "_obj")
# (object name, source, extra flags)
UNITS = [
    ("host", "sknnr_hip.hip", []),
    ("k_exact", "k_exact.hip", []),
    ("k_hamming", "k_hamming.hip", []),
    ("k_coarse1", "k_coarse1.hip", []),
    ("k_coarse2_a", "k_coarse2.hip", ["-DSKNNR_C2_PART=0"]),
    ("k_coarse2_b", "k_coarse2.hip", ["-DSKNNR_C2_PART=1"]),
]
SOURCES = sorted({u[1] for u in UNITS})
HEADERS = ["launch.hip.h", "coarse.hip.h", "coarse2.hip.h", "bucket.hip.h", "hamming.hip.h", "exact.hip.h",
           "../../include/sknnr_hip.h"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
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


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(p) > lib_m for p in deps)


def _compile_unit(hipcc, name, src, flags, obj_dir, verbose):
    obj = os.path.join(obj_dir, name + ".o")
    src_path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src_path), _newest_header()):
        return obj, ""
    cmd = [hipcc, *HIPCC_FLAGS, *flags, "-c", src_path, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stdout + proc.stderr)
    os.replace(obj + ".tmp", obj)
    return obj, proc.stderr


def build(force: bool = False, verbose: bool = False, extra_flags=(), out_path: str | None = None,
          jobs: int | None = None) -> str:
    """Compile the library if it is missing or older than its sources.  A variant (``extra_flags``
    and / or ``out_path``) gets an object directory of its own, keyed by its flags."""
    if out_path is None and not extra_flags and not force and not is_stale():
        return LIB_PATH
    target = out_path or LIB_PATH
    hipcc = hipcc_path()
    extra_flags = list(extra_flags)
    obj_dir = OBJ_DIR
    if extra_flags:
        obj_dir = OBJ_DIR + "_" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:10]
    os.makedirs(obj_dir, exist_ok=True)
    if force:
        for f in os.listdir(obj_dir):
            if f.endswith(".o"):
                os.remove(os.path.join(obj_dir, f))
    jobs = jobs or min(len(UNITS), max(1, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        futs = [pool.submit(_compile_unit, hipcc, name, src, [*flags, *extra_flags], obj_dir, verbose)
                for name, src, flags in UNITS]
        results = [f.result() for f in futs]
    objs = [r[0] for r in results]
    if verbose:
        for _, err in results:
            if err:
                print(err, file=sys.stderr)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", *objs, "-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + proc.stdout + proc.stderr)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
