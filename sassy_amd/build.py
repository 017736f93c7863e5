"""Builds libsassy_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch extension machinery: the product is a plain C-ABI shared library
(include/sassy.h, include/sassy_hip.h); Python reaches it through ctypes.
hipcc cross-compiles gfx950 without a GPU, so this runs on the CPU-only build box as well.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "libsassy_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
CFLAGS += os.environ.get("SASSY_EXTRA_CFLAGS", "").split()  # experiments (e.g. -DSASSY_NT_LOADS); build with --force

# (source, object name, extra flags): the scan kernel is compiled once per alphabet profile
UNITS = [
    ("scan_kernel.hip", "scan_ascii.o", ["-DSASSY_SCAN_PROFILE=0"]),
    ("scan_kernel.hip", "scan_dna.o", ["-DSASSY_SCAN_PROFILE=1"]),
    ("scan_kernel.hip", "scan_iupac.o", ["-DSASSY_SCAN_PROFILE=2"]),
    ("count_filter.hip", "count_filter.o", []),
    ("aux_kernels.hip", "aux_kernels.o", []),
    ("sort_kernels.hip", "sort_kernels.o", []),
    ("tiled_kernel.hip", "tiled_kernel.o", []),
    ("seed_kernels.hip", "seed_kernels.o", []),
    ("trace_kernel.hip", "trace_kernel.o", []),
    ("scan_driver.hip", "scan_driver.o", []),
    ("many_patterns.hip", "many_patterns.o", []),
    ("multi_device.hip", "multi_device.o", []),
    ("c_abi.hip", "c_abi.o", []),
]
HEADERS = ["common.h", "profiles.h", "tiled_step.h", "switches.h", "host_internal.h", os.path.join("..", "..", "include", "sassy.h"),
           os.path.join("..", "..", "include", "sassy_hip.h")]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(unit):
    src, obj, extra = unit
    srcp, objp = os.path.join(CSRC, src), os.path.join(OBJDIR, obj)
    deps = [srcp] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if _mtime(objp) >= max(_mtime(d) for d in deps):
        return objp, False
    cmd = [HIPCC] + CFLAGS + extra + ["-c", srcp, "-o", objp]
    subprocess.check_call(cmd)
    return objp, True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile (if stale) and link libsassy_hip.so; returns its path."""
    if not os.path.isdir(CSRC):  # a box that only carries the prebuilt library
        if os.path.exists(SO):
            return SO
        raise RuntimeError("sassy_amd/csrc is missing and no prebuilt libsassy_hip.so exists")
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for _, obj, _ in UNITS:
            try:
                os.remove(os.path.join(OBJDIR, obj))
            except FileNotFoundError:
                pass
    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        results = list(ex.map(_compile, UNITS))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _mtime(SO) < max(_mtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("linked", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
