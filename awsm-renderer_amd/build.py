"""build.py — compiles the native pieces in-tree (so the .so files travel to the GPU box with the snapshot).

  libawsm_hip.so   hipcc --offload-arch=gfx950 (cross-compiles without a GPU)   csrc/Makefile
  libawsm_host.so  g++ (no HIP)                                                  host/Makefile
"""
from __future__ import annotations

import os
import shutil
import subprocess

from . import PACKAGE_DIR


def _run_make(directory: str, target: str, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    proc = subprocess.run(["make", "-C", directory, target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e)
    if proc.returncode != 0:
        raise RuntimeError(f"building {target} in {directory} failed:\n{proc.stdout[-4000:]}")
    return proc.stdout


def build_hip(jobs: int = 3) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libawsm_hip.so cannot be built and there is no CPU fallback")
    proc = subprocess.run(["make", "-C", os.path.join(PACKAGE_DIR, "csrc"), f"-j{jobs}", "../libawsm_hip.so", f"HIPCC={hipcc}"],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc build failed:\n{proc.stdout[-4000:]}")
    return os.path.join(PACKAGE_DIR, "libawsm_hip.so")


def build_host() -> str:
    _run_make(os.path.join(PACKAGE_DIR, "host"), "../libawsm_host.so")
    return os.path.join(PACKAGE_DIR, "libawsm_host.so")


def build_all():
    return build_hip(), build_host()
