"""Build the HIP engine in-tree: `python -m climsim_amd.build` -> climsim_amd/libclimsim_hip.so.

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the GPU box
with the repo snapshot.  No other architecture is built.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "climsim_hip.hip")


def _deps():
    """Every source the .so is built from: the .hip file, all headers next to it and the public header."""
    import glob
    return [SRC] + sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + sorted(
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))


OUT = os.path.join(HERE, "libclimsim_hip.so")
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc on PATH or under /opt/rocm/bin)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", SRC, "-o", OUT + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
