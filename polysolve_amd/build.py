"""Build libpsolve_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libpsolve_hip.so")


def build(force: bool = False, jobs: int = 4) -> str:
    cmd = ["make", "-s", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError(f"build did not produce {LIB}")
    return LIB


if __name__ == "__main__":
    print(build())
