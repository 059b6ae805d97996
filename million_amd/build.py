"""Build libmillion_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m million_amd.build` or `make bindings` (the reference's `make bindings` runs
scripts/modeldb/bindings/setup.py, reference makefile:1-4).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libmillion_hip.so"
SOURCES = ["million_api.hip", "encode.hip", "attn_generic.hip", "attn_tile.hip", "attn_mfma.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-value"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmillion_hip.so)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "million_hip.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, "-o", str(LIB), *[str(CSRC / s) for s in SOURCES]]
    cwd = CSRC
    if os.environ.get("MILLION_SAVE_TEMPS"):      # ISA / IR dumps go to scratch (gpurun_out/ never ships to the GPU box)
        cwd = HERE.parent / "gpurun_out" / "save_temps"
        cwd.mkdir(parents=True, exist_ok=True)
        cmd.insert(1, "-save-temps")
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=str(cwd), capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libmillion_hip.so")
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
