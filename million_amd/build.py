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
LIB_DEBUG_IDS = HERE / "libmillion_hip_dbgids.so"      # same sources with -DMILLION_DEBUG_CHECK_IDS (page ids bounds-checked)
SOURCES = ["million_api.hip", "encode.hip", "attn_generic.hip", "attn_tile.hip", "attn_mfma.hip", "prefill.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-value", "-Wno-tautological-bitwise-compare", "-Wno-inline-asm"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmillion_hip.so)")


def needs_build(lib: Path = LIB) -> bool:
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "million_hip.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False, debug_ids: bool = False) -> Path:
    """Compile every source to an object in parallel (the attention kernels dominate: ~25 s), then link.
    debug_ids: the diagnostic variant libmillion_hip_dbgids.so (load it with MILLION_HIP_LIB=...): the three decode-attention
    kernels map page ids outside the pools to page 0 and count them (million_debug_bad_page_ids)."""
    LIB = LIB_DEBUG_IDS if debug_ids else globals()["LIB"]
    if not force and not needs_build(LIB):
        return LIB
    save_temps = bool(os.environ.get("MILLION_SAVE_TEMPS"))
    objdir = HERE.parent / "build" / ("obj_dbgids" if debug_ids else "obj")
    # ISA / IR dumps go to scratch (gpurun_out/ never ships to the GPU box)
    cwd = HERE.parent / "gpurun_out" / "save_temps" if save_temps else objdir
    objdir.mkdir(parents=True, exist_ok=True)
    cwd.mkdir(parents=True, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + (["-DMILLION_DEBUG_CHECK_IDS=1"] if debug_ids else [])
    cmds = []
    for src in SOURCES:
        cmd = [hipcc(), *cflags, "-c", "-o", str(objdir / (src + ".o")), str(CSRC / src)]
        if save_temps:
            cmd.insert(1, "-save-temps")
        cmds.append(cmd)
    if verbose:
        for cmd in cmds:
            print(" ".join(cmd), flush=True)
    procs = [subprocess.Popen(cmd, cwd=str(cwd), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for cmd in cmds]
    failed = False
    for src, pr in zip(SOURCES, procs):
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(out)
        elif verbose and out:
            sys.stderr.write(out)
    if failed:
        raise RuntimeError(f"hipcc failed building {LIB.name}")
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *[str(objdir / (src + ".o")) for src in SOURCES]]
    if verbose:
        print(" ".join(link), flush=True)
    r = subprocess.run(link, cwd=str(objdir), capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed linking {LIB.name}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug_ids="--debug-ids" in sys.argv))
