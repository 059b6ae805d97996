#!/usr/bin/env python3
"""Development helper: build A/B variants of libmillion_hip.so with different -DMILLION_EXP=<mask> values (and -DMILLION_DEV_BUILD, without which
csrc/dev_switches.h refuses the switch) into build/ab/ (git-ignored, but shipped to the GPU box), selected at run time with MILLION_HIP_LIB=<path>."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from million_amd import build as B  # noqa: E402

out = ROOT / "build" / "ab"
out.mkdir(parents=True, exist_ok=True)
procs = []
for m in sys.argv[1:]:
    lib = out / f"libmillion_exp{m}.so"
    cmd = [B.hipcc(), *B.FLAGS, "-DMILLION_DEV_BUILD=1", f"-DMILLION_EXP={m}", "-o", str(lib), *[str(B.CSRC / s) for s in B.SOURCES]]
    procs.append((m, subprocess.Popen(cmd, cwd=str(B.CSRC), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
for m, p in procs:
    o, _ = p.communicate()
    print(m, "ok" if p.returncode == 0 else "FAILED\n" + o)
