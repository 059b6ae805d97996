#!/usr/bin/env python3
"""Development helper: register / spill / scratch figures of the kernels in a built library, and (optionally) the
instruction-class sequence of one kernel - what hipcc made of a schedule.

  python tools/kernel_meta.py [lib.so] [substring of the kernel name] [--seq]

Reads the .hip_fatbin section of the library (llvm-objcopy), splits it into its gfx950 code objects and prints
.vgpr_count / .agpr_count / .vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size from the notes.  With
--seq the kernel is disassembled and every instruction mapped to one letter: M mfma, E v_exp, v other VALU, D ds_read,
d other LDS, G global / buffer, S scratch, w s_waitcnt, | s_barrier, J branch, n s_nop, a v_accvgpr, s other scalar."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = "/opt/rocm/lib/llvm/bin/"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = args[0] if args else str(ROOT / "million_amd" / "libmillion_hip.so")
pat = args[1] if len(args) > 1 else ""
seq = "--seq" in sys.argv

with tempfile.TemporaryDirectory() as td:
    fat = Path(td) / "fat.bin"
    subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, str(fat)], check=True)
    d = fat.read_bytes()
    i = k = 0
    while True:
        j = d.find(b"\x7fELF", i)
        if j < 0:
            break
        shoff = struct.unpack_from("<Q", d, j + 0x28)[0]
        shentsize, shnum = struct.unpack_from("<HH", d, j + 0x3A)
        size = shoff + shentsize * shnum
        co = Path(td) / f"co_{k}.o"
        co.write_bytes(d[j:j + size])
        k += 1
        i = j + size
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", str(co)], capture_output=True, text=True).stdout
        cur, row, hit = None, {}, []
        for line in notes.split("\n"):
            m = re.match(r"\s+\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\S+)", line)
            if not m:
                continue
            key, val = m.groups()
            if key == "agpr_count":          # first key of a kernel's record (keys are sorted)
                row = {"agpr_count": val}
            else:
                row[key] = val
            if key == "vgpr_spill_count" and pat in row.get("name", ""):
                print(f"{row['name'][:90]:90s} vgpr {row.get('vgpr_count')} agpr {row.get('agpr_count')} vgpr_spill {row.get('vgpr_spill_count')} "
                      f"sgpr_spill {row.get('sgpr_spill_count')} scratch {row.get('private_segment_fixed_size')} B")
                hit.append(row["name"])
        if seq and hit and pat:
            dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True).stdout.split("\n")
            start = [n for n, l in enumerate(dis) if hit[0] in l and l.endswith(">:")]
            if not start:
                continue
            s = []
            for line in dis[start[0] + 1:]:
                if line.endswith(">:") and "million" in line:
                    break
                m = re.match(r"\s+(\S+)", line)
                if not m:
                    continue
                op = m.group(1)
                s.append("M" if op.startswith("v_mfma") else "E" if op.startswith("v_exp") else "D" if op.startswith("ds_read") else
                         "d" if op.startswith("ds_") else "w" if op.startswith("s_waitcnt") else "|" if op.startswith("s_barrier") else
                         "J" if op.startswith(("s_cbranch", "s_branch")) else "a" if op.startswith("v_accvgpr") else
                         "S" if op.startswith("scratch") else "G" if op.startswith(("global", "buffer")) else
                         "v" if op.startswith("v_") else "n" if op.startswith("s_nop") else "s")
            s = "".join(s)
            print(f"{hit[0][:80]}: {len(s)} instructions")
            for n in range(0, len(s), 160):
                print(s[n:n + 160])
