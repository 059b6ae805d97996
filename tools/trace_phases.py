#!/usr/bin/env python3
"""Split a rocprofv3 --kernel-trace of `bench.py` into its phases and print, per phase, the mean duration of the fused
decode-attention kernel and the mean gap to the next dispatch of it.

bench.py launches the attention kernel in four ways, in this order: (1) inside the replayed step graphs (fused append, device
lengths), (2) eagerly with an event pair around every launch (two passes), (3) eagerly back to back (three regions), (4) as one
captured graph of the same launches, replayed (one warm replay + three regions).  Every region of (2)-(4) is preceded by a
device-side sleep kernel, which this tool uses as the separator.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --batch-per-gpu 8 --no-e2e --no-cpu-baseline
    python tools/trace_phases.py gpurun_out/kt [--label "8 requests"]
"""
import argparse
import csv
import glob
import sys

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--label", default="")
a = ap.parse_args()
files = glob.glob(f"{a.dir}/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit(f"no *kernel_trace.csv under {a.dir}")
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
phases, cur = [[]], None
for st, en, name in rows:
    if "sleep" in name.lower() or "spin" in name.lower():
        phases.append([])
        continue
    if "attn_stream_kernel" in name or "attn_mfma_kernel" in name or "attn_lean_kernel" in name:
        phases[-1].append((st, en))
names = ["steps (replayed step graphs: fused append, device lengths)"]
for ph in phases[1:]:
    names.append(None)
labels = {1: "eager behind a sleep, event pair per launch (warm-up pass)", 2: "eager behind a sleep, event pair per launch",
          3: "eager behind a sleep, back to back, region 1", 4: "eager behind a sleep, back to back, region 2",
          5: "eager behind a sleep, back to back, region 3 + the graph of the launches (capture run, warm replay, 3 timed replays)"}
print(f"# {a.label}  ({len(rows)} dispatches, {sum(len(p) for p in phases)} of the attention kernel, {len(phases) - 1} sleep separators)")
print(f"{'phase':62s} {'launches':>8s} {'kernel us':>10s} {'gap us':>8s} {'period us':>10s}")
if len(phases) >= 6 and len(phases[5]) > len(phases[4]):      # region 3 and the graph replays share the last separator: split by count
    n_eager = len(phases[4])
    tail = phases[5][n_eager:]
    phases[5] = phases[5][:n_eager]
    labels[5] = "eager behind a sleep, back to back, region 3"
    phases.append(tail)
    labels[len(phases) - 1] = "graph of the launches, no sleep in front (warm replay + 3 timed replays)"
for i, ph in enumerate(phases):
    if len(ph) < 2:
        continue
    dur = sum(e - s for s, e in ph) / len(ph) / 1e3
    gaps = [ph[k + 1][0] - ph[k][1] for k in range(len(ph) - 1)]
    gaps = [g for g in gaps if g < 200000]          # a step boundary / other kernels between two attention launches: not a gap of this mode
    gap = sum(gaps) / max(1, len(gaps)) / 1e3
    label = names[0] if i == 0 else labels.get(i, f"phase {i}")
    print(f"{label:62s} {len(ph):8d} {dur:10.2f} {gap:8.2f} {dur + gap:10.2f}")
