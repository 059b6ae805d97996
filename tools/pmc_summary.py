#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of bench.py into profiles/ (tracked).

Inputs (under gpurun_out/, produced on the GPU box by the commands in profiles/README.md):
  prof_kt/*/..._kernel_stats.csv       rocprofv3 --kernel-trace --stats
  prof_fetch|prof_write|prof_sq/*/..._counter_collection.csv   separate --pmc passes
HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled.
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
KERNEL = "attn_"            # the fused decode-attention launch: attn_stream_kernel (attn_mfma_kernel: fallback)

summary = {"kernel": KERNEL, "tag": tag}
import os


def newest(pattern):
    """gpurun merges every call's files into the same directories: keep the latest run of each."""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


ks = newest(str(src / "prof_kt" / "*" / "*_kernel_stats.csv"))
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    keep = [r for r in rows if "million::" in r["Name"] or "_ZN7million" in r["Name"]]
    with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            w.writerow(r)
    cand = [r for r in keep if KERNEL in r["Name"]]
    if cand:
        r = max(cand, key=lambda r: int(r["Calls"]))      # the variant the run actually used
        summary["kernel"] = r["Name"].split("(")[0]
        summary["rocprof_avg_ns"] = float(r["AverageNs"])
        summary["rocprof_calls"] = int(r["Calls"])
pmc = defaultdict(list)
for d in ("prof_fetch", "prof_write", "prof_sq"):
    for f in newest(str(src / d / "*" / "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                pmc[row["Counter_Name"]].append(float(row["Counter_Value"]))
means = {}
for c, v in pmc.items():
    v = v[len(v) // 4:]          # skip warm-up launches
    means[c] = sum(v) / len(v)
summary["pmc_mean_per_launch"] = means
if "FETCH_SIZE" in means:
    rd = 2.0 * means["FETCH_SIZE"] * 1024.0
    wr = means.get("WRITE_SIZE", 0.0) * 1024.0
    summary["hbm_read_bytes_per_launch"] = rd
    summary["hbm_write_bytes_per_launch"] = wr
    summary["hbm_bytes_per_launch"] = rd + wr
(out / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
bench = src / f"bench_{tag}.json"
cfg = {}
if bench.exists():
    line = [l for l in bench.read_text().splitlines() if l.startswith("{")][-1]
    (out / f"{tag}_bench.json").write_text(line + "\n")
    cfg = json.loads(line)["config"]
if "hbm_bytes_per_launch" in summary:
    (out / "pmc_traffic.json").write_text(json.dumps({
        "hbm_bytes_per_launch": summary["hbm_bytes_per_launch"], "ctx": cfg.get("ctx", 32768), "M": cfg.get("M", 64),
        "batch_per_gpu": cfg.get("batch_per_gpu", 1), "source": f"profiles/{tag}_pmc_summary.json"}, indent=1))
print(json.dumps(summary, indent=1))
