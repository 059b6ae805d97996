#!/bin/bash
# Round profile set, run on the GPU box from the repo root (gpurun): the bench line, rocprofv3 kernel stats at every
# BASELINE shape, the PMC passes (each counter set in its own run, --kernel-trace only), the in-kernel stamps.
# Outputs under gpurun_out/; tools/pmc_summary.py <tag> then copies the summaries into profiles/.
#   usage: bash tools/collect_profiles.sh r04
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-e2e --no-fp16-baseline"
run() { echo "== $*"; timeout -k 10 400 "$@" || { echo "FAILED: $*"; exit 1; }; }
# 1. the bench line itself (cpu baseline, fp16 baselines and the e2e record included)
run python3 $ROOT/bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || exit 1
# 2. kernel stats per shape
kt() { d=$1; shift; rm -rf $OUT/$d; run rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$d -- python3 $ROOT/bench.py --steps 64 --warmup 8 $Q "$@" > $OUT/$d.log 2>&1 || exit 1; }
kt prof_kt
kt prof_kt_bs2 --batch-per-gpu 2
kt prof_kt_bs4 --batch-per-gpu 4
kt prof_kt_128k --ctx 131072 --M 32
kt prof_kt_bs8 --batch-per-gpu 8
kt prof_kt_l2 --ctx 4096 --nh-k 32
kt prof_kt_g16 --nh 128
# 2b. the same traces split by launch mode (replayed steps / eager behind a sleep / graph of the launches) + ab_bench on this box
{
  for d in prof_kt:"1 request" prof_kt_bs2:"2 requests" prof_kt_bs4:"4 requests" prof_kt_bs8:"8 requests"; do
    python3 $ROOT/tools/trace_phases.py $OUT/${d%%:*} --label "${d#*:}" || exit 1
  done
  echo "# tools/ab_bench.py on the same box (one graph of the launches, replayed; no sleep in front)"
  (cd $ROOT && python3 tools/ab_bench.py --cfg 1,32768,64 2,32768,64 4,32768,64 8,32768,64 1,131072,64 1,131072,32 | grep us/launch) || exit 1
  echo "# shapes that left the streaming kernel in rounds 2-3 (more than 64 rounds per wave): now kind 1"
  (cd $ROOT && python3 tools/ab_bench.py --cfg 16,40000,64 --layers 4 | grep us/launch) || exit 1
  (cd $ROOT && python3 tools/ab_bench.py --cfg 8,32768,64 --nh-k 32 --layers 4 | grep us/launch) || exit 1
  (cd $ROOT && python3 tools/ab_bench.py --cfg 32,20000,64 --C 128 --layers 4 | grep us/launch) || exit 1
} > $OUT/shapes_$TAG.txt 2>&1
# 2c. the like-for-like N = 1 point of configs[3] (2 requests per GPU)
run python3 $ROOT/bench.py --gpus 1 --batch-per-gpu 2 --no-e2e > $OUT/bench_${TAG}_bs2.json 2> $OUT/bench_${TAG}_bs2.err || exit 1
# 3. counters (eager launches so that every dispatch is attributed)
pmc() { d=$1; shift; rm -rf $OUT/$d; run rocprofv3 --pmc "$@" --output-format csv -d $OUT/$d -- python3 $ROOT/bench.py --steps 8 --warmup 2 $Q --no-graph > $OUT/$d.log 2>&1 || exit 1; }
pmc prof_fetch FETCH_SIZE
pmc prof_write WRITE_SIZE
pmc prof_sq SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
# 4. in-kernel stamps
cd $ROOT
run python3 tools/stamp_profile.py > $OUT/stamps_$TAG.txt 2>&1 || exit 1
run python3 tools/stamp_profile.py --bs 2 > $OUT/stamps_${TAG}_bs2.txt 2>&1 || exit 1
run python3 tools/host_overhead.py > $OUT/host_overhead_$TAG.txt 2>&1 || exit 1
run python3 tools/encode_bench.py > $OUT/encode_bench_$TAG.txt 2>&1 || exit 1
echo done
