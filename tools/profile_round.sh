#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# 1. the default bench line, 2. rocprofv3 kernel stats of the same command, 3. rocprofv3 stats of the
# image leg alone, 4/5. FETCH_SIZE and WRITE_SIZE in separate PMC passes (never mixed with trace
# domains).  Everything lands under gpurun_out/prof_<round>/; tools/summarize_prof.py turns it into
# the files committed under profiles/<round>/.
set -eo pipefail
R=${1:-r01}
ROOT=$(pwd)
O=$ROOT/gpurun_out/prof_$R
mkdir -p "$O"
export TMPDIR=/tmp
IMG="--ann-corpus 0 --rgb-frames 0 --cosine-rows 0 --text-docs 0 --audio-seconds 0 --cpu-sample 0"
echo "[1/5] bench.py (default)"; python3 bench.py > "$O/bench_n1_full.json" 2> "$O/bench.err"
echo "[2/5] rocprofv3 --kernel-trace --stats (default bench, cpu leg off)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_full" -- python3 bench.py --cpu-sample 0 \
    > "$O/bench_under_rocprof.json" 2> "$O/rocprof_full.err"
echo "[3/5] rocprofv3 --kernel-trace --stats (image leg)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_image" -- python3 bench.py $IMG \
    > "$O/bench_image_under_rocprof.json" 2> "$O/rocprof_image.err"
echo "[4/5] rocprofv3 --pmc FETCH_SIZE (image leg)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 bench.py $IMG --steps 4 --warmup 1 \
    > /dev/null 2> "$O/rocprof_fetch.err"
echo "[5/5] rocprofv3 --pmc WRITE_SIZE (image leg)"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 bench.py $IMG --steps 4 --warmup 1 \
    > /dev/null 2> "$O/rocprof_write.err"
# keep what is merged back small: drop the per-dispatch traces, keep stats and counters
find "$O" -name '*_kernel_trace.csv' -delete
find "$O" -name '*_agent_info.csv' -delete
du -sh "$O"
