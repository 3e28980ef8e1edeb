#!/bin/bash
# The per-leg evidence of a round beyond tools/profile_round.sh (run through gpurun from the repo root):
#   bash tools/profile_round_extra.sh r02
# audio (kernel stats + PMC digest), Hamming at the shard sizes (sweep + kernel stats), PNG front end (kernel stats).
set -o pipefail
R=${1:-r02}
O=$(pwd)/gpurun_out/prof_${R}_extra
rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
echo "[audio]"; bash tools/prof_audio.sh $R > "$O/audio_digest.txt" 2>&1
cp gpurun_out/prof_audio_$R/bench.json "$O/audio_bench.json" 2>/dev/null
find gpurun_out/prof_audio_$R -name '*kernel_stats.csv' -exec cp {} "$O/audio_kernel_stats.csv" \;
echo "[hamming]"; python3 tools/bench_hamming.py --n 10000000 12500000 100000000 --nq 1 16 64 1024 4096 > "$O/bench_hamming.jsonl" 2> "$O/ham.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ham_stats" -- python3 tools/bench_hamming.py --n 12500000 --nq 4096 > /dev/null 2> "$O/ham_prof.err"
find "$O/ham_stats" -name '*kernel_stats.csv' -exec cp {} "$O/hamming_12m5_kernel_stats.csv" \;
echo "[png]"; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/png_stats" -- python3 tools/bench_png.py 1000 > "$O/png_bench.json" 2> "$O/png_prof.err"
find "$O/png_stats" -name '*kernel_stats.csv' -exec cp {} "$O/png_kernel_stats.csv" \;
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*_agent_info.csv' -delete
rm -rf "$O/ham_stats" "$O/png_stats"
ls -la "$O"
