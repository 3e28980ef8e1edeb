#!/bin/bash
# Round 4's per-leg evidence beyond tools/profile_round.sh (run through gpurun from the repo root, AFTER profile_round.sh):
#   bash tools/profile_round_r04.sh
# Everything lands under gpurun_out/prof_r04_extra/ (merged back by gpurun); the files judged are copied into profiles/r04/.
set -o pipefail
O=$(pwd)/gpurun_out/prof_r04_extra
rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
say() { echo "[$(date +%H:%M:%S)] $*"; }

say "ubench";        tools/ubench_valu.bin > "$O/ubench_valu.txt" 2>&1
say "hamming sweep"; python3 tools/bench_hamming.py --n 1250000 10000000 12500000 100000000 --nq 1 8 9 16 32 64 128 256 1024 4096 > "$O/bench_hamming.jsonl" 2> "$O/ham.err"
say "hamming stats"; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ham_stats" -- python3 tools/bench_hamming.py --n 12500000 --nq 4096 > /dev/null 2> "$O/ham_prof.err"
find "$O/ham_stats" -name '*kernel_stats.csv' -exec cp {} "$O/hamming_12m5_4096q_kernel_stats.csv" \;
for q in 16 32 64; do
  say "chain $q"
  rocprofv3 --kernel-trace --output-format csv -d "$O/tr_$q" -- python3 tools/bench_hamming.py --n 12500000 --nq $q > /dev/null 2> "$O/tr_$q.err" \
    && python3 tools/trace_chain.py "$O/tr_$q" > "$O/hamming_chain_${q}q.txt"
  rm -rf "$O/tr_$q"
done
say "direct phases"
for q in 1 8; do tools/prof_direct.bin 12500000 $q 10 1; done > "$O/hamming_direct_phases.txt" 2>&1
tools/prof_direct.bin 1250000 1 10 1 >> "$O/hamming_direct_phases.txt" 2>&1
tools/prof_direct.bin 1250000 8 10 1 >> "$O/hamming_direct_phases.txt" 2>&1
tools/prof_direct.bin 100000000 8 10 1 >> "$O/hamming_direct_phases.txt" 2>&1
say "direct pmc"
bash tools/pmc_summary.sh r04 hamming_direct8 hamming_direct_kernel 12500000 -- python3 tools/bench_hamming.py --n 12500000 --nq 8 --reps 20 > "$O/pmc_direct8.log" 2>&1
bash tools/pmc_summary.sh r04 hamming_direct1 hamming_direct_kernel 12500000 -- python3 tools/bench_hamming.py --n 12500000 --nq 1 --reps 20 > "$O/pmc_direct1.log" 2>&1
NQ=32 bash tools/pmc_hamming_scan_small.sh > "$O/pmc_scan_32q.txt" 2>&1
say "search batcher"; tools/bench_search_batcher.bin 1 2 8 16 32 64 128 256 512 > "$O/search_batcher.jsonl" 2> "$O/sb.err"
say "ingest batchers"; tools/bench_batcher.bin 16 64 256 > "$O/bench_batcher.jsonl" 2> "$O/bb.err"
say "image sizes"
for c in "512 512 1" "512 512 3" "256 256 1" "256 256 3" "300 200 1" "300 200 3" "640 480 1" "640 480 3" "641 481 1" "301 200 3" "1023 767 3" "1280 720 3" "1000 1000 4" "1920 1080 3"; do
  python3 tools/bench_image_sizes.py $c 2>> "$O/img.err"
done > "$O/bench_image_sizes.jsonl"
say "uploads"; python3 tools/bench_uploads.py > "$O/bench_uploads.json" 2> "$O/up.err"
say "jpeg"
for a in "--n 1000" "--n 500" "--n 8000" "--n 1000 --sub 1" "--n 1000 --side 512" "--n 1000 --restart-rows 1"; do python3 tools/bench_jpeg.py $a 2>> "$O/jpeg.err"; done > "$O/bench_jpeg.jsonl"
say "png";     python3 tools/bench_png.py 1000 > "$O/bench_png.jsonl" 2> "$O/png.err"; python3 tools/bench_png.py 8000 >> "$O/bench_png.jsonl" 2>> "$O/png.err"
say "cosine";  python3 tools/bench_cosine.py > "$O/bench_cosine.jsonl" 2> "$O/cos.err"
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*_agent_info.csv' -delete; rm -rf "$O/ham_stats"
cp gpurun_out/pmc_hamming_direct8/summary.json "$O/hamming_direct8_pmc_summary.json" 2>/dev/null
cp gpurun_out/pmc_hamming_direct1/summary.json "$O/hamming_direct1_pmc_summary.json" 2>/dev/null
ls -la "$O"
