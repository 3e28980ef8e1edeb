#!/bin/bash
# rocprofv3 evidence for the audio leg: kernel stats, then the PMC digest (separate passes).  usage: bash tools/prof_audio.sh <tag>
set -o pipefail
T=${1:-a}
O=$(pwd)/gpurun_out/prof_audio_$T
mkdir -p "$O"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 tools/bench_audio.py --only fused --steps 5 > "$O/bench.json" 2> "$O/err1.txt"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$O/pmc" -- python3 tools/bench_audio.py --only fused --steps 2 > /dev/null 2> "$O/err2.txt"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
    --output-format csv -d "$O/pmc2" -- python3 tools/bench_audio.py --only fused --steps 2 > /dev/null 2> "$O/err3.txt"
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*_agent_info.csv' -delete
cat "$O/bench.json"
python3 - "$O" <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/stats/**/*kernel_stats.csv",recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
python3 tools/pmc_all.py "$O/pmc"
python3 tools/pmc_kernel.py "$O/pmc2" wang_stream
