#!/bin/bash
# rocprofv3 kernel stats for an arbitrary python tool: bash tools/prof_cmd.sh <tag> <script> [args...]; prints the top kernels
set -o pipefail
T=$1; shift
O=$(pwd)/gpurun_out/prof_$T; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$@" > "$O/out.txt" 2> "$O/err.txt"
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*_agent_info.csv' -delete
cat "$O/out.txt"
python3 - "$O" <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/stats/**/*kernel_stats.csv",recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:24]:
    if 'at::native' in r['Name'] or 'rocclr' in r['Name']: continue
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_us={float(r['TotalDurationNs'])/1e3:10.1f}")
PY
