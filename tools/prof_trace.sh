#!/bin/bash
# per-dispatch durations (us) of kernels matching a substring, in dispatch order: bash tools/prof_trace.sh <substr> <script> [args...]
K=$1; shift
O=$(pwd)/gpurun_out/prof_trace; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 "$@" > "$O/out.txt" 2> "$O/err.txt"
python3 - "$O" "$K" <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[(r['Kernel_Name'].split('(')[0][-30:], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, int(r['Start_Timestamp'])) for r in rows if any(s in r['Kernel_Name'] for s in sys.argv[2].split(','))]
t0=None
for n,d,s in sel[-40:]:
    if t0 is None: t0=s
    print(f"{n:30s} {d:9.1f} us   start +{(s-t0)/1e3:9.1f}")
PY
rm -rf "$O"
