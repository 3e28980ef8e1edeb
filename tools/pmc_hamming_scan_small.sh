#!/bin/bash
# matrix-pipe / issue counters of the Hamming scan at 12.5 M codes x NQ (default 32) queries:
#   NQ=32 bash tools/pmc_hamming_scan_small.sh -> gpurun_out/pmc_ham_small/summary.txt
O=$(pwd)/gpurun_out/pmc_ham_small; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d "$O/p1" -- python3 tools/bench_hamming.py --n 12500000 --nq ${NQ:-32} --reps 2 > /dev/null 2> "$O/err1.txt"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d "$O/p2" -- python3 tools/bench_hamming.py --n 12500000 --nq ${NQ:-32} --reps 2 > /dev/null 2> "$O/err2.txt"
python3 - "$O" > "$O/summary.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
for p in ("p1", "p2"):
    rows = defaultdict(dict)
    for f in glob.glob(sys.argv[1] + f"/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hamming_scan_mfma" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                rows[int(r["Dispatch_Id"])]["grid"] = r.get("Grid_Size", "")
    big = sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:3]
    for d, c in big:
        print(p, "dispatch", d, {k: (v if isinstance(v, str) else round(v)) for k, v in sorted(c.items())})
PY
cat "$O/summary.txt"
find "$O" -name '*counter_collection.csv' -delete; find "$O" -name '*_agent_info.csv' -delete
