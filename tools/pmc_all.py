"""Digest of a rocprofv3 --pmc run per kernel: LDS occupancy and bank-conflict share, VALU occupancy, wait split.

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
              SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d <dir> -- python3 bench.py ...
    python tools/pmc_all.py <dir>

(this is how the audio twiddle-table and the cosine staging bank conflicts of round 1 were found)"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ucfp" not in n: continue
        n = n.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[n] += 1
rows = []
for n, c in tot.items():
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8
    if g < 1e5 * max(cnt[n],1): continue
    cu_cycles = g * 256
    rows.append((g, n, cnt[n], c))
for g, n, k, c in sorted(rows, reverse=True)[:14]:
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0); bc = c.get("SQ_LDS_BANK_CONFLICT", 0)
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(f"{n:60s} n={k:3d} Mcyc/launch={g/k/1e6:8.2f} LDSbusy={lds/(g*256):.2f} conflict/LDS={bc/max(lds,1):.2f} "
          f"VALUbusy={c.get('SQ_ACTIVE_INST_VALU',0)*4/(g*1024):.2f} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f}")
