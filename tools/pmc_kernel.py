"""Sum rocprofv3 --pmc counter_collection.csv rows per counter for kernels whose name contains a substring.
usage: python tools/pmc_kernel.py <dir> <kernel-substring>"""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} dispatches={cnt[k]:4d} avg={tot[k] / cnt[k]:.4g}")
