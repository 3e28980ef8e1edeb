import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cosine_norms" in r["Kernel_Name"]]
j = idx[-2]; e = idx[-1]
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:e+1]:
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1000:8.1f} us  +{(en - s) / 1000:7.1f} us  {r["Kernel_Name"][:70]}')
