import csv, glob, sys, collections
# prints the kernel timeline of the LAST search in the trace: start offsets and durations
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last but one hamming_final_select, walked back to its search's bound pass
idx = [i for i, r in enumerate(rows) if "hamming_final_select" in r["Kernel_Name"]]
last = idx[-2]
j = last
while j > 0 and "hamming_bound_mfma" not in rows[j]["Kernel_Name"]: j -= 1
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:last + 4]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1000:8.1f} us  +{(e - s) / 1000:7.1f} us  {r["Kernel_Name"][:60]}  grid {r.get("Grid_Size_X","?")} wg {r.get("Workgroup_Size_X","?")}')
