#!/bin/bash
# Issue-roofline inputs of ONE kernel -> profiles/<round>/<name>_pmc_summary.json (read by bench.py):
#   * tools/ubench_valu.bin: what integer / f32 VALU chains sustain on THIS chip in this call (lane-ops/s);
#   * three rocprofv3 --pmc passes over the given command: instruction counts, VALU / LDS busy cycles, waits.
# usage (GPU box, repo root):  bash tools/pmc_summary.sh <round> <name> <kernel-substring> <units-per-dispatch> -- <program> [args]
#   text :  bash tools/pmc_summary.sh r03 text text_hash_kernel 200000 -- python3 tools/text_only.py 200000
#   audio:  bash tools/pmc_summary.sh r03 audio wang_stream_kernel 2250000 -- python3 tools/bench_audio.py --only fused --steps 2
set -o pipefail
R=$1; NAME=$2; K=$3; UNITS=$4; shift 5
O=$(pwd)/gpurun_out/pmc_$NAME; rm -rf "$O"; mkdir -p "$O" "profiles/$R"; export TMPDIR=/tmp
tools/ubench_valu.bin > "$O/ubench.txt" 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$O/p1" -- "$@" > "$O/p1.out" 2> "$O/p1.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$O/p2" -- "$@" > "$O/p2.out" 2> "$O/p2.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d "$O/p3" -- "$@" > "$O/p3.out" 2> "$O/p3.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- "$@" > "$O/stats.out" 2> "$O/stats.err"
python3 - "$O" "$K" "$UNITS" "profiles/$R/${NAME}_pmc_summary.json" <<'PY'
import csv, glob, json, re, sys
o, ksub, units, dst = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
tot, cnt = {}, {}
for f in glob.glob(o + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            c = r["Counter_Name"]
            tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
            cnt[c] = cnt.get(c, 0) + 1
avg = {c: tot[c] / cnt[c] for c in tot}          # per dispatch
rates = {}
try:
    for line in open(o + "/ubench.txt"):
        m = re.match(r"(\S.*?)\s+([\d.]+) ms\s+([\d.]+) T lane-op/s", line)
        if m:
            rates[m.group(1).strip()] = float(m.group(3))
except OSError:
    pass
kern = None
for f in glob.glob(o + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Name"]:
            kern = {"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
            break
g = avg.get("GRBM_GUI_ACTIVE", 0.0)               # summed over the 8 XCDs
cyc = g / 8 if g else None                         # the dispatch's cycles
s = {"kernel_substring": ksub, "units_per_dispatch": units, "dispatches_seen": cnt, "counters_per_dispatch": avg,
     "kernel_stats_under_profiler": kern,
     "valu_wave_instr_per_unit": avg.get("SQ_INSTS_VALU", 0.0) / units,
     "salu_wave_instr_per_unit": avg.get("SQ_INSTS_SALU", 0.0) / units,
     "lds_wave_instr_per_unit": avg.get("SQ_INSTS_LDS", 0.0) / units,
     # SQ_ACTIVE_INST_* count, in quad-cycle samples, the cycles a SIMD's unit is issuing, summed over 1024 SIMDs
     "valu_busy_frac": avg.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (cyc * 1024) if cyc else None,
     "lds_busy_frac": avg.get("SQ_ACTIVE_INST_LDS", 0.0) * 4 / (cyc * 1024) if cyc else None,
     "lds_pipe_busy_frac": avg.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * 256) if cyc else None,
     "lds_bank_conflict_share": avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_LDS_IDX_ACTIVE"] if avg.get("SQ_LDS_IDX_ACTIVE") else None,
     "wave_wait_any_frac": avg.get("SQ_WAIT_ANY", 0.0) / avg["SQ_WAVE_CYCLES"] if avg.get("SQ_WAVE_CYCLES") else None,
     "clock_GHz_under_profiler": (cyc / (kern["avg_us"] * 1e3)) if (cyc and kern) else None,
     "ubench_valu_T_lane_ops_per_s": rates,
     "valu_peak_lane_ops_per_s": max(rates.values()) * 1e12 if rates else 39.3216e12,
     "note": "rocprofv3 --pmc in three passes + one --kernel-trace --stats pass (tools/pmc_summary.sh); peak = the best chain "
             "tools/ubench_valu.hip sustains on this chip in the same call"}
if kern:
    s["achieved_T_lane_ops_per_s_under_profiler"] = s["valu_wave_instr_per_unit"] * units * 64 / (kern["avg_us"] * 1e-6) / 1e12
# names bench.py's text leg reads
s["valu_wave_instr_per_doc"] = s["valu_wave_instr_per_unit"]
json.dump(s, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in s.items() if k not in ("counters_per_dispatch", "dispatches_seen")}, indent=1))
PY
cp "profiles/$R/${NAME}_pmc_summary.json" "$O/summary.json" 2>/dev/null; rm -rf "$O"/p1 "$O"/p2 "$O"/p3 "$O"/stats
