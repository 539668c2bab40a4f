#!/bin/bash
# k7: instruction counts per wave (PMC sets A + B) for the rates that matter, then the steady-state per-rate table
cd "$(dirname "$0")/../.."
SETS="A B" bash tools/pmc_sets.sh r5k7 "config5|--workload config5" "u6000|--workload custom --bauds 6000 --streams 65536" \
   "u12000|--workload custom --bauds 12000 --streams 65536" "u3000|--workload custom --bauds 3000 --streams 65536" \
   "u160|--workload custom --bauds 160 --streams 65536" "u375|--workload custom --bauds 375 --streams 65536" \
   "u800|--workload custom --bauds 800 --streams 65536" "u300|--workload custom --bauds 300 --streams 65536" 2>&1 | grep -v "^W2026" | tee gpurun_out/r5_exp7_pmc.txt
timeout -k 10 600 python bench.py --workload config5 --sub rates_65536 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_exp7_rates_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full_n1_config5.json"))
r = d["sub_records"]["rates_65536"]
print("headline", d["roofline"]["frac"], "min", r["min_frac"], "median", r["median_frac"], "max", r["max_frac"])
for b, v in sorted(r["by_baud"].items(), key=lambda kv: kv[1]["frac"])[:14]:
    print(b, v["bit_frames"], v["frac"], v["kernel_ms"], v["roundtrip_match_rate"])
PY
