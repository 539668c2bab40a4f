#!/bin/bash
# HBM bytes fetched per launch (FETCH_SIZE x 2 on gfx950) against the algorithmic bytes, uniform kernels at 65536 streams
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
for b in "$@"; do
  rm -rf gpurun_out/pmc_fetch_$b
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$b -- python3 $R/bench.py --workload custom --bauds $b --streams 65536 --sub "" --steps 6 --warmup 2 --preroll-ms 0 --min-region-ms 0 --no-cpu-baseline 2>/dev/null | tail -1 > $R/gpurun_out/pmc_fetch_$b.json )
  python - <<PY
import csv, glob, json
rows = [r for f in glob.glob("gpurun_out/pmc_fetch_$b/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
        if "demod" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = [float(r["Counter_Value"]) for r in rows][-6:]
d = json.loads(open("gpurun_out/pmc_fetch_$b.json").read())
alg = d["roofline"]["algorithmic_bytes_per_launch"]
fetched = 2 * 1024 * sum(v) / len(v)
print("$b baud: fetched %.0f MB, algorithmic %.0f MB, ratio %.3f, buffer %.0f MB" % (fetched / 1e6, alg / 1e6, fetched / alg, 65536 * 96000 / 1e6))
PY
done
