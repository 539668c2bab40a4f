#!/bin/bash
# Runs each hardware probe in its own process with a timeout.
cd "$(dirname "$0")"
mkdir -p ../gpurun_out
for p in galign gldslds bufalign buflds dsalign; do
  echo "=== $p"; timeout 60 ./hw_probe $p 2>&1 | tail -40; echo "rc=$?"
done 2>&1 | tee ../gpurun_out/hw_probe.log
