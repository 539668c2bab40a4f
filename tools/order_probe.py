#!/usr/bin/env python3
"""Why are later measurements of one bench.py process 2-4 % slower than the first?  Same Shard measured repeatedly
(thermal / clocks), then after freeing and re-allocating the 6.29 GB buffer (allocator / page placement), then after
10 s of idling."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
args = ap.parse_args([])
for k, v in dict(gpus=1, steps=20, warmup=5, workload="config5", sub="", min_region_ms=50.0, entry="auto", next_reps=3,
                 wav_files=256, sub_steps=0, rates_steps=0, cpu_budget_s=1.0, sub_cpu_sample=64, streams=0, bauds="",
                 rate_order="cycle", preroll_ms=300.0, no_cpu_baseline=True, cpu_sample_streams=0, gather_every=0,
                 gather_mode="root", dist_backend="nccl", share_gpu0=False, force_gather=False).items():
    setattr(args, k, v)
ctx = bench.Ctx(args)
torch = ctx.torch
def run(sh, tag):
    rec, _ = bench.measure(ctx, sh, 20, 5, 300.0, 0, 50.0)
    print(f"{tag:34s} frac {rec['roofline']['frac']:.4f}  kernel_ms {rec['roofline']['kernel_ms']:.4f}", flush=True)
sh = bench.Shard(ctx, "config5", 65536)
for i in range(4):
    run(sh, f"same shard, pass {i}")
ptr0 = sh.inputs[0].data_ptr()
del sh; torch.cuda.empty_cache()
sh = bench.Shard(ctx, "config5", 65536)
run(sh, f"re-allocated (same address: {sh.inputs[0].data_ptr() == ptr0})")
keep = [torch.empty(1 << 30, dtype=torch.uint8, device=ctx.dev) for _ in range(3)]   # shift the next allocation
del sh; torch.cuda.empty_cache()
sh = bench.Shard(ctx, "config5", 65536)
run(sh, f"re-allocated behind 3 GB (same: {sh.inputs[0].data_ptr() == ptr0})")
time.sleep(10)
run(sh, "after 10 s idle")
run(sh, "again")
