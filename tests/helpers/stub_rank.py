"""Stand-in for bench.py's rank side (CPU, gloo): lets tests/test_bench_launch.py exercise the
launcher half of `bench.py --gpus N` -- N rank processes, rank 0's JSON line relayed, a failing or
hanging rank turning into ONE diagnostic line and a non-zero exit code -- without a GPU.
Modes: ok | fail (last rank dies after the first collective) | silent (no line) | hang (last rank never
arrives at the barrier) | hang_after_headline (rank 0 has checkpointed a partial line, then the last rank hangs)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402  (plan / config_block: the same pure functions bench.run_rank builds its line from)

import time  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
hb = bench.Heartbeat(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
hb.beat("pg up")
ones = torch.ones(1, dtype=torch.int32)
dist.all_reduce(ones)
print(f"noise from rank {rank}", flush=True)
if mode == "fail" and rank == world - 1:
    hb.beat("about to die")
    os._exit(3)
if mode == "hang_after_headline" and rank == 0:
    hb.write_partial({"metric": "stub", "value": 123.0, "unit": "Msamples/s", "n_gpus": world, "steps": 1, "warmup": 0,
                      "ms_per_step": 1.0, "roofline": {"frac": 0.5}, "riders_pending": "config2"})
    hb.beat("sub-record config2")
if mode in ("hang", "hang_after_headline") and rank == world - 1:
    hb.beat("stuck before the barrier")
    time.sleep(3600)
dist.barrier()                # (a dead or hanging rank: rank 0 waits here, no result line can come out of the run)
if mode != "silent" and rank == 0:
    pl = bench.plan(world)
    # a full record shaped like run_rank's at N > 1 (figures are placeholders), through the SAME compaction
    roof = {"bound": "hbm", "achieved": 1.0, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None,
            "algorithmic_bytes_per_launch": 1, "kernel_ms": 1.0, "kernel_ms_median": 1.0, "padding": "x" * 3000}
    gather = {"ranks_seen": int(ones.item()), "gather_mode": "gather to rank 0", "gather_check": [True] * world,
              "gather_check_on_every_rank": True, "gather_every_steps": 3, "gathers_in_timed_region": 7,
              "gather_ms": {"median": 0.1, "max": 0.2, "bytes_per_rank": 1, "measured": 7}}
    sub = {"value": 1.0, "roofline": roof, "roundtrip_match_rate": 1.0, "entry": "afsk_demod_batch_uniform",
           "event_intervals": {"padding": "y" * 3000}, **gather}
    full = {"metric": "stub", "value": 1.0, "unit": "Msamples/s", "n_gpus": world, "steps": 1, "warmup": 0,
            "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16",
            "data": "synthetic",
            "config": bench.config_block(pl["main"], bench.WORKLOADS[pl["main"]][0], world, "gloo"),
            "roofline": roof, "sub_records": {k: dict(sub) for k in pl["subs"] + pl["next"]},
            "scaling_note": "z" * 3000, **gather}
    print(json.dumps(bench.compact_line(full, None)), flush=True)
dist.barrier() if mode != "fail" else None
dist.destroy_process_group() if mode != "fail" else None
