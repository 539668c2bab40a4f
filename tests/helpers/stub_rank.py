"""Stand-in for bench.py's rank side (CPU, gloo): lets tests/test_bench_launch.py exercise the
launcher half of `bench.py --gpus N` -- N processes under torch.distributed.run, rank 0's JSON
line relayed, a failing rank turning into a non-zero exit code -- without a GPU."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402  (plan / config_block: the same pure functions bench.run_rank builds its line from)

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ones = torch.ones(1, dtype=torch.int32)
dist.all_reduce(ones)
print(f"noise from rank {rank}", flush=True)
if mode == "fail" and rank == world - 1:
    os._exit(3)
if mode != "silent" and rank == 0:
    pl = bench.plan(world)
    print(json.dumps({"metric": "stub", "ranks_seen": int(ones.item()), "n_gpus": world,
                      "config": bench.config_block(pl["main"], bench.WORKLOADS[pl["main"]][0], world, "gloo"),
                      "sub_records": {k: {} for k in pl["subs"] + pl["next"]}}), flush=True)
dist.barrier() if mode != "fail" else None
dist.destroy_process_group() if mode != "fail" else None
