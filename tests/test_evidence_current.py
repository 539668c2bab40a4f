"""CPU suite: the committed evidence belongs to the committed kernels (r6).

`bench.py` attaches `roofline.traffic` (HBM bytes per launch, from rocprofv3 PMC passes) to the driver's line from
profiles/traffic_latest.json -- only when that file carries the hash of the kernel sources that are running
(`bench.kernel_source_hash()`).  A kernel change without a fresh PMC pass silently turns the figure into null.  This
test fails at commit time instead: `tools/gpu_round.sh prof <tag>` refreshes the file as the LAST GPU action of a round
(and fails loudly itself when the hashes differ)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_latest_was_measured_on_the_committed_kernel_sources():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert tj["kernel_source_hash"] == bench.kernel_source_hash(), (
        "profiles/traffic_latest.json was measured on other kernel sources: run `bash tools/gpu_round.sh prof r6` on the "
        "GPU (its last step refreshes the file) and `bash tools/collect_profiles.sh r6`")
    ent = tj["entries"]
    # the headline and the workloads the line reports traffic for
    for name in ("config5", "config5_lead", "config2", "config3"):
        assert name in ent, name
        e = ent[name]
        assert e["streams"] == bench.WORKLOADS[name][0]
        src = os.path.join(ROOT, e["source"])
        assert os.path.exists(src), e["source"]
        s = json.load(open(src))
        assert s["kernel_source_hash"] == tj["kernel_source_hash"] and s["hbm_bytes_per_launch"] == e["hbm_bytes_per_launch"]
        # the summary names the committed CSV it was computed from
        assert os.path.exists(os.path.join(ROOT, s["kernel_stats_file"])), s["kernel_stats_file"]


def test_attach_traffic_uses_it_and_says_where_it_comes_from():
    rec = {"roofline": {"traffic": None}}
    bench.attach_traffic(rec, "config5", 65536, bench.kernel_source_hash())
    assert rec["roofline"]["traffic"] and "not measured in this run" in rec["roofline"]["traffic_source"]
    rec = {"roofline": {"traffic": None}}
    bench.attach_traffic(rec, "config5", 65536, "0" * 16)
    assert rec["roofline"]["traffic"] is None and rec["roofline"]["traffic_source"].startswith("none:")
