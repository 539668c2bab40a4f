"""bench.py's self-launch half (`python bench.py --gpus N` from a bare interpreter): tested on
CPU with a gloo stub in place of the rank side."""
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "stub_rank.py")


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")
ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def bench_mod():
    import bench
    return bench


def _launch(mode):
    import bench
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.self_launch(2, [mode], script=STUB)
    return rc, buf.getvalue()


def test_launcher_relays_rank0_line_only():
    rc, out = _launch("ok")
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                      # ONE JSON line on stdout, the chatter went to stderr
    doc = json.loads(lines[0])
    assert doc["ranks_seen"] == 2 and doc["n_gpus"] == 2
    # the N > 1 line obeys the same cap as the N = 1 line and still carries what the scaling run is judged on
    assert len(lines[0]) < bench_mod().LINE_CAP
    assert doc["gather_check"] == [True, True] and doc["gather_ms"]["median"] == 0.1
    assert set(CONTRACT_KEYS) <= set(doc), set(CONTRACT_KEYS) - set(doc)
    assert doc["sub_records"]["config2"]["gather_check"] is True


def test_launcher_nonzero_when_a_rank_fails():
    rc, _ = _launch("fail")
    assert rc != 0


def test_launcher_nonzero_without_a_result_line():
    rc, out = _launch("silent")
    assert rc != 0 and out.strip() == ""


def test_bare_interpreter_refuses_more_gpus_than_visible():
    """No GPU here: the parent must say so and exit non-zero without starting ranks."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could really run it")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr and p.stdout.strip() == ""


def test_kernel_source_hash_is_stable_and_sensitive(tmp_path):
    import bench
    h = bench.kernel_source_hash()
    assert h == bench.kernel_source_hash() and len(h) == 16


def test_headline_workload_is_the_same_at_every_n():
    """One weak-scaling curve: the N = 1 and the N > 1 lines name the same workload (the config5
    shard, BASELINE.json configs[4] per GPU); only what rides along differs."""
    import bench
    blocks = {}
    for world in (1, 2, 4, 8):
        pl = bench.plan(world)
        assert pl["main"] == bench.HEADLINE == "config5"
        cb = bench.config_block(pl["main"], bench.WORKLOADS[pl["main"]][0], world)
        assert cb["workload"].startswith("configs[4]") and cb["streams_per_gpu"] == 65536
        assert cb["streams_total"] == 65536 * world and cb["bauds"] == [1200]
        blocks[world] = cb
    assert blocks[1]["workload"] == blocks[2]["workload"] == blocks[8]["workload"]
    assert blocks[8]["streams_total"] == 524288
    one, two = bench.plan(1), bench.plan(2)
    assert one["subs"] == ["config2", "config3", "config4"] and one["next"] == list(bench.NEXT_ROWS) + ["rates_4096", "rates_65536"]
    assert len(bench.ALL_RATES) == 36 and set((300, 1200, 2400, 100, 160, 96, 24, 12000)) <= set(bench.ALL_RATES)
    assert two["subs"] == ["config2"] and two["next"] == []
    # an explicit workload drops the riders unless --sub lists them; the next rows are never a headline
    assert bench.plan(1, "config2") == {"main": "config2", "subs": [], "next": []}
    assert "f5_wav_egress" in bench.NEXT_ROWS
    assert bench.plan(1, "custom", "config3,f2_gate") == {"main": "custom", "subs": ["config3"], "next": ["f2_gate"]}
    with pytest.raises(SystemExit):
        bench.plan(1, "f1_modulate")
    with pytest.raises(SystemExit):
        bench.plan(1, "", "nonsense")


def test_stub_rank_line_names_the_headline_workload():
    """The relayed N = 2 line and bench.plan(1) agree on the workload name."""
    import bench
    rc, out = _launch("ok")
    assert rc == 0
    doc = json.loads([ln for ln in out.splitlines() if ln.strip()][0])
    assert doc["config"]["workload"] == bench.config_block(bench.plan(1)["main"], 65536, 1)["workload"]


@pytest.mark.parametrize("record", ["r3_bench_n1.json", "r3_bench_n8_diagnostic_gloo_shared_gpu.json",
                                    "r2_bench_n1.json"])
def test_result_line_is_compact_and_complete(record):
    """The contract that broke in round 3 (a 26 KB line, of which the driver kept the last 8 KB): whatever the full
    record holds, the ONE line on stdout stays under LINE_CAP and keeps every key the driver and the judge parse."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", record)))
    line = bench.compact_line(full, "gpurun_out/bench_full_n1.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_CAP <= 4096, len(text)
    assert set(CONTRACT_KEYS) <= set(line)
    assert set(ROOF_KEYS) <= set(line["roofline"]) and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert line["value"] == full["value"] and line["config"] == full["config"]
    assert line["full_record"] == "gpurun_out/bench_full_n1.json"
    if full["n_gpus"] == 1:
        cb = line["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["value"] == full["cpu_baseline"]["value"]
        assert line["match_rate"] == full["match_rate"]
    else:
        assert line["ranks_seen"] == full["ranks_seen"] and line["gather_check"] == full["gather_check"]
    for name, sub in (full.get("sub_records") or {}).items():
        got = line["sub_records"][name]
        if "roofline" in sub:
            assert got["frac"] == sub["roofline"]["frac"] and got["value"] == sub["value"]


def test_result_line_survives_a_bloated_record():
    """Riders that do not exist yet cannot push the line over the cap: the summaries go, the contract stays."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_n1.json")))
    full["sub_records"].update({f"future_{i}": {"value": i, "roofline": {"frac": 0.5}, "match_rate": 1.0,
                                                "entry": "e" * 40} for i in range(200)})
    line = bench.compact_line(full, None)
    assert len(json.dumps(line)) < bench.LINE_CAP
    assert set(CONTRACT_KEYS) <= set(line) and line["sub_records"] == "see full_record"
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and "cpu_baseline" in line


def test_full_record_goes_to_a_file_not_to_the_streams(tmp_path, monkeypatch, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_n1.json")))
    path = bench.write_full_record(full, 1)
    assert path == os.path.join("gpurun_out", "bench_full_n1.json")
    assert json.load(open(tmp_path / path)) == full
    cap = capsys.readouterr()
    assert cap.out == "" and cap.err == ""


def test_cpu_baseline_calibration_is_a_tracked_artefact():
    """bench.py's 'reference-shaped' CPU figure (oracle/pyref.py on one core) is read with a ratio measured against
    the REAL reference in the build container: the ratio lives in a committed file written by
    tools/calibrate_cpu_reference.py, and the bench line carries it."""
    import bench
    cj = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_calibration.json")))
    assert cj["generated_by"] == "tools/calibrate_cpu_reference.py" and cj["cores_online"] >= 1 and cj["cpu"]
    assert set(cj["by_baud"]) == {"300", "1200", "2400"}
    for row in cj["by_baud"].values():
        assert 0.8 <= row["pyref_over_reference"] <= 1.3, row
        assert row["reference_msamples_per_s"] > 0 and row["c_oracle_over_reference"] > 10
    assert 0.8 <= cj["pyref_over_reference_1200"] <= 1.3
    cal = bench.cpu_calibration()
    assert cal["file"] == "profiles/cpu_reference_calibration.json"
    assert cal["pyref_over_reference"] == cj["pyref_over_reference_1200"]
    # and it survives the compaction of the result line
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_n1.json")))
    full["cpu_baseline"]["calibration"] = cal
    assert bench.compact_line(full, None)["cpu_baseline"]["calibration"] == cal
