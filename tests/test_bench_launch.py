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


def _launch(mode, **kw):
    import bench
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.self_launch(2, [mode], script=STUB, **kw)
    return rc, buf.getvalue()


def _one_line(out):
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                      # ONE JSON line on stdout whatever happened
    assert len(lines[0]) < bench_mod().LINE_CAP
    return json.loads(lines[0])


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


def test_launcher_says_which_rank_died():
    """A rank that dies: the launcher terminates the others (rank 0 sits in a barrier that will never complete)
    and prints ONE line that names the rank, its exit code and every rank's last heartbeat -- quickly."""
    import time
    t0 = time.monotonic()
    rc, out = _launch("fail", deadline_s=120.0, steps=7, warmup=3)
    took = time.monotonic() - t0
    doc = _one_line(out)
    assert rc == bench_mod().EXIT_RANK_FAILED and took < 60, (rc, took)
    assert doc["value"] is None and doc["n_gpus"] == 2 and doc["steps"] == 7 and doc["warmup"] == 3
    assert doc["metric"] == bench_mod().METRIC and "rank 1 -> exit code 3" in doc["error"]
    assert doc["heartbeats"]["1"]["phase"] == "about to die" and doc["heartbeats"]["0"]["phase"] == "pg up"
    assert doc["phase"] == "pg up" and doc["printed_by"] == "launcher"


def test_launcher_deadline_on_a_hanging_rank():
    """A rank that never arrives: after --deadline-s the launcher kills every rank and the line says who was where."""
    import time
    t0 = time.monotonic()
    rc, out = _launch("hang", deadline_s=8.0, grace_s=2.0)
    took = time.monotonic() - t0
    doc = _one_line(out)
    assert rc == bench_mod().EXIT_DEADLINE and took < 40, (rc, took)
    assert doc["value"] is None and "deadline of 8 s" in doc["error"] and "1" in doc["error"]
    assert doc["heartbeats"]["1"]["phase"] == "stuck before the barrier"
    assert doc["heartbeats"]["0"]["age_s"] is not None


def test_launcher_keeps_the_headline_when_a_rider_hangs():
    """Once rank 0 has checkpointed a line with numbers, a later hang costs the riders, not the headline."""
    rc, out = _launch("hang_after_headline", deadline_s=8.0, grace_s=2.0)
    doc = _one_line(out)
    assert rc == bench_mod().EXIT_DEADLINE
    assert doc["value"] == 123.0 and doc["incomplete"] is True and doc["riders_pending"] == "config2"
    assert "deadline" in doc["error"] and doc["heartbeats"]["0"]["phase"] == "sub-record config2"


def test_launcher_nonzero_without_a_result_line():
    rc, out = _launch("silent")
    doc = _one_line(out)
    assert rc == bench_mod().EXIT_NO_LINE and doc["value"] is None and "no result line" in doc["error"]


def test_rank_watchdog_prints_the_line_under_an_external_launcher(tmp_path):
    """The driver launches the ranks with torch.distributed.run itself: then no bench.py launcher exists and the
    in-rank watchdog must produce the line.  Rank 1 hangs in its main thread (a sleep stands in for a collective
    that never completes), the deadline passes, rank 0's watchdog thread prints and both exit non-zero."""
    code = (
        "import os, sys, time, types\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "rank = int(os.environ['RANK'])\n"
        "hb = bench.Heartbeat(rank, 2)\n"
        "args = types.SimpleNamespace(steps=5, warmup=1)\n"
        "dog = bench.Watchdog(hb, args, 3.0)\n"
        "hb.beat('pg up')\n"
        "if rank == 0: hb.beat('warm-up + timed region')\n"
        "time.sleep(600)\n")
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
               AFSK_BENCH_HB_DIR=str(tmp_path))
    env.pop("AFSK_BENCH_LAUNCHER", None)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=60) for p in procs]
    assert [p.returncode for p in procs] == [bench_mod().EXIT_DEADLINE] * 2
    assert outs[1][0].strip() == ""                                   # only one rank prints
    doc = _one_line(outs[0][0])
    assert doc["printed_by"] == "rank 0" and doc["value"] is None and doc["n_gpus"] == 2 and doc["steps"] == 5
    assert "deadline of 3 s" in doc["error"] and doc["phase"].startswith("warm-up + timed region [main thread at")
    assert doc["heartbeats"]["1"]["phase"].startswith("aborted in 'pg up'") or doc["heartbeats"]["1"]["phase"] == "pg up"


def test_rank_watchdog_sigterm_while_the_main_thread_is_stuck(tmp_path):
    """torchrun terminates the survivors of a failed rank with SIGTERM.  The main thread may sit in a C call that
    never returns to the interpreter (here: a blocking read on a pipe nobody writes to, through ctypes), so a Python
    signal handler would never run -- the watchdog thread still prints the line and ends the process.
    Rank 0 is gone (its heartbeat names a dead pid): the surviving rank 1 takes over the printing."""
    import signal
    import time
    code = (
        "import os, sys, time, types, ctypes\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "hb = bench.Heartbeat(1, 2)\n"
        "dog = bench.Watchdog(hb, types.SimpleNamespace(steps=5, warmup=1), 300.0)\n"
        "hb.beat('timed without the gather')\n"
        "r, w = os.pipe()\n"
        "print('READY', file=sys.stderr, flush=True)\n"
        "while True:\n"          # (a C loop that retries on EINTR, like a collective's progress loop)
        "    ctypes.CDLL(None).read(r, ctypes.create_string_buffer(8), 8)\n")
    (tmp_path / "rank0.json").write_text(json.dumps({"rank": 0, "pid": 2 ** 22 + 12345, "phase": "shard built",
                                                     "t": time.time(), "t_start": time.time(), "since_start_s": 1.0}))
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", AFSK_BENCH_HB_DIR=str(tmp_path))
    env.pop("AFSK_BENCH_LAUNCHER", None)
    p = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for ln in iter(p.stderr.readline, ""):                             # (heartbeats are silent at rank > 0)
        if "READY" in ln:
            break
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=30)
    assert p.returncode == bench_mod().EXIT_RANK_FAILED
    doc = _one_line(out)
    assert doc["printed_by"] == "rank 1" and "SIGTERM" in doc["error"]
    assert doc["heartbeats"]["0"] == {"phase": "shard built", "age_s": doc["heartbeats"]["0"]["age_s"],
                                      "since_start_s": 1.0, "alive": False}
    assert doc["phase"] == "shard built"                               # rank 0's last phase, from its heartbeat


def test_failure_line_stays_under_the_cap():
    import bench
    hbs = {str(r): {"phase": "x" * 120, "age_s": 1.0, "since_start_s": 2.0, "alive": True} for r in range(8)}
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r4_bench_full_n1.json")))
    partial = bench.compact_line(full, "gpurun_out/bench_full_n1.json")
    line = bench.failure_line(8, 20, 5, "e" * 2000, "p" * 500, hbs, partial)
    assert len(json.dumps(line)) < bench.LINE_CAP and line["value"] == full["value"] and line["incomplete"] is True
    line = bench.failure_line(8, 20, 5, "boom", "pg up", hbs, None)
    assert len(json.dumps(line)) < bench.LINE_CAP and line["value"] is None
    assert set(CONTRACT_KEYS) <= set(line)


def test_omp_num_threads_is_parsed_like_openmp_does():
    """A legal list-form or empty OMP_NUM_THREADS must not kill the launcher before any rank has started (r5 advisor)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_launch
    assert bench_launch.omp_threads("8,2", 4) == 8 and bench_launch.omp_threads(" 6 , 2", 4) == 6
    assert bench_launch.omp_threads("", 4) == 4 and bench_launch.omp_threads(None, 4) == 4
    assert bench_launch.omp_threads("auto", 4) == 4 and bench_launch.omp_threads("0", 4) == 4 and bench_launch.omp_threads(3, 4) == 3
    # and the launcher as a whole: stub ranks, a list-form value in the environment -> the line is relayed, exit 0
    old = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "8,2"
    try:
        rc, out = _launch("ok")
    finally:
        if old is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = old
    assert rc == 0 and _one_line(out)["ranks_seen"] == 2


def test_only_node_zero_prints_under_a_multi_node_external_launcher(tmp_path):
    """bench.py measures one node; if someone launches it across nodes the heartbeat directory is per node, and the
    lowest rank of every node would win its own claim file: ranks with GROUP_RANK != 0 never print (r5 advisor)."""
    code = (
        "import os, sys, time, types\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "hb = bench.Heartbeat(int(os.environ['RANK']), 4)\n"
        "dog = bench.Watchdog(hb, types.SimpleNamespace(steps=5, warmup=1), 2.0)\n"
        "time.sleep(600)\n")
    env = dict(os.environ, WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998", AFSK_BENCH_HB_DIR=str(tmp_path),
               RANK="2", GROUP_RANK="1")
    env.pop("AFSK_BENCH_LAUNCHER", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode == bench_mod().EXIT_DEADLINE and p.stdout.strip() == ""


def test_thread_budget_is_divided_by_the_local_world(monkeypatch):
    import bench
    whole = bench.usable_cpus(local_world=1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert bench.usable_cpus() == max(1, whole // 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert bench.usable_cpus() == whole


def test_launcher_counts_gpus_without_torch():
    """The launcher parent must not initialise HIP: devices are counted in sysfs (None / 0 without a KFD node)."""
    import bench
    n = bench.visible_gpus()
    assert n is None or n >= 0
    src = open(os.path.join(ROOT, "tools", "bench_launch.py")).read()     # launcher, heartbeats, watchdog: no torch at all
    assert "def self_launch" in src and "import torch" not in src and "torch.cuda" not in src


def test_rank_side_without_a_gpu_prints_one_diagnostic_line():
    """The product has no CPU fallback: the real rank code on a box without a HIP device ends in ONE line that says
    so (value null, the phase it was in), exit code non-zero -- in-process at N = 1 and under torch.distributed.run."""
    import torch
    if torch.cuda.device_count() >= 1:
        pytest.skip("a GPU is visible here")
    bench_py = os.path.join(ROOT, "bench.py")
    p = subprocess.run([sys.executable, bench_py, "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    doc = _one_line(p.stdout)
    assert p.returncode == bench_mod().EXIT_RANK_FAILED
    assert doc["value"] is None and "no HIP device" in doc["error"] and doc["phase"] == "checking for a HIP device"
    assert doc["steps"] == 2 and doc["n_gpus"] == 1 and set(CONTRACT_KEYS) <= set(doc)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(bench_mod().free_port()),
                        bench_py, "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    doc = _one_line(p.stdout)                                          # rank 0 prints, rank 1 only exits
    assert p.returncode != 0 and doc["printed_by"] == "rank 0" and doc["n_gpus"] == 2
    assert "no HIP device" in doc["error"] and set(doc["heartbeats"]) == {"0", "1"}


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
    assert one["subs"] == ["config2", "config3", "config4", "config5_lead"]
    # r6: the lead-in workload -- leads are a pure function of the GLOBAL stream index (a rank's shard of a larger job
    # gets the leads of its own streams), 0 ... 2047, and 7 of 8 clock indices are not multiples of 8 samples
    la = bench.Shard.host_leads("config5_lead", 0, 65536)
    lb = bench.Shard.host_leads("config5_lead", 32768, 4096)
    assert la.min() == 0 and la.max() == 2047 and (la[32768:32768 + 4096] == lb).all()
    assert 0.85 < float(((2 * la) & 15).astype(bool).mean()) < 0.90
    assert bench.Shard.host_leads("config5", 0, 16) is None
    assert one["next"] == ["f1_modulate", "f2_gate", "f2_chain", "f3_wav_ingest", "rates_65536"] == list(bench.DEFAULT_RIDERS)
    # the egress mirror and the 4096-stream table are measured on request only
    assert bench.plan(1, "", "f5_wav_egress,rates_4096,ragged_lengths")["next"] == ["f5_wav_egress", "rates_4096", "ragged_lengths"]
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


@pytest.mark.parametrize("record", ["archive/r3_bench_n1.json", "archive/r3_bench_n8_diagnostic_gloo_shared_gpu.json",
                                    "archive/r2_bench_n1.json"])
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
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r3_bench_n1.json")))
    full["sub_records"].update({f"future_{i}": {"value": i, "roofline": {"frac": 0.5}, "match_rate": 1.0,
                                                "entry": "e" * 40} for i in range(200)})
    line = bench.compact_line(full, None)
    assert len(json.dumps(line)) < bench.LINE_CAP
    assert set(CONTRACT_KEYS) <= set(line) and line["sub_records"] == "see full_record"
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and "cpu_baseline" in line


def test_full_record_goes_to_a_file_not_to_the_streams(tmp_path, monkeypatch, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r3_bench_n1.json")))
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
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r3_bench_n1.json")))
    full["cpu_baseline"]["calibration"] = cal
    assert bench.compact_line(full, None)["cpu_baseline"]["calibration"] == cal


def test_rider_rows_live_outside_bench_py_and_import_on_cpu():
    """bench.py (launcher, headline, configs 2-4, cpu_baseline) stays small; the f1 / f2 / f3 / f5 rows and the per-rate
    tables live in tools/bench_rows.py, are imported lazily and run inside try / except (a failing rider becomes
    {"error": ...}: tests/test_gpu_multi.py checks that on the GPU)."""
    import importlib
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    rows = importlib.import_module("bench_rows")
    for fn in ("measure_modulate", "measure_gate", "measure_wav_ingest", "measure_wav_egress", "measure_rates"):
        assert callable(getattr(rows, fn)) and not hasattr(bench, fn), fn
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import bench_rows" in src and src.count("def rider(") == 1
    assert len(src.splitlines()) < 1400
    # a failed rider's summary is its error, nothing else -- and the line stays parseable
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r4_bench_full_n1.json")))
    full["sub_records"]["f3_wav_ingest"] = {"error": "OSError: [Errno 2] No such file or directory: '/proc/afsk_no_such_dir'"}
    line = bench.compact_line(full, None)
    assert line["sub_records"]["f3_wav_ingest"] == {"error": full["sub_records"]["f3_wav_ingest"]["error"][:120]}
    assert line["value"] == full["value"] and len(json.dumps(line)) < bench.LINE_CAP
