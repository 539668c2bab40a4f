#!/usr/bin/env python3
"""bench_launch.py -- how a bench.py run starts, watches itself and fails: GPU count without HIP, the launcher of a
bare `python bench.py --gpus N` (own rank processes, deadline, exit codes), per-rank heartbeats, the watchdog thread
of every rank (deadline / SIGTERM under an external torchrun), and the ONE diagnostic JSON line of a run that did not
finish.  No torch, no HIP, nothing of the product: importable anywhere (tests/test_bench_launch.py drives it with stub
ranks).  bench.py re-exports every name; METRIC / the headline description / the line cap come from there
(`configure`)."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time

METRIC = ""                 # set by bench.py (configure): the contract's metric string
HEADLINE_DESC = ""          # description of the headline workload (config.workload of a line without numbers)
LINE_CAP = 4096


def configure(metric: str, headline_desc: str, line_cap: int) -> None:
    global METRIC, HEADLINE_DESC, LINE_CAP
    METRIC, HEADLINE_DESC, LINE_CAP = metric, headline_desc, line_cap


BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def usable_cpus(local_world: int | None = None) -> int:
    """CPUs THIS PROCESS may really use: the cgroup CPU quota where one is set (the GPU boxes show all 256 cores
    of the host to a container that gets the time of 16), else the affinity mask / core count -- divided by the
    number of rank processes on this node (LOCAL_WORLD_SIZE: one process per GPU, and eight ranks that each
    start a quota's worth of threads are throttled for whole 100 ms periods).  The CPU baseline runs on -- and
    reports as `cores` -- this many threads.  local_world=1 gives the whole box (the launcher's view)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    if local_world is None:
        try:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        except ValueError:
            local_world = 1
    return max(1, n // max(1, local_world))


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus() -> int | None:
    """GPUs this process could use, counted WITHOUT touching HIP or importing torch: the KFD topology in sysfs
    (a node with simd_count > 0 is a GPU), narrowed by a HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list.
    None when the topology cannot be read (the launcher then lets the ranks fail loudly themselves)."""
    import glob
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                   # no KFD driver: no AMD GPU in this machine / container
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not paths:
        return None
    n = 0
    for p in paths:
        try:
            for ln in open(p):
                k, _, v = ln.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = os.environ.get(var)
        if lst is not None:
            n = min(n, len([x for x in lst.split(",") if x.strip()]))
    return n


# --------------------------------------------------------------------------- heartbeats, deadline, diagnostic line
#
# The first real multi-GPU run of this code is the driver's scaling run: the only acceptable outcomes are a line
# with numbers or a line that says exactly what failed.  Three pieces:
#   * every rank writes a heartbeat per phase (stderr + one small file per rank in a shared directory);
#   * bare `python bench.py --gpus N`: the launcher (never touches the GPU, no torch import) watches its rank
#     children against --deadline-s and their exit codes, terminates them and prints ONE JSON line
#     (metric, n_gpus, error, phase, per-rank last heartbeat; rank 0's partial result when the headline was
#     already measured) and exits non-zero;
#   * under an external torchrun (the driver's own launch) and at N = 1 a watchdog THREAD in every rank does the
#     same from inside: deadline or SIGTERM (torchrun terminating the survivors of a failed rank) -> the line is
#     printed by rank 0, or by the lowest surviving rank, then os._exit.  It works while the main thread hangs in
#     a collective: the C-level signal handler only writes to a wake-up pipe that the thread reads.
HB_MAX_AGE_S = 1800.0        # heartbeat files older than this belong to an earlier run on the same port
EXIT_RANK_FAILED, EXIT_DEADLINE, EXIT_NO_LINE = 3, 4, 1


def omp_threads(value, default: int) -> int:
    """OMP_NUM_THREADS as OpenMP reads it -- a comma list gives one count per nesting level ("8,2"): the first -- or
    `default` when it is unset, empty or not a positive number.  (The launcher must not die of a legal value before any
    rank has started: there would be no diagnostic line.)"""
    try:
        n = int(str(value).split(",")[0].strip())
        return n if n > 0 else default
    except (TypeError, ValueError):
        return default


def hb_dir_for_env(env=None) -> str:
    """The heartbeat directory of this job: AFSK_BENCH_HB_DIR (set by the launcher), else one derived from the
    rendezvous port, so that the ranks of an external torchrun agree on it without talking to each other."""
    import tempfile
    env = os.environ if env is None else env
    d = env.get("AFSK_BENCH_HB_DIR")
    if not d:
        # the ranks of one torchrun share their parent (the agent); consecutive jobs on one port do not
        job = os.getppid() if "RANK" in env else os.getpid()
        d = os.path.join(tempfile.gettempdir(), "afsk_bench_hb_%s_%s_%s" % (env.get("MASTER_PORT", "solo"), os.getuid(), job))
    os.makedirs(d, exist_ok=True)
    return d


class Heartbeat:
    """Per-rank phase marks: a line on stderr and rank<r>.json in the job's heartbeat directory (atomic rename)."""

    def __init__(self, rank: int, world: int, hb_dir: str | None = None):
        self.rank, self.world = rank, world
        self.dir = hb_dir or hb_dir_for_env()
        self.t_start = time.time()
        self.phase = "start"
        # stderr stays small (the driver keeps one ~8 KB tail of stdout + stderr, and the result line must survive in
        # it): rank 0 of an N > 1 run narrates its phases, every other rank -- and an N = 1 run -- only writes the file
        self.quiet = not ((rank == 0 and world > 1) or os.environ.get("AFSK_BENCH_VERBOSE") == "1")
        self.beat("process up")

    def beat(self, phase: str) -> None:
        self.phase = phase
        now = time.time()
        doc = {"rank": self.rank, "pid": os.getpid(), "phase": phase, "t": now, "t_start": self.t_start,
               "since_start_s": round(now - self.t_start, 2)}
        try:
            tmp = os.path.join(self.dir, f".rank{self.rank}.{os.getpid()}.tmp")
            with open(tmp, "w") as f:
                json.dump(doc, f)
            os.replace(tmp, os.path.join(self.dir, f"rank{self.rank}.json"))
        except OSError:
            pass
        if not self.quiet:
            sys.stderr.write(f"bench.py[rank {self.rank}/{self.world}] +{now - self.t_start:7.2f}s  {phase}\n")
            sys.stderr.flush()
        # fault injection for the rehearsals of the failure paths (tests only): AFSK_BENCH_FAULT=<rank>:<die|hang>:<phase prefix>
        fault = os.environ.get("AFSK_BENCH_FAULT")
        if fault:
            r_, _, rest = fault.partition(":")
            kind, _, prefix = rest.partition(":")
            if r_ == str(self.rank) and prefix and phase.startswith(prefix):
                if kind == "die":
                    os._exit(7)
                if kind == "hang":
                    time.sleep(3600)

    def write_partial(self, line: dict) -> None:
        """Rank 0: the result line as it stands (headline measured, riders possibly missing) -- what the
        launcher or a watchdog prints, with the error attached, if the run dies later."""
        try:
            tmp = os.path.join(self.dir, f".partial.{os.getpid()}.tmp")
            with open(tmp, "w") as f:
                json.dump(line, f)
            os.replace(tmp, os.path.join(self.dir, "partial.json"))
        except OSError:
            pass


def read_heartbeats(hb_dir: str, world: int) -> dict:
    """{rank: {phase, age_s, since_start_s, alive}} of every rank that has written one (recent files only)."""
    out = {}
    now = time.time()
    for r in range(world):
        try:
            doc = json.load(open(os.path.join(hb_dir, f"rank{r}.json")))
        except (OSError, ValueError):
            out[str(r)] = {"phase": "no heartbeat", "age_s": None}
            continue
        if now - doc.get("t", 0) > HB_MAX_AGE_S:
            out[str(r)] = {"phase": "no heartbeat (stale file)", "age_s": None}
            continue
        alive = None
        try:
            os.kill(int(doc["pid"]), 0)
            alive = True
            try:                                      # an exited child its parent has not reaped yet still "exists"
                st = open(f"/proc/{int(doc['pid'])}/stat").read()
                if st[st.rindex(")") + 2] in "ZX":
                    alive = False
            except (OSError, ValueError, IndexError):
                pass
        except ProcessLookupError:
            alive = False
        except (OSError, ValueError, KeyError):
            pass
        out[str(r)] = {"phase": str(doc.get("phase"))[:120], "age_s": round(now - doc.get("t", now), 1),
                       "since_start_s": doc.get("since_start_s"), "alive": alive}
    return out


def read_partial(hb_dir: str) -> dict | None:
    try:
        p = os.path.join(hb_dir, "partial.json")
        if time.time() - os.path.getmtime(p) > HB_MAX_AGE_S:
            return None
        return json.load(open(p))
    except (OSError, ValueError):
        return None


def failure_line(n_gpus: int, steps, warmup, error: str, phase: str, heartbeats: dict, partial: dict | None = None,
                 printed_by: str = "launcher") -> dict:
    """The ONE line of a run that did not finish: the contract's keys (value null unless the headline had
    been measured: then rank 0's partial line, complete as far as it got) + error, phase, per-rank heartbeats."""
    if partial and partial.get("value") is not None:
        line = dict(partial)
        line["incomplete"] = True
    else:
        line = {"metric": METRIC, "value": None, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
                "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16",
                "data": "synthetic", "config": {"workload": HEADLINE_DESC}, "roofline": None}
    line["error"] = error[:600]
    line["phase"] = phase[:160]
    line["heartbeats"] = heartbeats
    line["printed_by"] = printed_by
    text = json.dumps(line)
    if len(text) >= LINE_CAP:                       # never over the cap: drop the bulky summaries first
        for victim in ("sub_records", "per_workload_value", "cpu_baseline"):
            if victim in line and len(json.dumps(line)) >= LINE_CAP:
                line[victim] = "see full_record"
    return line


def _kill_group(p, sig) -> None:
    try:
        os.killpg(p.pid, sig)
    except (ProcessLookupError, PermissionError, OSError):
        try:
            p.send_signal(sig)
        except (ProcessLookupError, OSError):
            pass


def self_launch(n: int, argv: list[str], script: str | None = None, deadline_s: float = 420.0,
                grace_s: float = 5.0, steps=None, warmup=None) -> int:
    """Parent of an N > 1 run: never touches the GPU (and never imports torch), starts N fresh rank processes
    itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, each in its own session), relays
    rank 0's JSON line.  If a rank exits non-zero or the deadline passes it terminates every rank (SIGTERM, then
    SIGKILL after grace_s), prints ONE diagnostic JSON line (failure_line) and returns non-zero.
    (`script` is this file; tests/test_bench_launch.py passes stubs to exercise relay, failure and deadline.)"""
    import signal
    import tempfile
    import threading
    hb_dir = tempfile.mkdtemp(prefix="afsk_bench_hb_")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool
    # host threads: the box's usable CPUs are shared by the N ranks (OpenMP, the I/O pool of libafsk_amd.so and
    # the CPU-oracle legs all size themselves from these)
    share = max(1, usable_cpus(local_world=1) // n)
    env["OMP_NUM_THREADS"] = str(min(omp_threads(env.get("OMP_NUM_THREADS"), share), share))
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(free_port()), "AFSK_BENCH_HB_DIR": hb_dir, "AFSK_BENCH_LAUNCHER": "1",
                "AFSK_BENCH_DEADLINE_S": str(deadline_s)})
    cmd = [sys.executable, "-u", script or BENCH_PY] + argv
    procs, readers = [], []
    lines: list = []

    def pump(r: int, stream) -> None:
        for ln in stream:
            if r == 0 and ln.startswith('{"metric"'):
                lines.append(ln.strip())
            else:
                sys.stderr.write(ln)

    t0 = time.monotonic()
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, text=True, start_new_session=True)
        th = threading.Thread(target=pump, args=(r, p.stdout), daemon=True)
        th.start()
        procs.append(p)
        readers.append(th)
    failure = None            # (exit code of the launcher, error text)
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failure = (EXIT_RANK_FAILED, "rank(s) exited non-zero: " + ", ".join(
                f"rank {r} -> " + (f"signal {-c}" if c < 0 else f"exit code {c}") for r, c in bad))
            break
        if all(c == 0 for c in codes):
            break
        if deadline_s > 0 and time.monotonic() - t0 > deadline_s:
            failure = (EXIT_DEADLINE, f"deadline of {deadline_s:g} s passed with rank(s) "
                       + ", ".join(str(r) for r, c in enumerate(codes) if c is None) + " still running")
            break
        time.sleep(0.05)
    hbs = read_heartbeats(hb_dir, n) if failure else None
    if failure:
        for p in procs:
            if p.poll() is None:
                _kill_group(p, signal.SIGTERM)
        t_end = time.monotonic() + grace_s
        while time.monotonic() < t_end and any(p.poll() is None for p in procs):
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                _kill_group(p, signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass
    for th in readers:
        th.join(timeout=2.0)
    rc = 0
    if failure:
        rc, err = failure
        partial = json.loads(lines[-1]) if lines else read_partial(hb_dir)
        print(json.dumps(failure_line(n, steps, warmup, err, (hbs.get("0") or {}).get("phase", "?"), hbs, partial)),
              flush=True)
    elif lines:
        print(lines[-1], flush=True)
    else:
        rc = EXIT_NO_LINE
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        print(json.dumps(failure_line(n, steps, warmup, "every rank exited with code 0 but rank 0 printed no result line",
                                      "?", read_heartbeats(hb_dir, n), read_partial(hb_dir))), flush=True)
    import shutil
    shutil.rmtree(hb_dir, ignore_errors=True)
    return rc


class Watchdog:
    """In-rank guard (external torchrun, N = 1): deadline and SIGTERM handling on a thread of its own.
    The main thread may hang inside a HIP or RCCL call for ever: Python-level signal handlers would never run,
    but signal.set_wakeup_fd() makes the C-level handler write the signal number into a pipe this thread reads.
    On SIGTERM / deadline: the printer (rank 0; or, if rank 0's process is gone, the lowest rank that wins the
    claim file) prints failure_line -- rank 0's partial result when the headline had been measured -- and every
    rank ends with os._exit.  Under bench.py's own launcher the launcher prints, the ranks only exit."""

    def __init__(self, hb: Heartbeat, args, deadline_s: float):
        import signal
        import threading
        self.hb, self.args = hb, args
        self.deadline = time.monotonic() + deadline_s if deadline_s > 0 else None
        self.deadline_s = deadline_s
        self.done = threading.Event()          # the result line is out: stand down
        self.under_launcher = os.environ.get("AFSK_BENCH_LAUNCHER") == "1"
        self.main_thread_id = threading.main_thread().ident
        self.rfd, self.wfd = os.pipe()
        os.set_blocking(self.wfd, False)
        os.set_blocking(self.rfd, False)
        signal.set_wakeup_fd(self.wfd, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: None)     # keep the process alive: the thread decides
        self.thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        self.thread.start()

    def stand_down(self) -> None:
        self.done.set()

    def _main_stack(self) -> str:
        import traceback
        fr = sys._current_frames().get(self.main_thread_id)
        if fr is None:
            return "?"
        st = traceback.extract_stack(fr)[-4:]
        return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(st))

    def _run(self) -> None:
        import select
        import signal
        while not self.done.is_set():
            try:
                ready, _, _ = select.select([self.rfd], [], [], 0.25)
            except OSError:
                return
            if self.done.is_set():
                return
            if ready:
                try:
                    data = os.read(self.rfd, 64)
                except OSError:
                    data = b""
                if int(signal.SIGTERM) in data:
                    self._fire(EXIT_RANK_FAILED, "SIGTERM received (another rank failed or the launcher gave up)")
                    return
            if self.deadline is not None and time.monotonic() > self.deadline:
                self._fire(EXIT_DEADLINE, f"deadline of {self.deadline_s:g} s passed")
                return

    def _fire(self, rc: int, why: str) -> None:
        hb = self.hb
        where = f"{hb.phase} [main thread at {self._main_stack()}]"
        hb.quiet = True
        last_phase = hb.phase
        hb.beat(f"aborted in '{last_phase}': {why}")
        sys.stderr.write(f"bench.py[rank {hb.rank}] {why}; was in: {where}\n")
        sys.stderr.flush()
        if not self.under_launcher:
            self._maybe_print(rc, why, where)
        try:
            sys.stdout.flush()
        except Exception:  # noqa: BLE001
            pass
        os._exit(rc)

    def _maybe_print(self, rc: int, why: str, where: str) -> None:
        hb = self.hb
        # bench.py measures ONE node (the contract's `--gpus N`): under an external multi-node torchrun the heartbeat
        # directory is per node, so only node 0's ranks may print -- otherwise every node's lowest rank would win its own
        # claim file and the job would print one failure line per node
        if os.environ.get("GROUP_RANK", "0") not in ("", "0"):
            return
        claim = os.path.join(hb.dir, "line_printed.claim")
        if hb.rank != 0:
            # rank 0 prints if it can; the others step in, lowest rank first, only if nobody has
            time.sleep(min(6.0, 1.5 * hb.rank))
        try:
            if os.path.exists(claim) and time.time() - os.path.getmtime(claim) > HB_MAX_AGE_S:
                os.unlink(claim)                                   # left over from an earlier job on this port
            fd = os.open(claim, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
            os.write(fd, str(hb.rank).encode())
            os.close(fd)
        except OSError:
            return                                                 # somebody else has printed the line
        hbs = read_heartbeats(hb.dir, hb.world)
        phase0 = (hbs.get("0") or {}).get("phase", "?")
        line = failure_line(hb.world, self.args.steps, self.args.warmup, f"rank {hb.rank}: {why}",
                            where if hb.rank == 0 else phase0, hbs, read_partial(hb.dir), f"rank {hb.rank}")
        print(json.dumps(line), flush=True)
