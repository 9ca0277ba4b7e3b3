"""bench.py as its own launcher, and what keeps an N-rank run's stdout to ONE JSON line: `python bench.py --gpus N` without a launcher
starts N rank processes (self_launch); native libraries' banners are kept off rank 0's stdout (stdout_to_stderr); the optional legs
behind the headline run under a budget that can never cost the headline (HeadlineGuard)."""
import json
import os
import sys
import time

from .common import BENCH_PY

REAL_STDOUT_FD = None   # the process's real stdout while fd 1 is routed to stderr (HeadlineGuard prints there)

class stdout_to_stderr:
    """Route fd 1 to fd 2 while native libraries initialise (RCCL prints a version banner on stdout): rank 0's stdout
    must carry exactly one JSON line."""

    def __enter__(self):
        global REAL_STDOUT_FD
        sys.stdout.flush()
        self._saved = os.dup(1)
        REAL_STDOUT_FD = self._saved
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:    # RCCL's banner is printf'ed: with stdout a pipe it sits in C's buffer and would come out at exit, behind the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        global REAL_STDOUT_FD
        os.dup2(self._saved, 1)
        os.close(self._saved)
        REAL_STDOUT_FD = None

class HeadlineGuard:
    """The optional legs that follow the headline at N > 1 (the peer-to-peer halo check, BASELINE config 5) are collective: a rank
    that fails alone leaves the others inside a barrier.  They must never cost the headline.  Once the headline line is complete it
    is armed with a budget; if the legs are not done by then, rank 0 prints the headline as it stands (plus a note saying what was
    cut short) on the real stdout and every rank leaves at once -- exit code 0, one JSON line, as the contract wants."""

    def __init__(self):
        self._timer = None

    def arm(self, line, seconds, what):
        import threading
        self.disarm()

        def fire():
            try:
                if line is not None:
                    line.setdefault("notes", []).append("%s did not finish within %d s and was cut short; the headline above is complete" % (what, seconds))
                    data = (json.dumps(line) + "\n").encode()
                    fd = REAL_STDOUT_FD if REAL_STDOUT_FD is not None else 1
                    while data:
                        data = data[os.write(fd, data):]
            finally:
                os._exit(0)

        self._timer = threading.Timer(seconds, fire)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

GUARD = HeadlineGuard()

def visible_devices():
    """HIP devices this process can see (torch is the plumbing the ranks use anyway; no context is created by the count)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0

def free_port():
    """A rendezvous port that is free now AND stays free until the rank processes bind it: drawn below the kernel's ephemeral range
    (32768+), from which any outgoing connection of this box may take a port between our probe and torch's bind (seen once:
    EADDRINUSE on a port bind(0) had just handed out)."""
    import random
    import socket
    rng = random.Random(os.getpid() ^ int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(20000, 30000)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", port))
                return port
            except OSError:
                continue
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

def self_launch(n, argv, worker=None, devices=None, limit_s=None, out=None, err=None):
    """`python bench.py --gpus N` without a launcher: be the launcher.  Starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, one device each through LOCAL_RANK, a free rendezvous port on 127.0.0.1), forwards rank 0's
    stdout -- the ONE JSON line -- to this process's stdout and every other rank's stdout to stderr, and returns the exit code:
    0 if every rank exited 0, otherwise the first non-zero code seen (the surviving ranks are terminated by PID, never by pattern).
    A rank that dies takes the launch down at once instead of leaving its peers in a collective until the watchdog fires.

    worker / devices / limit_s / out / err are for the CPU test of this logic (a stub worker, a pretended device count)."""
    import subprocess
    import threading
    out = out or sys.stdout
    err = err or sys.stderr
    have = visible_devices() if devices is None else devices
    if have < n:
        err.write("bench.py: %d devices requested, %d visible\n" % (n, have))
        return 2
    worker = worker or [sys.executable, BENCH_PY]
    limit_s = limit_s if limit_s is not None else float(os.environ.get("TETSIM_BENCH_WATCHDOG_S", "600")) + 30.0
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs, pumps = [], []

    def pump(src, dst, lock=threading.Lock()):
        for line in iter(src.readline, ""):
            with lock:
                dst.write(line)
                dst.flush()

    for r in range(n):
        p = subprocess.Popen(worker + list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=None,
                             text=True, bufsize=1)
        procs.append(p)
        th = threading.Thread(target=pump, args=(p.stdout, out if r == 0 else err), daemon=True)
        th.start()
        pumps.append(th)
    deadline = time.monotonic() + limit_s
    code, live = 0, set(range(n))
    while live and code == 0:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    err.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, rc))
        if time.monotonic() > deadline and live:
            err.write("bench.py: ranks %s still running after %.0f s; stopping them\n" % (sorted(live), limit_s))
            code = 124
        if live and code == 0:
            time.sleep(0.05)
    for r in live:                      # only on failure: the ranks that are still up
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    for th in pumps:
        th.join(timeout=5)
    return code
