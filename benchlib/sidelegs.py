"""The deadlines of bench.py.  Whoever runs the bench gives it a time of its own (the driver's lease: 1800 s) and kills it from
outside when that passes - silently, with whatever was already measured.  So every wait in here ends by itself, earlier:

  * the whole job ends TH_BENCH_TIMEOUT seconds (default 1500) after the process started - an ABSOLUTE point in time: the
    headline's stages (`Stages`), every side leg and the communicator's start (`comm_deadline`) are cut to what is left of it;
  * a side leg (frame loop, the strong-scaling legs, the sharded frame loop) gets TH_BENCH_LEG_TIMEOUT seconds (default 300; a
    leg takes 5-40 s).  A leg that raises is reported in its place.  A leg that never comes back - a collective one rank does
    not reach, a peer that died inside it - would keep every rank, and with them the headline that is already measured,
    waiting: when its deadline passes rank 0 prints the line as it stands (the leg marked as abandoned and named in
    `legs_failed`, the legs behind it as not run) and every rank ends there;
  * the library's communicator gets TH_BENCH_COMM_TIMEOUT seconds (default 180) to come up on every rank (`comm_deadline`):
    a rank standing inside ncclCommInitRank cannot be called back, so each rank starts the bench AGAIN as a child process
    with --no-library-comm (the counter block reduced through torch.distributed; a new rendezvous port) and ends with the
    child's status - a child, never an exec: this process has touched the GPU."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

T_START = time.time()


def job_seconds():
    return float(os.environ.get("TH_BENCH_TIMEOUT", "1500"))


def remaining():
    """seconds left of the whole job's time"""
    return T_START + job_seconds() - time.time()


class SideLegs:
    def __init__(self, line, rank):
        self.line, self.rank = line, rank
        self.limit = float(os.environ.get("TH_BENCH_LEG_TIMEOUT", "300"))
        self.seconds = self.limit
        self.lock = threading.Lock()
        self.running = None
        line.setdefault("legs_failed", [])      # (read this, not the exit status: a leg that hangs must not cost the headline its rc 0)

    def run(self, name, fn, pending=(), record=True):
        """line[name] = fn(), or what went wrong; `pending`: the legs that would follow (named in the line if this one hangs);
        record=False: a step that has no entry of its own unless it goes wrong (the job's last barrier)"""
        with self.lock:
            self.running = name
            self.seconds = max(min(self.limit, remaining() - 15.0), 1.0)      # (the job's own end comes first)
        if self.rank == 0:
            print("[bench] leg: %s" % name, file=sys.stderr, flush=True)
        # (the other ranks leave a little later than rank 0: its line first)
        timer = threading.Timer(self.seconds + (0.0 if self.rank == 0 else 10.0), self._abandon, (name, tuple(pending)))
        timer.daemon = True
        timer.start()
        try:
            result = fn()
        except Exception as e:            # noqa: BLE001
            result = {"error": "%s: %s" % (type(e).__name__, e)}
        with self.lock:                   # (held by a deadline that has passed: this thread stops here and the process ends)
            self.running = None
            timer.cancel()
            if record or result is not None:
                self.line[name] = result
            if isinstance(result, dict) and "error" in result:
                self.line["legs_failed"].append(name)

    def _abandon(self, name, pending):
        with self.lock:
            if self.running != name:
                return
            if self.rank == 0:
                self.line[name] = {"error": "no result within %.0f s: the leg was abandoned and the job ended here" % self.seconds}
                self.line["legs_failed"].append(name)
                for p in pending:
                    self.line.setdefault(p, {"skipped": "the leg `%s` before it did not come back" % name})
                try:
                    C.CDLL(None).fflush(None)
                except OSError:
                    pass
                print(json.dumps(self.line), flush=True)
            os._exit(0)


class Stages:
    """Where the job is, on stderr as it goes (rank 0; one short line per stage with the seconds since the start) - a run that is
    ended from outside leaves its last stage in the log -, and the deadline of the whole job (TH_BENCH_TIMEOUT seconds from the
    start of the process, default 1500: under the 1800 s the driver gives a run): a headline that never comes - a collective
    inside the timed region that some rank does not reach - ends with a line that says where it stood instead of silence."""

    def __init__(self, rank, world, describe):
        self.rank, self.world, self.describe, self.t0, self.now = rank, world, describe, T_START, "start"
        self._time, self._err = time, sys.stderr
        seconds = job_seconds()
        self.timer = threading.Timer(max(remaining(), 1.0) + (0.0 if rank == 0 else 10.0), self._expired, (seconds,))
        self.timer.daemon = True
        self.timer.start()

    def at(self, name):
        self.now = name
        if self.rank == 0:
            print("[bench] %7.1f s  %s" % (self._time.time() - self.t0, name), file=self._err, flush=True)

    def done(self):
        self.timer.cancel()

    def _expired(self, seconds):
        if self.rank == 0:
            line = dict(self.describe)
            line.update(value=None, error="no result within %.0f s; the job stood at: %s" % (seconds, self.now))
            try:
                C.CDLL(None).fflush(None)
            except OSError:
                pass
            print(json.dumps(line), flush=True)
        os._exit(3)


class comm_deadline:
    """`with comm_deadline(rank, argv): <communicator start + the ranks' agreement on it>`.  Should the body not end within
    TH_BENCH_COMM_TIMEOUT seconds (a rank that never arrives leaves the others inside ncclCommInitRank, and them nobody can call
    back), THIS process - every rank's, each by its own clock: the agreement inside the body ends on all of them or on none -
    starts `bench.py <argv> --no-library-comm` as a child with a rendezvous of its own, waits, and ends with the child's
    status.  The child's line says why (`rccl.fallback`)."""

    def __init__(self, rank, argv, bench):
        self.rank, self.argv, self.bench = rank, list(argv), bench
        self.seconds = float(os.environ.get("TH_BENCH_COMM_TIMEOUT", "180"))
        self.timer, self.lock, self.over = None, threading.Lock(), False

    def __enter__(self):
        self.timer = threading.Timer(max(min(self.seconds, remaining() - 60.0), 1.0), self._fresh_child)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        with self.lock:                   # (a deadline that has just passed holds it: this thread stops here and the process ends)
            self.over = True
            self.timer.cancel()
        return False

    def _fresh_child(self):
        self.lock.acquire()               # (never released: the body must not go on beside the child)
        if self.over:
            self.lock.release()
            return
        why = "the library's communicator had not come up on every rank within %.0f s" % self.seconds
        if self.rank == 0:
            print("[bench] %s: the bench starts again as a child process with --no-library-comm" % why, file=sys.stderr, flush=True)
        env = dict(os.environ)
        port = int(env.get("MASTER_PORT", "29511"))
        env["MASTER_PORT"] = str(20000 + (port * 7 + 13) % 20000)      # (the same on every rank, without a word exchanged)
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)                   # rank 0's child hosts the store of the new rendezvous
        env["TH_BENCH_TIMEOUT"] = "%.0f" % max(remaining() - 20.0, 30.0)
        env["TH_BENCH_COMM_FALLBACK"] = why
        argv = [a for a in self.argv if a != "--no-library-comm"] + ["--no-library-comm"]
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        try:
            rc = subprocess.call([sys.executable, self.bench] + argv, env=env)
        except OSError:
            rc = 5
        os._exit(rc)
