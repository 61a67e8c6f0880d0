"""The legs beside the headline (frame loop, configs 4 and 5, the sharded frame loop) must not cost the line.

A leg that raises is reported in its place.  A leg that never comes back - a collective one rank does not reach, a peer that
died inside it - would keep every rank, and with them the headline that is already measured, waiting until the launcher's
caller gives up: each leg therefore runs under a deadline, and when it passes rank 0 prints the line as it stands (the leg
marked as abandoned, the legs behind it as not run) and every rank ends there.  The deadlines are far above what the legs
take (TH_BENCH_LEG_TIMEOUT seconds, default 420; a leg takes 5-40 s)."""
import ctypes as C
import json
import os
import threading


class SideLegs:
    def __init__(self, line, rank):
        self.line, self.rank = line, rank
        self.seconds = float(os.environ.get("TH_BENCH_LEG_TIMEOUT", "420"))
        self.lock = threading.Lock()
        self.running = None

    def run(self, name, fn, pending=(), record=True):
        """line[name] = fn(), or what went wrong; `pending`: the legs that would follow (named in the line if this one hangs);
        record=False: a step that has no entry of its own unless it goes wrong (the job's last barrier)"""
        with self.lock:
            self.running = name
        if self.rank == 0:
            import sys
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

    def _abandon(self, name, pending):
        with self.lock:
            if self.running != name:
                return
            if self.rank == 0:
                self.line[name] = {"error": "no result within %.0f s: the leg was abandoned and the job ended here" % self.seconds}
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
    ended from outside leaves its last stage in the log -, and a deadline for the whole job (TH_BENCH_TIMEOUT seconds, default
    2400): a headline that never comes - a communicator that does not come up, a collective inside the timed region that some
    rank does not reach - ends with a line that says where it stood instead of silence."""

    def __init__(self, rank, world, describe):
        import sys
        import time
        self.rank, self.world, self.describe, self.t0, self.now = rank, world, describe, time.time(), "start"
        self._time, self._err = time, sys.stderr
        seconds = float(os.environ.get("TH_BENCH_TIMEOUT", "2400"))
        self.timer = threading.Timer(seconds + (0.0 if rank == 0 else 10.0), self._expired, (seconds,))
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
