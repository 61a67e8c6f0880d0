"""Host mirror of the reference's Timer (src/timer.js:1-80): same fields, same tick() rules.
Times are milliseconds held in Python floats (= JS doubles)."""
import math
import time as _time


def _date_now():
    return _time.time() * 1000.0


class Timer:
    def __init__(self, now=None, since=None):
        self.time = 0
        self.since = 0
        self.offset = 0
        self.rate = 1
        self.step = -1
        self.dt = 0
        self.paused = False
        self.end = -1
        self.loop = False
        self.reset(now, since)

    def now(self, now=None):
        now = _date_now() if now is None else now
        return (now - self.offset) * self.rate

    def tick(self, now=None):                       # src/timer.js:24-60
        time = self.time
        dt = 0
        if self.step >= 0:
            dt = self.step * self.rate
            time += dt
        else:
            past = time
            time = self.now(now)
            dt = time - past
        if self.paused:
            self.offset += dt
            dt = 0
        elif self.end < 0:
            self.time = time
        elif self.loop:
            self.time = math.fmod(time, self.end)    # JS % keeps the sign of the dividend
        else:
            self.time = (min if self.rate > 0 else max)(time, self.end)
            if self.time != time:
                self.paused = True
        self.dt = dt
        return self

    def seek(self, to):
        self.offset = -to
        return self

    def scrub(self, by):
        self.offset -= by
        return self

    def reset(self, now=None, since=None):
        now = _date_now() if now is None else now
        since = now if since is None else since
        self.since = self.offset = since
        self.time = self.now(now)
        return self
