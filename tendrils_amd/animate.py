"""Keyframe timelines and the player that drives `tendrils.state` from them: the host-side mirror of the reference's
animation utilities (src/animate/timeline.js:49-398, src/animate/index.js:13-129, src/animate/tween.js:10-48,
src/animate/join-curve.js:6-9, with the `lerp` / `bezier` / `clamp` packages they use).

A *frame* is a dict {to, time, ease, call}; a Timeline keeps frames sorted by time between two infinite end frames
and holds a playhead (time, gap = position between two frames, span = that pair with the in-between offset t);
`play()` also gathers what frames skipped since the last call would have set.  A Player owns named timelines and
applies each one's current span to its output object (plain dicts here; the demo points them at `tendrils.state`):
`apply` values are assigned, numbers are tweened from the output's current value along the span's ease curve, `call`
entries are invoked.  Same method names, argument order and arithmetic as the reference (pinned value for value by
tests/test_animate_reference.py against the reference's own compiled classes).
"""
import math

INF = float("inf")


def is_number(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def clamp(v, lo, hi):                      # the `clamp` package: min < max ? (v < min ? min : v > max ? max : v) : ...
    if lo < hi:
        return lo if v < lo else (hi if v > hi else v)
    return hi if v < hi else (lo if v > lo else v)


def lerp(a, b, t):                         # the `lerp` package
    return a * (1 - t) + b * t


def bezier(points, t):
    """1-D Bezier curve through control values (the `bezier` package): closed forms up to 4 points, de Casteljau beyond."""
    n = len(points)
    if n == 0:
        raise ValueError("Cannot create a interpolator with no elements")
    if n == 1:
        return points[0]
    if n == 2:
        return points[0] + (points[1] - points[0]) * t
    ut = 1 - t
    if n == 3:
        return (points[0] * ut + points[1] * t) * ut + (points[1] * ut + points[2] * t) * t
    if n == 4:
        a1 = points[1] * ut + points[2] * t
        return ((points[0] * ut + points[1] * t) * ut + a1 * t) * ut + (a1 * ut + (points[2] * ut + points[3] * t) * t) * t
    p = [points[r] * ut + points[r + 1] * t for r in range(n - 1)]
    while len(p) > 1:
        p = [p[r] * ut + p[r + 1] * t for r in range(len(p) - 1)]
    return p[0]


def join_curve(curve, align=1):            # src/animate/join-curve.js:6-9
    if not curve:
        return 0
    if len(curve) == 1:
        return curve[0]
    return (curve[-1] - curve[-2]) * align


# ---- tween (src/animate/tween.js) ---------------------------------------------------------------------------------
def tween_value(a, b, t, ease=None):
    if a == b or not is_number(a):
        return b
    return lerp(a, b, bezier(ease, t) if ease else t)


def _keys(obj):                            # own enumerable keys of an object or an array (colours are arrays)
    return list(obj.keys()) if isinstance(obj, dict) else list(range(len(obj)))


def _get(obj, k):
    if obj is None:
        return None
    if isinstance(obj, dict):
        return obj.get(k)
    return obj[k] if isinstance(k, int) and 0 <= k < len(obj) else None


def _tweenable(k, values, defaults):
    v = _get(values, k) if values else None
    return v if is_number(v) else (_get(defaults, k) if defaults else None)


def tween_props(a, b, t, ease=None, out=None):
    out = {} if out is None else out
    if not b:
        return out
    for k in _keys(b):
        va, vb = _tweenable(k, a, out), _tweenable(k, b, out)
        v = tween_value(va, vb, t, ease) if (is_number(va) and is_number(vb)) else (va if t < 1 else vb)
        if isinstance(out, list):
            while len(out) <= k:
                out.append(None)
        out[k] = v
    return out


def tween(span, out=None):
    """tween({a, b, t, ease}, out): the object form the Player uses"""
    return tween_props(span.get("a"), span.get("b"), span.get("t"), span.get("ease"), out)


# ---- timeline (src/animate/timeline.js) ---------------------------------------------------------------------------
def make_frame(*args):                     # src/animate/frame.js: (to, time, ease, call) or one frame object
    if len(args) > 1:
        to, time = args[0], args[1]
        return {"to": to, "time": time, "ease": args[2] if len(args) > 2 else None, "call": args[3] if len(args) > 3 else None}
    return args[0]


def _time(frame):
    return frame.get("time") if frame else None


def _after(a, b):                          # order(a, b) > 0; a missing time compares like JS undefined: never greater
    ta, tb = _time(a), _time(b)
    return ta is not None and tb is not None and ta > tb


def offset(a, b, time):
    lo = min(a["time"], b["time"])
    span = max(a["time"], b["time"]) - lo
    # JS: ((time-min)/(max-min) || 0): NaN (inf/inf, 0/0) and 0 fall to 0
    if math.isinf(time - lo) and math.isinf(span):
        q = 0
    elif span == 0:
        q = 0 if (time - lo) == 0 or math.isnan(time - lo) else math.copysign(INF, time - lo)
    else:
        q = (time - lo) / span
    if q != q or q == 0:
        q = 0
    return clamp(q, 0, 1)


def within(a, b, time):
    return min(a["time"], b["time"]) < time <= max(a["time"], b["time"])


def _accumulate(frame, out):
    out.setdefault("apply", {})
    to = frame.get("to") or {}
    for k in _keys(to):
        out["apply"][k] = to[k]
    if frame.get("call"):
        out.setdefault("call", []).extend(frame["call"])
    return out


class Timeline:
    def __init__(self, frames=None, infinite=True, rewind=False, symmetric=True):
        self.frames = self.setup(frames, infinite)
        self.time = 0
        self.gap = -1
        self.span = None
        self.symmetric = symmetric
        self.infinite = infinite
        self.rewind = rewind
        self.reverse = None                # (read by play(); never set by the reference either)

    # -- keyframes ---------------------------------------------------------------------------------------------
    def setup(self, frames=None, infinite=True):
        frames = list(frames or [])
        if infinite:
            frames = [{"time": -INF}] + frames + [{"time": INF}]
        # Array.prototype.sort with order(a, b) = a.time > b.time ? 1 : -1: a stable insertion gives the same result
        # for the inputs used here (distinct or already ordered times)
        out = []
        for f in frames:
            k = len(out)
            while k > 0 and _after(out[k - 1], f):
                k -= 1
            out.insert(k, f)
        self.frames = out
        return out

    def merge(self, frames):
        for f in frames:
            self.add(f)
        return frames

    def insert_frame(self, f, frame):
        self.frames.insert(f, frame)
        return self

    def add(self, *frame):
        adding = make_frame(*frame)
        f = self.index_of(adding)
        self.insert_frame(f, adding)
        return f

    def add_span(self, duration, *frame):
        f = self.add(*frame)
        t0 = self.frames[f]["time"] - duration
        past = self.frames[f - 1] if f - 1 >= 0 else None
        if duration and (past is None or past["time"] < t0):
            self.add(None, t0)
        return f

    # -- playback ----------------------------------------------------------------------------------------------
    def seek(self, time):
        if self.valid() and within(self.span["past"], self.span["next"], time):
            self.span["t"] = offset(self.span["past"], self.span["next"], time)
        else:
            self.set_time(time)
        return self.span

    def play(self, time):
        gap0 = max(self.gap, 0.5)
        span = self.seek(time)
        if self.valid():
            accumulated = {}
            passed = self.gap - gap0
            skipped = abs(passed)
            direction = (passed > 0) - (passed < 0)
            onwards = ((-direction if self.reverse else direction) > 0)
            if skipped > 0 and onwards:
                side = math.floor if direction < 0 else math.ceil
                f = 0
                while f < skipped:
                    _accumulate(self.frames[int(side(gap0 + (f * direction)))], accumulated)
                    f += 1
            span = dict(span)
            span.update(accumulated)
        return span

    def play_from(self, time=None, start=0):
        time = self.time if time is None else time
        self.seek(start)
        return self.play(time)

    def set_time(self, time):
        gap = self.gap_at(time)
        self.span = self.span_gap_at(time, gap, self.span)
        self.gap = gap
        self.time = time
        return self

    # -- queries -----------------------------------------------------------------------------------------------
    def index_of(self, frame):
        for k, other in enumerate(self.frames):
            if _after(other, frame):
                return k
        return len(self.frames)

    def gap_at(self, time):
        if len(self.frames) < 2:
            return -1
        nxt = -1
        for k, frame in enumerate(self.frames):
            if _time(frame) is not None and frame["time"] >= time:
                nxt = k
                break
        return (len(self.frames) - 1 if nxt < 0 else max(nxt, 1)) - 0.5

    def span_gap_at(self, time, gap=None, out=None):
        gap = self.gap_at(time) if gap is None else gap
        if gap < 0:
            return None
        out = {} if out is None else out
        past = self.frames[int(math.floor(gap))]
        nxt = self.frames[int(math.ceil(gap))]
        ease = nxt.get("ease")
        if self.rewind:
            if not self.symmetric:
                ease = past.get("ease")
            past, nxt = nxt, past
        out["past"], out["next"] = past, nxt
        out["a"], out["b"] = past.get("to"), nxt.get("to")
        out["t"] = offset(past, nxt, time)
        out["ease"] = ease
        return out

    # -- joining new frames to those before --------------------------------------------------------------------------
    def to(self, *frame):
        self.add(*frame)
        return self

    def ease_to(self, align, *frame):
        self.ease_join(self.add(*frame), align)
        return self

    def smooth_to(self, *frame):
        return self.ease_to(1, *frame)

    def flip_to(self, *frame):
        return self.ease_to(-1, *frame)

    def over(self, duration, *frame):
        self.add_span(duration, *frame)
        return self

    def ease_over(self, duration, align, *frame):
        self.ease_join(self.add_span(duration, *frame), align)
        return self

    def smooth_over(self, duration, *frame):
        return self.ease_over(duration, 1, *frame)

    def flip_over(self, duration, *frame):
        return self.ease_over(duration, -1, *frame)

    def ease_join(self, f, align):
        ease = None
        if f > 0:
            frame = self.frames[f]
            ease = frame["ease"] if frame.get("ease") else [0, 1]
            ease.insert(1, join_curve(self.frames[f - 1].get("ease"), align))
            frame["ease"] = ease
        return ease

    # -- etc ----------------------------------------------------------------------------------------------------
    def valid(self, gap=None, span=None):
        gap = self.gap if gap is None else gap
        span = self.span if span is None else span
        return gap > 0 and bool(span)

    def start(self):
        return self.frames[0]["time"] if self.frames else None

    def end(self):
        return self.frames[-1]["time"] if self.frames else None

    def duration(self):
        return (self.end() or 0) - (self.start() or 0)

    # the reference's camelCase names
    insertFrame, addSpan, playFrom, setTime, indexOf, gapAt, spanGapAt = insert_frame, add_span, play_from, set_time, index_of, gap_at, span_gap_at
    easeTo, smoothTo, flipTo, easeOver, smoothOver, flipOver, easeJoin = ease_to, smooth_to, flip_to, ease_over, smooth_over, flip_over, ease_join


def apply_span(span, out=None):            # src/animate/index.js:13-22 `apply`
    out = {} if out is None else out
    if span:
        ap = span.get("apply") or {}
        for k in _keys(ap):
            if isinstance(out, list):
                while len(out) <= k:
                    out.append(None)
            out[k] = ap[k]
        tween(span, out)
        for f in span.get("call") or []:
            f(out, span)
    return out


class Player:
    def __init__(self, tracks, outputs=None):
        self.tracks = tracks                # dict name -> frames list | Timeline (converted in place)
        self.outputs = {} if outputs is None else outputs
        self.add(self.tracks)

    def add(self, tracks):
        for key in list(tracks.keys()):
            track = tracks[key]
            self.tracks[key] = track if isinstance(track, Timeline) else Timeline(track)
        return self

    def each(self, f):
        for key in list(self.tracks.keys()):
            f(self.tracks[key], key)
        return self

    def apply(self, f, out=None):
        out = self.outputs if out is None else out

        def one(track, key):
            if not out.get(key):
                out[key] = {}
            track_out = out[key]
            return apply_span(f(track, key, track_out), track_out)
        self.each(one)
        return self

    def seek(self, time, out=None):
        return self.apply(lambda track, *_: track.seek(time), out)

    def play(self, time, out=None):
        return self.apply(lambda track, *_: track.play(time), out)

    def play_from(self, time, start, out=None):
        return self.apply(lambda track, *_: track.play_from(time, start), out)

    playFrom = play_from

    def frames(self):
        return {k: t.frames for k, t in self.tracks.items()}

    def start(self):
        return _reduce_min([t.start() for t in self.tracks.values()])

    def end(self):                          # (the reference takes the minimum here too: src/animate/index.js:121-124)
        return _reduce_min([t.end() for t in self.tracks.values()])

    def duration(self):
        return (self.end() or 0) - (self.start() or 0)


def _reduce_min(values):
    """reduce((acc, v) => Math.min(v, acc), tracks, null): Math.min treats null as 0"""
    acc = None
    for v in values:
        acc = min(v, 0 if acc is None else acc)
    return acc
