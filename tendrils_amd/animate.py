"""Keyframe tracks for the headless scene replay: what a host needs to drive `tendrils.state` the way the
reference's demo does with its `Player` / `Timeline` / `tween` (src/animate/index.js, timeline.js, tween.js; used at
src/demo.main.js:836-865, :928-953, :1027-1031, :1267-1274).  Method names (snake_case here), argument order and every
observable number are the reference's - pinned by tests/golden/animate_script.json, a script run on the reference's own
compiled classes (values after every play / seek / play_from, the playheads, edits made while playing, the return values
of the queries), and by tests/golden/scene_*.npz (the reference's Player driving the reference's Tendrils).

How it is built (the same design as tendrils_amd/js/animate.js):

* a track keeps its keys in time order in two lists in lock-step: `stamps` (plain numbers, bisected with the standard
  library) and `keys` (the records - dicts {"to", "time", "ease", "call"}, the caller's own where it passed dicts).  There
  are no end records: an open-ended track answers for "before the first key" / "after the last key" with two shared
  constants, and the reference's frame numbering (its -Infinity frame is number 0) is an offset applied at the surface;
* the playhead is a cached interval (`Held`: the two stamps it lies between, the records at either end, the curve) plus the
  reference's half-integer `gap`.  A seek that stays inside the cached interval only moves `t`; nothing is looked up again
  until the time leaves it.  Edits do not touch the playhead - as in the reference, a key added inside the cached
  interval is noticed when the time next leaves that interval;
* easing is one closed-form blend per number: the curve by de Casteljau in the a*(1-u) + b*u form of the `bezier`
  package (its two-point special case included), the mix as a*(1-w) + b*w (`lerp` package).

Outputs and key values are dicts or lists (colours); a key's "call" is a list of callables f(out, span).

One deliberate difference: frames of EQUAL time in a list handed to the constructor / setup() keep the order they were
given in.  The reference sorts that list with a comparator that answers -1 for a tie (`order`, src/animate/timeline.js),
which hands their order to the engine's sort algorithm (a V8 with TimSort turns A, B into B, A; an insertion sort need not)
- nothing the reference's source defines, so nothing is pinned there.  Ties met by add() follow the reference's rule.
"""
import math
from bisect import bisect_left, bisect_right
from dataclasses import dataclass

INF = math.inf
BEFORE = {"time": -INF}
AFTER = {"time": INF}


def numeric(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def identical(x, y):
    """Numbers, strings, booleans and None by value, everything else by identity."""
    if numeric(x) and numeric(y):
        return x == y
    if isinstance(x, (str, bool, type(None))):
        return type(x) is type(y) and x == y
    return x is y


def entries(any_):
    """(name, value) pairs of a dict or a list; nothing for None / scalars."""
    if isinstance(any_, dict):
        return list(any_.items())
    if isinstance(any_, (list, tuple)):
        return list(enumerate(any_))
    return []


def peek(any_, name):
    if isinstance(any_, dict):
        return any_.get(name)
    if isinstance(any_, (list, tuple)) and isinstance(name, int) and 0 <= name < len(any_):
        return any_[name]
    return None


def poke(out, name, value):
    if isinstance(out, list) and isinstance(name, int) and name >= len(out):
        out.extend([None] * (name + 1 - len(out)))
    out[name] = value


# ---- curves --------------------------------------------------------------------------------------------------------

def curve_at(points, u):
    """Value at `u` of the 1-D Bezier curve with the given control values."""
    n = len(points)
    if n < 2:
        return points[0] if n else u
    if n == 2:
        return points[0] + (points[1] - points[0]) * u
    w, v = list(points), 1 - u
    for top in range(n - 1, 0, -1):
        for r in range(top):
            w[r] = w[r] * v + w[r + 1] * u
    return w[0]


def join_curve(curve, align=1):
    """The control value that continues `curve` into the next one: its last leg, mirrored (align 1) or flipped (-1)."""
    n = len(curve) if curve else 0
    return 0 if n == 0 else (curve[0] if n == 1 else (curve[-1] - curve[-2]) * align)


# ---- blending ------------------------------------------------------------------------------------------------------

def tween_value(a, b, t, ease=None):
    if not numeric(a) or a == b:
        return b
    w = curve_at(ease, t) if ease is not None else t
    return a * (1 - w) + b * w


def tween_props(a, b, t, ease=None, out=None):
    """Every entry `b` names, written into `out`: numbers are blended from `a` (or, where `a` has no number of that name,
    from what `out` holds), anything else switches over when t reaches 1."""
    out = {} if out is None else out
    for k, target in entries(b):
        start = peek(a, k)
        start = start if numeric(start) else peek(out, k)
        target = target if numeric(target) else peek(out, k)
        poke(out, k, tween_value(start, target, t, ease) if numeric(start) and numeric(target) else (start if t < 1 else target))
    return out


def tween(span, out=None):
    """span = {"a", "b", "t", "ease"}: two numbers give the number, two collections are blended into `out`."""
    a, b, t, ease = span.get("a"), span.get("b"), span.get("t"), span.get("ease")
    return tween_value(a, b, t, ease) if numeric(b) else tween_props(a, b, t, ease, out)


def apply_span(span, out=None):
    """What a move of the playhead does to its output: keys that were jumped over land whole, the interval the head stands
    in is blended, then the jumped-over keys' calls run."""
    out = {} if out is None else out
    if span:
        for k, v in entries(span.get("apply")):
            poke(out, k, v)
        tween(span, out)
        for f in span.get("call") or ():
            f(out, span)
    return out


# ---- a track -------------------------------------------------------------------------------------------------------

def make_frame(*args):
    """(to, time, ease, call) or one ready record."""
    if len(args) > 1:
        return dict(zip(("to", "time", "ease", "call"), tuple(args) + (None,) * (4 - len(args))))
    return args[0]


def fraction(lo, hi, time):
    if hi == lo:
        return 1.0 if time > lo else 0
    f = (time - lo) / (hi - lo)
    return 0 if (f != f or f < 0) else (1 if f > 1 else f)      # NaN (an infinite interval) counts as its start


@dataclass
class Held:
    """The cached interval of a playhead."""
    lo: float
    hi: float
    past: dict
    next: dict
    ease: object
    t: float

    def span(self):
        return {"past": self.past, "next": self.next, "a": self.past.get("to"), "b": self.next.get("to"), "t": self.t,
                "ease": self.ease}


class Timeline:
    def __init__(self, frames=None, infinite=True, rewind=False, symmetric=True):
        self.infinite = infinite
        self.rewind = rewind            # the interval is handed out back to front (next <-> past)
        self.symmetric = symmetric      # rewinding keeps the later key's curve; if not, the curve of the key being approached
        self.reverse = False            # read by play() and splice_at(): the direction that counts as "onwards"
        self.time = 0
        self.gap = -1
        self.held = None
        self.setup(frames, infinite)

    # -- numbering: frame number = key index + base (an open-ended track counts its "before" end as frame 0)

    @property
    def base(self):
        return 1 if self.infinite else 0

    @property
    def size(self):
        return len(self.keys) + 2 * self.base

    def at(self, number):
        i = int(number) - self.base
        if 0 <= i < len(self.keys):
            return self.keys[i]
        if self.infinite and i in (-1, len(self.keys)):
            return BEFORE if i < 0 else AFTER
        return None

    @property
    def frames(self):
        """The reference's array, rebuilt on request (a view: edit through the methods)."""
        return [BEFORE, *self.keys, AFTER] if self.infinite else list(self.keys)

    @property
    def span(self):
        return self.held.span() if self.held else None

    # -- keys

    def setup(self, frames=None, infinite=None):
        if infinite is not None:
            self.infinite = infinite
        self.keys = sorted(frames or [], key=lambda frame: frame["time"])          # stable
        self.stamps = [frame["time"] for frame in self.keys]
        return self.frames

    def merge(self, frames):
        for _, frame in entries(frames):
            self.add(frame)
        return frames

    def insert_frame(self, number, frame):
        i = min(max(number - self.base, 0), len(self.keys))
        self.keys.insert(i, frame)
        self.stamps.insert(i, frame["time"])
        return self

    def index_of(self, frame):
        return bisect_right(self.stamps, frame["time"]) + self.base

    def add(self, *frame):
        record = make_frame(*frame)
        number = self.index_of(record)
        self.insert_frame(number, record)
        return number

    def add_span(self, duration, *frame):
        """The key, and before it a key without values `duration` earlier - where its transition starts - unless another
        key already stands inside that stretch.  Returns the number the key had when it went in."""
        number = self.add(*frame)
        start = self.at(number)["time"] - duration
        before = self.at(number - 1)
        if duration and (before is None or before["time"] < start):
            self.add(None, start)
        return number

    # -- playhead

    def gap_at(self, time):
        if self.size < 2:
            return -1
        i = bisect_left(self.stamps, time)
        number = max(i + self.base, 1) if (self.infinite or i < len(self.keys)) else self.size - 1
        return number - 0.5

    def interval(self, time, gap):
        if not gap >= 0:
            return None
        early, late = self.at(gap - 0.5), self.at(gap + 0.5)
        lo, hi = min(early["time"], late["time"]), max(early["time"], late["time"])
        turned = bool(self.rewind)
        return Held(lo, hi, late if turned else early, early if turned else late,
                    (early if (turned and not self.symmetric) else late).get("ease"), fraction(lo, hi, time))

    def span_gap_at(self, time, gap=None, out=None):
        h = self.interval(time, self.gap_at(time) if gap is None else gap)
        if h is None:
            return None
        out = {} if out is None else out
        out.update(h.span())
        return out

    def valid(self, gap=None, span=None):
        gap = self.gap if gap is None else gap
        span = self.span if span is None else span
        return span if (gap > 0 and span) else False

    def set_time(self, time):
        self.gap = self.gap_at(time)
        self.held = self.interval(time, self.gap)
        self.time = time
        return self

    def seek(self, time):
        h = self.held
        if self.gap > 0 and h is not None and h.lo < time <= h.hi:
            h.t = fraction(h.lo, h.hi, time)
        else:
            self.set_time(time)
        return self.span

    def play(self, time):
        """seek(), and what the head jumped over on its way (going onwards only) comes along as "apply" / "call"."""
        start = max(self.gap, 0.5)
        span = self.seek(time)
        if not self.valid():
            return span
        steps, down = int(abs(self.gap - start)), self.gap < start
        if steps > 0 and down == bool(self.reverse):
            span["apply"] = {}
            for j in range(steps):
                frame = self.at(start - 0.5 - j if down else start + 0.5 + j)
                span["apply"].update(entries(frame.get("to")))
                if frame.get("call"):
                    span["call"] = list(span.get("call") or ()) + list(frame["call"])
        return span

    def play_from(self, time=None, start=0):
        time = self.time if time is None else time
        self.seek(start)
        return self.play(time)

    # -- taking keys out (frame numbers; an open-ended track keeps its ends, and - like the reference - its last key)

    def splice(self, index=0, num=0, *adding):
        first, count = index, num
        if self.infinite:
            n = len(self.keys)
            wanted = n + index if index < 0 else index
            first = min(n, max(1, wanted))
            count = min(num - max(first - wanted, 0), n - first)
        i = max((self.size + first if first < 0 else min(first, self.size)) - self.base, 0)
        count = max(int(count), 0)
        gone = self.keys[i:i + count]
        self.keys[i:i + count] = list(adding)
        self.stamps[i:i + count] = [frame["time"] for frame in adding]
        return gone

    def splice_index(self, index, *adding):
        gone = self.splice(index, 1, *adding)
        return gone[0] if gone else None

    def splice_at(self, time, adjacent=-1, *adding):
        """Takes out the key next to `time`: adjacent -1 the one before it, 1 the one after (in the playing direction)."""
        gap = self.gap_at(time)
        ahead = (-adjacent if self.reverse else adjacent) > 0
        gone = self.splice(math.ceil(gap) if ahead else math.floor(gap), 1, *adding)
        return gone[0] if gone else None

    def splice_span(self, duration, start=0, *adding):
        a, b = self.gap_at(start), self.gap_at(start + duration)
        low = min(a, b)
        return self.splice(math.ceil(low), math.floor(max(a, b) - low), *adding)

    # -- adding keys, the chainable forms

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
        """The joined curve goes to the frame that now carries the number the key went in with: the transition's start key
        when add_span() made one (the key itself keeps its curve as given) - as the reference's timelines come out."""
        self.ease_join(self.add_span(duration, *frame), align)
        return self

    def smooth_over(self, duration, *frame):
        return self.ease_over(duration, 1, *frame)

    def flip_over(self, duration, *frame):
        return self.ease_over(duration, -1, *frame)

    def ease_join(self, number, align):
        """Gives frame `number` a curve that leaves the previous frame's curve without a kink: [first, joined, *rest]."""
        if not number > 0:
            return None
        frame = self.at(number)
        own = list(frame.get("ease") or (0, 1))
        frame["ease"] = [own[0], join_curve(self.at(number - 1).get("ease"), align), *own[1:]]
        return frame["ease"]

    def min_frame(self, *frame):
        """The frame as add() would take it, its "to" cut down to the entries that are not found - same name, same value - on
        the record standing before its place and on the one standing after the next (the comparison the reference makes)."""
        full = make_frame(*frame)
        number = self.index_of(full)
        to = full.get("to")

        def unlike(record):
            if not (record and record.get("to")) or record is to:
                return None
            if not entries(record) or not entries(to):
                return to
            kept = {k: v for k, v in entries(to) if not (k in record and identical(record[k], v))}
            return kept or None
        early, late = unlike(self.at(number - 1)), unlike(self.at(number + 1))
        both = {**dict(entries(early)), **dict(entries(late))}
        return {**full, "to": both if both else early}

    # -- extent

    def start(self):
        return self.at(0)["time"] if self.size else None

    def end(self):
        return self.at(self.size - 1)["time"] if self.size else None

    def duration(self):
        return (self.end() or 0) - (self.start() or 0)


# ---- tracks side by side -------------------------------------------------------------------------------------------

class Player:
    """`tracks`: a dict (or list) of key lists or Timelines - key lists are replaced, in place, by Timelines; `outputs`:
    the collection, keyed alike, whose members the tracks write into."""

    def __init__(self, tracks, outputs=None):
        self.tracks = tracks
        self.outputs = {} if outputs is None else outputs
        self.add(tracks)

    def add(self, tracks):
        for k, track in entries(tracks):
            poke(self.tracks, k, track if isinstance(track, Timeline) else Timeline(track))
        return self

    def import_(self, players):
        """Takes over the tracks of other players under their keys.  (The reference's `import` feeds each timeline to add() as
        if it were a collection of tracks and cannot work; nothing calls it there.  This is what its name promises.)"""
        for _, p in entries(players):
            self.add(p.tracks)
        return self

    def each(self, f):
        for k, track in entries(self.tracks):
            f(track, k)
        return self

    def apply(self, f, out=None):
        """f(track, key, out[key]) returns what to apply to out[key]: a span, {"apply": ...}, {"call": ...} or nothing."""
        out = self.outputs if out is None else out

        def one(track, key):
            mine = peek(out, key)
            if mine is None:
                mine = {}
                poke(out, key, mine)
            apply_span(f(track, key, mine), mine)
        return self.each(one)

    def seek(self, time, out=None):
        return self.apply(lambda track, *_: track.seek(time), out)

    def play(self, time, out=None):
        return self.apply(lambda track, *_: track.play(time), out)

    def play_from(self, time, start, out=None):
        return self.apply(lambda track, *_: track.play_from(time, start), out)

    def frames(self):
        return {k: track.frames for k, track in entries(self.tracks)}

    def extent(self, which):
        """Both ends are the smallest of zero and the tracks' own (the reference folds with a minimum from a null start, for
        the end as well - a player of open-ended tracks ends at 0); None without tracks."""
        ends = [getattr(track, which)() or 0 for _, track in entries(self.tracks)]
        return min(0, *ends) if ends else None

    def start(self):
        return self.extent("start")

    def end(self):
        return self.extent("end")

    def duration(self):
        return (self.end() or 0) - (self.start() or 0)
