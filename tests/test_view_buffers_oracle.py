"""Tendrils.buffers against the REFERENCE: scripts of its own setupBuffers / draw / copyBuffer / drawBuffer / stepBuffers /
clearView calls were run on the reference's Tendrils (oracle/gen_fixtures.py:gen_buffers -> tests/golden/buffers_*.npz) and
every image the scripts read on the way was kept.  Here the semantics the two hosts implement - which image draw() binds,
what clearView() leaves bound, what copyBuffer blends into what, how the ring rotates under a bound buffer - are restated
in a dozen lines over the CPU restatement's passes and held against those captures; tests/test_gpu_view_buffers.py replays
the same scripts on the hosts themselves.

Pinned: coverage of every image exactly; values to +-1 of 255 (the captured GL's varying interpolation, sin() and 8-bit
rounding are implementation-defined, as for the view pass - tests/test_view_oracle.py), +-2 after a copy of a copy."""
import numpy as np
import pytest

from helpers import blend_rgba8_over, buffers_fixture, golden


class RingModel:
    """src/index.js:172-184, 215-229, 278-340, 342-391 over the restatement's passes"""

    def __init__(self, O, m, cur, prev):
        self.O, self.m = O, m
        # src/index.js:28-64 (the restatement's table holds the uniforms of the passes; the draw's switches and the widths beside)
        self.state = dict(O.DEFAULT_STATE, autoClearView=False, autoFade=True, lineWidth=1, speedAlpha=0.000001,
                          colorMapAlpha=0.4, baseColor=[1, 1, 1, 0.5], flowColor=[1, 1, 1, 0.04], fadeColor=[0.1333, 0.1333, 0.1333, 0])
        self.state.update(m["state"])
        self.cur, self.prev = cur.copy(), prev.copy()
        fw, fh = m["viewRes"]
        self.flow = np.zeros((fh, fw, 4), np.float32)
        self.blank = lambda: np.zeros((fh, fw, 4), np.uint8)
        self.screen = self.blank()
        self.buffers = [self.blank() for _ in range(m["numBuffers"])]
        self.bound = None                   # None: the screen; else the bound image itself (GL binds the object)
        self.time, self.dt = m["time0"], 1000.0 / 60.0

    def target(self):
        return self.screen if self.bound is None else self.bound

    def write(self, img):
        self.target()[...] = img

    def run(self, op):
        O, s, what = self.O, self.state, op[0]
        if what == "tickStep":
            self.time += self.dt
            u = O.logic_uniforms(self.m["N"], self.m["N"], self.time, self.dt, view_size=self.m["viewSize"],
                                 **{k: v for k, v in s.items() if isinstance(v, (int, float)) and not isinstance(v, bool)})
            self.prev, self.cur = self.cur, O.logic_step(u, self.cur, self.flow)
        elif what == "draw":
            self.flow, _ = O.flow_deposit(self.cur, self.prev, self.flow, self.time, view_size=self.m["viewSize"], speedLimit=s["speedLimit"])
            self.bound = self.buffers[0] if self.buffers else None          # src/index.js:318-325
            if s["autoClearView"]:
                self.run(["clearView"])                                        # ... which leaves the screen bound
            if s["autoFade"]:
                self.run(["drawFade"])
            render = {k: s[k] for k in ("speedLimit", "flowDecay", "speedAlpha", "colorMapAlpha", "baseColor", "flowColor", "lineWidth")}
            out, _ = O.view_render(self.cur, self.prev, self.target().copy(), self.time, view_size=self.m["viewSize"], **render)
            self.write(out)
        elif what == "clearView":                                              # src/index.js:220-229
            for b in self.buffers:
                b[...] = 0
            self.screen[...] = 0
            self.bound = None
        elif what == "drawFade":
            if s["fadeColor"][3] > 0:
                self.run(["drawFill", s["fadeColor"]])
        elif what == "drawFill":
            self.write(blend_rgba8_over(np.array(op[1], np.float32), self.target()))
        elif what == "copyBuffer":
            if op[1] < len(self.buffers):
                self.write(blend_rgba8_over(self.buffers[op[1]], self.target()))
        elif what == "drawBuffer":                                             # src/index.js:359-367
            self.bound = None
            if s["autoClearView"]:
                self.screen[...] = 0
            self.run(["copyBuffer", 0 if op[1] is None else op[1]])
            self.run(["stepBuffers"])
        elif what == "stepBuffers":
            if len(self.buffers) > 1:
                self.buffers.insert(0, self.buffers.pop())
        elif what == "setupBuffers":
            while len(self.buffers) < op[1]:
                self.buffers.append(self.blank())
            while len(self.buffers) > op[1]:
                gone = self.buffers.pop()
                if self.bound is gone:
                    self.bound = None
        elif what == "set":
            s[op[1]] = op[2]
        elif what == "bind":
            self.bound = None if op[1] < 0 else self.buffers[op[1]]
        elif what == "viewport":
            pass
        elif what == "read":
            self.bound = None if op[1] < 0 else self.buffers[op[1]]
            return self.target().copy()
        else:
            raise ValueError(what)
        return None


def close_to_reference(got, ref, k, copied):
    assert (got.any(-1) == ref.any(-1)).all(), "read %d: coverage differs" % k
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32)).max(-1)
    assert diff.max() <= (2 if copied else 1), "read %d: off by %d" % (k, diff.max())


@pytest.mark.parametrize("path", golden("buffers"), ids=lambda p: p.split("/")[-1][:-4])
def test_ring_semantics_match_the_reference_capture(oracle, path):
    m, cur, prev, images = buffers_fixture(path)
    assert m["samples"] == 0 and len(images) == sum(1 for op in m["ops"] if op[0] == "read") >= 2
    model, k, copied = RingModel(oracle, m, cur, prev), 0, False
    for op, length in zip(m["ops"], m["lengths"]):
        got = model.run(op)
        copied = copied or op[0] in ("copyBuffer", "drawBuffer")
        assert len(model.buffers) == length
        if got is not None:
            close_to_reference(got, images[k], k, copied)
            k += 1
    assert k == len(images) and any(i.any() for i in images)
