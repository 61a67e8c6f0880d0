"""The view pass of Tendrils.draw() (src/index.js:315-337, src/render/index.vert:58-100): the restatement against
captures of the REFERENCE's own view render - the default framebuffer (RGBA8) after draw(), on a context without
multisampling - for the input states of the deposit fixtures.

What is pinned: the coverage, pixel for pixel (the view's width-1 lines follow the flow pass's rasteriser); the colours
to +-1 of 255 per channel - the varying interpolation, sin() and the 8-bit store rounding of the captured GL are
implementation-defined; on these captures at most 1 % of the touched pixels differ at all."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, golden


def view_fixture(path):
    d = np.load(path)
    m = json.loads(str(d["uniforms"]))
    src = np.load(os.path.join(GOLDEN, m["source"] + ".npz"))
    fw, fh = m["viewRes"]
    ref = np.zeros((fh * fw, 4), np.uint8)
    ref[d["idx"]] = d["val"]
    return m, src["current"], src["previous"], ref.reshape(fh, fw, 4)


def check_against_reference(got, ref):
    assert (got.any(-1) == ref.any(-1)).all(), "coverage differs"
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32)).max(-1)
    assert diff.max() <= 1
    assert (diff > 0).sum() <= max(1, int(0.01 * ref.any(-1).sum()))


@pytest.mark.parametrize("path", golden("view"), ids=lambda p: p.split("/")[-1][:-4])
def test_view_render_matches_reference_capture(oracle, path):
    m, cur, prev, ref = view_fixture(path)
    assert m["samples"] == 0 and m["render"]["lineWidth"] == 1
    fh, fw = ref.shape[:2]
    got, n = oracle.view_render(cur, prev, np.zeros((fh, fw, 4), np.uint8), m["time"], view_size=m["viewSize"], **m["render"])
    assert n >= ref.any(-1).sum() > 0
    check_against_reference(got, ref)


def test_view_fill_is_a_blended_quad(oracle):
    v = np.zeros((4, 5, 4), np.uint8)
    v[..., :] = [200, 100, 50, 255]
    out = oracle.view_fill(v, [0.1333, 0.1333, 0.1333, 0.25])
    want = np.round((0.1333 * 0.25 + v[0, 0, :3] / 255.0 * 0.75) * 255).astype(np.uint8)
    assert (out[..., :3] == want).all() and (out[..., 3] == round((0.25 * 0.25 + 0.75) * 255)).all()
