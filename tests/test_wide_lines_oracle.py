"""Lines wider than 1 (gl.lineWidth(flowWidth / lineWidth), src/index.js:302,336) in the CPU restatement.  UNPINNED against the
reference: the GL its fixtures were captured on reports ALIASED_LINE_WIDTH_RANGE = [1, 1] and draws every width as 1 - which
stays the default everywhere.  What is checked here is the restatement against its own stated definition (oracle/
tendrils_oracle.c, WIDE LINES): a line of width w covers the texel centres within an L1 distance of w/2 texels of the segment
(the width-1 construction - the hexagon of the two endpoint diamonds - with the diamonds scaled by w), which makes it exactly
w texels thick in its minor direction as OpenGL ES 2.0 3.4.2.1 asks of a non-antialiased wide line; width 1 is the pinned
case bit for bit.  CPU only."""
import numpy as np
import pytest

V = 64          # square target, viewSize (1, 1): texel k's centre at k + 0.5


def one_line(a, b, n=16):
    """A particle texture whose only drawable line runs from texel-space point a to b (row 2 pairs previous -> current)."""
    cur = np.full((n, n, 4), 0, np.float32)
    cur[..., :2] = -1e6
    prev = cur.copy()
    to_clip = lambda p: [p[0] / V * 2 - 1, p[1] / V * 2 - 1]
    prev[2, 3] = to_clip(a) + [0.002, -0.003]
    cur[2, 3] = to_clip(b) + [0.004, 0.001]
    return cur, prev


def l1_distance_to_segment(px, py, a, b):
    """min over t in [0, 1] of |px - x(t)| + |py - y(t)|: convex and piecewise linear in t - the ends and the two kinks"""
    ts = [0.0, 1.0]
    dx, dy = b[0] - a[0], b[1] - a[1]
    if dx:
        ts.append(min(1.0, max(0.0, (px - a[0]) / dx)))
    if dy:
        ts.append(min(1.0, max(0.0, (py - a[1]) / dy)))
    return min(abs(px - (a[0] + t * dx)) + abs(py - (a[1] + t * dy)) for t in ts)


@pytest.mark.parametrize("a,b,w", [((10.3, 20.4), (30.6, 20.4), 5),          # horizontal
                                   ((12.2, 8.3), (12.2, 41.7), 4),           # vertical
                                   ((9.3, 11.2), (41.8, 20.9), 3),           # x-major
                                   ((40.4, 50.3), (31.7, 14.2), 6),          # y-major, running down
                                   ((20.3, 20.6), (22.1, 21.4), 8),          # shorter than it is wide
                                   ((15.3, 30.2), (44.9, 52.4), 2.5),        # a width that is no integer
                                   ((-6.0, 30.3), (20.4, 36.2), 7),          # crossing the view's left edge (clipped)
                                   ((50.2, 58.3), (70.7, 69.1), 5)],         # leaving through the corner
                         ids=["horizontal", "vertical", "x_major", "y_major", "stub", "fractional", "clipped_left", "clipped_corner"])
def test_wide_line_covers_the_texels_within_half_its_width(oracle, a, b, w):
    cur, prev = one_line(a, b)
    _, fragments, cov = oracle.flow_deposit(cur, prev, np.zeros((V, V, 4), np.float32), 100.0, coverage=True, line_width=w)
    assert cov.max() == 1 and cov.sum() == fragments > 0           # a line never covers a texel twice
    margin = 0.08                                                   # endpoints and polygon vertices are snapped to 1/16 texel
    for y in range(V):
        for x in range(V):
            d = l1_distance_to_segment(x + 0.5, y + 0.5, a, b)
            if d < w / 2 - margin:
                assert cov[y, x] == 1, "texel (%d, %d) at L1 distance %.3f is not covered" % (x, y, d)
            elif d > w / 2 + margin:
                assert cov[y, x] == 0, "texel (%d, %d) at L1 distance %.3f is covered" % (x, y, d)


@pytest.mark.parametrize("w", [2, 3, 5, 8])
def test_wide_line_is_w_texels_thick_in_its_minor_direction(oracle, w):
    a, b = (8.3, 14.2), (52.7, 27.9)                                 # x-major
    cur, prev = one_line(a, b)
    _, _, cov = oracle.flow_deposit(cur, prev, np.zeros((V, V, 4), np.float32), 100.0, coverage=True, line_width=w)
    inner = cov[:, int(a[0] + w / 2 + 1):int(b[0] - w / 2 - 1)]
    assert (inner.sum(0) == w).all(), inner.sum(0)                   # a column of w fragments per step (ES 2.0 3.4.2.1)
    _, _, thin = oracle.flow_deposit(cur, prev, np.zeros((V, V, 4), np.float32), 100.0, coverage=True)
    assert ((thin > 0) <= (cov > 0)).all()                           # and it contains the width-1 line


def test_varying_is_constant_across_the_width(oracle):
    cur, prev = one_line((10.3, 20.4), (30.6, 20.4))
    got, _, cov = oracle.flow_deposit(cur, prev, np.zeros((V, V, 4), np.float32), 100.0, coverage=True, line_width=5)
    cols = [x for x in range(V) if cov[:, x].sum() == 5]
    assert len(cols) > 15
    for x in cols:
        col = got[cov[:, x] > 0, x].view(np.uint32)
        assert (col == col[0]).all()
    assert len({got[20, x, 0].tobytes() for x in cols}) == len(cols)     # ... and varies along the line


def test_width_one_is_the_pinned_rasteriser(oracle):
    from helpers import deposit_hashed_inputs
    cur, prev = deposit_hashed_inputs(64, 5, 1.2, 0.05, 17)
    base = np.zeros((48, 96, 4), np.float32)
    a, na, ca = oracle.flow_deposit(cur, prev, base, 300.0, view_size=(1.0, 0.5), coverage=True)
    for w in (1, 1.0, 0):                                                # (0: a zero-initialised struct of an older caller)
        b, nb, cb = oracle.flow_deposit(cur, prev, base, 300.0, view_size=(1.0, 0.5), coverage=True, line_width=w)
        assert na == nb and (ca == cb).all() and (a.view(np.uint32) == b.view(np.uint32)).all()
    c, nc, cc = oracle.flow_deposit(cur, prev, base, 300.0, view_size=(1.0, 0.5), coverage=True, line_width=5)
    assert nc > 3 * na and ((ca > 0) <= (cc > 0)).all()


def test_view_pass_draws_wide_lines_too(oracle):
    from helpers import deposit_hashed_inputs
    cur, prev = deposit_hashed_inputs(32, 9, 1.0, 0.05, 17)
    blank = np.zeros((40, 64, 4), np.uint8)
    thin, n1 = oracle.view_render(cur, prev, blank, 300.0, view_size=(1.0, 0.625))
    wide, n3 = oracle.view_render(cur, prev, blank, 300.0, view_size=(1.0, 0.625), line_width=3)
    assert n3 > 2 * n1 and (thin.any(-1) <= wide.any(-1)).all()
