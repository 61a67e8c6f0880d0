"""Flow deposit (Tendrils.draw()'s flow pass): the CPU restatement against captures of the
reference's own draw() (tests/golden/deposit_*.npz).  Coverage - which flow texels receive fragments - must
match texel for texel; values within the tolerance below (the rasteriser's varying interpolation is
implementation-defined arithmetic; blending order is the stream order).  CPU only."""
import numpy as np
import pytest

from helpers import golden, load

# per-channel tolerance of a deposited texel against the reference capture:
#   xy (velocities, |v| <= ~0.02):  5e-8 absolute;  z (time in ms): 1e-6 relative;  alpha: 1e-5 absolute
def deposit_close(got, ref, time):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    return (d[..., 0] <= 5e-8) & (d[..., 1] <= 5e-8) & (d[..., 2] <= 1e-6 * abs(time) + 1e-6) & (d[..., 3] <= 1e-5)


def deposit_inputs(fx):
    m = fx["meta"]
    if "hashed" in m and "current" not in fx:
        from helpers import deposit_hashed_inputs
        h = m["hashed"]
        fx["current"], fx["previous"] = deposit_hashed_inputs(m["N"], m["seed"], h["pos_range"], h["step"], h["inert"])
    fw, fh = m["viewRes"]
    base = fx["flow"] if "flow" in fx else np.zeros((fh, fw, 4), np.float32)
    ref = base.copy().reshape(-1, 4)
    ref[fx["idx"]] = fx["val"]
    return m, base, ref.reshape(fh, fw, 4)


@pytest.mark.parametrize("path", golden("deposit"), ids=lambda p: p.split("/")[-1][:-4])
def test_deposit_matches_reference_capture(oracle, path):
    fx = load(path)
    m, base, ref = deposit_inputs(fx)
    assert m["lineWidthRange"] == [1, 1]                    # the captured GL clamps flowWidth to 1
    got, fragments, cov = oracle.flow_deposit(fx["current"], fx["previous"], base, m["time"], view_size=m["viewSize"],
                                              speedLimit=m["speedLimit"], coverage=True)
    touched_ref = np.zeros(cov.size, bool)
    touched_ref[fx["idx"]] = True
    assert ((cov.ravel() > 0) == touched_ref).all(), "coverage differs in %d texels" % ((cov.ravel() > 0) != touched_ref).sum()
    assert fragments >= len(fx["idx"]) > 0
    assert deposit_close(got, ref, m["time"]).all()
    untouched = cov == 0
    assert (got[untouched].view(np.uint32) == base[untouched].view(np.uint32)).all()
    if "overlap" in path or "long" in path:
        assert cov.max() >= 3                               # several lines blended into one texel, in stream order


def test_deposit_pairing_quirk_and_inert(oracle):
    """Half of the vertex pairs read `current` twice (zero-length, nothing drawn), the last row pairs with
    `previous` again; pairs with an inert vertex draw nothing (documented deviation)."""
    n, v = 16, 256
    cell = v // n
    cur = np.zeros((n, n, 4), np.float32)
    for y in range(n):
        for x in range(n):
            cur[y, x] = [(x * cell + 6.3) / v * 2 - 1, (y * cell + 6.2) / v * 2 - 1, 0.004, 0.003]
    prev = cur.copy()
    prev[..., :2] += np.float32(2.5 / v * 2)
    _, _, cov = oracle.flow_deposit(cur, prev, np.zeros((v, v, 4), np.float32), 100.0, coverage=True)
    rows = sorted({int(y) // cell for y in np.nonzero(cov.any(1))[0]})
    assert rows == list(range(0, 8)) + [15]
    cur[3, 5] = [-1e6, -1e6, 0, 0]                          # dies this frame: no streak
    _, _, cov2 = oracle.flow_deposit(cur, prev, np.zeros((v, v, 4), np.float32), 100.0, coverage=True)
    assert cov2[3 * cell:4 * cell, 5 * cell:6 * cell].sum() == 0 and cov2.sum() < cov.sum()


def loop_reference(fx):
    m = fx["meta"]
    fw, fh = m["viewRes"]
    flow = np.zeros((fh * fw, 4), np.float32)
    flow[fx["flow_idx"]] = fx["flow_val"]
    return m, flow.reshape(fh, fw, 4)


def loop_close(states, flow, fx):
    """Tolerance of the closed loop against the reference's own K frames: the deposit's value tolerance feeds back
    into the next step through the flow tap, so states may drift by a few ulp (<= 2.5e-7 on positions in [-1, 1])
    and the final field carries K deposits' worth of the per-deposit tolerance; coverage stays identical."""
    m, ref_flow = loop_reference(fx)
    for k, s in enumerate(states):
        assert np.abs(s.astype(np.float64) - fx["out"][k]).max() <= 2.5e-7, "frame %d" % k
    assert ((flow != 0).any(-1) == (ref_flow != 0).any(-1)).all()
    d = np.abs(flow.astype(np.float64) - ref_flow)
    t = m["times"][-1]
    assert d[..., 0].max() <= 2e-7 and d[..., 1].max() <= 2e-7 and d[..., 2].max() <= 4e-6 * t and d[..., 3].max() <= 4e-5


def test_closed_loop_against_reference_frames(oracle):
    """K = 6 frames of the reference's own step() + draw(), restated: step, deposit, step on the deposited field ..."""
    import os
    from helpers import GOLDEN
    fx = load(os.path.join(GOLDEN, "loop_frames_64.npz"))
    m = fx["meta"]
    n = m["N"]
    fw, fh = m["viewRes"]
    cur, prev, flow = fx["state"], fx["state"], np.zeros((fh, fw, 4), np.float32)
    states = []
    for k in range(m["frames"]):
        u = oracle.logic_uniforms(n, n, m["times"][k], m["dts"][k], view_size=m["viewSize"], **oracle.DEFAULT_STATE)
        prev, cur = cur, oracle.logic_step(u, cur, flow)
        flow, _ = oracle.flow_deposit(cur, prev, flow, m["times"][k], view_size=m["viewSize"],
                                      speedLimit=oracle.DEFAULT_STATE["speedLimit"])
        states.append(cur)
    loop_close(states, flow, fx)
    assert (states[0].view(np.uint32) == fx["out"][0].view(np.uint32)).all()      # frame 0 has no wake yet: bit-exact
