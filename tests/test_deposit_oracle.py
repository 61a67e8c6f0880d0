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
