"""Respawn passes, CPU side: the oracle against the reference captures.  The GLSL hash
amplifies the platform's sin(), so parity here is statistical: moments, hard
bounds and the share of texels that agree."""
import numpy as np
import pytest

from helpers import bits_equal, golden, load


def oracle_spawn(oracle, fx):
    m = fx["meta"]
    if m["kind"] == "spawn_ball":
        return oracle.spawn_ball(m["N"], m["N"], **m["uniforms"])
    un = m["uniforms"]
    u = oracle.spawn_sample_uniforms(m["N"], m["N"], un["time"], m["samples"], m["apply"],
                                     spawnSize=un["spawnSize"], jitter=un["jitter"], speed=un["speed"],
                                     bias=un["bias"], flowDecay=un.get("flowDecay", 0.0),
                                     spawnMatrix=un["spawnMatrix"])
    if m["kind"] == "spawn_direct":
        return oracle.spawn_direct(u, fx["data"])
    return oracle.spawn_sample(u, fx["state"], fx["data"])


@pytest.mark.parametrize("path", golden("spawn"), ids=lambda p: p.split("/")[-1][:-4])
def test_spawn_statistics_match_reference(oracle, path):
    fx = load(path)
    got, ref = oracle_spawn(oracle, fx), fx["out"]
    if fx["meta"]["kind"] == "spawn_direct":
        # index.frag: no hash in the texel choice.  Velocities differ from the reference only through cos/sin
        # (|d| <= 2e-7 at |v| <= 0.3); positions are exact without jitter, within the jitter range with it
        # (uvToPos doubles it, spawnSize scales it: |d| <= 4*jitter*spawnSize)
        un = fx["meta"]["uniforms"]
        assert np.abs(got[..., 2:] - ref[..., 2:]).max() <= 2e-7
        bound = 4 * np.abs(np.array(un["jitter"]) * np.array(un["spawnSize"])) * (1 + 1e-5)
        d = np.abs(got[..., :2] - ref[..., :2])
        assert (d[..., 0] <= bound[0]).all() and (d[..., 1] <= bound[1]).all()
        if not any(un["jitter"]):
            assert (got[..., :2].view(np.uint32) == ref[..., :2].view(np.uint32)).all()
        return
    g, r = got.reshape(-1, 4).astype(np.float64), ref.reshape(-1, 4).astype(np.float64)
    for c in range(4):
        sd = max(r[:, c].std(), 1e-12)
        assert abs(g[:, c].mean() - r[:, c].mean()) <= 0.06 * sd + 1e-12, "component %d mean" % c
        assert abs(g[:, c].std() - r[:, c].std()) <= 0.03 * sd + 1e-12, "component %d spread" % c
    # a fair share of texels is untouched by the hash sensitivity
    assert (np.abs(got - ref) < 1e-3).all(-1).mean() > 0.15
    if fx["meta"]["kind"] == "spawn_ball":
        un = fx["meta"]["uniforms"]
        assert np.hypot(got[..., 0], got[..., 1]).max() <= un["radius"] * (1 + 1e-6)
        assert np.hypot(got[..., 2], got[..., 3]).max() <= un["speed"] * (1 + 1e-6)
        assert np.hypot(ref[..., 0], ref[..., 1]).max() <= un["radius"] * (1 + 1e-6)


def test_pinned_hash_is_the_rounded_true_sine(oracle):
    """to_random() = fract(fl32(sin(x)) * c) with the true sine: check against libm in fp64."""
    import math
    L = oracle.lib()
    rng = np.random.default_rng(0)
    f32 = np.float32
    for a, b in rng.uniform(-40, 60, (400, 2)).astype(np.float32):
        dt = f32(f32(a * f32(12.9898)) + f32(b * f32(78.233)))
        sn = f32(dt - f32(3.14) * np.floor(f32(dt / f32(3.14))))
        v = f32(f32(math.sin(float(sn))) * f32(43758.5453))
        assert L.to_random(float(a), float(b)) == f32(v - np.floor(v))


@pytest.mark.parametrize("path", golden("geometry"), ids=lambda p: p.split("/")[-1][:-4])
def test_geometry_triangles_bit_exact(oracle, path):
    """GeometrySpawner's triangle draw against the reference capture: coverage and (for the translucent case)
    the in-order blend of overlapping triangles, bit for bit."""
    fx = load(path)
    m = fx["meta"]
    got = oracle.triangles(fx["positions"], m["shape"], view_size=m["viewSize"], color=m["color"])
    assert (got.view(np.uint32) == fx["out"].view(np.uint32)).all()
    assert (fx["out"][..., 3] != 0).sum() > 1000
