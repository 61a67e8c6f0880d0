"""Shared helpers for the parity tests."""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(kind):
    """fixtures of one kind; the row-band fixture of the 4096^2 run has its own tests"""
    return sorted(p for p in glob.glob(os.path.join(GOLDEN, kind + "_*.npz")) if not p.endswith("_bands.npz"))


def load(path):
    d = np.load(path)
    out = {k: d[k] for k in d.files if k != "uniforms"}
    out["meta"] = json.loads(str(d["uniforms"]))
    out["name"] = os.path.basename(path)[:-4]
    return out


def bits_equal(a, b):
    """Bit equality of fp32 arrays, with any-NaN == any-NaN (NaN payloads are not pinned)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def state_overrides(meta):
    """Uniform values the reference actually used (Tendrils.state after overrides)."""
    return {k: v for k, v in meta["state"].items()}


def synth_frame(w, h, seed, shift8=(0, 0)):
    """Deterministic smooth RGBA8 frame (integer arithmetic only, so it is reproducible on any
    host): a seeded low-resolution grid, bilinearly upsampled.  shift8 = translation in 1/8 px."""
    rng = np.random.default_rng(seed)
    cell = 8
    gw, gh = w // cell + 4, h // cell + 4
    grid = rng.integers(0, 256, (gh, gw, 3), dtype=np.int64)
    ys = (np.arange(h, dtype=np.int64) * 8 + shift8[1] + 8 * cell)[:, None]
    xs = (np.arange(w, dtype=np.int64) * 8 + shift8[0] + 8 * cell)[None, :]
    gy, fy = ys // (8 * cell), ys % (8 * cell)
    gx, fx = xs // (8 * cell), xs % (8 * cell)
    q = 8 * cell
    out = np.empty((h, w, 4), np.uint8)
    for c in range(3):
        g = grid[..., c]
        top = g[gy, gx] * (q - fx) + g[gy, gx + 1] * fx
        bot = g[gy + 1, gx] * (q - fx) + g[gy + 1, gx + 1] * fx
        out[..., c] = ((top * (q - fy) + bot * fy) // (q * q)).astype(np.uint8)
    out[..., 3] = 255
    return out


def of_inputs(meta):
    """Inputs of an optical-flow fixture, regenerated from its seeds: (last frame, view frame,
    destination flow contents before the blended pass)."""
    fw, fh = meta["frame"]
    ow, oh = meta["out"]
    seed = meta["seed"]
    f0 = synth_frame(fw, fh, seed)
    f1 = synth_frame(fw, fh, seed, shift8=tuple(meta["shift8"]))
    rng = np.random.default_rng(seed)
    dst = np.zeros((oh, ow, 4), np.float32)
    dst[..., :2] = rng.uniform(-.01, .01, (oh, ow, 2))
    dst[..., 2] = meta["uniforms"]["time"] - 60.0
    dst[..., 3] = rng.uniform(0, 1, (oh, ow))
    return f0, f1, dst


def of_expected(fx, full):
    """Select the rows a fixture stores from a full [H, W, 4] result."""
    bands = fx["meta"].get("bands")
    if not bands:
        return full
    return np.concatenate([full[a:b] for a, b in bands])


# ---- packed state (TH_STATE_F16) - the build-defined encoding of include/tendrils_hip.h, mirrored in numpy ----
def pack_state(st):
    """[..., 4] f32 texels -> [..., 2] uint32 words (SNORM16 position over [-2,2) | fp16 velocity)."""
    st = np.asarray(st, np.float32)
    x, y = st[..., 0], st[..., 1]
    inert = (x == np.float32(-1e6)) & (y == np.float32(-1e6))
    nan = np.isnan(x) | np.isnan(y)
    with np.errstate(invalid="ignore"):
        qx = np.rint(np.clip(x * np.float32(16384), -32767, 32767)).astype(np.float32)
        qy = np.rint(np.clip(y * np.float32(16384), -32767, 32767)).astype(np.float32)
    qx = np.where(nan, 0, qx).astype(np.int32)
    qy = np.where(nan, 0, qy).astype(np.int32)
    qx = np.where(inert | nan, -32768, qx)
    qy = np.where(inert, -32768, np.where(nan, 0, qy))
    w0 = (qx.astype(np.uint32) & 0xffff) | ((qy.astype(np.uint32) & 0xffff) << 16)
    with np.errstate(over="ignore"):
        h = st[..., 2:].astype(np.float16).view(np.uint16).astype(np.uint32)
    w1 = h[..., 0] | (h[..., 1] << 16)
    return np.stack([w0, w1], -1).astype(np.uint32)


def unpack_state(w):
    """inverse of pack_state (exact)."""
    w = np.asarray(w, np.uint32)
    xs = (w[..., 0] & 0xffff).astype(np.uint16).view(np.int16).astype(np.int32)
    ys = (w[..., 0] >> 16).astype(np.uint16).view(np.int16).astype(np.int32)
    out = np.empty(w.shape[:-1] + (4,), np.float32)
    out[..., 0] = xs.astype(np.float32) * np.float32(2.0 ** -14)
    out[..., 1] = ys.astype(np.float32) * np.float32(2.0 ** -14)
    inert = (xs == -32768) & (ys == -32768)
    nan = (xs == -32768) & ~inert
    out[inert, 0] = out[inert, 1] = np.float32(-1e6)
    out[nan, 0] = out[nan, 1] = np.nan
    out[..., 2] = (w[..., 1] & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
    out[..., 3] = (w[..., 1] >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    return out


def hashed_state(n, seed, inert_mod=0, rows=None):
    """Deterministic integer-hash state (same generator as oracle/harness.js `stateGen`): every value has
    a 24-bit mantissa, so JS doubles and numpy agree exactly.  rows = (y0, y1) returns only that band."""
    y0, y1 = rows if rows else (0, n)
    q = (np.arange(y0 * n, y1 * n, dtype=np.uint64))

    def mix(v):
        v = v & 0xffffffff
        v = ((v ^ (v >> 16)) * 0x7feb352d) & 0xffffffff
        v = ((v ^ (v >> 15)) * 0x846ca68b) & 0xffffffff
        return (v ^ (v >> 16)) & 0xffffffff

    out = np.empty((y1 - y0, n, 4), np.float32)
    cols = []
    for c in range(4):
        hv = mix(q * 4 + c + seed).astype(np.int64)
        val = ((hv >> 8) - 8388608).astype(np.float64)
        cols.append(val / (8388608.0 if c < 2 else 838860800.0))
    st = np.stack(cols, -1).astype(np.float32).reshape(y1 - y0, n, 4)
    if inert_mod:
        b = mix(q * 4 + seed).astype(np.int64)
        st[(b % inert_mod == 0).reshape(y1 - y0, n)] = [-1e6, -1e6, 0, 0]
    out[:] = st
    return out


def _mix32(v):
    v = v & 0xffffffff
    v = ((v ^ (v >> 16)) * 0x7feb352d) & 0xffffffff
    v = ((v ^ (v >> 15)) * 0x846ca68b) & 0xffffffff
    return (v ^ (v >> 16)) & 0xffffffff


def curl_flow(w, h, seed, time, cells=12, mag=0.01, max_age=150.0):
    """Config C2's "curl-noise flow field" (the reference has none: SURVEY.md 8d defines it): a divergence-free
    field F = (dpsi/dy, -dpsi/dx) of a smooth seeded potential, written in the reference's flow format
    (Fx, Fy, deposit time, alpha).  psi = smoothstep-interpolated integer-hash lattice, evaluated on the texel
    corner grid in float64 with + - * only, so every machine regenerates the same bits; the discrete curl
    (differences of corner values) is exactly divergence-free."""
    lat = (_mix32(np.arange((cells + 2) * (cells + 2), dtype=np.uint64) + np.uint64(seed)) >> 8).astype(np.float64)
    lat = (lat / 8388608.0 - 1.0).reshape(cells + 2, cells + 2)

    def axis(n):
        g = np.arange(n + 1, dtype=np.float64) * cells / n         # corner k at k*cells/n, in [0, cells]
        c = np.minimum(np.floor(g), cells - 1)
        f = g - c
        return c.astype(np.int64), f * f * (3.0 - 2.0 * f)

    cx, sx = axis(w)
    cy, sy = axis(h)
    a = lat[cy][:, cx]; b = lat[cy][:, cx + 1]; c = lat[cy + 1][:, cx]; d = lat[cy + 1][:, cx + 1]
    top = a + (b - a) * sx[None, :]
    bot = c + (d - c) * sx[None, :]
    psi = top + (bot - top) * sy[:, None]                           # (h+1, w+1) corner values
    scale = mag * min(w, h) / (2.0 * cells)
    fx = (psi[1:, :-1] + psi[1:, 1:] - psi[:-1, :-1] - psi[:-1, 1:]) * (0.5 * scale)     # dpsi/dy at the texel centre
    fy = (psi[:-1, :-1] + psi[1:, :-1] - psi[:-1, 1:] - psi[1:, 1:]) * (0.5 * scale)     # -dpsi/dx
    age = (_mix32(np.arange(w * h, dtype=np.uint64) + np.uint64(seed * 7 + 1)) >> 8).astype(np.float64)
    age = (age / 16777216.0 * max_age).reshape(h, w)
    fl = np.empty((h, w, 4), np.float32)
    fl[..., 0] = fx
    fl[..., 1] = fy
    fl[..., 2] = (np.float64(time) - age)
    fl[..., 3] = 1.0
    return fl


def band_fixture_flow(fx):
    """Flow field of a *_bands fixture: stored, or regenerated from meta["flowGen"]."""
    g = fx["meta"].get("flowGen")
    if g is None:
        return fx["flow"]
    assert g["kind"] == "curl"
    return curl_flow(g["w"], g["h"], g["seed"], g["time"])


def deposit_hashed_inputs(n, seed, pos_range, step, inert_mod):
    """(current, previous) state textures of the large deposit fixture, from integer hashes (hashed_state) and
    float32 operations only, so that the fixture need not store them."""
    a = hashed_state(n, seed, inert_mod)                 # pos in [-1, 1), vel in [-.01, .01), some inert
    b = hashed_state(n, seed + 7919, 0)
    prev = a.copy()
    live = ~((a[..., 0] == -1e6) & (a[..., 1] == -1e6))
    prev[..., :2] = np.where(live[..., None], a[..., :2] * np.float32(pos_range), a[..., :2])
    cur = prev.copy()
    cur[..., :2] = np.where(live[..., None], prev[..., :2] + b[..., :2] * np.float32(step), prev[..., :2])
    cur[..., 2:] = np.where(live[..., None], b[..., 2:], prev[..., 2:])
    return cur.astype(np.float32), prev.astype(np.float32)


def buffers_fixture(path):
    """A Tendrils.buffers script captured on the reference (oracle/gen_fixtures.py:gen_buffers): meta (ops, state, ...), the
    two state textures, and the images its 'read' ops took, in order."""
    d = np.load(path)
    m = json.loads(str(d["uniforms"]))
    fw, fh = m["viewRes"]
    images = []
    for k in range(m["reads"]):
        img = np.zeros((fh * fw, 4), np.uint8)
        img[d["idx%d" % k]] = d["val%d" % k]
        images.append(img.reshape(fh, fw, 4))
    return m, d["current"], d["previous"], images


def blend_rgba8_over(src, dst):
    """A full-screen quad textured with the RGBA8 image `src` (texel for texel: copy.frag) - or of one colour, `src` a
    4-vector of floats - through SRC_ALPHA / ONE_MINUS_SRC_ALPHA into the RGBA8 image `dst`, in the view pass's own
    arithmetic (th_raster.hpp: dep_blend_rgba8 - fp32, a*b + c as two rounded operations, round half up)."""
    src = np.asarray(src)
    c = (src.astype(np.float32) / np.float32(255.0)) if src.dtype == np.uint8 else np.clip(src.astype(np.float32), 0, 1)
    sa = c[..., 3:4]
    da = np.float32(1.0) - sa
    o = c * sa + (dst.astype(np.float32) * np.float32(1.0 / 255.0)) * da
    return (np.clip(o, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
