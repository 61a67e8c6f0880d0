"""The synthetic workload of bench.py: sizes, constants, seeded state / frames / flow.  N, FLOW_W, FLOW_H are
read through this module at run time (`--flow-size`, the tools' TH_N)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N = 4096                        # C3: particles per rank = N*N
FLOW_W, FLOW_H = 1920, 1080
BYTES_PER_PARTICLE_STEP = 32    # 16 B state read + 16 B written; 8 + 8 with packed state
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK = 256 * 4 * 2.4e9 / 2     # wave64 VALU instructions per second: 2 cycles each on a SIMD-32
MAX_FUSED = 32                  # th::kMaxFusedSteps
PREROLL_MS = 60.0

CONFIGS = {
    # name: (width, global height as a function of world, rows per rank, state, steps per group, scaling)
    "c3": dict(width=N, rows=lambda w: N, gheight=lambda w: N * w, state="f32", group=32, scaling="weak",
               label="C3: 4096x4096 %s state (16.8M particles) per GPU"),
    # the metric's own particles ("16M particles at 1/2/4/8 MI355X") read as ONE texture over the GPUs: a 4096 // N-row band each
    "c3_strong": dict(width=N, rows=lambda w: N // w, gheight=lambda w: N, state="f32", group=32, scaling="strong",
                      label="C3 strong: ONE 4096x4096 %s state (16.8M particles) row-sharded over the GPUs"),
    "c4": dict(width=8192, rows=lambda w: 8192 // w, gheight=lambda w: 8192, state="f32", group=16, scaling="strong",
               label="C4: 8192x8192 %s state (67.1M particles) row-sharded over the GPUs"),
    "c5": dict(width=16384, rows=lambda w: 16384 // w, gheight=lambda w: 16384, state="f16", group=16, scaling="strong",
               label="C5: 16384x16384 %s state (268M particles) row-sharded over the GPUs"),
}


def synth_rows(width, rows, seed):
    rng = np.random.default_rng(seed)
    st = np.empty((rows, width, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (rows, width, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (rows, width, 2))
    return st


def synth_state(rank):
    return synth_rows(N, N, 12345 + rank)


def synth_frames():
    """frame0 = seeded band-limited pattern, frame1 = frame0 translated by (1.5, 0.7) px."""
    yy, xx = np.mgrid[0:FLOW_H, 0:FLOW_W].astype(np.float64)

    def pattern(dx, dy):
        img = np.zeros((FLOW_H, FLOW_W, 3))
        r = np.random.default_rng(778)
        for _ in range(24):
            fx, fy = r.uniform(-0.08, 0.08, 2)
            ph = r.uniform(0, 2 * np.pi, 3)
            amp = r.uniform(0.2, 1.0)
            for c in range(3):
                img[..., c] += amp * np.sin((xx - dx) * fx + (yy - dy) * fy + ph[c])
        img = (img - img.min()) / (img.max() - img.min())
        out = np.empty((FLOW_H, FLOW_W, 4), np.uint8)
        out[..., :3] = np.clip(np.rint(img * 255), 0, 255).astype(np.uint8)
        out[..., 3] = 255
        return out
    return pattern(0.0, 0.0), pattern(1.5, 0.7)


def synth_flow(time_ms):
    """Divergence-free seeded field in reference flow format (Fx, Fy, t_deposit, alpha)."""
    yy, xx = np.mgrid[0:FLOW_H, 0:FLOW_W].astype(np.float32)
    r = np.random.default_rng(4242)
    psi_x = np.zeros((FLOW_H, FLOW_W), np.float32)
    psi_y = np.zeros((FLOW_H, FLOW_W), np.float32)
    for _ in range(12):
        fx, fy = r.uniform(-0.05, 0.05, 2).astype(np.float32)
        ph = np.float32(r.uniform(0, 2 * np.pi))
        a = np.float32(r.uniform(0.3, 1.0))
        c = a * np.cos(xx * fx + yy * fy + ph)
        psi_x += c * fy        # d(psi)/dy
        psi_y += -c * fx       # -d(psi)/dx
    s = np.float32(0.01) / max(np.abs(psi_x).max(), np.abs(psi_y).max())
    fl = np.empty((FLOW_H, FLOW_W, 4), np.float32)
    fl[..., 0] = psi_x * s
    fl[..., 1] = psi_y * s
    fl[..., 2] = time_ms
    fl[..., 3] = 1.0
    return fl


