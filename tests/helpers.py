"""Shared helpers for the parity tests."""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(kind):
    return sorted(glob.glob(os.path.join(GOLDEN, kind + "_*.npz")))


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
