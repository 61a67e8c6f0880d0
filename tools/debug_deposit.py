import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from helpers import bits_equal, golden, load
from test_deposit_oracle import deposit_inputs
from test_gpu_deposit import gpu_deposit
for path in golden("deposit"):
    fx = load(path); m, base, ref = deposit_inputs(fx)
    got, frags = gpu_deposit(fx["current"], fx["previous"], base, m["time"], m["viewRes"], m["viewSize"], m["speedLimit"])
    want, n, cov = O.flow_deposit(fx["current"], fx["previous"], base, m["time"], view_size=m["viewSize"], speedLimit=m["speedLimit"], coverage=True)
    eq = bits_equal(got, want).all(-1)
    bad = np.argwhere(~eq)
    print(os.path.basename(path), "particles", fx["current"].shape, "view", m["viewRes"], "frags", frags, n, "bad texels", len(bad), "of touched", int((cov > 0).sum()), "max cov", int(cov.max()))
    for y, x in bad[:6]:
        print("   texel", x, y, "cov", int(cov.reshape(got.shape[:2])[y, x]), "got", got[y, x], "want", want[y, x])
