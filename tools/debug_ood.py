import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle')); sys.path.insert(0,os.path.join(ROOT,'tests'))
import oracle as O
import tendrils_amd as ta
from test_gpu_logic_parity import make_tendrils, seeded_case
from helpers import bits_equal
n=64
st, fl = seeded_case(n, 99)
st[0, 0, :2] = [3e6, 0.1]
st[0, 1, :2] = [1e30, -1e30]
st[0, 2, :2] = [np.inf, 0.0]
st[0, 3, :2] = [np.nan, 0.5]
st[0, 4, 2:] = [np.nan, 0.0]
st[0, 5, :2] = [-1e6, 0.25]
st[0, 6, :2] = [5e5, 5e5]
for overrides in ({}, {"noiseWeight": 0}):
    t = make_tendrils(n, (96, 54), (96, 54), overrides, ta.TH_MODE_EXACT)
    t.particles.upload_texels(st); t.flow.set_pixels(fl)
    t.timer.time = 4000.0; t.timer.tick(); t.step()
    got = t.particles.read(0)
    u = O.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize, **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
    want = O.logic_step(u, st, fl)
    bad=np.argwhere(~bits_equal(got,want).all(-1))
    print(overrides, 'bad lanes', bad.tolist())
    for y,x in bad[:8]:
        print('  in', st[y,x], 'got', got[y,x], 'want', want[y,x])
    t.dispose()
