import sys, time
sys.path.insert(0, '/root/repo')
from fluidsimulation_amd import PbfContext, scenes
DT = 0.0005
for kw in ({}, dict(xsph_c=0.01), dict(vorticity_eps=2e-6), dict(xsph_c=0.01, vorticity_eps=2e-6)):
    p, pos, mass, cols = scenes.dam_break(100, 100, 100, iterations=4, **kw)
    g = PbfContext(p, 0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
    for _ in range(10): g.step(DT)
    g.sync(); t0 = time.perf_counter()
    for _ in range(50): g.step(DT)
    g.sync(); t = (time.perf_counter() - t0) / 50
    g.profile_enable(True); g.profile_reset()
    for _ in range(20): g.step(DT)
    g.sync(); prof = g.profile_read(); g.profile_enable(False)
    print(kw, f"{t*1e3:.3f} ms/step", {k: (round(ms / max(c, 1) * 1e3, 1), c // 20) for k, (ms, c) in prof.items() if c})
