#!/usr/bin/env python3
"""Per-phase device time of the C3 scene as it evolves: at rest (the BASELINE window) and after
N sub-steps of the collapsing dam break.   python tools/evolved_probe.py [steps ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes
marks = [int(a) for a in sys.argv[1:]] or [0, 1000, 2000, 4000]
p, pos, mass, cols = scenes.c3_block()
g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
done = 0
for m in marks:
    for _ in range(m - done):
        g.step(0.0005)
    done = max(done, m)
    g.sync()
    for _ in range(5):
        g.step(0.0005)
    g.sync(); t0 = time.perf_counter()
    for _ in range(50):
        g.step(0.0005)
    g.sync(); ms = (time.perf_counter() - t0) / 50 * 1e3
    g.profile_enable(True); g.profile_reset()
    for _ in range(20):
        g.step(0.0005)
    g.sync(); prof = g.profile_read(); g.profile_enable(False)
    done += 75
    st = g.stats()
    ph = {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]}
    print(f"after {m}: {ms:.4f} ms/step  mean_nbr {st.mean_neighbor_count:.2f} max_nbr {st.max_neighbor_count} "
          f"mean|C| {st.mean_abs_C:.4f} vmax {st.max_speed:.2f}  {ph}", flush=True)
