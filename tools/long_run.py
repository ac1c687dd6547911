#!/usr/bin/env python3
"""Health check over a long run: C3 (1 M particles) for N sub-steps, stats every 500."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
p, pos, mass, cols = scenes.c3_block()
g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
t0 = time.perf_counter()
for s in range(1, steps + 1):
    g.step(0.0005)
    if s % 500 == 0:
        g.sync()
        st = g.stats()
        print(f"step {s}: {(time.perf_counter()-t0)/s*1e3:.3f} ms/step  max_nbr {st.max_neighbor_count} mean_nbr {st.mean_neighbor_count:.2f} "
              f"mean|C| {st.mean_abs_C:.4f} max|C| {st.max_abs_C:.3f} com {st.com[0]:.3f} {st.com[1]:.3f} vmax {st.max_speed:.2f} nonfinite {st.nonfinite}", flush=True)
