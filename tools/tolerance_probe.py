#!/usr/bin/env python3
"""Measurement for DESIGN.md section 3: a library built by tools/build_fp32_probe.py (pair geometry, W, grad W
and s_corr in fp32, hardware square root) against the CPU oracle, north_star's single-step bar
(1e-5 relative on positions AND velocities) and the 200-step statistics bar, plus its speed.
    python tools/tolerance_probe.py ab/libfp32.so
"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import binding as B
if len(sys.argv) > 1:
    B._lib = B.load_library(sys.argv[1])
from fluidsimulation_amd import PbfContext, scenes
from oracle import pyoracle
DT = 0.0005


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


for name, (p, pos, mass, cols), steps in (("C1 10k", scenes.c1_lattice(10, 100, 10, iterations=2), 1),
                                          ("C2 64k", scenes.c2_cube(), 1),
                                          ("C3 1M", scenes.c3_block(), 1)):
    g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
    o = pyoracle.Oracle(p, threads=16); o.upload(pos, uniform_mass=mass); o.set_colliders(cols)
    for _ in range(steps):
        g.step(DT); o.step(DT)
    g.sync()
    gp, gv, _ = g.download(); op, ov, _, _ = o.download()
    dp = np.abs(gp.astype(np.float64) - op)
    flipped = int((gp != op).any(axis=1).sum())
    print(f"{name}: 1 sub-step  rel_inf pos {rel(gp, op):.3e}  vel {rel(gv, ov):.3e}  (max|v| {np.abs(ov).max():.4f}, "
          f"max|dpos| {dp.max():.3e}, particles with a differing position {flipped} of {len(gp)})", flush=True)
p, pos, mass, cols = scenes.c1_lattice(10, 100, 10, iterations=2)
g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
o = pyoracle.Oracle(p, threads=16); o.upload(pos, uniform_mass=mass); o.set_colliders(cols)
for _ in range(200):
    g.step(DT); o.step(DT)
g.sync(); a, b = g.stats(), o.stats()
print(f"C1 200 steps: mean|C| {a.mean_abs_C:.6f} vs {b.mean_abs_C:.6f}  max|C| {a.max_abs_C:.5f} vs {b.max_abs_C:.5f}  "
      f"com_y {a.com[1]:.5f} vs {b.com[1]:.5f}")
p, pos, mass, cols = scenes.c3_block()
g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
for _ in range(20):
    g.step(DT)
g.sync(); t0 = time.perf_counter()
for _ in range(100):
    g.step(DT)
g.sync(); print(f"C3 speed: {(time.perf_counter() - t0) * 10:.4f} ms/sub-step")
g.profile_enable(True); g.profile_reset()
for _ in range(20):
    g.step(DT)
g.sync(); print({k: round(v[0] / v[1] * 1e3, 1) for k, v in g.profile_read().items() if v[1]})
