#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary with HOST buffers: pbf_upload (positions + velocities in),
one sub-step, pbf_download (positions + velocities out) every sub-step -- the way a caller that
keeps its particles in host memory (the reference's ParticleSet vectors) would drive it.
Not the bench metric (state stays resident in HBM there); DESIGN.md quotes it."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes

DT = 0.0005
p, pos, mass, cols = scenes.dam_break(100, 100, 100, iterations=4)
ctx = PbfContext(p, 0)
ctx.upload(pos, uniform_mass=mass)
ctx.set_colliders(cols)
for _ in range(5):
    ctx.step(DT)
out = ctx.download()
x, v = out[0], out[1]
steps = 30
t0 = time.perf_counter()
for _ in range(steps):
    ctx.upload(x, vel=v, uniform_mass=mass)
    ctx.step(DT)
    out = ctx.download()
    x, v = out[0], out[1]
dt = (time.perf_counter() - t0) / steps
n = pos.shape[0]
print(f"host-buffer round trip every sub-step: {dt*1e3:.3f} ms/sub-step = {n/dt:.3e} particle-substeps/s "
      f"({n} particles; 24 MB in + 24 MB out per sub-step)")
