#!/usr/bin/env python3
"""Probe: does replaying the sub-step as a HIP graph help launch-bound (small) scenes?
Captures 6 sub-steps (the period of the ctx's buffer rotation) on a torch stream and replays."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from fluidsimulation_amd import PbfContext, scenes

DT = 0.0005
for dims in [(40, 40, 40), (64, 64, 32), (100, 100, 100)]:
    p, pos, mass, cols = scenes.dam_break(*dims, iterations=4)
    ctx = PbfContext(p, 0)
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    ctx.upload(pos, uniform_mass=mass)
    ctx.set_colliders(cols)
    with torch.cuda.stream(st):
        for _ in range(12):
            ctx.step(DT)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(120):
            ctx.step(DT)
        st.synchronize()
        eager = (time.perf_counter() - t0) / 120
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(6):
                ctx.step(DT)
        for _ in range(3):
            g.replay()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        st.synchronize()
        graph = (time.perf_counter() - t0) / 120
    print(f"{dims}: eager {eager*1e3:.4f} ms/sub-step, graph {graph*1e3:.4f} ms/sub-step", flush=True)
    ctx.close() if hasattr(ctx, "close") else None
