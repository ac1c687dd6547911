#!/usr/bin/env python3
"""Overhead of the slab pipeline without a network: W slabs of one scene on ONE GPU, exchanged
in-process, against the plain single-ctx step.  (The RCCL transport itself needs >1 GPU.)"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from fluidsimulation_amd import PbfContext, scenes
from fluidsimulation_amd.slab import (HipSlabBackend, LocalExchange, SlabGroup, populate,
                                      rank_local_params, split_planes)
DT = 0.0005
block = (100, 100, 100)
steps, warm = 50, 10
worlds = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4]   # 0 = the plain single-ctx step
p, pos, mass, cols = scenes.dam_break(*block, iterations=4)
base = float("nan")
if 0 in worlds:
    g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
    for _ in range(warm): g.step(DT)
    g.sync(); t0 = time.perf_counter()
    for _ in range(steps): g.step(DT)
    g.sync(); base = (time.perf_counter() - t0) / steps
    print(f"single ctx: {base*1e3:.3f} ms/step")
    del g
for world in [w for w in worlds if w > 0]:
    planes = split_planes(pos, p.grid_spacing, world)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    bs = []
    for r in range(world):
        q = rank_local_params(p, pos, planes, r, world)
        b = HipSlabBackend(q, cols, mass, planes[r], planes[r + 1], device=0)
        populate(b, pos, mass, ids, planes[r], planes[r + 1], p.grid_spacing)
        bs.append(b)
    grp = SlabGroup(bs, LocalExchange(world), 4)
    for _ in range(warm): grp.step(DT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): grp.step(DT)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    print(f"{world} slab(s) on one GPU: {t*1e3:.3f} ms/step  (x{t/base:.2f} of single ctx)")
    del grp, bs
