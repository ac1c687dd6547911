#!/usr/bin/env python3
"""Addressing at the capacity limit: a ctx of 2^28 slots (the largest pbf_create accepts) holding a SMALL
scene -- the neighbour rows are cap * 4 bytes apart, so the sweeps' buffer loads run with scalar offsets of up
to 3 GiB and the builder's stores reach row k at k GiB -- against the CPU oracle, bit for bit, and against a
ctx of the scene's own size.   python tools/check_big_capacity.py [log2 capacity = 28] [max_neighbors = 48]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes
from oracle import pyoracle
import torch

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
p, pos, mass, cols = scenes.dam_break(24, 20, 16, iterations=4, max_neighbors=K)
rng = np.random.default_rng(5)
pos = np.ascontiguousarray(pos[rng.permutation(pos.shape[0])])
o = pyoracle.Oracle(p)
o.upload(pos, uniform_mass=mass); o.set_colliders(cols)
small = PbfContext(p, device=0)
small.upload(pos, uniform_mass=mass); small.set_colliders(cols)
p.capacity = 1 << lg
free0 = torch.cuda.mem_get_info(0)[0]
big = PbfContext(p, device=0)
print(f"capacity 2^{lg}: ctx holds {(free0 - torch.cuda.mem_get_info(0)[0]) / 2**30:.1f} GiB", flush=True)
big.upload(pos, uniform_mass=mass); big.set_colliders(cols)
for s in range(6):
    for c in (o, small, big):
        c.step(0.0005)
bp, bv, _ = big.download()
sp, sv, _ = small.download()
op, ov, _, _ = o.download()
ok = bp.tobytes() == op.tobytes() and bv.tobytes() == ov.tobytes() and sp.tobytes() == bp.tobytes()
st = big.stats()
print(f"{pos.shape[0]} particles, 6 sub-steps, mean neighbours {st.mean_neighbor_count:.2f}: identical with the oracle and the small ctx: {ok}")
sys.exit(0 if ok else 1)
