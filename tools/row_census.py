#!/usr/bin/env python3
"""Particles per neighbour-row format (32-bit / relative 16-bit / window-coded 16-bit) after one
sub-step of a dam-break block:    python tools/row_census.py NX NY NZ [NX NY NZ ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
for k in range(0, len(a), 3):
    p, pos, mass, cols = scenes.dam_break(*a[k:k + 3], iterations=4)
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    g.step(0.0005)
    g.sync()
    st = g.stats()
    print(a[k:k + 3], "grid", list(p.grid_dim), "origin", list(p.grid_origin_cell), "rows abs32/rel16/win16",
          list(st.row_format_particles), "mean nbrs %.2f" % st.mean_neighbor_count, flush=True)
    g.close()
