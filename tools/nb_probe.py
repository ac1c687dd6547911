#!/usr/bin/env python3
"""Times the grid build alone (predict + sort + neighbour table; no sweep consumes the rows)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes
p, pos, mass, cols = scenes.dam_break(100, 100, 100, iterations=4)
g = PbfContext(p, 0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
for _ in range(5):
    g.predict(0.0); g.build_grid()
g.sync(); g.profile_enable(True); g.profile_reset()
for _ in range(40):
    g.predict(0.0); g.build_grid()
g.sync(); prof = g.profile_read()
print({k: round(ms / max(c, 1) * 1e3, 1) for k, (ms, c) in prof.items() if c})
