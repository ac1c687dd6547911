#!/usr/bin/env python3
"""Long runs of one scene with the default neighbour-row formats against the same run restricted to
32-bit rows (pbf_set_row_formats): positions and velocities must stay bit-identical while the dam
break collapses and splashes -- every format switch, over-budget group and window code along the
way.    python tools/format_soak.py NX NY NZ STEPS [NX NY NZ STEPS ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes  # noqa: E402
from fluidsimulation_amd import binding as B  # noqa: E402

DT = 0.0005   # GLFWApp.cpp:393
a = [int(x) for x in sys.argv[1:]]
bad = 0
for k in range(0, len(a), 4):
    nx, ny, nz, steps = a[k:k + 4]
    p, pos, mass, cols = scenes.dam_break(nx, ny, nz, iterations=4)
    pos = np.ascontiguousarray(pos[np.random.default_rng(7).permutation(pos.shape[0])])
    res = []
    for mask in (B.PBF_ROWS_ALL, B.PBF_ROWS_ABS32):
        g = PbfContext(p, device=0)
        g.set_row_formats(mask)
        g.upload(pos, uniform_mass=mass)
        g.set_colliders(cols)
        census = np.zeros(3, np.int64)
        peak = 0
        t0 = time.perf_counter()
        for s in range(steps):
            g.step(DT)
            if s % max(1, steps // 8) == 0:
                st = g.stats()
                census = np.maximum(census, np.array(list(st.row_format_particles)))
                peak = max(peak, int(st.max_neighbor_count))
        g.sync()
        st = g.stats()
        res.append(g.download()[:2])
        print(f"{nx}x{ny}x{nz} mask {mask}: {steps} sub-steps in {time.perf_counter() - t0:.1f} s, mean nbrs "
              f"{st.mean_neighbor_count:.2f} max {st.max_neighbor_count} (peak at the checkpoints {peak}), most particles seen per format {census.tolist()}",
              flush=True)
        g.close()
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    print("   identical:", same, flush=True)
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
