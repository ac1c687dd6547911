#!/usr/bin/env python3
"""Per-step kernel timeline (durations and idle gaps) from a rocprofv3 --kernel-trace rocpd
database: tools/trace_steps.py <results.db> [steps]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = db.execute("select name, start, end from kernels order by start").fetchall()


def short(n):
    n = n.replace("void ", "")
    i = n.find("k_")
    if i >= 0:
        n = n[i:]
    return n.split("(")[0][:48]


names = [short(r[0]) for r in rows]
pred = [i for i, n in enumerate(names) if n.startswith("k_predict")]
if len(pred) < 3:   # a slab run: the split pass is the sub-step's first kernel
    pred = [i for i, n in enumerate(names) if n.startswith("k_slab_split")]
i0, i1 = pred[-steps - 1], pred[-1]
print(f"per step on the GPU timeline: {(rows[i1][1] - rows[i0][1]) / steps / 1e3:.1f} us")
agg = collections.OrderedDict()
gaps = 0
for k in range(i0, i1):
    d = agg.setdefault(names[k], [0, 0])
    d[0] += 1
    d[1] += rows[k][2] - rows[k][1]
    gaps += max(0, rows[k + 1][1] - rows[k][2])
tot = 0
for n, (c, t) in agg.items():
    print(f"{n:48s} {c / steps:5.1f}/step {t / steps / 1e3:8.1f} us/step")
    tot += t
print(f"busy {tot / steps / 1e3:.1f} us/step, idle {gaps / steps / 1e3:.1f} us/step")
print("-- last step")
for k in range(pred[-2], pred[-1] + 1):
    print(f"{names[k]:40s} dur {(rows[k][2] - rows[k][1]) / 1e3:7.1f}  gap_before {(rows[k][1] - rows[k - 1][2]) / 1e3:7.1f}")
