import sys, time, os
sys.argv = ["x"]
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import importlib.util
spec = importlib.util.spec_from_file_location("bss", "/root/repo/tools/bench_slab_self.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import torch, numpy as np
from fluidsimulation_amd import scenes, slab as S
DT = 0.0005
L = 2
p, pos, mass, cols = scenes.dam_break(100, 100, 100, iterations=4)
cx = S.cell_x(pos[:, 0], p.grid_spacing); lo, hi = int(cx.min()), int(cx.max()) + 41
q = S.rank_local_params(p, pos, [S.INT32_MIN, S.INT32_MAX], 0, 1)
b = S.HipSlabBackend(q, cols, mass, lo, hi, device=0, ghost_layers=L)
S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), lo, hi, p.grid_spacing)
ex = m.SelfExchange()
grp = S.SlabGroup([b], ex, 4)
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for nm in ["predict", "split_launch", "split_finish", "append_owned", "append_ghosts", "build", "lambda_pass", "delta_pass", "halo_pack", "halo_unpack", "finalize_stage"]:
    wrap(b, nm, nm)
wrap(ex, "counts", "x.counts"); wrap(ex, "run", "x.run")
for _ in range(10): grp.step(DT)
torch.cuda.synchronize(); T.clear(); t0 = time.perf_counter()
N = 50
for _ in range(N): grp.step(DT)
host = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"host {host/N*1e3:.3f} ms/step, wall {wall/N*1e3:.3f} ms/step")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"  {k:14s} {v/N*1e6:8.1f} us/step")
import torch.distributed as dist; dist.destroy_process_group()
