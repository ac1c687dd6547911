"""GPU: slab decomposition through the C-ABI slab entry points.  Several slabs share the one
GPU of the test box (in-process hand-off, and two processes over gloo with host staging); the
result must equal the single-domain oracle bit for bit.  The RCCL transport itself needs more
than one GPU and is exercised by bench.py --gpus N on the multi-GPU node."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.slab_common import DT, build_local_group, single_domain_oracle, slab_scene

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def hip_backend(p, cols, mass, lo, hi, ghost_layers=2):
    from fluidsimulation_amd.slab import HipSlabBackend
    q = p.copy()
    q.capacity = max(int(p.capacity), 4096)
    return HipSlabBackend(q, cols, mass, lo, hi, device=0, max_records=int(q.capacity),
                          ghost_layers=ghost_layers)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_hip_slabs_equal_single_domain(world):
    p, pos, mass, cols, vel = slab_scene()
    steps = 6
    group, _ = build_local_group(hip_backend, p, pos, mass, cols, vel, world)
    for _ in range(steps):
        group.step(DT)
    for b in group.backends:
        b.sync()
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(ids, np.arange(pos.shape[0]))
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


@pytest.mark.parametrize("layers", [1, 2])
def test_hip_slabs_with_viscosity(layers):
    p, pos, mass, cols, vel = slab_scene(nx=16, iterations=2, xsph_c=0.01, vorticity_eps=2e-6, vel_scale=2.0)
    make = lambda p, cols, mass, lo, hi: hip_backend(p, cols, mass, lo, hi, ghost_layers=layers)
    group, _ = build_local_group(make, p, pos, mass, cols, vel, 3)
    for _ in range(4):
        group.step(DT)
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, 4)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


@pytest.mark.parametrize("layers,viscous", [(4, False), (6, False), (4, True), (8, False)])
def test_hip_deep_ghost_layers_skip_halo_refreshes(layers, viscous):
    """communication-avoiding halo (slab.py): ghost positions are refreshed only when fewer than
    two valid ghost layers are left; results stay bit-identical with the single domain"""
    kw = dict(xsph_c=0.01, vorticity_eps=2e-6, vel_scale=2.0) if viscous else {}
    its = 4 if layers == 8 else 3
    p, pos, mass, cols, vel = slab_scene(nx=64 if layers == 8 else 48, ny=5, nz=5, iterations=its, **kw)
    make = lambda p, cols, mass, lo, hi: hip_backend(p, cols, mass, lo, hi, ghost_layers=layers)
    group, _ = build_local_group(make, p, pos, mass, cols, vel, 2, ghost_layers=layers)
    calls = []
    halo = group._halo
    group._halo = lambda what: (calls.append(what), halo(what))[1]
    steps = 5
    for _ in range(steps):
        group.step(DT)
    expect = {4: 1, 6: 0, 8: 0}[layers] + (1 if viscous else 0)
    assert sum(1 for w in calls if w == 0) == expect * steps
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


def test_hip_slabs_larger_block_vs_single_gpu_ctx():
    """40x20x20 = 16 000 particles in 4 slabs against ONE ctx stepping the whole scene."""
    from fluidsimulation_amd import PbfContext
    p, pos, mass, cols, vel = slab_scene(nx=40, ny=20, nz=20, iterations=4, vel_scale=1.0)
    group, _ = build_local_group(hip_backend, p, pos, mass, cols, vel, 4)
    g = PbfContext(p, device=0)
    g.upload(pos, vel=vel, uniform_mass=mass)
    g.set_colliders(cols)
    for _ in range(5):
        group.step(DT)
        g.step(DT)
    g.sync()
    ids, gpos, gvel = group.export()
    spos, svel, _ = g.download()
    assert np.array_equal(gpos, spos) and np.array_equal(gvel, svel)


def test_rank_local_grids_do_not_change_results():
    """every slab on its own narrow wrap-around grid (what bench.py --gpus N runs)"""
    from fluidsimulation_amd.slab import (LocalExchange, SlabGroup, populate, rank_local_params,
                                          split_planes)
    p, pos, mass, cols, vel = slab_scene(nx=48, ny=8, nz=8, iterations=3, vel_scale=3.0)
    world, steps = 4, 5
    planes = split_planes(pos, p.grid_spacing, world)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    backends = []
    for r in range(world):
        q = rank_local_params(p, pos, planes, r, world)
        assert int(q.grid_dim[0]) < int(p.grid_dim[0])
        b = hip_backend(q, cols, mass, planes[r], planes[r + 1])
        populate(b, pos, mass, ids, planes[r], planes[r + 1], p.grid_spacing, vel=vel)
        backends.append(b)
    group = SlabGroup(backends, LocalExchange(world), int(p.solver_iterations))
    for _ in range(steps):
        group.step(DT)
    _, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


def test_rebalancing_between_substeps_keeps_the_result():
    p, pos, mass, cols, vel = slab_scene(nx=28)
    steps = 6
    group, planes = build_local_group(hip_backend, p, pos, mass, cols, vel, 3)
    seen = {tuple(planes)}
    for _ in range(steps):
        group.step(DT)
        planes = group.rebalance(planes, [b.owned_count() for b in group.backends], tolerance=0.0)
        seen.add(tuple(planes))
    assert len(seen) > 1
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


def test_leaver_beyond_the_ghost_planes_is_reported():
    from fluidsimulation_amd.binding import PbfError
    p, pos, mass, cols, vel = slab_scene(vel_scale=0.1)
    vel[np.argmin(pos[:, 0])] = (1000.0, 0.0, 0.0)   # 0.5 m = 12 cells in one sub-step
    group, _ = build_local_group(hip_backend, p, pos, mass, cols, vel, 2)
    with pytest.raises(PbfError, match="ghost_layers"):
        group.step(DT)


def test_id_indexed_readback_is_refused_in_slab_mode():
    from fluidsimulation_amd import PbfError
    p, pos, mass, cols, vel = slab_scene()
    b = hip_backend(p, cols, mass, -5, 5)
    with pytest.raises(PbfError):
        b.ctx.download()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import json
    import torch.distributed as dist
    from fluidsimulation_amd.slab import SlabSimulation
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, pos, mass, cols, vel = slab_scene()
    # exactly what bench.py --gpus N builds (rank-local grid, TorchExchange), plus re-balancing
    sim = SlabSimulation(p, pos, mass, cols, rank=rank, world=world, device=0, vel=vel, rebalance_every=2)
    for _ in range(steps):
        sim.step(DT)
    sim.sync()
    st = sim.stats()
    ids, gpos, gvel = sim.export()
    np.savez(Path(out_dir) / f"rank{rank}.npz", ids=ids, pos=gpos, vel=gvel)
    (Path(out_dir) / f"stats{rank}.json").write_text(json.dumps({"stats": st, "planes": sim.planes}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_gpu_over_gloo(tmp_path):
    """the torch.distributed exchange path with GPU-resident slabs (payloads staged through the
    host because both ranks share the single GPU of this box)"""
    import torch.multiprocessing as mp
    world, steps = 2, 4
    mp.spawn(_worker, args=(world, _free_port(), steps, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    ids = np.concatenate([z["ids"] for z in parts])
    order = np.argsort(ids)
    gpos = np.concatenate([z["pos"] for z in parts])[order]
    gvel = np.concatenate([z["vel"] for z in parts])[order]
    p, pos, mass, cols, vel = slab_scene()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)
    # the all-reduced health figures are the same on both ranks and cover the whole scene
    import json
    st = [json.loads((tmp_path / f"stats{r}.json").read_text()) for r in range(world)]
    assert st[0] == st[1]
    assert st[0]["stats"]["n"] == pos.shape[0] and st[0]["stats"]["nonfinite"] == 0
    np.testing.assert_allclose(st[0]["stats"]["com"], opos.astype(np.float64).mean(axis=0), rtol=1e-6)
