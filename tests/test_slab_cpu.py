"""CPU: the slab-decomposition protocol (fluidsimulation_amd/slab.py) on the oracle backend --
in-process with 2-4 slabs and over torch.distributed gloo with world_size 2 -- must reproduce
the single-domain oracle bit for bit (cell ids, neighbour order and therefore every sum agree)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.slab_common import DT, build_local_group, single_domain_oracle, slab_scene

ROOT = Path(__file__).resolve().parent.parent


def oracle_backend(p, cols, mass, lo, hi):
    from oracle.slab_backend import OracleSlabBackend
    return OracleSlabBackend(p, cols, mass, lo, hi)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_local_slabs_equal_single_domain(world):
    p, pos, mass, cols, vel = slab_scene()
    steps = 6
    group, planes = build_local_group(oracle_backend, p, pos, mass, cols, vel, world)
    start_owner = np.searchsorted(np.array(planes[1:-1]), np.floor(pos[:, 0] / np.float32(p.grid_spacing)), side="right")
    for _ in range(steps):
        group.step(DT)
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(ids, np.arange(pos.shape[0]))
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)
    # the test really migrated particles across slab planes
    end_owner = np.searchsorted(np.array(planes[1:-1]), np.floor(opos[:, 0] / np.float32(p.grid_spacing)), side="right")
    assert (start_owner != end_owner).sum() > 0


@pytest.mark.parametrize("layers", [1, 2])
def test_local_slabs_with_viscosity(layers):
    """one ghost layer: lambda / omega of the ghosts come from their owners; two layers: from
    the local sweeps -- both must equal the single-domain run"""
    from oracle.slab_backend import OracleSlabBackend
    p, pos, mass, cols, vel = slab_scene(nx=16, iterations=2, xsph_c=0.01, vorticity_eps=2e-6, vel_scale=2.0)
    make = lambda p, cols, mass, lo, hi: OracleSlabBackend(p, cols, mass, lo, hi, ghost_layers=layers)
    group, _ = build_local_group(make, p, pos, mass, cols, vel, 3)
    for _ in range(4):
        group.step(DT)
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, 4)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


@pytest.mark.parametrize("layers,viscous", [(4, False), (6, False), (4, True)])
def test_deep_ghost_layers_skip_halo_refreshes(layers, viscous):
    """communication-avoiding halo: with L ghost layers the ghost positions are refreshed only
    when fewer than two valid layers are left (3 iterations: L = 4 -> one refresh, L = 6 -> none);
    the result must still equal the single-domain run bit for bit"""
    from oracle.slab_backend import OracleSlabBackend
    kw = dict(xsph_c=0.01, vorticity_eps=2e-6, vel_scale=2.0) if viscous else {}
    p, pos, mass, cols, vel = slab_scene(nx=48, ny=5, nz=5, iterations=3, **kw)
    make = lambda p, cols, mass, lo, hi: OracleSlabBackend(p, cols, mass, lo, hi, ghost_layers=layers)
    group, planes = build_local_group(make, p, pos, mass, cols, vel, 2, ghost_layers=layers)
    calls = []
    halo = group._halo
    group._halo = lambda what: (calls.append(what), halo(what))[1]
    steps = 5
    for _ in range(steps):
        group.step(DT)
    expect = {4: 1, 6: 0}[layers] + (1 if viscous else 0)
    assert sum(1 for w in calls if w == 0) == expect * steps
    ids, gpos, gvel = group.export()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)


def test_shift_planes_moves_one_cell_towards_balance():
    from fluidsimulation_amd.slab import INT32_MAX, INT32_MIN, shift_planes
    planes = [INT32_MIN, 10, 20, INT32_MAX]
    assert shift_planes(planes, [100, 100, 100]) == planes
    assert shift_planes(planes, [200, 100, 100]) == [INT32_MIN, 9, 20, INT32_MAX]
    assert shift_planes(planes, [100, 100, 300]) == [INT32_MIN, 10, 21, INT32_MAX]
    assert shift_planes(planes, [100, 300, 100]) == [INT32_MIN, 11, 19, INT32_MAX]
    # a slab never gets thinner than the ghost planes need
    assert shift_planes([INT32_MIN, 10, 14, INT32_MAX], [100, 300, 100]) == [INT32_MIN, 10, 14, INT32_MAX]
    assert shift_planes([INT32_MIN, 10, 15, INT32_MAX], [100, 300, 100]) == [INT32_MIN, 10, 14, INT32_MAX]


def test_rebalancing_between_substeps_keeps_the_result():
    """planes move one cell per sub-step (tolerance 0 forces a move whenever counts differ);
    the slab run must still equal the single-domain run bit for bit"""
    p, pos, mass, cols, vel = slab_scene(nx=28)
    steps = 6
    group, planes = build_local_group(oracle_backend, p, pos, mass, cols, vel, 3)
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
    """the one-round exchange keeps leavers behind as ghosts; a particle that jumps further than
    the ghost planes in one sub-step must fail loudly, not silently lose neighbours"""
    p, pos, mass, cols, vel = slab_scene(vel_scale=0.1)
    vel[np.argmin(pos[:, 0])] = (1000.0, 0.0, 0.0)   # 0.5 m = 12 cells in one sub-step
    group, _ = build_local_group(oracle_backend, p, pos, mass, cols, vel, 2)
    with pytest.raises(RuntimeError, match="ghost_layers"):
        group.step(DT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from fluidsimulation_amd.slab import SlabGroup, TorchExchange, populate, split_planes
    from oracle.slab_backend import OracleSlabBackend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, pos, mass, cols, vel = slab_scene()
    planes = split_planes(pos, p.grid_spacing, world)
    b = OracleSlabBackend(p, cols, mass, planes[rank], planes[rank + 1])
    populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
             p.grid_spacing, vel=vel)
    g = SlabGroup([b], TorchExchange(rank, world), int(p.solver_iterations))
    for _ in range(steps):
        g.step(DT)
    ids, gpos, gvel = g.export()
    np.savez(Path(out_dir) / f"rank{rank}.npz", ids=ids, pos=gpos, vel=gvel)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_equals_single_domain(tmp_path):
    import torch.multiprocessing as mp
    world, steps = 2, 5
    port = _free_port()
    mp.spawn(_gloo_worker, args=(world, port, steps, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    ids = np.concatenate([z["ids"] for z in parts])
    order = np.argsort(ids)
    gpos = np.concatenate([z["pos"] for z in parts])[order]
    gvel = np.concatenate([z["vel"] for z in parts])[order]
    p, pos, mass, cols, vel = slab_scene()
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(ids[order], np.arange(pos.shape[0]))
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)
    assert all(z["ids"].size > 0 for z in parts)


class _FakeComm:
    """In-process stand-in for binding.RcclComm with NCCL's matching rule: the k-th send from a
    to b pairs with b's k-th receive from a.  Threads play the ranks."""

    def __init__(self, rank, queues):
        self.rank, self.q = rank, queues

    def batch(self, ops, stream=0):
        import ctypes
        for snd, peer, ptr, nbytes in ops:
            if snd and nbytes:
                self.q[(self.rank, peer)].put(ctypes.string_at(ptr, nbytes))
        for snd, peer, ptr, nbytes in ops:
            if not snd and nbytes:
                data = self.q[(peer, self.rank)].get(timeout=60)
                assert len(data) == nbytes, "message sizes disagree between the two ends"
                ctypes.memmove(ptr, data, nbytes)


@pytest.mark.parametrize("world,layers", [(2, 2), (3, 2), (3, 4)])
def test_p2p_exchange_protocol_over_a_fake_rccl(world, layers):
    """slab.P2PExchange (what runs over RCCL through the C ABI on the multi-GPU node) driven by
    `world` threads over a transport with NCCL's ordering semantics, oracle backend: must equal
    the single-domain run bit for bit"""
    import queue
    import threading
    from fluidsimulation_amd.slab import P2PExchange, SlabGroup, populate, split_planes
    from oracle.slab_backend import OracleSlabBackend
    p, pos, mass, cols, vel = slab_scene(nx=60 if layers == 4 else 24, ny=5, nz=5)
    steps = 5
    planes = split_planes(pos, p.grid_spacing, world, min_width=2 * layers)
    queues = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    out, errs = {}, []

    def work(rank):
        try:
            b = OracleSlabBackend(p, cols, mass, planes[rank], planes[rank + 1], threads=1, ghost_layers=layers)
            populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                     p.grid_spacing, vel=vel)
            g = SlabGroup([b], P2PExchange(_FakeComm(rank, queues), rank, world), int(p.solver_iterations))
            for _ in range(steps):
                g.step(DT)
            out[rank] = g.export()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    ids = np.concatenate([out[r][0] for r in range(world)])
    order = np.argsort(ids)
    gpos = np.concatenate([out[r][1] for r in range(world)])[order]
    gvel = np.concatenate([out[r][2] for r in range(world)])[order]
    opos, ovel = single_domain_oracle(p, pos, mass, cols, vel, steps)
    assert np.array_equal(ids[order], np.arange(pos.shape[0]))
    assert np.array_equal(gpos, opos) and np.array_equal(gvel, ovel)
    assert all(q.empty() for q in queues.values())
