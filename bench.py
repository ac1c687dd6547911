#!/usr/bin/env python3
"""bench.py -- particle-substeps/s of the PBF sub-step on MI355X (BASELINE.json metric).

A "step" is one PBF sub-step (predict -> collide -> grid/sort -> neighbour table ->
4 x {density+lambda, delta-p} -> velocity update) over one synthetic dam-break block that is
already resident in HBM.  N=1 runs config C3 (1M particles, 4 iterations).  Under
torch.distributed (N>1) every rank owns an x-slab of the block (see fluidsimulation_amd/slab.py).

Prints ONE JSON line (rank 0) with the `roofline` of the dominant kernel (device time from HIP
events on the ctx stream) and the `cpu_baseline` (CPU oracle timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


DT = 0.0005
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# algorithmic bytes per particle, per launch of each phase
ALGO_BYTES = {
    "predict": 48 + 16 + 4,        # predict 48 + cell-id 16 + histogram 4
    "sort": 2 + 80,                # scan ~2 + scatter/reorder 80
    "lambda": 18,                  # density+lambda fused
    "delta": 28,                   # delta-p + collide
    "finalize": 48,
}
ALGO_BYTES_SUBSTEP_4IT = 382
ALGO_BYTES_PER_ITER = 46           # 18 + 28


PMC_KERNELS = {"lambda": "k_lambda<true, 0, false>", "delta": "k_delta<0, false, false>", "predict": "k_predict",
               "finalize": "k_finalize", "neighbors": "k_neighbors<true>"}
N_CU, N_SIMD, N_XCD = 256, 1024, 8     # MI355X_MICROARCH.md


def kernel_source_sha():
    """fingerprint of the device code the PMC passes profiled: sha256 over the kernel sources with
    comments and white space removed (a reworded comment is not another kernel).
    tools/summarize_prof.py stamps profiles/pmc_latest.json with it; a bench run of other sources
    must not quote that file's counters."""
    import hashlib
    import re
    csrc = ROOT / "fluidsimulation_amd" / "csrc"
    h = hashlib.sha256()
    for f in [csrc / "pbf_hip.hip", csrc / "pbf_device.h"] + sorted(csrc.glob("kernels/*.inc")):
        text = f.read_text()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", " ", text)                    # line comments (the sources hold no "//" in strings)
        h.update(f.name.encode())
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def load_pmc():
    """profiles/pmc_latest.json (per-kernel, per-launch averages of the rocprofv3 --pmc passes), or
    None when it is absent or was taken from other kernel sources than the ones in this tree"""
    f = ROOT / "profiles" / "pmc_latest.json"
    try:
        d = json.loads(f.read_text())
    except Exception:
        return None, "no profiles/pmc_latest.json"
    if d.get("kernel_source_sha") != kernel_source_sha():
        return None, (f"profiles/pmc_latest.json was taken from kernel sources {d.get('kernel_source_sha')}, "
                      f"this tree is {kernel_source_sha()}: counters not quoted")
    return d, None


def pmc_traffic(d, kernel_phase, iters=4):
    """HBM bytes per launch of the phase's kernel: FETCH_SIZE counts 64-B requests at half their
    size for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM; calibrated on k_finalize:
    32 MB read -> 16 MB counted), WRITE_SIZE is exact; both in KiB."""
    try:
        k = d["kernels"][PMC_KERNELS[kernel_phase]]
        nbytes = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        last = d["kernels"].get("k_delta<0, false, true>")
        if kernel_phase == "delta" and last and iters > 0:
            # the sub-step's last delta-p launch also carries the velocity update
            nbytes = ((iters - 1) * nbytes + (2.0 * last["FETCH_SIZE"] + last["WRITE_SIZE"]) * 1024.0) / iters
        return {"bytes_per_launch": nbytes, "particles_per_launch": d["particles_per_launch"]}
    except Exception:
        return None


def pmc_limiters(d, kernel_phase):
    """How busy each candidate limiter was during the kernel, from the counters alone:
    valu = SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs) x 4 / (kernel cycles x SIMDs);
    l1   = TCP_TOTAL_CACHE_ACCESSES / (kernel cycles x CUs)  (the vector L1 takes about one access --
           one distinct 32-byte sector -- per clock: tools/micro/gather_cost.hip);
    lds  = SQ_INSTS_LDS x 8 clocks (a 64-lane ds_read_b128 at 128 B/clk) / (kernel cycles x CUs);
    hbm  = (2 FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s.
    kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
    try:
        k = d["kernels"][PMC_KERNELS[kernel_phase]]
        cyc = k["GRBM_GUI_ACTIVE"] / N_XCD
        clock_hz = float(d.get("clock_hz") or 2.4e9)
        out = {"valu": k["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD),
               "l1": k["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cyc * N_CU),
               "lds": k.get("SQ_INSTS_LDS", 0.0) * 8.0 / (cyc * N_CU),
               "hbm": (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / (cyc / clock_hz) / (HBM_PEAK_GBS * 1e9),
               "valu_insts_per_wave": k["SQ_INSTS_VALU"] / k["SQ_WAVES"],
               "kernel_cycles": cyc}
        return {a: (round(b, 4) if a != "kernel_cycles" else round(b)) for a, b in out.items()}
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--block", type=int, nargs=3, default=None, help="nx ny nz of the particle block")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=40,
                    help="oracle sub-steps to time on the same block as the GPU (about 10 s of CPU work; 0 = skip)")
    ap.add_argument("--cpu-block", type=int, nargs=3, default=None)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--xsph", type=float, default=0.0, help="XSPH viscosity coefficient (BASELINE config 5: 0.01)")
    ap.add_argument("--vorticity", type=float, default=0.0, help="vorticity confinement epsilon (config 5: 2e-6)")
    ap.add_argument("--strong", action="store_true",
                    help="N>1: --block is the WHOLE scene, cut into N slabs (strong scaling, e.g. BASELINE's 16 M "
                         "case: --strong --block 256 250 250); default is weak scaling, one block per rank")
    ap.add_argument("--ghost-layers", type=int, default=0,
                    help="ghost cell planes per slab side for N>1; 0 = time 4 (one halo refresh per sub-step) and "
                         "2 (three, half the ghost particles) on this machine's links before the run and keep "
                         "the faster; results are bit-identical either way")
    ap.add_argument("--overlap", action="store_true",
                    help="with --ghost-layers L >= 2: overlap the ghost refresh with the interior lambda sweep "
                         "(pbf_slab_set_overlap); without --ghost-layers it is one of the timed candidates")
    ap.add_argument("--evolved-steps", type=int, default=2000,
                    help="N=1: after the BASELINE window, carry the same scene on until it has taken this many "
                         "sub-steps (the collapsing dam break) and time it again -> `evolved` (0 = skip)")
    ap.add_argument("--no-shuffle", action="store_true",
                    help="upload the block in generation order (x-major, z-fastest = already cell-sorted) instead "
                         "of a fixed random permutation")
    ap.add_argument("--no-sizes", action="store_true",
                    help="N=1: skip the two other single-GPU sizes BASELINE names (C2: 64 k particles, and the "
                         "16 M-particle scene of C4 on one GPU)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N>1: skip the secondary records (BASELINE's 16 M strong-scaling scene and the "
                         "8 M-per-GPU XSPH + vorticity weak-scaling scene)")
    ap.add_argument("--extras-strong-block", type=int, nargs=3, default=[256, 250, 250])
    ap.add_argument("--extras-weak-block", type=int, nargs=3, default=[200, 200, 200])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path with several ranks sharing one GPU "
                         "(payloads staged through the host); numbers from it are not bench results")
    return ap.parse_args()


def cpu_baseline(block, iters, steps):
    """CPU oracle (a port of the reference's OpenMP path) on a bounded sample of the workload."""
    from fluidsimulation_amd import scenes
    from oracle import pyoracle
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    p, pos, mass, cols = scenes.dam_break(*block, iterations=iters)
    o = pyoracle.Oracle(p, threads=threads)
    o.upload(pos, uniform_mass=mass)
    o.set_colliders(cols)
    o.step(DT)  # warm-up (first frame)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step(DT)
    dt = time.perf_counter() - t0
    n = pos.shape[0]
    # the reference hard-codes num_threads(8) in every pragma (Simulation.cpp:587,619,648,685):
    # a shorter second sample at that setting
    ref8 = None
    if threads > 8:
        o.set_threads(8)
        k8 = max(1, steps // 4)
        t0 = time.perf_counter()
        for _ in range(k8):
            o.step(DT)
        ref8 = n * k8 / (time.perf_counter() - t0)
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
    except OSError:
        pass
    return {
        "value": n * steps / dt,
        "unit": "particle-substeps/s",
        "cores": threads,
        "cpu_model": model,
        "host_cpus": os.cpu_count(),
        "value_at_8_threads": ref8,
        "kind": "port",
        "sample": f"{steps} sub-steps of the {block[0]}x{block[1]}x{block[2]} dam-break block "
                  f"({n} particles, {iters} iterations) after 1 warm-up, CPU oracle (oracle/pbf_oracle.c) "
                  f"with {threads} OpenMP threads",
    }


def shuffled(pos, enabled=True):
    """The generators emit the block x-major / z-fastest, which IS the cell-sorted order: uploaded
    like that the first counting sort would be a no-op and every particle's id would equal its
    sorted rank.  A fixed permutation removes that best case (ids, hence the order inside a cell,
    are then unrelated to position); the physics does not depend on it."""
    if not enabled:
        return pos, "generation order (already cell-sorted)"
    import numpy as np
    perm = np.random.default_rng(20240611).permutation(pos.shape[0])
    return np.ascontiguousarray(pos[perm]), "fixed random permutation (seed 20240611)"


def phase_table(ctx, stepper, barrier, ksteps):
    """per-kernel device time with HIP events on the ctx stream (the event records sit between
    launches, so such a pass is never the one a `value` comes from)"""
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(ksteps):
        stepper(DT)
    barrier()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    return {name: {"ms_per_launch": ms / calls, "launches_per_step": calls / ksteps, "ms_per_step": ms / ksteps}
            for name, (ms, calls) in prof.items() if calls}


def per_rank_breakdown(sim, rank, world, k, backend):
    """Every rank's own account of k sub-steps, gathered on all ranks: device time per phase from HIP
    events on the ctx stream -- predict (= the split pass of the native slab step), sort + neighbors
    (the grid), lambda + delta (the sweeps), exchange (count / payload / halo exchanges: pack, grouped
    send + receive, unpack; includes waiting for the neighbour) -- wall time, owned and ghost
    counts.  A profiled pass of its own (event records sit between the launches): never the pass a
    `value` comes from.  One hardware run then explains its own parallel efficiency."""
    import torch.distributed as dist
    ctx = sim.ctx
    sim.sync()
    dist.barrier()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(k):
        sim.step(DT)
    sim.sync()
    wall = (time.perf_counter() - t0) / k
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    owned = int(sim.backend.owned_count())
    mine = {"rank": rank, "owned": owned, "ghosts": int(ctx.n) - owned, "wall_ms_per_step": wall * 1e3,
            "device_ms_per_step": {name: ms / k for name, (ms, calls) in prof.items() if calls},
            "launches_per_step": {name: calls / k for name, (ms, calls) in prof.items() if calls}}
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def multi_gpu_extras(a, sim, rank, world, local, time_steps):
    """Inside a --gpus N launch: BASELINE.json's two multi-GPU configurations, each next to the same
    work on ONE GPU (rank 0 alone, one ctx), so the line carries measured efficiencies.
      strong_16m   : the 256x250x250 = 16 M-particle scene as N x-slabs (C4, strong scaling)
      weak_8m_visc : one 200^3 = 8 M-particle block per GPU with XSPH + vorticity (C5, weak scaling)
    Sizes follow --extras-strong-block / --extras-weak-block (the CPU-side rehearsal shrinks them)."""
    import torch
    import torch.distributed as dist
    from fluidsimulation_amd import PbfContext, scenes
    from fluidsimulation_amd.slab import SlabSimulation
    shared = getattr(sim.group.exchange, "comm", None)
    dev = "cuda" if a.backend == "nccl" else "cpu"
    layers = sim.ghost_layers
    warm, k = 5, 10

    def agree(ok):
        """every rank learns whether all ranks got through the last rank-local stage (allocation,
        scene set-up): a rank that failed must not leave the others inside a collective"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def one_gpu(p, pos, mass, cols):
        """ms per sub-step of one ctx on rank 0's GPU, told to every rank (None: it did not fit / failed)"""
        t = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
        if rank == 0:
            try:
                g = PbfContext(p, device=local)
                g.upload(pos, uniform_mass=mass)
                g.set_colliders(cols)
                for _ in range(warm):
                    g.step(DT)
                g.sync()
                t0 = time.perf_counter()
                for _ in range(k):
                    g.step(DT)
                g.sync()
                t[0] = (time.perf_counter() - t0) / k
                g.close()
            except Exception as e:  # noqa: BLE001
                print(f"[bench extras] one-GPU reference failed: {e}", file=sys.stderr)
            torch.cuda.empty_cache()
        dist.broadcast(t, 0)
        v = float(t.item())
        return v if v > 0 else None

    def as_slabs(p, pos, mass, cols):
        s, err = None, None
        try:
            s = SlabSimulation(p, pos, mass, cols, rank=rank, world=world, device=local, ghost_layers=layers,
                               comm=shared, transport="torch" if shared is None else "auto", overlap=sim.overlap)
        except Exception as e:  # noqa: BLE001  (rank-local: allocation, capacity)
            err = f"rank {rank}: {e}"
        if not agree(err is None):
            return None, err or "another rank could not set up its slab"
        try:   # pbf_slab_step fails collectively (every rank raises the same error in the same sub-step)
            for _ in range(warm):
                s.step(DT)
            t = time_steps(s.step, s.sync, k)
            st = s.stats()
            owned = s.owned_counts()
            per_rank = per_rank_breakdown(s, rank, world, k, a.backend)
            s.sync()
        except Exception as e:  # noqa: BLE001
            return None, str(e)
        finally:
            del s
            torch.cuda.empty_cache()
        return (t, st, owned, per_rank), None

    out = {}
    # C4: strong scaling of ONE 16 M-particle scene
    nx, ny, nz = a.extras_strong_block
    p, pos, mass, cols = scenes.dam_break(nx, ny, nz, iterations=a.iters)
    pos, _ = shuffled(pos, not a.no_shuffle)
    n = pos.shape[0]
    t1 = one_gpu(p, pos, mass, cols)
    res, err = as_slabs(p, pos, mass, cols)
    rec = {"workload": f"{nx}x{ny}x{nz} dam-break block ({n} particles), {a.iters} iterations, {world} x-slabs, "
                       f"{layers} ghost layers", "scaling": "strong", "steps": k, "warmup": warm,
           "ms_per_step_1gpu": None if t1 is None else t1 * 1e3, "value_1gpu": None if t1 is None else n / t1}
    if res:
        tn, st, owned, per_rank = res
        rec.update({"ms_per_step": tn * 1e3, "value": n / tn,
                    "parallel_efficiency": None if t1 is None else t1 / (world * tn), "owned_per_rank": owned,
                    "nonfinite": st["nonfinite"], "neighbor_overflow": st["neighbor_overflow"],
                    "per_rank": per_rank})
    else:
        rec["error"] = err
    out["strong_16m"] = rec
    del pos
    # C5: weak scaling, one 8 M-particle block per GPU, viscosity stages on
    bx, by, bz = a.extras_weak_block
    xs, vo = 0.01, 2e-6
    p1, pos1, mass, cols1 = scenes.dam_break(bx, by, bz, iterations=a.iters, xsph_c=xs, vorticity_eps=vo)
    pos1, _ = shuffled(pos1, not a.no_shuffle)
    t1 = one_gpu(p1, pos1, mass, cols1)
    n1 = pos1.shape[0]
    del pos1
    p, pos, mass, cols = scenes.dam_break(bx * world, by, bz, iterations=a.iters, xsph_c=xs, vorticity_eps=vo)
    pos, _ = shuffled(pos, not a.no_shuffle)
    n = pos.shape[0]
    res, err = as_slabs(p, pos, mass, cols)
    rec = {"workload": f"{bx}x{by}x{bz} block per GPU ({n1} particles each, {n} in all), {a.iters} iterations, "
                       f"XSPH {xs} + vorticity {vo}, {world} x-slabs, {layers} ghost layers", "scaling": "weak",
           "steps": k, "warmup": warm, "ms_per_step_1gpu_one_block": None if t1 is None else t1 * 1e3,
           "value_1gpu_one_block": None if t1 is None else n1 / t1}
    if res:
        tn, st, owned, per_rank = res
        rec.update({"ms_per_step": tn * 1e3, "value": n / tn,
                    "parallel_efficiency": None if t1 is None else t1 / tn, "owned_per_rank": owned,
                    "nonfinite": st["nonfinite"], "neighbor_overflow": st["neighbor_overflow"],
                    "per_rank": per_rank})
    else:
        rec["error"] = err
    out["weak_8m_visc"] = rec
    return out


def self_launch(a):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N
    ranks ourselves -- torch.distributed.run as a CHILD process, before this process has imported
    torch or touched a GPU (never an exec) -- relay the ranks' output and exit with the child's code.
    The one JSON line is printed by rank 0 of the child."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = 0
    for ln in proc.stdout:
        if ln.startswith("{"):
            lines += 1
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"[bench] the {a.gpus}-rank child printed {lines} JSON lines (expected 1)", file=sys.stderr)
        rc = 1
    return rc


def main():
    a = parse()
    # the world comes from the launcher's environment; --gpus must agree with it, and a bare
    # `python bench.py --gpus N` launches its N ranks itself: a multi-GPU line can not silently
    # come from one GPU
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to "
              f"measure a world other than the one asked for", file=sys.stderr)
        raise SystemExit(2)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PBF path has no CPU fallback")
    if a.backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    from fluidsimulation_amd import PbfContext, scenes

    block = tuple(a.block) if a.block else (100, 100, 100)
    ranks_seen = {"all_reduce_of_ones": 1, "pbf_comm_world": None, "devices": None}
    if world > 1:
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen["all_reduce_of_ones"] = int(ones.item())
        devs = [None] * world
        dist.all_gather_object(devs, f"{os.uname().nodename}:cuda{local}")
        ranks_seen["devices"] = devs
        if ranks_seen["all_reduce_of_ones"] != a.gpus:
            raise SystemExit(f"bench.py: {ranks_seen['all_reduce_of_ones']} ranks answered, --gpus {a.gpus}")

    if world == 1:
        p, pos, mass, cols = scenes.dam_break(*block, iterations=a.iters, xsph_c=a.xsph, vorticity_eps=a.vorticity)
        pos, upload_order = shuffled(pos, not a.no_shuffle)
        ctx = PbfContext(p, device=local)
        ctx.upload(pos, uniform_mass=mass)
        ctx.set_colliders(cols)
        n_total = pos.shape[0]
        stepper = ctx.step
        syncer = ctx.sync
        workload = f"C3: {block[0]}x{block[1]}x{block[2]} dam-break block ({n_total} particles), {a.iters} solver iterations, dt=5e-4"
        parallelism = "single GPU"
    else:
        from fluidsimulation_amd.slab import SlabSimulation
        # weak scaling: every rank brings one block; slabs along x
        scene_nx = block[0] if a.strong else block[0] * world
        p, pos, mass, cols = scenes.dam_break(scene_nx, block[1], block[2], iterations=a.iters, xsph_c=a.xsph,
                                              vorticity_eps=a.vorticity)
        pos, upload_order = shuffled(pos, not a.no_shuffle)
        # ghost_layers = 4: one halo refresh per sub-step instead of three (slab.py, "communication-
        # avoiding halo"); transport: grouped ncclSend/ncclRecv from the C-ABI library when the
        # group runs on nccl (torch.distributed stays the bootstrap and the fallback)
        def time_steps(step, sync, k):
            sync(); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                step(DT)
            sync(); torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                             device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)   # every rank sees the same figure
            return float(t.item()) / k

        layers, tuning = a.ghost_layers, None
        if layers > 0:
            sim = SlabSimulation(p, pos, mass, cols, rank=rank, world=world, device=local, ghost_layers=layers,
                                 overlap=a.overlap)
        else:
            # The trade between ghost work, exchange rounds and hidden link time depends on the links:
            # measure it here, before the run.  Candidates: ghost depth 4 (one refresh per sub-step), 2
            # (three, half the ghost particles), 2 with the refresh overlapped with the interior lambda
            # sweep (native step only), 1 (lambda exchanged too, a quarter of the ghost particles).  All
            # share one communicator; the fastest carries on, the others are dropped.  Results are
            # bit-identical for every candidate (tests/test_gpu_comm.py, tests/test_gpu_slab.py).
            cands, shared, tuning = [], None, {}
            for L, ov in ((4, False), (2, False), (2, True), (1, False)):
                s_ = SlabSimulation(p, pos, mass, cols, rank=rank, world=world, device=local, ghost_layers=L,
                                    overlap=ov, comm=shared, transport="auto" if (shared is not None or not cands) else "torch")
                if not cands:
                    shared = getattr(s_.group.exchange, "comm", None)
                if ov and not s_.overlap:      # no native step (rehearsal over gloo): nothing to overlap
                    del s_
                    continue
                cands.append(s_)
            for cand in cands:
                for _ in range(5):
                    cand.step(DT)
                tuning[f"L{cand.ghost_layers}" + ("+overlap" if cand.overlap else "")] = time_steps(cand.step, cand.sync, 15)
            best = min(range(len(cands)), key=lambda k: list(tuning.values())[k])
            sim = cands[best]
            layers = sim.ghost_layers
            for k, cand in enumerate(cands):
                cand.sync()
                cand.owns_comm = (k == best) and shared is not None   # the communicator now belongs to the survivor
            cands = None
        n_total = pos.shape[0]
        stepper = sim.step
        syncer = sim.sync
        ctx = sim.ctx
        comm = getattr(sim.group.exchange, "comm", None)   # the C-ABI library's RCCL communicator, if in use
        ranks_seen["pbf_comm_world"] = int(comm.world) if comm is not None and hasattr(comm, "world") else None
        workload = (f"{scene_nx}x{block[1]}x{block[2]} dam-break block ({n_total} particles) in "
                    f"{world} x-slabs, {a.iters} solver iterations, dt=5e-4")
        parallelism = f"slab{world}, {layers} ghost layers" + (" + overlapped refresh" if sim.overlap else "") + \
            (f" (picked from ms/sub-step {({k: round(v * 1e3, 4) for k, v in tuning.items()})})" if tuning else "") + \
            f", transport {sim.transport}"

    def barrier():
        syncer()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        stepper(DT)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        stepper(DT)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # health figures of the whole scene (N>1: per-rank stats combined by two small all-reduces)
    stats = ctx.stats().as_dict() if world == 1 else sim.stats()

    # per-kernel device time with HIP events on the ctx stream (a second pass: the event
    # records sit between launches, so this pass is not the one `value` comes from)
    roofline = None
    phases = {}
    pmc, pmc_note = load_pmc()
    if not a.no_profile:
        ksteps = max(1, min(a.steps, 50))
        phases = phase_table(ctx, stepper, barrier, ksteps)
        n_local = ctx.n
        algo = dict(ALGO_BYTES)
        if "finalize" not in phases and "delta" in phases:
            # the velocity update rides along with the last delta-p launch of a sub-step (k_delta<FINAL>):
            # its 48 algorithmic bytes are spread over the sub-step's delta launches
            algo["delta"] = ALGO_BYTES["delta"] + ALGO_BYTES["finalize"] / phases["delta"]["launches_per_step"]
        cand = {k: v for k, v in phases.items() if k in algo}
        if cand:
            dom = max(cand, key=lambda k: cand[k]["ms_per_step"])
            per_launch_s = cand[dom]["ms_per_launch"] * 1e-3
            achieved = algo[dom] * n_local / per_launch_s / 1e9
            traffic, limiters, achieved_hbm = None, None, None
            # the PMC summary derives its particle count from SQ_WAVES x 64 (rounded up per block)
            if pmc and abs((pmc.get("particles_per_launch") or 0) - n_local) < 1024:
                tr = pmc_traffic(pmc, dom, a.iters)
                traffic = tr["bytes_per_launch"] if tr else None
                limiters = {k: pmc_limiters(pmc, k) for k in ("lambda", "delta", "neighbors")}
                achieved_hbm = traffic / per_launch_s / 1e9 if traffic else None
            elif pmc:
                pmc_note = "profiles/pmc_latest.json profiled another particle count: counters not quoted"
            busiest = None
            if limiters and limiters.get(dom):
                lim = {k: v for k, v in limiters[dom].items() if k in ("valu", "l1", "lds", "hbm")}
                busiest = max(lim, key=lim.get)
            # `bound` names the roofline `achieved` / `peak` / `frac` are priced against (HBM;
            # the path has no MFMA work).  What the counters say actually limits the kernel is `limited_by`.
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                        "achieved_is": "algorithmic bytes (SURVEY.md 8(d)) / measured launch time",
                        "achieved_hbm_gbs": achieved_hbm,
                        "limited_by": busiest, "limiters": limiters, "pmc_note": pmc_note,
                        "algorithmic_bytes_per_launch": algo[dom] * n_local,
                        "algorithmic_bytes_per_particle": algo[dom],
                        "avg_launch_ms": cand[dom]["ms_per_launch"], "particles_per_launch": n_local,
                        # the HBM fraction of every phase it prices, lambda and delta-p separately
                        "per_kernel": {k: {"algorithmic_bytes_per_particle": algo[k],
                                           "avg_launch_ms": v["ms_per_launch"],
                                           "achieved": algo[k] * n_local / (v["ms_per_launch"] * 1e-3) / 1e9,
                                           "frac": algo[k] * n_local / (v["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                       for k, v in cand.items()}}

    # the same scene once it is no lattice at rest any more (N = 1): `value` above is the BASELINE
    # window (sub-steps warmup .. warmup + steps of the block at rest, ~20 neighbours per particle)
    evolved = None
    if world == 1 and a.evolved_steps > 0:
        done = a.warmup + a.steps + (0 if a.no_profile else max(1, min(a.steps, 50)))
        for _ in range(max(0, a.evolved_steps - done)):
            stepper(DT)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            stepper(DT)
        barrier()
        t_ev = time.perf_counter() - t0
        st = ctx.stats().as_dict()
        evolved = {"after_substeps": max(done, a.evolved_steps), "steps": a.steps, "ms_per_step": t_ev / a.steps * 1e3,
                   "value": n_total * a.steps / t_ev, "mean_neighbors": st["mean_neighbor_count"],
                   "max_neighbors": st["max_neighbor_count"], "mean_abs_C": st["mean_abs_C"],
                   "max_speed": st["max_speed"], "nonfinite": st["nonfinite"],
                   "neighbor_overflow": st["neighbor_overflow"],
                   "phases": None if a.no_profile else phase_table(ctx, stepper, barrier, max(1, min(a.steps, 50)))}

    # N = 1: the metric is quoted "at 1M and 16M particles"; C2 (64 k, the README CUDA scene's count) is
    # BASELINE's other single-GPU configuration.  Same kernels, same arithmetic, 20 timed sub-steps each.
    sizes = None
    if world == 1 and not a.no_sizes and not a.block:
        sizes = {}
        torch.cuda.empty_cache()
        # C5's scene (64 M particles = 8 M per GPU on eight GPUs, XSPH 0.01 + vorticity 2e-6) on ONE GPU: it fits
        # (about 35 GiB of the 288), so the weak-scaling configuration has a measured one-GPU figure of the
        # whole scene beside it
        for name, blk, visc in (("C2_64k", (40, 40, 40), None), ("C4_16m_one_gpu", (256, 250, 250), None),
                                ("C5_64m_one_gpu", (1600, 200, 200), (0.01, 2e-6))):
            kw = {} if visc is None else {"xsph_c": visc[0], "vorticity_eps": visc[1]}
            q, qpos, qmass, qcols = scenes.dam_break(*blk, iterations=a.iters, **kw)
            qpos, _ = shuffled(qpos, not a.no_shuffle)
            try:
                g = PbfContext(q, device=local)
            except Exception as e:  # noqa: BLE001 -- a smaller GPU: say so, keep the line
                sizes[name] = {"error": str(e)}
                continue
            g.upload(qpos, uniform_mass=qmass)
            g.set_colliders(qcols)
            k_warm, k_time = (10, 20) if qpos.shape[0] <= 20_000_000 else (5, 10)
            for _ in range(k_warm):
                g.step(DT)
            g.sync()
            t0 = time.perf_counter()
            for _ in range(k_time):
                g.step(DT)
            g.sync()
            t = (time.perf_counter() - t0) / k_time
            st = g.stats().as_dict()
            sizes[name] = {"workload": f"{blk[0]}x{blk[1]}x{blk[2]} dam-break block ({qpos.shape[0]} particles), "
                                       f"{a.iters} iterations" + ("" if visc is None else f", XSPH {visc[0]} + vorticity {visc[1]}"),
                           "ms_per_step": t * 1e3, "value": qpos.shape[0] / t,
                           "steps": k_time, "warmup": k_warm, "mean_neighbors": st["mean_neighbor_count"],
                           "mean_abs_C": st["mean_abs_C"], "max_speed": st["max_speed"],
                           "nonfinite": st["nonfinite"], "neighbor_overflow": st["neighbor_overflow"]}
            g.close()
            del g, qpos
            torch.cuda.empty_cache()

    # N > 1: every rank's phase breakdown of the headline scene (a pass of its own, after the timed one)
    per_rank = None
    if world > 1 and not a.no_profile:
        per_rank = per_rank_breakdown(sim, rank, world, max(1, min(a.steps, 20)), a.backend)

    # N > 1: the two multi-GPU configurations BASELINE.json names, measured inside the same launch
    extras = None
    if world > 1 and not a.no_extras:
        extras = multi_gpu_extras(a, sim, rank, world, local, time_steps)

    if rank == 0:
        value = n_total * a.steps / elapsed
        cpu = None
        if a.cpu_steps > 0 and world == 1:
            cb = tuple(a.cpu_block) if a.cpu_block else block
            cpu = cpu_baseline(cb, a.iters, a.cpu_steps)
        out = {
            "metric": "particle-substeps/sec (PBF, 4 iters)",
            "value": value,
            "unit": "particle-substeps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if (a.strong and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if a.backend == "nccl" else "synthetic (REHEARSAL over gloo: not a bench result)",
            "config": {"workload": workload + (f", XSPH {a.xsph} + vorticity {a.vorticity}" if (a.xsph or a.vorticity) else ""),
                       "particles": n_total, "iterations": a.iters, "upload_order": upload_order,
                       "timed_window": f"sub-steps {a.warmup}..{a.warmup + a.steps} from the block at rest",
                       "arithmetic": "f32 state and accumulation, f64 SPH-kernel evaluation (as the reference CPU path)", "parallelism": parallelism,
                       "substeps_per_s": a.steps / elapsed},
            "hbm_roofline_frac_substep": (290 - 2 * ALGO_BYTES_PER_ITER + a.iters * ALGO_BYTES_PER_ITER) * value / (HBM_PEAK_GBS * 1e9 * world),
            "roofline": roofline,
            # VALU-busy fraction of the three instruction-heavy kernels from the committed PMC pass
            # (SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x 1024 SIMDs)); null when the pass is stale
            "roofline_valu": None if not (roofline and roofline.get("limiters")) else
                             {k: (v or {}).get("valu") for k, v in roofline["limiters"].items()},
            "phases": phases,
            "evolved": evolved,
            "other_sizes": sizes,
            "per_rank": per_rank,
            "extras": extras,
            "cpu_baseline": cpu,
            "health": {"nonfinite": stats["nonfinite"], "neighbor_overflow": stats["neighbor_overflow"],
                       "max_neighbors": stats["max_neighbor_count"],
                       "mean_neighbors": stats["mean_neighbor_count"], "mean_abs_C": stats["mean_abs_C"]},
        }
        print(json.dumps(out))
    if world > 1:
        import torch as _t
        _t.cuda.synchronize()
        dist.barrier()
        sim.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
