"""GPU: pbf_build_cells_noalloc (the allocation-free cell binning) against pbf_build_grid +
pbf_get_cells of a ctx with the same grid, bit for bit, through the C ABI -- two calls in a row,
the second with a smaller input (the library's fixed scratch array keeps the first call's
contents).  Collected last (file name) so that a failure here cannot hide another test.
"""
import ctypes as C

import numpy as np
import pytest

from fluidsimulation_amd import PbfContext, binding as B

pytestmark = pytest.mark.gpu


def ctx_cells(pos, grid_dim, origin, spacing):
    """the existing entry points: upload -> pbf_build_grid -> pbf_get_cells"""
    p = B.default_params()
    p.capacity = max(1, pos.shape[0])
    p.grid_dim[:] = list(grid_dim)
    p.grid_origin_cell[:] = list(origin)
    p.grid_spacing = spacing
    p.effective_radius = spacing
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=1.0)
    g.build_grid()
    _, key, sid, start = g.cells()
    g.close()
    return key, sid, start


def noalloc_cells(pos, grid_dim, origin, spacing):
    import torch
    lib = B.load_library()
    n = pos.shape[0]
    ncell = int(np.prod(grid_dim))
    dev = torch.device("cuda", 0)
    d_xyz = torch.from_numpy(np.ascontiguousarray(pos, np.float32)).to(dev)
    # outputs pre-filled with a pattern: every word must be written by the call
    d_key = torch.full((max(1, n),), -1, dtype=torch.int32, device=dev)
    d_sid = torch.full((max(1, n),), -1, dtype=torch.int32, device=dev)
    d_start = torch.full((ncell + 1,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    gd = (C.c_int32 * 3)(*grid_dim)
    go = (C.c_int32 * 3)(*origin)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.pbf_build_cells_noalloc(C.c_void_p(d_xyz.data_ptr()), n, C.byref(gd), C.byref(go),
                                     C.c_float(spacing), C.c_void_p(d_key.data_ptr()),
                                     C.c_void_p(d_sid.data_ptr()), C.c_void_p(d_start.data_ptr()), 0,
                                     C.c_void_p(stream))
    assert rc == B.PBF_OK, lib.pbf_last_error(None)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] >= free_before, "the call took device memory"
    return (d_key.cpu().numpy().view(np.uint32)[:n], d_sid.cpu().numpy().view(np.uint32)[:n],
            d_start.cpu().numpy().view(np.uint32))


def scattered(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    # crowd a few cells and repeat positions: the order inside a cell must still be by index
    pos[:40] = pos[0] + rng.uniform(0, 0.2, (40, 3)).astype(np.float32)
    pos[n // 2] = pos[n // 2 + 1]
    return pos


def test_noalloc_cells_equal_the_ctx_path_twice_second_call_smaller():
    cases = [
        # (n, grid, origin, spacing, coordinate range): the first wraps around the grid on purpose
        (200_000, (48, 40, 56), (-3, -2, -5), 0.25, (-2.0, 11.0)),
        (7_001, (9, 11, 7), (0, 0, 0), 1.0, (0.0, 6.9)),
    ]
    for k, (n, grid, origin, spacing, (lo, hi)) in enumerate(cases):
        pos = scattered(n, lo, hi, seed=100 + k)
        want = ctx_cells(pos, grid, origin, spacing)
        got = noalloc_cells(pos, grid, origin, spacing)
        for name, w, g in zip(("cell_key", "sorted_id", "cell_start"), want, got):
            assert np.array_equal(w, g), f"call {k}: {name} differs from pbf_build_grid's"
        assert got[2][-1] == n and np.array_equal(np.sort(got[1]), np.arange(n, dtype=np.uint32))


def test_noalloc_rejects_sizes_beyond_its_scratch_and_handles_empty_input():
    import torch
    lib = B.load_library()
    dev = torch.device("cuda", 0)
    buf = torch.zeros(64, dtype=torch.int32, device=dev)
    ptr = C.c_void_p(buf.data_ptr())
    go = (C.c_int32 * 3)(0, 0, 0)
    big = (C.c_int32 * 3)(128, 128, 128)     # 2^21 cells > PBF_NOALLOC_MAX_CELLS
    assert lib.pbf_build_cells_noalloc(ptr, 4, C.byref(big), C.byref(go), C.c_float(1.0), ptr, ptr, ptr, 0,
                                       None) == B.PBF_ERR_CAPACITY
    assert b"PBF_NOALLOC_MAX_CELLS" in lib.pbf_last_error(None)
    small = (C.c_int32 * 3)(3, 3, 3)
    assert lib.pbf_build_cells_noalloc(ptr, (1 << 20) + 1, C.byref(small), C.byref(go), C.c_float(1.0), ptr,
                                       ptr, ptr, 0, None) == B.PBF_ERR_CAPACITY
    assert b"PBF_NOALLOC_MAX_PARTICLES" in lib.pbf_last_error(None)
    assert lib.pbf_build_cells_noalloc(ptr, 4, C.byref(small), C.byref(go), C.c_float(0.0), ptr, ptr, ptr, 0,
                                       None) == B.PBF_ERR_INVALID
    # n = 0: cell_start is all zeros, nothing else is touched
    start = torch.full((28,), -1, dtype=torch.int32, device=dev)
    assert lib.pbf_build_cells_noalloc(None, 0, C.byref(small), C.byref(go), C.c_float(1.0), None, None,
                                       C.c_void_p(start.data_ptr()), 0, None) == B.PBF_OK
    torch.cuda.synchronize()
    assert int(start.abs().sum().item()) == 0


def _call(lib, d_xyz, n, grid, origin, spacing, d_key, d_sid, d_start, stream):
    gd = (C.c_int32 * 3)(*grid)
    go = (C.c_int32 * 3)(*origin)
    return lib.pbf_build_cells_noalloc(C.c_void_p(d_xyz.data_ptr()), n, C.byref(gd), C.byref(go), C.c_float(spacing),
                                       C.c_void_p(d_key.data_ptr()), C.c_void_p(d_sid.data_ptr()),
                                       C.c_void_p(d_start.data_ptr()), 0, C.c_void_p(stream))


def test_noalloc_call_is_capturable_into_a_hip_graph_and_replays_on_new_inputs():
    """the callers the entry point is for: under stream capture.  One warm-up call per device happens
    OUTSIDE capture (the first call resolves the scratch array's address, which makes the runtime load
    the code object: include/pbf.h says so); after that the call is memsets and kernel launches on the
    caller's stream and nothing else -- captured once, replayed on two different inputs, each replay
    equal to pbf_build_grid + pbf_get_cells."""
    import torch
    lib = B.load_library()
    dev = torch.device("cuda", 0)
    n, grid, origin, spacing = 60_000, (32, 24, 40), (-2, 0, -3), 0.5
    ncell = int(np.prod(grid))
    d_xyz = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    d_key = torch.zeros(n, dtype=torch.int32, device=dev)
    d_sid = torch.zeros(n, dtype=torch.int32, device=dev)
    d_start = torch.zeros(ncell + 1, dtype=torch.int32, device=dev)
    assert _call(lib, d_xyz, 0, grid, origin, spacing, d_key, d_sid, d_start,
                 torch.cuda.current_stream().cuda_stream) == B.PBF_OK          # warm-up, not captured
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        rc = _call(lib, d_xyz, n, grid, origin, spacing, d_key, d_sid, d_start, torch.cuda.current_stream().cuda_stream)
    assert rc == B.PBF_OK, lib.pbf_last_error(None)
    for seed in (7, 8):
        pos = scattered(n, -1.0, 15.0, seed)
        d_xyz.copy_(torch.from_numpy(pos))
        for t in (d_key, d_sid, d_start):
            t.fill_(-1)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = ctx_cells(pos, grid, origin, spacing)
        got = (d_key.cpu().numpy().view(np.uint32), d_sid.cpu().numpy().view(np.uint32),
               d_start.cpu().numpy().view(np.uint32))
        for name, w, g in zip(("cell_key", "sorted_id", "cell_start"), want, got):
            assert np.array_equal(w, g), f"replay with seed {seed}: {name} differs from pbf_build_grid's"


def test_noalloc_worst_case_every_particle_in_one_cell_stays_bounded():
    """the rank-by-index pass is O(cell occupancy) per particle and this entry point has no crowd
    limit: its particle limit IS the bound.  All 2^20 particles at one point = 2^40 compare steps;
    the call must still finish in seconds and give the right answer (index order)."""
    import time
    import torch
    lib = B.load_library()
    dev = torch.device("cuda", 0)
    n, grid = 1 << 20, (8, 8, 8)
    d_xyz = torch.full((n, 3), 3.25, dtype=torch.float32, device=dev)
    d_key = torch.zeros(n, dtype=torch.int32, device=dev)
    d_sid = torch.zeros(n, dtype=torch.int32, device=dev)
    d_start = torch.zeros(8 * 8 * 8 + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert _call(lib, d_xyz, n, grid, (0, 0, 0), 1.0, d_key, d_sid, d_start,
                 torch.cuda.current_stream().cuda_stream) == B.PBF_OK
    torch.cuda.synchronize()
    took = time.perf_counter() - t0
    print(f"2^20 particles in one cell: {took:.2f} s")
    assert took < 30.0
    assert torch.equal(d_sid, torch.arange(n, dtype=torch.int32, device=dev))
    key = (3 * 8 + 3) * 8 + 3
    start = d_start.cpu().numpy()
    assert (start[:key + 1] == 0).all() and (start[key + 1:] == n).all()
