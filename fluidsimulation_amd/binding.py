"""ctypes mirror of include/pbf.h and loader of the HIP extension (libpbf_hip.so).

Plumbing only: the product is the C-ABI library.  There is NO CPU fallback here --
if the extension is missing or no GPU is usable, the calls raise (PbfError).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PBF_ABI_VERSION = 2
PBF_ROWS_ABS32, PBF_ROWS_REL16, PBF_ROWS_WIN16, PBF_ROWS_ALL, PBF_ROWS_UNSTAGED = 1, 2, 4, 7, 8
PBF_MAX_COLLIDERS = 16

PBF_OK = 0
PBF_ERR_INVALID = 1
PBF_ERR_HIP = 2
PBF_ERR_CAPACITY = 3
PBF_ERR_NEIGHBOR_OVERFLOW = 4
PBF_ERR_NO_DEVICE = 5
PBF_ERR_NOT_INITIALIZED = 6

PBF_COLLIDER_AABB = 2   # Collider::ColliderTypes::AABB  (reference Collider.h:10-17)
PBF_COLLIDER_PLANE = 4  # Collider::ColliderTypes::PLANE
PBF_COLLIDER_HARD_BOX = 100  # the CUDA path's baked-in box (cuda_simulation.cu:473-511)

PBF_RHO0_FIRST_FRAME_MEAN = 0
PBF_RHO0_FIXED = 1
PBF_GRAD_POLY6 = 0
PBF_GRAD_SPIKY = 1


class PbfCollider(C.Structure):
    _fields_ = [("type", C.c_int32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("d", C.c_float)]


PBF_COMM_ID_BYTES = 128


PBF_CONSTRAINTS_PBD, PBF_CONSTRAINTS_XPBD = 0, 1


class PbfDistanceConstraint(C.Structure):
    _fields_ = [("p0", C.c_uint32), ("p1", C.c_uint32), ("rest_length", C.c_float),
                ("stiffness", C.c_float), ("compliance", C.c_float)]


def constraint_array(cons):
    """cons: iterable of (p0, p1, rest_length, stiffness, compliance)"""
    cons = list(cons)
    arr = (PbfDistanceConstraint * max(1, len(cons)))()
    for k, (a, b, rest, st, cm) in enumerate(cons):
        arr[k].p0, arr[k].p1, arr[k].rest_length = int(a), int(b), float(rest)
        arr[k].stiffness, arr[k].compliance = float(st), float(cm)
    return arr, len(cons)


class PbfDemParams(C.Structure):
    _fields_ = [("particle_radius", C.c_float), ("static_friction", C.c_float),
                ("kinematic_friction", C.c_float), ("iterations", C.c_int32)]


def dem_params(particle_radius, static_friction=1.0, kinematic_friction=0.75, iterations=3):
    """SimParams' DEM constants (reference Simulation.cpp:335,349-350; iterations :148)"""
    d = PbfDemParams()
    d.particle_radius, d.static_friction = float(particle_radius), float(static_friction)
    d.kinematic_friction, d.iterations = float(kinematic_friction), int(iterations)
    return d


class PbfP2P(C.Structure):
    _fields_ = [("send", C.c_int32), ("peer", C.c_int32), ("buf", C.c_void_p), ("bytes", C.c_uint64)]


class PbfParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("capacity", C.c_uint32),
        ("max_neighbors", C.c_uint32),
        ("grid_dim", C.c_int32 * 3),
        ("grid_origin_cell", C.c_int32 * 3),
        ("grid_spacing", C.c_float),
        ("effective_radius", C.c_float),
        ("gravity", C.c_float * 3),
        ("epsilon", C.c_float),
        ("restitution", C.c_float),
        ("scorr_k", C.c_double),
        ("scorr_dq", C.c_float),
        ("scorr_dt_power", C.c_int32),
        ("rest_density_mode", C.c_int32),
        ("rest_density", C.c_float),
        ("solver_iterations", C.c_int32),
        ("gradient_kernel", C.c_int32),
        ("container_enabled", C.c_int32),
        ("container_min", C.c_float * 3),
        ("container_max", C.c_float * 3),
        ("xsph_c", C.c_float),
        ("vorticity_eps", C.c_float),
    ]

    def copy(self) -> "PbfParams":
        q = PbfParams()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(PbfParams))
        return q


class PbfStats(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("nonfinite", C.c_uint32),
        ("neighbor_overflow", C.c_uint32),
        ("max_neighbor_count", C.c_uint32),
        ("mean_neighbor_count", C.c_double),
        ("mean_abs_C", C.c_double),
        ("max_abs_C", C.c_double),
        ("mean_C", C.c_double),
        ("com", C.c_double * 3),
        ("max_speed", C.c_double),
        ("rest_density", C.c_float),
        ("row_format_particles", C.c_uint32 * 3),
    ]

    def as_dict(self) -> dict:
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


def default_params() -> PbfParams:
    """The reference CPU path's constants (mirror of pbf_default_params)."""
    p = PbfParams()
    p.abi_version = PBF_ABI_VERSION
    p.capacity = 0
    p.max_neighbors = 64
    p.grid_dim[:] = [64, 64, 64]          # Simulation.cpp:46
    p.grid_spacing = 1.0                  # NeighborSearch.cpp:17
    p.effective_radius = 1.0              # Simulation.cpp:88
    p.gravity[:] = [0.0, -9.8, 0.0]       # Simulation.cpp:16
    p.epsilon = 1.0e-6                    # Simulation.cpp:644
    p.restitution = 0.99                  # Particle.cpp:333
    p.scorr_k = 0.1                       # Simulation.cpp:698
    p.scorr_dq = 0.3                      # Simulation.cpp:700
    p.scorr_dt_power = 1                  # Simulation.cpp:702
    p.rest_density_mode = PBF_RHO0_FIRST_FRAME_MEAN
    p.rest_density = 0.8                  # Simulation.cpp:18
    p.solver_iterations = 1               # ConstraintSolver.cpp:4
    p.gradient_kernel = PBF_GRAD_POLY6
    return p


def plane(normal, d) -> PbfCollider:
    c = PbfCollider()
    c.type = PBF_COLLIDER_PLANE
    c.a[:] = [float(x) for x in normal]
    c.d = float(d)
    return c


def aabb(mn, mx) -> PbfCollider:
    c = PbfCollider()
    c.type = PBF_COLLIDER_AABB
    c.a[:] = [float(x) for x in mn]
    c.b[:] = [float(x) for x in mx]
    return c


def hard_box(mn, mx, boundary_damping=0.1) -> PbfCollider:
    """integrate_pbd_d's wall response (cuda_simulation.cu:473-511) as a collider: clamp to the box (open
    top), point the velocity component back inside, multiply the whole velocity by boundary_damping"""
    c = PbfCollider()
    c.type = PBF_COLLIDER_HARD_BOX
    c.a[:] = [float(x) for x in mn]
    c.b[:] = [float(x) for x in mx]
    c.d = float(boundary_damping)
    return c


def collider_array(cols):
    arr = (PbfCollider * max(1, len(cols)))()
    for i, c in enumerate(cols):
        arr[i] = c
    return arr


class PbfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pbf error {code}: {msg}")
        self.code = code


_REPO = Path(__file__).resolve().parent.parent
LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libpbf_hip.so"

# every symbol include/pbf.h declares
SYMBOLS = [
    "pbf_default_params", "pbf_create", "pbf_destroy", "pbf_last_error", "pbf_upload",
    "pbf_upload_device", "pbf_set_colliders", "pbf_set_solver_iterations", "pbf_set_gravity",
    "pbf_set_row_formats",
    "pbf_predict", "pbf_build_grid", "pbf_compute_rest_density", "pbf_solve", "pbf_finalize",
    "pbf_step", "pbf_sync", "pbf_download", "pbf_get_scalars", "pbf_get_cells",
    "pbf_get_neighbors", "pbf_get_stats", "pbf_get_rest_density", "pbf_num_particles",
    "pbf_profile_enable", "pbf_profile_reset", "pbf_profile_read", "pbf_num_phases",
    "pbf_phase_name", "pbf_solve_lambda", "pbf_solve_delta", "pbf_finalize_stage", "pbf_set_stream",
    "pbf_export_positions", "pbf_snapshot_save", "pbf_snapshot_load",
    "pbf_slab_set_bounds", "pbf_slab_split", "pbf_slab_split_done", "pbf_slab_append_owned",
    "pbf_slab_append_ghosts",
    "pbf_slab_pack_halo", "pbf_slab_unpack_halo", "pbf_slab_export", "pbf_selftest_sqrt",
    "pbf_set_boundary", "pbf_num_boundary_particles", "pbf_get_boundary",
    "pbf_get_boundary_neighbors",
    "pbf_comm_unique_id", "pbf_comm_create", "pbf_comm_destroy", "pbf_comm_rank", "pbf_comm_world",
    "pbf_comm_last_error", "pbf_comm_batch", "pbf_slab_halo_exchange", "pbf_step_dem",
    "pbf_set_distance_constraints", "pbf_project_constraints", "pbf_get_constraint_lambdas",
    "pbf_comm_local_id", "pbf_slab_attach", "pbf_slab_step", "pbf_slab_owned_count",
    "pbf_slab_append_owned_host", "pbf_slab_export_host", "pbf_build_cells_noalloc", "pbf_slab_rebalance",
    "pbf_slab_set_overlap", "pbf_slab_step_dem",
]

_lib = None


def _fp(dtype):
    return np.ctypeslib.ndpointer(dtype=dtype, flags="C_CONTIGUOUS")


def load_library(path: os.PathLike | None = None):
    """dlopen libpbf_hip.so and declare prototypes.  Raises if the extension is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise PbfError(PBF_ERR_NO_DEVICE,
                       f"HIP extension not built: {p} (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    # PyTorch-ROCm bundles its own libamdhip64.so.7 and loads it by path; libpbf_hip.so needs the
    # same soname.  Whichever is loaded first serves both, and a process must end up with ONE
    # HIP runtime (torch buffers are handed to the pbf_slab_* calls): let torch load first.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for single-GPU use
        pass
    lib = C.CDLL(str(p))
    vp = C.c_void_p
    lib.pbf_default_params.argtypes = [C.POINTER(PbfParams)]
    lib.pbf_create.argtypes = [C.POINTER(PbfParams), C.c_int, C.POINTER(vp)]
    lib.pbf_destroy.argtypes = [vp]
    lib.pbf_last_error.argtypes = [vp]
    lib.pbf_last_error.restype = C.c_char_p
    lib.pbf_upload.argtypes = [vp, vp, vp, vp, C.c_float, C.c_uint32]
    lib.pbf_upload_device.argtypes = [vp, vp, vp, vp, C.c_float, C.c_uint32]
    lib.pbf_set_colliders.argtypes = [vp, C.POINTER(PbfCollider), C.c_uint32]
    lib.pbf_set_solver_iterations.argtypes = [vp, C.c_int32]
    lib.pbf_set_gravity.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    lib.pbf_set_row_formats.argtypes = [vp, C.c_uint32]
    lib.pbf_predict.argtypes = [vp, C.c_float]
    lib.pbf_build_grid.argtypes = [vp]
    lib.pbf_compute_rest_density.argtypes = [vp]
    lib.pbf_solve.argtypes = [vp, C.c_int32, C.c_float]
    lib.pbf_finalize.argtypes = [vp, C.c_float]
    lib.pbf_step.argtypes = [vp, C.c_float]
    lib.pbf_sync.argtypes = [vp]
    lib.pbf_download.argtypes = [vp, vp, vp, vp]
    lib.pbf_get_scalars.argtypes = [vp, vp, vp, vp]
    lib.pbf_get_cells.argtypes = [vp, vp, vp, vp, vp]
    lib.pbf_get_neighbors.argtypes = [vp, vp, vp, C.c_uint32]
    lib.pbf_get_stats.argtypes = [vp, C.POINTER(PbfStats)]
    lib.pbf_get_rest_density.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pbf_num_particles.argtypes = [vp]
    lib.pbf_num_particles.restype = C.c_uint32
    lib.pbf_profile_enable.argtypes = [vp, C.c_int]
    lib.pbf_profile_reset.argtypes = [vp]
    lib.pbf_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_uint32]
    lib.pbf_num_phases.restype = C.c_uint32
    lib.pbf_phase_name.argtypes = [C.c_uint32]
    lib.pbf_phase_name.restype = C.c_char_p
    lib.pbf_solve_lambda.argtypes = [vp, C.c_float]
    lib.pbf_solve_delta.argtypes = [vp, C.c_float]
    lib.pbf_finalize_stage.argtypes = [vp, C.c_int32, C.c_float]
    lib.pbf_set_stream.argtypes = [vp, vp]
    lib.pbf_export_positions.argtypes = [vp, vp]
    lib.pbf_snapshot_save.argtypes = [vp, C.c_char_p]
    lib.pbf_snapshot_load.argtypes = [vp, C.c_char_p]
    lib.pbf_slab_set_bounds.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.pbf_slab_split.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp]
    lib.pbf_slab_split_done.argtypes = [vp, C.POINTER(C.c_uint32 * 6)]
    lib.pbf_slab_append_owned.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.pbf_slab_append_ghosts.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.pbf_slab_pack_halo.argtypes = [vp, C.c_int32, vp, vp]
    lib.pbf_slab_unpack_halo.argtypes = [vp, C.c_int32, vp, C.c_uint32, vp, C.c_uint32]
    lib.pbf_slab_export.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.pbf_selftest_sqrt.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.pbf_set_distance_constraints.argtypes = [vp, C.POINTER(PbfDistanceConstraint), C.c_uint32, C.c_int32]
    lib.pbf_project_constraints.argtypes = [vp, C.c_float]
    lib.pbf_get_constraint_lambdas.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    lib.pbf_step_dem.argtypes = [vp, C.c_float, C.POINTER(PbfDemParams)]
    lib.pbf_comm_unique_id.argtypes = [vp]
    lib.pbf_comm_local_id.argtypes = [vp]
    lib.pbf_slab_attach.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_uint32]
    lib.pbf_slab_step.argtypes = [vp, C.c_float]
    lib.pbf_slab_set_overlap.argtypes = [vp, C.c_int32]
    lib.pbf_slab_step_dem.argtypes = [vp, C.c_float, C.POINTER(PbfDemParams)]
    lib.pbf_slab_owned_count.argtypes = [vp]
    lib.pbf_slab_rebalance.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.c_int32,
                                       C.POINTER(C.c_int32)]
    lib.pbf_slab_owned_count.restype = C.c_uint32
    lib.pbf_comm_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.pbf_comm_destroy.argtypes = [vp]
    lib.pbf_comm_rank.argtypes = [vp]
    lib.pbf_comm_rank.restype = C.c_int32
    lib.pbf_comm_world.argtypes = [vp]
    lib.pbf_comm_world.restype = C.c_int32
    lib.pbf_comm_last_error.argtypes = [vp]
    lib.pbf_comm_last_error.restype = C.c_char_p
    lib.pbf_comm_batch.argtypes = [vp, vp, C.POINTER(PbfP2P), C.c_uint32]
    lib.pbf_slab_halo_exchange.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.pbf_set_boundary.argtypes = [vp, vp, vp, C.c_float, C.c_uint32, C.c_uint32]
    lib.pbf_num_boundary_particles.argtypes = [vp]
    lib.pbf_num_boundary_particles.restype = C.c_uint32
    lib.pbf_get_boundary.argtypes = [vp, vp, vp, vp]
    lib.pbf_get_boundary_neighbors.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32]
    lib.pbf_build_cells_noalloc.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3),
                                            C.c_float, vp, vp, vp, C.c_int, vp]
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RcclComm:
    """One rank of an RCCL communicator behind the C ABI (pbf_comm_*).  `unique_id()` on rank 0,
    the 128 bytes travel to the other ranks by any means, then every rank constructs."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = -1):
        self.lib = load_library()
        assert len(unique_id) == PBF_COMM_ID_BYTES
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), PBF_COMM_ID_BYTES)
        rc = self.lib.pbf_comm_create(buf, rank, world, device, C.byref(self._h))
        if rc != PBF_OK:
            raise PbfError(rc, (self.lib.pbf_comm_last_error(None) or b"pbf_comm_create failed").decode())
        self.rank, self.world = rank, world

    @staticmethod
    def local_id() -> bytes:
        """id of the in-process loopback transport (ranks = threads of this process; tests)"""
        lib = load_library()
        buf = C.create_string_buffer(PBF_COMM_ID_BYTES)
        lib.pbf_comm_local_id(buf)
        return buf.raw

    @staticmethod
    def unique_id() -> bytes:
        lib = load_library()
        buf = C.create_string_buffer(PBF_COMM_ID_BYTES)
        rc = lib.pbf_comm_unique_id(buf)
        if rc != PBF_OK:
            raise PbfError(rc, (lib.pbf_comm_last_error(None) or b"pbf_comm_unique_id failed").decode())
        return buf.raw

    def batch(self, ops, stream: int = 0):
        """ops: (send: bool, peer: int, device pointer: int, bytes: int), one grouped exchange,
        asynchronous on `stream` (a hipStream_t handle)"""
        arr = (PbfP2P * max(1, len(ops)))()
        for i, (snd, peer, ptr, nbytes) in enumerate(ops):
            arr[i].send, arr[i].peer, arr[i].buf, arr[i].bytes = int(bool(snd)), int(peer), ptr or None, int(nbytes)
        rc = self.lib.pbf_comm_batch(self._h, C.c_void_p(stream or None), arr, len(ops))
        if rc != PBF_OK:
            raise PbfError(rc, self.lib.pbf_comm_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h:
            self.lib.pbf_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PbfContext:
    """Thin owner of one pbf_ctx.  Mirrors the C ABI one to one; raises PbfError on failure."""

    def __init__(self, params: PbfParams, device: int = -1):
        self.lib = load_library()
        self.params = params.copy()
        self._h = C.c_void_p()
        rc = self.lib.pbf_create(C.byref(self.params), int(device), C.byref(self._h))
        if rc != PBF_OK:
            msg = self.lib.pbf_last_error(self._h if self._h else None)
            raise PbfError(rc, (msg or b"pbf_create failed").decode())
        self.n = 0
        # test lever (tools/fuzz_*.py): PBF_ROW_FORMATS=5 puts every scene, however small, through
        # the neighbour-row format only large scenes would pick (pbf_set_row_formats)
        if os.environ.get("PBF_ROW_FORMATS"):
            self.set_row_formats(int(os.environ["PBF_ROW_FORMATS"]))

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h:
            self.lib.pbf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != PBF_OK:
            raise PbfError(rc, self.lib.pbf_last_error(self._h).decode())

    # -- setup
    def upload(self, pos, vel=None, mass=None, uniform_mass: float = 1.0):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        n = pos.shape[0]
        vel = None if vel is None else np.ascontiguousarray(vel, dtype=np.float32).reshape(n, 3)
        mass = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32).reshape(n)
        self._ck(self.lib.pbf_upload(self._h, _ptr(pos), _ptr(vel), _ptr(mass),
                                     C.c_float(uniform_mass), n))
        self.n = n

    def upload_device(self, d_pos: int, d_vel: int, d_mass: int, uniform_mass: float, n: int):
        self._ck(self.lib.pbf_upload_device(self._h, C.c_void_p(d_pos), C.c_void_p(d_vel or None),
                                            C.c_void_p(d_mass or None), C.c_float(uniform_mass), n))
        self.n = n

    def set_colliders(self, cols):
        self._ck(self.lib.pbf_set_colliders(self._h, collider_array(cols), len(cols)))

    def set_solver_iterations(self, it: int):
        self._ck(self.lib.pbf_set_solver_iterations(self._h, it))

    def set_boundary(self, pos, mass=None, uniform_mass: float = 1.0, max_boundary_neighbors: int = 0):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        nb = pos.shape[0]
        mass = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32).reshape(nb)
        self._ck(self.lib.pbf_set_boundary(self._h, _ptr(pos), _ptr(mass), C.c_float(uniform_mass),
                                           nb, max_boundary_neighbors))
        self.nb = nb
        self.kb = max_boundary_neighbors or int(self.params.max_neighbors)

    def boundary(self):
        """(volume, density, lambda) of the boundary particles, boundary index order"""
        out = [np.zeros(self.nb, np.float32) for _ in range(3)]
        self._ck(self.lib.pbf_get_boundary(self._h, *[_ptr(a) for a in out]))
        return tuple(out)

    def boundary_neighbors(self, stride_f=None, stride_b=None):
        """(fluid counts, fluid->boundary rows, boundary counts, boundary->fluid rows)"""
        stride_f = stride_f or self.kb
        stride_b = stride_b or self.kb
        fc, fl = np.zeros(self.n, np.uint32), np.zeros((self.n, stride_f), np.uint32)
        bc, bl = np.zeros(self.nb, np.uint32), np.zeros((self.nb, stride_b), np.uint32)
        self._ck(self.lib.pbf_get_boundary_neighbors(self._h, _ptr(fc), _ptr(fl), stride_f,
                                                     _ptr(bc), _ptr(bl), stride_b))
        return fc, fl, bc, bl

    def set_row_formats(self, mask):
        """PBF_ROWS_* the fluid neighbour table may use (results never depend on it)"""
        self._ck(self.lib.pbf_set_row_formats(self._h, mask))

    def set_gravity(self, gx, gy, gz):
        self._ck(self.lib.pbf_set_gravity(self._h, gx, gy, gz))

    # -- the sub-steps
    def predict(self, dt):
        self._ck(self.lib.pbf_predict(self._h, C.c_float(dt)))

    def build_grid(self):
        self._ck(self.lib.pbf_build_grid(self._h))

    def compute_rest_density(self):
        self._ck(self.lib.pbf_compute_rest_density(self._h))

    def solve(self, dt, iterations: int = 0):
        self._ck(self.lib.pbf_solve(self._h, iterations, C.c_float(dt)))

    def finalize(self, dt):
        self._ck(self.lib.pbf_finalize(self._h, C.c_float(dt)))

    def step(self, dt):
        self._ck(self.lib.pbf_step(self._h, C.c_float(dt)))

    def set_distance_constraints(self, cons, mode=PBF_CONSTRAINTS_XPBD):
        arr, k = constraint_array(cons)
        self._ck(self.lib.pbf_set_distance_constraints(self._h, arr, k, mode))
        self.ncons = k

    def project_constraints(self, dt):
        self._ck(self.lib.pbf_project_constraints(self._h, C.c_float(dt)))

    def constraint_lambdas(self):
        lam = np.zeros(max(1, getattr(self, "ncons", 0)), np.float32)
        lv = C.c_uint32()
        self._ck(self.lib.pbf_get_constraint_lambdas(self._h, _ptr(lam), C.byref(lv)))
        return lam[:getattr(self, "ncons", 0)], int(lv.value)

    def step_dem(self, dt, dem: "PbfDemParams"):
        self._ck(self.lib.pbf_step_dem(self._h, C.c_float(dt), C.byref(dem)))

    def sync(self):
        self._ck(self.lib.pbf_sync(self._h))

    # -- read-back
    def download(self):
        n = self.n
        pos = np.empty((n, 3), np.float32)
        vel = np.empty((n, 3), np.float32)
        pred = np.empty((n, 3), np.float32)
        self._ck(self.lib.pbf_download(self._h, _ptr(pos), _ptr(vel), _ptr(pred)))
        return pos, vel, pred

    def scalars(self):
        n = self.n
        d = np.empty(n, np.float32)
        c = np.empty(n, np.float32)
        lam = np.empty(n, np.float32)
        self._ck(self.lib.pbf_get_scalars(self._h, _ptr(d), _ptr(c), _ptr(lam)))
        return d, c, lam

    def cells(self):
        n = self.n
        g = self.params.grid_dim
        ncell = int(g[0]) * int(g[1]) * int(g[2])
        xyz = np.empty((n, 3), np.int32)
        key = np.empty(n, np.uint32)
        sid = np.empty(n, np.uint32)
        start = np.empty(ncell + 1, np.uint32)
        self._ck(self.lib.pbf_get_cells(self._h, _ptr(xyz), _ptr(key), _ptr(sid), _ptr(start)))
        return xyz, key, sid, start

    def neighbors(self):
        n = self.n
        stride = int(self.params.max_neighbors)
        counts = np.empty(n, np.uint32)
        nb = np.zeros((n, stride), np.uint32)
        self._ck(self.lib.pbf_get_neighbors(self._h, _ptr(counts), _ptr(nb), stride))
        return counts, nb

    def stats(self) -> PbfStats:
        s = PbfStats()
        self._ck(self.lib.pbf_get_stats(self._h, C.byref(s)))
        return s

    def rest_density(self) -> float:
        v = C.c_float()
        self._ck(self.lib.pbf_get_rest_density(self._h, C.byref(v)))
        return float(v.value)

    # -- split iteration / staged finalize / slab decomposition (device pointers are ints)
    def solve_lambda(self, dt):
        self._ck(self.lib.pbf_solve_lambda(self._h, C.c_float(dt)))

    def solve_delta(self, dt):
        self._ck(self.lib.pbf_solve_delta(self._h, C.c_float(dt)))

    def finalize_stage(self, stage, dt):
        self._ck(self.lib.pbf_finalize_stage(self._h, stage, C.c_float(dt)))

    def set_stream(self, stream_handle: int):
        self._ck(self.lib.pbf_set_stream(self._h, C.c_void_p(stream_handle)))

    def export_positions(self, d_xyz: int):
        """N packed float3 in particle-index order into device memory (asynchronous)"""
        self._ck(self.lib.pbf_export_positions(self._h, C.c_void_p(d_xyz)))

    def snapshot_save(self, path):
        self._ck(self.lib.pbf_snapshot_save(self._h, str(path).encode()))

    def snapshot_load(self, path):
        self._ck(self.lib.pbf_snapshot_load(self._h, str(path).encode()))
        self.n = int(self.lib.pbf_num_particles(self._h))

    def slab_set_bounds(self, x_lo: int, x_hi: int, ghost_layers: int = 2):
        self._ck(self.lib.pbf_slab_set_bounds(self._h, x_lo, x_hi, ghost_layers))

    def slab_split(self, d_mig_left: int, d_mig_right: int, d_ghost_left: int, d_ghost_right: int,
                   max_records: int, d_counts: int):
        self._ck(self.lib.pbf_slab_split(self._h, C.c_void_p(d_mig_left), C.c_void_p(d_mig_right),
                                         C.c_void_p(d_ghost_left), C.c_void_p(d_ghost_right),
                                         max_records, C.c_void_p(d_counts)))

    def slab_split_done(self, counts):
        arr = (C.c_uint32 * 6)(*[int(v) for v in counts])
        self._ck(self.lib.pbf_slab_split_done(self._h, C.byref(arr)))
        self.n = int(self.lib.pbf_num_particles(self._h))

    def slab_append_owned(self, d_left: int, count_left: int, d_right: int = 0, count_right: int = 0):
        self._ck(self.lib.pbf_slab_append_owned(self._h, C.c_void_p(d_left or None), count_left,
                                                C.c_void_p(d_right or None), count_right))
        self.n = int(self.lib.pbf_num_particles(self._h))

    def slab_append_ghosts(self, d_left: int, count_left: int, d_right: int, count_right: int):
        self._ck(self.lib.pbf_slab_append_ghosts(self._h, C.c_void_p(d_left or None), count_left,
                                                 C.c_void_p(d_right or None), count_right))
        self.n = int(self.lib.pbf_num_particles(self._h))

    def slab_pack_halo(self, what: int, d_left: int, d_right: int):
        self._ck(self.lib.pbf_slab_pack_halo(self._h, what, C.c_void_p(d_left or None),
                                             C.c_void_p(d_right or None)))

    def slab_unpack_halo(self, what: int, d_left: int, count_left: int, d_right: int, count_right: int):
        self._ck(self.lib.pbf_slab_unpack_halo(self._h, what, C.c_void_p(d_left or None), count_left,
                                               C.c_void_p(d_right or None), count_right))

    def slab_halo_exchange(self, comm: "RcclComm", what: int, left: int, right: int, d_send_left: int,
                           d_send_right: int, d_recv_left: int, d_recv_right: int):
        self._ck(self.lib.pbf_slab_halo_exchange(self._h, comm._h, what, left, right,
                                                 C.c_void_p(d_send_left or None), C.c_void_p(d_send_right or None),
                                                 C.c_void_p(d_recv_left or None), C.c_void_p(d_recv_right or None)))

    def slab_attach(self, comm: "RcclComm", left: int, right: int, max_records: int = 0):
        self._ck(self.lib.pbf_slab_attach(self._h, comm._h, left, right, max_records))
        self._comm = comm   # keep the communicator alive as long as the ctx uses it

    def slab_set_overlap(self, on: bool):
        """ghost refresh overlapped with the interior lambda sweep (pbf_slab_step; results do not change)"""
        self._ck(self.lib.pbf_slab_set_overlap(self._h, 1 if on else 0))

    def slab_step(self, dt):
        self._ck(self.lib.pbf_slab_step(self._h, C.c_float(dt)))
        self.n = int(self.lib.pbf_num_particles(self._h))

    def slab_step_dem(self, dt, dem: "PbfDemParams"):
        self._ck(self.lib.pbf_slab_step_dem(self._h, C.c_float(dt), C.byref(dem)))
        self.n = int(self.lib.pbf_num_particles(self._h))

    def slab_rebalance(self, planes, x_first: int, n_planes: int, max_move: int):
        """collective; returns (new planes, number of planes moved)"""
        comm = getattr(self, "_comm", None)
        if comm is None:
            raise PbfError(PBF_ERR_INVALID, "pbf_slab_attach has not been called")
        # the C side reads planes[world] and writes planes[1 .. world - 1]: it cannot check the length
        if len(planes) != comm.world + 1:
            raise PbfError(PBF_ERR_INVALID, f"planes must hold world + 1 = {comm.world + 1} entries, got {len(planes)}")
        arr = (C.c_int32 * len(planes))(*[int(v) for v in planes])
        moved = C.c_int32()
        self._ck(self.lib.pbf_slab_rebalance(self._h, arr, int(x_first), int(n_planes), int(max_move),
                                             C.byref(moved)))
        return [int(v) for v in arr], int(moved.value)

    def slab_owned_count(self) -> int:
        return int(self.lib.pbf_slab_owned_count(self._h))

    def slab_export(self, d_records: int, max_records: int) -> int:
        k = C.c_uint32()
        self._ck(self.lib.pbf_slab_export(self._h, C.c_void_p(d_records), max_records, C.byref(k)))
        return int(k.value)

    def selftest_sqrt(self):
        bad, first = C.c_uint64(), C.c_uint32()
        self._ck(self.lib.pbf_selftest_sqrt(self._h, C.byref(bad), C.byref(first)))
        return int(bad.value), int(first.value)

    # -- device-time profile
    def profile_enable(self, on=True):
        self._ck(self.lib.pbf_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._ck(self.lib.pbf_profile_reset(self._h))

    def profile_read(self) -> dict:
        k = int(self.lib.pbf_num_phases())
        ms = (C.c_double * k)()
        calls = (C.c_uint64 * k)()
        self._ck(self.lib.pbf_profile_read(self._h, ms, calls, k))
        return {self.lib.pbf_phase_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(k)}
