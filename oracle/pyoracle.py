"""ctypes door to the CPU oracle (oracle/libpbf_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (fluidsimulation_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from fluidsimulation_amd.binding import (PBF_OK, PbfCollider, PbfParams, PbfStats,
                                         collider_array)

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR / "libpbf_oracle.so"
REF_LIB_PATH = _DIR / "_ref" / "libpbf_ref.so"

_lib = None
_ref = None


def build(force: bool = False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < (_DIR / "pbf_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR)], check=True, capture_output=True)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    lib = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    f3 = C.POINTER(C.c_float)
    lib.orc_default_params.argtypes = [C.POINTER(PbfParams)]
    lib.orc_create.argtypes = [C.POINTER(PbfParams), C.POINTER(vp)]
    lib.orc_destroy.argtypes = [vp]
    lib.orc_set_threads.argtypes = [vp, C.c_int]
    lib.orc_upload.argtypes = [vp, vp, vp, vp, C.c_float, C.c_uint32]
    lib.orc_set_colliders.argtypes = [vp, C.POINTER(PbfCollider), C.c_uint32]
    lib.orc_set_solver_iterations.argtypes = [vp, C.c_int32]
    lib.orc_set_boundary.argtypes = [vp, vp, vp, C.c_float, C.c_uint32]
    lib.orc_get_boundary.argtypes = [vp, vp, vp, vp]
    lib.orc_get_boundary_neighbors.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32]
    lib.orc_set_distance_constraints.argtypes = [vp, vp, C.c_uint32, C.c_int32]
    lib.orc_project_constraints.argtypes = [vp, C.c_float]
    lib.orc_get_constraint_lambdas.argtypes = [vp, vp]
    lib.orc_step_dem.argtypes = [vp, C.c_float, vp]
    lib.orc_set_gravity.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    for name in ("orc_predict_positions", "orc_collision_detection", "orc_compute_correction",
                 "orc_apply_solver_results", "orc_step"):
        getattr(lib, name).argtypes = [vp, C.c_float]
    for name in ("orc_spatial_search", "orc_naive_search", "orc_compute_rest_density",
                 "orc_compute_density", "orc_compute_lambdas", "orc_update_predict_position"):
        getattr(lib, name).argtypes = [vp]
    lib.orc_update_velocity.argtypes = [vp, C.c_float]
    lib.orc_vorticity_omega.argtypes = [vp]
    lib.orc_vorticity_apply.argtypes = [vp, C.c_float]
    lib.orc_xsph.argtypes = [vp]
    lib.orc_set_array.argtypes = [vp, C.c_int, vp]
    lib.orc_get_array.argtypes = [vp, C.c_int, vp]
    lib.orc_download.argtypes = [vp, vp, vp, vp, vp]
    lib.orc_get_scalars.argtypes = [vp, vp, vp, vp]
    lib.orc_get_cells.argtypes = [vp, vp, vp]
    lib.orc_neighbor_total.argtypes = [vp]
    lib.orc_neighbor_total.restype = C.c_uint64
    lib.orc_neighbor_max.argtypes = [vp]
    lib.orc_neighbor_max.restype = C.c_uint32
    lib.orc_get_neighbors.argtypes = [vp, vp, vp, C.c_uint32]
    lib.orc_get_stats.argtypes = [vp, C.POINTER(PbfStats)]
    lib.orc_rest_density.argtypes = [vp]
    lib.orc_rest_density.restype = C.c_float
    lib.orc_num_particles.argtypes = [vp]
    lib.orc_num_particles.restype = C.c_uint32
    lib.orc_poly6_w.argtypes = [C.c_float, C.c_float]
    lib.orc_poly6_w.restype = C.c_float
    lib.orc_spiky_w.argtypes = [C.c_float, C.c_float]
    lib.orc_spiky_w.restype = C.c_float
    lib.orc_poly6_w_gradient.argtypes = [f3, C.c_float, C.c_float, f3]
    lib.orc_poly6_w_gradient.restype = None
    lib.orc_spiky_w_gradient.argtypes = [f3, C.c_float, C.c_float, f3]
    lib.orc_spiky_w_gradient.restype = None
    lib.orc_hash_value.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.orc_hash_value.restype = C.c_uint32
    lib.orc_flooring.argtypes = [f3, C.c_float, C.POINTER(C.c_int32)]
    lib.orc_flooring.restype = None
    lib.orc_point_plane.argtypes = [f3, f3, C.c_float]
    lib.orc_point_aabb.argtypes = [f3, f3, f3]
    lib.orc_setup_sim_params.argtypes = [f3, f3, f3]
    lib.orc_setup_sim_params.restype = None
    _lib = lib
    return lib


def load_ref():
    """The reference's own leaf sources compiled into oracle/_ref (None if absent)."""
    global _ref
    if _ref is not None:
        return _ref
    if not REF_LIB_PATH.exists():
        return None
    lib = C.CDLL(str(REF_LIB_PATH))
    f3 = C.POINTER(C.c_float)
    lib.ref_poly6_w.argtypes = [C.c_float, C.c_float]
    lib.ref_poly6_w.restype = C.c_float
    lib.ref_spiky_w.argtypes = [C.c_float, C.c_float]
    lib.ref_spiky_w.restype = C.c_float
    lib.ref_poly6_w_gradient.argtypes = [f3, C.c_float, C.c_float, f3]
    lib.ref_poly6_w_gradient.restype = None
    lib.ref_spiky_w_gradient.argtypes = [f3, C.c_float, C.c_float, f3]
    lib.ref_spiky_w_gradient.restype = None
    lib.ref_point_plane.argtypes = [f3, f3, C.c_float]
    lib.ref_point_aabb.argtypes = [f3, f3, f3]
    lib.ref_glm_distance.argtypes = [f3, f3]
    lib.ref_glm_distance.restype = C.c_float
    lib.ref_glm_dot.argtypes = [f3, f3]
    lib.ref_glm_dot.restype = C.c_float
    lib.ref_glm_pow.argtypes = [C.c_double, C.c_int]
    lib.ref_glm_pow.restype = C.c_double
    if hasattr(lib, "ref_glm_distance2_many"):   # doors of tests/golden/make_ref_loops.py
        lib.ref_glm_distance2.argtypes = [f3, f3]
        lib.ref_glm_distance2.restype = C.c_float
        for name in ("ref_glm_distance2_many", "ref_glm_distance_many"):
            getattr(lib, name).argtypes = [f3, f3, C.c_uint32, f3]
            getattr(lib, name).restype = None
        lib.ref_poly6_w_many.argtypes = [f3, C.c_float, C.c_uint32, f3]
        lib.ref_poly6_w_many.restype = None
        lib.ref_poly6_w_gradient_many.argtypes = [f3, f3, f3, C.c_float, C.c_uint32, f3]
        lib.ref_poly6_w_gradient_many.restype = None
        lib.ref_glm_scaled_dot_self_many.argtypes = [C.c_float, f3, C.c_uint32, f3]
        lib.ref_glm_scaled_dot_self_many.restype = None
        lib.ref_glm_scale_many.argtypes = [f3, f3, C.c_uint32, f3]
        lib.ref_glm_scale_many.restype = None
    # Particle.cpp / Constraints.cpp / ConstraintSolver.cpp doors (present when the image held
    # cuda_runtime.h at build time; ref_has_particles() says which build this is)
    lib.ref_has_particles.restype = C.c_int
    if lib.ref_has_particles():
        u32p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        lib.ref_particleset_collide.argtypes = [C.c_uint32, f3, f3, f3, f3, C.c_float, C.c_uint32, i32p, f3, f3, f3]
        lib.ref_particleset_collide.restype = C.c_int
        lib.ref_particleset_update.argtypes = [C.c_uint32, f3, f3, f3, C.c_float]
        lib.ref_particleset_update.restype = None
        lib.ref_solve_distance_constraints.argtypes = [C.c_int32, C.c_float, C.c_uint32, C.c_uint32, f3, f3,
                                                       C.c_uint32, u32p, u32p, f3, f3, f3, f3]
        lib.ref_solve_distance_constraints.restype = C.c_int
    _ref = lib
    return lib


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_params() -> PbfParams:
    p = PbfParams()
    load().orc_default_params(C.byref(p))
    return p


class Oracle:
    """One orc_ctx; method names follow the reference's Simulation member functions."""

    def __init__(self, params: PbfParams, threads: int = 8):
        self.lib = load()
        self.params = params.copy()
        self._h = C.c_void_p()
        rc = self.lib.orc_create(C.byref(self.params), C.byref(self._h))
        if rc != PBF_OK:
            raise RuntimeError(f"orc_create failed: {rc}")
        self.lib.orc_set_threads(self._h, threads)
        self.n = 0

    def set_threads(self, threads: int):
        self.lib.orc_set_threads(self._h, int(threads))

    def close(self):
        if getattr(self, "_h", None) and self._h:
            self.lib.orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != PBF_OK:
            raise RuntimeError(f"oracle error {rc}")

    def upload(self, pos, vel=None, mass=None, uniform_mass: float = 1.0):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        n = pos.shape[0]
        vel = None if vel is None else np.ascontiguousarray(vel, dtype=np.float32).reshape(n, 3)
        mass = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32).reshape(n)
        self._ck(self.lib.orc_upload(self._h, _ptr(pos), _ptr(vel), _ptr(mass),
                                     C.c_float(uniform_mass), n))
        self.n = n

    def set_colliders(self, cols):
        self._ck(self.lib.orc_set_colliders(self._h, collider_array(cols), len(cols)))

    def set_boundary(self, pos, mass=None, uniform_mass: float = 1.0):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        nb = pos.shape[0]
        mass = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32).reshape(nb)
        self._ck(self.lib.orc_set_boundary(self._h, _ptr(pos), _ptr(mass), C.c_float(uniform_mass), nb))
        self.nb = nb

    def boundary(self):
        """(volume, density, lambda) of the boundary particles"""
        out = [np.zeros(self.nb, np.float32) for _ in range(3)]
        self._ck(self.lib.orc_get_boundary(self._h, *[_ptr(a) for a in out]))
        return tuple(out)

    def boundary_neighbors(self, stride_f=64, stride_b=64):
        """(fluid counts, fluid->boundary lists, boundary counts, boundary->fluid lists)"""
        fc, fl = np.zeros(self.n, np.uint32), np.zeros((self.n, stride_f), np.uint32)
        bc, bl = np.zeros(self.nb, np.uint32), np.zeros((self.nb, stride_b), np.uint32)
        self._ck(self.lib.orc_get_boundary_neighbors(self._h, _ptr(fc), _ptr(fl), stride_f,
                                                     _ptr(bc), _ptr(bl), stride_b))
        return fc, fl, bc, bl

    def set_distance_constraints(self, cons, mode=1):
        from fluidsimulation_amd.binding import constraint_array
        arr, k = constraint_array(cons)
        self._ck(self.lib.orc_set_distance_constraints(self._h, C.byref(arr), k, mode))
        self.ncons = k

    def project_constraints(self, dt):
        self._ck(self.lib.orc_project_constraints(self._h, C.c_float(dt)))

    def constraint_lambdas(self):
        lam = np.zeros(max(1, getattr(self, "ncons", 0)), np.float32)
        self._ck(self.lib.orc_get_constraint_lambdas(self._h, _ptr(lam)))
        return lam[:getattr(self, "ncons", 0)]

    def step_dem(self, dt, dem):
        self._ck(self.lib.orc_step_dem(self._h, C.c_float(dt), C.byref(dem)))

    def set_solver_iterations(self, it):
        self._ck(self.lib.orc_set_solver_iterations(self._h, it))

    def set_gravity(self, gx, gy, gz):
        self._ck(self.lib.orc_set_gravity(self._h, gx, gy, gz))

    def predict_positions(self, dt):
        self._ck(self.lib.orc_predict_positions(self._h, C.c_float(dt)))

    def collision_detection(self, dt):
        self._ck(self.lib.orc_collision_detection(self._h, C.c_float(dt)))

    def spatial_search(self):
        self._ck(self.lib.orc_spatial_search(self._h))

    def naive_search(self):
        self._ck(self.lib.orc_naive_search(self._h))

    def compute_rest_density(self):
        self._ck(self.lib.orc_compute_rest_density(self._h))

    def compute_density(self):
        self._ck(self.lib.orc_compute_density(self._h))

    def compute_lambdas(self):
        self._ck(self.lib.orc_compute_lambdas(self._h))

    def compute_correction(self, dt):
        self._ck(self.lib.orc_compute_correction(self._h, C.c_float(dt)))

    def update_predict_position(self):
        self._ck(self.lib.orc_update_predict_position(self._h))

    def apply_solver_results(self, dt):
        self._ck(self.lib.orc_apply_solver_results(self._h, C.c_float(dt)))

    def step(self, dt):
        self._ck(self.lib.orc_step(self._h, C.c_float(dt)))

    def update_velocity(self, dt):
        self._ck(self.lib.orc_update_velocity(self._h, C.c_float(dt)))

    def vorticity_omega(self):
        self._ck(self.lib.orc_vorticity_omega(self._h))

    def vorticity_apply(self, dt):
        self._ck(self.lib.orc_vorticity_apply(self._h, C.c_float(dt)))

    def xsph(self):
        self._ck(self.lib.orc_xsph(self._h))

    ARRAYS = {"pos": (0, 3), "vel": (1, 3), "pred": (2, 3), "lambda": (3, 1), "omega": (4, 3)}

    def set_array(self, name, data):
        which, w = self.ARRAYS[name]
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(self.n, w)
        self._ck(self.lib.orc_set_array(self._h, which, _ptr(data)))

    def get_array(self, name):
        which, w = self.ARRAYS[name]
        out = np.empty((self.n, w), np.float32)
        self._ck(self.lib.orc_get_array(self._h, which, _ptr(out)))
        return out

    def download(self):
        n = self.n
        out = [np.empty((n, 3), np.float32) for _ in range(4)]
        self._ck(self.lib.orc_download(self._h, *[_ptr(a) for a in out]))
        return tuple(out)  # pos, vel, pred, new

    def scalars(self):
        n = self.n
        out = [np.empty(n, np.float32) for _ in range(3)]
        self._ck(self.lib.orc_get_scalars(self._h, *[_ptr(a) for a in out]))
        return tuple(out)  # density, C, lambda

    def cells(self):
        xyz = np.empty((self.n, 3), np.int32)
        h = np.empty(self.n, np.uint32)
        self._ck(self.lib.orc_get_cells(self._h, _ptr(xyz), _ptr(h)))
        return xyz, h

    def neighbors(self, stride: int | None = None):
        mx = int(self.lib.orc_neighbor_max(self._h))
        stride = max(1, mx) if stride is None else stride
        counts = np.empty(self.n, np.uint32)
        nb = np.zeros((self.n, stride), np.uint32)
        self._ck(self.lib.orc_get_neighbors(self._h, _ptr(counts), _ptr(nb), stride))
        return counts, nb

    def stats(self) -> PbfStats:
        s = PbfStats()
        self._ck(self.lib.orc_get_stats(self._h, C.byref(s)))
        return s

    def rest_density(self) -> float:
        return float(self.lib.orc_rest_density(self._h))
