/*
 * pbf.h -- C ABI of the MI355X-native Position-Based-Fluids sub-step.
 *
 * This is the drop-in boundary.  The reference has no plugin/FFI
 * API; the cut is the level its own CUDA path already exposes as free host
 * functions over raw device pointers (/root/reference/cuda_simulation.cuh:15-90)
 * beneath the C++ class surface Simulation / ParticleSystem / NeighborSearch /
 * ConstraintSolver (Simulation.h:50-77).  Each entry point below cites the
 * reference interface it replaces.  Semantics follow the reference's OpenMP CPU
 * path (Simulation::Step, Simulation.cpp:80-138), not its CUDA kernels.
 *
 * Conventions
 *   - every function returns an int status (PBF_OK == 0); nothing aborts/exits
 *     (the reference's getLastCudaError() exits the process, cuda_simulation.cu:393)
 *   - one pbf_ctx == one device + one HIP stream; a ctx is not thread safe,
 *     distinct ctxs are independent (no __constant__ singleton as in
 *     cuda_simulation.cuh:13)
 *   - host arrays are caller-owned, interleaved xyz float triples (the layout of
 *     std::vector<glm::vec3>, Particle.h:56-61) indexed by ORIGINAL particle id
 *   - step entry points are asynchronous on the ctx stream; pbf_sync / download /
 *     get_* synchronise
 *
 * Plain C, no torch / HIP types in any signature.
 */
#ifndef PBF_H_
#define PBF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBF_ABI_VERSION 2

/* status codes */
enum {
    PBF_OK = 0,
    PBF_ERR_INVALID = 1,       /* bad argument / bad state                             */
    PBF_ERR_HIP = 2,           /* a HIP runtime call failed (see pbf_last_error)        */
    PBF_ERR_CAPACITY = 3,      /* n exceeds ctx capacity                                */
    PBF_ERR_NEIGHBOR_OVERFLOW = 4, /* a particle has more than max_neighbors neighbours */
    PBF_ERR_NO_DEVICE = 5,     /* no HIP device / extension cannot run                  */
    PBF_ERR_NOT_INITIALIZED = 6 /* Simulation::Step returns false (Simulation.cpp:82)   */
};

/* collider kinds: values of Collider::ColliderTypes (Collider.h:10-17) */
enum {
    PBF_COLLIDER_AABB = 2,
    PBF_COLLIDER_PLANE = 4,
    /* not a Collider class of the reference: the hard box its CUDA path bakes into integrate_pbd_d
     * (cuda_simulation.cu:473-511; x, z in [-1, 1], y >= 0, boundary_damping 0.1 in the README scene) */
    PBF_COLLIDER_HARD_BOX = 100
};

/* One collider, copied by value into the ctx (the reference borrows raw
 * Collider* from the app, Simulation.cpp:252-256).
 *   PLANE: a = normal, d = offset; hit iff dot(a, p*) <= d   (CollisionDetection.cpp:62-67)
 *   AABB : a = min, b = max;      hit iff strictly inside    (CollisionDetection.cpp:29-34)
 *   HARD_BOX: a = min, b = max (b[1], the top, is not used: the box is open), d = boundary damping.  The CUDA
 *          path's wall response, in its order (x >= b[0], x <= a[0], z >= b[2], z <= a[2], y <= a[1]): the
 *          predicted coordinate is clamped to the wall, the velocity component is pointed back inside
 *          (-+|v|) and the WHOLE velocity vector is multiplied by d; no re-prediction.  Parity unpinned
 *          (CUDA-only code); checked against this repo's oracle. */
typedef struct pbf_collider {
    int32_t type;
    float a[3];
    float b[3];
    float d;
} pbf_collider;

#define PBF_MAX_COLLIDERS 16

/* rest-density modes */
enum {
    PBF_RHO0_FIRST_FRAME_MEAN = 0, /* Simulation::ComputeRestDensity, Simulation.cpp:273-300 */
    PBF_RHO0_FIXED = 1             /* SetupSimParams rho0 = 1000, Simulation.cpp:318         */
};

/* gradient kernel used for lambda / delta-p */
enum {
    PBF_GRAD_POLY6 = 0, /* what the reference actually calls (Simulation.cpp:657,665,706) */
    PBF_GRAD_SPIKY = 1  /* SPHKernel::Spiky_W_Gradient (SPHKernel.h:78-97), never called
                           by the reference                                               */
};

/* All solver parameters.  Replaces SimParams (Simulation.h:22-48) + SimWorldDesc
 * (Simulation.h:14-20) + the constants hard-coded in Simulation::Step.  Defaults
 * (pbf_default_params) reproduce the reference CPU path exactly. */
typedef struct pbf_params {
    uint32_t abi_version;       /* PBF_ABI_VERSION                                         */
    uint32_t capacity;          /* max particles this ctx can hold (owned + ghosts)        */
    uint32_t max_neighbors;     /* rows of the neighbour table (m_search_cache bound)      */
    int32_t  grid_dim[3];       /* wrap-around cell grid; 64^3 at Simulation.cpp:46        */
    int32_t  grid_origin_cell[3]; /* Flooring cell that maps to grid index (0,0,0); the CUDA
                                   path's world_origin (Simulation.cpp:340) in cell units.
                                   Only the internal key depends on it, never the results  */
    float    grid_spacing;      /* m_grid_spacing = 1.f, NeighborSearch.cpp:17             */
    float    effective_radius;  /* h; effective_radius = 1.f, Simulation.cpp:88            */
    float    gravity[3];        /* (0, m_world_desc.gravity, 0), Simulation.cpp:16,592     */
    float    epsilon;           /* 1.0e-6f, Simulation.cpp:644                             */
    float    restitution;       /* 0.99f, Particle.cpp:333                                 */
    double   scorr_k;           /* 0.1 (scorr = -0.1), Simulation.cpp:698                  */
    float    scorr_dq;          /* 0.3f, Simulation.cpp:700                                */
    int32_t  scorr_dt_power;    /* 1 on the CPU path (Simulation.cpp:702); the CUDA path
                                   uses 3 (cuda_simulation.cu:973)                         */
    int32_t  rest_density_mode; /* PBF_RHO0_*                                              */
    float    rest_density;      /* value for PBF_RHO0_FIXED                                */
    int32_t  solver_iterations; /* ConstraintSolver::m_solver_iteration, default 1
                                   (ConstraintSolver.cpp:4)                                */
    int32_t  gradient_kernel;   /* PBF_GRAD_*                                              */
    /* ---- build-side extensions, no reference counterpart (parity unpinned) ---- */
    int32_t  container_enabled; /* per-iteration clamp of new positions into a tank       */
    float    container_min[3];
    float    container_max[3];
    float    xsph_c;            /* XSPH viscosity coefficient (0 = off)                    */
    float    vorticity_eps;     /* vorticity confinement strength (0 = off)                */
} pbf_params;

/* per-step health metrics (SURVEY.md section 5 "Metrics") */
typedef struct pbf_stats {
    uint32_t n;
    uint32_t nonfinite;          /* particles with a non-finite position                   */
    uint32_t neighbor_overflow;  /* particles that hit max_neighbors in the last search; if that
                                    one was clean, rows cut in ANY sub-step since the upload   */
    uint32_t max_neighbor_count;
    double   mean_neighbor_count;
    double   mean_abs_C;         /* mean |rho_i/rho0 - 1| of the last solver iteration      */
    double   max_abs_C;
    double   mean_C;
    double   com[3];             /* mass-weighted centre of mass                            */
    double   max_speed;
    float    rest_density;
    uint32_t row_format_particles[3]; /* particles whose neighbour row is stored as 32-bit slots /
                                    16-bit offsets from the particle / 16-bit window codes (DESIGN.md
                                    section 2; no reference counterpart)                       */
} pbf_stats;

typedef struct pbf_ctx pbf_ctx;

/* fills *p with the reference CPU path's constants */
int pbf_default_params(pbf_params* p);

/* Simulation::Initialize + ParticleSystem::SetupCUDAMemory + NeighborSearch::InitializeCUDA
 * (Simulation.cpp:42-72, ParticleSystem.cpp:116-348, NeighborSearch.cpp:41-74):
 * allocates every device array once.  device < 0 selects the current device. */
int pbf_create(const pbf_params* params, int device, pbf_ctx** out);
/* ParticleSystem::Release + NeighborSearch::Release (ParticleSystem.cpp:59-97, NeighborSearch.cpp:76-98) */
int pbf_destroy(pbf_ctx* ctx);
/* message of the last failure on this ctx (never NULL) */
const char* pbf_last_error(const pbf_ctx* ctx);

/* ParticleSystem::AllocateParticles + the cudaMemcpy block (ParticleSystem.cpp:99-103,188-260):
 * pos/vel are n x 3 floats, mass n floats (NULL = uniform_mass for all).  Resets the
 * first-frame flag (Simulation.cpp:17). */
int pbf_upload(pbf_ctx* ctx, const float* pos, const float* vel, const float* mass,
               float uniform_mass, uint32_t n);
/* same, from device-resident buffers (inputs already in HBM) */
int pbf_upload_device(pbf_ctx* ctx, const void* d_pos, const void* d_vel, const void* d_mass,
                      float uniform_mass, uint32_t n);
/* Simulation::AddCollider (Simulation.cpp:252-256); replaces the whole list, order kept */
int pbf_set_colliders(pbf_ctx* ctx, const pbf_collider* colliders, uint32_t count);
/* ConstraintSolver::setSolverIteration (ConstraintSolver.cpp:25-28) */
int pbf_set_solver_iterations(pbf_ctx* ctx, int32_t iterations);
/* Simulation::setGravity (Simulation.cpp:307-310) */
int pbf_set_gravity(pbf_ctx* ctx, float gx, float gy, float gz);
/* Which entry formats the fluid neighbour table may use from the next pbf_build_grid on (no
 * reference counterpart; results never depend on it): bit 0 = 32-bit slots (always allowed), bit 1 =
 * 16-bit offsets from the particle, bit 2 = 16-bit window codes.  Default
 * PBF_ROWS_ALL; the narrower masks exist for measurements and to put small test scenes through the
 * format, or the build path, only large scenes would pick. */
#define PBF_ROWS_ABS32 1u
#define PBF_ROWS_REL16 2u
#define PBF_ROWS_WIN16 4u
#define PBF_ROWS_ALL   7u
#define PBF_ROWS_UNSTAGED 8u /* test lever: build every group on the neighbour kernel's global path (the
                                path of groups whose candidate windows exceed the LDS budget) */
int pbf_set_row_formats(pbf_ctx* ctx, uint32_t mask);

/* Simulation::PredictPositions + CollisionDetection (Simulation.cpp:578-599,723-751;
 * ParticleSet::TestCollision/OnCollision, Particle.cpp:217-338); replaces integratePBD
 * (cuda_simulation.cuh:21-27) */
int pbf_predict(pbf_ctx* ctx, float dt);
/* NeighborSearch::SpatialSearch (NeighborSearch.cpp:131-236); replaces calculate_hash +
 * sort_particles + reorder_data (cuda_simulation.cuh:31-47): cell id -> counting sort ->
 * prefix scan -> reorder -> neighbour table */
int pbf_build_grid(pbf_ctx* ctx);
/* Simulation::ComputeRestDensity (Simulation.cpp:273-300); no-op unless
 * rest_density_mode == PBF_RHO0_FIRST_FRAME_MEAN and this is the first frame */
int pbf_compute_rest_density(pbf_ctx* ctx);
/* iterations x { ComputeDensity, ComputeLambdas, ComputeSPHParticlesCorrection,
 * UpdatePredictPosition } (Simulation.cpp:112-118,607-721); replaces solve_sph_fluid
 * (cuda_simulation.cuh:69-79).  iterations <= 0 uses the ctx's solver_iterations. */
int pbf_solve(pbf_ctx* ctx, int32_t iterations, float dt);
/* Simulation::ApplySolverResults -> ParticleSet::Update (Simulation.cpp:787-791,
 * Particle.cpp:207-215), then XSPH viscosity / vorticity confinement when enabled */
int pbf_finalize(pbf_ctx* ctx, float dt);
/* Simulation::Step (Simulation.cpp:80-138) = predict, build_grid, [rest density],
 * solve, finalize */
int pbf_step(pbf_ctx* ctx, float dt);
/* blocks until the ctx stream is idle; returns the sticky device-side error if any
 * (PBF_ERR_NEIGHBOR_OVERFLOW: rows cut at max_neighbors in this or an earlier sub-step; or a cell key
 * that held more than max(q^3 x (max_neighbors + 1), 2^34 / n) particles, q = ceil(sqrt(3)
 * grid_spacing / effective_radius) -- a pile in one cell, whose rows overflow for any arrangement, or
 * a grid far too small for the scene: the sort and the neighbour build, O(occupancy) per particle,
 * are skipped from that sub-step on and the state is undefined until the next upload) */
int pbf_sync(pbf_ctx* ctx);

/* copies state back in ORIGINAL id order; any pointer may be NULL.
 * pos/vel/pred: n x 3 (m_positions, m_velocity, m_predict_positions == m_new_positions) */
int pbf_download(pbf_ctx* ctx, float* pos, float* vel, float* pred);
/* m_density, m_C, m_lambda (Particle.h:65-67) in original id order; NULLs allowed */
int pbf_get_scalars(pbf_ctx* ctx, float* density, float* C, float* lambda);
/* result of the last pbf_build_grid, for the bit-exact checks:
 *   cell_xyz[n*3] : NeighborSearch::Flooring of every particle, by original id
 *   cell_key[n]   : wrapped linear cell id, by original id
 *   sorted_id[n]  : original id of the particle stored at each sorted slot
 *   cell_start[num_cells+1] : exclusive prefix sum of cell occupancy */
int pbf_get_cells(pbf_ctx* ctx, int32_t* cell_xyz, uint32_t* cell_key, uint32_t* sorted_id,
                  uint32_t* cell_start);
/* NeighborSearch::FetchNeighbors (NeighborSearch.cpp:238-241) for all particles:
 * counts[n] and, for particle i, neighbors[i*stride + k] = ORIGINAL id of its k-th
 * neighbour in the reference's m_search_cache order.  stride >= max count. */
int pbf_get_neighbors(pbf_ctx* ctx, uint32_t* counts, uint32_t* neighbors, uint32_t stride);
int pbf_get_stats(pbf_ctx* ctx, pbf_stats* out);
/* current rest density (m_rest_density) */
int pbf_get_rest_density(pbf_ctx* ctx, float* out);
uint32_t pbf_num_particles(const pbf_ctx* ctx);

/* the two halves of one solver iteration (pbf_solve == iterations x {lambda, delta}) and the
 * stages of pbf_finalize (0 = ParticleSet::Update, 1 = vorticity omega, 2 = vorticity apply,
 * 3 = XSPH), exposed so that a slab-decomposed caller can refresh ghost data in between */
int pbf_solve_lambda(pbf_ctx* ctx, float dt);
int pbf_solve_delta(pbf_ctx* ctx, float dt);
int pbf_finalize_stage(pbf_ctx* ctx, int32_t stage, float dt);

/* ---- render / export hand-off ------------------------------------
 * pbf_export_positions: what ParticleSystem::UpdateGLBUfferData uploads and Renderer::
 * RenderParticles draws (ParticleSystem.cpp:372-389, Renderer.cpp:121-127) and what the
 * reference's StepCUDA writes through the mapped VBO (Simulation.cpp:160-168): N tightly packed
 * float3 positions in particle-index order, written into caller-owned DEVICE memory (a mapped
 * GL buffer, a torch tensor ...) on the ctx stream, asynchronous. */
int pbf_export_positions(pbf_ctx* ctx, void* d_xyz);
/* raw SoA snapshot / restart file (no reference counterpart): 64-byte header ("PBFSNAP1", n,
 * flags, rest density in force, uniform mass), then pos[n][3], vel[n][3] and, with
 * per-particle masses, mass[n] -- fp32, particle-index order.  A run resumed with
 * pbf_snapshot_load continues bit-identically (the sorted order is a function of the state). */
int pbf_snapshot_save(pbf_ctx* ctx, const char* path);
int pbf_snapshot_load(pbf_ctx* ctx, const char* path);

/* ---- boundary-particle walls --------------------------------------
 * Replaces Simulation::InitializeBoundaryCudaData (Simulation.cpp:492-526: hash -> sort ->
 * reorder -> compute_boundary_volume, cuda_simulation.cuh:49-53) and the boundary arguments of
 * solve_sph_fluid (cuda_simulation.cuh:69-79).  The reference couples boundary particles only
 * in its CUDA kernels (K5, K7, K9 and the boundary halves of K6/K8/K10); here that coupling is
 * evaluated in the CPU path's arithmetic and frozen-list semantics (PARITY UNPINNED: checked
 * against this repo's oracle only).  pos: nb x 3 floats, mass: nb floats or NULL =
 * uniform_mass.  The set is static and survives pbf_upload; nb = 0 removes it.
 * max_boundary_neighbors = rows of the three boundary tables (0 = max_neighbors).  Allocates;
 * synchronises.  In slab mode every rank sets the wall particles within reach of its slab and
 * ghost planes (the whole set is fine: volumes need each wall particle's wall neighbours). */
int pbf_set_boundary(pbf_ctx* ctx, const float* pos, const float* mass, float uniform_mass,
                     uint32_t nb, uint32_t max_boundary_neighbors);
uint32_t pbf_num_boundary_particles(const pbf_ctx* ctx);
/* m_volume, m_density, m_lambda of the boundary ParticleSet (Particle.h:65-68) in boundary
 * index order; NULLs allowed */
int pbf_get_boundary(pbf_ctx* ctx, float* volume, float* density, float* lambda);
/* the fluid->boundary and boundary->fluid rows of the last pbf_build_grid, by original index
 * (as pbf_get_neighbors) */
int pbf_get_boundary_neighbors(pbf_ctx* ctx, uint32_t* fluid_counts, uint32_t* fluid_to_boundary,
                               uint32_t stride_f, uint32_t* boundary_counts,
                               uint32_t* boundary_to_fluid, uint32_t stride_b);

/* ---- PBD granular / DEM solver -------------------------------------
 * Replaces solve_pbd_dem (cuda_simulation.cuh:81-90, cuda_simulation.cu:2158-2225; its only call
 * site is commented out at Simulation.cpp:202-213): `iterations` Jacobi passes of the
 * non-penetration correction (compute_distance_correction, :1867-1926: every particle within one
 * diameter, fluid set then boundary set, 27 cells in z, y, x order around the particle's CURRENT
 * cell, bucket membership frozen at the sort), one Coulomb friction pass
 * (compute_friction_correction, :2084-2156) and the velocity update (finalize_correction,
 * :1491-1518).  PARITY UNPINNED: CUDA-only dead code in the reference; checked against this
 * repo's CPU restatement in plain fp32. */
typedef struct pbf_dem_params {
    float   particle_radius;     /* SimParams.particle_radius, Simulation.cpp:326,335 */
    float   static_friction;     /* 1.0f,  Simulation.cpp:349 */
    float   kinematic_friction;  /* 0.75f, Simulation.cpp:350 */
    int32_t iterations;          /* `iterations` of StepCUDA (3, Simulation.cpp:148) */
} pbf_dem_params;
/* one sub-step: pbf_predict (+ colliders), counting sort (no neighbour table), the DEM solve,
 * velocity update.  Boundary particles set with pbf_set_boundary take part. */
int pbf_step_dem(pbf_ctx* ctx, float dt, const pbf_dem_params* dem);

/* ---- generic constraint projection ---------------------------------
 * DistanceConstraint::SolvePBDConstraint / SolveXPBDConstraint (Constraints.cpp:61-120) under
 * ConstraintSolver::SolvePBDConstraints / SolveXPBDConstraints (ConstraintSolver.cpp:35-72):
 * `solver_iterations` Gauss-Seidel sweeps over the list IN LIST ORDER on the new positions,
 * between the density iterations and the velocity update (Simulation::ProjectConstraints,
 * Simulation.cpp:121,775-780).  On the device the list is cut into levels of constraints that
 * share no particle (a constraint's level = 1 + the highest level of an earlier constraint on
 * either of its particles), executed level by level: every particle sees its constraints in list
 * order, so the result is the sequential Gauss-Seidel result bit for bit.  Particles are
 * addressed by ORIGINAL index; stiffness is clamped to [0, 1] (Constraint::setStiffness).
 * Slab mode (set the list after pbf_slab_set_bounds, the same list on every rank): the indices are the
 * GLOBAL particle ids the slab's records carry; pbf_slab_step projects the list level by level between the
 * solver iterations and the velocity update -- a constraint on every rank that holds both endpoints (owned
 * or ghost), the ghost positions refreshed from their owners after every level (one exchange per level and
 * solver iteration) -- bit-identical with one ctx over the whole scene.  A constraint whose endpoints are
 * further apart along x than the ghost planes reach cannot be projected by the owner of an endpoint: it is
 * counted on the device and the next pbf_slab_step fails on EVERY rank (PBF_ERR_INVALID).
 * pbf_project_constraints itself refuses a slab ctx (it cannot refresh the ghost copies). */
enum { PBF_CONSTRAINTS_PBD = 0, PBF_CONSTRAINTS_XPBD = 1 };  /* PBD_MODE, ConstraintSolver.h */
typedef struct pbf_distance_constraint {
    uint32_t p0, p1;      /* m_particles[0], m_particles[1] */
    float    rest_length; /* m_rest_length */
    float    stiffness;   /* m_stiffness  (PBD)  */
    float    compliance;  /* m_compliance (XPBD) */
} pbf_distance_constraint;
/* Simulation::AddStaticConstraints (Simulation.cpp:258-266): replaces the list; count = 0 clears */
int pbf_set_distance_constraints(pbf_ctx* ctx, const pbf_distance_constraint* list, uint32_t count,
                                 int32_t mode);
/* Simulation::ProjectConstraints: part of pbf_step when a list is set; callable on its own
 * between pbf_solve and pbf_finalize */
int pbf_project_constraints(pbf_ctx* ctx, float dt);
/* XPBD multipliers m_lambda after the last projection, list order; levels = device schedule depth */
int pbf_get_constraint_lambdas(pbf_ctx* ctx, float* lambdas, uint32_t* levels);

/* ---- slab decomposition over the GPUs of one node.  No counterpart in the
 * reference, which is single-GPU (cuda_simulation.cu has no multi-device code).  A ctx owns
 * the cells whose x index is in [x_lo, x_hi); neighbours exchange particle records of
 * PBF_RECORD_FLOATS floats: pos.xyz, mass, vel.xyz, id (as bits), pred.xyz, 0.  All buffers
 * are DEVICE pointers owned by the caller (e.g. torch tensors handed to RCCL). */
#define PBF_RECORD_FLOATS 12
enum { PBF_HALO_PRED_LAMBDA = 0, PBF_HALO_VELOCITY = 1, PBF_HALO_OMEGA = 2, PBF_HALO_POSITION = 3 };
/* run this ctx's kernels on the caller's HIP stream (hipStream_t) instead of its own */
int pbf_set_stream(pbf_ctx* ctx, void* hip_stream);
/* INT32_MIN / INT32_MAX mark an open end; requires PBF_RHO0_FIXED.  ghost_layers = number of
 * boundary cell planes sent to each neighbour: 1 is enough for the neighbour sweeps; with 2 the
 * ghosts' lambda of a solver iteration is computed locally (bit-identical) and one halo
 * refresh per iteration is saved.  A slab with a neighbour on both sides must be at least
 * 2 * ghost_layers cell planes wide (PBF_ERR_INVALID otherwise): a migrant arriving from one side
 * must not already lie in the boundary planes that face the other neighbour */
int pbf_slab_set_bounds(pbf_ctx* ctx, int32_t x_lo, int32_t x_hi, int32_t ghost_layers);
/* after pbf_predict, asynchronous, one pass over the slab: drops last step's ghosts; keeps the
 * particles whose predicted cell is still owned; packs the others as migrant records into
 * d_mig_left / d_mig_right; packs the stayers of the boundary cell planes ([x_lo, x_lo+L) for the
 * left neighbour, [x_hi-L, x_hi) for the right one) as ghost records into d_ghost_left /
 * d_ghost_right and remembers them for the halo refreshes.  d_counts (device, 6 x uint32:
 * {stay, too_far, mig_left, ghost_left, mig_right, ghost_right}) is written on the ctx stream, so the
 * caller can forward [2..3] / [4..5] to its neighbours before anyone synchronises.  The migrant
 * buffers must stay untouched until pbf_slab_append_ghosts has run. */
int pbf_slab_split(pbf_ctx* ctx, void* d_mig_left, void* d_mig_right, void* d_ghost_left,
                   void* d_ghost_right, uint32_t max_records, void* d_counts);
/* host bookkeeping once the caller has read d_counts back (the one host synchronisation of a
 * sub-step is the caller's).  Fails when a buffer overflowed or when too_far != 0: a particle
 * left its slab by more than ghost_layers cell planes in one sub-step, which the one-round
 * exchange (leavers stay behind as ghosts) does not cover. */
int pbf_slab_split_done(pbf_ctx* ctx, const uint32_t counts[6]);
/* appends two record buffers as owned particles: the migrants received from the left / right
 * neighbour (which also join that side's boundary list: their sender keeps them as ghosts), or
 * the initial population */
int pbf_slab_append_owned(pbf_ctx* ctx, const void* d_from_left, uint32_t count_left,
                          const void* d_from_right, uint32_t count_right);
/* appends the ghost records received from the left / right neighbour, each followed by this
 * slab's own migrants to that side (re-read from the pbf_slab_split buffers): ghosts of side s =
 * neighbour's boundary-plane particles ++ particles that just left towards s */
int pbf_slab_append_ghosts(pbf_ctx* ctx, const void* d_from_left, uint32_t count_left,
                           const void* d_from_right, uint32_t count_right);
/* after pbf_build_grid: one float4 per boundary-plane particle / per ghost, in pack order */
int pbf_slab_pack_halo(pbf_ctx* ctx, int32_t what, void* d_left, void* d_right);
int pbf_slab_unpack_halo(pbf_ctx* ctx, int32_t what, const void* d_from_left, uint32_t count_left,
                         const void* d_from_right, uint32_t count_right);
/* all owned particles as records, arbitrary order (synchronises) */
int pbf_slab_export(pbf_ctx* ctx, void* d_records, uint32_t max_records, uint32_t* count);

/* ---- RCCL transport of the slab exchange (no counterpart in the reference).
 * One pbf_comm = one rank of an RCCL communicator; RCCL is bound at run time (librccl.so.1), so
 * nothing here is needed on a single GPU.  Bootstrap: rank 0 calls pbf_comm_unique_id and hands
 * the PBF_COMM_ID_BYTES bytes to every rank by any means (torch.distributed, a file, MPI);
 * every rank then calls pbf_comm_create (collective: returns when all `world` ranks joined). */
#define PBF_COMM_ID_BYTES 128
typedef struct pbf_comm pbf_comm;
typedef struct pbf_p2p {
    int32_t  send;   /* 1 = ncclSend, 0 = ncclRecv */
    int32_t  peer;   /* rank */
    void*    buf;    /* device pointer */
    uint64_t bytes;  /* 0 = skipped on both ends (sizes are agreed beforehand) */
} pbf_p2p;
int pbf_comm_unique_id(void* id_out);
/* an id that selects the in-process loopback transport instead of RCCL: the `world` ranks are
 * threads of ONE process (each with its own ctx and stream, possibly on one GPU); same matching
 * rule and stream semantics, implemented with events and device-to-device copies.  For tests of
 * the multi-rank protocol where only one GPU is available. */
int pbf_comm_local_id(void* id_out);
int pbf_comm_create(const void* id, int32_t rank, int32_t world, int32_t device, pbf_comm** out);
int pbf_comm_destroy(pbf_comm* comm);
int32_t pbf_comm_rank(const pbf_comm* comm);
int32_t pbf_comm_world(const pbf_comm* comm);
const char* pbf_comm_last_error(const pbf_comm* comm);
/* one ncclGroupStart .. ncclGroupEnd of sends / receives, enqueued on hip_stream (hipStream_t):
 * asynchronous, ordered with the kernels of a ctx that runs on the same stream.  Messages
 * between one pair of ranks match in the order they are listed. */
int pbf_comm_batch(pbf_comm* comm, void* hip_stream, const pbf_p2p* ops, uint32_t n_ops);
/* one halo refresh in one call, on the ctx stream: pbf_slab_pack_halo into d_send_*, one grouped
 * exchange with the slab neighbours (rank < 0 = open end), pbf_slab_unpack_halo from d_recv_*.
 * Buffers: one float4 per boundary-plane particle / per ghost. */
int pbf_slab_halo_exchange(pbf_ctx* ctx, pbf_comm* comm, int32_t what, int32_t left_rank,
                           int32_t right_rank, void* d_send_left, void* d_send_right,
                           void* d_recv_left, void* d_recv_right);

/* host-pointer variants of pbf_slab_append_owned / pbf_slab_export (staged through device memory
 * by the library; synchronise) for callers that hold no device allocator, e.g. the C++ driver */
int pbf_slab_append_owned_host(pbf_ctx* ctx, const float* records, uint32_t count);
int pbf_slab_export_host(pbf_ctx* ctx, float* records, uint32_t max_records, uint32_t* count);
/* the native slab runtime: attach a communicator and the neighbour ranks (< 0 = open end) to a
 * slab ctx -- the ctx allocates its exchange buffers (max_records particles per message; 0 =
 * capacity / 4) -- then pbf_slab_step runs one whole sub-step of the slab INCLUDING the
 * exchanges: predict, split, counts to the neighbours, the sub-step's one host synchronisation,
 * migrants + ghost planes, grid, iterations x { lambda, delta-p } with a ghost refresh only when
 * fewer than two valid ghost layers are left, velocity update (+ vorticity / XSPH stages).
 * Every rank of the communicator calls it once per sub-step.  The ctx's bounds may be moved
 * between sub-steps (pbf_slab_set_bounds: re-balancing).
 * Failure is collective: the sub-step's counters of EVERY rank travel to every rank (64 bytes per
 * pair, ahead of the read-back), each rank runs all ranks' checks, and when any rank's exchange
 * buffer or capacity is too small (PBF_ERR_CAPACITY) or a particle crossed more than ghost_layers
 * planes (PBF_ERR_INVALID), ALL ranks skip the payload exchange and return that same error, naming
 * the rank, from this same call -- no rank is left waiting in a receive.  The slab's state is lost
 * then (pbf_num_particles == 0). */
int pbf_slab_attach(pbf_ctx* ctx, pbf_comm* comm, int32_t left_rank, int32_t right_rank,
                    uint32_t max_records);
int pbf_slab_step(pbf_ctx* ctx, float dt);
/* The granular sub-step (pbf_step_dem) of this rank's slab, exchanges included; needs ghost_layers >= 2.
 * Same collective failure behaviour as pbf_slab_step.  Every pass runs over owned + ghost particles; the
 * ghost positions are refreshed from their owners whenever fewer than two exact ghost layers are left (a
 * pass uses up one).  Bit-identical with pbf_step_dem on one ctx over the whole scene. */
int pbf_slab_step_dem(pbf_ctx* ctx, float dt, const pbf_dem_params* dem);
/* Overlap of the ghost refresh with the lambda sweep it precedes (pbf_slab_step, ghost_layers >= 2): the
 * pack runs on the ctx stream, the grouped send / receive and the unpack on a second stream of the ctx,
 * the sweep of the interior cell planes (which read no ghost) meanwhile on the ctx stream, and the sweeps
 * of the two edge windows behind an event of the second stream.  Costs two more launches and two event
 * waits per refresh and hides the link time of the refresh; which is faster depends on the links
 * (bench.py --gpus N times both before the run).  Results do not depend on it.  Every rank of the
 * communicator must use the same setting only for speed, not for correctness.  Default: off. */
int pbf_slab_set_overlap(pbf_ctx* ctx, int32_t on);
/* Collective re-balancing between sub-steps (every rank of the communicator calls it with the same
 * arguments): the per-x-plane counts of the owned particles of all ranks are gathered on every rank
 * (cell planes [x_first, x_first + n_planes), n_planes <= 4096; particles outside count in the end
 * bins), every rank computes the same equal-count planes and moves each interior plane towards its
 * target by at most max_move cells (1 <= max_move <= ghost_layers - 1: the particles of a moved
 * plane travel as ordinary migrants of the next sub-step), keeping interior slabs at least
 * 2 * ghost_layers planes wide.  planes[world + 1] (planes[0] = INT32_MIN, planes[world] =
 * INT32_MAX) is updated in place, identically on every rank, and this ctx's bounds follow
 * (pbf_slab_set_bounds).  *moved = number of planes that changed.  Synchronises. */
int pbf_slab_rebalance(pbf_ctx* ctx, int32_t* planes, int32_t x_first, uint32_t n_planes, int32_t max_move,
                       int32_t* moved);
/* owned particles after the last sub-step (its ghosts sit behind them until the next split) */
uint32_t pbf_slab_owned_count(const pbf_ctx* ctx);

/* ---- cell binning without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers
 * that cannot allocate device memory -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc -- and therefore cannot create a pbf_ctx.  Same result as
 * pbf_build_grid + pbf_get_cells on a ctx with the same grid (NeighborSearch::Flooring cell of
 * every particle, wrapped linear key, counting sort with the reference's bucket order, exclusive
 * prefix sum), on caller-owned DEVICE buffers:
 *   d_xyz        in : n packed float3 positions (layout of pbf_upload_device / pbf_export_positions)
 *   d_cell_key   out: n x uint32, wrapped linear cell id by particle index
 *   d_sorted_id  out: n x uint32, particle index stored at each sorted slot
 *   d_cell_start out: (cells + 1) x uint32, exclusive prefix sum of the cell occupancy
 * The histogram, the scatter cursor, the unordered scatter result and the scan's tile totals live
 * in ONE fixed device array inside the library (12 MiB + 2 KiB on every GPU it runs on, held until
 * the process ends), so the call performs no hipMalloc / hipFree and calls no allocating routine.
 * Limits: n <= PBF_NOALLOC_MAX_PARTICLES and grid cells <= PBF_NOALLOC_MAX_CELLS, else
 * PBF_ERR_CAPACITY (message: pbf_last_error(NULL)).  Asynchronous on hip_stream (hipStream_t; NULL =
 * the null stream) of `device` (< 0 = the current device).  All calls on one GPU share that GPU's
 * array: calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
#define PBF_NOALLOC_MAX_PARTICLES (1u << 20)
#define PBF_NOALLOC_MAX_CELLS (1u << 20)
int pbf_build_cells_noalloc(const void* d_xyz, uint32_t n, const int32_t grid_dim[3],
                            const int32_t grid_origin_cell[3], float grid_spacing, void* d_cell_key,
                            void* d_sorted_id, void* d_cell_start, int device, void* hip_stream);

/* self-test (no reference counterpart): compares the kernels' 6-instruction correctly rounded
 * fp32 square root with the compiler's sqrtf over EVERY non-negative float; *mismatches must
 * be 0 (first_bad = bit pattern of the first differing input, 0xffffffff if none) */
int pbf_selftest_sqrt(pbf_ctx* ctx, uint64_t* mismatches, uint32_t* first_bad);

/* per-phase device time of the kernels launched by this ctx, measured with HIP events on
 * the ctx stream when enabled (replaces the ImGui chrono panel, Simulation.cpp:126-133).
 * phase names: pbf_phase_name(i), i < pbf_num_phases(). ms[i] accumulates, calls[i] counts. */
int pbf_profile_enable(pbf_ctx* ctx, int enable);
int pbf_profile_reset(pbf_ctx* ctx);
int pbf_profile_read(pbf_ctx* ctx, double* ms, uint64_t* calls, uint32_t max_phases);
uint32_t pbf_num_phases(void);
const char* pbf_phase_name(uint32_t i);

#ifdef __cplusplus
}
#endif
#endif /* PBF_H_ */
