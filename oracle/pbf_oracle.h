/*
 * pbf_oracle.h -- CPU oracle: TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's OpenMP CPU path (Simulation::Step and
 * the components it calls).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (libpbf_hip.so) never does.
 *
 * Pinning status: pinned against the reference's own sources compiled into oracle/_ref
 * (oracle/Makefile; tests/test_oracle.py "..._equals_compiled_reference" and the fixtures
 * tests/golden/ref_*.npz those sources produced): SPHKernel and point/plane/AABB arithmetic,
 * the collision response and re-prediction (ParticleSet::TestCollision / OnCollision), the
 * velocity update (ParticleSet::Update) and the PBD / XPBD distance-constraint solve
 * (ConstraintSolver over DistanceConstraint).  Hash / flooring / derived constants are pinned by
 * source-derived known answers.  The reference holds no golden vectors or
 * tests of its own and its Simulation.cpp / NeighborSearch.cpp cannot be built here, so the
 * density / lambda / delta-p loops and the neighbour search are pinned only by those pieces
 * plus NaiveSearch == SpatialSearch.  XSPH viscosity, vorticity confinement, the spiky-gradient
 * option, the container clamp, scorr_dt_power != 1, the boundary-particle coupling and the DEM
 * solver have no CPU implementation in the reference: PARITY UNPINNED for those.
 */
#ifndef PBF_ORACLE_H_
#define PBF_ORACLE_H_

#include <stdint.h>
#include "pbf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* the reference CPU path's constants (same values as pbf_default_params) */
int orc_default_params(pbf_params* p);
int orc_create(const pbf_params* params, orc_ctx** out);
int orc_destroy(orc_ctx* ctx);
int orc_set_threads(orc_ctx* ctx, int num_threads); /* reference: num_threads(8) */
int orc_upload(orc_ctx* ctx, const float* pos, const float* vel, const float* mass,
               float uniform_mass, uint32_t n);
int orc_set_colliders(orc_ctx* ctx, const pbf_collider* colliders, uint32_t count);
/* static boundary particles (cuda_simulation.cu K5/K7/K9 and the boundary halves of K6/K8/K10,
 * in the CPU path's arithmetic; PARITY UNPINNED); nb = 0 removes them */
int orc_set_boundary(orc_ctx* ctx, const float* pos, const float* mass, float uniform_mass, uint32_t nb);
int orc_get_boundary(orc_ctx* ctx, float* volume, float* density, float* lambda);
int orc_get_boundary_neighbors(orc_ctx* ctx, uint32_t* fluid_counts, uint32_t* fluid_to_boundary,
                               uint32_t stride_f, uint32_t* boundary_counts,
                               uint32_t* boundary_to_fluid, uint32_t stride_b);
/* PBD granular / DEM sub-step (cuda_simulation.cu:1730-2225 restated in plain fp32; PARITY
 * UNPINNED) */
int orc_step_dem(orc_ctx* ctx, float dt, const pbf_dem_params* dem);
/* DistanceConstraint list under ConstraintSolver::SolveConstraints (Constraints.cpp:61-120,
 * ConstraintSolver.cpp:12-72), sequential Gauss-Seidel in list order */
int orc_set_distance_constraints(orc_ctx* ctx, const pbf_distance_constraint* list, uint32_t count, int32_t mode);
int orc_project_constraints(orc_ctx* ctx, float dt);
int orc_get_constraint_lambdas(orc_ctx* ctx, float* lambdas);
int orc_set_solver_iterations(orc_ctx* ctx, int32_t iterations);
int orc_set_gravity(orc_ctx* ctx, float gx, float gy, float gz);

/* the sub-steps of Simulation::Step, one function per reference function */
int orc_predict_positions(orc_ctx* ctx, float dt);       /* Simulation.cpp:578-599 */
int orc_collision_detection(orc_ctx* ctx, float dt);     /* Simulation.cpp:723-751 */
int orc_spatial_search(orc_ctx* ctx);                    /* NeighborSearch.cpp:131-236 */
int orc_naive_search(orc_ctx* ctx);                      /* NeighborSearch.cpp:103-129 */
int orc_compute_rest_density(orc_ctx* ctx);              /* Simulation.cpp:273-300 */
int orc_compute_density(orc_ctx* ctx);                   /* Simulation.cpp:607-640 */
int orc_compute_lambdas(orc_ctx* ctx);                   /* Simulation.cpp:642-679 */
int orc_compute_correction(orc_ctx* ctx, float dt);      /* Simulation.cpp:681-712 */
int orc_update_predict_position(orc_ctx* ctx);           /* Simulation.cpp:714-721 */
int orc_apply_solver_results(orc_ctx* ctx, float dt);    /* Simulation.cpp:787-791 */
int orc_step(orc_ctx* ctx, float dt);                    /* Simulation.cpp:80-138 */
/* the stages of orc_apply_solver_results, and raw array access (0 pos, 1 vel, 2 pred == new,
 * 3 lambda, 4 omega), for the slab-decomposition tests */
int orc_update_velocity(orc_ctx* ctx, float dt);         /* Particle.cpp:207-215 */
int orc_vorticity_omega(orc_ctx* ctx);
int orc_vorticity_apply(orc_ctx* ctx, float dt);
int orc_xsph(orc_ctx* ctx);
int orc_set_array(orc_ctx* ctx, int which, const float* data);
int orc_get_array(orc_ctx* ctx, int which, float* data);

int orc_download(orc_ctx* ctx, float* pos, float* vel, float* pred, float* newp);
int orc_get_scalars(orc_ctx* ctx, float* density, float* C, float* lambda);
int orc_get_cells(orc_ctx* ctx, int32_t* cell_xyz, uint32_t* hash);
uint64_t orc_neighbor_total(orc_ctx* ctx);
uint32_t orc_neighbor_max(orc_ctx* ctx);
int orc_get_neighbors(orc_ctx* ctx, uint32_t* counts, uint32_t* neighbors, uint32_t stride);
int orc_get_stats(orc_ctx* ctx, pbf_stats* out);
float orc_rest_density(orc_ctx* ctx);
uint32_t orc_num_particles(orc_ctx* ctx);

/* known-answer access to the leaf arithmetic */
float orc_poly6_w(float distance, float h);                                   /* SPHKernel.h:9-30 */
void orc_poly6_w_gradient(const float diff[3], float distance, float h, float out[3]); /* :32-54 */
float orc_spiky_w(float distance, float h);                                   /* :56-75 */
void orc_spiky_w_gradient(const float diff[3], float distance, float h, float out[3]); /* :78-97 */
uint32_t orc_hash_value(int32_t x, int32_t y, int32_t z);      /* NeighborSearch.cpp:255-266 */
void orc_flooring(const float p[3], float spacing, int32_t out[3]); /* NeighborSearch.cpp:243-252 */
int orc_point_plane(const float p[3], const float n[3], float d);  /* CollisionDetection.cpp:62-67 */
int orc_point_aabb(const float p[3], const float mn[3], const float mx[3]); /* :29-34 */
/* Simulation::SetupSimParams derived constants (Simulation.cpp:315-326) */
void orc_setup_sim_params(float* effective_radius, float* particle_radius, float* volume);

#ifdef __cplusplus
}
#endif
#endif
