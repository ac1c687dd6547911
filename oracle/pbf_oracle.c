/*
 * pbf_oracle.c -- CPU oracle: TEST INFRASTRUCTURE ONLY (see pbf_oracle.h).
 *
 * Restates, function by function, the reference's OpenMP CPU path.  Every function
 * cites the /root/reference file:line it follows.  Arithmetic is kept op-for-op:
 * fp32 vectors (glm::vec3 semantics: component-wise, dot = (x*x + y*y) + z*z,
 * glm/detail/func_geometric.inl:65-72), fp64 SPH kernel evaluation with the
 * float literal M_PI = 3.1415926f (common.h:8).  Build with -ffp-contract=off.
 *
 * Differences from the reference that do not change results:
 *   - debug std::cout in hot loops (Simulation.cpp:636,672) and ImGui/GL calls removed
 *   - unordered_map<uint, HashEntry*> (NeighborSearch.h:67) restated as an open-addressing
 *     table keyed by the same 32-bit hash; buckets keep ascending particle index
 *   - m_search_cache (vector<vector<size_t>>) restated as CSR built in two passes
 *   - pow(h, 9) hoisted out of the pair loops (pure function of h)
 */
#include "pbf_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.1415926f /* common.h:8 */

typedef struct {
    uint32_t key;
    int32_t head, tail;
    int32_t used;
} bucket_t;

struct orc_ctx {
    pbf_params P;
    int threads;
    uint32_t n;
    float *pos, *pred, *newp, *vel, *force; /* n*3, Particle.h:56-61 */
    float *mass, *mass_inv, *density, *C, *lambda; /* Particle.h:63-67 */
    float *vel2, *omega; /* build-side: XSPH / vorticity scratch */
    pbf_collider colliders[PBF_MAX_COLLIDERS];
    uint32_t num_colliders;
    /* search state */
    int32_t* cell;   /* n*3 */
    uint32_t* hash;  /* n */
    bucket_t* table;
    uint32_t table_size; /* power of two */
    int32_t* next;   /* n, chain inside a bucket in insertion order */
    uint64_t* nbr_off; /* n+1 */
    uint32_t* nbr;
    uint64_t nbr_cap;
    /* boundary particles (static): positions, mass, volume (K5), and per
     * iteration density / C / lambda (K7, K9) */
    uint32_t nb;
    float *bpos, *bmass, *bvol, *bdensity, *bC, *blambda;
    int32_t* bcell;      /* nb*3 */
    uint32_t* border;    /* boundary indices sorted by (cell x, y, z, index) */
    uint64_t *bb_off, *fb_off, *bf_off; /* boundary-boundary (static), fluid->boundary, boundary->fluid */
    uint32_t *bb, *fb, *bf;
    uint64_t fb_cap, bf_cap;
    pbf_distance_constraint* cons; /* m_static_constraints, Simulation.h */
    float* cons_lambda;
    uint32_t ncons;
    int32_t cons_mode;
    int first_frame;      /* m_first_frame, Simulation.cpp:17 */
    float rest_density;   /* m_rest_density */
    int32_t iterations;   /* ConstraintSolver::m_solver_iteration */
};

/* ---------------------------------------------------------------- leaf arithmetic */

/* SPHKernel::Poly6_W, SPHKernel.h:9-30 */
float orc_poly6_w(float distance, float effective_radius) {
    if (distance >= 0 && distance <= effective_radius) {
        const double h = (double)effective_radius;
        const double d = (double)distance;
        double h2 = h * h;
        double h9 = pow(h, 9);
        double d2 = d * d;
        double q = h2 - d2;
        double q3 = q * q * q;
        double result = (315.0 / (64.0 * REF_PI * h9)) * q3;
        return (float)result;
    }
    return 0.0f;
}

/* SPHKernel::Poly6_W_Gradient, SPHKernel.h:32-54 */
void orc_poly6_w_gradient(const float diff[3], float distance, float effective_radius,
                          float out[3]) {
    if (distance >= 0 && distance <= effective_radius) {
        const double h = (double)effective_radius;
        const double d = (double)distance;
        double h2 = h * h;
        double h9 = pow(h, 9);
        double d2 = d * d;
        double q = h2 - d2;
        double q2 = q * q;
        double scalar = (-945.0 / (32.0 * REF_PI * h9));
        float s = (float)scalar * (float)q2; /* both rounded to float first, :46 */
        out[0] = s * diff[0];
        out[1] = s * diff[1];
        out[2] = s * diff[2];
        return;
    }
    out[0] = out[1] = out[2] = 0.0f;
}

/* SPHKernel::Spiky_W, SPHKernel.h:56-75 */
float orc_spiky_w(float distance, float effective_radius) {
    if (distance >= 0 && distance <= effective_radius) {
        const double h = (double)effective_radius;
        const double d = (double)distance;
        double h6 = pow(h, 6);
        double q = h - d;
        double q3 = q * q * q;
        return (float)((15.0 / (REF_PI * h6)) * q3);
    }
    return 0.0f;
}

/* SPHKernel::Spiky_W_Gradient, SPHKernel.h:78-97 (divides by distance: inf/NaN at 0) */
void orc_spiky_w_gradient(const float diff[3], float distance, float effective_radius,
                          float out[3]) {
    if (distance >= 0 && distance <= effective_radius) {
        const double h = (double)effective_radius;
        const double d = (double)distance;
        double h6 = pow(h, 6);
        double q = h - d;
        double q2 = q * q;
        double scalar = (-45.0 / (REF_PI * h6)) * (q2 / distance);
        float s = (float)scalar;
        out[0] = s * diff[0];
        out[1] = s * diff[1];
        out[2] = s * diff[2];
        return;
    }
    out[0] = out[1] = out[2] = 0.0f;
}

/* hoisted forms used in the pair loops: identical values, pow() evaluated once */
typedef struct {
    float h;
    double h2;
    double w_coef;    /* 315/(64 pi h^9) */
    float g_coef;     /* (float)(-945/(32 pi h^9)) */
    double spiky_coef; /* -45/(pi h^6) */
    double hd;
    int grad_kernel;
} kern_t;

static kern_t make_kern(float hf, int grad_kernel) {
    kern_t k;
    const double h = (double)hf;
    double h9 = pow(h, 9);
    double h6 = pow(h, 6);
    k.h = hf;
    k.hd = h;
    k.h2 = h * h;
    k.w_coef = (315.0 / (64.0 * REF_PI * h9));
    k.g_coef = (float)(-945.0 / (32.0 * REF_PI * h9));
    k.spiky_coef = (-45.0 / (REF_PI * h6));
    k.grad_kernel = grad_kernel;
    return k;
}

static inline float k_poly6(const kern_t* k, float distance) {
    if (distance >= 0 && distance <= k->h) {
        const double d = (double)distance;
        double d2 = d * d;
        double q = k->h2 - d2;
        double q3 = q * q * q;
        return (float)(k->w_coef * q3);
    }
    return 0.0f;
}

static inline void k_grad(const kern_t* k, const float diff[3], float distance, float out[3]) {
    if (distance >= 0 && distance <= k->h) {
        const double d = (double)distance;
        float s;
        if (k->grad_kernel == PBF_GRAD_SPIKY) {
            double q = k->hd - d;
            double q2 = q * q;
            double scalar = k->spiky_coef * (q2 / distance);
            s = (float)scalar;
        } else {
            double d2 = d * d;
            double q = k->h2 - d2;
            double q2 = q * q;
            s = k->g_coef * (float)q2;
        }
        out[0] = s * diff[0];
        out[1] = s * diff[1];
        out[2] = s * diff[2];
        return;
    }
    out[0] = out[1] = out[2] = 0.0f;
}

/* glm::dot on vec3, func_geometric.inl:65-72 */
static inline float dot3(const float a[3], const float b[3]) {
    float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2];
    return tx + ty + tz;
}

/* glm::distance(p0,p1) = sqrt(dot(p1-p0, p1-p0)), func_geometric.inl:95-115 */
static inline float dist3(const float p0[3], const float p1[3]) {
    float v[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    return sqrtf(dot3(v, v));
}

/* glm::distance2(p0,p1) = dot(p1-p0, p1-p0), gtx/norm.inl:92-99 */
static inline float dist2_3(const float p0[3], const float p1[3]) {
    float v[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    return dot3(v, v);
}

/* NeighborSearch::GetHashValue, NeighborSearch.cpp:255-266 (int32 wrap-around) */
uint32_t orc_hash_value(int32_t x, int32_t y, int32_t z) {
    const uint32_t p1 = 73856093u * (uint32_t)x;
    const uint32_t p2 = 19349663u * (uint32_t)y;
    const uint32_t p3 = 83492791u * (uint32_t)z;
    return p1 + p2 + p3;
}

/* NeighborSearch::Flooring, NeighborSearch.cpp:243-252 */
void orc_flooring(const float p[3], float spacing, int32_t out[3]) {
    out[0] = (int32_t)floorf(p[0] / spacing);
    out[1] = (int32_t)floorf(p[1] / spacing);
    out[2] = (int32_t)floorf(p[2] / spacing);
}

/* CollisionDetection::PointPlaneIntersection, CollisionDetection.cpp:62-67 */
int orc_point_plane(const float p[3], const float n[3], float d) {
    float dot_val = dot3(n, p);
    return dot_val <= d;
}

/* CollisionDetection::PointAABBIntersection, CollisionDetection.cpp:29-34 */
int orc_point_aabb(const float p[3], const float mn[3], const float mx[3]) {
    return (p[0] < mx[0] && p[0] > mn[0] && p[1] < mx[1] && p[1] > mn[1] && p[2] < mx[2] &&
            p[2] > mn[2]);
}

/* Simulation::SetupSimParams, Simulation.cpp:315-326 */
void orc_setup_sim_params(float* effective_radius, float* particle_radius, float* volume) {
    /* volatile: evaluate powf at run time with libm, as the reference binary does, instead of
     * letting the compiler fold it (the folded value differs from glibc's by one ulp in r) */
    volatile float particle_mass = 0.015f;
    volatile float n_kernel_particles = 20.f;
    volatile float rest_density = 1000.f;
    float v = n_kernel_particles * particle_mass / rest_density;
    float h = powf(((3.0f * v) / (4.0f * REF_PI)), 1.0f / 3.0f);
    float r = powf((REF_PI / (6.0f * n_kernel_particles)), 1.0f / 3.0f) * h;
    if (effective_radius) *effective_radius = h;
    if (particle_radius) *particle_radius = r;
    if (volume) *volume = v;
}

/* ---------------------------------------------------------------- ctx plumbing */

static void* xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

int orc_default_params(pbf_params* p) {
    if (!p) return PBF_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    p->abi_version = PBF_ABI_VERSION;
    p->capacity = 0;
    p->max_neighbors = 64;
    p->grid_dim[0] = p->grid_dim[1] = p->grid_dim[2] = 64; /* Simulation.cpp:46 */
    p->grid_spacing = 1.f;                                 /* NeighborSearch.cpp:17 */
    p->effective_radius = 1.f;                             /* Simulation.cpp:88 */
    p->gravity[0] = 0.f;
    p->gravity[1] = -9.8f;                                 /* Simulation.cpp:16 */
    p->gravity[2] = 0.f;
    p->epsilon = 1.0e-6f;                                  /* Simulation.cpp:644 */
    p->restitution = 0.99f;                                /* Particle.cpp:333 */
    p->scorr_k = 0.1;                                      /* Simulation.cpp:698 */
    p->scorr_dq = 0.3f;                                    /* Simulation.cpp:700 */
    p->scorr_dt_power = 1;                                 /* Simulation.cpp:702 */
    p->rest_density_mode = PBF_RHO0_FIRST_FRAME_MEAN;      /* Simulation.cpp:103-107 */
    p->rest_density = 0.8f;                                /* Simulation.cpp:18 */
    p->solver_iterations = 1;                              /* ConstraintSolver.cpp:4 */
    p->gradient_kernel = PBF_GRAD_POLY6;
    return PBF_OK;
}

int orc_create(const pbf_params* params, orc_ctx** out) {
    if (!params || !out || params->abi_version != PBF_ABI_VERSION) return PBF_ERR_INVALID;
    if (params->grid_spacing <= 0.f || params->effective_radius <= 0.f) return PBF_ERR_INVALID;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    if (!c) return PBF_ERR_INVALID;
    c->P = *params;
    c->threads = 8; /* num_threads(8), Simulation.cpp:587 */
    c->first_frame = 1;
    c->rest_density = params->rest_density;
    c->iterations = params->solver_iterations;
    *out = c;
    return PBF_OK;
}

static void free_particles(orc_ctx* c) {
    free(c->pos); free(c->pred); free(c->newp); free(c->vel); free(c->force);
    free(c->mass); free(c->mass_inv); free(c->density); free(c->C); free(c->lambda);
    free(c->vel2); free(c->omega);
    free(c->cell); free(c->hash); free(c->table); free(c->next); free(c->nbr_off); free(c->nbr);
    c->pos = c->pred = c->newp = c->vel = c->force = NULL;
    c->mass = c->mass_inv = c->density = c->C = c->lambda = NULL;
    c->vel2 = c->omega = NULL;
    c->cell = NULL; c->hash = NULL; c->table = NULL; c->next = NULL; c->nbr_off = NULL;
    c->nbr = NULL; c->nbr_cap = 0; c->table_size = 0;
}

static void free_boundary(orc_ctx* c);

int orc_destroy(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    free_particles(c);
    free_boundary(c);
    free(c->cons); free(c->cons_lambda);
    free(c);
    return PBF_OK;
}

int orc_set_threads(orc_ctx* c, int t) {
    if (!c || t < 1) return PBF_ERR_INVALID;
    c->threads = t;
    return PBF_OK;
}

/* ParticleSet::ParticleSet, Particle.cpp:172-188: predict = new = pos as the scene
 * generators set them (Simulation.cpp:567-569), velocity as given, force 0,
 * massInv = 1.f / mass */
int orc_upload(orc_ctx* c, const float* pos, const float* vel, const float* mass,
               float uniform_mass, uint32_t n) {
    if (!c || (n && !pos)) return PBF_ERR_INVALID;
    free_particles(c);
    c->n = n;
    size_t n3 = (size_t)n * 3;
    c->pos = (float*)xcalloc(n3, 4); c->pred = (float*)xcalloc(n3, 4);
    c->newp = (float*)xcalloc(n3, 4); c->vel = (float*)xcalloc(n3, 4);
    c->force = (float*)xcalloc(n3, 4); c->vel2 = (float*)xcalloc(n3, 4);
    c->omega = (float*)xcalloc(n3, 4);
    c->mass = (float*)xcalloc(n, 4); c->mass_inv = (float*)xcalloc(n, 4);
    c->density = (float*)xcalloc(n, 4); c->C = (float*)xcalloc(n, 4);
    c->lambda = (float*)xcalloc(n, 4);
    c->cell = (int32_t*)xcalloc(n3, 4); c->hash = (uint32_t*)xcalloc(n, 4);
    c->next = (int32_t*)xcalloc(n, 4);
    c->nbr_off = (uint64_t*)xcalloc((size_t)n + 1, 8);
    uint32_t ts = 16;
    while (ts < 4u * n) ts <<= 1;
    c->table_size = ts;
    c->table = (bucket_t*)xcalloc(ts, sizeof(bucket_t));
    if (n) {
        memcpy(c->pos, pos, n3 * 4);
        memcpy(c->pred, pos, n3 * 4);
        memcpy(c->newp, pos, n3 * 4);
        if (vel) memcpy(c->vel, vel, n3 * 4);
    }
    for (uint32_t i = 0; i < n; ++i) {
        float m = mass ? mass[i] : uniform_mass;
        c->mass[i] = m;
        c->mass_inv[i] = 1.f / m;
    }
    c->first_frame = 1;
    c->rest_density = c->P.rest_density;
    return PBF_OK;
}

int orc_set_colliders(orc_ctx* c, const pbf_collider* cols, uint32_t count) {
    if (!c || count > PBF_MAX_COLLIDERS || (count && !cols)) return PBF_ERR_INVALID;
    for (uint32_t i = 0; i < count; ++i)
        if (cols[i].type != PBF_COLLIDER_PLANE && cols[i].type != PBF_COLLIDER_AABB &&
            cols[i].type != PBF_COLLIDER_HARD_BOX)
            return PBF_ERR_INVALID;
    if (count) memcpy(c->colliders, cols, count * sizeof(pbf_collider));
    c->num_colliders = count;
    return PBF_OK;
}

int orc_set_solver_iterations(orc_ctx* c, int32_t it) {
    if (!c || it < 0) return PBF_ERR_INVALID;
    c->iterations = it;
    return PBF_OK;
}

int orc_set_gravity(orc_ctx* c, float gx, float gy, float gz) {
    if (!c) return PBF_ERR_INVALID;
    c->P.gravity[0] = gx; c->P.gravity[1] = gy; c->P.gravity[2] = gz;
    return PBF_OK;
}

/* ---------------------------------------------------------------- the step */

/* Simulation::PredictPositions, Simulation.cpp:578-599 */
int orc_predict_positions(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const float* g = c->P.gravity;
#pragma omp parallel for schedule(static) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        float* f = c->force + 3 * i;
        float* v = c->vel + 3 * i;
        const float* p = c->pos + 3 * i;
        float m = c->mass[i];
        /* m_force = m_mass * vec3(0, g, 0) */
        f[0] = m * g[0]; f[1] = m * g[1]; f[2] = m * g[2];
        /* v = v + dt * massInv * force : (dt*massInv) is a scalar product first */
        float s = dt * c->mass_inv[i];
        v[0] = v[0] + s * f[0]; v[1] = v[1] + s * f[1]; v[2] = v[2] + s * f[2];
        float* pr = c->pred + 3 * i;
        pr[0] = p[0] + dt * v[0]; pr[1] = p[1] + dt * v[1]; pr[2] = p[2] + dt * v[2];
        float* np = c->newp + 3 * i;
        np[0] = pr[0]; np[1] = pr[1]; np[2] = pr[2];
    }
    return PBF_OK;
}

/* ParticleSet::OnCollision, Particle.cpp:248-338 (PLANE and AABB branches; the SPHERE
 * branch reads an uninitialised v_r and OBB is a TODO -- neither is reproduced) */
static void on_collision(orc_ctx* c, int i, const pbf_collider* col, float dt) {
    float* v = c->vel + 3 * i;
    float* pr = c->pred + 3 * i;
    float normal[3];
    if (col->type == PBF_COLLIDER_AABB) {
        /* Particle.cpp:272-310: picks the face with the LARGEST distance (sic) */
        float diff2max[3] = {col->b[0] - pr[0], col->b[1] - pr[1], col->b[2] - pr[2]};
        float diff2min[3] = {pr[0] - col->a[0], pr[1] - col->a[1], pr[2] - col->a[2]};
        static const float preset[6][3] = {{1, 0, 0},  {0, 1, 0},  {0, 0, 1},
                                           {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
        float min_dist = diff2max[0];
        int pick = 0;
        for (int k = 1; k < 3; ++k)
            if (min_dist < diff2max[k]) min_dist = diff2max[k], pick = k;
        for (int k = 0; k < 3; ++k)
            if (min_dist < diff2min[k]) min_dist = diff2min[k], pick = 3 + k;
        normal[0] = preset[pick][0]; normal[1] = preset[pick][1]; normal[2] = preset[pick][2];
    } else {
        normal[0] = col->a[0]; normal[1] = col->a[1]; normal[2] = col->a[2];
    }
    /* v_r = v - 2.f * dot(v, normal) * normal, Particle.cpp:306,325 */
    float s = 2.f * dot3(v, normal);
    v[0] = v[0] - s * normal[0]; v[1] = v[1] - s * normal[1]; v[2] = v[2] - s * normal[2];
    /* restitution, Particle.cpp:333 */
    float e = c->P.restitution;
    v[0] = e * v[0]; v[1] = e * v[1]; v[2] = e * v[2];
    /* re-prediction from the OLD position, Particle.cpp:336-337 */
    const float* p = c->pos + 3 * i;
    pr[0] = p[0] + dt * v[0]; pr[1] = p[1] + dt * v[1]; pr[2] = p[2] + dt * v[2];
    float* np = c->newp + 3 * i;
    np[0] = pr[0]; np[1] = pr[1]; np[2] = pr[2];
}

/* The box the reference's CUDA path bakes into integrate_pbd_d (cuda_simulation.cu:473-511), as a collider:
 * clamp the predicted coordinate, point the velocity component back inside, damp the whole vector; the
 * predicted position is not recomputed.  PARITY UNPINNED (CUDA-only). */
static void hard_box(orc_ctx* c, int i, const pbf_collider* col) {
    float* v = c->vel + 3 * i;
    float* pr = c->pred + 3 * i;
    const float damp = col->d;
    if (pr[0] >= col->b[0]) { pr[0] = col->b[0]; v[0] = -fabsf(v[0]); v[0] *= damp; v[1] *= damp; v[2] *= damp; }
    if (pr[0] <= col->a[0]) { pr[0] = col->a[0]; v[0] = fabsf(v[0]); v[0] *= damp; v[1] *= damp; v[2] *= damp; }
    if (pr[2] >= col->b[2]) { pr[2] = col->b[2]; v[2] = -fabsf(v[2]); v[0] *= damp; v[1] *= damp; v[2] *= damp; }
    if (pr[2] <= col->a[2]) { pr[2] = col->a[2]; v[2] = fabsf(v[2]); v[0] *= damp; v[1] *= damp; v[2] *= damp; }
    if (pr[1] <= col->a[1]) { pr[1] = col->a[1]; v[1] = fabsf(v[1]); v[0] *= damp; v[1] *= damp; v[2] *= damp; }
    float* np = c->newp + 3 * i;
    np[0] = pr[0]; np[1] = pr[1]; np[2] = pr[2];
}

/* Simulation::CollisionDetection, Simulation.cpp:723-751 (serial in the reference;
 * particles are independent, so the loop order over i does not matter) */
int orc_collision_detection(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
#pragma omp parallel for schedule(static) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        for (uint32_t j = 0; j < c->num_colliders; ++j) {
            const pbf_collider* col = &c->colliders[j];
            if (col->type == PBF_COLLIDER_HARD_BOX) {
                hard_box(c, i, col);
                continue;
            }
            const float* pr = c->pred + 3 * i;
            int hit = (col->type == PBF_COLLIDER_PLANE) ? orc_point_plane(pr, col->a, col->d)
                                                        : orc_point_aabb(pr, col->a, col->b);
            if (hit) on_collision(c, i, col, dt);
        }
    }
    return PBF_OK;
}

static bucket_t* table_find(orc_ctx* c, uint32_t key, int insert) {
    uint32_t mask = c->table_size - 1;
    uint32_t s = (key * 2654435761u) & mask;
    for (;;) {
        bucket_t* b = &c->table[s];
        if (!b->used) {
            if (!insert) return NULL;
            b->used = 1; b->key = key; b->head = b->tail = -1;
            return b;
        }
        if (b->key == key) return b;
        s = (s + 1) & mask;
    }
}

static int ensure_nbr(orc_ctx* c, uint64_t total) {
    if (total > c->nbr_cap) {
        free(c->nbr);
        c->nbr = (uint32_t*)malloc((size_t)(total ? total : 1) * 4);
        if (!c->nbr) { c->nbr_cap = 0; return PBF_ERR_INVALID; }
        c->nbr_cap = total;
    }
    return PBF_OK;
}

static int boundary_search(orc_ctx* c);

/* NeighborSearch::SpatialSearch, NeighborSearch.cpp:131-236 */
int orc_spatial_search(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const float spacing = c->P.grid_spacing;
    const float h = c->P.effective_radius;
    const float square_h = h * h; /* :139 */
    memset(c->table, 0, (size_t)c->table_size * sizeof(bucket_t));
    /* :154-185 bucket fill, serial, ascending i */
    for (int i = 0; i < n; ++i) {
        orc_flooring(c->pred + 3 * i, spacing, c->cell + 3 * i);
        uint32_t hv = orc_hash_value(c->cell[3 * i], c->cell[3 * i + 1], c->cell[3 * i + 2]);
        c->hash[i] = hv;
        bucket_t* b = table_find(c, hv, 1);
        c->next[i] = -1;
        if (b->head < 0) b->head = i; else c->next[b->tail] = i;
        b->tail = i;
    }
    /* :187-234 cache fill; pass 0 counts, pass 1 writes (same visiting order) */
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            uint64_t acc = 0;
            for (int i = 0; i < n; ++i) { uint64_t k = c->nbr_off[i + 1]; c->nbr_off[i] = acc; acc += k; }
            c->nbr_off[n] = acc;
            int rc = ensure_nbr(c, acc);
            if (rc) return rc;
        }
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
        for (int i = 0; i < n; ++i) {
            const float* pi = c->pred + 3 * i;
            const int32_t* g = c->cell + 3 * i;
            uint64_t w = (pass == 1) ? c->nbr_off[i] : 0;
            uint64_t cnt = 0;
            for (int32_t x = -1; x < 2; ++x)
                for (int32_t y = -1; y < 2; ++y)
                    for (int32_t z = -1; z < 2; ++z) {
                        uint32_t hv = orc_hash_value(g[0] + x, g[1] + y, g[2] + z);
                        bucket_t* b = table_find(c, hv, 0);
                        if (!b) continue;
                        for (int32_t pj = b->head; pj >= 0; pj = c->next[pj]) {
                            if (pj == i) continue;
                            float distance2 = dist2_3(pi, c->pred + 3 * pj);
                            if (distance2 <= square_h) {
                                if (pass == 1) c->nbr[w++] = (uint32_t)pj;
                                ++cnt;
                            }
                        }
                    }
            if (pass == 0) c->nbr_off[i + 1] = cnt;
        }
    }
    if (c->nb) return boundary_search(c);
    return PBF_OK;
}

/* NeighborSearch::NaiveSearch, NeighborSearch.cpp:103-129 (uses m_new_positions).
 * The reference appends j to cache[i] and i to cache[j] in one serial double loop;
 * the resulting per-particle order is ascending index, which is what is built here. */
int orc_naive_search(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const float h = c->P.effective_radius;
    const float square_h = h * h;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            uint64_t acc = 0;
            for (int i = 0; i < n; ++i) { uint64_t k = c->nbr_off[i + 1]; c->nbr_off[i] = acc; acc += k; }
            c->nbr_off[n] = acc;
            int rc = ensure_nbr(c, acc);
            if (rc) return rc;
        }
#pragma omp parallel for schedule(dynamic, 16) num_threads(c->threads)
        for (int i = 0; i < n; ++i) {
            uint64_t w = (pass == 1) ? c->nbr_off[i] : 0, cnt = 0;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                /* distance2(new[min], new[max]) as the i<j loop evaluates it */
                int a = i < j ? i : j, b = i < j ? j : i;
                float distance2 = dist2_3(c->newp + 3 * a, c->newp + 3 * b);
                if (distance2 <= square_h) {
                    if (pass == 1) c->nbr[w++] = (uint32_t)j;
                    ++cnt;
                }
            }
            if (pass == 0) c->nbr_off[i + 1] = cnt;
        }
    }
    return PBF_OK;
}

/* ---------------------------------------------------------------- boundary particles
 * Build-side (PARITY UNPINNED): the boundary-particle coupling of the reference's CUDA kernels
 * (cuda_simulation.cu: K5 compute_boundary_volume_d :331-374, pbf_density_boundary :665-701,
 * K7 compute_boundary_density_d :1128-1212, pbf_lambda_boundary :846-884, K9
 * compute_boundary_lambdas_d :1296-1384, pbf_correction_boundary :990-1046) restated in the CPU
 * path's conventions: fp32 vectors, SPH kernels evaluated in double, frozen per-sub-step
 * neighbour lists (dist2 <= h*h on predicted positions), 27-cell sweep x outermost / z
 * innermost, ascending boundary index inside a cell.  Boundary particles never move. */

static int bcell_cmp(const int32_t* a, const int32_t* b) {
    for (int k = 0; k < 3; ++k)
        if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
    return 0;
}

static orc_ctx* g_sort_ctx; /* qsort has no context argument; set_boundary is single-threaded */
static int border_cmp(const void* pa, const void* pb) {
    uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
    int c = bcell_cmp(g_sort_ctx->bcell + 3 * (size_t)a, g_sort_ctx->bcell + 3 * (size_t)b);
    if (c) return c;
    return a < b ? -1 : (a > b ? 1 : 0);
}

/* [lo, hi) range of `border` whose cell equals `cell` */
static void bcell_range(const orc_ctx* c, const int32_t cell[3], uint32_t* lo, uint32_t* hi) {
    uint32_t a = 0, b = c->nb;
    while (a < b) {
        uint32_t m = a + (b - a) / 2;
        if (bcell_cmp(c->bcell + 3 * (size_t)c->border[m], cell) < 0) a = m + 1; else b = m;
    }
    *lo = a;
    b = c->nb;
    while (a < b) {
        uint32_t m = a + (b - a) / 2;
        if (bcell_cmp(c->bcell + 3 * (size_t)c->border[m], cell) <= 0) a = m + 1; else b = m;
    }
    *hi = a;
}

static void free_boundary(orc_ctx* c) {
    free(c->bpos); free(c->bmass); free(c->bvol); free(c->bdensity); free(c->bC); free(c->blambda);
    free(c->bcell); free(c->border); free(c->bb_off); free(c->fb_off); free(c->bf_off);
    free(c->bb); free(c->fb); free(c->bf);
    c->bpos = c->bmass = c->bvol = c->bdensity = c->bC = c->blambda = NULL;
    c->bcell = NULL; c->border = NULL; c->bb_off = c->fb_off = c->bf_off = NULL;
    c->bb = c->fb = c->bf = NULL; c->fb_cap = c->bf_cap = 0; c->nb = 0;
}

/* neighbours of point p (cell g) among the boundary particles, visiting order as above;
 * skip = boundary index to leave out (or UINT32_MAX); out may be NULL (count only) */
static uint64_t boundary_sweep(const orc_ctx* c, const float p[3], const int32_t g[3], uint32_t skip,
                               float square_h, uint32_t* out) {
    uint64_t cnt = 0;
    for (int32_t x = -1; x < 2; ++x)
        for (int32_t y = -1; y < 2; ++y)
            for (int32_t z = -1; z < 2; ++z) {
                int32_t cell[3] = {g[0] + x, g[1] + y, g[2] + z};
                uint32_t lo, hi;
                bcell_range(c, cell, &lo, &hi);
                for (uint32_t t = lo; t < hi; ++t) {
                    uint32_t b = c->border[t];
                    if (b == skip) continue;
                    if (dist2_3(p, c->bpos + 3 * (size_t)b) <= square_h) {
                        if (out) out[cnt] = b;
                        ++cnt;
                    }
                }
            }
    return cnt;
}

int orc_set_boundary(orc_ctx* c, const float* pos, const float* mass, float uniform_mass, uint32_t nb) {
    if (!c || (nb && !pos)) return PBF_ERR_INVALID;
    free_boundary(c);
    if (!nb) return PBF_OK;
    c->nb = nb;
    c->bpos = (float*)xcalloc((size_t)nb * 3, 4); c->bmass = (float*)xcalloc(nb, 4);
    c->bvol = (float*)xcalloc(nb, 4); c->bdensity = (float*)xcalloc(nb, 4);
    c->bC = (float*)xcalloc(nb, 4); c->blambda = (float*)xcalloc(nb, 4);
    c->bcell = (int32_t*)xcalloc((size_t)nb * 3, 4); c->border = (uint32_t*)xcalloc(nb, 4);
    c->bb_off = (uint64_t*)xcalloc((size_t)nb + 1, 8);
    c->bf_off = (uint64_t*)xcalloc((size_t)nb + 1, 8);
    memcpy(c->bpos, pos, (size_t)nb * 12);
    for (uint32_t b = 0; b < nb; ++b) {
        c->bmass[b] = mass ? mass[b] : uniform_mass;
        orc_flooring(c->bpos + 3 * (size_t)b, c->P.grid_spacing, c->bcell + 3 * (size_t)b);
        c->border[b] = b;
    }
    g_sort_ctx = c;
    qsort(c->border, nb, 4, border_cmp);
    const float h = c->P.effective_radius, square_h = h * h;
    /* static boundary-boundary lists */
    for (uint32_t b = 0; b < nb; ++b)
        c->bb_off[b + 1] = c->bb_off[b] + boundary_sweep(c, c->bpos + 3 * (size_t)b, c->bcell + 3 * (size_t)b, b, square_h, NULL);
    c->bb = (uint32_t*)xcalloc(c->bb_off[nb], 4);
    for (uint32_t b = 0; b < nb; ++b)
        boundary_sweep(c, c->bpos + 3 * (size_t)b, c->bcell + 3 * (size_t)b, b, square_h, c->bb + c->bb_off[b]);
    /* K5 compute_boundary_volume_d: V_b = m_b / (m_b W(0) + sum_b' m_b' W(|p_b - p_b'|)) */
    const kern_t k = make_kern(h, c->P.gradient_kernel);
    for (uint32_t b = 0; b < nb; ++b) {
        float rho = c->bmass[b] * k_poly6(&k, 0);
        for (uint64_t a = c->bb_off[b]; a < c->bb_off[b + 1]; ++a) {
            uint32_t o = c->bb[a];
            float dist = dist3(c->bpos + 3 * (size_t)b, c->bpos + 3 * (size_t)o);
            rho += c->bmass[o] * k_poly6(&k, dist);
        }
        c->bvol[b] = c->bmass[b] / rho;
    }
    return PBF_OK;
}

/* per sub-step, after the fluid lists: fluid->boundary and boundary->fluid lists */
static int boundary_search(orc_ctx* c) {
    const int n = (int)c->n;
    const uint32_t nb = c->nb;
    const float h = c->P.effective_radius, square_h = h * h;
    free(c->fb_off);
    c->fb_off = (uint64_t*)xcalloc((size_t)n + 1, 8);
    for (int i = 0; i < n; ++i)
        c->fb_off[i + 1] = c->fb_off[i] + boundary_sweep(c, c->pred + 3 * i, c->cell + 3 * i, UINT32_MAX, square_h, NULL);
    if (c->fb_off[n] > c->fb_cap) {
        free(c->fb);
        c->fb = (uint32_t*)xcalloc(c->fb_off[n], 4);
        c->fb_cap = c->fb_off[n];
    }
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i)
        boundary_sweep(c, c->pred + 3 * i, c->cell + 3 * i, UINT32_MAX, square_h, c->fb + c->fb_off[i]);
    /* boundary -> fluid through the fluid buckets (same sweep as SpatialSearch) */
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            uint64_t acc = 0;
            for (uint32_t b = 0; b < nb; ++b) { uint64_t k = c->bf_off[b + 1]; c->bf_off[b] = acc; acc += k; }
            c->bf_off[nb] = acc;
            if (acc > c->bf_cap) {
                free(c->bf);
                c->bf = (uint32_t*)xcalloc(acc, 4);
                c->bf_cap = acc;
            }
        }
        for (uint32_t b = 0; b < nb; ++b) {
            const float* pb = c->bpos + 3 * (size_t)b;
            const int32_t* g = c->bcell + 3 * (size_t)b;
            uint64_t w = (pass == 1) ? c->bf_off[b] : 0, cnt = 0;
            for (int32_t x = -1; x < 2; ++x)
                for (int32_t y = -1; y < 2; ++y)
                    for (int32_t z = -1; z < 2; ++z) {
                        bucket_t* bk = table_find(c, orc_hash_value(g[0] + x, g[1] + y, g[2] + z), 0);
                        if (!bk) continue;
                        for (int32_t pj = bk->head; pj >= 0; pj = c->next[pj]) {
                            /* a hash bucket can hold aliased far cells: the exact cell test keeps
                             * the list a function of geometry only */
                            const int32_t* cj = c->cell + 3 * pj;
                            if (cj[0] != g[0] + x || cj[1] != g[1] + y || cj[2] != g[2] + z) continue;
                            if (dist2_3(pb, c->pred + 3 * pj) <= square_h) {
                                if (pass == 1) c->bf[w++] = (uint32_t)pj;
                                ++cnt;
                            }
                        }
                    }
            if (pass == 0) c->bf_off[b + 1] = cnt;
        }
    }
    return PBF_OK;
}

/* K7 compute_boundary_density_d + K9 compute_boundary_lambdas_d on the current predicted
 * fluid positions */
static void boundary_density_lambda(orc_ctx* c) {
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    const float rho0 = c->rest_density, inv_rho0 = 1.f / rho0, epsilon = c->P.epsilon;
    const int nb = (int)c->nb;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int b = 0; b < nb; ++b) {
        const float* pb = c->bpos + 3 * (size_t)b;
        float rho = (rho0 * c->bvol[b]) * k_poly6(&k, 0);
        float grad_sum = 0.f;
        for (uint64_t a = c->bb_off[b]; a < c->bb_off[b + 1]; ++a) {
            uint32_t o = c->bb[a];
            const float* po = c->bpos + 3 * (size_t)o;
            float diff[3] = {pb[0] - po[0], pb[1] - po[1], pb[2] - po[2]};
            float dist = dist3(pb, po);
            rho += (rho0 * c->bvol[o]) * k_poly6(&k, dist);
            float g[3];
            k_grad(&k, diff, dist, g);
            float s = inv_rho0 * ((rho0 * c->bvol[o]) / c->bmass[o]);
            float gj[3] = {s * g[0], s * g[1], s * g[2]};
            grad_sum += dot3(gj, gj);
        }
        for (uint64_t a = c->bf_off[b]; a < c->bf_off[b + 1]; ++a) {
            uint32_t j = c->bf[a];
            const float* pj = c->pred + 3 * (size_t)j;
            float diff[3] = {pb[0] - pj[0], pb[1] - pj[1], pb[2] - pj[2]};
            float dist = dist3(pb, pj);
            rho += c->mass[j] * k_poly6(&k, dist);
            float g[3];
            k_grad(&k, diff, dist, g);
            float gj[3] = {inv_rho0 * g[0], inv_rho0 * g[1], inv_rho0 * g[2]};
            grad_sum += dot3(gj, gj);
        }
        c->bdensity[b] = rho;
        c->bC[b] = rho / rho0 - 1.f;
        float lam = -c->bC[b];
        lam /= grad_sum + epsilon;
        c->blambda[b] = lam;
    }
}

int orc_get_boundary(orc_ctx* c, float* volume, float* density, float* lambda) {
    if (!c) return PBF_ERR_INVALID;
    if (volume) memcpy(volume, c->bvol, (size_t)c->nb * 4);
    if (density) memcpy(density, c->bdensity, (size_t)c->nb * 4);
    if (lambda) memcpy(lambda, c->blambda, (size_t)c->nb * 4);
    return PBF_OK;
}

int orc_get_boundary_neighbors(orc_ctx* c, uint32_t* fluid_counts, uint32_t* fluid_to_boundary,
                               uint32_t stride_f, uint32_t* boundary_counts,
                               uint32_t* boundary_to_fluid, uint32_t stride_b) {
    if (!c) return PBF_ERR_INVALID;
    if (!c->nb || !c->fb_off) return PBF_ERR_INVALID;
    for (uint32_t i = 0; i < c->n; ++i) {
        uint64_t k = c->fb_off[i + 1] - c->fb_off[i];
        if (fluid_counts) fluid_counts[i] = (uint32_t)k;
        if (fluid_to_boundary)
            for (uint64_t a = 0; a < k && a < stride_f; ++a)
                fluid_to_boundary[(size_t)i * stride_f + a] = c->fb[c->fb_off[i] + a];
    }
    for (uint32_t b = 0; b < c->nb; ++b) {
        uint64_t k = c->bf_off[b + 1] - c->bf_off[b];
        if (boundary_counts) boundary_counts[b] = (uint32_t)k;
        if (boundary_to_fluid)
            for (uint64_t a = 0; a < k && a < stride_b; ++a)
                boundary_to_fluid[(size_t)b * stride_b + a] = c->bf[c->bf_off[b] + a];
    }
    return PBF_OK;
}

/* Simulation::ComputeRestDensity, Simulation.cpp:273-300: CURRENT positions, neighbour
 * lists from the predicted ones, serial fp32 mean in index order */
int orc_compute_rest_density(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    c->rest_density = 0;
    for (int i = 0; i < n; ++i) {
        c->density[i] = c->mass[i] * k_poly6(&k, 0);
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            float distance = dist3(c->pos + 3 * i, c->pos + 3 * j);
            c->density[i] += c->mass[j] * k_poly6(&k, distance);
        }
        c->rest_density += c->density[i];
    }
    c->rest_density /= (float)n;
    c->first_frame = 0;
    return PBF_OK;
}

/* Simulation::ComputeDensity, Simulation.cpp:607-640 */
int orc_compute_density(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    const float rho0 = c->rest_density;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        float dens = c->mass[i] * k_poly6(&k, 0);
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            float distance = dist3(c->pred + 3 * i, c->pred + 3 * j);
            float res = k_poly6(&k, distance);
            dens += c->mass[j] * res;
        }
        if (c->nb) /* pbf_density_boundary, cuda_simulation.cu:665-701 */
            for (uint64_t a = c->fb_off[i]; a < c->fb_off[i + 1]; ++a) {
                uint32_t b = c->fb[a];
                float distance = dist3(c->pred + 3 * i, c->bpos + 3 * (size_t)b);
                dens += (rho0 * c->bvol[b]) * k_poly6(&k, distance);
            }
        c->density[i] = dens;
        c->C[i] = dens / rho0 - 1.f;
    }
    return PBF_OK;
}

static void boundary_density_lambda(orc_ctx* c);

/* Simulation::ComputeLambdas, Simulation.cpp:642-679 */
int orc_compute_lambdas(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    const float epsilon = c->P.epsilon;
    const float inv_rho0 = 1.f / c->rest_density;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        float lam = -c->C[i];
        /* gradientC_i = (1/rho0) * Poly6_W_Gradient(vec3(0), 0, h): zero vector, :657.
         * (The spiky option would divide by zero here; the self term is defined as 0.) */
        float gradientC_sum = 0.f;
        if (k.grad_kernel == PBF_GRAD_POLY6) {
            float zero[3] = {0, 0, 0}, g0[3];
            k_grad(&k, zero, 0.f, g0);
            float gi[3] = {inv_rho0 * g0[0], inv_rho0 * g0[1], inv_rho0 * g0[2]};
            gradientC_sum = dot3(gi, gi);
        }
        const float* pi = c->pred + 3 * i;
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            const float* pj = c->pred + 3 * j;
            float diff[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            float distance = dist3(pi, pj);
            float g[3];
            k_grad(&k, diff, distance, g);
            float gj[3] = {inv_rho0 * g[0], inv_rho0 * g[1], inv_rho0 * g[2]};
            float dot_value = dot3(gj, gj);
            gradientC_sum += dot_value;
        }
        if (c->nb) /* pbf_lambda_boundary, cuda_simulation.cu:846-884 */
            for (uint64_t a = c->fb_off[i]; a < c->fb_off[i + 1]; ++a) {
                uint32_t b = c->fb[a];
                const float* pb = c->bpos + 3 * (size_t)b;
                float diff[3] = {pi[0] - pb[0], pi[1] - pb[1], pi[2] - pb[2]};
                float distance = dist3(pi, pb);
                float g[3];
                k_grad(&k, diff, distance, g);
                float s = inv_rho0 * ((c->rest_density * c->bvol[b]) / c->mass[i]);
                float gj[3] = {s * g[0], s * g[1], s * g[2]};
                gradientC_sum += dot3(gj, gj);
            }
        lam /= gradientC_sum + epsilon;
        c->lambda[i] = lam;
    }
    if (c->nb) boundary_density_lambda(c);
    return PBF_OK;
}

/* Simulation::ComputeSPHParticlesCorrection, Simulation.cpp:681-712, then (build-side,
 * no reference counterpart) the optional container clamp */
int orc_compute_correction(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    const float inv_rho0 = 1.f / c->rest_density;
    const float w_dq = k_poly6(&k, c->P.scorr_dq * c->P.effective_radius); /* :700 */
    const double scorr_k = -c->P.scorr_k;
    /* scorr * x * dt (Simulation.cpp:702); for scorr_dt_power != 1 (build-side) dt is
     * replaced by (double)dt multiplied that many times */
    double dt_pow = 1.0;
    for (int t = 0; t < c->P.scorr_dt_power; ++t) dt_pow = dt_pow * (double)dt;
    if (c->P.scorr_dt_power == 1) dt_pow = (double)dt;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        const float* pi = c->pred + 3 * i;
        float* np = c->newp + 3 * i;
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            const float* pj = c->pred + 3 * j;
            float diff[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            float distance = dist3(pi, pj);
            /* artificial pressure, :698-702 */
            double scorr = scorr_k;
            double x = k_poly6(&k, distance) / w_dq; /* float / float, then promoted */
            x = pow(x, 4);
            scorr = scorr * x * dt_pow;
            float g[3];
            k_grad(&k, diff, distance, g);
            float s = inv_rho0 * (c->lambda[i] + c->lambda[j] + (float)scorr);
            np[0] += s * g[0]; np[1] += s * g[1]; np[2] += s * g[2];
        }
        if (c->nb) /* pbf_correction_boundary, cuda_simulation.cu:990-1046: no s_corr term */
            for (uint64_t a = c->fb_off[i]; a < c->fb_off[i + 1]; ++a) {
                uint32_t b = c->fb[a];
                const float* pb = c->bpos + 3 * (size_t)b;
                float diff[3] = {pi[0] - pb[0], pi[1] - pb[1], pi[2] - pb[2]};
                float distance = dist3(pi, pb);
                float g[3];
                k_grad(&k, diff, distance, g);
                float s = inv_rho0 * (c->lambda[i] + c->blambda[b]);
                np[0] += s * g[0]; np[1] += s * g[1]; np[2] += s * g[2];
            }
        if (c->P.container_enabled) {
            for (int a = 0; a < 3; ++a) {
                if (np[a] < c->P.container_min[a]) np[a] = c->P.container_min[a];
                if (np[a] > c->P.container_max[a]) np[a] = c->P.container_max[a];
            }
        }
    }
    return PBF_OK;
}

/* Simulation::UpdatePredictPosition, Simulation.cpp:714-721 */
int orc_update_predict_position(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    memcpy(c->pred, c->newp, (size_t)c->n * 12);
    return PBF_OK;
}

/* build-side (PARITY UNPINNED): vorticity confinement and XSPH viscosity after
 * Macklin & Mueller 2013 eqs. 15-17, on the frozen neighbour lists and the updated
 * positions, with m_j / rho0 volume weights:
 *   omega_i = sum_j (v_j - v_i) x gradW(p_i - p_j)
 *   eta_i   = sum_j (m_j / rho0) |omega_j| gradW(p_i - p_j);  N = eta / |eta|
 *   v_i    += dt * eps_v * (N x omega_i)
 *   v_i    <- v_i + c * sum_j (m_j / rho0) (v_j - v_i) W(|p_i - p_j|)      (Jacobi) */
int orc_vorticity_omega(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        const float* pi = c->pos + 3 * i;
        const float* vi = c->vel + 3 * i;
        float w[3] = {0, 0, 0};
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            const float* pj = c->pos + 3 * j;
            const float* vj = c->vel + 3 * j;
            float diff[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            float distance = dist3(pi, pj);
            float g[3];
            k_grad(&k, diff, distance, g);
            float u[3] = {vj[0] - vi[0], vj[1] - vi[1], vj[2] - vi[2]};
            w[0] += u[1] * g[2] - g[1] * u[2];
            w[1] += u[2] * g[0] - g[2] * u[0];
            w[2] += u[0] * g[1] - g[0] * u[1];
        }
        c->omega[3 * i] = w[0]; c->omega[3 * i + 1] = w[1]; c->omega[3 * i + 2] = w[2];
    }
    return PBF_OK;
}

int orc_vorticity_apply(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    const float inv_rho0 = 1.f / c->rest_density;
    const float eps_v = c->P.vorticity_eps;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        const float* pi = c->pos + 3 * i;
        float eta[3] = {0, 0, 0};
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            const float* pj = c->pos + 3 * j;
            float diff[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            float distance = dist3(pi, pj);
            float g[3];
            k_grad(&k, diff, distance, g);
            const float* wj = c->omega + 3 * j;
            float s = (c->mass[j] * inv_rho0) * sqrtf(dot3(wj, wj));
            eta[0] += s * g[0]; eta[1] += s * g[1]; eta[2] += s * g[2];
        }
        float len = sqrtf(dot3(eta, eta));
        float* v2 = c->vel2 + 3 * i;
        const float* v = c->vel + 3 * i;
        v2[0] = v[0]; v2[1] = v[1]; v2[2] = v[2];
        if (len > 0.f) {
            float N[3] = {eta[0] / len, eta[1] / len, eta[2] / len};
            const float* w = c->omega + 3 * i;
            float f[3] = {N[1] * w[2] - w[1] * N[2], N[2] * w[0] - w[2] * N[0],
                          N[0] * w[1] - w[0] * N[1]};
            float s = dt * eps_v;
            v2[0] = v[0] + s * f[0]; v2[1] = v[1] + s * f[1]; v2[2] = v[2] + s * f[2];
        }
    }
    memcpy(c->vel, c->vel2, (size_t)n * 12);
    return PBF_OK;
}

int orc_xsph(orc_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
    const kern_t k = make_kern(c->P.effective_radius, c->P.gradient_kernel);
    const float inv_rho0 = 1.f / c->rest_density;
    const float cx = c->P.xsph_c;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        const float* pi = c->pos + 3 * i;
        const float* vi = c->vel + 3 * i;
        float acc[3] = {0, 0, 0};
        for (uint64_t a = c->nbr_off[i]; a < c->nbr_off[i + 1]; ++a) {
            uint32_t j = c->nbr[a];
            const float* pj = c->pos + 3 * j;
            const float* vj = c->vel + 3 * j;
            float distance = dist3(pi, pj);
            float s = (c->mass[j] * inv_rho0) * k_poly6(&k, distance);
            acc[0] += s * (vj[0] - vi[0]); acc[1] += s * (vj[1] - vi[1]);
            acc[2] += s * (vj[2] - vi[2]);
        }
        float* v2 = c->vel2 + 3 * i;
        v2[0] = vi[0] + cx * acc[0]; v2[1] = vi[1] + cx * acc[1]; v2[2] = vi[2] + cx * acc[2];
    }
    memcpy(c->vel, c->vel2, (size_t)n * 12);
    return PBF_OK;
}

/* ParticleSet::Update, Particle.cpp:207-215 (true division by dt) */
int orc_update_velocity(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    const int n = (int)c->n;
#pragma omp parallel for schedule(static) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        float* v = c->vel + 3 * i;
        float* p = c->pos + 3 * i;
        const float* np = c->newp + 3 * i;
        v[0] = (np[0] - p[0]) / dt; v[1] = (np[1] - p[1]) / dt; v[2] = (np[2] - p[2]) / dt;
        p[0] = np[0]; p[1] = np[1]; p[2] = np[2];
    }
    return PBF_OK;
}

/* Simulation::ApplySolverResults -> ParticleSet::Update, Simulation.cpp:787-791,
 * Particle.cpp:207-215 (true division by dt) */
int orc_apply_solver_results(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    orc_update_velocity(c, dt);
    if (c->P.vorticity_eps != 0.f) {
        orc_vorticity_omega(c);
        orc_vorticity_apply(c, dt);
    }
    if (c->P.xsph_c != 0.f) orc_xsph(c);
    return PBF_OK;
}

/* Simulation::Step, Simulation.cpp:80-138 */
int orc_step(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    if (!c->pos && c->n) return PBF_ERR_NOT_INITIALIZED;
    orc_predict_positions(c, dt);
    orc_collision_detection(c, dt);
    int rc = orc_spatial_search(c);
    if (rc) return rc;
    if (c->first_frame) {
        if (c->P.rest_density_mode == PBF_RHO0_FIRST_FRAME_MEAN) orc_compute_rest_density(c);
        c->first_frame = 0;
    }
    for (int32_t it = 0; it < c->iterations; ++it) {
        orc_compute_density(c);
        orc_compute_lambdas(c);
        orc_compute_correction(c, dt);
        orc_update_predict_position(c);
    }
    orc_project_constraints(c, dt); /* Simulation.cpp:121; no-op over an empty list */
    orc_apply_solver_results(c, dt);
    return PBF_OK;
}

/* ---------------------------------------------------------------- getters */

/* ---------------------------------------------------------------- PBD granular / DEM
 * Build-side (PARITY UNPINNED): solve_pbd_dem of the reference's CUDA path
 * (cuda_simulation.cu:1730-2225; dead code there, call site commented out at
 * Simulation.cpp:202-213) restated in plain fp32 with helper_math's operator forms
 * (length = sqrtf(dot), dot = (x*x + y*y) + z*z).  Buckets are the ones of the last
 * SpatialSearch (membership frozen for the sub-step, like the CUDA path's sorted cells); the
 * 27-cell sweep is centred on the particle's CURRENT cell (calcGridPos(pos), :1887) and runs z
 * outermost, x innermost (:1891-1906), ascending index inside a cell. */

typedef struct { float two_r, mu_s, mu_k; } dem_t;

/* pbd_distance_correction (+_boundary), :1730-1865: one neighbour's term; w1 = its massInv */
static inline void dem_distance_term(const dem_t* D, const float pi[3], const float pj[3], float w0,
                                     float w1, float acc[3]) {
    float v[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
    float dist = sqrtf(dot3(v, v));
    float cj[3] = {0.f, 0.f, 0.f};
    if (dist <= D->two_r) {
        float w_sum = w0 + w1;
        float C = dist - D->two_r;
        float n[3] = {v[0] / dist, v[1] / dist, v[2] / dist};
        float s = -w0 * (1.f / w_sum) * C;
        cj[0] = s * n[0]; cj[1] = s * n[1]; cj[2] = s * n[2];
    }
    acc[0] += cj[0]; acc[1] += cj[1]; acc[2] += cj[2];
}

/* pbd_friction_correction (+_boundary), :1928-2082; d1 = neighbour's displacement this
 * sub-step (NULL for a boundary particle) */
static inline void dem_friction_term(const dem_t* D, const float pi[3], const float oi[3],
                                     const float pj[3], const float* oj, float w0, float w1,
                                     float acc[3]) {
    float v[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
    float dist = sqrtf(dot3(v, v));
    float cj[3] = {0.f, 0.f, 0.f};
    if (dist <= D->two_r) {
        float w_sum = w0 + w1;
        float n[3] = {v[0] / dist, v[1] / dist, v[2] / dist};
        float penetration = D->two_r - dist;
        float dx[3] = {pi[0] - oi[0], pi[1] - oi[1], pi[2] - oi[2]};
        if (oj) { /* (predict0 - original0) + (predict1 - original1), :1978 (sic: a sum) */
            dx[0] = dx[0] + (pj[0] - oj[0]); dx[1] = dx[1] + (pj[1] - oj[1]); dx[2] = dx[2] + (pj[2] - oj[2]);
        }
        float dn = dot3(dx, n);
        float t[3] = {dx[0] - dn * n[0], dx[1] - dn * n[1], dx[2] - dn * n[2]};
        float threshold = D->mu_s * penetration;
        float len = sqrtf(dot3(t, t));
        if (len > threshold) {
            float coeff = fminf(D->mu_k * penetration / len, 1.f);
            t[0] = coeff * t[0]; t[1] = coeff * t[1]; t[2] = coeff * t[2];
        }
        float s = -(w0 / w_sum);
        cj[0] += s * t[0]; cj[1] += s * t[1]; cj[2] += s * t[2];
    }
    acc[0] += cj[0]; acc[1] += cj[1]; acc[2] += cj[2];
}

/* one Jacobi pass: newp = pred + correction, then pred = newp (apply_correction, :1466-1488) */
static void dem_pass(orc_ctx* c, const dem_t* D, int friction) {
    const int n = (int)c->n;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c->threads)
    for (int i = 0; i < n; ++i) {
        const float* pi = c->pred + 3 * i;
        const float* oi = c->pos + 3 * i;
        const float w0 = 1.f / c->mass[i];
        int32_t g[3];
        orc_flooring(pi, c->P.grid_spacing, g);
        float corr[3] = {0.f, 0.f, 0.f};
        for (int32_t z = -1; z <= 1; ++z)
            for (int32_t y = -1; y <= 1; ++y)
                for (int32_t x = -1; x <= 1; ++x) {
                    const int32_t cell[3] = {g[0] + x, g[1] + y, g[2] + z};
                    bucket_t* bk = table_find(c, orc_hash_value(cell[0], cell[1], cell[2]), 0);
                    if (!bk) continue;
                    float part[3] = {0.f, 0.f, 0.f}; /* the per-cell `correction` / `result` */
                    for (int32_t pj = bk->head; pj >= 0; pj = c->next[pj]) {
                        const int32_t* cj = c->cell + 3 * pj; /* bucket cell, frozen at the search */
                        if (cj[0] != cell[0] || cj[1] != cell[1] || cj[2] != cell[2]) continue;
                        if (pj == i) continue; /* adds a zero term in the reference: no effect */
                        const float w1 = 1.f / c->mass[pj];
                        if (friction) dem_friction_term(D, pi, oi, c->pred + 3 * pj, c->pos + 3 * pj, w0, w1, part);
                        else dem_distance_term(D, pi, c->pred + 3 * pj, w0, w1, part);
                    }
                    corr[0] += part[0]; corr[1] += part[1]; corr[2] += part[2];
                }
        if (c->nb)
            for (int32_t z = -1; z <= 1; ++z)
                for (int32_t y = -1; y <= 1; ++y)
                    for (int32_t x = -1; x <= 1; ++x) {
                        const int32_t cell[3] = {g[0] + x, g[1] + y, g[2] + z};
                        uint32_t lo, hi;
                        bcell_range(c, cell, &lo, &hi);
                        float part[3] = {0.f, 0.f, 0.f};
                        for (uint32_t t = lo; t < hi; ++t) {
                            const uint32_t b = c->border[t];
                            const float w1 = 1.f / c->bmass[b];
                            if (friction) dem_friction_term(D, pi, oi, c->bpos + 3 * (size_t)b, NULL, w0, w1, part);
                            else dem_distance_term(D, pi, c->bpos + 3 * (size_t)b, w0, w1, part);
                        }
                        corr[0] += part[0]; corr[1] += part[1]; corr[2] += part[2];
                    }
        float* np = c->newp + 3 * i;
        np[0] = pi[0] + corr[0]; np[1] = pi[1] + corr[1]; np[2] = pi[2] + corr[2];
    }
    memcpy(c->pred, c->newp, (size_t)c->n * 12);
}

int orc_step_dem(orc_ctx* c, float dt, const pbf_dem_params* dem) {
    if (!c || !dem) return PBF_ERR_INVALID;
    if (!c->pos && c->n) return PBF_ERR_NOT_INITIALIZED;
    if (!(dem->particle_radius > 0.f) || dem->iterations < 0) return PBF_ERR_INVALID;
    orc_predict_positions(c, dt);
    orc_collision_detection(c, dt);
    int rc = orc_spatial_search(c); /* buckets (the neighbour lists are not used here) */
    if (rc) return rc;
    const dem_t D = {2.f * dem->particle_radius, dem->static_friction, dem->kinematic_friction};
    for (int32_t it = 0; it < dem->iterations; ++it) dem_pass(c, &D, 0);
    dem_pass(c, &D, 1);
    orc_update_velocity(c, dt); /* finalize_correction, :1491-1518 */
    return PBF_OK;
}

/* ---------------------------------------------------------------- generic constraints
 * Simulation::ProjectConstraints -> ConstraintSolver::SolveConstraints (Simulation.cpp:775-780,
 * ConstraintSolver.cpp:12-72) over DistanceConstraint (Constraints.cpp:61-120), operating on
 * m_new_positions.  (The reference's Constraint holds legacy Particle objects; here the two
 * particles are indices into the ParticleSet.)  Pinned by the reference itself: Constraints.cpp
 * and ConstraintSolver.cpp are compiled into oracle/_ref (oracle/Makefile) and
 * tests/test_oracle.py::test_constraint_solver_equals_compiled_reference compares bit for bit. */
#define CONSTRAINT_EPSILON 0.000000000000001f /* Constraints.cpp:5 */

int orc_set_distance_constraints(orc_ctx* c, const pbf_distance_constraint* list, uint32_t count, int32_t mode) {
    if (!c || (count && !list)) return PBF_ERR_INVALID;
    if (mode != PBF_CONSTRAINTS_PBD && mode != PBF_CONSTRAINTS_XPBD) return PBF_ERR_INVALID;
    free(c->cons); free(c->cons_lambda);
    c->cons = NULL; c->cons_lambda = NULL; c->ncons = 0; c->cons_mode = mode;
    if (!count) return PBF_OK;
    c->cons = (pbf_distance_constraint*)xcalloc(count, sizeof(*list));
    c->cons_lambda = (float*)xcalloc(count, 4);
    memcpy(c->cons, list, (size_t)count * sizeof(*list));
    for (uint32_t k = 0; k < count; ++k) { /* Constraint::setStiffness, Constraints.cpp:25-33 */
        float s = c->cons[k].stiffness;
        c->cons[k].stiffness = s > 1.f ? 1.f : (s < 0.f ? 0.f : s);
    }
    c->ncons = count;
    return PBF_OK;
}

int orc_project_constraints(orc_ctx* c, float dt) {
    if (!c) return PBF_ERR_INVALID;
    if (!c->ncons) return PBF_OK;
    for (uint32_t k = 0; k < c->ncons; ++k)
        if (c->cons[k].p0 >= c->n || c->cons[k].p1 >= c->n) return PBF_ERR_INVALID;
    const int xpbd = c->cons_mode == PBF_CONSTRAINTS_XPBD;
    if (xpbd) /* InitLambda + ComputeCompliance, ConstraintSolver.cpp:55-60 */
        for (uint32_t k = 0; k < c->ncons; ++k) c->cons_lambda[k] = 0.0f;
    for (int32_t it = 0; it < c->iterations; ++it)
        for (uint32_t k = 0; k < c->ncons; ++k) {
            const pbf_distance_constraint* q = &c->cons[k];
            float* p0 = c->newp + 3 * (size_t)q->p0;
            float* p1 = c->newp + 3 * (size_t)q->p1;
            const float w0 = 1.f / c->mass[q->p0], w1 = 1.f / c->mass[q->p1];
            float w_sum = w0 + w1;
            float distance = dist3(p0, p1);
            float C = distance - q->rest_length;
            float v[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
            if (distance < CONSTRAINT_EPSILON) distance = CONSTRAINT_EPSILON;
            if (!xpbd) { /* SolvePBDConstraint, Constraints.cpp:61-89 */
                float n[3] = {v[0] / distance, v[1] / distance, v[2] / distance};
                float s = (1.f / w_sum) * C;
                float corr[3] = {s * n[0], s * n[1], s * n[2]};
                float a = q->stiffness * -w0, b = q->stiffness * w1;
                p0[0] += a * corr[0]; p0[1] += a * corr[1]; p0[2] += a * corr[2];
                p1[0] += b * corr[0]; p1[1] += b * corr[1]; p1[2] += b * corr[2];
            } else { /* SolveXPBDConstraint, Constraints.cpp:91-120 */
                const float ct = q->compliance / (dt * dt);
                float delta_lambda = (-C - ct * c->cons_lambda[k]) / (w_sum + ct);
                float corr[3] = {(delta_lambda * v[0]) / distance, (delta_lambda * v[1]) / distance,
                                 (delta_lambda * v[2]) / distance};
                c->cons_lambda[k] += delta_lambda;
                float b = -w1;
                p0[0] += w0 * corr[0]; p0[1] += w0 * corr[1]; p0[2] += w0 * corr[2];
                p1[0] += b * corr[0]; p1[1] += b * corr[1]; p1[2] += b * corr[2];
            }
        }
    memcpy(c->pred, c->newp, (size_t)c->n * 12); /* the device path keeps ONE predicted/new array */
    return PBF_OK;
}

int orc_get_constraint_lambdas(orc_ctx* c, float* lambdas) {
    if (!c || !lambdas) return PBF_ERR_INVALID;
    if (c->ncons) memcpy(lambdas, c->cons_lambda, (size_t)c->ncons * 4);
    return PBF_OK;
}

int orc_download(orc_ctx* c, float* pos, float* vel, float* pred, float* newp) {
    if (!c) return PBF_ERR_INVALID;
    size_t b = (size_t)c->n * 12;
    if (pos) memcpy(pos, c->pos, b);
    if (vel) memcpy(vel, c->vel, b);
    if (pred) memcpy(pred, c->pred, b);
    if (newp) memcpy(newp, c->newp, b);
    return PBF_OK;
}

int orc_get_scalars(orc_ctx* c, float* density, float* C, float* lambda) {
    if (!c) return PBF_ERR_INVALID;
    size_t b = (size_t)c->n * 4;
    if (density) memcpy(density, c->density, b);
    if (C) memcpy(C, c->C, b);
    if (lambda) memcpy(lambda, c->lambda, b);
    return PBF_OK;
}

int orc_get_cells(orc_ctx* c, int32_t* cell_xyz, uint32_t* hash) {
    if (!c) return PBF_ERR_INVALID;
    if (cell_xyz) memcpy(cell_xyz, c->cell, (size_t)c->n * 12);
    if (hash) memcpy(hash, c->hash, (size_t)c->n * 4);
    return PBF_OK;
}

uint64_t orc_neighbor_total(orc_ctx* c) { return (c && c->nbr_off) ? c->nbr_off[c->n] : 0; }

uint32_t orc_neighbor_max(orc_ctx* c) {
    uint32_t m = 0;
    if (!c || !c->nbr_off) return 0;
    for (uint32_t i = 0; i < c->n; ++i) {
        uint64_t k = c->nbr_off[i + 1] - c->nbr_off[i];
        if (k > m) m = (uint32_t)k;
    }
    return m;
}

int orc_get_neighbors(orc_ctx* c, uint32_t* counts, uint32_t* neighbors, uint32_t stride) {
    if (!c || !counts) return PBF_ERR_INVALID;
    for (uint32_t i = 0; i < c->n; ++i) {
        uint64_t k = c->nbr_off[i + 1] - c->nbr_off[i];
        counts[i] = (uint32_t)k;
        if (neighbors) {
            if (k > stride) return PBF_ERR_CAPACITY;
            memcpy(neighbors + (size_t)i * stride, c->nbr + c->nbr_off[i], (size_t)k * 4);
        }
    }
    return PBF_OK;
}

int orc_get_stats(orc_ctx* c, pbf_stats* s) {
    if (!c || !s) return PBF_ERR_INVALID;
    memset(s, 0, sizeof(*s));
    s->n = c->n;
    s->rest_density = c->rest_density;
    double sa = 0, sc = 0, mx = 0, msum = 0, com[3] = {0, 0, 0}, vmax = 0;
    uint64_t tot = 0;
    for (uint32_t i = 0; i < c->n; ++i) {
        const float* p = c->pos + 3 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) { s->nonfinite++; continue; }
        double a = fabs((double)c->C[i]);
        sa += a; sc += c->C[i];
        if (a > mx) mx = a;
        double m = c->mass[i];
        msum += m;
        com[0] += m * p[0]; com[1] += m * p[1]; com[2] += m * p[2];
        const float* v = c->vel + 3 * i;
        double sp = sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);
        if (sp > vmax) vmax = sp;
        if (c->nbr_off) {
            uint64_t k = c->nbr_off[i + 1] - c->nbr_off[i];
            tot += k;
            if (k > s->max_neighbor_count) s->max_neighbor_count = (uint32_t)k;
        }
    }
    if (c->n) {
        s->mean_abs_C = sa / c->n; s->mean_C = sc / c->n;
        s->mean_neighbor_count = (double)tot / c->n;
    }
    s->max_abs_C = mx;
    if (msum > 0) { s->com[0] = com[0] / msum; s->com[1] = com[1] / msum; s->com[2] = com[2] / msum; }
    s->max_speed = vmax;
    return PBF_OK;
}

/* raw array access for the slab-decomposition tests: 0 pos, 1 vel, 2 pred (== new), 3 lambda,
 * 4 omega */
static float* orc_array(orc_ctx* c, int which, int* width) {
    switch (which) {
        case 0: *width = 3; return c->pos;
        case 1: *width = 3; return c->vel;
        case 2: *width = 3; return c->pred;
        case 3: *width = 1; return c->lambda;
        case 4: *width = 3; return c->omega;
        default: *width = 0; return NULL;
    }
}

int orc_set_array(orc_ctx* c, int which, const float* data) {
    int w;
    if (!c || !data) return PBF_ERR_INVALID;
    float* a = orc_array(c, which, &w);
    if (!a) return PBF_ERR_INVALID;
    memcpy(a, data, (size_t)c->n * w * 4);
    if (which == 2) memcpy(c->newp, data, (size_t)c->n * 12);
    return PBF_OK;
}

int orc_get_array(orc_ctx* c, int which, float* data) {
    int w;
    if (!c || !data) return PBF_ERR_INVALID;
    float* a = orc_array(c, which, &w);
    if (!a) return PBF_ERR_INVALID;
    memcpy(data, a, (size_t)c->n * w * 4);
    return PBF_OK;
}

float orc_rest_density(orc_ctx* c) { return c ? c->rest_density : 0.f; }
uint32_t orc_num_particles(orc_ctx* c) { return c ? c->n : 0; }
