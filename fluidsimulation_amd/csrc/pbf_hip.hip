// pbf_hip.hip -- MI355X (gfx950) Position-Based-Fluids sub-step behind include/pbf.h.
//
// Data layout in HBM (all arrays live in SORTED order: cell-major, original id inside a
// cell -- the order of the reference's hash buckets, NeighborSearch.cpp:154-185):
//   pos [cap] float4  (x, y, z, mass)         m_positions + m_mass
//   vel [cap] float4  (vx, vy, vz, 0)          m_velocity
//   prd [cap] float4  (x*, y*, z*, lambda)     m_predict_positions == m_new_positions, m_lambda
//   id  [cap] u32     original particle id
//   nbr [K][cap] u32  neighbour table, row k = k-th neighbour of every particle
//                      (m_search_cache transposed so a wave reads one row coalesced)
// pos/vel/prd/id are double-buffered: the counting sort gathers them into the other copy.
// prd has a third copy for the Jacobi hand-over (UpdatePredictPosition).
//
// Kernels are one thread per particle, 256-thread workgroups (4 wave64), no MFMA: every
// phase is HBM/L2-bandwidth or latency bound.  Compiled with -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "pbf.h"
#include "pbf_device.h"

using namespace pbf;

namespace {

constexpr int BLOCK = 256;
constexpr int SWEEP_BATCH = 4;                    // neighbour-row entries / gathers in flight
constexpr int SCAN_ITEMS = 8;                     // items per thread in the scan tiles
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;     // 2048 cells per workgroup
constexpr uint32_t MAX_SCAN_BLOCKS = 131072;      // second level: one workgroup loops over the tile sums
constexpr float REF_PI = 3.1415926f;              // reference common.h:8

enum Phase {
    PH_PREDICT = 0,
    PH_SORT,
    PH_NEIGHBORS,
    PH_REST_DENSITY,
    PH_LAMBDA,
    PH_DELTA,
    PH_FINALIZE,
    PH_VISCOSITY,
    PH_COUNT
};
const char* kPhaseNames[PH_COUNT] = {"predict", "sort",     "neighbors", "rest_density",
                                     "lambda",  "delta",    "finalize",  "viscosity"};

// ------------------------------------------------------------------------------------
// device flags written by kernels, read at sync
struct DevFlags {
    uint32_t neighbor_overflow;  // particles whose neighbour count exceeded K
    uint32_t max_count;
    uint32_t x_wrapped;          // a particle's cell lies outside the grid in x (its key wrapped)
};

// ------------------------------------------------------------------------------------
// upload / download helpers

__global__ void k_pack(const float* __restrict__ pos3, const float* __restrict__ vel3,
                       const float* __restrict__ mass, float uniform_mass, uint32_t n,
                       float4* __restrict__ pos, float4* __restrict__ vel, float4* __restrict__ prd,
                       uint32_t* __restrict__ id) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float x = pos3[3 * (size_t)i], y = pos3[3 * (size_t)i + 1], z = pos3[3 * (size_t)i + 2];
    float m = mass ? mass[i] : uniform_mass;
    pos[i] = make_float4(x, y, z, m);
    prd[i] = make_float4(x, y, z, 0.f);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vel3) {
        v.x = vel3[3 * (size_t)i];
        v.y = vel3[3 * (size_t)i + 1];
        v.z = vel3[3 * (size_t)i + 2];
    }
    vel[i] = v;
    id[i] = i;
}

// scatter a float4 array's xyz back to original-id order as float triples
__global__ void k_unpack3(const float4* __restrict__ a, const uint32_t* __restrict__ id, uint32_t n,
                          float* __restrict__ out3) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (id[i] & 0x80000000u) return;  // slab ghost
    float4 v = a[i];
    size_t o = 3 * (size_t)id[i];
    out3[o] = v.x;
    out3[o + 1] = v.y;
    out3[o + 2] = v.z;
}

// mass (pos.w) back to original-id order
__global__ void k_unpack_w(const float4* __restrict__ a, const uint32_t* __restrict__ id, uint32_t n,
                           float* __restrict__ out) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (id[i] & 0x80000000u) return;  // slab ghost
    out[id[i]] = a[i].w;
}

__global__ void k_unpack_scalars(const float* __restrict__ dens, const float* __restrict__ C,
                                 const float4* __restrict__ prd, const uint32_t* __restrict__ id,
                                 uint32_t n, float* __restrict__ o_dens, float* __restrict__ o_C,
                                 float* __restrict__ o_lam) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t o = id[i];
    o_dens[o] = dens[i];
    o_C[o] = C[i];
    o_lam[o] = prd[i].w;
}

// ------------------------------------------------------------------------------------
// a6 + a7: Simulation::PredictPositions (Simulation.cpp:578-599) fused with
// Simulation::CollisionDetection / ParticleSet::OnCollision (Simulation.cpp:723-751,
// Particle.cpp:217-338), then NeighborSearch::Flooring + cell histogram (a8, first third).

// ------------------------------------------------------------------------------------
// wave-aggregated cell atomics.  Particles arrive nearly cell-sorted (last sub-step's order),
// so equal keys sit in runs of adjacent lanes: one ballot finds the run heads, the head lane
// issues ONE atomic for the whole run (about 5x fewer atomics at 4.8 particles per cell).
// Equal keys that are not adjacent simply issue separate atomics -- still correct.

struct RunInfo {
    uint32_t start;   // lane of the run head
    uint32_t len;     // lanes in the run
    bool head;
};

__device__ __forceinline__ RunInfo key_run(uint32_t key, bool active) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t prev = __shfl_up(key, 1, 64);
    bool prev_active = __shfl_up((int)active, 1, 64) != 0;
    bool head = active && (lane == 0 || !prev_active || prev != key);
    // a run also ends where an inactive lane begins: mark those as heads for the end search
    unsigned long long heads = __ballot(head || !active);
    unsigned long long below = heads & ((2ull << lane) - 1ull);       // bits 0..lane
    uint32_t start = 63u - (uint32_t)__builtin_clzll(below | 1ull);
    if (!(below)) start = 0;
    unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
    uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;
    RunInfo r;
    r.start = start;
    r.len = end - start;
    r.head = head;
    return r;
}

__device__ __forceinline__ void histogram_add(uint32_t* cell_count, uint32_t key, bool active) {
    RunInfo r = key_run(key, active);
    if (r.head) atomicAdd(&cell_count[key], r.len);
}

// returns this lane's slot: the run head reserves r.len consecutive slots
__device__ __forceinline__ uint32_t cursor_take(uint32_t* cursor, uint32_t key, bool active) {
    RunInfo r = key_run(key, active);
    uint32_t base = 0;
    if (r.head) base = atomicAdd(&cursor[key], r.len);
    base = __shfl(base, (int)r.start, 64);
    return base + ((threadIdx.x & 63u) - r.start);
}

__device__ __forceinline__ void collide(const ColliderSet& cs, float px, float py, float pz,
                                        float dt, float& vx, float& vy, float& vz, float& qx,
                                        float& qy, float& qz) {
    for (uint32_t c = 0; c < cs.count; ++c) {
        const pbf_collider& col = cs.c[c];
        float nx, ny, nz;
        if (col.type == PBF_COLLIDER_PLANE) {
            // CollisionDetection::PointPlaneIntersection, CollisionDetection.cpp:62-67
            float dot_val = dot3(col.a[0], col.a[1], col.a[2], qx, qy, qz);
            if (!(dot_val <= col.d)) continue;
            nx = col.a[0];
            ny = col.a[1];
            nz = col.a[2];
        } else {
            // CollisionDetection::PointAABBIntersection, CollisionDetection.cpp:29-34
            bool inside = qx < col.b[0] && qx > col.a[0] && qy < col.b[1] && qy > col.a[1] &&
                          qz < col.b[2] && qz > col.a[2];
            if (!inside) continue;
            // Particle.cpp:272-303: face with the LARGEST distance (sic)
            float dmax[3] = {col.b[0] - qx, col.b[1] - qy, col.b[2] - qz};
            float dmin[3] = {qx - col.a[0], qy - col.a[1], qz - col.a[2]};
            float min_dist = dmax[0];
            int pick = 0;
            if (min_dist < dmax[1]) min_dist = dmax[1], pick = 1;
            if (min_dist < dmax[2]) min_dist = dmax[2], pick = 2;
            if (min_dist < dmin[0]) min_dist = dmin[0], pick = 3;
            if (min_dist < dmin[1]) min_dist = dmin[1], pick = 4;
            if (min_dist < dmin[2]) min_dist = dmin[2], pick = 5;
            float sgn = pick < 3 ? 1.f : -1.f;
            int ax = pick % 3;
            nx = ax == 0 ? sgn : 0.f;
            ny = ax == 1 ? sgn : 0.f;
            nz = ax == 2 ? sgn : 0.f;
        }
        // v_r = v - 2.f * dot(v, n) * n  (Particle.cpp:306,325)
        float s = 2.f * dot3(vx, vy, vz, nx, ny, nz);
        vx = vx - s * nx;
        vy = vy - s * ny;
        vz = vz - s * nz;
        // restitution (Particle.cpp:333)
        vx = cs.restitution * vx;
        vy = cs.restitution * vy;
        vz = cs.restitution * vz;
        // re-prediction from the old position (Particle.cpp:336)
        qx = px + dt * vx;
        qy = py + dt * vy;
        qz = pz + dt * vz;
    }
}

// a6 + a7 for one particle: PredictPositions then the collider loop
__device__ __forceinline__ void predict_one(const float4& p, const float4& v, float dt, float gx,
                                            float gy, float gz, const ColliderSet& cs, float& vx,
                                            float& vy, float& vz, float& qx, float& qy, float& qz) {
    float m = p.w;
    float minv = 1.f / m;  // m_massInv, Particle.cpp:184
    // m_force = m_mass * vec3(0, g, 0); v = v + dt * massInv * force
    float fx = m * gx, fy = m * gy, fz = m * gz;
    float s = dt * minv;
    vx = v.x + s * fx;
    vy = v.y + s * fy;
    vz = v.z + s * fz;
    qx = p.x + dt * vx;
    qy = p.y + dt * vy;
    qz = p.z + dt * vz;
    collide(cs, p.x, p.y, p.z, dt, vx, vy, vz, qx, qy, qz);
}

__global__ __launch_bounds__(BLOCK) void k_predict(const float4* __restrict__ pos,
                                                   float4* __restrict__ vel,
                                                   float4* __restrict__ prd,
                                                   uint32_t* __restrict__ key,
                                                   uint32_t* __restrict__ cell_count, uint32_t n,
                                                   float dt, float gx, float gy, float gz,
                                                   ColliderSet cs, GridConsts G, int keep_vel) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n;
    uint32_t k = 0xffffffffu;
    if (active) {
        float4 p = pos[i];
        float4 v = vel[i];
        float vx, vy, vz, qx, qy, qz;
        predict_one(p, v, dt, gx, gy, gz, cs, vx, vy, vz, qx, qy, qz);
        // the predicted velocity is dead when the sub-step ends in the fused velocity update
        // (v = (new - p) / dt reads no velocity): then it is neither stored nor sorted
        if (keep_vel) vel[i] = make_float4(vx, vy, vz, 0.f);
        prd[i] = make_float4(qx, qy, qz, 0.f);
        k = cell_key(G, floor_cell(qx, G.spacing), floor_cell(qy, G.spacing),
                     floor_cell(qz, G.spacing));
        key[i] = k;
    }
    if (cell_count) histogram_add(cell_count, k, active);
}

// re-key without predicting (pbf_build_grid called on its own)
__global__ __launch_bounds__(BLOCK) void k_rekey(const float4* __restrict__ prd,
                                                 uint32_t* __restrict__ key,
                                                 uint32_t* __restrict__ cell_count, uint32_t n,
                                                 GridConsts G, uint32_t first = 0) {
    uint32_t i = first + blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n;
    uint32_t k = 0xffffffffu;
    if (active) {
        float4 q = prd[i];
        k = cell_key(G, floor_cell(q.x, G.spacing), floor_cell(q.y, G.spacing),
                     floor_cell(q.z, G.spacing));
        key[i] = k;
    }
    histogram_add(cell_count, k, active);
}

// ------------------------------------------------------------------------------------
// a8, second third: exclusive prefix scan of the cell histogram.
// wave64 inclusive scan with DPP-backed shuffles, LDS across the 4 waves of a workgroup.

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over the workgroup; returns the exclusive prefix
// and the workgroup total in `total`
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds_waves,
                                                    uint32_t& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) lds_waves[wid] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        uint32_t s = lds_waves[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// loads SCAN_ITEMS consecutive counters of one thread as two 16-byte vectors when in range
__device__ __forceinline__ void load_items(const uint32_t* __restrict__ a, uint32_t base,
                                           uint32_t n, uint32_t (&c)[SCAN_ITEMS]) {
    if (base + SCAN_ITEMS <= n) {
        const uint4 lo = *reinterpret_cast<const uint4*>(a + base);
        const uint4 hi = *reinterpret_cast<const uint4*>(a + base + 4);
        c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w;
        c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) c[k] = (base + k < n) ? a[base + k] : 0u;
    }
}

__global__ __launch_bounds__(BLOCK) void k_scan_tile_sums(const uint32_t* __restrict__ count,
                                                          uint32_t ncell,
                                                          uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t lds_waves[BLOCK / 64];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS];
    load_items(count, base, ncell, c);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += c[k];
    uint32_t total;
    block_excl_scan(s, lds_waves, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one workgroup scans up to MAX_SCAN_BLOCKS tile sums in place (exclusive)
__global__ __launch_bounds__(BLOCK) void k_scan_sums(uint32_t* __restrict__ tile_sums,
                                                     uint32_t ntiles) {
    __shared__ uint32_t lds_waves[BLOCK / 64];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < ntiles ? tile_sums[i] : 0u;
        uint32_t total;
        uint32_t ex = block_excl_scan(v, lds_waves, total);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += total;
    }
}

// writes cell_start (exclusive prefix), a cursor copy for the scatter, and clears the
// histogram for the next sub-step (so no per-step memset is needed)
// FOLD: tile_sums holds the raw per-tile totals and every workgroup sums the ones in front of it
// itself (grids of up to FOLD_MAX_TILES tiles: saves the single-workgroup k_scan_sums launch);
// otherwise tile_sums already holds exclusive prefixes.  Workgroup 0 also clears the device
// flags of the neighbour build that follows (no per-step memset).
constexpr uint32_t FOLD_MAX_TILES = 2048;

template <bool FOLD>
__global__ __launch_bounds__(BLOCK) void k_scan_apply(uint32_t* __restrict__ count, uint32_t ncell,
                                                      const uint32_t* __restrict__ tile_sums,
                                                      uint32_t* __restrict__ cell_start,
                                                      uint32_t* __restrict__ cursor,
                                                      DevFlags* __restrict__ flags) {
    static_assert(SCAN_ITEMS == 8, "vector path below moves two uint4 per thread");
    __shared__ uint32_t lds_waves[BLOCK / 64];
    if (flags && blockIdx.x == 0 && threadIdx.x == 0) {
        flags->neighbor_overflow = 0u;
        flags->max_count = 0u;
        flags->x_wrapped = 0u;
    }
    uint32_t tile_base;
    if (FOLD) {
        uint32_t part = 0;
        for (uint32_t t = threadIdx.x; t < blockIdx.x; t += BLOCK) part += tile_sums[t];
        uint32_t tot;
        block_excl_scan(part, lds_waves, tot);
        tile_base = tot;
    } else {
        tile_base = tile_sums[blockIdx.x];
    }
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS];
    load_items(count, base, ncell, c);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += c[k];
    uint32_t total;
    uint32_t ex = block_excl_scan(s, lds_waves, total) + tile_base;
    uint32_t e[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        e[k] = ex;
        ex += c[k];
    }
    if (base + SCAN_ITEMS <= ncell) {
        const uint4 lo = make_uint4(e[0], e[1], e[2], e[3]), hi = make_uint4(e[4], e[5], e[6], e[7]);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(cell_start + base) = lo;
        *reinterpret_cast<uint4*>(cell_start + base + 4) = hi;
        *reinterpret_cast<uint4*>(cursor + base) = lo;
        *reinterpret_cast<uint4*>(cursor + base + 4) = hi;
        *reinterpret_cast<uint4*>(count + base) = z;
        *reinterpret_cast<uint4*>(count + base + 4) = z;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            uint32_t idx = base + k;
            if (idx < ncell) {
                cell_start[idx] = e[k];
                cursor[idx] = e[k];
                count[idx] = 0u;
            }
        }
    }
    // sentinel: cell_start[ncell] = n
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == BLOCK - 1) cell_start[ncell] = ex;
}

// ------------------------------------------------------------------------------------
// a8, last third: counting-sort scatter (arbitrary order inside a cell), then a rank-by-id
// pass inside each cell that makes the order deterministic and equal to a stable sort of
// the original indices -- the reference's bucket order -- fused with the reorder gather.

__global__ __launch_bounds__(BLOCK) void k_scatter(const uint32_t* __restrict__ key,
                                                   const uint32_t* __restrict__ id,
                                                   uint32_t* __restrict__ cursor, uint32_t n,
                                                   uint32_t* __restrict__ tmp_src,
                                                   uint32_t* __restrict__ tmp_id) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n;
    uint32_t k = active ? key[i] : 0xffffffffu;
    uint32_t slot = cursor_take(cursor, k, active);
    if (active) {
        tmp_src[slot] = i;
        tmp_id[slot] = id[i];
    }
}

__global__ __launch_bounds__(BLOCK) void k_reorder(
    const uint32_t* __restrict__ key, const uint32_t* __restrict__ cell_start,
    const uint32_t* __restrict__ tmp_src, const uint32_t* __restrict__ tmp_id, uint32_t n,
    const float4* __restrict__ pos_in, const float4* __restrict__ vel_in,
    const float4* __restrict__ prd_in, float4* __restrict__ pos_out, float4* __restrict__ vel_out,
    float4* __restrict__ prd_out, uint32_t* __restrict__ id_out, uint32_t* __restrict__ slot_of) {
    uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n) return;
    uint32_t src = tmp_src[s];
    uint32_t my = tmp_id[s];
    uint32_t k = key[src];
    uint32_t a = cell_start[k], b = cell_start[k + 1];
    // rank inside the cell by original id (bit 31 marks slab ghosts and is not part of the id)
    const uint32_t my_id = my & 0x7fffffffu;
    uint32_t rank = 0;
    for (uint32_t t = a; t < b; ++t) rank += ((tmp_id[t] & 0x7fffffffu) < my_id) ? 1u : 0u;
    uint32_t dst = a + rank;
    if (slot_of) slot_of[src] = dst;
    pos_out[dst] = pos_in[src];
    if (vel_out) vel_out[dst] = vel_in[src];
    prd_out[dst] = prd_in[src];
    id_out[dst] = my;
}

// ------------------------------------------------------------------------------------
// a8: the 27-cell sweep of NeighborSearch::SpatialSearch (NeighborSearch.cpp:187-234):
// x outermost, z innermost, bucket order inside a cell, j != i, dist2 <= h*h (inclusive).
// Cells are z-fastest in the key, so the visiting order equals the reference's.

// Query particles q[0..n) and target particles t (cell_start is the TARGET set's) may be
// different sets sorted on the same grid (fluid <-> boundary particles); SELF = same set, the
// particle itself is left out.
template <bool SELF>
__device__ __forceinline__ void neighbors_global_path(uint32_t i, const float4* __restrict__ q,
                                                     const float4* __restrict__ prd,
                                                     const uint32_t* __restrict__ cell_start,
                                                     uint32_t n, uint32_t cap, uint32_t K,
                                                     GridConsts G, float square_h,
                                                     uint32_t* __restrict__ nbr,
                                                     uint32_t* __restrict__ nbr_count,
                                                     DevFlags* __restrict__ flags) {
    if (i >= n) return;
    const float4 xi = q[i];
    const int32_t cx = floor_cell(xi.x, G.spacing), cy = floor_cell(xi.y, G.spacing),
                  cz = floor_cell(xi.z, G.spacing);
    const uint32_t wz = (uint32_t)wrap(cz - G.oz, G.gz);
    // the three z-cells of one (dx,dy) are consecutive keys unless the z index wraps
    const bool zrun = wz >= 1u && wz + 1u < (uint32_t)G.gz;
    uint32_t cnt = 0;
    uint32_t* out = nbr + i;  // advances one row (cap entries) per accepted neighbour
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 pix = {xi.x, xi.x}, piy = {xi.y, xi.y}, piz = {xi.z, xi.z};
    // candidate range [a,b), in chunks of 32: first a branch-free test pass that records one
    // accept bit per candidate (4 loads in flight, two candidates per packed fp32 op -- each
    // half of a packed op rounds like the scalar op, so this is still glm's dot order), then an
    // emit pass over the set bits in ascending order (the reference's cache order).
    auto sweep = [&](uint32_t a, uint32_t b) {
        for (uint32_t s = a; s < b; s += 32) {
            const uint32_t len = b - s < 32u ? b - s : 32u;
            const uint32_t last = s + len - 1;
            uint32_t mask = 0;
            for (uint32_t u = 0; u < len; u += 4) {
                const uint32_t j = s + u;
                const uint32_t j1 = j + 1 < last ? j + 1 : last, j2 = j + 2 < last ? j + 2 : last,
                               j3 = j + 3 < last ? j + 3 : last;
                const float4 c0 = prd[j], c1 = prd[j1], c2 = prd[j2], c3 = prd[j3];
                // glm::distance2(pi, pj) = dot(pj - pi, pj - pi)
                f2 ex = (f2){c0.x, c1.x} - pix, ey = (f2){c0.y, c1.y} - piy, ez = (f2){c0.z, c1.z} - piz;
                const f2 da = (ex * ex + ey * ey) + ez * ez;
                ex = (f2){c2.x, c3.x} - pix;
                ey = (f2){c2.y, c3.y} - piy;
                ez = (f2){c2.z, c3.z} - piz;
                const f2 db = (ex * ex + ey * ey) + ez * ez;
                const uint32_t nib = (da.x <= square_h ? 1u : 0u) | (da.y <= square_h ? 2u : 0u) |
                                     (db.x <= square_h ? 4u : 0u) | (db.y <= square_h ? 8u : 0u);
                mask |= nib << u;
            }
            // drop the clamped tail duplicates and the particle itself (p_j == i, :216)
            if (len < 32u) mask &= (1u << len) - 1u;
            if (SELF) {
                const uint32_t self = i - s;  // wraps to a huge value when i < s
                if (self < 32u) mask &= ~(1u << self);
            }
            while (mask) {
                const uint32_t k = (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1u;
                if (cnt < K) {
                    *out = s + k;
                    out += cap;
                }
                ++cnt;
            }
        }
    };
    for (int dx = -1; dx <= 1; ++dx) {
        const uint32_t wx = (uint32_t)wrap(cx + dx - G.ox, G.gx);
        for (int dy = -1; dy <= 1; ++dy) {
            const uint32_t wy = (uint32_t)wrap(cy + dy - G.oy, G.gy);
            const uint32_t row = (wx * (uint32_t)G.gy + wy) * (uint32_t)G.gz;
            if (zrun) {
                sweep(cell_start[row + wz - 1u], cell_start[row + wz + 2u]);
            } else {
                for (int dz = -1; dz <= 1; ++dz) {
                    const uint32_t k = row + (uint32_t)wrap(cz + dz - G.oz, G.gz);
                    sweep(cell_start[k], cell_start[k + 1]);
                }
            }
        }
    }
    nbr_count[i] = cnt < K ? cnt : K;
    if (cnt > K) {  // rare: keep the per-particle atomics off the common path
        atomicAdd(&flags->neighbor_overflow, 1u);
        atomicMax(&flags->max_count, cnt);
    }
}


// LDS-staged form of the same sweep.  A workgroup owns 256 consecutive sorted particles, i.e.
// a run of ~54 consecutive cells of one or two z-columns.  For each of the nine (dx,dy)
// columns the candidates of ALL its particles form ONE contiguous slot range
// [cell_start[kf+off-1], cell_start[kl+off+2]) (kf/kl = first/last key of the workgroup), so
// the workgroup stages nine coalesced ranges (~2400 positions) into LDS as SoA x[], y[], z[]
// once, and every thread then reads its candidates four at a time with ds_read_b128 -- the four
// x (y, z) of consecutive candidates arrive already packed for v_pk_* math, with no global
// load and no address arithmetic in the inner loop.  Visiting order and arithmetic are those
// of neighbors_global_path (which remains the path for workgroups that touch a grid edge,
// where keys wrap, or whose ranges exceed the LDS budget).
constexpr uint32_t NB_LDS_T = 3072;   // staged candidates per workgroup (36 KB as SoA fp32)

template <bool SELF>
__global__ __launch_bounds__(BLOCK) void k_neighbors(const float4* __restrict__ q,
                                                     const float4* __restrict__ prd,
                                                     const uint32_t* __restrict__ cell_start,
                                                     uint32_t n, uint32_t cap, uint32_t K,
                                                     GridConsts G, float square_h,
                                                     uint32_t* __restrict__ nbr,
                                                     uint32_t* __restrict__ nbr_count,
                                                     DevFlags* __restrict__ flags) {
    __shared__ __align__(16) float sx[NB_LDS_T + 32];
    __shared__ __align__(16) float sy[NB_LDS_T + 32];
    __shared__ __align__(16) float sz[NB_LDS_T + 32];
    __shared__ uint32_t sA[9], sBase[10], sKey[2];
    __shared__ int sOk;

    const uint32_t i0 = xcd_block(blockIdx.x, gridDim.x) * BLOCK;
    const uint32_t i = i0 + threadIdx.x;
    const bool active = i < n;
    const uint32_t ilast = (i0 + BLOCK <= n ? i0 + BLOCK : n) - 1;

    float4 xi = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t key = 0;
    bool interior = true;
    if (active) {
        xi = q[i];
        const int32_t ux = floor_cell(xi.x, G.spacing) - G.ox;
        if (SELF && (ux < 0 || ux >= G.gx)) flags->x_wrapped = 1u;   // cell planes are no longer slot runs
        const uint32_t wx = (uint32_t)wrap(ux, G.gx);
        const uint32_t wy = (uint32_t)wrap(floor_cell(xi.y, G.spacing) - G.oy, G.gy);
        const uint32_t wz = (uint32_t)wrap(floor_cell(xi.z, G.spacing) - G.oz, G.gz);
        key = (wx * (uint32_t)G.gy + wy) * (uint32_t)G.gz + wz;
        interior = wx >= 1u && wx + 1u < (uint32_t)G.gx && wy >= 1u && wy + 1u < (uint32_t)G.gy &&
                   wz >= 1u && wz + 1u < (uint32_t)G.gz;
        if (i == i0) sKey[0] = key;
        if (i == ilast) sKey[1] = key;
    }
    if (threadIdx.x == 0) sOk = 1;
    __syncthreads();
    bool lds_path = true;
    {
        if (threadIdx.x < 9) {
            // key window of the workgroup shifted to column (dx,dy), clamped to the grid: every
            // interior particle's three cells lie inside it (edge particles take the global path)
            const int dx = (int)threadIdx.x / 3 - 1, dy = (int)threadIdx.x % 3 - 1;
            const int32_t off = (dx * G.gy + dy) * G.gz;
            const int32_t ncell = G.gx * G.gy * G.gz;
            int32_t lo = (int32_t)sKey[0] + off - 1, hi = (int32_t)sKey[1] + off + 2;
            lo = lo < 0 ? 0 : (lo > ncell ? ncell : lo);
            hi = hi < lo ? lo : (hi > ncell ? ncell : hi);
            const uint32_t A = cell_start[lo];
            const uint32_t B = cell_start[hi];
            sA[threadIdx.x] = A;
            sBase[threadIdx.x + 1] = B - A;   // length for now
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            sBase[0] = 0;
            for (int r = 0; r < 9; ++r) {
                const uint32_t len = sBase[r + 1];
                sBase[r] = acc;
                acc += len;
            }
            sBase[9] = acc;
            if (acc > NB_LDS_T) sOk = 0;
        }
        __syncthreads();
        lds_path = sOk != 0;
    }
    if (!lds_path) {
        neighbors_global_path<SELF>(i, q, prd, cell_start, n, cap, K, G, square_h, nbr, nbr_count, flags);
        return;
    }
    // stage the nine ranges, coalesced
    for (int r = 0; r < 9; ++r) {
        const uint32_t A = sA[r], base = sBase[r], len = sBase[r + 1] - base;
        for (uint32_t m = threadIdx.x; m < len; m += BLOCK) {
            const float4 v = prd[A + m];
            sx[base + m] = v.x;
            sy[base + m] = v.y;
            sz[base + m] = v.z;
        }
    }
    __syncthreads();
    if (!active) return;
    if (!interior) {  // grid-edge cell: its neighbour keys wrap, the staged window does not apply
        neighbors_global_path<SELF>(i, q, prd, cell_start, n, cap, K, G, square_h, nbr, nbr_count, flags);
        return;
    }

    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 pix = {xi.x, xi.x, xi.x, xi.x}, piy = {xi.y, xi.y, xi.y, xi.y},
             piz = {xi.z, xi.z, xi.z, xi.z};
    uint32_t cnt = 0;
    uint32_t* out = nbr + i;
    for (int r = 0; r < 9; ++r) {   // r = (dx+1)*3 + (dy+1): x outermost, then y; z inside the run
        const int dx = r / 3 - 1, dy = r % 3 - 1;
        const int32_t off = (dx * G.gy + dy) * G.gz;
        const uint32_t a = cell_start[(uint32_t)((int32_t)key + off - 1)];
        const uint32_t b = cell_start[(uint32_t)((int32_t)key + off + 2)];
        const uint32_t A = sA[r], base = sBase[r];
        const uint32_t la = base + (a - A), lb = base + (b - A);   // LDS index range
        const uint32_t to_slot = A - base;                          // LDS index -> sorted slot
        const uint32_t lself = base + (i - A);                      // only meaningful for r == 4
        for (uint32_t s = la & ~3u; s < lb; s += 32) {
            uint32_t mask = 0;
            for (uint32_t u = 0; u < 32u && s + u < lb; u += 4) {
                const f4 cx4 = *reinterpret_cast<const f4*>(&sx[s + u]);
                const f4 cy4 = *reinterpret_cast<const f4*>(&sy[s + u]);
                const f4 cz4 = *reinterpret_cast<const f4*>(&sz[s + u]);
                // glm::distance2(pi, pj) = dot(pj - pi, pj - pi), four candidates per op
                const f4 ex = cx4 - pix, ey = cy4 - piy, ez = cz4 - piz;
                const f4 d2 = (ex * ex + ey * ey) + ez * ez;
                const uint32_t nib = (d2.x <= square_h ? 1u : 0u) | (d2.y <= square_h ? 2u : 0u) |
                                     (d2.z <= square_h ? 4u : 0u) | (d2.w <= square_h ? 8u : 0u);
                mask |= nib << u;
            }
            if (s < la) mask &= ~((1u << (la - s)) - 1u);           // entries before the range
            if (lb - s < 32u) mask &= (1u << (lb - s)) - 1u;        // entries after it
            if (SELF && r == 4) {                                    // the particle itself (:216)
                const uint32_t self = lself - s;
                if (self < 32u) mask &= ~(1u << self);
            }
            while (mask) {
                const uint32_t k = (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1u;
                if (cnt < K) {
                    *out = s + k + to_slot;
                    out += cap;
                }
                ++cnt;
            }
        }
    }
    nbr_count[i] = cnt < K ? cnt : K;
    if (cnt > K) {
        atomicAdd(&flags->neighbor_overflow, 1u);
        atomicMax(&flags->max_count, cnt);
    }
}

// ------------------------------------------------------------------------------------
// a9: Simulation::ComputeRestDensity (Simulation.cpp:273-300): CURRENT positions with the
// neighbour lists of the predicted ones; mean taken serially in original-id order in fp32.

__global__ __launch_bounds__(BLOCK) void k_rest_density_terms(
    const float4* __restrict__ pos, const uint32_t* __restrict__ id,
    const uint32_t* __restrict__ nbr, const uint32_t* __restrict__ nbr_count, uint32_t n,
    uint32_t cap, KernConsts kc, float* __restrict__ dens_by_id) {
    uint32_t i = xcd_block(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 pi = pos[i];
    float dens = pi.w * kc.w_zero;
    for_each_neighbor<SWEEP_BATCH>(i, nbr_count[i], nbr, cap, pos, [&](uint32_t, const float4& pj) {
        float ex = pj.x - pi.x, ey = pj.y - pi.y, ez = pj.z - pi.z;
        float distance = sqrt_rn(dot3(ex, ey, ez, ex, ey, ez));
        dens += pj.w * poly6_strict(kc, distance);
    });
    dens_by_id[id[i]] = dens;
}

__global__ void k_rest_density_mean(const float* __restrict__ dens_by_id, uint32_t n,
                                    float* __restrict__ rho0) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float acc = 0.f;
    for (uint32_t i = 0; i < n; ++i) acc += dens_by_id[i];
    acc /= (float)n;
    *rho0 = acc;
}

// Sweep window: the sorted slots [cell_start[key_a], cell_start[key_b]) when a window is given
// (slab mode: the owned cell planes plus as many ghost planes as the iteration still needs; the
// grid key has x outermost, so a run of cell planes is a contiguous run of slots -- unless some
// particle's cell wrapped around the grid in x, which k_neighbors flags), else [0, n).
// Returns false for threads outside it.  Blocks are spread over the window XCD-aware.
__device__ __forceinline__ bool sweep_index(const uint32_t* __restrict__ win_start, uint32_t key_a,
                                            uint32_t key_b, const DevFlags* __restrict__ flags,
                                            uint32_t n, uint32_t& i) {
    uint32_t i0 = 0, cnt = n;
    if (win_start && !flags->x_wrapped) {
        i0 = win_start[key_a];
        cnt = win_start[key_b] - i0;
    }
    const uint32_t nbk = (cnt + BLOCK - 1) / BLOCK;
    if (blockIdx.x >= nbk) return false;
    const uint32_t k = xcd_block(blockIdx.x, nbk) * BLOCK + threadIdx.x;
    i = i0 + k;
    return k < cnt;
}

// ------------------------------------------------------------------------------------
// a13 + a14 fused: Simulation::ComputeDensity and ComputeLambdas (Simulation.cpp:607-679)
// share one pass over the neighbour row: rho_i and sum_j |gradW/rho0|^2 are independent.
// lambda is written into prd[i].w (no reader uses .w during this kernel).

// BND: boundary-particle terms of pbf_density_boundary / pbf_lambda_boundary
// (cuda_simulation.cu:665-701,846-884) over the fluid->boundary rows, after the fluid terms.
template <bool UNIFORM_MASS, int GRAD, bool BND>
__global__ __launch_bounds__(BLOCK) void k_lambda(float4* prd, const float4* __restrict__ pos,
                                                  const uint32_t* __restrict__ nbr,
                                                  const uint32_t* __restrict__ nbr_count,
                                                  uint32_t n, uint32_t cap, KernConsts kc,
                                                  const float* __restrict__ rho0_ptr,
                                                  float uniform_mass,
                                                  float* __restrict__ dens_out,
                                                  float* __restrict__ C_out,
                                                  const float4* __restrict__ bpos,
                                                  const uint32_t* __restrict__ fb,
                                                  const uint32_t* __restrict__ fb_count,
                                                  const uint32_t* __restrict__ win_start,
                                                  uint32_t win_a, uint32_t win_b,
                                                  const DevFlags* __restrict__ flags) {
    uint32_t i;
    if (!sweep_index(win_start, win_a, win_b, flags, n, i)) return;
    const float rho0 = *rho0_ptr;
    const float inv_rho0 = 1.f / rho0;
    const float4 xi = prd[i];
    float dens = (UNIFORM_MASS ? uniform_mass : pos[i].w) * kc.w_zero;
    float grad_sum = 0.f;  // self term: |(1/rho0) gradW(0)|^2 == 0 (Simulation.cpp:657-658)
    for_each_neighbor<SWEEP_BATCH>(i, nbr_count[i], nbr, cap, prd, [&](uint32_t j, const float4& xj) {
        float mj = UNIFORM_MASS ? uniform_mass : pos[j].w;
        float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        float w, gs;
        kernel_pair<GRAD>(kc, distance, w, gs);
        dens += mj * w;
        // gradientC_j = (1/rho0) * (gs * diff); sum += dot(gC, gC)
        float gx = inv_rho0 * (gs * dx), gy = inv_rho0 * (gs * dy), gz = inv_rho0 * (gs * dz);
        grad_sum += dot3(gx, gy, gz, gx, gy, gz);
    });
    if (BND) {
        const float mi = UNIFORM_MASS ? uniform_mass : pos[i].w;
        for_each_neighbor<SWEEP_BATCH>(i, fb_count[i], fb, cap, bpos, 0u, [&](uint32_t, const float4& pb) {
            float dx = xi.x - pb.x, dy = xi.y - pb.y, dz = xi.z - pb.z;
            float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
            float psi = rho0 * pb.w;  // rho0 * V_b
            dens += psi * poly6_strict(kc, distance);
            float gs = grad_scalar_strict<GRAD>(kc, distance);
            float s = inv_rho0 * (psi / mi);
            float gx = s * (gs * dx), gy = s * (gs * dy), gz = s * (gs * dz);
            grad_sum += dot3(gx, gy, gz, gx, gy, gz);
        });
    }
    float C = dens / rho0 - 1.f;
    float lam = -C;
    lam /= grad_sum + kc.epsilon;
    if (dens_out) {   // per-particle rho / C are outputs for the caller only: no kernel reads them
        dens_out[i] = dens;
        C_out[i] = C;
    }
    reinterpret_cast<float*>(prd)[4 * (size_t)i + 3] = lam;
}

// a15 + a16: Simulation::ComputeSPHParticlesCorrection (Simulation.cpp:681-712) writing the
// Jacobi hand-over copy directly (UpdatePredictPosition, :714-721), plus the build-side
// container clamp.  Accumulates onto new_i in neighbour-row order like the reference.

// FINAL: the last iteration of a sub-step without constraints / viscosity also performs
// ParticleSet::Update (k_finalize: v = (new - p) / dt; p = new) for its own particle -- the
// streaming pass over pos / vel / prd that k_finalize would make rides along with a kernel that
// is instruction-bound anyway.  No other thread reads pos or vel in this kernel.
template <int GRAD, bool BND, bool FINAL>
__global__ __launch_bounds__(BLOCK) void k_delta(const float4* __restrict__ prd,
                                                 float4* __restrict__ prd_out,
                                                 const uint32_t* __restrict__ nbr,
                                                 const uint32_t* __restrict__ nbr_count,
                                                 uint32_t n, uint32_t cap, KernConsts kc,
                                                 const float* __restrict__ rho0_ptr,
                                                 int container, float3_ cmin, float3_ cmax,
                                                 const float4* __restrict__ bprd,
                                                 const uint32_t* __restrict__ fb,
                                                 const uint32_t* __restrict__ fb_count,
                                                 float4* __restrict__ pos, float4* __restrict__ vel,
                                                 float dt, const uint32_t* __restrict__ win_start,
                                                 uint32_t win_a, uint32_t win_b,
                                                 const DevFlags* __restrict__ flags) {
    uint32_t i;
    if (!sweep_index(win_start, win_a, win_b, flags, n, i)) return;
    const float inv_rho0 = 1.f / *rho0_ptr;
    const float4 xi = prd[i];
    float nx = xi.x, ny = xi.y, nz = xi.z;
    for_each_neighbor<SWEEP_BATCH>(i, nbr_count[i], nbr, cap, prd, [&](uint32_t, const float4& xj) {
        float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        float w, gs;
        kernel_pair<GRAD>(kc, distance, w, gs);
        float sc = scorr_strict(kc, w);
        // result = (1/rho0) * (lambda_i + lambda_j + (float)scorr) * gradW
        float s = inv_rho0 * (xi.w + xj.w + sc);
        nx += s * (gs * dx);
        ny += s * (gs * dy);
        nz += s * (gs * dz);
    });
    if (BND) {  // pbf_correction_boundary (cuda_simulation.cu:990-1046): no s_corr term
        for_each_neighbor<SWEEP_BATCH>(i, fb_count[i], fb, cap, bprd, 0u, [&](uint32_t, const float4& pb) {
            float dx = xi.x - pb.x, dy = xi.y - pb.y, dz = xi.z - pb.z;
            float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
            float gs = grad_scalar_strict<GRAD>(kc, distance);
            float s = inv_rho0 * (xi.w + pb.w);
            nx += s * (gs * dx);
            ny += s * (gs * dy);
            nz += s * (gs * dz);
        });
    }
    if (container) {
        nx = nx < cmin.x ? cmin.x : nx;
        nx = nx > cmax.x ? cmax.x : nx;
        ny = ny < cmin.y ? cmin.y : ny;
        ny = ny > cmax.y ? cmax.y : ny;
        nz = nz < cmin.z ? cmin.z : nz;
        nz = nz > cmax.z ? cmax.z : nz;
    }
    prd_out[i] = make_float4(nx, ny, nz, xi.w);
    if (FINAL) {
        const float4 p = pos[i];
        vel[i] = make_float4((nx - p.x) / dt, (ny - p.y) / dt, (nz - p.z) / dt, 0.f);
        pos[i] = make_float4(nx, ny, nz, p.w);
    }
}

// ------------------------------------------------------------------------------------
// Boundary particles (PARITY UNPINNED): the Akinci coupling of the
// reference's CUDA kernels (cuda_simulation.cu K5 :331-374, K7 :1128-1212, K9 :1296-1384 and the
// boundary halves of K6/K8/K10) in the CPU path's arithmetic and frozen-list semantics, as
// checked against the CPU oracle.  The set is static: sorted once on the ctx grid as
//   bpos [nb] float4 (x, y, z, V_b)     bprd [nb] float4 (x, y, z, lambda_b)
// with three neighbour tables: bb (boundary-boundary, static), fb (fluid -> boundary) and bf
// (boundary -> fluid), the last two rebuilt with the fluid table every sub-step.

__global__ __launch_bounds__(BLOCK) void k_boundary_pack(const float* __restrict__ pos3,
                                                         const float* __restrict__ mass,
                                                         float uniform_mass, uint32_t nb,
                                                         float4* __restrict__ raw,
                                                         uint32_t* __restrict__ id) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nb) return;
    raw[i] = make_float4(pos3[3 * (size_t)i], pos3[3 * (size_t)i + 1], pos3[3 * (size_t)i + 2],
                         mass ? mass[i] : uniform_mass);
    id[i] = i;
}

// rank by index inside the cell (= stable sort) + gather, as k_reorder
__global__ __launch_bounds__(BLOCK) void k_boundary_reorder(
    const uint32_t* __restrict__ key, const uint32_t* __restrict__ cell_start,
    const uint32_t* __restrict__ tmp_src, const uint32_t* __restrict__ tmp_id, uint32_t nb,
    const float4* __restrict__ raw, float4* __restrict__ bprd, float* __restrict__ bmass,
    uint32_t* __restrict__ bid) {
    uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= nb) return;
    const uint32_t src = tmp_src[s], my = tmp_id[s];
    const uint32_t k = key[src];
    const uint32_t a = cell_start[k], b = cell_start[k + 1];
    uint32_t rank = 0;
    for (uint32_t t = a; t < b; ++t) rank += (tmp_id[t] < my) ? 1u : 0u;
    const uint32_t dst = a + rank;
    const float4 v = raw[src];
    bprd[dst] = make_float4(v.x, v.y, v.z, 0.f);
    bmass[dst] = v.w;
    bid[dst] = my;
}

// K5 compute_boundary_volume_d: V_b = m_b / (m_b W(0) + sum_b' m_b' W(|p_b - p_b'|))
__global__ __launch_bounds__(BLOCK) void k_boundary_volume(const float4* __restrict__ bprd,
                                                           const float* __restrict__ bmass,
                                                           const uint32_t* __restrict__ bb,
                                                           const uint32_t* __restrict__ bb_count,
                                                           uint32_t nb, KernConsts kc,
                                                           float4* __restrict__ bpos) {
    uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nb) return;
    const float4 pb = bprd[b];
    const float mb = bmass[b];
    float rho = mb * kc.w_zero;
    for_each_neighbor<SWEEP_BATCH>(b, bb_count[b], bb, nb, bprd, [&](uint32_t o, const float4& po) {
        float ex = po.x - pb.x, ey = po.y - pb.y, ez = po.z - pb.z;
        float distance = sqrt_rn(dot3(ex, ey, ez, ex, ey, ez));
        rho += bmass[o] * poly6_strict(kc, distance);
    });
    bpos[b] = make_float4(pb.x, pb.y, pb.z, mb / rho);
}

// the boundary-boundary part of K7 + K9: it comes first in both running sums and never changes
// (static positions), so the prefix of each sum is kept and the per-iteration kernel continues
// from it -- same fp32 summation order as summing everything each time
template <int GRAD>
__global__ __launch_bounds__(BLOCK) void k_boundary_static(const float4* __restrict__ bpos,
                                                           const float* __restrict__ bmass,
                                                           const uint32_t* __restrict__ bb,
                                                           const uint32_t* __restrict__ bb_count,
                                                           uint32_t nb, KernConsts kc,
                                                           const float* __restrict__ rho0_ptr,
                                                           float* __restrict__ rho_s,
                                                           float* __restrict__ grad_s) {
    uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nb) return;
    const float rho0 = *rho0_ptr;
    const float inv_rho0 = 1.f / rho0;
    const float4 pb = bpos[b];
    float rho = (rho0 * pb.w) * kc.w_zero;
    float grad_sum = 0.f;
    for_each_neighbor<SWEEP_BATCH>(b, bb_count[b], bb, nb, bpos, [&](uint32_t o, const float4& po) {
        float dx = pb.x - po.x, dy = pb.y - po.y, dz = pb.z - po.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        float psi = rho0 * po.w;
        rho += psi * poly6_strict(kc, distance);
        float gs = grad_scalar_strict<GRAD>(kc, distance);
        float s = inv_rho0 * (psi / bmass[o]);
        float gx = s * (gs * dx), gy = s * (gs * dy), gz = s * (gs * dz);
        grad_sum += dot3(gx, gy, gz, gx, gy, gz);
    });
    rho_s[b] = rho;
    grad_s[b] = grad_sum;
}

// K7 compute_boundary_density_d + K9 compute_boundary_lambdas_d on the current predicted
// fluid positions; lambda_b goes into bprd[b].w
template <bool UNIFORM_MASS, int GRAD>
__global__ __launch_bounds__(BLOCK) void k_boundary_lambda(
    const float4* __restrict__ bpos, float4* bprd, const float4* __restrict__ prd,
    const float4* __restrict__ pos, const uint32_t* __restrict__ bf,
    const uint32_t* __restrict__ bf_count, uint32_t nb, KernConsts kc,
    const float* __restrict__ rho0_ptr, float uniform_mass, const float* __restrict__ rho_s,
    const float* __restrict__ grad_s, float* __restrict__ bdens) {
    uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nb) return;
    const float rho0 = *rho0_ptr;
    const float inv_rho0 = 1.f / rho0;
    const float4 pb = bpos[b];
    float rho = rho_s[b];
    float grad_sum = grad_s[b];
    for_each_neighbor<SWEEP_BATCH>(b, bf_count[b], bf, nb, prd, 0u, [&](uint32_t j, const float4& xj) {
        float mj = UNIFORM_MASS ? uniform_mass : pos[j].w;
        float dx = pb.x - xj.x, dy = pb.y - xj.y, dz = pb.z - xj.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        rho += mj * poly6_strict(kc, distance);
        float gs = grad_scalar_strict<GRAD>(kc, distance);
        float gx = inv_rho0 * (gs * dx), gy = inv_rho0 * (gs * dy), gz = inv_rho0 * (gs * dz);
        grad_sum += dot3(gx, gy, gz, gx, gy, gz);
    });
    bdens[b] = rho;
    float C = rho / rho0 - 1.f;
    float lam = -C;
    lam /= grad_sum + kc.epsilon;
    reinterpret_cast<float*>(bprd)[4 * (size_t)b + 3] = lam;
}

__global__ __launch_bounds__(BLOCK) void k_boundary_unpack(const float4* __restrict__ bpos,
                                                           const float* __restrict__ bdens,
                                                           const float4* __restrict__ bprd,
                                                           const uint32_t* __restrict__ bid,
                                                           uint32_t nb, float* __restrict__ o_vol,
                                                           float* __restrict__ o_dens,
                                                           float* __restrict__ o_lam) {
    uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nb) return;
    const uint32_t o = bid[b];
    o_vol[o] = bpos[b].w;
    o_dens[o] = bdens[b];
    o_lam[o] = bprd[b].w;
}

// ------------------------------------------------------------------------------------
// PBD granular / DEM solver (PARITY UNPINNED): solve_pbd_dem of the
// reference's CUDA path (cuda_simulation.cu:1730-2225, dead code there) in plain fp32 on this
// repo's sorted arrays.  One thread per particle sweeps the 27 cells around its CURRENT cell
// (calcGridPos(pos), :1887) in z, y, x order over the buckets frozen at the sort; per-cell
// partial sums are added to the total like the reference's `corr += pbd_*_correction(cell)`.
// FRICTION = false: compute_distance_correction + apply_correction (:1867-1926,1466-1488);
// FRICTION = true : compute_friction_correction + apply_correction (:2084-2156).

struct DemConsts {
    float two_r, mu_s, mu_k;
};

template <bool FRICTION>
__device__ __forceinline__ void dem_term(const DemConsts& D, const float4& xi, const float4& oi,
                                         const float4& xj, const float4& oj, bool moving, float w0,
                                         float w1, float& ax, float& ay, float& az) {
    const float vx = xi.x - xj.x, vy = xi.y - xj.y, vz = xi.z - xj.z;
    const float dist = sqrt_rn(dot3(vx, vy, vz, vx, vy, vz));
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (dist <= D.two_r) {
        const float w_sum = w0 + w1;
        const float nx = vx / dist, ny = vy / dist, nz = vz / dist;
        if (!FRICTION) {
            const float C = dist - D.two_r;
            const float s = -w0 * (1.f / w_sum) * C;
            cx = s * nx;
            cy = s * ny;
            cz = s * nz;
        } else {
            const float penetration = D.two_r - dist;
            float dx = xi.x - oi.x, dy = xi.y - oi.y, dz = xi.z - oi.z;
            if (moving) {  // (predict0 - original0) + (predict1 - original1), :1978
                dx = dx + (xj.x - oj.x);
                dy = dy + (xj.y - oj.y);
                dz = dz + (xj.z - oj.z);
            }
            const float dn = dot3(dx, dy, dz, nx, ny, nz);
            float tx = dx - dn * nx, ty = dy - dn * ny, tz = dz - dn * nz;
            const float threshold = D.mu_s * penetration;
            const float len = sqrt_rn(dot3(tx, ty, tz, tx, ty, tz));
            if (len > threshold) {
                const float coeff = fminf(D.mu_k * penetration / len, 1.f);
                tx = coeff * tx;
                ty = coeff * ty;
                tz = coeff * tz;
            }
            const float s = -(w0 / w_sum);
            cx += s * tx;
            cy += s * ty;
            cz += s * tz;
        }
    }
    ax += cx;
    ay += cy;
    az += cz;
}

template <bool FRICTION, bool UNIFORM_MASS>
__global__ __launch_bounds__(BLOCK) void k_dem_pass(
    const float4* __restrict__ prd, float4* __restrict__ prd_out, const float4* __restrict__ pos,
    const uint32_t* __restrict__ cell_start, uint32_t n, GridConsts G, DemConsts D,
    float uniform_mass, const float4* __restrict__ bprd, const float* __restrict__ bmass,
    const uint32_t* __restrict__ b_cell_start) {
    const uint32_t i = xcd_block(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 xi = prd[i];
    const float4 oi = pos[i];
    const float w0 = 1.f / (UNIFORM_MASS ? uniform_mass : oi.w);
    const int32_t cx = floor_cell(xi.x, G.spacing), cy = floor_cell(xi.y, G.spacing),
                  cz = floor_cell(xi.z, G.spacing);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const uint32_t k = cell_key(G, cx + dx, cy + dy, cz + dz);
                const uint32_t a = cell_start[k], b = cell_start[k + 1];
                float px = 0.f, py = 0.f, pz = 0.f;
                for (uint32_t j = a; j < b; ++j) {
                    if (j == i) continue;
                    const float4 xj = prd[j];
                    const float4 oj = pos[j];
                    const float w1 = 1.f / (UNIFORM_MASS ? uniform_mass : oj.w);
                    dem_term<FRICTION>(D, xi, oi, xj, oj, true, w0, w1, px, py, pz);
                }
                sx += px;
                sy += py;
                sz += pz;
            }
    if (b_cell_start) {
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const uint32_t k = cell_key(G, cx + dx, cy + dy, cz + dz);
                    const uint32_t a = b_cell_start[k], b = b_cell_start[k + 1];
                    float px = 0.f, py = 0.f, pz = 0.f;
                    for (uint32_t j = a; j < b; ++j) {
                        const float4 xj = bprd[j];
                        dem_term<FRICTION>(D, xi, oi, xj, xj, false, w0, 1.f / bmass[j], px, py, pz);
                    }
                    sx += px;
                    sy += py;
                    sz += pz;
                }
    }
    prd_out[i] = make_float4(xi.x + sx, xi.y + sy, xi.z + sz, xi.w);
}

// ------------------------------------------------------------------------------------
// Generic constraint projection: DistanceConstraint::SolvePBDConstraint /
// SolveXPBDConstraint (Constraints.cpp:61-120) in the list order of
// ConstraintSolver::Solve{PBD,XPBD}Constraints (ConstraintSolver.cpp:35-72).  The host cuts the
// list into levels of constraints that share no particle; inside a level every constraint is
// independent, across levels the list order is kept per particle -> sequential Gauss-Seidel
// result, bit for bit.  Constraint arrays are stored in level order (SoA).

constexpr float CONSTRAINT_EPSILON = 0.000000000000001f;  // Constraints.cpp:5

struct ConstraintSoA {
    const uint32_t *p0, *p1;
    const float *rest, *stiffness, *compliance;
    float* lambda;
};

__global__ __launch_bounds__(BLOCK) void k_slot_by_id(const uint32_t* __restrict__ id, uint32_t n,
                                                      uint32_t* __restrict__ slot_by_id) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) slot_by_id[id[i]] = i;
}

template <bool XPBD>
__device__ __forceinline__ void solve_distance_constraint(uint32_t c, const ConstraintSoA& K,
                                                          const uint32_t* __restrict__ slot_by_id,
                                                          const float4* __restrict__ pos, float4* prd,
                                                          float dt) {
    const uint32_t s0 = slot_by_id[K.p0[c]], s1 = slot_by_id[K.p1[c]];
    float4 a = prd[s0], b = prd[s1];
    const float w0 = 1.f / pos[s0].w, w1 = 1.f / pos[s1].w;  // m_massInv
    const float w_sum = w0 + w1;
    // glm::distance(p0, p1) = length(p1 - p0)
    const float ex = b.x - a.x, ey = b.y - a.y, ez = b.z - a.z;
    float distance = sqrt_rn(dot3(ex, ey, ez, ex, ey, ez));
    const float C = distance - K.rest[c];
    const float vx = a.x - b.x, vy = a.y - b.y, vz = a.z - b.z;
    if (distance < CONSTRAINT_EPSILON) distance = CONSTRAINT_EPSILON;
    if (!XPBD) {
        const float nx = vx / distance, ny = vy / distance, nz = vz / distance;
        const float s = (1.f / w_sum) * C;
        const float cx = s * nx, cy = s * ny, cz = s * nz;
        const float st = K.stiffness[c];
        const float fa = st * -w0, fb = st * w1;
        a.x += fa * cx; a.y += fa * cy; a.z += fa * cz;
        b.x += fb * cx; b.y += fb * cy; b.z += fb * cz;
    } else {
        const float ct = K.compliance[c] / (dt * dt);  // Constraint::ComputeCompliance
        const float lam = K.lambda[c];
        const float delta_lambda = (-C - ct * lam) / (w_sum + ct);
        const float cx = (delta_lambda * vx) / distance, cy = (delta_lambda * vy) / distance,
                    cz = (delta_lambda * vz) / distance;
        K.lambda[c] = lam + delta_lambda;
        const float fb = -w1;
        a.x += w0 * cx; a.y += w0 * cy; a.z += w0 * cz;
        b.x += fb * cx; b.y += fb * cy; b.z += fb * cz;
    }
    prd[s0] = a;
    prd[s1] = b;
}

// one (large) level, many workgroups
template <bool XPBD>
__global__ __launch_bounds__(BLOCK) void k_constraints_level(uint32_t first, uint32_t count,
                                                             ConstraintSoA K,
                                                             const uint32_t* __restrict__ slot_by_id,
                                                             const float4* __restrict__ pos,
                                                             float4* prd, float dt) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < count) solve_distance_constraint<XPBD>(first + t, K, slot_by_id, pos, prd, dt);
}

// a run of small levels in ONE workgroup: a barrier (with its workgroup-scope fence) between
// levels instead of a launch per level -- a rope listed end to end is one level per constraint
template <bool XPBD>
__global__ __launch_bounds__(BLOCK) void k_constraints_levels(const uint32_t* __restrict__ level_start,
                                                              uint32_t first_level, uint32_t n_levels,
                                                              ConstraintSoA K,
                                                              const uint32_t* __restrict__ slot_by_id,
                                                              const float4* __restrict__ pos,
                                                              float4* prd, float dt) {
    for (uint32_t l = first_level; l < first_level + n_levels; ++l) {
        const uint32_t a = level_start[l], b = level_start[l + 1];
        for (uint32_t c = a + threadIdx.x; c < b; c += BLOCK)
            solve_distance_constraint<XPBD>(c, K, slot_by_id, pos, prd, dt);
        __threadfence_block();
        __syncthreads();
    }
}

// a18: ParticleSet::Update (Particle.cpp:207-215): v = (new - p) / dt; p = new
__global__ __launch_bounds__(BLOCK) void k_finalize(float4* __restrict__ pos,
                                                    float4* __restrict__ vel,
                                                    const float4* __restrict__ prd, uint32_t n,
                                                    float dt) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 p = pos[i];
    float4 q = prd[i];
    vel[i] = make_float4((q.x - p.x) / dt, (q.y - p.y) / dt, (q.z - p.z) / dt, 0.f);
    pos[i] = make_float4(q.x, q.y, q.z, p.w);
}

// ------------------------------------------------------------------------------------
// a20 (no reference counterpart): vorticity confinement + XSPH viscosity after Macklin &
// Mueller 2013 eqs. 15-17 on the frozen neighbour rows and the updated positions.

template <int GRAD>
__global__ __launch_bounds__(BLOCK) void k_vorticity(const float4* __restrict__ pos,
                                                     const float4* __restrict__ vel,
                                                     const uint32_t* __restrict__ nbr,
                                                     const uint32_t* __restrict__ nbr_count,
                                                     uint32_t n, uint32_t cap, KernConsts kc,
                                                     float4* __restrict__ omega) {
    uint32_t i = xcd_block(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 pi = pos[i];
    const float4 vi = vel[i];
    float wx = 0.f, wy = 0.f, wz = 0.f;
    for_each_neighbor<SWEEP_BATCH>(i, nbr_count[i], nbr, cap, pos, [&](uint32_t j, const float4& pj) {
        float4 vj = vel[j];
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        float gs = grad_scalar_strict<GRAD>(kc, distance);
        float gx = gs * dx, gy = gs * dy, gz = gs * dz;
        float ux = vj.x - vi.x, uy = vj.y - vi.y, uz = vj.z - vi.z;
        wx += uy * gz - gy * uz;
        wy += uz * gx - gz * ux;
        wz += ux * gy - gx * uy;
    });
    omega[i] = make_float4(wx, wy, wz, sqrtf(dot3(wx, wy, wz, wx, wy, wz)));
}

template <int GRAD>
__global__ __launch_bounds__(BLOCK) void k_vorticity_apply(
    const float4* __restrict__ pos, const float4* __restrict__ vel,
    const float4* __restrict__ omega, const uint32_t* __restrict__ nbr,
    const uint32_t* __restrict__ nbr_count, uint32_t n, uint32_t cap, KernConsts kc,
    const float* __restrict__ rho0_ptr, float dt, float eps_v, float4* __restrict__ vel_out) {
    uint32_t i = xcd_block(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float inv_rho0 = 1.f / *rho0_ptr;
    const float4 pi = pos[i];
    float ex = 0.f, ey = 0.f, ez = 0.f;
    for_each_neighbor<SWEEP_BATCH>(i, nbr_count[i], nbr, cap, pos, [&](uint32_t j, const float4& pj) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        float gs = grad_scalar_strict<GRAD>(kc, distance);
        float s = (pj.w * inv_rho0) * omega[j].w;
        ex += s * (gs * dx);
        ey += s * (gs * dy);
        ez += s * (gs * dz);
    });
    float len = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
    float4 v = vel[i];
    if (len > 0.f) {
        float Nx = ex / len, Ny = ey / len, Nz = ez / len;
        float4 w = omega[i];
        float fx = Ny * w.z - w.y * Nz;
        float fy = Nz * w.x - w.z * Nx;
        float fz = Nx * w.y - w.x * Ny;
        float s = dt * eps_v;
        v.x = v.x + s * fx;
        v.y = v.y + s * fy;
        v.z = v.z + s * fz;
    }
    vel_out[i] = v;
}

__global__ __launch_bounds__(BLOCK) void k_xsph(const float4* __restrict__ pos,
                                                const float4* __restrict__ vel,
                                                const uint32_t* __restrict__ nbr,
                                                const uint32_t* __restrict__ nbr_count,
                                                uint32_t n, uint32_t cap, KernConsts kc,
                                                const float* __restrict__ rho0_ptr, float c,
                                                float4* __restrict__ vel_out) {
    uint32_t i = xcd_block(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float inv_rho0 = 1.f / *rho0_ptr;
    const float4 pi = pos[i];
    const float4 vi = vel[i];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for_each_neighbor<SWEEP_BATCH>(i, nbr_count[i], nbr, cap, pos, [&](uint32_t j, const float4& pj) {
        float4 vj = vel[j];
        float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
        float distance = sqrt_rn(dot3(dx, dy, dz, dx, dy, dz));
        float s = (pj.w * inv_rho0) * poly6_strict(kc, distance);
        ax += s * (vj.x - vi.x);
        ay += s * (vj.y - vi.y);
        az += s * (vj.z - vi.z);
    });
    vel_out[i] = make_float4(vi.x + c * ax, vi.y + c * ay, vi.z + c * az, 0.f);
}

// ------------------------------------------------------------------------------------
// slab decomposition (no reference counterpart).  A rank owns the cells whose
// x index lies in [x_lo, x_hi).  Records are 12 floats: pos.xyz, mass, vel.xyz, id (bits),
// prd.xyz, 0.  Ghost particles carry bit 31 of their id.  Compaction order is arbitrary
// (device atomics): the counting sort re-establishes the canonical (cell, id) order.

constexpr uint32_t GHOST_BIT = 0x80000000u;

struct SlabCounters {
    uint32_t exported;
};

__device__ __forceinline__ void write_record(float* rec, const float4& p, const float4& v,
                                             uint32_t id, const float4& q) {
    float4* r = reinterpret_cast<float4*>(rec);
    r[0] = p;
    r[1] = make_float4(v.x, v.y, v.z, __uint_as_float(id));
    r[2] = make_float4(q.x, q.y, q.z, 0.f);
}

// A ghost needs its old position and velocity only for the viscosity stages and its mass only
// when masses differ: otherwise the record shrinks to one float4 (prd.xyz, id bits) -- a third of
// the bytes on the link.
__device__ __forceinline__ void write_ghost(float* buf, uint32_t g, const float4& p, const float4& v,
                                            uint32_t id, const float4& q, int compact) {
    if (compact) reinterpret_cast<float4*>(buf)[g] = make_float4(q.x, q.y, q.z, __uint_as_float(id));
    else write_record(buf + 12 * (size_t)g, p, v, id, q);
}

// counter slots of the split (device memory the caller may forward to its neighbours)
enum { SPLIT_STAY = 0, SPLIT_TOO_FAR = 1, SPLIT_MIG_LEFT = 2, SPLIT_GHOST_LEFT = 3, SPLIT_MIG_RIGHT = 4, SPLIT_GHOST_RIGHT = 5 };

constexpr int SPLIT_ITEMS = 8;                       // particles per thread
constexpr int SPLIT_CHUNK = BLOCK * SPLIT_ITEMS;     // particles per block
constexpr int SPLIT_TILES = SPLIT_ITEMS * (BLOCK / 64);   // wave-sized tiles per block (<= 64)

// after predict, one pass: drop last step's ghosts, keep stayers (compacted into the spare
// buffers), pack leavers as migrant records, pack the stayers of the boundary cell planes as
// ghost records (side 0: cells [lo0, hi0), side 1: [lo1, hi1); a thin slab's particle may be in
// both) and remember their compacted index for the halo refreshes.
// Same-address atomics retire about one per 10 ns, so a block classifies SPLIT_CHUNK particles,
// scans its wave tiles in LDS and reserves its output ranges with ONE atomic per counter.
// PREDICT: the kernel also IS the sub-step's predict + collide pass (k_predict): it reads the old
// pos / vel, and what it writes (and what it classifies on) is the predicted state -- the
// stand-alone k_predict launch and one round trip of vel / prd through HBM disappear.
// Stayers also get their cell key and histogram entry here (as k_predict does for a single ctx),
// so only the particles appended afterwards (migrants in, ghosts) are keyed by k_rekey.
struct PredictArgs {
    float dt, gx, gy, gz;
    ColliderSet cs;
    int keep_vel;   // 0: the stayers' velocities are dead (see k_predict) and are not stored
};

template <bool PREDICT>
__global__ __launch_bounds__(BLOCK) void k_slab_split(
    const float4* __restrict__ pos, const float4* __restrict__ vel, const float4* __restrict__ prd,
    const uint32_t* __restrict__ id, uint32_t n, float spacing, int32_t x_lo, int32_t x_hi,
    int32_t lo0, int32_t hi0, int32_t lo1, int32_t hi1, float4* __restrict__ pos_o,
    float4* __restrict__ vel_o, float4* __restrict__ prd_o, uint32_t* __restrict__ id_o,
    float* __restrict__ mig0, float* __restrict__ mig1, float* __restrict__ gh0,
    float* __restrict__ gh1, uint32_t max_records, uint32_t* __restrict__ blist0,
    uint32_t* __restrict__ blist1, uint32_t* __restrict__ cnt, uint32_t* __restrict__ key_o,
    uint32_t* __restrict__ cell_count, GridConsts G, PredictArgs A, int compact_ghosts) {
    __shared__ uint32_t tile[3][SPLIT_TILES];
    __shared__ uint32_t block_base[3];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t chunk0 = blockIdx.x * SPLIT_CHUNK;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // the state a particle leaves this kernel with: (pos, velocity, predicted position)
    auto state = [&](uint32_t i, float4& p, float4& v, float4& q) {
        p = pos[i];
        v = vel[i];
        if (PREDICT) {
            float vx, vy, vz, qx, qy, qz;
            predict_one(p, v, A.dt, A.gx, A.gy, A.gz, A.cs, vx, vy, vz, qx, qy, qz);
            v = make_float4(vx, vy, vz, 0.f);
            q = make_float4(qx, qy, qz, 0.f);
        } else {
            q = prd[i];
        }
    };
    // per item: bits 0-1 class (0 none, 1 stay, 2 to the left, 3 to the right), bit 2 / 3 = in
    // the left / right boundary planes.  The state of every item stays in registers between the
    // classification and the write-out (one read of pos / vel / id per particle).
    uint32_t flags = 0;
    uint32_t too_far = 0;
    float4 P[SPLIT_ITEMS], V[SPLIT_ITEMS], Q[SPLIT_ITEMS];
    uint32_t ID[SPLIT_ITEMS];
#pragma unroll
    for (int k = 0; k < SPLIT_ITEMS; ++k) {
        const uint32_t i = chunk0 + k * BLOCK + threadIdx.x;
        uint32_t pid = GHOST_BIT;
        if (i < n) pid = id[i];
        ID[k] = pid;
        P[k] = V[k] = Q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t f = 0;
        if (!(pid & GHOST_BIT)) {
            state(i, P[k], V[k], Q[k]);
            const int32_t cx = floor_cell(Q[k].x, spacing);
            if (cx < x_lo) {
                f = 2;
                too_far += (int64_t)cx < (int64_t)x_lo - (hi0 - lo0);
            } else if (cx >= x_hi) {
                f = 3;
                too_far += (int64_t)cx >= (int64_t)x_hi + (hi1 - lo1);
            } else {
                f = 1u | ((cx >= lo0 && cx < hi0) ? 4u : 0u) | ((cx >= lo1 && cx < hi1) ? 8u : 0u);
            }
        }
        flags |= f << (4 * k);
        const unsigned long long ms = __ballot((f & 3u) == 1u), m0 = __ballot(f & 4u), m1 = __ballot(f & 8u);
        if (lane == 0) {
            tile[0][k * (BLOCK / 64) + wave] = (uint32_t)__popcll(ms);
            tile[1][k * (BLOCK / 64) + wave] = (uint32_t)__popcll(m0);
            tile[2][k * (BLOCK / 64) + wave] = (uint32_t)__popcll(m1);
        }
    }
    if (too_far) atomicAdd(&cnt[SPLIT_TOO_FAR], too_far);
    __syncthreads();
    if (wave == 0) {
        const int slot[3] = {SPLIT_STAY, SPLIT_GHOST_LEFT, SPLIT_GHOST_RIGHT};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t v = lane < SPLIT_TILES ? tile[a][lane] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(inc, d, 64);
                if ((int)lane >= d) inc += t;
            }
            const uint32_t total = __shfl(inc, 63, 64);
            if (lane < SPLIT_TILES) tile[a][lane] = inc - v;
            if (lane == 0) block_base[a] = total ? atomicAdd(&cnt[slot[a]], total) : 0u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPLIT_ITEMS; ++k) {
        const uint32_t f = (flags >> (4 * k)) & 15u;
        const bool stay = (f & 3u) == 1u, in0 = f & 4u, in1 = f & 8u;
        const unsigned long long ms = __ballot(stay), m0 = __ballot(in0), m1 = __ballot(in1);
        const float4 p = P[k], v = V[k], q = Q[k];
        const uint32_t pid = ID[k];
        // cell key + histogram of the stayers (every lane takes part in the wave-level run search)
        uint32_t ck = 0xffffffffu;
        if (stay)
            ck = cell_key(G, floor_cell(q.x, G.spacing), floor_cell(q.y, G.spacing), floor_cell(q.z, G.spacing));
        histogram_add(cell_count, ck, stay);
        if (!f) continue;
        const int t = k * (BLOCK / 64) + wave;
        if (stay) {
            const uint32_t s = block_base[0] + tile[0][t] + (uint32_t)__popcll(ms & lt);
            pos_o[s] = p;
            if (!PREDICT || A.keep_vel) vel_o[s] = v;
            prd_o[s] = q;
            id_o[s] = pid;
            key_o[s] = ck;
            if (in0) {
                const uint32_t g = block_base[1] + tile[1][t] + (uint32_t)__popcll(m0 & lt);
                if (g < max_records) {
                    write_ghost(gh0, g, p, v, pid, q, compact_ghosts);
                    blist0[g] = s;
                }
            }
            if (in1) {
                const uint32_t g = block_base[2] + tile[2][t] + (uint32_t)__popcll(m1 & lt);
                if (g < max_records) {
                    write_ghost(gh1, g, p, v, pid, q, compact_ghosts);
                    blist1[g] = s;
                }
            }
        } else {   // leavers are rare: plain atomics
            float* mig = (f & 3u) == 2u ? mig0 : mig1;
            const uint32_t m = atomicAdd(&cnt[(f & 3u) == 2u ? SPLIT_MIG_LEFT : SPLIT_MIG_RIGHT], 1u);
            if (m < max_records) write_record(mig + 12 * (size_t)m, p, v, pid, q);
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_slab_append(const float* __restrict__ rec, uint32_t count,
                                                       uint32_t base, uint32_t ghost_flag,
                                                       float4* __restrict__ pos,
                                                       float4* __restrict__ vel,
                                                       float4* __restrict__ prd,
                                                       uint32_t* __restrict__ id) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const float4* r = reinterpret_cast<const float4*>(rec + 12 * (size_t)i);
    float4 p = r[0], v = r[1], q = r[2];
    uint32_t pid = (__float_as_uint(v.w) & ~GHOST_BIT) | ghost_flag;
    pos[base + i] = p;
    vel[base + i] = make_float4(v.x, v.y, v.z, 0.f);
    prd[base + i] = make_float4(q.x, q.y, q.z, 0.f);
    id[base + i] = pid;
}

// compact ghost records (write_ghost): the old position is the predicted one, the velocity zero --
// neither is read for a ghost without the viscosity stages
__global__ __launch_bounds__(BLOCK) void k_slab_append_compact(const float4* __restrict__ rec, uint32_t count,
                                                               uint32_t base, float mass,
                                                               float4* __restrict__ pos,
                                                               float4* __restrict__ vel,
                                                               float4* __restrict__ prd,
                                                               uint32_t* __restrict__ id) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const float4 r = rec[i];
    pos[base + i] = make_float4(r.x, r.y, r.z, mass);
    vel[base + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    prd[base + i] = make_float4(r.x, r.y, r.z, 0.f);
    id[base + i] = (__float_as_uint(r.w) & ~GHOST_BIT) | GHOST_BIT;
}

// migrants that arrived from one side join that side's boundary list: blist[at + e] = base + e
__global__ __launch_bounds__(BLOCK) void k_slab_list_range(uint32_t* __restrict__ blist, uint32_t at,
                                                           uint32_t base, uint32_t count) {
    uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
    if (e < count) blist[at + e] = base + e;
}

// halo refresh, both sides in one launch: boundary list (pre-sort indices) -> sorted slots ->
// float4 payload
__global__ __launch_bounds__(BLOCK) void k_slab_pack_halo(
    const float4* __restrict__ src, const uint32_t* __restrict__ blist0,
    const uint32_t* __restrict__ blist1, const uint32_t* __restrict__ slot_of, uint32_t count0,
    uint32_t count1, float4* __restrict__ out0, float4* __restrict__ out1) {
    uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
    if (e < count0) {
        out0[e] = src[slot_of[blist0[e]]];
    } else if (e < count0 + count1) {
        e -= count0;
        out1[e] = src[slot_of[blist1[e]]];
    }
}

__global__ __launch_bounds__(BLOCK) void k_slab_unpack_halo(
    float4* __restrict__ dst, const uint32_t* __restrict__ slot_of, uint32_t base0, uint32_t count0,
    uint32_t base1, uint32_t count1, const float4* __restrict__ in0, const float4* __restrict__ in1,
    int keep_w) {
    uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t slot;
    float4 v;
    if (e < count0) {
        slot = slot_of[base0 + e];
        v = in0[e];
    } else if (e < count0 + count1) {
        e -= count0;
        slot = slot_of[base1 + e];
        v = in1[e];
    } else {
        return;
    }
    if (keep_w) v.w = dst[slot].w;
    dst[slot] = v;
}

__global__ __launch_bounds__(BLOCK) void k_slab_export(const float4* __restrict__ pos,
                                                       const float4* __restrict__ vel,
                                                       const float4* __restrict__ prd,
                                                       const uint32_t* __restrict__ id, uint32_t n,
                                                       float* __restrict__ rec, uint32_t max_records,
                                                       uint32_t* __restrict__ counter) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t pid = id[i];
    if (pid & GHOST_BIT) return;
    uint32_t s = atomicAdd(counter, 1u);
    if (s < max_records) write_record(rec + 12 * (size_t)s, pos[i], vel[i], pid, prd[i]);
}

// ------------------------------------------------------------------------------------
// diagnostics

__global__ __launch_bounds__(BLOCK) void k_cells_by_id(const float4* __restrict__ prd,
                                                       const uint32_t* __restrict__ id, uint32_t n,
                                                       GridConsts G, int32_t* __restrict__ xyz,
                                                       uint32_t* __restrict__ key) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 q = prd[i];
    int32_t cx = floor_cell(q.x, G.spacing), cy = floor_cell(q.y, G.spacing),
            cz = floor_cell(q.z, G.spacing);
    uint32_t o = id[i];
    xyz[3 * (size_t)o] = cx;
    xyz[3 * (size_t)o + 1] = cy;
    xyz[3 * (size_t)o + 2] = cz;
    key[o] = cell_key(G, cx, cy, cz);
}

__global__ __launch_bounds__(BLOCK) void k_neighbors_by_id(const uint32_t* __restrict__ nbr,
                                                           const uint32_t* __restrict__ nbr_count,
                                                           const uint32_t* __restrict__ id,
                                                           const uint32_t* __restrict__ target_id,
                                                           uint32_t n, uint32_t cap, uint32_t stride,
                                                           uint32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t o = id[i];
    uint32_t cnt = nbr_count[i];
    counts[o] = cnt;
    for (uint32_t k = 0; k < cnt && k < stride; ++k)
        out[(size_t)o * stride + k] = target_id[nbr[(size_t)k * cap + i]];
}

// exhaustive check of sqrt_rn against the compiler's correctly rounded sqrtf
__global__ __launch_bounds__(BLOCK) void k_selftest_sqrt(uint32_t first, uint32_t last,
                                                         unsigned long long* __restrict__ bad,
                                                         uint32_t* __restrict__ first_bad) {
    unsigned long long local = 0;
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t b = (uint64_t)first + blockIdx.x * BLOCK + threadIdx.x; b <= last; b += stride) {
        float x = __uint_as_float((uint32_t)b);
        float a = sqrt_rn(x), r = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(r)) {
            ++local;
            atomicMin(first_bad, (uint32_t)b);
        }
    }
    if (local) atomicAdd(bad, local);
}

struct StatPartial {
    double sum_abs_C, sum_C, max_abs_C, msum, mx, my, mz, max_speed2;
    unsigned long long nbr_total, count;   // count = owned (non-ghost) particles seen
    uint32_t nonfinite, max_nbr;
};

__global__ __launch_bounds__(BLOCK) void k_stats(const float4* __restrict__ pos,
                                                 const float4* __restrict__ vel,
                                                 const float* __restrict__ C,
                                                 const uint32_t* __restrict__ nbr_count,
                                                 const uint32_t* __restrict__ id, uint32_t n,
                                                 StatPartial* __restrict__ partials) {
    __shared__ StatPartial sh[BLOCK / 64];
    StatPartial a;
    memset(&a, 0, sizeof(a));
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        if (id[i] & 0x80000000u) continue;  // slab ghost
        a.count++;
        float4 p = pos[i];
        if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {
            a.nonfinite++;
            continue;
        }
        double c = C[i], ac = fabs(c);
        a.sum_abs_C += ac;
        a.sum_C += c;
        a.max_abs_C = ac > a.max_abs_C ? ac : a.max_abs_C;
        double m = p.w;
        a.msum += m;
        a.mx += m * p.x;
        a.my += m * p.y;
        a.mz += m * p.z;
        float4 v = vel[i];
        double s2 = (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z;
        a.max_speed2 = s2 > a.max_speed2 ? s2 : a.max_speed2;
        uint32_t k = nbr_count[i];
        a.nbr_total += k;
        a.max_nbr = k > a.max_nbr ? k : a.max_nbr;
    }
    // wave reduction with shuffles, then LDS across waves
    for (int off = 32; off > 0; off >>= 1) {
        a.sum_abs_C += __shfl_down(a.sum_abs_C, off, 64);
        a.sum_C += __shfl_down(a.sum_C, off, 64);
        double m1 = __shfl_down(a.max_abs_C, off, 64);
        a.max_abs_C = m1 > a.max_abs_C ? m1 : a.max_abs_C;
        a.msum += __shfl_down(a.msum, off, 64);
        a.mx += __shfl_down(a.mx, off, 64);
        a.my += __shfl_down(a.my, off, 64);
        a.mz += __shfl_down(a.mz, off, 64);
        double s1 = __shfl_down(a.max_speed2, off, 64);
        a.max_speed2 = s1 > a.max_speed2 ? s1 : a.max_speed2;
        a.nbr_total += __shfl_down(a.nbr_total, off, 64);
        a.count += __shfl_down(a.count, off, 64);
        a.nonfinite += __shfl_down(a.nonfinite, off, 64);
        uint32_t k1 = __shfl_down(a.max_nbr, off, 64);
        a.max_nbr = k1 > a.max_nbr ? k1 : a.max_nbr;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        StatPartial t = sh[0];
        for (int w = 1; w < BLOCK / 64; ++w) {
            const StatPartial& b = sh[w];
            t.sum_abs_C += b.sum_abs_C;
            t.sum_C += b.sum_C;
            t.max_abs_C = b.max_abs_C > t.max_abs_C ? b.max_abs_C : t.max_abs_C;
            t.msum += b.msum;
            t.mx += b.mx;
            t.my += b.my;
            t.mz += b.mz;
            t.max_speed2 = b.max_speed2 > t.max_speed2 ? b.max_speed2 : t.max_speed2;
            t.nbr_total += b.nbr_total;
            t.count += b.count;
            t.nonfinite += b.nonfinite;
            t.max_nbr = b.max_nbr > t.max_nbr ? b.max_nbr : t.max_nbr;
        }
        partials[blockIdx.x] = t;
    }
}

}  // namespace

// ======================================================================================
// host side

struct pbf_ctx {
    pbf_params P;
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n = 0, cap = 0, K = 0, ncell = 0, ntiles = 0;
    int cur = 0;  // which copy of pos/vel/id is live
    float4* pos[2] = {nullptr, nullptr};
    float4* vel[2] = {nullptr, nullptr};
    float4* prd[3] = {nullptr, nullptr, nullptr};
    int prd_cur = 0;  // live predicted/new positions
    uint32_t* id[2] = {nullptr, nullptr};
    uint32_t *key = nullptr, *tmp_src = nullptr, *tmp_id = nullptr;
    uint32_t *cell_count = nullptr, *cell_start = nullptr, *cursor = nullptr, *tile_sums = nullptr;
    uint32_t *nbr = nullptr, *nbr_count = nullptr;
    float *dens = nullptr, *C = nullptr, *rho0 = nullptr;
    float4* omega = nullptr;
    void* staging = nullptr;  // cap * 16 B scratch for up/downloads
    size_t staging_bytes = 0;
    DevFlags* flags = nullptr;
    StatPartial* stat_partials = nullptr;
    ColliderSet colliders;
    bool first_frame = true;
    // slab decomposition state
    bool slab = false;
    int32_t x_lo = INT32_MIN, x_hi = INT32_MAX, ghost_layers = 1;
    uint32_t* slot_of = nullptr;          // pre-sort index -> sorted slot of the last sort
    uint32_t* blist[2] = {nullptr, nullptr};  // boundary-plane particles (pre-sort indices)
    uint32_t bcount[2] = {0, 0};
    uint32_t ghost_base[2] = {0, 0}, ghost_count[2] = {0, 0};
    bool split_pending = false, in_exchange = false;
    const float* mig_out[2] = {nullptr, nullptr};   // this step's migrant records (caller's buffers)
    uint32_t mig_out_count[2] = {0, 0}, split_max = 0;
    SlabCounters* slab_cnt = nullptr;
    // boundary particles (static set; see k_boundary_*)
    uint32_t nb = 0, Kb = 0;
    float4 *bpos = nullptr, *bprd = nullptr;
    float *bmass = nullptr, *bdens = nullptr, *b_rho_s = nullptr, *b_grad_s = nullptr;
    uint32_t *bid = nullptr, *b_cell_start = nullptr;
    uint32_t *bb = nullptr, *bb_count = nullptr, *bf = nullptr, *bf_count = nullptr;
    uint32_t *fb = nullptr, *fb_count = nullptr;
    bool b_static_valid = false;
    // distance constraints (level-ordered SoA on the device; see k_constraints_*)
    uint32_t ncons = 0, ncons_levels = 0;
    int32_t cons_mode = 0;
    uint32_t *cons_p0 = nullptr, *cons_p1 = nullptr, *cons_level_start = nullptr, *slot_by_id = nullptr;
    float *cons_rest = nullptr, *cons_stiff = nullptr, *cons_compl = nullptr, *cons_lambda = nullptr;
    std::vector<uint32_t> cons_order;                // level-ordered position -> list index
    struct ConsBatch { uint32_t first_level, n_levels, first, count; bool one_block; };
    std::vector<ConsBatch> cons_batches;
    uint32_t cons_max_particle = 0;
    // native slab step (pbf_slab_attach / pbf_slab_step): exchange buffers owned by the ctx
    pbf_comm* link_comm = nullptr;
    int32_t link_left = -1, link_right = -1;
    uint32_t link_max = 0;
    float *x_mig[2] = {nullptr, nullptr}, *x_gh[2] = {nullptr, nullptr};      // out: records
    float *x_rmig[2] = {nullptr, nullptr}, *x_rgh[2] = {nullptr, nullptr};    // in: records
    float4 *x_hsend[2] = {nullptr, nullptr}, *x_hrecv[2] = {nullptr, nullptr};
    uint32_t *x_counts = nullptr;       // device: 8 own counters + 4 received
    uint32_t *x_host_counts = nullptr;  // pinned host mirror (12)
    bool owns_stream = true;
    bool uniform = false;      // every particle has uniform_mass (no per-neighbour mass gather)
    float uniform_mass = 0.f;
    bool keyed = false;  // histogram of the live prd already accumulated (k_predict / k_slab_split) ...
    uint32_t keyed_count = 0;  // ... for the first keyed_count particles (pre-sort order)
    bool grid_valid = false;
    int32_t iterations = 1;
    std::string err;
    // profiling
    bool prof = false;
    struct Span {
        int phase;
        hipEvent_t a, b;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> free_events;
    double prof_ms[PH_COUNT] = {0};
    uint64_t prof_calls[PH_COUNT] = {0};
};

namespace {

thread_local std::string g_create_err = "no error";

int fail(pbf_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    else g_create_err = msg;
    return code;
}

#define HIP_CK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), PBF_ERR_HIP,                                                   \
                        std::string(#call) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                            ":" + std::to_string(__LINE__) + ")");                          \
    } while (0)

#define LAUNCH_CK(c) HIP_CK(c, hipGetLastError())

inline uint32_t nblocks(uint32_t n) { return (n + BLOCK - 1) / BLOCK; }

// SPHKernel constants: same expressions, same libm pow, as the reference evaluates per pair
KernConsts make_kern(const pbf_params& P, float dt) {
    KernConsts k;
    memset(&k, 0, sizeof(k));
    const double h = (double)P.effective_radius;
    const double h9 = pow(h, 9);
    const double h6 = pow(h, 6);
    k.h = P.effective_radius;
    k.hd = h;
    k.h2 = h * h;
    k.hh = P.effective_radius * P.effective_radius;
    k.w_coef = (315.0 / (64.0 * REF_PI * h9));
    k.g_coef = (float)(-945.0 / (32.0 * REF_PI * h9));
    k.spiky_coef = (-45.0 / (REF_PI * h6));
    k.scorr_k = -P.scorr_k;
    k.w_zero = (float)(k.w_coef * (k.h2 * k.h2 * k.h2));
    {
        float dq = P.scorr_dq * P.effective_radius;
        double d = (double)dq, q = k.h2 - d * d;
        k.w_dq = (dq >= 0 && dq <= k.h) ? (float)(k.w_coef * (q * q * q)) : 0.f;
    }
    k.inv_w_dq = k.w_dq != 0.f ? 1.0 / (double)k.w_dq : 0.0;
    k.dt_pow = 1.0;
    for (int t = 0; t < P.scorr_dt_power; ++t) k.dt_pow = k.dt_pow * (double)dt;
    k.epsilon = P.epsilon;
    return k;
}

GridConsts make_grid(const pbf_params& P) {
    GridConsts g;
    g.gx = P.grid_dim[0];
    g.gy = P.grid_dim[1];
    g.gz = P.grid_dim[2];
    g.ox = P.grid_origin_cell[0];
    g.oy = P.grid_origin_cell[1];
    g.oz = P.grid_origin_cell[2];
    g.spacing = P.grid_spacing;
    return g;
}

// ---- profiling spans (HIP events on the ctx stream)
hipEvent_t get_event(pbf_ctx* c) {
    if (!c->free_events.empty()) {
        hipEvent_t e = c->free_events.back();
        c->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void fold_spans(pbf_ctx* c) {
    if (c->spans.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->prof_ms[s.phase] += ms;
            c->prof_calls[s.phase] += 1;
        }
        c->free_events.push_back(s.a);
        c->free_events.push_back(s.b);
    }
    c->spans.clear();
}

struct PhaseScope {
    pbf_ctx* c;
    hipEvent_t a = nullptr;
    int phase;
    PhaseScope(pbf_ctx* ctx, int ph) : c(ctx), phase(ph) {
        if (c->prof) {
            a = get_event(c);
            if (a) (void)hipEventRecord(a, c->stream);
        }
    }
    ~PhaseScope() {
        if (a) {
            hipEvent_t b = get_event(c);
            if (b) {
                (void)hipEventRecord(b, c->stream);
                c->spans.push_back({phase, a, b});
                if (c->spans.size() >= 8192) fold_spans(c);
            } else {
                c->free_events.push_back(a);
            }
        }
    }
};

int validate_params(const pbf_params& P, std::string& why) {
    if (P.abi_version != PBF_ABI_VERSION) { why = "abi_version mismatch"; return PBF_ERR_INVALID; }
    if (P.capacity == 0) { why = "capacity must be > 0"; return PBF_ERR_INVALID; }
    if (P.capacity > (1u << 28)) { why = "capacity must not exceed 2^28 particles per ctx (32-bit gather offsets)"; return PBF_ERR_INVALID; }
    if (P.max_neighbors == 0 || P.max_neighbors > 1024) { why = "max_neighbors out of range"; return PBF_ERR_INVALID; }
    for (int a = 0; a < 3; ++a)
        if (P.grid_dim[a] < 3 || P.grid_dim[a] > 4096) { why = "grid_dim must be in [3, 4096]"; return PBF_ERR_INVALID; }
    uint64_t ncell = (uint64_t)P.grid_dim[0] * P.grid_dim[1] * P.grid_dim[2];
    if (ncell > (uint64_t)MAX_SCAN_BLOCKS * SCAN_TILE) { why = "grid has too many cells (max 268435456)"; return PBF_ERR_INVALID; }
    if (!(P.grid_spacing > 0.f) || !(P.effective_radius > 0.f)) { why = "grid_spacing and effective_radius must be > 0"; return PBF_ERR_INVALID; }
    if (P.effective_radius > P.grid_spacing) { why = "effective_radius must not exceed grid_spacing (27-cell sweep)"; return PBF_ERR_INVALID; }
    if (P.solver_iterations < 0 || P.scorr_dt_power < 0 || P.scorr_dt_power > 8) { why = "bad iteration / scorr_dt_power"; return PBF_ERR_INVALID; }
    if (P.gradient_kernel != PBF_GRAD_POLY6 && P.gradient_kernel != PBF_GRAD_SPIKY) { why = "bad gradient_kernel"; return PBF_ERR_INVALID; }
    if (P.rest_density_mode != PBF_RHO0_FIRST_FRAME_MEAN && P.rest_density_mode != PBF_RHO0_FIXED) { why = "bad rest_density_mode"; return PBF_ERR_INVALID; }
    return PBF_OK;
}

template <typename T>
hipError_t dalloc(T** p, size_t count) {
    return hipMalloc((void**)p, (count ? count : 1) * sizeof(T));
}

int set_device(pbf_ctx* c) {
    HIP_CK(c, hipSetDevice(c->device));
    return PBF_OK;
}

#define ENTER(c)                                       \
    if (!(c)) return PBF_ERR_INVALID;                  \
    do {                                               \
        int rc_ = set_device(c);                       \
        if (rc_) return rc_;                           \
    } while (0)

// sort + neighbour table of the live prd (histogram must already be in cell_count)
// move_vel = false: the velocities are dead (see k_predict) and stay where they are
int sort_and_neighbors(pbf_ctx* c, bool with_neighbors = true, bool move_vel = true) {
    const uint32_t n = c->n;
    GridConsts G = make_grid(c->P);
    {
        PhaseScope ps(c, PH_SORT);
        k_scan_tile_sums<<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums);
        if (c->ntiles <= FOLD_MAX_TILES) {
            k_scan_apply<true><<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums,
                                                                   c->cell_start, c->cursor, c->flags);
        } else {
            k_scan_sums<<<1, BLOCK, 0, c->stream>>>(c->tile_sums, c->ntiles);
            k_scan_apply<false><<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums,
                                                                    c->cell_start, c->cursor, c->flags);
        }
        if (n) {
            const int s = c->cur, d = 1 - c->cur;
            const int ps_ = c->prd_cur, pd = (c->prd_cur + 1) % 3;
            k_scatter<<<nblocks(n), BLOCK, 0, c->stream>>>(c->key, c->id[s], c->cursor, n,
                                                           c->tmp_src, c->tmp_id);
            k_reorder<<<nblocks(n), BLOCK, 0, c->stream>>>(
                c->key, c->cell_start, c->tmp_src, c->tmp_id, n, c->pos[s], c->vel[s],
                c->prd[ps_], c->pos[d], move_vel ? c->vel[d] : nullptr, c->prd[pd], c->id[d],
                c->slab ? c->slot_of : nullptr);
            c->cur = d;
            c->prd_cur = pd;
        }
        LAUNCH_CK(c);
    }
    if (!with_neighbors) {
        c->keyed = false;
        c->grid_valid = false;   // cells are sorted, but there is no neighbour table
        return PBF_OK;
    }
    if (n) {
        PhaseScope ps(c, PH_NEIGHBORS);   // the device flags were cleared by k_scan_apply
        k_neighbors<true><<<nblocks(n), BLOCK, 0, c->stream>>>(
            c->prd[c->prd_cur], c->prd[c->prd_cur], c->cell_start, n, c->cap, c->K, G,
            c->P.effective_radius * c->P.effective_radius, c->nbr, c->nbr_count, c->flags);
        if (c->nb) {
            // fluid -> boundary and boundary -> fluid rows, frozen for the sub-step like the fluid's
            const float hh = c->P.effective_radius * c->P.effective_radius;
            k_neighbors<false><<<nblocks(n), BLOCK, 0, c->stream>>>(
                c->prd[c->prd_cur], c->bpos, c->b_cell_start, n, c->cap, c->Kb, G, hh, c->fb,
                c->fb_count, c->flags);
            k_neighbors<false><<<nblocks(c->nb), BLOCK, 0, c->stream>>>(
                c->bpos, c->prd[c->prd_cur], c->cell_start, c->nb, c->nb, c->Kb, G, hh, c->bf,
                c->bf_count, c->flags);
        }
        LAUNCH_CK(c);
    } else if (c->nb) {
        HIP_CK(c, hipMemsetAsync(c->bf_count, 0, (size_t)c->nb * 4, c->stream));
    }
    c->keyed = false;
    c->grid_valid = true;
    return PBF_OK;
}

int download3(pbf_ctx* c, const float4* src, float* host) {
    if (!host || !c->n) return PBF_OK;
    float* st = (float*)c->staging;
    k_unpack3<<<nblocks(c->n), BLOCK, 0, c->stream>>>(src, c->id[c->cur], c->n, st);
    LAUNCH_CK(c);
    HIP_CK(c, hipMemcpyAsync(host, st, (size_t)c->n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    return PBF_OK;
}

void free_link(pbf_ctx* c) {
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(c->x_mig[k]); (void)hipFree(c->x_gh[k]); (void)hipFree(c->x_rmig[k]);
        (void)hipFree(c->x_rgh[k]); (void)hipFree(c->x_hsend[k]); (void)hipFree(c->x_hrecv[k]);
        c->x_mig[k] = c->x_gh[k] = c->x_rmig[k] = c->x_rgh[k] = nullptr;
        c->x_hsend[k] = c->x_hrecv[k] = nullptr;
    }
    (void)hipFree(c->x_counts);
    if (c->x_host_counts) (void)hipHostFree(c->x_host_counts);
    c->x_counts = c->x_host_counts = nullptr;
    c->link_comm = nullptr;
    c->link_max = 0;
}

void free_constraints(pbf_ctx* c) {
    (void)hipFree(c->cons_p0); (void)hipFree(c->cons_p1); (void)hipFree(c->cons_level_start);
    (void)hipFree(c->slot_by_id); (void)hipFree(c->cons_rest); (void)hipFree(c->cons_stiff);
    (void)hipFree(c->cons_compl); (void)hipFree(c->cons_lambda);
    c->cons_p0 = c->cons_p1 = c->cons_level_start = c->slot_by_id = nullptr;
    c->cons_rest = c->cons_stiff = c->cons_compl = c->cons_lambda = nullptr;
    c->ncons = c->ncons_levels = 0;
    c->cons_order.clear();
    c->cons_batches.clear();
}

void free_boundary(pbf_ctx* c) {
    (void)hipFree(c->bpos); (void)hipFree(c->bprd); (void)hipFree(c->bmass); (void)hipFree(c->bdens);
    (void)hipFree(c->b_rho_s); (void)hipFree(c->b_grad_s); (void)hipFree(c->bid);
    (void)hipFree(c->b_cell_start); (void)hipFree(c->bb); (void)hipFree(c->bb_count);
    (void)hipFree(c->bf); (void)hipFree(c->bf_count); (void)hipFree(c->fb); (void)hipFree(c->fb_count);
    c->bpos = c->bprd = nullptr;
    c->bmass = c->bdens = c->b_rho_s = c->b_grad_s = nullptr;
    c->bid = c->b_cell_start = c->bb = c->bb_count = c->bf = c->bf_count = c->fb = c->fb_count = nullptr;
    c->nb = 0;
    c->Kb = 0;
    c->b_static_valid = false;
}

int check_flags(pbf_ctx* c) {
    DevFlags f;
    HIP_CK(c, hipMemcpyAsync(&f, c->flags, sizeof(f), hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    if (f.neighbor_overflow)
        return fail(c, PBF_ERR_NEIGHBOR_OVERFLOW,
                    std::to_string(f.neighbor_overflow) + " particles exceed max_neighbors=" +
                        std::to_string(c->K) + " (largest count " + std::to_string(f.max_count) +
                        "); raise pbf_params.max_neighbors");
    return PBF_OK;
}

}  // namespace

extern "C" {

int pbf_default_params(pbf_params* p) {
    if (!p) return PBF_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    p->abi_version = PBF_ABI_VERSION;
    p->capacity = 0;
    p->max_neighbors = 64;
    p->grid_dim[0] = p->grid_dim[1] = p->grid_dim[2] = 64;  // Simulation.cpp:46
    p->grid_spacing = 1.f;                                  // NeighborSearch.cpp:17
    p->effective_radius = 1.f;                              // Simulation.cpp:88
    p->gravity[0] = 0.f;
    p->gravity[1] = -9.8f;  // Simulation.cpp:16
    p->gravity[2] = 0.f;
    p->epsilon = 1.0e-6f;    // Simulation.cpp:644
    p->restitution = 0.99f;  // Particle.cpp:333
    p->scorr_k = 0.1;        // Simulation.cpp:698
    p->scorr_dq = 0.3f;      // Simulation.cpp:700
    p->scorr_dt_power = 1;   // Simulation.cpp:702
    p->rest_density_mode = PBF_RHO0_FIRST_FRAME_MEAN;
    p->rest_density = 0.8f;    // Simulation.cpp:18
    p->solver_iterations = 1;  // ConstraintSolver.cpp:4
    p->gradient_kernel = PBF_GRAD_POLY6;
    return PBF_OK;
}

const char* pbf_last_error(const pbf_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int pbf_create(const pbf_params* params, int device, pbf_ctx** out) {
    if (out) *out = nullptr;
    if (!params || !out) return fail(nullptr, PBF_ERR_INVALID, "null argument");
    std::string why;
    int rc = validate_params(*params, why);
    if (rc) return fail(nullptr, rc, why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PBF_ERR_NO_DEVICE, "no HIP device visible: the PBF path has no CPU fallback");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) return fail(nullptr, PBF_ERR_INVALID, "device index out of range");
    pbf_ctx* c = new (std::nothrow) pbf_ctx();
    if (!c) return fail(nullptr, PBF_ERR_INVALID, "out of host memory");
    c->P = *params;
    c->device = device;
    c->cap = params->capacity;
    c->K = params->max_neighbors;
    c->ncell = (uint32_t)params->grid_dim[0] * params->grid_dim[1] * params->grid_dim[2];
    c->ntiles = (c->ncell + SCAN_TILE - 1) / SCAN_TILE;
    c->iterations = params->solver_iterations;
    c->err = "no error";
    memset(&c->colliders, 0, sizeof(c->colliders));
    c->colliders.restitution = params->restitution;
#define CREATE_CK(call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);                \
            pbf_destroy(c);                                                                    \
            return fail(nullptr, PBF_ERR_HIP, m_);                                             \
        }                                                                                      \
    } while (0)
    CREATE_CK(hipSetDevice(device));
    CREATE_CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t cap = c->cap;
    for (int b = 0; b < 2; ++b) {
        CREATE_CK(dalloc(&c->pos[b], cap));
        CREATE_CK(dalloc(&c->vel[b], cap));
        CREATE_CK(dalloc(&c->id[b], cap));
    }
    for (int b = 0; b < 3; ++b) CREATE_CK(dalloc(&c->prd[b], cap));
    CREATE_CK(dalloc(&c->key, cap));
    CREATE_CK(dalloc(&c->tmp_src, cap));
    CREATE_CK(dalloc(&c->tmp_id, cap));
    CREATE_CK(dalloc(&c->cell_count, (size_t)c->ncell));
    CREATE_CK(dalloc(&c->cell_start, (size_t)c->ncell + 1));
    CREATE_CK(dalloc(&c->cursor, (size_t)c->ncell));
    CREATE_CK(dalloc(&c->tile_sums, (size_t)c->ntiles));
    CREATE_CK(dalloc(&c->nbr, cap * (size_t)c->K));
    CREATE_CK(dalloc(&c->nbr_count, cap));
    CREATE_CK(dalloc(&c->dens, cap));
    CREATE_CK(dalloc(&c->C, cap));
    CREATE_CK(dalloc(&c->rho0, 1));
    CREATE_CK(dalloc(&c->omega, cap));
    c->staging_bytes = cap * 16;
    CREATE_CK(hipMalloc(&c->staging, c->staging_bytes ? c->staging_bytes : 16));
    CREATE_CK(dalloc(&c->flags, 1));
    CREATE_CK(dalloc(&c->stat_partials, 1024));
    CREATE_CK(dalloc(&c->slot_of, cap));
    CREATE_CK(dalloc(&c->blist[0], cap));
    CREATE_CK(dalloc(&c->blist[1], cap));
    CREATE_CK(dalloc(&c->slab_cnt, 1));
    CREATE_CK(hipMemsetAsync(c->cell_count, 0, (size_t)c->ncell * 4, c->stream));
    CREATE_CK(hipMemsetAsync(c->flags, 0, sizeof(DevFlags), c->stream));
    CREATE_CK(hipMemsetAsync(c->nbr_count, 0, cap * 4, c->stream));
    CREATE_CK(hipMemsetAsync(c->dens, 0, cap * 4, c->stream));
    CREATE_CK(hipMemsetAsync(c->C, 0, cap * 4, c->stream));
    CREATE_CK(hipMemcpyAsync(c->rho0, &c->P.rest_density, 4, hipMemcpyHostToDevice, c->stream));
    CREATE_CK(hipStreamSynchronize(c->stream));
#undef CREATE_CK
    *out = c;
    return PBF_OK;
}

int pbf_destroy(pbf_ctx* c) {
    if (!c) return PBF_ERR_INVALID;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(c->pos[b]);
        (void)hipFree(c->vel[b]);
        (void)hipFree(c->id[b]);
    }
    for (int b = 0; b < 3; ++b) (void)hipFree(c->prd[b]);
    (void)hipFree(c->key);
    (void)hipFree(c->tmp_src);
    (void)hipFree(c->tmp_id);
    (void)hipFree(c->cell_count);
    (void)hipFree(c->cell_start);
    (void)hipFree(c->cursor);
    (void)hipFree(c->tile_sums);
    (void)hipFree(c->nbr);
    (void)hipFree(c->nbr_count);
    (void)hipFree(c->dens);
    (void)hipFree(c->C);
    (void)hipFree(c->rho0);
    (void)hipFree(c->omega);
    (void)hipFree(c->staging);
    (void)hipFree(c->flags);
    (void)hipFree(c->stat_partials);
    (void)hipFree(c->slot_of);
    (void)hipFree(c->blist[0]);
    (void)hipFree(c->blist[1]);
    (void)hipFree(c->slab_cnt);
    free_boundary(c);
    free_constraints(c);
    free_link(c);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PBF_OK;
}

static int upload_common(pbf_ctx* c, const float* d_pos, const float* d_vel, const float* d_mass,
                         float uniform_mass, uint32_t n) {
    c->cur = 0;
    c->prd_cur = 0;
    if (n) {
        k_pack<<<nblocks(n), BLOCK, 0, c->stream>>>(d_pos, d_vel, d_mass, uniform_mass, n, c->pos[0],
                                                    c->vel[0], c->prd[0], c->id[0]);
        LAUNCH_CK(c);
    }
    HIP_CK(c, hipMemcpyAsync(c->rho0, &c->P.rest_density, 4, hipMemcpyHostToDevice, c->stream));
    if (c->keyed) HIP_CK(c, hipMemsetAsync(c->cell_count, 0, (size_t)c->ncell * 4, c->stream));
    HIP_CK(c, hipMemsetAsync(c->nbr_count, 0, (size_t)c->cap * 4, c->stream));
    HIP_CK(c, hipMemsetAsync(c->flags, 0, sizeof(DevFlags), c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    c->n = n;
    c->uniform = (d_mass == nullptr);
    c->uniform_mass = uniform_mass;
    c->first_frame = true;
    c->keyed = false;
    c->grid_valid = false;
    c->b_static_valid = false;
    return PBF_OK;
}

int pbf_upload(pbf_ctx* c, const float* pos, const float* vel, const float* mass,
               float uniform_mass, uint32_t n) {
    ENTER(c);
    if (n > c->cap) return fail(c, PBF_ERR_CAPACITY, "n exceeds ctx capacity");
    if (n && !pos) return fail(c, PBF_ERR_INVALID, "pos is null");
    if (!mass && !(uniform_mass > 0.f)) return fail(c, PBF_ERR_INVALID, "uniform_mass must be > 0");
    float *dp = nullptr, *dv = nullptr, *dm = nullptr;
    if (n) {
        // stage the raw triples in spare device arrays that k_pack does not write:
        // pos -> prd[1] (16 B/particle >= 12), vel -> prd[2], mass -> staging
        dp = (float*)c->prd[1];
        HIP_CK(c, hipMemcpyAsync(dp, pos, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
        if (vel) {
            dv = (float*)c->prd[2];
            HIP_CK(c, hipMemcpyAsync(dv, vel, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
        }
        if (mass) {
            dm = (float*)c->staging;
            HIP_CK(c, hipMemcpyAsync(dm, mass, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        }
    }
    return upload_common(c, dp, dv, dm, uniform_mass, n);
}

int pbf_upload_device(pbf_ctx* c, const void* d_pos, const void* d_vel, const void* d_mass,
                      float uniform_mass, uint32_t n) {
    ENTER(c);
    if (n > c->cap) return fail(c, PBF_ERR_CAPACITY, "n exceeds ctx capacity");
    if (n && !d_pos) return fail(c, PBF_ERR_INVALID, "d_pos is null");
    if (!d_mass && !(uniform_mass > 0.f)) return fail(c, PBF_ERR_INVALID, "uniform_mass must be > 0");
    return upload_common(c, (const float*)d_pos, (const float*)d_vel, (const float*)d_mass,
                         uniform_mass, n);
}

int pbf_set_colliders(pbf_ctx* c, const pbf_collider* cols, uint32_t count) {
    if (!c) return PBF_ERR_INVALID;
    if (count > PBF_MAX_COLLIDERS) return fail(c, PBF_ERR_INVALID, "too many colliders");
    if (count && !cols) return fail(c, PBF_ERR_INVALID, "colliders is null");
    for (uint32_t i = 0; i < count; ++i)
        if (cols[i].type != PBF_COLLIDER_PLANE && cols[i].type != PBF_COLLIDER_AABB)
            return fail(c, PBF_ERR_INVALID,
                        "unsupported collider type (only PLANE and AABB have a defined response "
                        "in the reference, Particle.cpp:248-338)");
    for (uint32_t i = 0; i < count; ++i) c->colliders.c[i] = cols[i];
    c->colliders.count = count;
    return PBF_OK;
}

int pbf_set_solver_iterations(pbf_ctx* c, int32_t it) {
    if (!c) return PBF_ERR_INVALID;
    if (it < 0) return fail(c, PBF_ERR_INVALID, "iterations must be >= 0");
    c->iterations = it;
    return PBF_OK;
}

int pbf_set_gravity(pbf_ctx* c, float gx, float gy, float gz) {
    if (!c) return PBF_ERR_INVALID;
    c->P.gravity[0] = gx;
    c->P.gravity[1] = gy;
    c->P.gravity[2] = gz;
    return PBF_OK;
}

int pbf_set_boundary(pbf_ctx* c, const float* pos, const float* mass, float uniform_mass,
                     uint32_t nb, uint32_t max_boundary_neighbors) {
    ENTER(c);
    if (nb && !pos) return fail(c, PBF_ERR_INVALID, "pos is null");
    if (nb && !mass && !(uniform_mass > 0.f)) return fail(c, PBF_ERR_INVALID, "uniform_mass must be > 0");
    if (max_boundary_neighbors > 1024) return fail(c, PBF_ERR_INVALID, "max_boundary_neighbors out of range");
    HIP_CK(c, hipStreamSynchronize(c->stream));
    free_boundary(c);
    c->grid_valid = false;  // the fluid<->boundary rows of the last search are gone
    if (!nb) return PBF_OK;
    const uint32_t Kb = max_boundary_neighbors ? max_boundary_neighbors : c->K;
    const size_t cap = c->cap;
    float *d_pos3 = nullptr, *d_mass = nullptr;
    float4* raw = nullptr;
    uint32_t *key = nullptr, *tsrc = nullptr, *tid = nullptr, *ident = nullptr;
    auto drop_tmp = [&]() {
        (void)hipFree(d_pos3); (void)hipFree(d_mass); (void)hipFree(raw); (void)hipFree(key);
        (void)hipFree(tsrc); (void)hipFree(tid); (void)hipFree(ident);
    };
#define B_CK(call)                                                                   \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            (void)hipStreamSynchronize(c->stream);                                   \
            drop_tmp();                                                              \
            free_boundary(c);                                                        \
            return fail(c, PBF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                            \
    } while (0)
    B_CK(dalloc(&c->bpos, (size_t)nb));
    B_CK(dalloc(&c->bprd, (size_t)nb));
    B_CK(dalloc(&c->bmass, (size_t)nb));
    B_CK(dalloc(&c->bdens, (size_t)nb));
    B_CK(dalloc(&c->b_rho_s, (size_t)nb));
    B_CK(dalloc(&c->b_grad_s, (size_t)nb));
    B_CK(dalloc(&c->bid, (size_t)nb));
    B_CK(dalloc(&c->b_cell_start, (size_t)c->ncell + 1));
    B_CK(dalloc(&c->bb, (size_t)nb * Kb));
    B_CK(dalloc(&c->bb_count, (size_t)nb));
    B_CK(dalloc(&c->bf, (size_t)nb * Kb));
    B_CK(dalloc(&c->bf_count, (size_t)nb));
    B_CK(dalloc(&c->fb, cap * Kb));
    B_CK(dalloc(&c->fb_count, cap));
    B_CK(dalloc(&d_pos3, (size_t)nb * 3));
    if (mass) B_CK(dalloc(&d_mass, (size_t)nb));
    B_CK(dalloc(&raw, (size_t)nb));
    B_CK(dalloc(&key, (size_t)nb));
    B_CK(dalloc(&tsrc, (size_t)nb));
    B_CK(dalloc(&tid, (size_t)nb));
    B_CK(dalloc(&ident, (size_t)nb));
    B_CK(hipMemcpyAsync(d_pos3, pos, (size_t)nb * 12, hipMemcpyHostToDevice, c->stream));
    if (mass) B_CK(hipMemcpyAsync(d_mass, mass, (size_t)nb * 4, hipMemcpyHostToDevice, c->stream));
    B_CK(hipMemsetAsync(c->bdens, 0, (size_t)nb * 4, c->stream));
    B_CK(hipMemsetAsync(c->bf_count, 0, (size_t)nb * 4, c->stream));
    B_CK(hipMemsetAsync(c->fb_count, 0, cap * 4, c->stream));
    // the cell histogram is all zero between sub-steps unless a predict is pending
    if (c->keyed) {
        B_CK(hipMemsetAsync(c->cell_count, 0, (size_t)c->ncell * 4, c->stream));
        c->keyed = false;
    }
    const GridConsts G = make_grid(c->P);
    const uint32_t nbb = nblocks(nb);
    k_boundary_pack<<<nbb, BLOCK, 0, c->stream>>>(d_pos3, d_mass, uniform_mass, nb, raw, ident);
    k_rekey<<<nbb, BLOCK, 0, c->stream>>>(raw, key, c->cell_count, nb, G);
    k_scan_tile_sums<<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums);
    k_scan_sums<<<1, BLOCK, 0, c->stream>>>(c->tile_sums, c->ntiles);
    k_scan_apply<false><<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums,
                                                            c->b_cell_start, c->cursor, nullptr);
    k_scatter<<<nbb, BLOCK, 0, c->stream>>>(key, ident, c->cursor, nb, tsrc, tid);
    k_boundary_reorder<<<nbb, BLOCK, 0, c->stream>>>(key, c->b_cell_start, tsrc, tid, nb, raw, c->bprd,
                                                     c->bmass, c->bid);
    B_CK(hipGetLastError());
    B_CK(hipMemsetAsync(c->flags, 0, sizeof(DevFlags), c->stream));
    k_neighbors<true><<<nbb, BLOCK, 0, c->stream>>>(c->bprd, c->bprd, c->b_cell_start, nb, nb, Kb, G,
                                                    c->P.effective_radius * c->P.effective_radius,
                                                    c->bb, c->bb_count, c->flags);
    k_boundary_volume<<<nbb, BLOCK, 0, c->stream>>>(c->bprd, c->bmass, c->bb, c->bb_count, nb,
                                                    make_kern(c->P, 0.f), c->bpos);
    B_CK(hipGetLastError());
    B_CK(hipStreamSynchronize(c->stream));
#undef B_CK
    drop_tmp();
    c->nb = nb;
    c->Kb = Kb;
    int rc = check_flags(c);
    if (rc) {
        std::string why = c->err;
        free_boundary(c);
        return fail(c, rc, "boundary-boundary table: " + why);
    }
    return PBF_OK;
}

uint32_t pbf_num_boundary_particles(const pbf_ctx* c) { return c ? c->nb : 0; }

int pbf_get_boundary(pbf_ctx* c, float* volume, float* density, float* lambda) {
    ENTER(c);
    const uint32_t nb = c->nb;
    if (!nb) return PBF_OK;
    float* d = nullptr;
    HIP_CK(c, dalloc(&d, (size_t)nb * 3));
    k_boundary_unpack<<<nblocks(nb), BLOCK, 0, c->stream>>>(c->bpos, c->bdens, c->bprd, c->bid, nb, d,
                                                            d + nb, d + 2 * (size_t)nb);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && volume) e = hipMemcpyAsync(volume, d, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && density) e = hipMemcpyAsync(density, d + nb, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && lambda) e = hipMemcpyAsync(lambda, d + 2 * (size_t)nb, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, PBF_ERR_HIP, std::string("get_boundary: ") + hipGetErrorString(e));
    return PBF_OK;
}

int pbf_get_boundary_neighbors(pbf_ctx* c, uint32_t* fluid_counts, uint32_t* fluid_to_boundary,
                               uint32_t stride_f, uint32_t* boundary_counts,
                               uint32_t* boundary_to_fluid, uint32_t stride_b) {
    ENTER(c);
    if (!c->nb) return fail(c, PBF_ERR_INVALID, "no boundary particles");
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    if (!fluid_counts || !fluid_to_boundary || !boundary_counts || !boundary_to_fluid || !stride_f || !stride_b)
        return fail(c, PBF_ERR_INVALID, "null output");
    const uint32_t n = c->n, nb = c->nb;
    uint32_t *d_fc = nullptr, *d_fl = nullptr, *d_bc = nullptr, *d_bl = nullptr;
    hipError_t e = dalloc(&d_fc, (size_t)n);
    if (e == hipSuccess) e = dalloc(&d_fl, (size_t)n * stride_f);
    if (e == hipSuccess) e = dalloc(&d_bc, (size_t)nb);
    if (e == hipSuccess) e = dalloc(&d_bl, (size_t)nb * stride_b);
    if (e == hipSuccess) e = hipMemsetAsync(d_fl, 0, (size_t)(n ? n : 1) * stride_f * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_bl, 0, (size_t)nb * stride_b * 4, c->stream);
    if (e == hipSuccess) {
        if (n)
            k_neighbors_by_id<<<nblocks(n), BLOCK, 0, c->stream>>>(c->fb, c->fb_count, c->id[c->cur], c->bid, n,
                                                                   c->cap, stride_f, d_fc, d_fl);
        k_neighbors_by_id<<<nblocks(nb), BLOCK, 0, c->stream>>>(c->bf, c->bf_count, c->bid, c->id[c->cur], nb,
                                                                nb, stride_b, d_bc, d_bl);
        e = hipGetLastError();
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(fluid_counts, d_fc, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(fluid_to_boundary, d_fl, (size_t)n * stride_f * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(boundary_counts, d_bc, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(boundary_to_fluid, d_bl, (size_t)nb * stride_b * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_fc); (void)hipFree(d_fl); (void)hipFree(d_bc); (void)hipFree(d_bl);
    if (e != hipSuccess) return fail(c, PBF_ERR_HIP, std::string("get_boundary_neighbors: ") + hipGetErrorString(e));
    return PBF_OK;
}

static int predict_impl(pbf_ctx* c, float dt, bool keep_vel) {
    ENTER(c);
    if (c->keyed) HIP_CK(c, hipMemsetAsync(c->cell_count, 0, (size_t)c->ncell * 4, c->stream));
    if (c->n) {
        PhaseScope ps(c, PH_PREDICT);
        k_predict<<<nblocks(c->n), BLOCK, 0, c->stream>>>(
            c->pos[c->cur], c->vel[c->cur], c->prd[c->prd_cur], c->key, c->slab ? nullptr : c->cell_count, c->n, dt,
            c->P.gravity[0], c->P.gravity[1], c->P.gravity[2], c->colliders, make_grid(c->P), keep_vel ? 1 : 0);
        LAUNCH_CK(c);
    }
    c->keyed = !c->slab;
    c->keyed_count = c->keyed ? c->n : 0;
    c->grid_valid = false;
    return PBF_OK;
}

int pbf_predict(pbf_ctx* c, float dt) { return predict_impl(c, dt, true); }

static int build_grid_impl(pbf_ctx* c, bool move_vel) {
    ENTER(c);
    if (c->split_pending || c->in_exchange)
        return fail(c, PBF_ERR_INVALID, "slab exchange in flight: split_done / append_ghosts come first");
    const uint32_t first = c->keyed ? (c->keyed_count < c->n ? c->keyed_count : c->n) : 0u;
    if (first < c->n) {   // particles not keyed yet (all of them, or the ones appended after a split)
        k_rekey<<<nblocks(c->n - first), BLOCK, 0, c->stream>>>(c->prd[c->prd_cur], c->key, c->cell_count,
                                                                c->n, make_grid(c->P), first);
        LAUNCH_CK(c);
    }
    return sort_and_neighbors(c, true, move_vel);
}

int pbf_build_grid(pbf_ctx* c) { return build_grid_impl(c, true); }

int pbf_compute_rest_density(pbf_ctx* c) {
    ENTER(c);
    if (c->P.rest_density_mode != PBF_RHO0_FIRST_FRAME_MEAN || !c->first_frame) return PBF_OK;
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    if (c->n) {
        PhaseScope ps(c, PH_REST_DENSITY);
        float* by_id = (float*)c->staging;
        k_rest_density_terms<<<nblocks(c->n), BLOCK, 0, c->stream>>>(
            c->pos[c->cur], c->id[c->cur], c->nbr, c->nbr_count, c->n, c->cap,
            make_kern(c->P, 0.f), by_id);
        k_rest_density_mean<<<1, 64, 0, c->stream>>>(by_id, c->n, c->rho0);
        LAUNCH_CK(c);
    }
    c->b_static_valid = false;  // rho0 changed
    c->first_frame = false;
    return PBF_OK;
}

// keys of the first cell of the plane `depth` ghost planes left of the slab and of the first plane
// beyond `depth` ghost planes on the right (an open end: the whole side).  False when the slab's
// ghost planes do not fit the rank-local grid unwrapped (then sweep everything).
static bool slab_window(const pbf_ctx* c, int32_t depth, uint32_t* key_a, uint32_t* key_b) {
    const int64_t gx = c->P.grid_dim[0], plane = (int64_t)c->P.grid_dim[1] * c->P.grid_dim[2];
    const int64_t ox = c->P.grid_origin_cell[0], L = c->ghost_layers;
    int64_t a = 0, b = gx;
    if (c->x_lo != INT32_MIN) {
        if ((int64_t)c->x_lo - L - ox < 0) return false;
        a = (int64_t)c->x_lo - depth - ox;
    }
    if (c->x_hi != INT32_MAX) {
        if ((int64_t)c->x_hi + L - ox > gx) return false;
        b = (int64_t)c->x_hi + depth - ox;
    }
    if (depth > L || a < 0 || b > gx || a >= b) return false;
    *key_a = (uint32_t)(a * plane);
    *key_b = (uint32_t)(b * plane);
    return true;
}

// ghost_depth >= 0 (slab mode): sweep only the owned cell planes plus that many ghost planes per
// side (sweep_index); -1: every particle
// keep_scalars = false: rho_i / C_i of this lambda sweep are not stored (a later sweep of the same
// sub-step overwrites them before anyone can read them)
static int solve_half(pbf_ctx* c, int half, float dt, bool fuse_finalize = false, int32_t ghost_depth = -1,
                      bool keep_scalars = true) {
    const KernConsts kc = make_kern(c->P, dt);
    const uint32_t nb = nblocks(c->n);
    const uint32_t* win_start = nullptr;
    uint32_t win_a = 0, win_b = 0;
    if (ghost_depth >= 0 && c->slab && slab_window(c, ghost_depth, &win_a, &win_b)) win_start = c->cell_start;
    const bool spiky = c->P.gradient_kernel == PBF_GRAD_SPIKY;
    float4* x = c->prd[c->prd_cur];
    if (half == 0) {
        PhaseScope ps(c, PH_LAMBDA);
#define LAUNCH_LAMBDA(U, G, B)                                                                \
    k_lambda<U, G, B><<<nb, BLOCK, 0, c->stream>>>(x, c->pos[c->cur], c->nbr, c->nbr_count, c->n, \
                                                   c->cap, kc, c->rho0, c->uniform_mass,          \
                                                   keep_scalars ? c->dens : nullptr, c->C, c->bpos, \
                                                   c->fb, c->fb_count, win_start, win_a, win_b, c->flags)
#define LAUNCH_LAMBDA_UG(U, G)            \
    do {                                  \
        if (c->nb) LAUNCH_LAMBDA(U, G, true); \
        else LAUNCH_LAMBDA(U, G, false);  \
    } while (0)
        if (c->uniform && !spiky) LAUNCH_LAMBDA_UG(true, PBF_GRAD_POLY6);
        else if (c->uniform) LAUNCH_LAMBDA_UG(true, PBF_GRAD_SPIKY);
        else if (!spiky) LAUNCH_LAMBDA_UG(false, PBF_GRAD_POLY6);
        else LAUNCH_LAMBDA_UG(false, PBF_GRAD_SPIKY);
#undef LAUNCH_LAMBDA_UG
#undef LAUNCH_LAMBDA
        if (c->nb) {
            const uint32_t nbb = nblocks(c->nb);
            if (!c->b_static_valid) {
                if (spiky)
                    k_boundary_static<PBF_GRAD_SPIKY><<<nbb, BLOCK, 0, c->stream>>>(
                        c->bpos, c->bmass, c->bb, c->bb_count, c->nb, kc, c->rho0, c->b_rho_s, c->b_grad_s);
                else
                    k_boundary_static<PBF_GRAD_POLY6><<<nbb, BLOCK, 0, c->stream>>>(
                        c->bpos, c->bmass, c->bb, c->bb_count, c->nb, kc, c->rho0, c->b_rho_s, c->b_grad_s);
                c->b_static_valid = true;
            }
#define LAUNCH_BLAMBDA(U, G)                                                                     \
    k_boundary_lambda<U, G><<<nbb, BLOCK, 0, c->stream>>>(c->bpos, c->bprd, x, c->pos[c->cur], c->bf, \
                                                          c->bf_count, c->nb, kc, c->rho0,       \
                                                          c->uniform_mass, c->b_rho_s, c->b_grad_s, c->bdens)
            if (c->uniform && !spiky) LAUNCH_BLAMBDA(true, PBF_GRAD_POLY6);
            else if (c->uniform) LAUNCH_BLAMBDA(true, PBF_GRAD_SPIKY);
            else if (!spiky) LAUNCH_BLAMBDA(false, PBF_GRAD_POLY6);
            else LAUNCH_BLAMBDA(false, PBF_GRAD_SPIKY);
#undef LAUNCH_BLAMBDA
        }
    } else {
        PhaseScope ps(c, PH_DELTA);
        float3_ cmin = {c->P.container_min[0], c->P.container_min[1], c->P.container_min[2]};
        float3_ cmax = {c->P.container_max[0], c->P.container_max[1], c->P.container_max[2]};
        float4* y = c->prd[(c->prd_cur + 1) % 3];
#define LAUNCH_DELTA(G, B, F)                                                                     \
    k_delta<G, B, F><<<nb, BLOCK, 0, c->stream>>>(x, y, c->nbr, c->nbr_count, c->n, c->cap, kc,   \
                                                  c->rho0, c->P.container_enabled, cmin, cmax,    \
                                                  c->bprd, c->fb, c->fb_count, c->pos[c->cur],    \
                                                  c->vel[c->cur], dt, win_start, win_a, win_b, c->flags)
#define LAUNCH_DELTA_F(G, B)                  \
    do {                                      \
        if (fuse_finalize) LAUNCH_DELTA(G, B, true); \
        else LAUNCH_DELTA(G, B, false);       \
    } while (0)
        if (!spiky && !c->nb) LAUNCH_DELTA_F(PBF_GRAD_POLY6, false);
        else if (!spiky) LAUNCH_DELTA_F(PBF_GRAD_POLY6, true);
        else if (!c->nb) LAUNCH_DELTA_F(PBF_GRAD_SPIKY, false);
        else LAUNCH_DELTA_F(PBF_GRAD_SPIKY, true);
#undef LAUNCH_DELTA_F
#undef LAUNCH_DELTA
        c->prd_cur = (c->prd_cur + 1) % 3;
    }
    LAUNCH_CK(c);
    return PBF_OK;
}

// whether the sub-step's velocity update can ride along with the last delta-p kernel
static bool can_fuse_finalize(const pbf_ctx* c) {
    return c->iterations > 0 && c->n && !c->ncons && c->P.vorticity_eps == 0.f && c->P.xsph_c == 0.f;
}

int pbf_solve(pbf_ctx* c, int32_t iterations, float dt) {
    ENTER(c);
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    if (iterations <= 0) iterations = c->iterations;
    if (!c->n) return PBF_OK;
    for (int32_t it = 0; it < iterations; ++it) {
        int rc = solve_half(c, 0, dt);
        if (rc) return rc;
        if ((rc = solve_half(c, 1, dt))) return rc;
    }
    return PBF_OK;
}

int pbf_solve_lambda(pbf_ctx* c, float dt) {
    ENTER(c);
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    return c->n ? solve_half(c, 0, dt) : PBF_OK;
}

int pbf_solve_delta(pbf_ctx* c, float dt) {
    ENTER(c);
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    return c->n ? solve_half(c, 1, dt) : PBF_OK;
}

static int finalize_stage(pbf_ctx* c, int stage, float dt) {
    const uint32_t nb = nblocks(c->n);
    const KernConsts kc = make_kern(c->P, dt);
    const bool spiky = c->P.gradient_kernel == PBF_GRAD_SPIKY;
    float4* va = c->vel[c->cur];
    float4* vb = c->vel[1 - c->cur];  // spare copy (the next sort overwrites it): Jacobi target
    switch (stage) {
        case 0: {
            PhaseScope ps(c, PH_FINALIZE);
            k_finalize<<<nb, BLOCK, 0, c->stream>>>(c->pos[c->cur], va, c->prd[c->prd_cur], c->n, dt);
            break;
        }
        case 1: {
            PhaseScope ps(c, PH_VISCOSITY);
            if (spiky)
                k_vorticity<PBF_GRAD_SPIKY><<<nb, BLOCK, 0, c->stream>>>(
                    c->pos[c->cur], va, c->nbr, c->nbr_count, c->n, c->cap, kc, c->omega);
            else
                k_vorticity<PBF_GRAD_POLY6><<<nb, BLOCK, 0, c->stream>>>(
                    c->pos[c->cur], va, c->nbr, c->nbr_count, c->n, c->cap, kc, c->omega);
            break;
        }
        case 2: {
            PhaseScope ps(c, PH_VISCOSITY);
            if (spiky)
                k_vorticity_apply<PBF_GRAD_SPIKY><<<nb, BLOCK, 0, c->stream>>>(
                    c->pos[c->cur], va, c->omega, c->nbr, c->nbr_count, c->n, c->cap, kc, c->rho0,
                    dt, c->P.vorticity_eps, vb);
            else
                k_vorticity_apply<PBF_GRAD_POLY6><<<nb, BLOCK, 0, c->stream>>>(
                    c->pos[c->cur], va, c->omega, c->nbr, c->nbr_count, c->n, c->cap, kc, c->rho0,
                    dt, c->P.vorticity_eps, vb);
            c->vel[c->cur] = vb;  // the two velocity copies trade places
            c->vel[1 - c->cur] = va;
            break;
        }
        case 3: {
            PhaseScope ps(c, PH_VISCOSITY);
            k_xsph<<<nb, BLOCK, 0, c->stream>>>(c->pos[c->cur], va, c->nbr, c->nbr_count, c->n,
                                                c->cap, kc, c->rho0, c->P.xsph_c, vb);
            c->vel[c->cur] = vb;
            c->vel[1 - c->cur] = va;
            break;
        }
        default:
            return fail(c, PBF_ERR_INVALID, "bad finalize stage");
    }
    LAUNCH_CK(c);
    return PBF_OK;
}

int pbf_finalize_stage(pbf_ctx* c, int32_t stage, float dt) {
    ENTER(c);
    if (!c->n) return PBF_OK;
    if (stage > 0 && !c->grid_valid) return fail(c, PBF_ERR_INVALID, "viscosity needs the neighbour table");
    return finalize_stage(c, stage, dt);
}

int pbf_finalize(pbf_ctx* c, float dt) {
    ENTER(c);
    if (!c->n) return PBF_OK;
    int rc = finalize_stage(c, 0, dt);
    if (rc) return rc;
    if (c->P.vorticity_eps != 0.f || c->P.xsph_c != 0.f) {
        if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "viscosity needs the neighbour table");
        if (c->P.vorticity_eps != 0.f) {
            if ((rc = finalize_stage(c, 1, dt))) return rc;
            if ((rc = finalize_stage(c, 2, dt))) return rc;
        }
        if (c->P.xsph_c != 0.f && (rc = finalize_stage(c, 3, dt))) return rc;
    }
    return PBF_OK;
}

// ---------------------------------------------------------------- slab decomposition

int pbf_set_stream(pbf_ctx* c, void* stream) {
    ENTER(c);
    HIP_CK(c, hipStreamSynchronize(c->stream));
    fold_spans(c);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)stream;
    c->owns_stream = false;
    return PBF_OK;
}

int pbf_slab_set_bounds(pbf_ctx* c, int32_t x_lo, int32_t x_hi, int32_t ghost_layers) {
    if (!c) return PBF_ERR_INVALID;
    if (x_lo >= x_hi) return fail(c, PBF_ERR_INVALID, "empty slab");
    if (ghost_layers < 1 || ghost_layers > 8) return fail(c, PBF_ERR_INVALID, "ghost_layers must be in [1, 8]");
    if (c->P.rest_density_mode != PBF_RHO0_FIXED)
        return fail(c, PBF_ERR_INVALID, "slab mode needs PBF_RHO0_FIXED (the first-frame mean is a serial global sum)");
    if (c->ncons) return fail(c, PBF_ERR_INVALID, "constraints are not available in slab mode");
    c->slab = true;
    c->x_lo = x_lo;
    c->x_hi = x_hi;
    c->ghost_layers = ghost_layers;
    return PBF_OK;
}

static int slab_split_impl(pbf_ctx* c, void* d_mig_left, void* d_mig_right, void* d_ghost_left,
                           void* d_ghost_right, uint32_t max_records, void* d_counts, bool predict,
                           float dt, bool compact_ghosts = false, bool keep_vel = true) {
    ENTER(c);
    if (!c->slab) return fail(c, PBF_ERR_INVALID, "pbf_slab_set_bounds has not been called");
    if (c->split_pending) return fail(c, PBF_ERR_INVALID, "pbf_slab_split_done has not been called");
    if (!d_mig_left || !d_mig_right || !d_ghost_left || !d_ghost_right || !d_counts || !max_records)
        return fail(c, PBF_ERR_INVALID, "bad argument");
    HIP_CK(c, hipMemsetAsync(d_counts, 0, 6 * sizeof(uint32_t), c->stream));
    if (c->keyed) HIP_CK(c, hipMemsetAsync(c->cell_count, 0, (size_t)c->ncell * 4, c->stream));  // stale histogram
    const int s = c->cur, d = 1 - c->cur;
    const int ps_ = c->prd_cur, pd = (c->prd_cur + 1) % 3;
    if (max_records > c->cap) max_records = c->cap;   // the boundary lists hold cap entries
    if (c->n) {
        // empty plane ranges for open ends
        const bool open0 = c->x_lo == INT32_MIN, open1 = c->x_hi == INT32_MAX;
        const int32_t L = c->ghost_layers;
        const int32_t lo0 = open0 ? 0 : c->x_lo, hi0 = open0 ? 0 : (int32_t)((int64_t)c->x_lo + L);
        const int32_t lo1 = open1 ? 0 : (int32_t)((int64_t)c->x_hi - L), hi1 = open1 ? 0 : c->x_hi;
        PredictArgs A;
        A.dt = dt;
        A.gx = c->P.gravity[0];
        A.gy = c->P.gravity[1];
        A.gz = c->P.gravity[2];
        A.cs = c->colliders;
        A.keep_vel = keep_vel ? 1 : 0;
        const uint32_t grid = (c->n + SPLIT_CHUNK - 1) / SPLIT_CHUNK;
#define LAUNCH_SPLIT(PRED_)                                                                             \
    k_slab_split<PRED_><<<grid, BLOCK, 0, c->stream>>>(                                                 \
        c->pos[s], c->vel[s], c->prd[ps_], c->id[s], c->n, c->P.grid_spacing, c->x_lo, c->x_hi, lo0,  \
        hi0, lo1, hi1, c->pos[d], c->vel[d], c->prd[pd], c->id[d], (float*)d_mig_left,                \
        (float*)d_mig_right, (float*)d_ghost_left, (float*)d_ghost_right, max_records, c->blist[0],   \
        c->blist[1], (uint32_t*)d_counts, c->key, c->cell_count, make_grid(c->P), A,                  \
        compact_ghosts ? 1 : 0)
        if (predict) {
            PhaseScope ps(c, PH_PREDICT);
            LAUNCH_SPLIT(true);
        } else {
            LAUNCH_SPLIT(false);
        }
#undef LAUNCH_SPLIT
        LAUNCH_CK(c);
    }
    c->cur = d;
    c->prd_cur = pd;
    c->keyed = true;       // the stayers are keyed; their number is known at pbf_slab_split_done
    c->keyed_count = 0;
    c->grid_valid = false;
    c->split_pending = true;
    c->split_max = max_records;
    c->mig_out[0] = (const float*)d_mig_left;
    c->mig_out[1] = (const float*)d_mig_right;
    return PBF_OK;
}

int pbf_slab_split(pbf_ctx* c, void* d_mig_left, void* d_mig_right, void* d_ghost_left,
                   void* d_ghost_right, uint32_t max_records, void* d_counts) {
    return slab_split_impl(c, d_mig_left, d_mig_right, d_ghost_left, d_ghost_right, max_records,
                           d_counts, false, 0.f);
}

int pbf_slab_split_done(pbf_ctx* c, const uint32_t counts[6]) {
    ENTER(c);
    if (!c->split_pending) return fail(c, PBF_ERR_INVALID, "no split in flight");
    if (!counts) return fail(c, PBF_ERR_INVALID, "null counts");
    c->split_pending = false;
    c->n = 0;
    c->ghost_count[0] = c->ghost_count[1] = 0;
    c->bcount[0] = c->bcount[1] = 0;
    c->mig_out_count[0] = c->mig_out_count[1] = 0;
    if (counts[SPLIT_STAY] > c->cap) return fail(c, PBF_ERR_INVALID, "stay count exceeds capacity (wrong counts?)");
    if (counts[SPLIT_TOO_FAR])
        return fail(c, PBF_ERR_INVALID, "a particle crossed more than ghost_layers cell planes beyond its slab in one sub-step");
    if (counts[SPLIT_MIG_LEFT] > c->split_max || counts[SPLIT_MIG_RIGHT] > c->split_max)
        return fail(c, PBF_ERR_CAPACITY, "migration buffer too small");
    if (counts[SPLIT_GHOST_LEFT] > c->split_max || counts[SPLIT_GHOST_RIGHT] > c->split_max)
        return fail(c, PBF_ERR_CAPACITY, "ghost buffer too small");
    c->n = counts[SPLIT_STAY];
    c->keyed_count = c->n;
    c->bcount[0] = counts[SPLIT_GHOST_LEFT];
    c->bcount[1] = counts[SPLIT_GHOST_RIGHT];
    c->mig_out_count[0] = counts[SPLIT_MIG_LEFT];
    c->mig_out_count[1] = counts[SPLIT_MIG_RIGHT];
    c->in_exchange = true;
    return PBF_OK;
}

static int slab_append_one(pbf_ctx* c, const void* d_records, uint32_t count, bool ghost, bool compact = false) {
    if ((uint64_t)c->n + count > c->cap) return fail(c, PBF_ERR_CAPACITY, "slab exceeds ctx capacity");
    if (count) {
        if (!d_records) return fail(c, PBF_ERR_INVALID, "null records");
        if (compact)
            k_slab_append_compact<<<nblocks(count), BLOCK, 0, c->stream>>>(
                (const float4*)d_records, count, c->n, c->uniform_mass, c->pos[c->cur], c->vel[c->cur],
                c->prd[c->prd_cur], c->id[c->cur]);
        else
            k_slab_append<<<nblocks(count), BLOCK, 0, c->stream>>>(
                (const float*)d_records, count, c->n, ghost ? GHOST_BIT : 0u, c->pos[c->cur],
                c->vel[c->cur], c->prd[c->prd_cur], c->id[c->cur]);
        LAUNCH_CK(c);
    }
    c->n += count;
    c->grid_valid = false;
    return PBF_OK;
}

int pbf_slab_append_owned(pbf_ctx* c, const void* d_from_left, uint32_t count_left,
                          const void* d_from_right, uint32_t count_right) {
    ENTER(c);
    if (!c->slab) return fail(c, PBF_ERR_INVALID, "pbf_slab_set_bounds has not been called");
    if (c->split_pending) return fail(c, PBF_ERR_INVALID, "pbf_slab_split_done has not been called");
    if (c->ghost_count[0] || c->ghost_count[1])
        return fail(c, PBF_ERR_INVALID, "owned particles must be appended before ghosts");
    const void* rec[2] = {d_from_left, d_from_right};
    const uint32_t cnt[2] = {count_left, count_right};
    for (int side = 0; side < 2; ++side) {
        const uint32_t base = c->n;
        int rc = slab_append_one(c, rec[side], cnt[side], false);
        if (rc) return rc;
        if (c->in_exchange && cnt[side]) {
            // the sender keeps its migrants as ghosts: list them for the halo refreshes
            if ((uint64_t)c->bcount[side] + cnt[side] > c->cap)
                return fail(c, PBF_ERR_CAPACITY, "boundary list full");
            k_slab_list_range<<<nblocks(cnt[side]), BLOCK, 0, c->stream>>>(c->blist[side], c->bcount[side],
                                                                          base, cnt[side]);
            LAUNCH_CK(c);
            c->bcount[side] += cnt[side];
        }
    }
    return PBF_OK;
}

static int slab_append_ghosts_impl(pbf_ctx* c, const void* d_from_left, uint32_t count_left,
                                   const void* d_from_right, uint32_t count_right, bool compact) {
    ENTER(c);
    if (!c->slab) return fail(c, PBF_ERR_INVALID, "pbf_slab_set_bounds has not been called");
    if (!c->in_exchange) return fail(c, PBF_ERR_INVALID, "pbf_slab_split / pbf_slab_split_done come first");
    const void* rec[2] = {d_from_left, d_from_right};
    const uint32_t cnt[2] = {count_left, count_right};
    for (int side = 0; side < 2; ++side) {
        c->ghost_base[side] = c->n;
        int rc = slab_append_one(c, rec[side], cnt[side], true, compact);
        if (rc) return rc;
        // this slab's own migrants to that side stay here as ghosts (their new owner lists them
        // in the same order, right after its boundary-plane particles)
        rc = slab_append_one(c, c->mig_out[side], c->mig_out_count[side], true);
        if (rc) return rc;
        c->ghost_count[side] = c->n - c->ghost_base[side];
    }
    c->in_exchange = false;
    return PBF_OK;
}

int pbf_slab_append_ghosts(pbf_ctx* c, const void* d_from_left, uint32_t count_left,
                           const void* d_from_right, uint32_t count_right) {
    return slab_append_ghosts_impl(c, d_from_left, count_left, d_from_right, count_right, false);
}

static float4* halo_array(pbf_ctx* c, int32_t what) {
    switch (what) {
        case 0: return c->prd[c->prd_cur];
        case 1: return c->vel[c->cur];
        case 2: return c->omega;
        case 3: return c->pos[c->cur];
        default: return nullptr;
    }
}

int pbf_slab_pack_halo(pbf_ctx* c, int32_t what, void* d_left, void* d_right) {
    ENTER(c);
    if (!c->slab || !c->grid_valid) return fail(c, PBF_ERR_INVALID, "halo refresh needs a built slab grid");
    float4* src = halo_array(c, what);
    if (!src) return fail(c, PBF_ERR_INVALID, "bad argument");
    const uint32_t k0 = c->bcount[0], k1 = c->bcount[1];
    if (!(k0 + k1)) return PBF_OK;
    if ((k0 && !d_left) || (k1 && !d_right)) return fail(c, PBF_ERR_INVALID, "null buffer");
    k_slab_pack_halo<<<nblocks(k0 + k1), BLOCK, 0, c->stream>>>(
        src, c->blist[0], c->blist[1], c->slot_of, k0, k1, (float4*)d_left, (float4*)d_right);
    LAUNCH_CK(c);
    return PBF_OK;
}

int pbf_slab_unpack_halo(pbf_ctx* c, int32_t what, const void* d_from_left, uint32_t count_left,
                         const void* d_from_right, uint32_t count_right) {
    ENTER(c);
    if (!c->slab || !c->grid_valid) return fail(c, PBF_ERR_INVALID, "halo refresh needs a built slab grid");
    float4* dst = halo_array(c, what);
    if (!dst) return fail(c, PBF_ERR_INVALID, "bad argument");
    if (count_left != c->ghost_count[0] || count_right != c->ghost_count[1])
        return fail(c, PBF_ERR_INVALID, "halo size differs from the ghost count");
    if (c->in_exchange) return fail(c, PBF_ERR_INVALID, "ghosts have not been appended");
    if (!(count_left + count_right)) return PBF_OK;
    if ((count_left && !d_from_left) || (count_right && !d_from_right))
        return fail(c, PBF_ERR_INVALID, "null buffer");
    // positions keep their own mass in .w; everything else is replaced whole
    k_slab_unpack_halo<<<nblocks(count_left + count_right), BLOCK, 0, c->stream>>>(
        dst, c->slot_of, c->ghost_base[0], count_left, c->ghost_base[1], count_right,
        (const float4*)d_from_left, (const float4*)d_from_right, what == 3);
    LAUNCH_CK(c);
    return PBF_OK;
}

int pbf_slab_halo_exchange(pbf_ctx* c, pbf_comm* comm, int32_t what, int32_t left_rank,
                           int32_t right_rank, void* d_send_left, void* d_send_right,
                           void* d_recv_left, void* d_recv_right) {
    ENTER(c);
    if (!comm) return fail(c, PBF_ERR_INVALID, "null communicator");
    const uint32_t s0 = c->bcount[0], s1 = c->bcount[1], r0 = c->ghost_count[0], r1 = c->ghost_count[1];
    if ((left_rank < 0 && (s0 || r0)) || (right_rank < 0 && (s1 || r1)))
        return fail(c, PBF_ERR_INVALID, "boundary particles / ghosts on an open end");
    if ((r0 && !d_recv_left) || (r1 && !d_recv_right)) return fail(c, PBF_ERR_INVALID, "null buffer");
    int rc = pbf_slab_pack_halo(c, what, d_send_left, d_send_right);
    if (rc) return rc;
    // receives first: a rank whose two neighbours are the same rank (world 2 is not periodic, but a
    // rehearsal on one rank is) still pairs message k with message k
    pbf_p2p ops[4] = {{1, left_rank, d_send_left, (uint64_t)s0 * 16}, {0, left_rank, d_recv_left, (uint64_t)r0 * 16},
                      {1, right_rank, d_send_right, (uint64_t)s1 * 16}, {0, right_rank, d_recv_right, (uint64_t)r1 * 16}};
    rc = pbf_comm_batch(comm, (void*)c->stream, ops, 4);
    if (rc) return fail(c, rc, std::string("halo exchange: ") + pbf_comm_last_error(comm));
    return pbf_slab_unpack_halo(c, what, d_recv_left, r0, d_recv_right, r1);
}

int pbf_slab_attach(pbf_ctx* c, pbf_comm* comm, int32_t left_rank, int32_t right_rank,
                    uint32_t max_records) {
    ENTER(c);
    if (!c->slab) return fail(c, PBF_ERR_INVALID, "pbf_slab_set_bounds has not been called");
    if (!comm) return fail(c, PBF_ERR_INVALID, "null communicator");
    const int32_t world = pbf_comm_world(comm), me = pbf_comm_rank(comm);
    if (left_rank >= world || right_rank >= world || left_rank == me || right_rank == me)
        return fail(c, PBF_ERR_INVALID, "bad neighbour rank");
    HIP_CK(c, hipStreamSynchronize(c->stream));
    free_link(c);
    if (!max_records) max_records = c->cap / 4 > 65536u ? c->cap / 4 : 65536u;
    if (max_records > c->cap) max_records = c->cap;
#define L_CK(call)                                                                       \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            free_link(c);                                                                \
            return fail(c, PBF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                                \
    } while (0)
    for (int k = 0; k < 2; ++k) {
        L_CK(dalloc(&c->x_mig[k], (size_t)max_records * PBF_RECORD_FLOATS));
        L_CK(dalloc(&c->x_gh[k], (size_t)max_records * PBF_RECORD_FLOATS));
        L_CK(dalloc(&c->x_rmig[k], (size_t)max_records * PBF_RECORD_FLOATS));
        L_CK(dalloc(&c->x_rgh[k], (size_t)max_records * PBF_RECORD_FLOATS));
        L_CK(dalloc(&c->x_hsend[k], (size_t)max_records));
        L_CK(dalloc(&c->x_hrecv[k], (size_t)max_records));
    }
    L_CK(dalloc(&c->x_counts, (size_t)12));
    // on the ctx stream and waited for: a null-stream memset is not ordered with the (non-blocking)
    // ctx stream and may land after the first split has written its counters
    L_CK(hipMemsetAsync(c->x_counts, 0, 12 * sizeof(uint32_t), c->stream));   // open ends stay 0 for good
    L_CK(hipStreamSynchronize(c->stream));
    L_CK(hipHostMalloc((void**)&c->x_host_counts, 12 * sizeof(uint32_t), hipHostMallocDefault));
#undef L_CK
    c->link_comm = comm;
    c->link_left = left_rank;
    c->link_right = right_rank;
    c->link_max = max_records;
    return PBF_OK;
}

// one PBF sub-step of this rank's slab, exchanges included: the protocol of
// fluidsimulation_amd/slab.py::SlabGroup.step for one slab, issued from C++ so that the launches
// behind the sub-step's one host synchronisation follow each other within microseconds
int pbf_slab_step(pbf_ctx* c, float dt) {
    ENTER(c);
    if (!c->link_comm) return fail(c, PBF_ERR_INVALID, "pbf_slab_attach has not been called");
    pbf_comm* comm = c->link_comm;
    const int32_t L = c->link_left, R = c->link_right;
    // predict + collide + split + key/histogram of the stayers in ONE pass over the slab
    // ghosts travel as one float4 each unless a later stage reads their velocity / old position
    // (viscosity) or masses differ; every rank takes the same decision from the shared parameters
    const bool compact = c->uniform && c->P.vorticity_eps == 0.f && c->P.xsph_c == 0.f;
    // the sub-step ends in the fused velocity update, which reads no velocity: the predicted ones
    // are neither stored nor sorted (records still carry them)
    const bool vel_dead = can_fuse_finalize(c);
    int rc = slab_split_impl(c, c->x_mig[0], c->x_mig[1], c->x_gh[0], c->x_gh[1], c->link_max, c->x_counts, true, dt,
                             compact, !vel_dead);
    if (rc) return rc;
    // counts to the neighbours device-to-device, then the ONE read-back of the sub-step
    {
        uint32_t* d = c->x_counts;
        pbf_p2p ops[4] = {{1, L, d + 2, L < 0 ? 0u : 8u}, {0, L, d + 8, L < 0 ? 0u : 8u},
                          {1, R, d + 4, R < 0 ? 0u : 8u}, {0, R, d + 10, R < 0 ? 0u : 8u}};
        rc = pbf_comm_batch(comm, (void*)c->stream, ops, 4);
        if (rc) return fail(c, rc, std::string("count exchange: ") + pbf_comm_last_error(comm));
        HIP_CK(c, hipMemcpyAsync(c->x_host_counts, d, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_CK(c, hipStreamSynchronize(c->stream));
    }
    const uint32_t* h = c->x_host_counts;
    const uint32_t own[6] = {h[0], h[1], h[2], h[3], h[4], h[5]};
    const uint32_t from_left[2] = {h[8], h[9]}, from_right[2] = {h[10], h[11]};   // {migrants, ghosts}
    if ((rc = pbf_slab_split_done(c, own))) return rc;
    if (from_left[0] > c->link_max || from_left[1] > c->link_max || from_right[0] > c->link_max ||
        from_right[1] > c->link_max)
        return fail(c, PBF_ERR_CAPACITY, "incoming migrants / ghosts exceed the exchange buffers (pbf_slab_attach max_records)");
    {
        const uint64_t rb = PBF_RECORD_FLOATS * sizeof(float), gb = compact ? sizeof(float4) : rb;
        pbf_p2p ops[8] = {{1, L, c->x_mig[0], own[2] * rb}, {0, L, c->x_rmig[0], from_left[0] * rb},
                          {1, L, c->x_gh[0], own[3] * gb},  {0, L, c->x_rgh[0], from_left[1] * gb},
                          {1, R, c->x_mig[1], own[4] * rb}, {0, R, c->x_rmig[1], from_right[0] * rb},
                          {1, R, c->x_gh[1], own[5] * gb},  {0, R, c->x_rgh[1], from_right[1] * gb}};
        if (L < 0) for (int k = 0; k < 4; ++k) ops[k].bytes = 0;
        if (R < 0) for (int k = 4; k < 8; ++k) ops[k].bytes = 0;
        rc = pbf_comm_batch(comm, (void*)c->stream, ops, 8);
        if (rc) return fail(c, rc, std::string("payload exchange: ") + pbf_comm_last_error(comm));
    }
    if ((rc = pbf_slab_append_owned(c, c->x_rmig[0], L < 0 ? 0 : from_left[0], c->x_rmig[1], R < 0 ? 0 : from_right[0]))) return rc;
    if ((rc = slab_append_ghosts_impl(c, c->x_rgh[0], L < 0 ? 0 : from_left[1], c->x_rgh[1], R < 0 ? 0 : from_right[1], compact))) return rc;
    if (c->bcount[0] > c->link_max || c->bcount[1] > c->link_max || c->ghost_count[0] > c->link_max ||
        c->ghost_count[1] > c->link_max)
        return fail(c, PBF_ERR_CAPACITY, "halo exceeds the exchange buffers (pbf_slab_attach max_records)");
    if ((rc = build_grid_impl(c, !vel_dead))) return rc;
    auto halo = [&](int32_t what) {
        return pbf_slab_halo_exchange(c, comm, what, L, R, c->x_hsend[0], c->x_hsend[1], c->x_hrecv[0], c->x_hrecv[1]);
    };
    const bool vort = c->P.vorticity_eps != 0.f, xsph = c->P.xsph_c != 0.f, viscous = vort || xsph;
    const int32_t layers = c->ghost_layers;
    const bool local_sweeps = layers >= 2;
    int32_t valid = layers;   // ghost layers whose data is still exact (slab.py, "communication-avoiding halo")
    const bool fused = can_fuse_finalize(c);
    for (int32_t it = 0; it < c->iterations; ++it) {
        if (local_sweeps && valid < 2) {
            if ((rc = halo(PBF_HALO_PRED_LAMBDA))) return rc;
            valid = layers;
        }
        // what the sweeps of this iteration still have to produce: lambda on valid - 1 ghost planes,
        // delta-p on valid - 2 (the last iteration: only what the owned particles need)
        const bool last = it + 1 == c->iterations;
        const int32_t depth_l = !local_sweeps ? -1 : (last ? 1 : valid - 1);
        const int32_t depth_d = !local_sweeps ? -1 : (last ? 0 : valid - 2);
        if (c->n && (rc = solve_half(c, 0, dt, false, depth_l, last))) return rc;
        if (!local_sweeps && (rc = halo(PBF_HALO_PRED_LAMBDA))) return rc;
        if (c->n && (rc = solve_half(c, 1, dt, last && fused, depth_d))) return rc;
        if (local_sweeps) valid -= 2;
        else if (!last && (rc = halo(PBF_HALO_PRED_LAMBDA))) return rc;
    }
    if (fused) return PBF_OK;
    if (viscous && (rc = halo(PBF_HALO_PRED_LAMBDA))) return rc;
    if ((rc = pbf_finalize_stage(c, 0, dt))) return rc;
    if (vort) {
        if ((rc = pbf_finalize_stage(c, 1, dt))) return rc;
        if (!local_sweeps && (rc = halo(PBF_HALO_OMEGA))) return rc;
        if ((rc = pbf_finalize_stage(c, 2, dt))) return rc;
        if (xsph && (rc = halo(PBF_HALO_VELOCITY))) return rc;
    }
    if (xsph && (rc = pbf_finalize_stage(c, 3, dt))) return rc;
    return PBF_OK;
}

// host-pointer variants for callers without a device allocator of their own (the C++ driver)
int pbf_slab_append_owned_host(pbf_ctx* c, const float* records, uint32_t count) {
    ENTER(c);
    if (!count) return PBF_OK;
    if (!records) return fail(c, PBF_ERR_INVALID, "null records");
    float* d = nullptr;
    HIP_CK(c, dalloc(&d, (size_t)count * PBF_RECORD_FLOATS));
    hipError_t e = hipMemcpyAsync(d, records, (size_t)count * PBF_RECORD_FLOATS * sizeof(float),
                                  hipMemcpyHostToDevice, c->stream);
    int rc = PBF_OK;
    if (e != hipSuccess) rc = fail(c, PBF_ERR_HIP, std::string("append_owned_host: ") + hipGetErrorString(e));
    if (rc == PBF_OK) rc = pbf_slab_append_owned(c, d, count, nullptr, 0);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    return rc;
}

int pbf_slab_export_host(pbf_ctx* c, float* records, uint32_t max_records, uint32_t* count) {
    ENTER(c);
    if (!count || (max_records && !records)) return fail(c, PBF_ERR_INVALID, "null output");
    float* d = nullptr;
    HIP_CK(c, dalloc(&d, (size_t)max_records * PBF_RECORD_FLOATS));
    int rc = pbf_slab_export(c, d, max_records, count);
    if (rc == PBF_OK && *count) {
        hipError_t e = hipMemcpyAsync(records, d, (size_t)*count * PBF_RECORD_FLOATS * sizeof(float),
                                      hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, PBF_ERR_HIP, std::string("export_host: ") + hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

uint32_t pbf_slab_owned_count(const pbf_ctx* c) {
    return c ? c->n - c->ghost_count[0] - c->ghost_count[1] : 0;
}

int pbf_slab_export(pbf_ctx* c, void* d_records, uint32_t max_records, uint32_t* count) {
    ENTER(c);
    if (!count) return fail(c, PBF_ERR_INVALID, "null count");
    *count = 0;
    if (!c->slab_cnt) return fail(c, PBF_ERR_INVALID, "ctx not initialised");
    uint32_t* counter = &c->slab_cnt->exported;
    HIP_CK(c, hipMemsetAsync(counter, 0, 4, c->stream));
    if (c->n) {
        k_slab_export<<<nblocks(c->n), BLOCK, 0, c->stream>>>(c->pos[c->cur], c->vel[c->cur],
                                                            c->prd[c->prd_cur], c->id[c->cur], c->n,
                                                            (float*)d_records, max_records, counter);
        LAUNCH_CK(c);
    }
    uint32_t h = 0;
    HIP_CK(c, hipMemcpyAsync(&h, counter, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    if (h > max_records) return fail(c, PBF_ERR_CAPACITY, "export buffer too small");
    *count = h;
    return PBF_OK;
}

int pbf_step(pbf_ctx* c, float dt) {
    ENTER(c);
    const bool vel_dead = can_fuse_finalize(c);   // the last delta-p launch rewrites every velocity
    int rc = predict_impl(c, dt, !vel_dead);
    if (rc) return rc;
    rc = sort_and_neighbors(c, true, !vel_dead);
    if (rc) return rc;
    if (c->first_frame) {
        rc = pbf_compute_rest_density(c);
        if (rc) return rc;
        c->first_frame = false;
    }
    if (can_fuse_finalize(c)) {
        for (int32_t it = 0; it < c->iterations; ++it) {
            const bool last = it + 1 == c->iterations;
            if ((rc = solve_half(c, 0, dt, false, -1, last))) return rc;
            if ((rc = solve_half(c, 1, dt, last))) return rc;
        }
        return PBF_OK;
    }
    rc = pbf_solve(c, c->iterations, dt);
    if (rc) return rc;
    if (c->ncons && (rc = pbf_project_constraints(c, dt))) return rc;  // Simulation.cpp:121
    return pbf_finalize(c, dt);
}

int pbf_set_distance_constraints(pbf_ctx* c, const pbf_distance_constraint* list, uint32_t count,
                                 int32_t mode) {
    ENTER(c);
    if (count && !list) return fail(c, PBF_ERR_INVALID, "null constraint list");
    if (mode != PBF_CONSTRAINTS_PBD && mode != PBF_CONSTRAINTS_XPBD) return fail(c, PBF_ERR_INVALID, "bad constraint mode");
    if (c->slab && count) return fail(c, PBF_ERR_INVALID, "constraints are not available in slab mode (particles are addressed by global index)");
    HIP_CK(c, hipStreamSynchronize(c->stream));
    free_constraints(c);
    c->cons_mode = mode;
    if (!count) return PBF_OK;
    // levels: a constraint runs one level after the latest earlier constraint on either particle
    uint32_t max_p = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (list[k].p0 == list[k].p1) return fail(c, PBF_ERR_INVALID, "constraint joins a particle to itself");
        max_p = list[k].p0 > max_p ? list[k].p0 : max_p;
        max_p = list[k].p1 > max_p ? list[k].p1 : max_p;
    }
    if (max_p >= c->cap) return fail(c, PBF_ERR_INVALID, "constraint particle index exceeds ctx capacity");
    std::vector<uint32_t> last((size_t)max_p + 1, 0), level(count);
    uint32_t n_levels = 0;
    for (uint32_t k = 0; k < count; ++k) {
        const uint32_t l = (last[list[k].p0] > last[list[k].p1] ? last[list[k].p0] : last[list[k].p1]) + 1;
        level[k] = l - 1;
        last[list[k].p0] = last[list[k].p1] = l;
        n_levels = l > n_levels ? l : n_levels;
    }
    std::vector<uint32_t> start((size_t)n_levels + 1, 0);
    for (uint32_t k = 0; k < count; ++k) start[level[k] + 1]++;
    for (uint32_t l = 0; l < n_levels; ++l) start[l + 1] += start[l];
    std::vector<uint32_t> cursor(start.begin(), start.end() - 1), order(count);
    for (uint32_t k = 0; k < count; ++k) order[cursor[level[k]]++] = k;   // stable inside a level
    std::vector<uint32_t> p0(count), p1(count);
    std::vector<float> rest(count), stiff(count), compl_(count);
    for (uint32_t t = 0; t < count; ++t) {
        const pbf_distance_constraint& q = list[order[t]];
        p0[t] = q.p0;
        p1[t] = q.p1;
        rest[t] = q.rest_length;
        stiff[t] = q.stiffness > 1.f ? 1.f : (q.stiffness < 0.f ? 0.f : q.stiffness);  // setStiffness
        compl_[t] = q.compliance;
    }
    // launches: a level of more than 4 workgroups' worth gets its own grid, runs of smaller levels
    // share one workgroup
    const uint32_t big = 4 * BLOCK;
    for (uint32_t l = 0; l < n_levels;) {
        const uint32_t sz = start[l + 1] - start[l];
        if (sz > big) {
            c->cons_batches.push_back({l, 1, start[l], sz, false});
            ++l;
        } else {
            uint32_t e = l;
            while (e < n_levels && start[e + 1] - start[e] <= big) ++e;
            c->cons_batches.push_back({l, e - l, start[l], start[e] - start[l], true});
            l = e;
        }
    }
#define C_CK(call)                                                                       \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            free_constraints(c);                                                         \
            return fail(c, PBF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                                \
    } while (0)
    C_CK(dalloc(&c->cons_p0, (size_t)count));
    C_CK(dalloc(&c->cons_p1, (size_t)count));
    C_CK(dalloc(&c->cons_rest, (size_t)count));
    C_CK(dalloc(&c->cons_stiff, (size_t)count));
    C_CK(dalloc(&c->cons_compl, (size_t)count));
    C_CK(dalloc(&c->cons_lambda, (size_t)count));
    C_CK(dalloc(&c->cons_level_start, (size_t)n_levels + 1));
    C_CK(dalloc(&c->slot_by_id, (size_t)c->cap));
    // everything on the ctx stream, then waited for (the host vectors die at return; null-stream
    // copies / memsets are not ordered with the non-blocking ctx stream)
    C_CK(hipMemcpyAsync(c->cons_p0, p0.data(), (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
    C_CK(hipMemcpyAsync(c->cons_p1, p1.data(), (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
    C_CK(hipMemcpyAsync(c->cons_rest, rest.data(), (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
    C_CK(hipMemcpyAsync(c->cons_stiff, stiff.data(), (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
    C_CK(hipMemcpyAsync(c->cons_compl, compl_.data(), (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
    C_CK(hipMemcpyAsync(c->cons_level_start, start.data(), ((size_t)n_levels + 1) * 4, hipMemcpyHostToDevice, c->stream));
    C_CK(hipMemsetAsync(c->cons_lambda, 0, (size_t)count * 4, c->stream));
    C_CK(hipStreamSynchronize(c->stream));
#undef C_CK
    c->cons_order = order;
    c->ncons = count;
    c->ncons_levels = n_levels;
    c->cons_max_particle = max_p;
    return PBF_OK;
}

int pbf_project_constraints(pbf_ctx* c, float dt) {
    ENTER(c);
    if (!c->ncons || !c->n) return PBF_OK;
    if (c->cons_max_particle >= c->n) return fail(c, PBF_ERR_INVALID, "constraint particle index exceeds the particle count");
    const bool xpbd = c->cons_mode == PBF_CONSTRAINTS_XPBD;
    const ConstraintSoA K = {c->cons_p0, c->cons_p1, c->cons_rest, c->cons_stiff, c->cons_compl, c->cons_lambda};
    k_slot_by_id<<<nblocks(c->n), BLOCK, 0, c->stream>>>(c->id[c->cur], c->n, c->slot_by_id);
    if (xpbd) HIP_CK(c, hipMemsetAsync(c->cons_lambda, 0, (size_t)c->ncons * 4, c->stream));  // InitLambda
    float4* prd = c->prd[c->prd_cur];
    const float4* pos = c->pos[c->cur];
    for (int32_t it = 0; it < c->iterations; ++it)
        for (const auto& b : c->cons_batches) {
            if (b.one_block) {
                if (xpbd) k_constraints_levels<true><<<1, BLOCK, 0, c->stream>>>(c->cons_level_start, b.first_level, b.n_levels, K, c->slot_by_id, pos, prd, dt);
                else k_constraints_levels<false><<<1, BLOCK, 0, c->stream>>>(c->cons_level_start, b.first_level, b.n_levels, K, c->slot_by_id, pos, prd, dt);
            } else {
                if (xpbd) k_constraints_level<true><<<nblocks(b.count), BLOCK, 0, c->stream>>>(b.first, b.count, K, c->slot_by_id, pos, prd, dt);
                else k_constraints_level<false><<<nblocks(b.count), BLOCK, 0, c->stream>>>(b.first, b.count, K, c->slot_by_id, pos, prd, dt);
            }
        }
    LAUNCH_CK(c);
    return PBF_OK;
}

int pbf_get_constraint_lambdas(pbf_ctx* c, float* lambdas, uint32_t* levels) {
    ENTER(c);
    if (levels) *levels = c->ncons_levels;
    if (!lambdas || !c->ncons) return PBF_OK;
    std::vector<float> tmp(c->ncons);
    HIP_CK(c, hipMemcpyAsync(tmp.data(), c->cons_lambda, (size_t)c->ncons * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    for (uint32_t t = 0; t < c->ncons; ++t) lambdas[c->cons_order[t]] = tmp[t];
    return PBF_OK;
}

int pbf_step_dem(pbf_ctx* c, float dt, const pbf_dem_params* dem) {
    ENTER(c);
    if (!dem) return fail(c, PBF_ERR_INVALID, "null dem params");
    if (!(dem->particle_radius > 0.f) || dem->iterations < 0) return fail(c, PBF_ERR_INVALID, "bad dem params");
    if (c->slab) return fail(c, PBF_ERR_INVALID, "the DEM solver is not available in slab mode");
    int rc = pbf_predict(c, dt);
    if (rc) return rc;
    if ((rc = sort_and_neighbors(c, false))) return rc;
    const uint32_t n = c->n;
    if (!n) return PBF_OK;
    const DemConsts D = {2.f * dem->particle_radius, dem->static_friction, dem->kinematic_friction};
    const GridConsts G = make_grid(c->P);
    {
        PhaseScope ps(c, PH_DELTA);
        for (int32_t it = 0; it <= dem->iterations; ++it) {
            const bool friction = it == dem->iterations;
            const float4* x = c->prd[c->prd_cur];
            float4* y = c->prd[(c->prd_cur + 1) % 3];
#define LAUNCH_DEM(F, U)                                                                          \
    k_dem_pass<F, U><<<nblocks(n), BLOCK, 0, c->stream>>>(x, y, c->pos[c->cur], c->cell_start, n, G, D, \
                                                          c->uniform_mass, c->bprd, c->bmass,     \
                                                          c->nb ? c->b_cell_start : nullptr)
            if (friction && c->uniform) LAUNCH_DEM(true, true);
            else if (friction) LAUNCH_DEM(true, false);
            else if (c->uniform) LAUNCH_DEM(false, true);
            else LAUNCH_DEM(false, false);
#undef LAUNCH_DEM
            c->prd_cur = (c->prd_cur + 1) % 3;
        }
        LAUNCH_CK(c);
    }
    return finalize_stage(c, 0, dt);
}

int pbf_sync(pbf_ctx* c) {
    ENTER(c);
    return check_flags(c);
}

int pbf_download(pbf_ctx* c, float* pos, float* vel, float* pred) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "id-indexed read-back is not available in slab mode (ids are global): use pbf_slab_export");
    int rc = download3(c, c->pos[c->cur], pos);
    if (rc) return rc;
    rc = download3(c, c->vel[c->cur], vel);
    if (rc) return rc;
    rc = download3(c, c->prd[c->prd_cur], pred);
    if (rc) return rc;
    HIP_CK(c, hipStreamSynchronize(c->stream));
    return PBF_OK;
}

int pbf_get_scalars(pbf_ctx* c, float* density, float* C, float* lambda) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "id-indexed read-back is not available in slab mode (ids are global): use pbf_slab_export");
    const uint32_t n = c->n;
    if (!n) return PBF_OK;
    // staging holds cap*16 B: three cap-float planes fit
    float* st = (float*)c->staging;
    float *o_d = st, *o_c = st + c->cap, *o_l = st + 2 * (size_t)c->cap;
    k_unpack_scalars<<<nblocks(n), BLOCK, 0, c->stream>>>(c->dens, c->C, c->prd[c->prd_cur],
                                                          c->id[c->cur], n, o_d, o_c, o_l);
    LAUNCH_CK(c);
    if (density) HIP_CK(c, hipMemcpyAsync(density, o_d, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (C) HIP_CK(c, hipMemcpyAsync(C, o_c, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (lambda) HIP_CK(c, hipMemcpyAsync(lambda, o_l, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    return PBF_OK;
}

int pbf_get_cells(pbf_ctx* c, int32_t* cell_xyz, uint32_t* cell_key_out, uint32_t* sorted_id,
                  uint32_t* cell_start) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "id-indexed read-back is not available in slab mode (ids are global): use pbf_slab_export");
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    const uint32_t n = c->n;
    if (n && (cell_xyz || cell_key_out)) {
        int32_t* xyz = (int32_t*)c->staging;           // n*12 B
        uint32_t* key = (uint32_t*)c->staging + 3 * (size_t)c->cap;  // + n*4 B  (cap*16 total)
        k_cells_by_id<<<nblocks(n), BLOCK, 0, c->stream>>>(c->prd[c->prd_cur], c->id[c->cur], n,
                                                           make_grid(c->P), xyz, key);
        LAUNCH_CK(c);
        if (cell_xyz) HIP_CK(c, hipMemcpyAsync(cell_xyz, xyz, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
        if (cell_key_out) HIP_CK(c, hipMemcpyAsync(cell_key_out, key, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (n && sorted_id)
        HIP_CK(c, hipMemcpyAsync(sorted_id, c->id[c->cur], (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (cell_start)
        HIP_CK(c, hipMemcpyAsync(cell_start, c->cell_start, ((size_t)c->ncell + 1) * 4,
                                 hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    return PBF_OK;
}

int pbf_get_neighbors(pbf_ctx* c, uint32_t* counts, uint32_t* neighbors, uint32_t stride) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "id-indexed read-back is not available in slab mode (ids are global): use pbf_slab_export");
    if (!c->grid_valid) return fail(c, PBF_ERR_INVALID, "pbf_build_grid must run first");
    if (!counts || !neighbors || stride == 0) return fail(c, PBF_ERR_INVALID, "null output");
    const uint32_t n = c->n;
    if (!n) return PBF_OK;
    uint32_t *d_counts = nullptr, *d_out = nullptr;
    HIP_CK(c, dalloc(&d_counts, (size_t)n));
    hipError_t e = dalloc(&d_out, (size_t)n * stride);
    if (e != hipSuccess) {
        (void)hipFree(d_counts);
        return fail(c, PBF_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    (void)hipMemsetAsync(d_out, 0, (size_t)n * stride * 4, c->stream);
    k_neighbors_by_id<<<nblocks(n), BLOCK, 0, c->stream>>>(c->nbr, c->nbr_count, c->id[c->cur],
                                                           c->id[c->cur], n, c->cap, stride,
                                                           d_counts, d_out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(neighbors, d_out, (size_t)n * stride * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_counts);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, PBF_ERR_HIP, std::string("get_neighbors: ") + hipGetErrorString(e));
    return PBF_OK;
}

int pbf_get_stats(pbf_ctx* c, pbf_stats* out) {
    ENTER(c);
    if (!out) return fail(c, PBF_ERR_INVALID, "null output");
    memset(out, 0, sizeof(*out));
    out->n = c->n;
    HIP_CK(c, hipMemcpyAsync(&out->rest_density, c->rho0, 4, hipMemcpyDeviceToHost, c->stream));
    DevFlags f;
    HIP_CK(c, hipMemcpyAsync(&f, c->flags, sizeof(f), hipMemcpyDeviceToHost, c->stream));
    uint32_t nb = nblocks(c->n);
    if (nb > 1024) nb = 1024;
    std::vector<StatPartial> parts(nb);
    if (c->n) {
        k_stats<<<nb, BLOCK, 0, c->stream>>>(c->pos[c->cur], c->vel[c->cur], c->C, c->nbr_count, c->id[c->cur], c->n,
                                             c->stat_partials);
        LAUNCH_CK(c);
        HIP_CK(c, hipMemcpyAsync(parts.data(), c->stat_partials, nb * sizeof(StatPartial),
                                 hipMemcpyDeviceToHost, c->stream));
    }
    HIP_CK(c, hipStreamSynchronize(c->stream));
    out->neighbor_overflow = f.neighbor_overflow;
    double sa = 0, sc = 0, mx = 0, ms = 0, cx = 0, cy = 0, cz = 0, v2 = 0;
    unsigned long long tot = 0, owned = 0;
    for (auto& p : parts) {
        owned += p.count;
        sa += p.sum_abs_C;
        sc += p.sum_C;
        mx = p.max_abs_C > mx ? p.max_abs_C : mx;
        ms += p.msum;
        cx += p.mx;
        cy += p.my;
        cz += p.mz;
        v2 = p.max_speed2 > v2 ? p.max_speed2 : v2;
        tot += p.nbr_total;
        out->nonfinite += p.nonfinite;
        out->max_neighbor_count = p.max_nbr > out->max_neighbor_count ? p.max_nbr : out->max_neighbor_count;
    }
    out->n = (uint32_t)owned;   // slab mode: ghosts are not counted
    if (owned) {
        out->mean_abs_C = sa / owned;
        out->mean_C = sc / owned;
        out->mean_neighbor_count = (double)tot / owned;
    }
    out->max_abs_C = mx;
    if (ms > 0) {
        out->com[0] = cx / ms;
        out->com[1] = cy / ms;
        out->com[2] = cz / ms;
    }
    out->max_speed = sqrt(v2);
    return PBF_OK;
}

int pbf_export_positions(pbf_ctx* c, void* d_xyz) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "id-indexed read-back is not available in slab mode (ids are global): use pbf_slab_export");
    if (!c->n) return PBF_OK;
    if (!d_xyz) return fail(c, PBF_ERR_INVALID, "null output");
    k_unpack3<<<nblocks(c->n), BLOCK, 0, c->stream>>>(c->pos[c->cur], c->id[c->cur], c->n, (float*)d_xyz);
    LAUNCH_CK(c);
    return PBF_OK;
}

// snapshot file: SnapHeader, pos[n][3], vel[n][3], then mass[n] unless uniform; all in
// particle-index order, little-endian fp32
struct SnapHeader {
    char magic[8];          // "PBFSNAP1"
    uint32_t n;
    uint32_t flags;         // bit 0: first frame still pending, bit 1: per-particle mass follows
    float rest_density;     // the device value (fixed, or the first-frame mean once computed)
    float uniform_mass;
    uint32_t reserved[10];
};
static_assert(sizeof(SnapHeader) == 64, "snapshot header is 64 bytes");

int pbf_snapshot_save(pbf_ctx* c, const char* path) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "snapshots are per ctx in particle-index order: not available in slab mode, use pbf_slab_export");
    if (!path) return fail(c, PBF_ERR_INVALID, "null path");
    SnapHeader hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, "PBFSNAP1", 8);
    hd.n = c->n;
    hd.flags = (c->first_frame ? 1u : 0u) | (c->uniform ? 0u : 2u);
    hd.uniform_mass = c->uniform_mass;
    HIP_CK(c, hipMemcpyAsync(&hd.rest_density, c->rho0, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    std::vector<float> buf((size_t)c->n * 3);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(c, PBF_ERR_INVALID, "cannot open snapshot file for writing");
    int rc = PBF_OK;
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1;
    const float4* arrays[2] = {c->pos[c->cur], c->vel[c->cur]};
    for (int a = 0; a < 2 && ok && rc == PBF_OK; ++a) {
        rc = download3(c, arrays[a], buf.data());
        if (rc == PBF_OK && c->n) ok = fwrite(buf.data(), 12, c->n, f) == c->n;
    }
    if (rc == PBF_OK && ok && !c->uniform && c->n) {
        float* st = (float*)c->staging;
        k_unpack_w<<<nblocks(c->n), BLOCK, 0, c->stream>>>(c->pos[c->cur], c->id[c->cur], c->n, st);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(buf.data(), st, (size_t)c->n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = fail(c, PBF_ERR_HIP, "snapshot: mass read-back failed");
        else
            ok = fwrite(buf.data(), 4, c->n, f) == c->n;
    }
    if (fclose(f) != 0) ok = false;
    if (rc != PBF_OK) return rc;
    if (!ok) return fail(c, PBF_ERR_INVALID, "short write to snapshot file");
    return PBF_OK;
}

int pbf_snapshot_load(pbf_ctx* c, const char* path) {
    ENTER(c);
    if (c->slab) return fail(c, PBF_ERR_INVALID, "snapshots are not available in slab mode");
    if (!path) return fail(c, PBF_ERR_INVALID, "null path");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(c, PBF_ERR_INVALID, "cannot open snapshot file");
    SnapHeader hd;
    if (fread(&hd, sizeof(hd), 1, f) != 1 || memcmp(hd.magic, "PBFSNAP1", 8) != 0) {
        fclose(f);
        return fail(c, PBF_ERR_INVALID, "not a PBFSNAP1 file");
    }
    if (hd.n > c->cap) {
        fclose(f);
        return fail(c, PBF_ERR_CAPACITY, "snapshot holds more particles than the ctx capacity");
    }
    const bool per_particle = hd.flags & 2u;
    std::vector<float> pos((size_t)hd.n * 3), vel((size_t)hd.n * 3), mass(per_particle ? hd.n : 0);
    bool ok = fread(pos.data(), 12, hd.n, f) == hd.n && fread(vel.data(), 12, hd.n, f) == hd.n;
    if (ok && per_particle) ok = fread(mass.data(), 4, hd.n, f) == hd.n;
    fclose(f);
    if (!ok) return fail(c, PBF_ERR_INVALID, "truncated snapshot file");
    int rc = pbf_upload(c, pos.data(), vel.data(), per_particle ? mass.data() : nullptr,
                        hd.uniform_mass, hd.n);
    if (rc != PBF_OK) return rc;
    // resume where the run stopped: the rest density in force and whether frame 1 is done
    HIP_CK(c, hipMemcpyAsync(c->rho0, &hd.rest_density, 4, hipMemcpyHostToDevice, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    c->first_frame = hd.flags & 1u;
    return PBF_OK;
}

int pbf_get_rest_density(pbf_ctx* c, float* out) {
    ENTER(c);
    if (!out) return fail(c, PBF_ERR_INVALID, "null output");
    HIP_CK(c, hipMemcpyAsync(out, c->rho0, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    return PBF_OK;
}

uint32_t pbf_num_particles(const pbf_ctx* c) { return c ? c->n : 0; }

int pbf_profile_enable(pbf_ctx* c, int enable) {
    ENTER(c);
    if (!enable) fold_spans(c);
    c->prof = enable != 0;
    return PBF_OK;
}

int pbf_profile_reset(pbf_ctx* c) {
    ENTER(c);
    fold_spans(c);
    for (int i = 0; i < PH_COUNT; ++i) {
        c->prof_ms[i] = 0;
        c->prof_calls[i] = 0;
    }
    return PBF_OK;
}

int pbf_profile_read(pbf_ctx* c, double* ms, uint64_t* calls, uint32_t max_phases) {
    ENTER(c);
    fold_spans(c);
    for (uint32_t i = 0; i < max_phases && i < PH_COUNT; ++i) {
        if (ms) ms[i] = c->prof_ms[i];
        if (calls) calls[i] = c->prof_calls[i];
    }
    return PBF_OK;
}

int pbf_selftest_sqrt(pbf_ctx* c, uint64_t* mismatches, uint32_t* first_bad) {
    ENTER(c);
    if (!mismatches || !first_bad) return fail(c, PBF_ERR_INVALID, "null output");
    unsigned long long* d_bad = (unsigned long long*)c->staging;
    uint32_t* d_first = (uint32_t*)((char*)c->staging + 8);
    HIP_CK(c, hipMemsetAsync(d_bad, 0, 8, c->stream));
    HIP_CK(c, hipMemsetAsync(d_first, 0xff, 4, c->stream));
    // every non-negative float from the smallest subnormal to +inf (0x7f800000), plus zero
    k_selftest_sqrt<<<4096, BLOCK, 0, c->stream>>>(0u, 0x7f800000u, d_bad, d_first);
    LAUNCH_CK(c);
    unsigned long long hb = 0;
    HIP_CK(c, hipMemcpyAsync(&hb, d_bad, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipMemcpyAsync(first_bad, d_first, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
    *mismatches = hb;
    return PBF_OK;
}

uint32_t pbf_num_phases(void) { return PH_COUNT; }

const char* pbf_phase_name(uint32_t i) { return i < PH_COUNT ? kPhaseNames[i] : ""; }

}  // extern "C"
