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
// Kernels are one thread per particle, 256-thread workgroups (4 wave64), no MFMA: the sweeps
// are VALU-issue bound, the rest HBM/L2-bandwidth or latency bound.
// Compiled with -ffp-contract=off.
//
// ONE translation unit (the kernels live in an anonymous namespace next to the ctx that
// launches them), laid out in files by topic:
//   kernels/*.inc   device code: state, sort, neighbors, solver, boundary, dem, constraints,
//                   viscosity, slab, diag
//   this file       constants, pbf_ctx, host helpers (error plumbing, profiling spans, sort_and_neighbors)
//   abi/*.inc       the extern "C" entry points of include/pbf.h: lifecycle, boundary, substep,
//                   slab, step, constraints, dem, readback, noalloc
// pbf_comm.hip (the RCCL / loopback transport) is a translation unit of its own.

#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <mutex>
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
    PH_EXCHANGE,   // slab mode: count / payload / halo exchanges (pack, grouped send + receive, unpack)
    PH_COUNT
};
const char* kPhaseNames[PH_COUNT] = {"predict", "sort",     "neighbors", "rest_density", "lambda",
                                     "delta",   "finalize", "viscosity", "exchange"};

#include "kernels/state.inc"
#include "kernels/sort.inc"
#include "kernels/neighbors.inc"
#include "kernels/solver.inc"
#include "kernels/boundary.inc"
#include "kernels/dem.inc"
#include "kernels/constraints.inc"
#include "kernels/viscosity.inc"
#include "kernels/slab.inc"
#include "kernels/diag.inc"

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
    uint32_t* nbr_base = nullptr;   // nine window bases per group of 256 slots (WIN16 rows, pbf_device.h)
    uint32_t row_formats = PBF_ROWS_ALL;   // pbf_set_row_formats
    // Crowd limit: the rank-by-id pass and the neighbour build cost O(occupancy of the cell KEY) per
    // particle, so a pile of particles in one cell -- or a grid far too small for the scene, whose keys
    // alias many cells -- turns them quadratic.  A sub-step is refused (DevFlags::crowded_cell) when
    // some key holds more than crowd_threshold(n) particles: beyond BOTH q^3 (K + 1) -- a cell of side s
    // splits into q^3 sub-cubes (q = ceil(sqrt(3) s / h)) whose diagonal is at most h, so an un-aliased
    // cell with more occupants has K + 2 mutual neighbours and overflows its rows whatever the
    // arrangement -- AND 2^34 / n, i.e. only where n x occupancy says seconds of work, not milliseconds
    // (small wrap-around grids alias on purpose and stay legal).
    uint32_t crowd_limit = 0;              // q^3 (K + 1); 0 = unchecked
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
    std::vector<uint32_t> cons_level_start_host;     // slab mode runs level by level (a ghost refresh after each)
    uint32_t cons_max_particle = 0;
    // native slab step (pbf_slab_attach / pbf_slab_step): exchange buffers owned by the ctx
    pbf_comm* link_comm = nullptr;
    int32_t link_left = -1, link_right = -1;
    uint32_t link_max = 0;
    float *x_mig[2] = {nullptr, nullptr}, *x_gh[2] = {nullptr, nullptr};      // out: records
    float *x_rmig[2] = {nullptr, nullptr}, *x_rgh[2] = {nullptr, nullptr};    // in: records
    float4 *x_hsend[2] = {nullptr, nullptr}, *x_hrecv[2] = {nullptr, nullptr};
    uint32_t *x_counts = nullptr;       // device: (1 + world) counter blocks of XC_WORDS: own, then every rank's
    bool x_counts_clean = false;        // the six split counters of the own block are zero (the last sort cleared them)
    uint32_t *x_host_counts = nullptr;  // pinned host mirror of the same size
    uint32_t *x_hist = nullptr, *x_host_hist = nullptr;   // re-balancing: (1 + world) plane histograms
    uint32_t x_hist_planes = 0;
    bool overlap = false;               // pbf_slab_set_overlap
    hipStream_t xstream = nullptr;      // second stream of the overlapped ghost refresh (created on demand)
    hipEvent_t x_packed = nullptr, x_arrived = nullptr;
    bool owns_stream = true;
    bool uniform = false;      // every particle has uniform_mass (no per-neighbour mass gather)
    float uniform_mass = 0.f;
    bool keyed = false;  // histogram of the live prd already accumulated (k_predict / k_slab_split) ...
    uint32_t keyed_count = 0;  // ... for the first keyed_count particles (pre-sort order)
    bool grid_valid = false;
    bool slots_valid = false;   // slot_of / cell_start describe the live arrays (a sort without a neighbour table: the DEM path)
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
// row_depth >= 0 (slab mode): neighbour rows only for the owned cell planes plus that many ghost planes
// per side (k_neighbors' query window); -1: every particle
uint32_t crowd_threshold(const pbf_ctx* c) {
    if (!c->crowd_limit) return 0u;
    const uint64_t by_work = (1ull << 34) / (c->n ? c->n : 1u);
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c->crowd_limit, by_work), 0x40000000ull);
}

int sort_and_neighbors(pbf_ctx* c, bool with_neighbors = true, bool move_vel = true, int32_t row_depth = -1) {
    const uint32_t n = c->n;
    const uint32_t crowd = crowd_threshold(c);
    GridConsts G = make_grid(c->P);
    {
        PhaseScope ps(c, PH_SORT);
        k_scan_tile_sums<<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums);
        // an attached slab: this sub-step's split counters have been exchanged and read back by now
        uint32_t* spent = c->link_comm ? c->x_counts : nullptr;
        if (c->ntiles <= FOLD_MAX_TILES) {
            k_scan_apply<true><<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums,
                                                                   c->cell_start, c->cursor, c->flags, crowd, spent);
        } else {
            k_scan_sums<<<1, BLOCK, 0, c->stream>>>(c->tile_sums, c->ntiles);
            k_scan_apply<false><<<c->ntiles, BLOCK, 0, c->stream>>>(c->cell_count, c->ncell, c->tile_sums,
                                                                    c->cell_start, c->cursor, c->flags, crowd, spent);
        }
        if (spent) c->x_counts_clean = true;
        if (n) {
            const int s = c->cur, d = 1 - c->cur;
            const int ps_ = c->prd_cur, pd = (c->prd_cur + 1) % 3;
            k_scatter<<<nblocks(n), BLOCK, 0, c->stream>>>(c->key, c->id[s], c->cursor, n,
                                                           c->tmp_src, c->tmp_id);
            k_reorder<<<nblocks(n), BLOCK, 0, c->stream>>>(
                c->key, c->cell_start, c->tmp_src, c->tmp_id, n, c->pos[s], c->vel[s],
                c->prd[ps_], c->pos[d], move_vel ? c->vel[d] : nullptr, c->prd[pd], c->id[d],
                c->slab ? c->slot_of : nullptr, c->flags);
            c->cur = d;
            c->prd_cur = pd;
        }
        LAUNCH_CK(c);
    }
    if (!with_neighbors) {
        c->keyed = false;
        c->grid_valid = false;   // cells are sorted, but there is no neighbour table
        c->slots_valid = true;
        return PBF_OK;
    }
    if (n) {
        PhaseScope ps(c, PH_NEIGHBORS);   // the device flags were cleared by k_scan_apply
        int32_t qx_lo = INT32_MIN, qx_hi = INT32_MAX;
        const uint32_t run_limit = crowd ? 3u * crowd : 0xffffffffu;
        if (row_depth >= 0 && c->slab) {
            if (c->x_lo != INT32_MIN) qx_lo = (int32_t)std::max<int64_t>((int64_t)c->x_lo - row_depth, INT32_MIN);
            if (c->x_hi != INT32_MAX) qx_hi = (int32_t)std::min<int64_t>((int64_t)c->x_hi + row_depth, INT32_MAX);
        }
        k_neighbors<true><<<nblocks(n), BLOCK, 0, c->stream>>>(
            c->prd[c->prd_cur], c->prd[c->prd_cur], c->cell_start, n, c->cap, c->K, G,
            c->P.effective_radius * c->P.effective_radius, c->nbr, c->nbr_count, c->flags, c->nbr_base, c->row_formats, run_limit, qx_lo, qx_hi);
        if (c->nb) {
            // fluid -> boundary and boundary -> fluid rows, frozen for the sub-step like the fluid's
            const float hh = c->P.effective_radius * c->P.effective_radius;
            k_neighbors<false><<<nblocks(n), BLOCK, 0, c->stream>>>(
                c->prd[c->prd_cur], c->bpos, c->b_cell_start, n, c->cap, c->Kb, G, hh, c->fb,
                c->fb_count, c->flags, nullptr, PBF_ROWS_ABS32, 0xffffffffu, qx_lo, qx_hi);
            k_neighbors<false><<<nblocks(c->nb), BLOCK, 0, c->stream>>>(
                c->bpos, c->prd[c->prd_cur], c->cell_start, c->nb, c->nb, c->Kb, G, hh, c->bf,
                c->bf_count, c->flags, nullptr, PBF_ROWS_ABS32, run_limit, INT32_MIN, INT32_MAX);
        }
        LAUNCH_CK(c);
    } else if (c->nb) {
        HIP_CK(c, hipMemsetAsync(c->bf_count, 0, (size_t)c->nb * 4, c->stream));
    }
    c->keyed = false;
    c->grid_valid = true;
    c->slots_valid = true;
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
    c->x_counts_clean = false;
    (void)hipFree(c->x_hist);
    if (c->x_host_hist) (void)hipHostFree(c->x_host_hist);
    c->x_hist = c->x_host_hist = nullptr;
    c->x_hist_planes = 0;
    c->link_comm = nullptr;
    c->link_max = 0;
}

void free_overlap(pbf_ctx* c) {
    if (c->x_packed) (void)hipEventDestroy(c->x_packed);
    if (c->x_arrived) (void)hipEventDestroy(c->x_arrived);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    c->x_packed = c->x_arrived = nullptr;
    c->xstream = nullptr;
    c->overlap = false;
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
    c->cons_level_start_host.clear();
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
    // with a boundary set the fluid<->boundary tables (max_boundary_neighbors rows) share the counters
    const std::string rows = "max_neighbors=" + std::to_string(c->K) +
                             (c->nb ? " (fluid rows) or max_boundary_neighbors=" + std::to_string(c->Kb) +
                                          " (fluid<->boundary rows)"
                                    : std::string());
    const char* cure = c->nb ? "); raise pbf_params.max_neighbors / pbf_set_boundary's max_boundary_neighbors"
                             : "); raise pbf_params.max_neighbors";
    if (f.crowded_cell)
        return fail(c, PBF_ERR_NEIGHBOR_OVERFLOW,
                    "a cell key held " + std::to_string(f.crowded_cell) + " particles (limit " +
                        std::to_string(crowd_threshold(c)) + " for " + std::to_string(c->n) +
                        " particles): particles piled up in one cell -- its neighbour rows overflow for any "
                        "arrangement -- or grid_dim is far too small for the scene; the sort and the neighbour "
                        "build, O(occupancy) per particle, were skipped from that sub-step on and the state is "
                        "undefined until the next upload");
    if (f.unreachable_constraints)
        return fail(c, PBF_ERR_INVALID,
                    std::to_string(f.unreachable_constraints) + " distance constraint projection(s) of owned particles were "
                    "skipped: the other endpoint lies beyond this slab's ghost planes (constraints must be shorter than "
                    "ghost_layers cells along x)");
    if (f.neighbor_overflow)
        return fail(c, PBF_ERR_NEIGHBOR_OVERFLOW,
                    std::to_string(f.neighbor_overflow) + " particles exceed " + rows + " (largest count " +
                        std::to_string(f.max_count) + cure);
    if (f.truncated_total)   // an earlier sub-step of this run: the state has been computed from cut rows
        return fail(c, PBF_ERR_NEIGHBOR_OVERFLOW,
                    std::to_string(f.truncated_total) + " neighbour rows were cut at " + rows +
                        " in earlier sub-steps since the last upload (largest count " +
                        std::to_string(f.truncated_max) + cure);
    return PBF_OK;
}

}  // namespace

extern "C" {

#include "abi/lifecycle.inc"
#include "abi/boundary.inc"
#include "abi/substep.inc"
#include "abi/constraints.inc"
#include "abi/slab.inc"
#include "abi/step.inc"
#include "abi/dem.inc"
#include "abi/readback.inc"
#include "abi/noalloc.inc"

}  // extern "C"
