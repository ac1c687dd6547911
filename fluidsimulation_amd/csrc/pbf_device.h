// pbf_device.h -- device-side arithmetic shared by the PBF kernels (gfx950).
//
// These forms restate the reference CPU path op for op (fp64 SPH kernel evaluation with
// the float literal pi, fp32 vectors with glm's (x*x + y*y) + z*z dot) so results are
// bit-identical with the CPU oracle; the translation unit is compiled with
// -ffp-contract=off and every fused multiply-add below is an explicit fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbf.h"

namespace pbf {

// Host-computed constants of SPHKernel (reference SPHKernel.h:9-97); pow(h,9) is
// evaluated once on the host with the same libm call the reference makes per pair.
struct KernConsts {
    double h2;          // (double)h * (double)h
    double hd;          // (double)h
    double w_coef;      // 315 / (64 pi h^9)
    double spiky_coef;  // -45 / (pi h^6)
    double scorr_k;     // -scorr_k
    double dt_pow;      // (double)dt multiplied scorr_dt_power times
    double inv_w_dq;    // 1.0 / (double)w_dq
    float h;            // effective radius
    float hh;           // h * h in fp32 (square_h, NeighborSearch.cpp:139)
    float g_coef;       // (float)(-945 / (32 pi h^9))
    float w_zero;       // Poly6_W(0, h)
    float w_dq;         // Poly6_W(scorr_dq * h, h)
    float epsilon;
};

struct GridConsts {
    int32_t gx, gy, gz;
    int32_t ox, oy, oz;  // Flooring cell that maps to grid index (0,0,0)
    float spacing;
};

struct ColliderSet {
    pbf_collider c[PBF_MAX_COLLIDERS];
    uint32_t count;
    float restitution;
};

struct float3_ {
    float x, y, z;
};

// XCD-aware workgroup order: hardware deals workgroups round-robin over the 8 XCDs (each
// with a private 4 MiB L2), so blocks b and b+8 share an L2.  Remap so that each XCD walks
// ONE contiguous eighth of the (cell-sorted) particle range: a block's neighbour gathers
// then hit lines its own XCD already holds.  Bijective for any grid size.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t nb) {
    const uint32_t xcd = b & 7u, q = nb >> 3, r = nb & 7u;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (b >> 3);
}

// glm::dot on vec3 (reference include/glm/detail/func_geometric.inl:65-72)
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    float tx = ax * bx, ty = ay * by, tz = az * bz;
    return tx + ty + tz;
}

// NeighborSearch::Flooring (NeighborSearch.cpp:243-252)
__device__ __forceinline__ int32_t floor_cell(float p, float spacing) {
    return (int32_t)floorf(p / spacing);
}

__device__ __forceinline__ int32_t wrap(int32_t c, int32_t g) {
    // almost every cell lies inside the grid; an integer modulo by a run-time divisor costs ~25
    // instructions, three of them per particle
    if ((uint32_t)c < (uint32_t)g) return c;
    int32_t w = c % g;
    return w < 0 ? w + g : w;
}

__device__ __forceinline__ uint32_t cell_key(const GridConsts& G, int32_t cx, int32_t cy, int32_t cz) {
    return ((uint32_t)wrap(cx - G.ox, G.gx) * (uint32_t)G.gy + (uint32_t)wrap(cy - G.oy, G.gy)) *
               (uint32_t)G.gz +
           (uint32_t)wrap(cz - G.oz, G.gz);
}

// ---- STRICT: SPHKernel::Poly6_W (SPHKernel.h:9-30)
__device__ __forceinline__ float poly6_strict(const KernConsts& k, float distance) {
    if (distance >= 0 && distance <= k.h) {
        const double d = (double)distance;
        double d2 = d * d;
        double q = k.h2 - d2;
        double q3 = q * q * q;
        return (float)(k.w_coef * q3);
    }
    return 0.0f;
}

// ---- SPHKernel::Poly6_W_Gradient / Spiky_W_Gradient scalar factor
// (SPHKernel.h:32-54, 78-97); the caller multiplies by diff
template <int GRAD>
__device__ __forceinline__ float grad_scalar_strict(const KernConsts& k, float distance) {
    if (distance >= 0 && distance <= k.h) {
        const double d = (double)distance;
        if (GRAD == PBF_GRAD_SPIKY) {
            double q = k.hd - d;
            double q2 = q * q;
            double scalar = k.spiky_coef * (q2 / distance);
            return (float)scalar;
        }
        double d2 = d * d;
        double q = k.h2 - d2;
        double q2 = q * q;
        return k.g_coef * (float)q2;
    }
    return 0.0f;
}

// Poly6_W and the Poly6 gradient scalar of one distance in one go: they share q and q*q.
// Same values as poly6_strict + grad_scalar_strict<PBF_GRAD_POLY6>, with two exact shortcuts:
//  - q = fma(-d, d, h2): d holds a 24-bit value, so d * d is exact in fp64 and h2 - d * d rounds
//    once either way;
//  - the support test `0 <= distance <= h` becomes q = max(q, 0): distance is a square root (>= 0
//    or NaN); distance > h  <=>  (double)d * d > (double)h * h = h2  <=>  q < 0, and a clamped q
//    gives W = (float)(coef * 0) = 0 and a gradient scalar of g_coef * 0.f = -0.f where the
//    reference returns +0 -- the same "0 * diff" deviation in the sign of a zero that DESIGN.md
//    section 3 already lists (a sum only shows it when it is itself exactly -0).  NaN: max(NaN, 0) = 0,
//    like the reference's failed test.
__device__ __forceinline__ void poly6_pair(const KernConsts& k, float distance, float& w, float& gs) {
    const double d = (double)distance;
    const double q = fmax(fma(-d, d, k.h2), 0.0);
    const double q2 = q * q;
    const double q3 = q2 * q;  // (q * q) * q
    w = (float)(k.w_coef * q3);
    gs = k.g_coef * (float)q2;
}

template <int GRAD>
__device__ __forceinline__ void kernel_pair(const KernConsts& k, float distance, float& w, float& gs) {
    if (GRAD == PBF_GRAD_POLY6) {
        poly6_pair(k, distance, w, gs);
    } else {
        w = poly6_strict(k, distance);
        gs = grad_scalar_strict<GRAD>(k, distance);
    }
}

// Correctly rounded fp32 square root (glm::distance -> sqrtf).  The compiler's expansion of
// sqrtf costs ~17 instructions (input scaling for subnormals, v_sqrt_f32, two-sided fix-up);
// for inputs >= 2^-96 one rsq and one Markstein-style fma correction give the same bits in 5
// (verified EXHAUSTIVELY against sqrtf over every float in [2^-96, FLT_MAX] by
// pbf_selftest_sqrt, tests/test_gpu_parity.py).  Zero, subnormal-range and non-finite inputs
// take the library path.
__device__ __forceinline__ float sqrt_rn(float x) {
    if (x >= 1.2621774e-29f && x <= 3.4028235e38f) {  // [2^-96, FLT_MAX]
        float r = __builtin_amdgcn_rsqf(x);
        float y = x * r;
        float h = 0.5f * r;
        float e = fmaf(-y, y, x);
        y = fmaf(e, h, y);
        return y;
    }
    return sqrtf(x);
}

// sqrt_rn for a squared distance whose root ONLY feeds the SPH kernel functions above: they map +inf and NaN
// to the same zero weight / gradient (a failed support test; fmax(NaN, 0) = 0), so +inf may take the
// rsq path, where it turns into NaN -- one comparison per pair instead of two in the VALU-bound sweeps.
__device__ __forceinline__ float sqrt_rn_for_kernel(float x) {
    if (x >= 1.2621774e-29f) {  // [2^-96, +inf]; NaN fails the test
        float r = __builtin_amdgcn_rsqf(x);
        float y = x * r;
        float h = 0.5f * r;
        float e = fmaf(-y, y, x);
        y = fmaf(e, h, y);
        return y;
    }
    return sqrtf(x);
}

// The x and y components of a float4 record as one register pair (they ARE one: a 16-byte load fills an
// aligned quad), for v_pk_* arithmetic on both; every packed operation rounds each half like the scalar one.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f xy_of(const float4& v) { return v2f{v.x, v.y}; }

// x^4 for a double that holds an fp32 value (the reference's pow(x, 4), Simulation.cpp:701,
// where x = float / float): x*x has at most 48 significant bits and is exact in double, so
// (x*x)*(x*x) rounds ONCE -- it is the correctly rounded x^4, which is what glibc's pow
// returns in all but ~1e-5 of cases, and the result is cast to float afterwards.
__device__ __forceinline__ double pow4_of_float(double x) {
    double a = x * x;
    return a * a;
}

// ---- artificial pressure (Simulation.cpp:698-702), returns (float)scorr.
// x = W / W(dq) is a float division in the reference; here it is the product with the
// double reciprocal rounded to float, which is the correctly rounded float quotient:
// the product is within 2^-52 (relative) of the exact quotient, and the quotient of two
// 24-bit floats lies at least 2^-49 away from any float rounding boundary.  (For subnormal
// quotients ties can differ, but then x^4 underflows to 0 in the final float anyway.)
__device__ __forceinline__ float scorr_strict(const KernConsts& k, float w) {
    float xf = (float)((double)w * k.inv_w_dq);
    double x = pow4_of_float((double)xf);
    double s = k.scorr_k * x;
    s = s * k.dt_pow;
    return (float)s;
}

// ---- neighbour table layout.  The table of a set with `cap` slots holds the k-th neighbour of
// particle i in row k (m_search_cache transposed: row k of every particle contiguous, one coalesced
// read per wave); a row is cap 32-bit words.  Rows are walked four at a time; the builder
// (k_neighbors) pads a particle's list to a multiple of four with a valid index, so a sweep loads
// the four entries of a group without a per-entry predicate or dummy select, from a UNIFORM row base
// (scalar registers) plus the lane's offset, and the gathers of the masked-off tail never leave the
// target array.
// Entry format, per group of 256 consecutive slots (the builder's workgroup), in bits 31:30 of
// every count word of the group (the count is the low 30 bits):
//   0 (ABS32): the neighbour's sorted slot, 32 bits, at word i of the row;
//   1 (REL16): the slot RELATIVE to the particle's own, 16 bits signed -- particles are cell-sorted,
//      so a neighbour is at most about one x-plane of cells away in slot order; picked when the
//      group's whole candidate window lies within +-32767 slots of every particle of the group
//      (cell planes of up to ~30 k particles); decoding is the add the gather's address needs anyway;
//   2 (WIN16): 16 bits = (column r << 12) | index into the group's staged candidate windows (the
//      builder's LDS index, < 4096), decoded as win_base[group][r] + index with the nine window
//      bases of the group in a side table (36 bytes per 256 particles) that a sweeping wave keeps in LDS --
//      for any scene size, whenever the group was built on the LDS path.
//   Both 16-bit formats sit dense at halfword (i & 255) of the group's 1 KiB segment of the row
//   (bytes [1024 (i >> 8), +512)), so a wave reads 128 contiguous bytes per row instead of 256: the
//   rows are the bulk of the sweeps' HBM traffic (20 neighbours x 8 sweeps per particle and
//   sub-step), halved.
// Tables whose query and target sets differ (fluid <-> boundary) are always format 0.
__host__ __device__ __forceinline__ uint32_t nbr_rows(uint32_t K) { return (K + 3u) & ~3u; }
constexpr uint32_t NBR_GROUP_SHIFT = 8;   // 256 slots: one k_neighbors workgroup
constexpr uint32_t NBR_FMT_SHIFT = 30;
constexpr uint32_t NBR_COUNT_MASK = 0x3fffffffu;
constexpr uint32_t NBR_ABS32 = 0u, NBR_REL16 = 1u, NBR_WIN16 = 2u;
constexpr uint32_t NBR_WIN_STRIDE = 16;   // LDS words per group in a sweep's copy of the window bases (9 used)
__host__ __device__ __forceinline__ size_t nbr_base_words(uint32_t cap) { return ((size_t)(cap >> NBR_GROUP_SHIFT) + 2) * 9; }

__device__ __forceinline__ uint32_t nbr_offset(uint32_t i, bool compact) {   // byte offset inside a row
    return compact ? ((i >> NBR_GROUP_SHIFT) << (NBR_GROUP_SHIFT + 2)) + ((i & ((1u << NBR_GROUP_SHIFT) - 1u)) << 1)
                   : i << 2;
}

// entry k of particle i (diagnostics; the sweeps walk rows with for_each_neighbor)
__device__ __forceinline__ uint32_t nbr_entry(const uint32_t* __restrict__ nbr, const uint32_t* __restrict__ win_base,
                                              uint32_t count_word, uint32_t k, uint32_t i, uint32_t cap) {
    const uint32_t fmt = count_word >> NBR_FMT_SHIFT;
    const char* p = reinterpret_cast<const char*>(nbr) + (size_t)k * cap * 4 + nbr_offset(i, fmt != NBR_ABS32);
    if (fmt == NBR_REL16) return (uint32_t)((int32_t)i + (int32_t)*reinterpret_cast<const int16_t*>(p));
    if (fmt == NBR_WIN16) {
        const uint32_t e = *reinterpret_cast<const uint16_t*>(p);
        return win_base[(size_t)(i >> NBR_GROUP_SHIFT) * 9 + (e >> 12)] + (e & 0xfffu);
    }
    return *reinterpret_cast<const uint32_t*>(p);
}

// A sweep's view of the window bases.  `base` is the table's side array (nullptr: a table that never
// uses WIN16), `lds` a workgroup array of NBR_WIN_LDS_WORDS(blockDim.x) words.  Nothing is read unless
// some lane of the wave holds a WIN16 row; then the wave copies the nine bases of its (at most two: 64
// consecutive slots need not start on a group boundary) groups into its own 32 words of `lds` -- LDS
// operations of one wave execute in order, so this needs no workgroup barrier and the sweeps of
// REL16 / ABS32 groups pay one ballot.
struct NbrWin {
    const uint32_t* base;
    uint32_t* lds;
};
__host__ __device__ constexpr uint32_t NBR_WIN_LDS_WORDS(uint32_t block) { return (block / 64u) * 2u * NBR_WIN_STRIDE; }

// The calling lanes are consecutive slots in ascending lane order (all sweeps index particles that
// way) and call this together.  Returns the lane's group's table (nine words), or nullptr when the
// wave has no WIN16 row.
__device__ __forceinline__ const uint32_t* nbr_win_table(const NbrWin& win, uint32_t i, uint32_t fmt, uint32_t cap) {
    if (!win.base || !__builtin_amdgcn_ballot_w64(fmt == NBR_WIN16)) return nullptr;
    uint32_t* t = win.lds + (threadIdx.x >> 6) * (2u * NBR_WIN_STRIDE);
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i) >> NBR_GROUP_SHIFT;
    // the active lanes share the 32 words among themselves (a full wave: lanes 0..31, one each)
    const uint64_t ex = __builtin_amdgcn_ballot_w64(true);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u));
    const uint32_t n_active = (uint32_t)__builtin_popcountll(ex);
    for (uint32_t w = rank; w < 2u * NBR_WIN_STRIDE; w += n_active) {
        const uint32_t g = g0 + w / NBR_WIN_STRIDE, r = w % NBR_WIN_STRIDE;
        const bool in = r < 9u && (size_t)g * 9 + r < nbr_base_words(cap);
        t[w] = in ? win.base[(size_t)g * 9 + r] : 0u;
    }
    __builtin_amdgcn_wave_barrier();
    return t + ((i >> NBR_GROUP_SHIFT) - g0) * NBR_WIN_STRIDE;
}

// The rows are read with BUFFER loads: the descriptor holds the (uniform) address of the current group of
// four rows, the lane's byte offset inside a row is loop-invariant, and row u of the group is the scalar
// offset u * row_bytes -- so a row load costs no vector instruction for its address (global loads need a
// 64-bit per-lane address: one v_lshl_add_u64 per load, ~5 % of a VALU-bound sweep's instruction time).
// cap <= 2^28: off < row_bytes <= 2^30 and 3 * row_bytes + off < 2^32.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t nbr_rows_rsrc(const char* rows) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rows), 0, 0xffffffffu, 0x00020000);
}

// the four entries of the group of rows behind `rsrc` for the lane at byte offset off, as BYTE offsets of
// the neighbours' float4 records (slot << 4: the shift rides on the add that decodes the entry)
__device__ __forceinline__ void nbr_load4(__amdgpu_buffer_rsrc_t rsrc, uint32_t row_bytes, uint32_t off,
                                          uint32_t fmt, uint32_t i, const uint32_t* win_tbl, uint32_t (&jb)[4]) {
    if (fmt == NBR_REL16) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            jb[u] = (i + (uint32_t)(int32_t)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, off, u * row_bytes, 0)) << 4;
    } else if (fmt == NBR_WIN16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, off, u * row_bytes, 0);
            jb[u] = (win_tbl[e >> 12] + (e & 0xfffu)) << 4;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) jb[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, u * row_bytes, 0) << 4;
    }
}

// Visit the neighbour row of particle i in order, four entries and four position gathers in
// flight, the next four entries prefetched behind them.  f(j, xj) is called in row order (the
// summation order of the reference, Simulation.cpp:628,660,692).
template <int BATCH, typename F>
__device__ __forceinline__ void for_each_neighbor(uint32_t i, uint32_t count_word,
                                                  const uint32_t* __restrict__ nbr, uint32_t cap,
                                                  const NbrWin& win,
                                                  const float4* __restrict__ x, F&& f) {
    static_assert(BATCH == 4, "rows are padded to groups of four");
    const uint32_t rel16 = count_word >> NBR_FMT_SHIFT;   // the entry format
    const uint32_t cnt = count_word & NBR_COUNT_MASK;
    const uint32_t off = nbr_offset(i, rel16 != NBR_ABS32);
    const uint32_t row_bytes = cap * 4u;
    const char* __restrict__ rows = reinterpret_cast<const char*>(nbr);   // uniform: row k0 of every particle
    const uint32_t* win_tbl = nbr_win_table(win, i, rel16, cap);
    uint32_t jn[4] = {0u, 0u, 0u, 0u};
    if (cnt) nbr_load4(nbr_rows_rsrc(rows), row_bytes, off, rel16, i, win_tbl, jn);
    for (uint32_t k0 = 0; k0 < cnt; k0 += 4) {
        const uint32_t jb[4] = {jn[0], jn[1], jn[2], jn[3]};
        float4 xj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            // 32-bit byte offset (capacity <= 2^28): the load takes the base from scalar registers
            xj[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(x) + jb[u]);
        rows += 4 * (size_t)row_bytes;
        if (k0 + 4 < cnt) nbr_load4(nbr_rows_rsrc(rows), row_bytes, off, rel16, i, win_tbl, jn);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < cnt) f(jb[u] >> 4, xj[u]);
    }
}

// The same walk with TWO per-particle arrays gathered per neighbour (position + velocity, position
// + vorticity): both gathers of all four neighbours are in flight before the first pair is
// evaluated.  f(j, xj, yj).
template <int BATCH, typename F>
__device__ __forceinline__ void for_each_neighbor2(uint32_t i, uint32_t count_word,
                                                   const uint32_t* __restrict__ nbr, uint32_t cap,
                                                   const NbrWin& win,
                                                   const float4* __restrict__ x,
                                                   const float4* __restrict__ y, F&& f) {
    static_assert(BATCH == 4, "rows are padded to groups of four");
    const uint32_t rel16 = count_word >> NBR_FMT_SHIFT;   // the entry format
    const uint32_t cnt = count_word & NBR_COUNT_MASK;
    const uint32_t off = nbr_offset(i, rel16 != NBR_ABS32);
    const uint32_t row_bytes = cap * 4u;
    const char* __restrict__ rows = reinterpret_cast<const char*>(nbr);
    const uint32_t* win_tbl = nbr_win_table(win, i, rel16, cap);
    uint32_t jn[4] = {0u, 0u, 0u, 0u};
    if (cnt) nbr_load4(nbr_rows_rsrc(rows), row_bytes, off, rel16, i, win_tbl, jn);
    for (uint32_t k0 = 0; k0 < cnt; k0 += 4) {
        const uint32_t jb[4] = {jn[0], jn[1], jn[2], jn[3]};
        float4 xj[4], yj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xj[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(x) + jb[u]);
            yj[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(y) + jb[u]);
        }
        rows += 4 * (size_t)row_bytes;
        if (k0 + 4 < cnt) nbr_load4(nbr_rows_rsrc(rows), row_bytes, off, rel16, i, win_tbl, jn);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < cnt) f(jb[u] >> 4, xj[u], yj[u]);
    }
}

}  // namespace pbf
