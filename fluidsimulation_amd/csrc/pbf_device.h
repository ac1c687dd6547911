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
    float h;            // effective radius
    float hh;           // h * h in fp32 (square_h, NeighborSearch.cpp:139)
    float g_coef;       // (float)(-945 / (32 pi h^9))
    float w_zero;       // Poly6_W(0, h)
    float w_dq;         // Poly6_W(scorr_dq * h, h)
    float epsilon;
    float dt;
    int32_t scorr_dt_power;
    int32_t grad_kernel;
};

struct GridConsts {
    int32_t gx, gy, gz;
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
    int32_t w = c % g;
    return w < 0 ? w + g : w;
}

__device__ __forceinline__ uint32_t cell_key(const GridConsts& G, int32_t cx, int32_t cy, int32_t cz) {
    return ((uint32_t)wrap(cx, G.gx) * (uint32_t)G.gy + (uint32_t)wrap(cy, G.gy)) * (uint32_t)G.gz +
           (uint32_t)wrap(cz, G.gz);
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

// ---- STRICT: SPHKernel::Poly6_W_Gradient / Spiky_W_Gradient scalar factor
// (SPHKernel.h:32-54, 78-97); the caller multiplies by diff
__device__ __forceinline__ float grad_scalar_strict(const KernConsts& k, float distance) {
    if (distance >= 0 && distance <= k.h) {
        const double d = (double)distance;
        if (k.grad_kernel == PBF_GRAD_SPIKY) {
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

// x^4 for x >= 0, correctly rounded (double-double square of a double-double square);
// stands in for the reference's pow(x, 4) (Simulation.cpp:701), which glibc rounds
// correctly in all but ~1e-5 of cases -- and the result is then cast to float.
__device__ __forceinline__ double pow4_cr(double x) {
    double a = x * x;
    double ae = fma(x, x, -a);
    double p = a * a;
    double pe = fma(a, a, -p);
    double t = (2.0 * a) * ae;
    return p + (pe + t);
}

// ---- STRICT artificial pressure (Simulation.cpp:698-702), returns (float)scorr
__device__ __forceinline__ float scorr_strict(const KernConsts& k, float w) {
    double x = (double)(w / k.w_dq);
    x = pow4_cr(x);
    double s = k.scorr_k * x;
    for (int t = 0; t < k.scorr_dt_power; ++t) s = s * (double)k.dt;
    return (float)s;
}

}  // namespace pbf
