// SlabDriver.h -- host side of the slab decomposition in plain C++ (no HIP, no Python): scene
// partitioning, rank-local parameters and the per-rank step loop over the C ABI's native slab
// runtime (pbf_slab_attach / pbf_slab_step).  The reference is single-GPU and has no
// counterpart; the per-rank step is still Simulation::Step (Simulation.cpp:80-138) on the
// rank's slab.  Mirrors fluidsimulation_amd/slab.py (split_planes, rank_local_params,
// populate), which drives the same entry points from Python.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "Vec3.h"
#include "pbf.h"

namespace slab {

// NeighborSearch::Flooring on the x axis, in fp32 like the kernels
inline int32_t cell_x(float x, float spacing) { return (int32_t)std::floor(x / spacing); }

// planes[0..world]: open ends, interior planes snapped to cell planes at equal particle counts;
// interior slabs at least min_width cells wide
inline std::vector<int32_t> split_planes(const std::vector<Vec3>& pos, float spacing, int world,
                                         int min_width) {
    std::vector<int32_t> planes{INT32_MIN};
    if (world > 1 && !pos.empty()) {
        int32_t lo = INT32_MAX, hi = INT32_MIN;
        for (const Vec3& p : pos) {
            const int32_t c = cell_x(p.x, spacing);
            lo = std::min(lo, c);
            hi = std::max(hi, c);
        }
        std::vector<uint64_t> csum((size_t)(hi - lo) + 1, 0);
        for (const Vec3& p : pos) csum[(size_t)(cell_x(p.x, spacing) - lo)]++;
        for (size_t k = 1; k < csum.size(); ++k) csum[k] += csum[k - 1];
        const double total = (double)csum.back();
        for (int r = 1; r < world; ++r) {
            const double target = total * r / world;
            const size_t k = (size_t)(std::lower_bound(csum.begin(), csum.end(), target,
                                                       [](uint64_t a, double b) { return (double)a < b; }) -
                                      csum.begin()) + 1;
            int32_t plane = lo + (int32_t)k;
            if (planes.back() != INT32_MIN) plane = std::max(plane, planes.back() + min_width);
            planes.push_back(plane);
        }
    } else {
        for (int r = 1; r < world; ++r) planes.push_back(planes.back() == INT32_MIN ? 0 : planes.back() + min_width);
    }
    planes.push_back(INT32_MAX);
    return planes;
}

// this rank's pbf_params: capacity for its share (+ headroom) and a rank-local cell grid that
// spans only the slab's x range (+ ghost planes and a drift margin)
inline pbf_params rank_local_params(const pbf_params& P, const std::vector<Vec3>& pos,
                                    const std::vector<int32_t>& planes, int rank, int world,
                                    double headroom, int ghost_layers) {
    pbf_params p = P;
    const size_t share = (pos.size() + (size_t)world - 1) / (size_t)world;
    p.capacity = (uint32_t)((double)share * headroom) + 65536u;
    int32_t x_min = INT32_MAX, x_max = INT32_MIN;
    if (P.container_enabled) {
        x_min = (int32_t)std::floor(P.container_min[0] / P.grid_spacing);
        x_max = (int32_t)std::floor(std::min(P.container_max[0], 1e30f) / P.grid_spacing);
    } else {
        for (const Vec3& q : pos) {
            const int32_t c = cell_x(q.x, P.grid_spacing);
            x_min = std::min(x_min, c);
            x_max = std::max(x_max, c);
        }
    }
    const int32_t lo = planes[rank] != INT32_MIN ? std::max(planes[rank], x_min) : x_min;
    const int32_t hi = planes[rank + 1] != INT32_MAX ? std::min(planes[rank + 1], x_max + 1) : x_max + 1;
    const int32_t margin = ghost_layers + 2;
    const int32_t width = std::max(hi - lo, 1) + 2 * margin;
    int32_t gx = 16;
    while (gx < width) gx *= 2;
    p.grid_dim[0] = std::min(gx, P.grid_dim[0]);
    p.grid_origin_cell[0] = lo - margin;
    return p;
}

// the global cell planes pbf_slab_rebalance histograms over: the tank when the scene has one, else
// the initial block with a margin of 64 cells (at most 4096 planes)
inline void plane_range(const pbf_params& P, const std::vector<Vec3>& pos, int32_t* x_first, uint32_t* n_planes) {
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    if (P.container_enabled) {
        lo = (int32_t)std::floor(P.container_min[0] / P.grid_spacing);
        hi = (int32_t)std::floor(std::min(P.container_max[0], 1e30f) / P.grid_spacing) + 1;
    } else {
        for (const Vec3& q : pos) {
            const int32_t c = cell_x(q.x, P.grid_spacing);
            lo = std::min(lo, c - 64);
            hi = std::max(hi, c + 65);
        }
        if (pos.empty()) lo = 0, hi = 1;
    }
    *x_first = lo;
    *n_planes = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)hi - lo, 4096));
}

// particle records (include/pbf.h: PBF_RECORD_FLOATS) of the particles whose cell x lies in
// [x_lo, x_hi): pos.xyz, mass, vel.xyz, id bits, pred.xyz, 0
inline std::vector<float> records_in(const std::vector<Vec3>& pos, float mass, float spacing,
                                     int32_t x_lo, int32_t x_hi) {
    std::vector<float> rec;
    for (size_t i = 0; i < pos.size(); ++i) {
        const int32_t c = cell_x(pos[i].x, spacing);
        if (c < x_lo || c >= x_hi) continue;
        const uint32_t id = (uint32_t)i;
        float idf;
        std::memcpy(&idf, &id, 4);
        const float r[PBF_RECORD_FLOATS] = {pos[i].x, pos[i].y, pos[i].z, mass, 0.f, 0.f, 0.f, idf,
                                            pos[i].x, pos[i].y, pos[i].z, 0.f};
        rec.insert(rec.end(), r, r + PBF_RECORD_FLOATS);
    }
    return rec;
}

inline uint32_t record_id(const float* rec) {
    uint32_t id;
    std::memcpy(&id, rec + 7, 4);
    return id & 0x7fffffffu;
}

}  // namespace slab
