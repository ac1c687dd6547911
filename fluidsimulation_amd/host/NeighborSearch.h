// NeighborSearch.h -- mirrors the reference's NeighborSearch call surface (reference
// NeighborSearch.h:28-54): SpatialSearch fills a per-particle neighbour cache that
// FetchNeighbors returns.  The search itself is the device cell grid + counting sort +
// neighbour table behind pbf_build_grid; the cache is downloaded lazily.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>
#include "ParticleSystem.h"

struct GridSize {
    uint32_t x, y, z;
};

class NeighborSearch {
public:
    NeighborSearch(std::shared_ptr<ParticleSystem> particle_system, GridSize grid_size);

    void Initialize();  // NeighborSearch.cpp:27-39
    // NeighborSearch.cpp:131-236.  As in the reference, the grid spacing is a property of the
    // searcher, not of the radius argument; the radius must equal the ctx's effective_radius.
    bool SpatialSearch(float effective_radius);
    // NeighborSearch.cpp:238-241: neighbours of particle i (original ids), in the reference's
    // cache order
    const std::vector<size_t>& FetchNeighbors(size_t i);

    GridSize m_grid_size;
    uint32_t m_num_grid_cells;

    void invalidate() { m_cache_valid = false; }

private:
    bool refresh_cache();
    std::shared_ptr<ParticleSystem> m_particle_system;
    std::vector<std::vector<size_t>> m_search_cache;
    bool m_cache_valid = false;
};
