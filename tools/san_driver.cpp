// Walks the host code of the C ABI that the two C++ drivers do not reach -- read-backs, the boundary set, distance
// constraints (level scheduling), the granular step, snapshots, the row-format switch, staged calls, the
// no-allocation entry point, and the error paths of bad arguments -- for the host-side AddressSanitizer / UBSan
// build (tools/sanitize_host.sh).  It checks return codes and a few invariants, not parity: that is the tests' job.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pbf.h"

#define REQUIRE(cond)                                                                 \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "%s:%d: %s -- %s\n", __FILE__, __LINE__, #cond, pbf_last_error(ctx)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static pbf_collider plane(float nx, float ny, float nz, float d) {
    pbf_collider c;
    memset(&c, 0, sizeof c);
    c.type = PBF_COLLIDER_PLANE;
    c.a[0] = nx; c.a[1] = ny; c.a[2] = nz;
    c.d = d;
    return c;
}

int main() {
    pbf_ctx* ctx = nullptr;
    const float h = 0.0415283032f, d = 0.0246621166f, dt = 0.0005f, mass = 0.015f;
    const int nx = 20, ny = 16, nz = 12;
    const uint32_t n = nx * ny * nz;
    std::vector<float> pos(3 * (size_t)n), vel(3 * (size_t)n, 0.f), pm(n);
    uint32_t seed = 12345u;
    auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.f; };
    for (int i = 0, k = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int l = 0; l < nz; ++l, ++k) {
                pos[3 * k + 0] = 0.5f * d + d * i + 0.1f * d * (rnd() - 0.5f);
                pos[3 * k + 1] = 0.5f * d + d * j + 0.1f * d * (rnd() - 0.5f);
                pos[3 * k + 2] = 0.5f * d + d * l + 0.1f * d * (rnd() - 0.5f);
                pm[k] = mass * (0.75f + 0.5f * rnd());
            }
    pbf_params P;
    pbf_default_params(&P);
    P.capacity = n + 100;
    P.max_neighbors = 64;
    P.effective_radius = P.grid_spacing = h;
    P.grid_dim[0] = P.grid_dim[1] = P.grid_dim[2] = 32;
    P.grid_origin_cell[0] = P.grid_origin_cell[1] = P.grid_origin_cell[2] = -1;
    P.epsilon = 100.f;
    P.rest_density_mode = PBF_RHO0_FIXED;
    P.rest_density = 1000.f;
    P.solver_iterations = 3;
    P.xsph_c = 0.01f;
    P.vorticity_eps = 2e-6f;

    // ---- misuse first: every call must fail cleanly
    REQUIRE(pbf_create(nullptr, 0, &ctx) != PBF_OK);
    REQUIRE(pbf_create(&P, 0, nullptr) != PBF_OK);
    {
        pbf_params bad = P;
        bad.grid_dim[0] = 2;
        REQUIRE(pbf_create(&bad, 0, &ctx) != PBF_OK && ctx == nullptr);
        bad = P;
        bad.capacity = (1u << 28) + 1u;
        REQUIRE(pbf_create(&bad, 0, &ctx) != PBF_OK && ctx == nullptr);
    }
    REQUIRE(pbf_step(nullptr, dt) != PBF_OK);
    REQUIRE(pbf_create(&P, 0, &ctx) == PBF_OK);
    REQUIRE(pbf_upload(ctx, pos.data(), nullptr, nullptr, mass, P.capacity + 1) != PBF_OK);
    REQUIRE(pbf_upload(ctx, nullptr, nullptr, nullptr, mass, 5) != PBF_OK);
    REQUIRE(pbf_solve(ctx, 1, dt) != PBF_OK);              // no grid yet
    REQUIRE(pbf_project_constraints(ctx, dt) == PBF_OK);   // no list: nothing to do

    // ---- a viscous run with per-particle masses, staged calls, every read-back
    const pbf_collider cols[5] = {plane(0, 1, 0, 0.f), plane(1, 0, 0, 0.f), plane(-1, 0, 0, -1.5f * nx * d),
                                  plane(0, 0, 1, 0.f), plane(0, 0, -1, -1.0f * nz * d)};
    REQUIRE(pbf_upload(ctx, pos.data(), vel.data(), pm.data(), 0.f, n) == PBF_OK);
    REQUIRE(pbf_set_colliders(ctx, cols, 5) == PBF_OK);
    REQUIRE(pbf_set_colliders(ctx, cols, PBF_MAX_COLLIDERS + 1) != PBF_OK);
    for (int s = 0; s < 3; ++s) REQUIRE(pbf_step(ctx, dt) == PBF_OK);
    REQUIRE(pbf_predict(ctx, dt) == PBF_OK);
    REQUIRE(pbf_build_grid(ctx) == PBF_OK);
    REQUIRE(pbf_compute_rest_density(ctx) == PBF_OK);
    REQUIRE(pbf_solve_lambda(ctx, dt) == PBF_OK);
    REQUIRE(pbf_solve_delta(ctx, dt) == PBF_OK);
    REQUIRE(pbf_solve(ctx, 2, dt) == PBF_OK);
    {
        std::vector<float> dens(n), C(n), lam(n), p2(3 * (size_t)n), v2(3 * (size_t)n), q2(3 * (size_t)n);
        std::vector<int32_t> cxyz(3 * (size_t)n);
        std::vector<uint32_t> ckey(n), sid(n), cstart(32 * 32 * 32 + 1), cnt(n), nb((size_t)n * 64);
        REQUIRE(pbf_get_scalars(ctx, dens.data(), C.data(), lam.data()) == PBF_OK);
        REQUIRE(pbf_get_cells(ctx, cxyz.data(), ckey.data(), sid.data(), cstart.data()) == PBF_OK);
        REQUIRE(cstart[32 * 32 * 32] == n);
        REQUIRE(pbf_get_neighbors(ctx, cnt.data(), nb.data(), 64) == PBF_OK);
        for (uint32_t i = 0; i < n; ++i) REQUIRE(cnt[i] <= 64 && sid[i] < n);
        REQUIRE(pbf_finalize(ctx, dt) == PBF_OK);
        REQUIRE(pbf_download(ctx, p2.data(), v2.data(), q2.data()) == PBF_OK);
        pbf_stats st;
        REQUIRE(pbf_get_stats(ctx, &st) == PBF_OK);
        REQUIRE(st.nonfinite == 0);
        float rho0 = 0.f;
        REQUIRE(pbf_get_rest_density(ctx, &rho0) == PBF_OK && rho0 == 1000.f);
        REQUIRE(pbf_num_particles(ctx) == n);
    }
    for (uint32_t mask : {1u, 3u, 5u, 7u}) {
        REQUIRE(pbf_set_row_formats(ctx, mask) == PBF_OK);
        REQUIRE(pbf_step(ctx, dt) == PBF_OK);
    }
    REQUIRE(pbf_set_row_formats(ctx, 0x100u) != PBF_OK);   // unknown bit
    REQUIRE(pbf_set_row_formats(ctx, PBF_ROWS_ALL | PBF_ROWS_UNSTAGED) == PBF_OK);
    REQUIRE(pbf_step(ctx, dt) == PBF_OK);
    REQUIRE(pbf_set_row_formats(ctx, PBF_ROWS_ALL) == PBF_OK);
    REQUIRE(pbf_sync(ctx) == PBF_OK);

    // ---- snapshot round trip, garbage file
    REQUIRE(pbf_snapshot_save(ctx, "/tmp/pbf_san.snap") == PBF_OK);
    REQUIRE(pbf_snapshot_load(ctx, "/tmp/pbf_san.snap") == PBF_OK);
    {
        FILE* f = fopen("/tmp/pbf_san.bad", "wb");
        const char junk[40] = "PBFSNAP1 this is not a snapshot";
        fwrite(junk, 1, sizeof junk, f);
        fclose(f);
        REQUIRE(pbf_snapshot_load(ctx, "/tmp/pbf_san.bad") != PBF_OK);
        REQUIRE(pbf_snapshot_load(ctx, "/tmp/does/not/exist") != PBF_OK);
    }
    REQUIRE(pbf_step(ctx, dt) == PBF_OK);

    // ---- boundary particles: a floor of wall particles under the block
    {
        std::vector<float> wall;
        for (int i = -2; i < nx + 2; ++i)
            for (int l = -2; l < nz + 2; ++l)
                for (int j = 1; j <= 2; ++j) {
                    wall.push_back(0.5f * d + d * i);
                    wall.push_back(-d * j + 0.5f * d);
                    wall.push_back(0.5f * d + d * l);
                }
        const uint32_t nb = (uint32_t)(wall.size() / 3);
        REQUIRE(pbf_set_boundary(ctx, wall.data(), nullptr, mass, nb, 0) == PBF_OK);
        REQUIRE(pbf_num_boundary_particles(ctx) == nb);
        for (int s = 0; s < 2; ++s) REQUIRE(pbf_step(ctx, dt) == PBF_OK);
        std::vector<float> vol(nb), bd(nb), bl(nb);
        REQUIRE(pbf_get_boundary(ctx, vol.data(), bd.data(), bl.data()) == PBF_OK);
        std::vector<uint32_t> fc(n), fb((size_t)n * 64), bc(nb), bf((size_t)nb * 64);
        REQUIRE(pbf_get_boundary_neighbors(ctx, fc.data(), fb.data(), 64, bc.data(), bf.data(), 64) == PBF_OK);
        REQUIRE(pbf_set_boundary(ctx, nullptr, nullptr, mass, 0, 0) == PBF_OK);   // removes it
    }

    // ---- distance constraints: a chain (as many levels as links), a star (one level per spoke), random pairs
    {
        std::vector<pbf_distance_constraint> list;
        for (uint32_t k = 0; k + 1 < 200; ++k) list.push_back({k, k + 1, d, 0.9f, 1e-6f});
        for (uint32_t k = 1; k < 40; ++k) list.push_back({300u, 300u + k, 2.f * d, 1.5f, 0.f});   // stiffness clamps to 1
        for (int k = 0; k < 2000; ++k) {
            uint32_t a = (uint32_t)(rnd() * n) % n, b = (uint32_t)(rnd() * n) % n;
            if (a != b) list.push_back({a, b, 3.f * d, 0.2f, 1e-4f});
        }
        for (int mode : {PBF_CONSTRAINTS_PBD, PBF_CONSTRAINTS_XPBD}) {
            REQUIRE(pbf_set_distance_constraints(ctx, list.data(), (uint32_t)list.size(), mode) == PBF_OK);
            for (int s = 0; s < 2; ++s) REQUIRE(pbf_step(ctx, dt) == PBF_OK);
            std::vector<float> lam(list.size());
            uint32_t levels = 0;
            REQUIRE(pbf_get_constraint_lambdas(ctx, lam.data(), &levels) == PBF_OK && levels >= 199);
        }
        pbf_distance_constraint self = {7u, 7u, d, 1.f, 0.f}, far_ = {1u, P.capacity + 5u, d, 1.f, 0.f};
        REQUIRE(pbf_set_distance_constraints(ctx, &self, 1, PBF_CONSTRAINTS_PBD) != PBF_OK);
        REQUIRE(pbf_set_distance_constraints(ctx, &far_, 1, PBF_CONSTRAINTS_PBD) != PBF_OK);
        REQUIRE(pbf_set_distance_constraints(ctx, list.data(), 5, 99) != PBF_OK);
        REQUIRE(pbf_set_distance_constraints(ctx, nullptr, 0, PBF_CONSTRAINTS_PBD) == PBF_OK);
    }

    // ---- the granular step
    {
        const pbf_dem_params dem = {0.5f * d, 1.0f, 0.75f, 3};
        for (int s = 0; s < 3; ++s) REQUIRE(pbf_step_dem(ctx, dt, &dem) == PBF_OK);
        REQUIRE(pbf_step_dem(ctx, dt, nullptr) != PBF_OK);
        REQUIRE(pbf_sync(ctx) == PBF_OK);
    }

    // ---- device-pointer entry points: export, the no-allocation cell build
    {
        float* dxyz = nullptr;
        uint32_t *dkey = nullptr, *did = nullptr, *dstart = nullptr;
        const int32_t gd[3] = {32, 32, 32}, go[3] = {-1, -1, -1};
        REQUIRE(hipMalloc((void**)&dxyz, (size_t)n * 12) == hipSuccess);
        REQUIRE(hipMalloc((void**)&dkey, (size_t)n * 4) == hipSuccess);
        REQUIRE(hipMalloc((void**)&did, (size_t)n * 4) == hipSuccess);
        REQUIRE(hipMalloc((void**)&dstart, (size_t)(32 * 32 * 32 + 1) * 4) == hipSuccess);
        REQUIRE(pbf_export_positions(ctx, dxyz) == PBF_OK);
        REQUIRE(pbf_sync(ctx) == PBF_OK);
        REQUIRE(pbf_build_cells_noalloc(dxyz, n, gd, go, h, dkey, did, dstart, 0, nullptr) == PBF_OK);
        REQUIRE(pbf_build_cells_noalloc(dxyz, PBF_NOALLOC_MAX_PARTICLES + 1, gd, go, h, dkey, did, dstart, 0, nullptr) != PBF_OK);
        REQUIRE(pbf_build_cells_noalloc(nullptr, n, gd, go, h, dkey, did, dstart, 0, nullptr) != PBF_OK);
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        std::vector<uint32_t> hs(32 * 32 * 32 + 1);
        REQUIRE(hipMemcpy(hs.data(), dstart, hs.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(hs.back() == n);
        (void)hipFree(dxyz); (void)hipFree(dkey); (void)hipFree(did); (void)hipFree(dstart);
    }

    // ---- profiling spans
    {
        REQUIRE(pbf_profile_enable(ctx, 1) == PBF_OK);
        REQUIRE(pbf_profile_reset(ctx) == PBF_OK);
        for (int s = 0; s < 3; ++s) REQUIRE(pbf_step(ctx, dt) == PBF_OK);
        std::vector<double> ms(pbf_num_phases());
        std::vector<uint64_t> calls(pbf_num_phases());
        REQUIRE(pbf_profile_read(ctx, ms.data(), calls.data(), pbf_num_phases()) == PBF_OK);
        REQUIRE(pbf_profile_read(ctx, ms.data(), calls.data(), 2) == PBF_OK);   // a short caller array
        for (uint32_t i = 0; i < pbf_num_phases(); ++i) REQUIRE(pbf_phase_name(i) != nullptr);
        REQUIRE(pbf_profile_enable(ctx, 0) == PBF_OK);
    }

    // ---- slab entry points called out of order
    {
        uint32_t counts[6] = {0, 0, 0, 0, 0, 0};
        REQUIRE(pbf_slab_split_done(ctx, counts) != PBF_OK);
        REQUIRE(pbf_slab_step(ctx, dt) != PBF_OK);
        REQUIRE(pbf_slab_set_bounds(ctx, 5, 5, 2) != PBF_OK);
        REQUIRE(pbf_slab_set_bounds(ctx, 0, 8, 0) != PBF_OK);
    }
    REQUIRE(pbf_upload(ctx, nullptr, nullptr, nullptr, mass, 0) == PBF_OK);   // empty
    REQUIRE(pbf_step(ctx, dt) == PBF_OK);
    REQUIRE(pbf_destroy(ctx) == PBF_OK);
    ctx = nullptr;
    printf("san_driver: ok\n");
    return 0;
}
