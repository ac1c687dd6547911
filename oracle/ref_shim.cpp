// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY, in-container only.
//
// extern "C" doors into the reference's OWN sources, compiled where they lie under
// /root/reference (never copied): SPHKernel.h (header-only), CollisionDetection.cpp and
// Collider.cpp.  These are the only pieces of the reference's CPU path that build with
// this image's toolchain without stand-ins: Particle.cpp / NeighborSearch.cpp /
// Simulation.cpp need cuda_runtime.h, GL and ImGui headers the image lacks (treated as
// unbuildable).  The only accommodation is -D__forceinline=inline for the
// MSVC keyword at Collider.h:26.  Output goes to oracle/_ref/ (git-ignored).
#include "SPHKernel.h"
#include "CollisionDetection.h"

extern "C" {

float ref_poly6_w(float distance, float h) { return SPHKernel::Poly6_W(distance, h); }

void ref_poly6_w_gradient(const float* diff, float distance, float h, float* out) {
    glm::vec3 g = SPHKernel::Poly6_W_Gradient(glm::vec3(diff[0], diff[1], diff[2]), distance, h);
    out[0] = g.x; out[1] = g.y; out[2] = g.z;
}

float ref_spiky_w(float distance, float h) { return SPHKernel::Spiky_W(distance, h); }

void ref_spiky_w_gradient(const float* diff, float distance, float h, float* out) {
    glm::vec3 g = SPHKernel::Spiky_W_Gradient(glm::vec3(diff[0], diff[1], diff[2]), distance, h);
    out[0] = g.x; out[1] = g.y; out[2] = g.z;
}

int ref_point_plane(const float* p, const float* n, float d) {
    PlaneCollider plane(glm::vec3(n[0], n[1], n[2]), d);
    return CollisionDetection::PointPlaneIntersection(glm::vec3(p[0], p[1], p[2]), &plane) ? 1 : 0;
}

int ref_point_aabb(const float* p, const float* mn, const float* mx) {
    AABB box(glm::vec3(mn[0], mn[1], mn[2]), glm::vec3(mx[0], mx[1], mx[2]));
    return CollisionDetection::PointAABBIntersection(glm::vec3(p[0], p[1], p[2]), &box) ? 1 : 0;
}

// glm helpers the step loop leans on, evaluated by the vendored glm itself
float ref_glm_distance(const float* a, const float* b) {
    return glm::distance(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}

float ref_glm_dot(const float* a, const float* b) {
    return glm::dot(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}

double ref_glm_pow(double x, int e) { return glm::pow(x, e); }

}  // extern "C"
