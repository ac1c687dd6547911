// Vec3.h -- three packed floats, layout-compatible with the glm::vec3 the reference stores
// in std::vector<glm::vec3> (reference Particle.h:56-61), so host arrays pass straight
// through the C ABI as interleaved xyz triples.
#pragma once

struct Vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
static_assert(sizeof(Vec3) == 12, "Vec3 must be three packed floats");
