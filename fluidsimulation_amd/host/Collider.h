// Collider.h -- collider value types of the headless host facade.
// Mirrors the reference's Collider family (reference Collider.h:7-109) for the two kinds
// whose particle response the reference defines (Particle.cpp:272-329): PLANE and AABB.
// Sphere/OBB exist as type tags only: the reference's sphere response reads an
// uninitialised vector (Particle.cpp:267) and its OBB response is a TODO (:312-318), so
// Simulation::AddCollider refuses them instead of inventing behaviour.
#pragma once
#include "Vec3.h"

class Collider {
public:
    enum class ColliderTypes { POINT = 0, SPHERE, AABB, OBB, PLANE };  // Collider.h:10-17
    explicit Collider(ColliderTypes type) : m_type(type) {}
    virtual ~Collider() = default;
    ColliderTypes getColliderTypes() const { return m_type; }

private:
    ColliderTypes m_type;
};

// hit iff dot(m_normal, p) <= m_d (CollisionDetection.cpp:62-67)
class PlaneCollider final : public Collider {
public:
    PlaneCollider(Vec3 plane_normal, float distance_from_origin)
        : Collider(ColliderTypes::PLANE), m_normal(plane_normal), m_d(distance_from_origin) {}
    Vec3 m_normal;
    float m_d;
};

// hit iff strictly inside (CollisionDetection.cpp:29-34)
class AABB final : public Collider {
public:
    AABB(Vec3 min, Vec3 max) : Collider(ColliderTypes::AABB), m_min(min), m_max(max) {}
    Vec3 getCenter() const { return {(m_max.x + m_min.x) * 0.5f, (m_max.y + m_min.y) * 0.5f, (m_max.z + m_min.z) * 0.5f}; }
    Vec3 m_min;
    Vec3 m_max;
};
