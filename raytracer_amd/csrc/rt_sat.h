// rt_sat.h -- the build-time primitives of the octree, shared by the host builder and the device builder so
// that both evaluate literally the same expressions: triangle/box overlap (13-axis separating-axis test,
// src/collision_detection.h:89-138 of the reference), the stored face normal (src/triangle.h:28-32) and the
// boxes of a node's eight octants (src/ADS/Octree/octree.cc:81-97).
#pragma once

#include "rt_math.h"

namespace rt {

constexpr float kAabbEps = 1e-6f;  // octree.cc:30

// src/triangle.h:28-32: cross(b - a, c - b), normalised with the guarded division.
RT_HD f3 face_normal3(f3 a, f3 b, f3 c) { return normalize(cross(b - a, c - b)); }

// std::min({a,b,c}) / std::max({a,b,c}) return the first smallest / largest element.
RT_HD float sat_min3(float a, float b, float c) {
    float m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return m;
}
RT_HD float sat_max3(float a, float b, float c) {
    float m = a;
    if (m < b) m = b;
    if (m < c) m = c;
    return m;
}

RT_HD bool sat_axis_separates(f3 axis, f3 v0, f3 v1, f3 v2, f3 half) {
    const float p0 = dot(v0, axis), p1 = dot(v1, axis), p2 = dot(v2, axis);
    const float r = half.x * fabsf(axis.x) + half.y * fabsf(axis.y) + half.z * fabsf(axis.z);
    return sat_min3(p0, p1, p2) > r || sat_max3(p0, p1, p2) < -r;
}

// collision_detection.h:89-138.  `normal` is the triangle's stored, normalised face normal.
RT_HD bool sat_tri_box(f3 a, f3 b, f3 c, f3 normal, f3 bmin, f3 bmax) {
    const f3 center = (bmin + bmax) * 0.5f;
    const f3 half = (bmax - bmin) * 0.5f;
    const f3 v0 = a - center, v1 = b - center, v2 = c - center;
    const f3 e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;
    // 1. box axes
    if (sat_min3(v0.x, v1.x, v2.x) > half.x || sat_max3(v0.x, v1.x, v2.x) < -half.x) return false;
    if (sat_min3(v0.y, v1.y, v2.y) > half.y || sat_max3(v0.y, v1.y, v2.y) < -half.y) return false;
    if (sat_min3(v0.z, v1.z, v2.z) > half.z || sat_max3(v0.z, v1.z, v2.z) < -half.z) return false;
    // 2. triangle plane
    const float p0 = dot(v0, normal);
    const float r = half.x * fabsf(normal.x) + half.y * fabsf(normal.y) + half.z * fabsf(normal.z);
    if (fabsf(p0) > r) return false;
    // 3. nine edge cross products, in the reference's order
    if (sat_axis_separates(mk(0, -e0.z, e0.y), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(0, -e1.z, e1.y), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(0, -e2.z, e2.y), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(e0.z, 0, -e0.x), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(e1.z, 0, -e1.x), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(e2.z, 0, -e2.x), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(-e0.y, e0.x, 0), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(-e1.y, e1.x, 0), v0, v1, v2, half)) return false;
    if (sat_axis_separates(mk(-e2.y, e2.x, 0), v0, v1, v2, half)) return false;
    return true;
}

// The per-node terms of octree.cc:81-97 and the box of octant o = (bx<<2)|(by<<1)|bz built from them.
struct OctantFrame {
    f3 bmin, half, min_eps, max_eps;
};
RT_HD OctantFrame octant_frame(f3 bmin, f3 bmax) {
    OctantFrame f;
    f.bmin = bmin;
    f.half = div_guard(vabs(bmax - bmin), 2.0f);
    f.min_eps = vabs(bmin) * kAabbEps;
    f.max_eps = vabs(bmax) * kAabbEps;
    return f;
}
RT_HD void octant_box(const OctantFrame& f, unsigned o, f3& cmin, f3& cmax) {
    const f3 omin = mk(f.bmin.x + (f.half.x * (float)((o & 4) >> 2)), f.bmin.y + (f.half.y * (float)((o & 2) >> 1)),
                       f.bmin.z + (f.half.z * (float)(o & 1)));
    const f3 omax = omin + f.half;
    cmin = omin - f.min_eps;
    cmax = omax + f.max_eps;
}

}  // namespace rt
