// Host-side axis-aligned box with the reference's construction rules (src/aabb.cuh:28-107,
// 219-223; src/interval.cuh:51-54,78): every constructor pads an axis thinner than 0.001 by
// 0.001 on both sides, unions pick a.min when a.min <= b.min.  Used by the scene builder and
// the BVH4 builder; the padding order matters for bit-exact bounds.
#ifndef CRT_HOST_BOX_H
#define CRT_HOST_BOX_H

#include <limits>
#include <math.h>

namespace crt {

struct Box {
    float mn[3], mx[3];
};

inline void box_pad(Box& b, float delta = 0.001f) {
    for (int a = 0; a < 3; a++) {
        if (b.mx[a] - b.mn[a] < delta) {
            float lo = b.mn[a] - delta, hi = b.mx[a] + delta;
            b.mn[a] = lo;
            b.mx[a] = hi;
        }
    }
}
inline Box box_empty() {
    const float inf = std::numeric_limits<float>::infinity();
    Box b;
    for (int a = 0; a < 3; a++) { b.mn[a] = inf; b.mx[a] = -inf; }
    box_pad(b);   // inf - 0.001 stays inf: kept for fidelity with aabb::empty
    return b;
}
// aabb(point a, point b)
inline Box box_from_points(const float* p, const float* q) {
    Box b;
    for (int a = 0; a < 3; a++) {
        if (p[a] <= q[a]) { b.mn[a] = p[a]; b.mx[a] = q[a]; }
        else { b.mn[a] = q[a]; b.mx[a] = p[a]; }
    }
    box_pad(b);
    return b;
}
// aabb(point a, point b, point c)
inline Box box_from_points(const float* p, const float* q, const float* r) {
    Box b;
    for (int a = 0; a < 3; a++) {
        b.mn[a] = fminf(p[a], fminf(q[a], r[a]));
        b.mx[a] = fmaxf(p[a], fmaxf(q[a], r[a]));
    }
    box_pad(b);
    return b;
}
// aabb(aabb a, aabb b)
inline Box box_union(const Box& p, const Box& q) {
    Box b;
    for (int a = 0; a < 3; a++) {
        b.mn[a] = p.mn[a] <= q.mn[a] ? p.mn[a] : q.mn[a];
        b.mx[a] = p.mx[a] >= q.mx[a] ? p.mx[a] : q.mx[a];
    }
    box_pad(b);
    return b;
}
// aabb(bbox.x + dx, bbox.y + dy, bbox.z + dz)
inline Box box_offset(const Box& p, const float* off) {
    Box b;
    for (int a = 0; a < 3; a++) { b.mn[a] = p.mn[a] + off[a]; b.mx[a] = p.mx[a] + off[a]; }
    box_pad(b);
    return b;
}
// aabb::longest_axis (aabb.cuh:163-166)
inline int box_longest_axis(const Box& b) {
    float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}

}  // namespace crt

#endif
