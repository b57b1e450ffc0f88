// The hierarchy the tree modes walk on the device.
//
// The reference's BVH4 (bvh.cuh:68-310) is a median split over object boxes; with a few very
// large primitives among many small ones (room walls around a mesh) its upper boxes span the
// whole scene and almost every ray visits them.  The closest hit only depends on
//   * each primitive's OWN box in the reference tree (its leaf slot; every ancestor box contains
//     it and the slab test is monotone in the box, so "all ancestors and the leaf box are hit"
//     is the same predicate as "the leaf box is hit"),
//   * the primitive test, and
//   * the reference's visiting rank for exact ties (consider() in pt_device.h),
// none of which involves the shape of the hierarchy above the leaves.  So the device walks a
// hierarchy of its own over the same leaf boxes: a binned surface-area-heuristic build (binary,
// one primitive per leaf; the cheapest of 31 bin boundaries on each of the three axes: against 15
// boundaries on the widest axis only, the 976-triangle scene renders in 8.74 ms instead of 9.11 and
// the 1 M-triangle frame in 345 ms instead of 360; an exact sweep over every split position for
// nodes of 32 primitives or fewer added nothing) collapsed to four children per node, inner boxes
// = exact unions of leaf boxes.  Same hits, fewer node visits.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "crt_internal.h"

namespace crt {

namespace {

struct Bounds {
    float lo[3], hi[3];
    void clear() {
        for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const Bounds& b) {
        for (int a = 0; a < 3; a++) { lo[a] = std::fmin(lo[a], b.lo[a]); hi[a] = std::fmax(hi[a], b.hi[a]); }
    }
    void grow(const float p[3]) {
        for (int a = 0; a < 3; a++) { lo[a] = std::fmin(lo[a], p[a]); hi[a] = std::fmax(hi[a], p[a]); }
    }
    double area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        if (!(x >= 0 && y >= 0 && z >= 0)) return 0.0;
        return 2.0 * (x * y + y * z + z * x);
    }
};

struct Prim {
    Bounds box;
    float centre[3];
    int32_t object;
};

struct BinaryNode {
    Bounds box;
    int32_t left = -1, right = -1;      // children, or
    int32_t object = -1;                // the primitive of a leaf
};

#ifndef CRT_ACCEL_BINS
#define CRT_ACCEL_BINS 32
#endif
constexpr int kBins = CRT_ACCEL_BINS;

inline int bin_of(const Prim& p, int axis, float lo, float scale) {
    const int b = (int)((p.centre[axis] - lo) * scale);
    return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
}

// Binary SAH build over prims[first, last); returns the index of the subtree's root in `out`.
int32_t build_binary(std::vector<Prim>& prims, int first, int last, std::vector<BinaryNode>& out) {
    struct Task { int first, last; int32_t node; };
    std::vector<Task> stack;
    const int32_t root = (int32_t)out.size();
    out.emplace_back();
    stack.push_back({first, last, root});
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        Bounds box, cbox;
        box.clear(); cbox.clear();
        for (int i = t.first; i < t.last; i++) { box.grow(prims[i].box); cbox.grow(prims[i].centre); }
        out[t.node].box = box;
        const int count = t.last - t.first;
        if (count == 1) { out[t.node].object = prims[t.first].object; continue; }
        // split: the cheapest (surface area x primitives on either side) of the bin boundaries of all three axes
        int mid = t.first + count / 2;
        double best = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_split = -1;
        for (int axis = 0; axis < 3; axis++) {
            const float extent = cbox.hi[axis] - cbox.lo[axis];
            if (!(extent > 0.0f && std::isfinite(extent))) continue;
            int bin_count[kBins] = {0};
            Bounds bin_box[kBins];
            for (int b = 0; b < kBins; b++) bin_box[b].clear();
            const float scale = (float)kBins / extent;
            for (int i = t.first; i < t.last; i++) {
                const int b = bin_of(prims[i], axis, cbox.lo[axis], scale);
                bin_count[b]++;
                bin_box[b].grow(prims[i].box);
            }
            double right_area[kBins];
            int right_count[kBins];
            Bounds acc;
            acc.clear();
            int n = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bin_box[b]);
                n += bin_count[b];
                right_area[b] = acc.area();
                right_count[b] = n;
            }
            acc.clear();
            n = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bin_box[b]);
                n += bin_count[b];
                if (n == 0 || right_count[b + 1] == 0) continue;
                const double cost = acc.area() * n + right_area[b + 1] * right_count[b + 1];
                if (cost < best) { best = cost; best_axis = axis; best_split = b; }
            }
        }
        if (best_axis >= 0) {
            const float lo = cbox.lo[best_axis], scale = (float)kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
            Prim* begin = prims.data() + t.first;
            Prim* split = std::partition(begin, prims.data() + t.last, [&](const Prim& p) { return bin_of(p, best_axis, lo, scale) <= best_split; });
            mid = t.first + (int)(split - begin);
        }
        if (mid <= t.first || mid >= t.last) mid = t.first + count / 2;     // all centres in one place
        const int32_t l = (int32_t)out.size();
        out.emplace_back();
        const int32_t r = (int32_t)out.size();
        out.emplace_back();
        out[t.node].left = l;
        out[t.node].right = r;
        stack.push_back({mid, t.last, r});
        stack.push_back({t.first, mid, l});
    }
    return root;
}

}  // namespace

// Four-wide hierarchy over the leaf slots of the reference tree `ref`.  Node 0 is the root and
// a child's number is larger than its parent's; `depth` = levels of the result.
void build_accel(const crt_bvh4& ref, std::vector<AccelNode>& nodes, int& depth) {
    std::vector<Prim> prims;
    for (int s = 0; s < ref.slot_count; s++) {
        if (ref.object_index[s] < 0) continue;
        Prim p;
        for (int a = 0; a < 3; a++) {
            p.box.lo[a] = ref.min_bounds[3 * s + a];
            p.box.hi[a] = ref.max_bounds[3 * s + a];
            p.centre[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
            if (!std::isfinite(p.centre[a])) p.centre[a] = 0.0f;
        }
        p.object = ref.object_index[s];
        prims.push_back(p);
    }
    nodes.clear();
    depth = 1;
    auto empty_node = [] {
        AccelNode n;
        for (int i = 0; i < 4; i++) {
            for (int a = 0; a < 3; a++) { n.lo[i][a] = std::numeric_limits<float>::infinity(); n.hi[i][a] = -std::numeric_limits<float>::infinity(); }
            n.child[i] = -1;
            n.object[i] = -1;
        }
        return n;
    };
    if (prims.empty()) { nodes.push_back(empty_node()); return; }
    std::vector<BinaryNode> bin;
    bin.reserve(prims.size() * 2);
    const int32_t root = build_binary(prims, 0, (int)prims.size(), bin);

    // collapse: a wide node takes a binary node's two children and keeps replacing the inner
    // child with the largest box by its two children until it has four (or only leaves)
    struct Task { int32_t binary; int32_t wide; int level; };
    std::vector<Task> stack;
    nodes.push_back(empty_node());
    stack.push_back({root, 0, 1});
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        depth = std::max(depth, t.level);
        int32_t kids[4];
        int n = 0;
        if (bin[t.binary].object >= 0) kids[n++] = t.binary;               // a tree of one primitive
        else { kids[n++] = bin[t.binary].left; kids[n++] = bin[t.binary].right; }
        while (n < 4) {
            int pick = -1;
            double largest = -1.0;
            for (int i = 0; i < n; i++) {
                if (bin[kids[i]].object >= 0) continue;
                const double a = bin[kids[i]].box.area();
                if (a > largest) { largest = a; pick = i; }
            }
            if (pick < 0) break;
            const int32_t inner = kids[pick];
            kids[pick] = bin[inner].left;
            kids[n++] = bin[inner].right;
        }
        // inner children get their node numbers now (always larger than the parent's)
        int32_t wide_of[4] = {-1, -1, -1, -1};
        for (int i = 0; i < n; i++) {
            const BinaryNode& k = bin[kids[i]];
            for (int a = 0; a < 3; a++) { nodes[t.wide].lo[i][a] = k.box.lo[a]; nodes[t.wide].hi[i][a] = k.box.hi[a]; }
            if (k.object >= 0) nodes[t.wide].object[i] = k.object;
        }
        for (int i = n - 1; i >= 0; i--) {
            if (bin[kids[i]].object >= 0) continue;
            wide_of[i] = (int32_t)nodes.size();
            nodes.push_back(empty_node());
            nodes[t.wide].child[i] = wide_of[i];
            stack.push_back({kids[i], wide_of[i], t.level + 1});
        }
    }
}

// The reference tree itself in the same form (development comparison).
void accel_from_reference(const crt_bvh4& ref, std::vector<AccelNode>& nodes) {
    const int count = ref.slot_count / 4;
    nodes.assign((size_t)count, AccelNode());
    for (int n = 0; n < count; n++)
        for (int i = 0; i < 4; i++) {
            const int s = n * 4 + i;
            for (int a = 0; a < 3; a++) { nodes[n].lo[i][a] = ref.min_bounds[3 * s + a]; nodes[n].hi[i][a] = ref.max_bounds[3 * s + a]; }
            nodes[n].child[i] = ref.children[s];
            nodes[n].object[i] = ref.object_index[s];
        }
}

}  // namespace crt
