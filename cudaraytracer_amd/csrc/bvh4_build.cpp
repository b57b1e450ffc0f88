// BVH4 builder producing the reference's flat layout bit for bit.
//
// Behaviour being reproduced (src/bvh.cuh:68-133 build, :210-243 flatten, :299-310 entry):
//   node(range): keep the visible objects of the range; 0 -> empty node; <= 4 -> leaf node
//   holding them in their current order; otherwise take the union box, choose its longest
//   axis, std::sort by bbox.min on that axis, cut at k*n/4 and recurse on the four parts.
//   Flattening is pre-order, a child's node number is assigned when its slot is visited,
//   leaf slots carry the object's own (padded) box and its index in the scene order.
//
// What is done differently: objects are sorted as indices over a contiguous key array
// instead of pointers to polymorphic objects, each recursion level copies only its own
// range, and leaf slots know their object index directly (the reference searches the
// object vector linearly per leaf, O(N^2)).  std::sort sees the same sequence of keys and
// the same comparison outcomes, so libstdc++'s introsort returns the same permutation.
#include <algorithm>
#include <limits>

#include "crt_internal.h"

namespace crt {

namespace {

struct Node {
    int32_t child[4];   // kind 1: object index, kind 2: index into the node pool
    uint8_t kind[4];    // 0 empty, 1 leaf object, 2 subtree
    Box box;
};

struct Builder {
    const std::vector<Box>& boxes;
    const std::vector<uint8_t>& visible;
    std::vector<Node> pool;

    Builder(const std::vector<Box>& b, const std::vector<uint8_t>& v) : boxes(b), visible(v) {}

    Box union_of(const std::vector<int32_t>& ids) const {
        Box bb = box_empty();
        for (int32_t id : ids) bb = box_union(bb, boxes[id]);
        return bb;
    }

    // objs[start, end) is this node's candidate range.
    int32_t build(const std::vector<int32_t>& objs, size_t start, size_t end) {
        std::vector<int32_t> vis;
        vis.reserve(end - start);
        for (size_t i = start; i < end; i++)
            if (visible[objs[i]]) vis.push_back(objs[i]);
        const size_t span = vis.size();

        int32_t self = (int32_t)pool.size();
        pool.emplace_back();
        Node node;
        for (int i = 0; i < 4; i++) { node.child[i] = -1; node.kind[i] = 0; }

        if (span == 0) {
            node.box = box_empty();
        } else if (span <= 4) {
            for (size_t i = 0; i < span; i++) { node.child[i] = vis[i]; node.kind[i] = 1; }
            node.box = union_of(vis);
        } else {
            const int axis = box_longest_axis(union_of(vis));
            const Box* bx = boxes.data();
            std::sort(vis.begin(), vis.end(), [bx, axis](int32_t a, int32_t b) { return bx[a].mn[axis] < bx[b].mn[axis]; });
            const size_t cut[5] = {0, 1 * span / 4, 2 * span / 4, 3 * span / 4, span};
            for (int i = 0; i < 4; i++) {
                node.child[i] = build(vis, cut[i], cut[i + 1]);
                node.kind[i] = 2;
            }
            node.box = box_empty();
            for (int i = 0; i < 4; i++) node.box = box_union(node.box, pool[node.child[i]].box);
        }
        pool[self] = node;
        return self;
    }
};

}  // namespace

int build_bvh4(const std::vector<Box>& boxes, const std::vector<uint8_t>& visible, std::vector<float>& min_bounds,
               std::vector<float>& max_bounds, std::vector<int32_t>& children, std::vector<int32_t>& object_index) {
    Builder b(boxes, visible);
    std::vector<int32_t> all(boxes.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (int32_t)i;
    const int32_t root = b.build(all, 0, all.size());

    // The pool is already in pre-order (a node is appended before its children are built and
    // children are built in slot order), which is exactly the numbering flatten_tree assigns.
    const size_t nodes = b.pool.size();
    const size_t slots = nodes * 4;
    min_bounds.assign(slots * 3, 0.0f);
    max_bounds.assign(slots * 3, 0.0f);
    children.assign(slots, -1);
    object_index.assign(slots, -1);
    const float inf = std::numeric_limits<float>::infinity();
    (void)root;
    for (size_t n = 0; n < nodes; n++) {
        const Node& node = b.pool[n];
        for (int i = 0; i < 4; i++) {
            const size_t s = n * 4 + i;
            if (node.kind[i] == 1) {
                const Box& ob = boxes[node.child[i]];
                for (int a = 0; a < 3; a++) { min_bounds[3 * s + a] = ob.mn[a]; max_bounds[3 * s + a] = ob.mx[a]; }
                object_index[s] = node.child[i];
            } else if (node.kind[i] == 2) {
                const Box& cb = b.pool[node.child[i]].box;
                for (int a = 0; a < 3; a++) { min_bounds[3 * s + a] = cb.mn[a]; max_bounds[3 * s + a] = cb.mx[a]; }
                children[s] = node.child[i];
            } else {
                for (int a = 0; a < 3; a++) { min_bounds[3 * s + a] = inf; max_bounds[3 * s + a] = -inf; }
            }
        }
    }
    return (int)slots;
}

}  // namespace crt

extern "C" int crt_bvh4_build(int n, const float* bbox_min, const float* bbox_max, const int32_t* visible,
                              int capacity_slots, float* min_bounds, float* max_bounds, int32_t* children,
                              int32_t* object_index) {
    crt::clear_error();
    if (n < 0 || (n > 0 && (!bbox_min || !bbox_max))) return crt::set_error(CRT_ERR_INVALID, "crt_bvh4_build: bad arguments");
    try {
        std::vector<crt::Box> boxes((size_t)n);
        std::vector<uint8_t> vis((size_t)n, 1);
        for (int i = 0; i < n; i++) {
            for (int a = 0; a < 3; a++) { boxes[i].mn[a] = bbox_min[3 * i + a]; boxes[i].mx[a] = bbox_max[3 * i + a]; }
            if (visible) vis[i] = visible[i] != 0;
        }
        std::vector<float> mn, mx;
        std::vector<int32_t> ch, oi;
        int slots = crt::build_bvh4(boxes, vis, mn, mx, ch, oi);
        if (min_bounds || max_bounds || children || object_index) {
            if (!min_bounds || !max_bounds || !children || !object_index)
                return crt::set_error(CRT_ERR_INVALID, "crt_bvh4_build: pass all four output arrays or none");
            if (capacity_slots < slots) return crt::set_error(CRT_ERR_INVALID, "crt_bvh4_build: output capacity too small");
            std::copy(mn.begin(), mn.end(), min_bounds);
            std::copy(mx.begin(), mx.end(), max_bounds);
            std::copy(ch.begin(), ch.end(), children);
            std::copy(oi.begin(), oi.end(), object_index);
        }
        return slots;
    } catch (const std::exception& e) {
        return crt::set_error(CRT_ERR_INVALID, std::string("crt_bvh4_build: ") + e.what());
    }
}
