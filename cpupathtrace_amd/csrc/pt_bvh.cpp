// pt_bvh.cpp -- builds the BVH the reference would build, so closest-hit ties and traversal order are the reference's.
//
// The reference's Scene constructor wraps every object in a leaf AABB and calls impl::constructBVH
// (src/scene/scene.cpp:12-102,153-162).  Which leaf a ray reports when two hits tie, and which subtrees are pruned, depend on
// that tree's exact shape, so the device traverses the same tree.  The algorithm per node:
//   1. per axis, the median of the boxes' LOW coordinates: the element nth_element leaves at index n/2 - 1   (:24-36)
//   2. per axis, the summed surface area of the two groups {low <= median} / {low > median}, fp32, in input order (:38-62)
//   3. the axis with the smallest sum wins, the first one on ties                                           (:64-72)
//   4. stable partition in input order                                                                       (:74-87)
//   5. while left has more than twice right's elements (and more than one), move left's LAST element to right's end (:89-94)
//   6. recurse; the parent box is the union of the two child boxes                                           (bounding_box.cpp:8-10,18-24)
// This is host code on the scene-build path, not on the per-sample path; it runs once per scene and is parallel over
// subtrees.
#include "pt_bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <deque>
#include <future>
#include <limits>

namespace ptb {

namespace {

// std::min / std::max on floats, spelled out so the order of operands is the reference's
inline float fmin_std(float a, float b) {
    return (b < a) ? b : a;
}
inline float fmax_std(float a, float b) {
    return (a < b) ? b : a;
}

struct Builder {
    std::vector<Node> &nodes;
    std::atomic<int32_t> next_inner;
    std::atomic<int> spare_threads;

    Builder(std::vector<Node> &n, int32_t first_inner, int threads) : nodes(n), next_inner(first_inner), spare_threads(threads - 1) {}

    // returns (node index, depth)
    std::pair<int32_t, uint32_t> build(std::vector<int32_t> ids) {
        const size_t n = ids.size();
        if(n == 1) {
            return {ids[0], 1U};
        }

        float medians[3];
        {
            std::vector<float> coords(n);
            for(int dim = 0; dim < 3; dim++) {
                for(size_t i = 0; i < n; i++) {
                    coords[i] = nodes[ids[i]].box.lo[dim];
                }
                auto nth = coords.begin() + (static_cast<long>(n) / 2 - 1);
                std::nth_element(coords.begin(), nth, coords.end());
                medians[dim] = *nth;
            }
        }

        const float inf = std::numeric_limits<float>::infinity();
        float surface_areas[3];
        for(int dim = 0; dim < 3; dim++) {
            float clo[2][3], chi[2][3];
            for(int g = 0; g < 2; g++) {
                for(int k = 0; k < 3; k++) {
                    clo[g][k] = inf;
                    chi[g][k] = -inf;
                }
            }
            for(size_t i = 0; i < n; i++) {
                const Box &b = nodes[ids[i]].box;
                const int g = b.lo[dim] <= medians[dim] ? 0 : 1;
                for(int k = 0; k < 3; k++) {
                    clo[g][k] = fmin_std(clo[g][k], b.lo[k]);
                    chi[g][k] = fmax_std(chi[g][k], b.hi[k]);
                }
            }
            float surface_area = 0.0F;
            for(int g = 0; g < 2; g++) {
                const float d0 = chi[g][0] - clo[g][0];
                const float d1 = chi[g][1] - clo[g][1];
                const float d2 = chi[g][2] - clo[g][2];
                surface_area += 2 * (d0 * d1 + d1 * d2 + d0 * d2);
            }
            surface_areas[dim] = surface_area;
        }

        int axis = 0;
        float min_surface = surface_areas[0];
        for(int dim = 1; dim < 3; dim++) {
            if(surface_areas[dim] < min_surface) {
                min_surface = surface_areas[dim];
                axis = dim;
            }
        }

        std::vector<int32_t> left, right;
        left.reserve(n / 2 + 1);
        right.reserve((n + 1) / 2 + 1);
        for(size_t i = 0; i < n; i++) {
            if(nodes[ids[i]].box.lo[axis] <= medians[axis]) {
                left.push_back(ids[i]);
            }
            else {
                right.push_back(ids[i]);
            }
        }
        while(left.size() > 1 && left.size() > 2 * right.size()) {
            right.push_back(left.back());
            left.pop_back();
        }
        std::vector<int32_t>().swap(ids);

        std::pair<int32_t, uint32_t> l, r;
        bool forked = false;
        if(n > 16384) {
            int spare = spare_threads.load(std::memory_order_relaxed);
            while(spare > 0 && !spare_threads.compare_exchange_weak(spare, spare - 1)) {
            }
            if(spare > 0) {
                forked = true;
                auto fut = std::async(std::launch::async, [this, &left]() { return build(std::move(left)); });
                r = build(std::move(right));
                l = fut.get();
                spare_threads.fetch_add(1);
            }
        }
        if(!forked) {
            l = build(std::move(left));
            r = build(std::move(right));
        }

        const int32_t me = next_inner.fetch_add(1);
        Node &nd = nodes[me];
        const Box &lb = nodes[l.first].box;
        const Box &rb = nodes[r.first].box;
        for(int k = 0; k < 3; k++) {
            nd.box.lo[k] = fmin_std(lb.lo[k], rb.lo[k]);
            nd.box.hi[k] = fmax_std(lb.hi[k], rb.hi[k]);
        }
        nd.left = l.first;
        nd.right = r.first;
        nd.obj = -1;
        return {me, std::max(l.second, r.second) + 1U};
    }
};

} // namespace

Tree build_reference_bvh(const std::vector<Box> &leaf_boxes, int threads) {
    Tree tree;
    const size_t n = leaf_boxes.size();
    if(n == 0) {
        return tree;
    }
    tree.nodes.resize(2 * n - 1);
    std::vector<int32_t> ids(n);
    for(size_t i = 0; i < n; i++) {
        tree.nodes[i].box = leaf_boxes[i];
        tree.nodes[i].obj = static_cast<int32_t>(i);
        ids[i] = static_cast<int32_t>(i);
    }
    Builder builder(tree.nodes, static_cast<int32_t>(n), std::max(threads, 1));
    auto result = builder.build(std::move(ids));
    tree.root = result.first;
    tree.depth = result.second;
    return tree;
}

void leaves_depth_first(const Tree &tree, std::vector<int32_t> &out_objects) {
    out_objects.clear();
    if(tree.root < 0) {
        return;
    }
    std::vector<int32_t> stack{tree.root};
    while(!stack.empty()) {
        const int32_t i = stack.back();
        stack.pop_back();
        const Node &nd = tree.nodes[i];
        if(nd.left < 0) {
            out_objects.push_back(nd.obj);
        }
        else {
            stack.push_back(nd.right);
            stack.push_back(nd.left);
        }
    }
}

void dump_preorder(const Tree &tree, std::vector<int32_t> &out_obj, std::vector<Box> &out_box) {
    out_obj.clear();
    out_box.clear();
    if(tree.root < 0) {
        return;
    }
    std::vector<int32_t> stack{tree.root};
    while(!stack.empty()) {
        const int32_t i = stack.back();
        stack.pop_back();
        const Node &nd = tree.nodes[i];
        out_box.push_back(nd.box);
        if(nd.left < 0) {
            out_obj.push_back(nd.obj);
        }
        else {
            out_obj.push_back(-1);
            stack.push_back(nd.right);
            stack.push_back(nd.left);
        }
    }
}

FlatBvh flatten_breadth_first(const Tree &tree, const std::vector<uint32_t> &leaf_ref, bool align_siblings) {
    FlatBvh flat;
    if(tree.root < 0) {
        return flat;
    }
    const Node &root = tree.nodes[tree.root];
    flat.root_box = root.box;
    if(root.left < 0) {
        flat.root_ref = leaf_ref[root.obj];
        return flat;
    }

    // Breadth-first numbering of the inner nodes.  When both children of a node are inner nodes their two records are
    // placed in ONE aligned 128-byte line (even index, odd index): HBM is fetched in 128-byte lines, a walk that enters
    // the near child very often enters the far child later, and the second record then costs no further line.  An unused
    // slot is left where the running index is odd (about one slot in eight).
    std::vector<int32_t> order; // order[slot] = node index or -1 for a padding slot
    order.reserve(tree.nodes.size() / 2 + tree.nodes.size() / 8 + 2);
    std::vector<uint32_t> pair_index(tree.nodes.size(), 0xffffffffu);
    order.push_back(tree.root);
    pair_index[tree.root] = 0;
    size_t level_end = 1; // the root is level 0
    flat.level_begin.push_back(0U);
    for(size_t head = 0; head < order.size(); head++) {
        if(head == level_end) {
            flat.level_begin.push_back(static_cast<uint32_t>(head));
            level_end = order.size();
        }
        if(order[head] < 0) {
            continue;
        }
        const Node &nd = tree.nodes[order[head]];
        const bool left_inner = tree.nodes[nd.left].left >= 0;
        const bool right_inner = tree.nodes[nd.right].left >= 0;
        if(left_inner && right_inner && (order.size() & 1U) != 0 && align_siblings) {
            order.push_back(-1);
        }
        for(int32_t child : {nd.left, nd.right}) {
            if(tree.nodes[child].left >= 0) {
                pair_index[child] = static_cast<uint32_t>(order.size());
                order.push_back(child);
            }
        }
    }

    flat.n_pairs = static_cast<uint32_t>(order.size());
    flat.level_begin.push_back(flat.n_pairs);
    flat.root_ref = 0;
    flat.pairs.assign(16 * order.size(), 0.0F);
    auto ref_of = [&](int32_t node) -> uint32_t {
        const Node &c = tree.nodes[node];
        return c.left < 0 ? leaf_ref[c.obj] : pair_index[node];
    };
    for(size_t i = 0; i < order.size(); i++) {
        if(order[i] < 0) {
            continue;
        }
        const Node &nd = tree.nodes[order[i]];
        const Box &l = tree.nodes[nd.left].box;
        const Box &r = tree.nodes[nd.right].box;
        float *q = &flat.pairs[16 * i];
        q[0] = l.lo[0];
        q[1] = l.lo[1];
        q[2] = l.lo[2];
        q[3] = l.hi[0];
        q[4] = l.hi[1];
        q[5] = l.hi[2];
        q[6] = r.lo[0];
        q[7] = r.lo[1];
        q[8] = r.lo[2];
        q[9] = r.hi[0];
        q[10] = r.hi[1];
        q[11] = r.hi[2];
        const uint32_t lr = ref_of(nd.left);
        const uint32_t rr = ref_of(nd.right);
        static_assert(sizeof(float) == sizeof(uint32_t), "float is 32 bits");
        __builtin_memcpy(&q[12], &lr, 4);
        __builtin_memcpy(&q[13], &rr, 4);
        q[14] = 0.0F;
        q[15] = 0.0F;
    }
    return flat;
}

} // namespace ptb
