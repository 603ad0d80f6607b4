// pt_bvh.h -- host-side construction of the reference's BVH topology and its breadth-first flattening.
#ifndef PT_BVH_H
#define PT_BVH_H

#include <cstdint>
#include <vector>

namespace ptb {

struct Box {
    float lo[3];
    float hi[3];
};

// One node of the reference-shaped tree (AABB, include/PathTrace/scene/bounding_box.h:24-57).  Nodes [0, n_objects) are the
// leaves in object construction order; inner nodes follow in allocation order.
struct Node {
    Box box;
    int32_t left = -1;  // inner: child node indices
    int32_t right = -1;
    int32_t obj = -1;   // leaf: object index
};

struct Tree {
    std::vector<Node> nodes;
    int32_t root = -1; // -1: empty scene
    uint32_t depth = 0; // number of levels (a single leaf has depth 1)
};

// impl::constructBVH (src/scene/scene.cpp:12-102) over one leaf per object.  `threads` > 1 builds independent subtrees in
// parallel; the resulting tree is identical for any thread count.
Tree build_reference_bvh(const std::vector<Box> &leaf_boxes, int threads);

// Leaves in Scene::registerEmissiveObjects order (depth-first, left before right; scene.cpp:183-208).
void leaves_depth_first(const Tree &tree, std::vector<int32_t> &out_objects);

// Pre-order dump used by the parity tests: obj index or -1, box.
void dump_preorder(const Tree &tree, std::vector<int32_t> &out_obj, std::vector<Box> &out_box);

struct FlatBvh {
    // 16 floats per inner node, breadth-first (layout: pt_types.h)
    std::vector<float> pairs;
    uint32_t n_pairs = 0;
    uint32_t root_ref = 0xffffffffu;
    Box root_box{};
    std::vector<uint32_t> level_begin; // first pair slot of every level of inner nodes, plus the end of the last level
};

// leaf_ref[i] = reference word of object i (PT_REF_LEAF | kind bit | typed index)
FlatBvh flatten_breadth_first(const Tree &tree, const std::vector<uint32_t> &leaf_ref, bool align_siblings);

} // namespace ptb

#endif
