// pt_build.h -- scene construction ON the device: leaf records and the reference-shaped BVH, built level by level in HBM.
#ifndef PT_BUILD_H
#define PT_BUILD_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

// Raw object arrays, already in device memory (uploaded as the caller handed them over, include/pt_hip.h: pt_scene_desc).
struct PtBuildInput {
    uint32_t n_objects = 0;
    uint32_t n_triangles = 0;
    uint32_t n_spheres = 0;
    const float *tri_pos = nullptr;         // [n_triangles][9]
    const float *tri_nrm = nullptr;         // [n_triangles][9] or nullptr: face normals
    const uint8_t *tri_cull = nullptr;      // [n_triangles]
    const uint32_t *tri_material = nullptr; // [n_triangles]
    const uint32_t *tri_obj = nullptr;      // [n_triangles] object index (construction order)
    const float *sph = nullptr;             // [n_spheres][4]
    const uint32_t *sph_material = nullptr; // [n_spheres]
    const uint32_t *sph_obj = nullptr;      // [n_spheres]
    bool align_siblings = true;
};

// Device buffers the builder fills.  tris / tri_shade / spheres / sph_meta are allocated by the caller; pairs and dfs by the builder
// (hipMalloc; ownership passes to the caller).
struct PtBuildOutput {
    float4 *tris = nullptr;      // PT_TRI_QUADS per triangle  (pt_types.h)
    float4 *tri_shade = nullptr; // 8 per triangle
    float4 *spheres = nullptr;   // 1 per sphere
    uint2 *sph_meta = nullptr;   // 1 per sphere
    float4 *pairs = nullptr;     // 4 per pair slot
    uint32_t *dfs = nullptr;     // object indices, depth-first left-to-right (Scene::registerEmissiveObjects order)
    uint32_t n_pairs = 0;
    uint32_t depth = 0;
    uint32_t root_ref = 0xffffffffu;
    float root_lo[3] = {0, 0, 0};
    float root_hi[3] = {0, 0, 0};
    float build_ms = 0.0F; // device time of the tree construction (HIP events), records excluded
    std::vector<uint32_t> level_begin; // first pair slot of every level of inner nodes, plus the end of the last level
};

// Returns hipSuccess or the first HIP error; `error_text` (optional) receives a static description of the failing step.
// Requires n_objects >= 2.
hipError_t pt_build_scene_device(hipStream_t stream, const PtBuildInput &in, PtBuildOutput &out, const char **error_text);


// Turns the local indices of the tree's references into record indices (pt_types.h): recs = [leaf records | n_pairs pair records from
// record pair_base on]; a reference to pair slot p becomes pair_base + p, one to sphere i becomes sphere_base + i, triangles keep theirs.
hipError_t pt_link_records(hipStream_t stream, float4 *recs, uint32_t pair_base, uint32_t n_pairs, uint32_t sphere_base);

// Positions of the objects whose bit is set in `mask_bits` (host array, one bit per object) within the depth-first leaf order
// `dfs` (device).  On return `ordered` lists those objects in depth-first order.
hipError_t pt_build_order_subset(hipStream_t stream, const uint32_t *dfs, uint32_t n_objects, const std::vector<uint32_t> &mask_bits, uint32_t n_selected,
                                 std::vector<uint32_t> &ordered);

#endif
