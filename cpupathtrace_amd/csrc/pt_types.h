// pt_types.h -- device-side layouts shared by the host code and the kernels of libpathtrace_hip.so.
//
// HBM layout of a scene (all arrays 16-byte aligned, indices 32-bit):
//   pairs     float4[4 * n_pairs]   one 64-byte record per INNER node of the reference's BVH: the boxes of its two children
//                                   and a 32-bit reference for each child.  Records are in breadth-first order, so the top
//                                   of the tree has the lowest indices.
//                                     q0 = (L.lo.x, L.lo.y, L.lo.z, L.hi.x)
//                                     q1 = (L.hi.y, L.hi.z, R.lo.x, R.lo.y)
//                                     q2 = (R.lo.z, R.hi.x, R.hi.y, R.hi.z)
//                                     q3 = (bits L.ref, bits R.ref, 0, 0)
//   tris      float4[4 * (n_tris + 1 + n_spheres)]  the LEAF records of the traversal, 64 bytes each (the stride of a pair record: a walk fetches
//                                   the record it stands on with one address computation whatever it is).  Triangle t at record t, for Moeller-Trumbore:
//                                     q0 = (a.x, a.y, a.z, ab.x)  q1 = (ab.y, ab.z, ac.x, ac.y)  q2 = (ac.z, bits material, bits (obj | cull << 31), 0)  q3 = 0
//                                   then one spare record, then sphere i at record n_tris + 1 + i: q0 = (origin.xyz, radius), rest 0
//                                   ab = b - a and ac = c - a are the fp32 differences the reference forms on every call
//                                   (src/scene/object.cpp:127-128,149-150), formed once on the host.
//   tri_shade float4[8 * n_tris]    128-byte (one HBM line) record per triangle for the shading kernel: the three words of `tris`
//                                   followed by the vertex normals (na.xyz, nb.x) (nb.yz, nc.xy) (nc.z, 0, 0, 0) and padding
//   spheres   float4[n_spheres]     (origin.xyz, radius);  sph_meta uint2[n_spheres] = (material, obj)
//   materials float4[4 * n_mat]     diffuse, specular, emission, (ior, bits bsdf, bits one_way, 0)
//   lights    float4[2 * n_lights]  (pos.xyz, 0), spectrum rgba
//   emis      float4[4 * n_emis]    one record per emissive object in registration order (scene.cpp:183-208):
//                                     triangle: (a.xyz, b.x) (b.yz, c.xy) (c.z, bits ref, bits cull, bits material) (emission rgba)
//                                     sphere:   (origin.xyz, radius) (0) (0, bits ref, 0, bits material) (emission rgba)
//   emis_cdf  float[n_emis]         normalised inclusive prefix sums (scene.cpp:169-180)
//
// Child / object reference (32 bits): bit 31 = leaf; leaf: bit 30 = sphere; PT_REF_NONE marks "no node".
// The references INSIDE the tree (the two of every pair record, root_ref, and therefore every hit the traversal reports) carry in their low
// 30 bits the index of the referenced record in ONE array `recs` = [leaf records `tris` | pair records `pairs`]: triangle t = t,
// sphere i = n_tris + 1 + i, inner node with pair slot p = pair_base + p -- so a walk fetches the record it stands on with one shift
// and one add whatever it is.  (pt_scene_create builds the two arrays with local indices and links them: pt_link_records.)
// Consumers of a hit: triangle index = low bits; sphere index = low bits - (n_tris + 1).
#ifndef PT_TYPES_H
#define PT_TYPES_H

#include <stdint.h>

#define PT_REF_LEAF 0x80000000u
#define PT_REF_SPHERE 0x40000000u
#define PT_REF_INDEX 0x3fffffffu
#define PT_REF_NONE 0xffffffffu /* also: a leaf holding the NullObject of an empty scene */

#define PT_TRI_QUADS 4      /* float4 per record of `tris` */
#define PT_MAX_NEE 32       /* light samples per path vertex: point lights + object samples (bits of the visibility mask of a slot) */
#define PT_MAX_CANDIDATES 8 /* closed candidates of the per-pixel estimator (worker.cpp:183-185) */
#define PT_MAX_DEPTH 128    /* deepest supported BVH */

struct PtDevScene {
    const float4 *recs;  /* leaf records, then pair records: what the tree's references index */
    const float4 *pairs; /* = recs + 4 * pair_base */
    const float4 *tris;  /* = recs */
    const float4 *tri_shade;
    const float4 *spheres;
    const uint2 *sph_meta;
    const float4 *materials;
    const float4 *lights;
    const float4 *emis;
    const float *emis_cdf;
    float root_lo[3];
    float root_hi[3];
    uint32_t root_ref; /* reference of the root node (a leaf when the scene has one object) */
    uint32_t pair_base; /* record index of pair slot 0: n_tris + 1 + n_spheres rounded up to even (siblings share a 128-byte line) */
    uint32_t n_pairs;
    uint32_t n_tris;
    uint32_t n_spheres;
    uint32_t n_lights;
    uint32_t n_emis;
    uint32_t n_materials;
    uint32_t n_object_samples; /* min(2 + int(log10(E + 1)), E), scene.cpp:226 */
    uint32_t n_lds_pairs;      /* small scenes: all pair records are staged in LDS by the path kernel (else 0) */
    uint32_t n_lds_tris;       /* small scenes: all triangle records are staged in LDS (else 0) */
};

// Derived camera state, Camera::Camera (src/camera.cpp:53-76)
struct PtDevCamera {
    float origin[3];
    float forward[3];
    float up[3];
    float right[3];
    float aperture_width_half;
    float aperture_height_half;
    int32_t aperture_kind;
    float hex_ratio;
    float focal_plane_dist;
};

struct PtDevOptions {
    int32_t image_width;
    int32_t image_height;
    int32_t min_sample_count;
    int32_t max_sample_count;
    float epsilon;
    float pixel_width;  /* 1 / image_width  (worker.cpp:27) */
    float pixel_height; /* 1 / image_height (worker.cpp:28) */
    int32_t stats_sample_count;    /* worker.cpp:158 */
    int32_t candidate_batch_count; /* worker.cpp:159 */
    int32_t check_sample_count;    /* worker.cpp:161-164 */
};

// Per-pixel adaptive estimator, the locals of processItem's pixel loop (worker.cpp:172-192); one per stream slot.
struct PtEstimator {
    float pixel_value[4];
    float contribution_mean[4];
    float contribution_m2[4];
    float sample_aggregate[4];
    float candidate_mean[4];
    float candidate_m2[4];
    int32_t collected_sample_count;
    int32_t contribution_count;
    int32_t stats_sample_index;
    int32_t candidate_count;
    int32_t remaining_checks;
    int32_t n_candidates;
    int32_t pixel_sample; /* loop counter of worker.cpp:193 */
    int32_t pad;
};

struct PtCandidate {
    float mean[4];
    float m2[4];
    int32_t count;
    int32_t pad[3];
};

// What a launch of the path kernel reports besides the per-wave work counters: streams that rendered their whole rectangle
// (the host checks it against the number of streams when statistics are read back).
struct PtDevCounters {
    unsigned long long streams_done;
    unsigned long long pad[7];
};

#endif
