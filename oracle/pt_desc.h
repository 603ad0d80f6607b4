/*
 * oracle/pt_desc.h -- flat scene / camera / job description shared by the two CPU checkers
 * (oracle/pt_oracle.c = restatement, oracle/ref_shim.cpp = compiled reference behind a C shim).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cpupathtrace_amd/ or include/ includes this file; the
 * product declares the same layouts on its own in include/pt_hip.h so that one set of numpy arrays
 * can be handed to all three implementations.
 */
#ifndef PT_ORACLE_DESC_H
#define PT_ORACLE_DESC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PTO_OBJ_TRIANGLE = 0, PTO_OBJ_SPHERE = 1 };
enum { PTO_BSDF_LAMBERTIAN = 0, PTO_BSDF_GLASS = 1, PTO_BSDF_MIRROR = 2 };
enum { PTO_APERTURE_NONE = 0, PTO_APERTURE_CIRCULAR = 1, PTO_APERTURE_HEXAGONAL = 2 };

/* One ConstantMaterial + BSDF pair (reference: include/PathTrace/scene/material.h:53-68,
 * include/PathTrace/scene/propagation.h:58-108).  64 bytes. */
typedef struct pto_material {
    float diffuse[4];
    float specular[4]; /* Material::getSpecularColor default is white, src/scene/material.cpp:15-17 */
    float emission[4];
    float ior;
    int32_t bsdf;
    int32_t one_way; /* MirrorBRDF(one_way), propagation.h:85-92 */
    int32_t pad;
} pto_material;

/* Objects in construction order (the order of the vector handed to Scene::Scene, scene.h:32).
 * obj_kind[i] says whether object i is the next triangle or the next sphere of the typed arrays. */
typedef struct pto_scene_desc {
    uint32_t n_objects;
    const uint8_t *obj_kind;
    uint32_t n_triangles;
    const float *tri_pos;          /* [n_triangles][9]  a, b, c            (object.h:126-128) */
    const float *tri_nrm;          /* [n_triangles][9]  normal_a/b/c       (object.h:129-131) */
    const uint8_t *tri_cull;       /* [n_triangles]     cull_backface      (object.h:134)     */
    const uint32_t *tri_material;  /* [n_triangles] */
    uint32_t n_spheres;
    const float *sph;              /* [n_spheres][4]    origin xyz, radius (object.h:101-103) */
    const uint32_t *sph_material;  /* [n_spheres] */
    uint32_t n_materials;
    const pto_material *materials;
    uint32_t n_point_lights;
    const float *light_pos;        /* [n_point_lights][3]  (light.h:55) */
    const float *light_spectrum;   /* [n_point_lights][4]  (light.h:56) */
} pto_scene_desc;

/* Arguments of the Camera constructor, camera.h:92,108-109. */
typedef struct pto_camera_params {
    float origin[3];
    float look_at[3];
    float up[3];
    float focal_length;
    float height;
    float aspect_ratio;
    float aperture_width;
    float aperture_height;
    int32_t aperture_kind;
    float hex_ratio;
    float focal_plane_dist;
} pto_camera_params;

/* RenderOptions, worker.h:14-31 (allow_bias is never read by the reference). */
typedef struct pto_options {
    int32_t image_width;
    int32_t image_height;
    int32_t min_sample_count;
    int32_t max_sample_count;
    float epsilon;
} pto_options;

/* One deterministic unit of work: processItem(WorkItem(job, x, y, w, h), engine) with the engine in
 * raw state `rng_state` (worker.h:44-69).  RandomEngine(seed) has state seed ^ (~seed << 32), base.h:26. */
typedef struct pto_stream {
    int32_t x, y, w, h;
    uint64_t rng_state;
} pto_stream;

#ifdef __cplusplus
}
#endif

#endif
