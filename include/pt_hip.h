/*
 * pt_hip.h -- C ABI of libpathtrace_hip.so, the MI355X (gfx950) implementation of CPUPathTrace's hot path.
 *
 * The reference has no FFI layer: its boundary for this path is the C++ API of include/PathTrace/worker.h and
 * include/PathTrace/scene/scene.h (all citations are file:line under the reference tree).  Each entry point below names
 * the reference interface it replaces; INTEGRATION.md shows the binding a maintainer adds on the reference side.
 * Plain C types only: pointers + sizes in, status code out (0 = PT_OK); pt_last_error() describes the last failure of the
 * calling thread.  The caller owns every host buffer; the library owns device memory behind the opaque pt_scene handle.
 * There is no CPU fallback: without a usable HIP device every compute entry point fails with PT_ERR_NO_DEVICE.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID 1     /* bad argument (null pointer, negative size, index out of range) */
#define PT_ERR_NO_DEVICE 2   /* no HIP device / device index out of range */
#define PT_ERR_HIP 3         /* a HIP runtime call failed */
#define PT_ERR_UNSUPPORTED 4 /* scene outside what the kernels cover (see pt_scene_create) */
#define PT_ERR_NOMEM 5

enum { PT_OBJ_TRIANGLE = 0, PT_OBJ_SPHERE = 1 };
enum { PT_BSDF_LAMBERTIAN = 0, PT_BSDF_GLASS = 1, PT_BSDF_MIRROR = 2 };
enum { PT_APERTURE_NONE = 0, PT_APERTURE_CIRCULAR = 1, PT_APERTURE_HEXAGONAL = 2 };
#define PT_NO_MATERIAL 0xFFFFFFFFu /* object keeps the default handler: white Lambertian (src/scene/object.cpp:9-11,32) */

/* ConstantMaterial + BSDF behind one ConstantMaterialHandler (scene/material.h:53-68, scene/propagation.h:58-108,
 * scene/object.h:26-40).  specular is Material::getSpecularColor (white unless a subclass overrides it). */
typedef struct pt_material {
    float diffuse[4];
    float specular[4];
    float emission[4];
    float ior;
    int32_t bsdf;    /* PT_BSDF_* */
    int32_t one_way; /* MirrorBRDF(one_way) */
    int32_t pad;
} pt_material;

/* The std::vector<std::unique_ptr<Object>> and std::vector<std::unique_ptr<LightSource>> handed to Scene::Scene
 * (scene/scene.h:32), flattened.  Objects keep their construction order: obj_kind[i] tells whether object i is the next
 * entry of the triangle arrays or of the sphere arrays.  Object indices reported by pt_intersect_batch are positions in
 * this order. */
typedef struct pt_scene_desc {
    uint32_t n_objects;
    const uint8_t *obj_kind;      /* [n_objects] PT_OBJ_* */
    uint32_t n_triangles;
    const float *tri_pos;         /* [n_triangles][9] a, b, c                  (scene/object.h:126-128) */
    const float *tri_nrm;         /* [n_triangles][9] normal_a, _b, _c; NULL = face normals (object.cpp:118-124) */
    const uint8_t *tri_cull;      /* [n_triangles] cull_backface              (scene/object.h:134) */
    const uint32_t *tri_material; /* [n_triangles] index into materials or PT_NO_MATERIAL */
    uint32_t n_spheres;
    const float *sph;             /* [n_spheres][4] origin xyz, radius        (scene/object.h:101-103) */
    const uint32_t *sph_material; /* [n_spheres] */
    uint32_t n_materials;
    const pt_material *materials;
    uint32_t n_point_lights;
    const float *light_pos;       /* [n_point_lights][3] PointLightSource::pos      (scene/light.h:55) */
    const float *light_spectrum;  /* [n_point_lights][4] PointLightSource::spectrum (scene/light.h:56) */
} pt_scene_desc;

/* Arguments of Camera::Camera (camera.h:92,108-109). */
typedef struct pt_camera_params {
    float origin[3];
    float look_at[3];
    float up[3];
    float focal_length;
    float height;
    float aspect_ratio;
    float aperture_width;
    float aperture_height;
    int32_t aperture_kind; /* PT_APERTURE_* : nullptr / CircularApertureSampler / HexagonalApertureSampler(hex_ratio) */
    float hex_ratio;
    float focal_plane_dist;
} pt_camera_params;

/* RenderOptions (worker.h:14-31); allow_bias is never read by the reference and has no field here. */
typedef struct pt_options {
    int32_t image_width;
    int32_t image_height;
    int32_t min_sample_count;
    int32_t max_sample_count;
    float epsilon;
} pt_options;

/* WorkItem (worker.h:44-62). */
typedef struct pt_tile {
    int32_t x, y, w, h;
} pt_tile;

/* One processItem(WorkItem(job, x, y, w, h), engine) call (worker.h:69): the pixels of the rectangle are rendered in row-major
 * order through ONE engine whose raw xorshift state is rng_state on entry (base.h:24-38). */
typedef struct pt_stream {
    int32_t x, y, w, h;
    uint64_t rng_state;
} pt_stream;

/* Work done by one render call, counted on the device by the kernel itself; the time is measured with HIP events around the launch
 * on the library's stream. */
typedef struct pt_stats {
    uint64_t samples;           /* getSample calls (worker.cpp:194) */
    uint64_t rays_traced;       /* closest-hit + shadow rays walked through the tree */
    uint64_t shadow_rays_traced;
    uint64_t node_visits;       /* inner nodes visited = pairs of AABB slab tests */
    uint64_t leaf_tests;        /* Triangle/Sphere::getIntersection calls */
    uint64_t vertices;          /* path vertices shaded */
    uint64_t launches;          /* launches of the path kernel (one per render call) */
    double kernel_ms;           /* duration of the launch */
    uint64_t wave_steps;        /* traversal steps executed by wavefronts (each serves up to 64 walks) */
    uint64_t shading_passes;    /* shading passes executed by wavefronts */
    uint64_t wavefronts;        /* wavefronts of the launch */
    uint64_t slot_rows;         /* rows of 64 stream slots per wavefront */
} pt_stats;

typedef struct pt_scene pt_scene;

/* Number of usable HIP devices (0 when there is none or the runtime cannot be initialised). */
int pt_device_count(void);

const char *pt_last_error(void);

/* Scene::Scene (src/scene/scene.cpp:153-181): builds the reference's BVH topology (impl::constructBVH, scene.cpp:12-102) -- on
 * the device, level by level (pt_build.hip), for scenes of PT_BUILD_DEVICE_MIN (1024) objects and more, by the same recursion on the host
 * for smaller ones; the two produce the same tree bit for bit -- registers emissive objects (scene.cpp:183-208) and lays everything out
 * in device arrays on `device`.
 * PT_ERR_UNSUPPORTED: more than 32 light samples per path vertex (point lights + min(2 + log10(E + 1), E) object samples: Scene::sampleLights
 * itself has no bound, scene.cpp:226,231; 32 are the bits of this library's per-vertex visibility mask),
 * or a BVH deeper than 128 levels.
 * Thread safety: calls on DIFFERENT scenes may run concurrently; render and intersection calls on the SAME scene are serialised inside
 * the library (one workspace per scene), so callers that run processItem from several threads on one Scene -- as the reference's
 * doWork does (src/worker.cpp:328-362) -- are correct, they just do not overlap on the device. */
int pt_scene_create(int device, const pt_scene_desc *desc, pt_scene **out);
void pt_scene_destroy(pt_scene *scene);

/* Introspection for tests: node count of the reference-topology BVH (2 * n_objects - 1), its depth, emissive object count. */
int pt_scene_info(const pt_scene *scene, uint64_t *n_nodes, uint32_t *depth, uint32_t *n_emissive);
/* Emissive objects in Scene::registerEmissiveObjects order (scene.cpp:183-208) with their normalised cumulative selection
 * probabilities (scene.cpp:167-180); at most `capacity` entries are written, the count is returned in *n_written. */
int pt_scene_emissive(const pt_scene *scene, int32_t *out_obj, float *out_cdf, uint64_t capacity, uint64_t *n_written);
/* Pre-order dump of the BVH: out_obj[i] = object index of a leaf or -1 for an inner node, out_box[i] = low xyz, high xyz. */
int pt_scene_bvh_dump(const pt_scene *scene, int32_t *out_obj, float *out_box, uint64_t capacity, uint64_t *n_written);

/* Scene::getIntersection (scene/scene.h:41, src/scene/scene.cpp:210-220) for n rays (origin xyz, direction xyz each).
 * out_t < 0 means miss (then out_obj = -1); otherwise out_obj is the construction-order index of the closest object. */
int pt_intersect_batch(pt_scene *scene, const float *rays, size_t n, float *out_t, int32_t *out_obj);

/* processItem (worker.h:69, src/worker.cpp:149-326) for n independent streams, all in flight at once on the device.
 * out_image is the full row-major image (image_width * image_height * 4 floats, index (y * width + x) * 4 as image/image.h:80-89);
 * only pixels covered by a stream are written.  out_states[i] (may be NULL) receives stream i's engine state afterwards. */
int pt_render_streams(pt_scene *scene, const pt_camera_params *camera, const pt_options *options, const pt_stream *streams, size_t n,
                      float *out_image, uint64_t *out_states, pt_stats *stats);

/* processItem for ONE WorkItem, returning only the item's rectangle: out_tile is item->w * item->h * 4 floats, row-major inside the
 * rectangle (the Image<> processItem returns, worker.h:69); *out_state (may be NULL) receives the engine state afterwards.  The frame the
 * item belongs to never exists on the host. */
int pt_render_item(pt_scene *scene, const pt_camera_params *camera, const pt_options *options, const pt_stream *item, float *out_tile,
                   uint64_t *out_state, pt_stats *stats);

/* processJob (worker.h:83-84, src/worker.cpp:389-424) restricted to the given tiles: every pixel is its own 1x1 stream whose
 * engine is RandomEngine(pt_pixel_seed(base_seed, x, y)) -- the reference seeds its workers from std::random_device
 * (worker.cpp:369-382), so any seeding conforms; this one makes the image independent of tiling and of the GPU count. */
int pt_render_tiles(pt_scene *scene, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles,
                    uint64_t base_seed, float *out_image, pt_stats *stats);

/* Same, reporting progress the way processJob's progress_callback does (worker.h:75-84, src/worker.cpp:354-360): `progress(completed,
 * total, user)` is called from the CALLING thread, never concurrently, with completed = 1 .. n_tiles in increasing order, while the
 * device is still rendering (the kernel counts finished tiles in host-visible memory; the host polls).  NULL = no reporting.
 * The callback runs while the library holds the scene's render lock: it must not call back into the library with the same scene (a
 * preview through processItem would wait for itself), and it must not throw through this C interface (src/host/worker.cpp keeps a
 * C++ callback's exception and throws it again after the call). */
typedef void (*pt_progress_fn)(int completed, int total, void *user);
int pt_render_tiles_progress(pt_scene *scene, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles,
                             uint64_t base_seed, float *out_image, pt_stats *stats, pt_progress_fn progress, void *user);

/* processJob on several devices of one node -- the multi-device form of doWorkParallel (src/worker.cpp:364-387).  scenes[i] are replicas of
 * one scene created on different devices (pt_scene_create(device_i, same desc)); the tiles are dealt round-robin to the scenes (tile k to
 * scenes[k % n_scenes]; along the diagonals of the tile grid when its rows hold a multiple of n_scenes tiles, so that no device renders
 * whole columns of the frame), one host thread per scene, and only the rectangles of a device's own tiles are copied into out_image.
 * The image is identical for every n_scenes (per-pixel engines).  stats, if not NULL, is an array of n_scenes entries.  progress as in
 * pt_render_tiles_progress, counted over all devices and never concurrent -- but with n_scenes > 1 it is called from the library's
 * worker threads (as the reference calls it from its worker threads, worker.h:75-78), not from the calling thread.
 * MEASURED ONLY WITH REPLICAS ON ONE DEVICE so far (tests): no multi-GPU node was available to the build. */
int pt_render_tiles_multi(pt_scene *const *scenes, int n_scenes, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles,
                          size_t n_tiles, uint64_t base_seed, float *out_image, pt_stats *stats, pt_progress_fn progress, void *user);

/* Same, writing into DEVICE memory (e.g. a torch tensor's data_ptr) and ordered on `stream` (a hipStream_t, NULL = the
 * library's own stream followed by a synchronisation).  Used for the multi-GPU gather over RCCL.  Does not wait for the device unless
 * `stats` is given; every waiting entry point checks that the launch rendered all its streams, this one with stats or PT_VERIFY=1. */
int pt_render_tiles_device(pt_scene *scene, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles,
                           uint64_t base_seed, float *d_out_image, void *stream, pt_stats *stats);

/* The tile list processJob builds (worker.cpp:398-414): tile_size = clamp(min(w, h) / 4, 1, 32), row-major, edge tiles clipped.
 * Returns the tile count; fills at most `capacity` entries. */
size_t pt_job_tiles(int32_t image_width, int32_t image_height, pt_tile *out, size_t capacity);

uint64_t pt_pixel_seed(uint64_t base_seed, int32_t x, int32_t y);
/* RandomEngine(seed) raw state (base.h:26). */
uint64_t pt_rng_seed_to_state(uint64_t seed);

/* toneMap / gammaCorrect / postProcess (include/PathTrace/post_processing.h:14,22,30; src/post_processing.cpp:32-187) on an
 * rgba f32 frame (row-major, y * width + x as image/image.h:82), in place.  `steps` = PT_POST_TONE_MAP | PT_POST_GAMMA; both =
 * postProcess (tone mapping first).  `gamma` is gammaCorrect's argument (the reference's default is 1.8).
 * pt_post_process takes a HOST buffer (uploads, processes, downloads); pt_post_process_device works on DEVICE memory and is
 * ordered on `stream` (a hipStream_t, NULL = the default stream), which it synchronises before returning. */
#define PT_POST_TONE_MAP 1u
#define PT_POST_GAMMA 2u
int pt_post_process(int device, float *rgba, int32_t width, int32_t height, uint32_t steps, float gamma);
int pt_post_process_device(int device, float *d_rgba, int32_t width, int32_t height, uint32_t steps, float gamma, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* PT_HIP_H */
