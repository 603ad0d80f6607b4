/*
 * oracle/pt_oracle.h -- entry points of the plain-C CPU restatement of CPUPathTrace's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.c).  The entry points mirror oracle/ref_shim.cpp one for one
 * (same argument lists, prefix oracle_ instead of ref_) so tests can run one harness against both.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "pt_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Work counters behind the algorithmic-bytes formula of SURVEY.md 8(d). */
typedef struct oracle_counters {
    uint64_t samples;      /* getSample calls */
    uint64_t scene_queries; /* R: Scene::getIntersection calls (camera + bounce + shadow rays) */
    uint64_t aabb_tests;   /* A: AABB::getIntersection calls */
    uint64_t tri_tests;    /* T: Triangle::getIntersection calls */
    uint64_t sphere_tests;
    uint64_t vertices;     /* V: path vertices (hits of camera/bounce rays) */
    uint64_t shadow_rays;
} oracle_counters;

void oracle_rng_draws(uint64_t seed, uint64_t n, uint32_t *out);
uint64_t oracle_rng_state_after(uint64_t seed, uint64_t n_draws);
void oracle_uniform_floats(uint64_t seed, float a, float b, uint64_t n, float *out);
uint64_t oracle_bernoulli(uint64_t seed, double p, uint64_t n, uint8_t *out_flags);

void oracle_aabb_intersect(uint64_t n, const float *boxes, const float *rays, float *out_t);

void oracle_tri_intersect(uint64_t n, const float *tri, const uint8_t *cull, const float *rays, float *out_t);
void oracle_tri_normal(uint64_t n, const float *tri, const float *nrm, const float *pos, float *out_n);
void oracle_tri_props(uint64_t n, const float *tri, float *out_area, float *out_box, float *out_face_normal);
void oracle_tri_sample(uint64_t n, const float *tri, const uint8_t *cull, const uint64_t *states, float *out_pos, float *out_p, uint8_t *out_cull,
                       uint64_t *out_states);
void oracle_sphere_intersect(uint64_t n, const float *sph, const float *rays, float *out_t);
void oracle_sphere_normal(uint64_t n, const float *sph, const float *pos, float *out_n);
void oracle_sphere_props(uint64_t n, const float *sph, float *out_area, float *out_box);
void oracle_sphere_sample(uint64_t n, const float *sph, const uint64_t *states, float *out_pos, float *out_p, uint64_t *out_states);

void oracle_bsdf_propagate(int kind, int one_way, uint64_t n, const float *rays, const float *pos, const float *nrm, float epsilon, const float *ior,
                           const uint64_t *states, float *out_ray, float *out_factor, float *out_pd, uint64_t *out_states);
void oracle_bsdf_spectrum(int kind, int one_way, uint64_t n, const float *from_dir, const float *to_dir, const float *nrm, const float *light_rgba,
                          const float *diffuse, const float *specular, int synthetic, float *out_rgba, float *out_shade, float *out_p);

void oracle_camera_shoot(const pto_camera_params *cp, uint64_t n, const float *xy, float pixel_width, float pixel_height, const uint64_t *states,
                         float *out_ray, uint64_t *out_states);

void *oracle_scene_create(const pto_scene_desc *d);
void oracle_scene_destroy(void *h);
void oracle_scene_intersect(void *h, uint64_t n, const float *rays, float *out_t, int32_t *out_obj);
void oracle_scene_sample_lights(void *h, uint64_t n, const float *pos, const uint64_t *states, int max_lights, int32_t *out_count, float *out_pos,
                                float *out_rgba, float *out_pd, uint64_t *out_states);
uint64_t oracle_bvh_dump(const pto_scene_desc *d, int32_t *out_obj, float *out_box);

void oracle_get_sample(void *h, const pto_camera_params *cp, const pto_options *op, uint64_t n, const float *xy_camera, const uint64_t *states,
                       float *out_rgba, uint8_t *out_collected, uint64_t *out_states);
void oracle_render_streams(void *h, const pto_camera_params *cp, const pto_options *op, const pto_stream *streams, uint64_t n, float *out_image,
                           uint64_t *out_states, int n_threads);

/* Counters accumulated by oracle_get_sample / oracle_render_streams / oracle_scene_intersect on this scene since the last reset. */
void oracle_counters_reset(void *h);
void oracle_counters_get(void *h, oracle_counters *out);

#ifdef __cplusplus
}
#endif

#endif
