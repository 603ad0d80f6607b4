/*
 * oracle/pt_oracle.c -- plain-C CPU restatement of CPUPathTrace's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load the library
 * built from this file; the product (cpupathtrace_amd/, include/) never includes, links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit for bit against the compiled, unmodified reference
 * (oracle/_ref/libptref.so, recipe in oracle/Makefile) by tests/test_oracle_vs_reference.py in the build container, and
 * against the golden vectors that the reference produced (tests/golden/*.npz, generator tests/golden/make_golden.py)
 * everywhere, including the GPU box where /root/reference does not exist.
 *
 * The restatement keeps the reference's structure (recursive ordered BVH traversal, std::vector-of-lights loop,
 * per-pixel estimator) and its exact fp32/fp64 evaluation order; it replaces the pointer-based object graph with flat
 * arrays.  Third-party arithmetic on the path, absent from /root/reference:
 *   - libstdc++ 11.4 <random>: uniform_real_distribution<float> = 1 draw, float(draw)/2^32, clamp to nextafter(1,0)
 *     (bits/random.tcc:3348-3382); bernoulli_distribution = 2 draws, low word first, in double (bits/random.h:3635-3644).
 *   - glibc 2.35 libm: sinf, cosf, powf, acosf, sqrtf, log10 -- called here exactly as the reference calls them.
 * Argument evaluation order at propagation.cpp:97 is the clang one (left to right: r1 is the first draw).
 *
 * Citations are file:line under /root/reference.
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------------ */
/* util/vector.h, util/color.h                                                                                        */
/* ------------------------------------------------------------------------------------------------------------------ */

typedef struct {
    float e[3];
} v3;
typedef struct {
    float e[4];
} c4;

static inline v3 v3_make(float x, float y, float z) {
    v3 r = {{x, y, z}};
    return r;
}
static inline v3 v3_ld(const float *p) {
    return v3_make(p[0], p[1], p[2]);
}
/* vector.h:40-58 */
static inline v3 v3_sub(v3 a, v3 b) {
    return v3_make(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]);
}
static inline v3 v3_add(v3 a, v3 b) {
    return v3_make(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]);
}
/* vector.h:76-84 */
static inline v3 v3_scale(v3 a, float f) {
    return v3_make(a.e[0] * f, a.e[1] * f, a.e[2] * f);
}
/* vector.h:122-130 */
static inline v3 v3_neg(v3 a) {
    return v3_make(-a.e[0], -a.e[1], -a.e[2]);
}
/* vector.h:193-201: accumulates from 0, left to right */
static inline float v3_dot(v3 a, v3 b) {
    float d = 0.0F;
    d += a.e[0] * b.e[0];
    d += a.e[1] * b.e[1];
    d += a.e[2] * b.e[2];
    return d;
}
/* vector.h:137-146 */
static inline float v3_len2(v3 a) {
    float l = 0.0F;
    l += a.e[0] * a.e[0];
    l += a.e[1] * a.e[1];
    l += a.e[2] * a.e[2];
    return l;
}
/* vector.h:153 */
static inline float v3_len(v3 a) {
    return sqrtf(v3_len2(a));
}
/* vector.h:161-167: reciprocal, then multiply */
static inline v3 v3_normalize(v3 a) {
    float inv = 1.0F / v3_len(a);
    return v3_scale(a, inv);
}
/* vector.h:235-237 */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
/* vector.h:250-255: v - (n * 2) * d */
static inline v3 v3_reflect(v3 v, v3 n) {
    float d = v3_dot(v, n);
    return v3_sub(v, v3_scale(v3_scale(n, 2.0F), d));
}
/* std::min / std::max as libstdc++ defines them */
static inline float f_min(float a, float b) {
    return (b < a) ? b : a;
}
static inline float f_max(float a, float b) {
    return (a < b) ? b : a;
}
/* vector.h:204-224 */
static inline v3 v3_min(v3 a, v3 b) {
    return v3_make(f_min(a.e[0], b.e[0]), f_min(a.e[1], b.e[1]), f_min(a.e[2], b.e[2]));
}
static inline v3 v3_max(v3 a, v3 b) {
    return v3_make(f_max(a.e[0], b.e[0]), f_max(a.e[1], b.e[1]), f_max(a.e[2], b.e[2]));
}

static inline c4 c4_make(float r, float g, float b, float a) {
    c4 c = {{r, g, b, a}};
    return c;
}
static inline c4 c4_ld(const float *p) {
    return c4_make(p[0], p[1], p[2], p[3]);
}
static inline c4 c4_add(c4 a, c4 b) {
    return c4_make(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2], a.e[3] + b.e[3]);
}
static inline c4 c4_sub(c4 a, c4 b) {
    return c4_make(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2], a.e[3] - b.e[3]);
}
static inline c4 c4_mul(c4 a, c4 b) {
    return c4_make(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2], a.e[3] * b.e[3]);
}
static inline c4 c4_scale(c4 a, float f) {
    return c4_make(a.e[0] * f, a.e[1] * f, a.e[2] * f, a.e[3] * f);
}
/* vector.h:96-104: true division per component */
static inline c4 c4_div(c4 a, float d) {
    return c4_make(a.e[0] / d, a.e[1] / d, a.e[2] / d, a.e[3] / d);
}

typedef struct {
    v3 o, d;
} ray_t;

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-1..a-3: base.h:24-58 and libstdc++ distributions                                                                 */
/* ------------------------------------------------------------------------------------------------------------------ */

typedef struct {
    uint64_t s;
} rng_t;

static inline uint64_t rng_seed_to_state(uint64_t seed) {
    return seed ^ (~seed << 32); /* base.h:26 */
}

static inline uint32_t rng_draw(rng_t *r) {
    uint64_t result = r->s * 0xD989BCACC137DCD5ULL; /* base.h:29 */
    r->s ^= r->s >> 11;
    r->s ^= r->s << 31;
    r->s ^= r->s >> 18;
    return (uint32_t)(result >> 32);
}

/* generate_canonical<float, 24>: one draw (random.tcc:3348-3382) */
static inline float rng_canonical_f(rng_t *r) {
    float sum = 0.0F;
    float tmp = 1.0F;
    sum += (float)rng_draw(r) * tmp;
    tmp *= 4294967296.0F;
    float ret = sum / tmp;
    if(ret >= 1.0F) {
        ret = nextafterf(1.0F, 0.0F);
    }
    return ret;
}

/* uniform_real_distribution<float>(a, b)(re) (random.h:1865-1871) */
static inline float rng_uniform(rng_t *r, float a, float b) {
    return rng_canonical_f(r) * (b - a) + a;
}

/* bernoulli_distribution(p)(re) (random.h:3635-3644): two draws, the first is the low word */
static inline int rng_bernoulli(rng_t *r, double p) {
    double sum = 0.0;
    double tmp = 1.0;
    sum += (double)rng_draw(r) * tmp;
    tmp *= 4294967296.0;
    sum += (double)rng_draw(r) * tmp;
    tmp *= 4294967296.0;
    double ret = sum / tmp;
    if(ret >= 1.0) {
        ret = nextafter(1.0, 0.0);
    }
    return (ret - 0.0) < p * (1.0 - 0.0);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* scene representation                                                                                               */
/* ------------------------------------------------------------------------------------------------------------------ */

typedef struct {
    v3 lo, hi;
    int left, right; /* inner node children, -1 on a leaf */
    int obj;         /* leaf: object index; -1 on an inner node; -2 NullObject */
} node_t;

typedef struct {
    /* objects in construction order */
    uint32_t n_obj;
    uint8_t *kind;
    uint32_t *kidx; /* index into the typed arrays */
    uint32_t n_tri;
    float *tri_pos;
    float *tri_nrm;
    uint8_t *tri_cull;
    uint32_t *tri_mat;
    uint32_t n_sph;
    float *sph;
    uint32_t *sph_mat;
    uint32_t n_mat;
    pto_material *mat;
    uint32_t n_lights;
    float *light_pos;
    float *light_spec;
    /* BVH (scene.cpp:12-102) */
    node_t *nodes;
    size_t n_nodes, cap_nodes;
    int root;
    /* emissive registry (scene.cpp:183-208, 167-180) */
    int *emissive;
    float *cdf;
    int n_emissive;
    /* counters */
    pthread_mutex_t cnt_lock;
    oracle_counters cnt;
} scene_t;

/* default material of an object without a handler: white Lambertian (object.cpp:9-11) */
static const pto_material DEFAULT_MATERIAL = {{1.0F, 1.0F, 1.0F, 1.0F}, {1.0F, 1.0F, 1.0F, 1.0F}, {0.0F, 0.0F, 0.0F, 0.0F}, 1.0F, PTO_BSDF_LAMBERTIAN, 0, 0};

static const pto_material *obj_material(const scene_t *s, int obj) {
    uint32_t m = s->kind[obj] == PTO_OBJ_TRIANGLE ? s->tri_mat[s->kidx[obj]] : s->sph_mat[s->kidx[obj]];
    return m == 0xFFFFFFFFU ? &DEFAULT_MATERIAL : &s->mat[m];
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-4: AABB::getIntersection, bounding_box.cpp:38-73                                                                 */
/* ------------------------------------------------------------------------------------------------------------------ */

static float aabb_intersect(v3 lo, v3 hi, ray_t ray) {
    float zero = 0.0F;
    float ix = fabsf(ray.d.e[0]) > zero ? 1.0F / ray.d.e[0] : FLT_MAX;
    float iy = fabsf(ray.d.e[1]) > zero ? 1.0F / ray.d.e[1] : FLT_MAX;
    float iz = fabsf(ray.d.e[2]) > zero ? 1.0F / ray.d.e[2] : FLT_MAX;

    v3 ld = v3_sub(lo, ray.o);
    v3 hd = v3_sub(hi, ray.o);

    float t1 = ld.e[0] * ix;
    float t2 = hd.e[0] * ix;
    float t3 = ld.e[1] * iy;
    float t4 = hd.e[1] * iy;
    float t5 = ld.e[2] * iz;
    float t6 = hd.e[2] * iz;

    float t_min = f_max(f_max(f_min(t1, t2), f_min(t3, t4)), f_min(t5, t6));
    float t_max = f_min(f_min(f_max(t1, t2), f_max(t3, t4)), f_max(t5, t6));

    float t = t_min;
    if(t_max < zero || t_min > t_max) {
        return -1.0F;
    }
    if(t_min < zero && t_min <= t_max && t_max >= zero) {
        t = zero;
    }
    return t;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-6, a-7: Triangle, object.cpp:118-207                                                                             */
/* ------------------------------------------------------------------------------------------------------------------ */

static float tri_intersect(const float *p, int cull, ray_t ray) {
    const float epsilon = 1E-6F;
    v3 a = v3_ld(p), b = v3_ld(p + 3), c = v3_ld(p + 6);
    v3 ab = v3_sub(b, a);
    v3 ac = v3_sub(c, a);
    v3 pvec = v3_cross(ray.d, ac);
    float det = v3_dot(ab, pvec);

    if(cull) {
        if(det <= epsilon) {
            return -1.0F;
        }
    }
    else {
        if(fabsf(det) <= epsilon) {
            return -1.0F;
        }
    }

    float inv_det = 1.0F / det;
    v3 tvec = v3_sub(ray.o, a);
    float u = v3_dot(tvec, pvec) * inv_det;
    if(u < 0 || u > 1) {
        return -1.0F;
    }
    v3 qvec = v3_cross(tvec, ab);
    float v = v3_dot(ray.d, qvec) * inv_det;
    if(v < 0 || u + v > 1) {
        return -1.0F;
    }
    return v3_dot(ac, qvec) * inv_det;
}

static v3 tri_normal(const float *p, const float *n, v3 pos) {
    v3 a = v3_ld(p), b = v3_ld(p + 3), c = v3_ld(p + 6);
    v3 ab = v3_sub(b, a);
    v3 ac = v3_sub(c, a);
    v3 ap = v3_sub(pos, a);

    float d00 = v3_dot(ab, ab);
    float d01 = v3_dot(ab, ac);
    float d11 = v3_dot(ac, ac);
    float d20 = v3_dot(ap, ab);
    float d21 = v3_dot(ap, ac);

    float inv_d = 1.0F / (d00 * d11 - d01 * d01);
    float v = (d11 * d20 - d01 * d21) * inv_d;
    float w = (d00 * d21 - d01 * d20) * inv_d;
    float u = 1.0F - v - w;

    v3 na = v3_ld(n), nb = v3_ld(n + 3), nc = v3_ld(n + 6);
    return v3_normalize(v3_add(v3_add(v3_scale(na, u), v3_scale(nb, v)), v3_scale(nc, w)));
}

/* Triangle::Triangle, object.cpp:118-124 */
static v3 tri_face_normal(const float *p) {
    v3 a = v3_ld(p), b = v3_ld(p + 3), c = v3_ld(p + 6);
    return v3_normalize(v3_cross(v3_sub(b, a), v3_sub(c, a)));
}

static void tri_bounds(const float *p, v3 *lo, v3 *hi) {
    v3 a = v3_ld(p), b = v3_ld(p + 3), c = v3_ld(p + 6);
    *lo = v3_min(v3_min(a, b), c); /* object.cpp:184-186 */
    *hi = v3_max(v3_max(a, b), c);
}

static float tri_area(const float *p) {
    v3 a = v3_ld(p), b = v3_ld(p + 3), c = v3_ld(p + 6);
    return v3_len(v3_cross(v3_sub(b, a), v3_sub(c, a))) / 2.0F; /* object.cpp:188-190 */
}

static void tri_sample(const float *p, rng_t *re, v3 *pos, float *pd) {
    v3 a = v3_ld(p), b = v3_ld(p + 3), c = v3_ld(p + 6);
    float r1 = rng_uniform(re, 0.0F, 1.0F);
    float r2 = rng_uniform(re, 0.0F, 1.0F);
    float rr1 = sqrtf(r1);
    *pos = v3_add(v3_add(v3_scale(a, 1.0F - rr1), v3_scale(b, rr1 * (1.0F - r2))), v3_scale(c, rr1 * r2)); /* object.cpp:200 */
    float area = v3_len(v3_cross(v3_sub(b, a), v3_sub(c, a))) / 2.0F;
    *pd = 1.0F / area;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-8: Sphere, object.cpp:68-116                                                                                     */
/* ------------------------------------------------------------------------------------------------------------------ */

static float sphere_intersect(const float *s, ray_t ray) {
    v3 origin = v3_ld(s);
    float radius2 = s[3] * s[3];
    v3 co = v3_sub(ray.o, origin);
    float d = v3_dot(ray.d, co);
    float discriminant = d * d - v3_len2(co) + radius2;
    if(discriminant >= 0) {
        return -(d + sqrtf(discriminant));
    }
    return -1.0F;
}

static v3 sphere_normal(const float *s, v3 pos) {
    return v3_normalize(v3_sub(pos, v3_ld(s)));
}

static float sphere_area(const float *s) {
    const float pi = (float)M_PI;
    return 4.0F * pi * (s[3] * s[3]);
}

static void sphere_sample(const float *s, rng_t *re, v3 *pos, float *pd) {
    const float pi = (float)M_PI;
    float radius2 = s[3] * s[3];
    float theta = 2.0F * pi * rng_uniform(re, 0.0F, 1.0F);
    float phi = acosf(1.0F - 2.0F * rng_uniform(re, 0.0F, 1.0F));
    float x = sinf(phi) * cosf(theta);
    float y = sinf(phi) * sinf(theta);
    float z = cosf(phi);
    *pos = v3_add(v3_ld(s), v3_scale(v3_make(x, y, z), s[3]));
    *pd = 1.0F / (4.0F * pi * radius2);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* object dispatch (the reference's virtual calls)                                                                    */
/* ------------------------------------------------------------------------------------------------------------------ */

static float obj_intersect(const scene_t *s, int obj, ray_t ray, oracle_counters *cnt) {
    if(obj < 0) {
        return -1.0F; /* NullObject, object.cpp:52-54 */
    }
    uint32_t k = s->kidx[obj];
    if(s->kind[obj] == PTO_OBJ_TRIANGLE) {
        cnt->tri_tests++;
        return tri_intersect(s->tri_pos + 9 * (size_t)k, s->tri_cull[k], ray);
    }
    cnt->sphere_tests++;
    return sphere_intersect(s->sph + 4 * (size_t)k, ray);
}

static v3 obj_normal(const scene_t *s, int obj, v3 pos) {
    uint32_t k = s->kidx[obj];
    if(s->kind[obj] == PTO_OBJ_TRIANGLE) {
        return tri_normal(s->tri_pos + 9 * (size_t)k, s->tri_nrm + 9 * (size_t)k, pos);
    }
    return sphere_normal(s->sph + 4 * (size_t)k, pos);
}

static void obj_bounds(const scene_t *s, int obj, v3 *lo, v3 *hi) {
    uint32_t k = s->kidx[obj];
    if(s->kind[obj] == PTO_OBJ_TRIANGLE) {
        tri_bounds(s->tri_pos + 9 * (size_t)k, lo, hi);
    }
    else {
        const float *sp = s->sph + 4 * (size_t)k;
        v3 d = v3_make(sp[3], sp[3], sp[3]); /* object.cpp:90-93 */
        *lo = v3_sub(v3_ld(sp), d);
        *hi = v3_add(v3_ld(sp), d);
    }
}

static float obj_area(const scene_t *s, int obj) {
    uint32_t k = s->kidx[obj];
    return s->kind[obj] == PTO_OBJ_TRIANGLE ? tri_area(s->tri_pos + 9 * (size_t)k) : sphere_area(s->sph + 4 * (size_t)k);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-10: impl::constructBVH, scene.cpp:12-102                                                                         */
/* ------------------------------------------------------------------------------------------------------------------ */

/* value of the k-th smallest element (what std::nth_element leaves at position k); three-way quickselect */
static float select_kth(float *a, int n, int k) {
    int lo = 0, hi = n - 1;
    for(;;) {
        if(lo >= hi) {
            return a[k];
        }
        float x = a[lo], y = a[lo + (hi - lo) / 2], z = a[hi];
        float pivot = (x < y) ? ((y < z) ? y : (x < z ? z : x)) : ((x < z) ? x : (y < z ? z : y));
        int lt = lo, i = lo, gt = hi;
        while(i <= gt) {
            if(a[i] < pivot) {
                float t = a[lt];
                a[lt] = a[i];
                a[i] = t;
                lt++;
                i++;
            }
            else if(a[i] > pivot) {
                float t = a[gt];
                a[gt] = a[i];
                a[i] = t;
                gt--;
            }
            else {
                i++;
            }
        }
        if(k < lt) {
            hi = lt - 1;
        }
        else if(k > gt) {
            lo = gt + 1;
        }
        else {
            return pivot;
        }
    }
}

static int node_new(scene_t *s) {
    if(s->n_nodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 64;
        s->nodes = (node_t *)realloc(s->nodes, s->cap_nodes * sizeof(node_t));
    }
    return (int)s->n_nodes++;
}

/* ids: leaf node indices in input order */
static int bvh_build(scene_t *s, int *ids, int n, float *scratch) {
    if(n == 1) {
        return ids[0]; /* scene.cpp:17-19 */
    }

    float medians[3];
    for(int dim = 0; dim < 3; dim++) {
        for(int i = 0; i < n; i++) {
            scratch[i] = s->nodes[ids[i]].lo.e[dim];
        }
        medians[dim] = select_kth(scratch, n, n / 2 - 1); /* scene.cpp:32-35 */
    }

    float surface_areas[3];
    for(int dim = 0; dim < 3; dim++) {
        v3 clo[2], chi[2];
        for(int g = 0; g < 2; g++) {
            clo[g] = v3_make(INFINITY, INFINITY, INFINITY);
            chi[g] = v3_make(-INFINITY, -INFINITY, -INFINITY);
        }
        for(int i = 0; i < n; i++) {
            const node_t *b = &s->nodes[ids[i]];
            int index = b->lo.e[dim] <= medians[dim] ? 0 : 1;
            clo[index] = v3_min(clo[index], b->lo);
            chi[index] = v3_max(chi[index], b->hi);
        }
        float surface_area = 0.0F;
        for(int g = 0; g < 2; g++) {
            v3 d = v3_sub(chi[g], clo[g]);
            surface_area += 2 * (d.e[0] * d.e[1] + d.e[1] * d.e[2] + d.e[0] * d.e[2]); /* scene.cpp:58 */
        }
        surface_areas[dim] = surface_area;
    }

    int min_index = 0;
    float min_surface = surface_areas[0];
    for(int dim = 1; dim < 3; dim++) {
        if(surface_areas[dim] < min_surface) {
            min_surface = surface_areas[dim];
            min_index = dim;
        }
    }

    int *left = (int *)malloc(sizeof(int) * (size_t)n);
    int *right = (int *)malloc(sizeof(int) * (size_t)n);
    int nl = 0, nr = 0;
    for(int i = 0; i < n; i++) {
        if(s->nodes[ids[i]].lo.e[min_index] <= medians[min_index]) {
            left[nl++] = ids[i];
        }
        else {
            right[nr++] = ids[i];
        }
    }
    /* scene.cpp:90-94: move the last left element to the end of right */
    while(nl > 1 && nl > 2 * nr) {
        right[nr++] = left[--nl];
    }

    int l = bvh_build(s, left, nl, scratch);
    int r = bvh_build(s, right, nr, scratch);
    free(left);
    free(right);

    int me = node_new(s);
    node_t *nd = &s->nodes[me];
    nd->lo = v3_min(s->nodes[l].lo, s->nodes[r].lo); /* bounding_box.cpp:8-10,18-24 */
    nd->hi = v3_max(s->nodes[l].hi, s->nodes[r].hi);
    nd->left = l;
    nd->right = r;
    nd->obj = -1;
    return me;
}

/* Scene::registerEmissiveObjects, scene.cpp:183-208 */
static void register_emissive(scene_t *s, int node) {
    const node_t *nd = &s->nodes[node];
    if(nd->left < 0) {
        if(nd->obj < 0) {
            return; /* NullObject: default material has no emission */
        }
        const pto_material *m = obj_material(s, nd->obj);
        float emissive_power = (m->emission[0] + m->emission[1] + m->emission[2]) * m->emission[3];
        if(emissive_power <= 0.0F) {
            return;
        }
        float object_probability = emissive_power * obj_area(s, nd->obj);
        if(object_probability <= 0.0F) {
            return;
        }
        s->emissive[s->n_emissive] = nd->obj;
        s->cdf[s->n_emissive] = object_probability;
        s->n_emissive++;
    }
    else {
        register_emissive(s, nd->left);
        register_emissive(s, nd->right);
    }
}

static void *dup_mem(const void *p, size_t bytes) {
    void *r = malloc(bytes ? bytes : 1);
    if(p != NULL && bytes) {
        memcpy(r, p, bytes);
    }
    return r;
}

void *oracle_scene_create(const pto_scene_desc *d) {
    scene_t *s = (scene_t *)calloc(1, sizeof(scene_t));
    pthread_mutex_init(&s->cnt_lock, NULL);
    s->n_obj = d->n_objects;
    s->n_tri = d->n_triangles;
    s->n_sph = d->n_spheres;
    s->n_mat = d->n_materials;
    s->n_lights = d->n_point_lights;
    s->kind = (uint8_t *)dup_mem(d->obj_kind, d->n_objects);
    s->kidx = (uint32_t *)malloc(sizeof(uint32_t) * (d->n_objects + 1));
    s->tri_pos = (float *)dup_mem(d->tri_pos, sizeof(float) * 9 * (size_t)d->n_triangles);
    s->tri_nrm = (float *)malloc(sizeof(float) * 9 * (size_t)d->n_triangles + 4);
    if(d->tri_nrm != NULL) {
        memcpy(s->tri_nrm, d->tri_nrm, sizeof(float) * 9 * (size_t)d->n_triangles);
    }
    else {
        for(uint32_t i = 0; i < d->n_triangles; i++) {
            v3 fn = tri_face_normal(s->tri_pos + 9 * (size_t)i);
            for(int k = 0; k < 9; k++) {
                s->tri_nrm[9 * (size_t)i + k] = fn.e[k % 3];
            }
        }
    }
    s->tri_cull = (uint8_t *)dup_mem(d->tri_cull, d->n_triangles);
    s->tri_mat = (uint32_t *)dup_mem(d->tri_material, sizeof(uint32_t) * d->n_triangles);
    s->sph = (float *)dup_mem(d->sph, sizeof(float) * 4 * d->n_spheres);
    s->sph_mat = (uint32_t *)dup_mem(d->sph_material, sizeof(uint32_t) * d->n_spheres);
    s->mat = (pto_material *)dup_mem(d->materials, sizeof(pto_material) * d->n_materials);
    s->light_pos = (float *)dup_mem(d->light_pos, sizeof(float) * 3 * d->n_point_lights);
    s->light_spec = (float *)dup_mem(d->light_spectrum, sizeof(float) * 4 * d->n_point_lights);

    uint32_t ti = 0, si = 0;
    for(uint32_t i = 0; i < d->n_objects; i++) {
        s->kidx[i] = s->kind[i] == PTO_OBJ_TRIANGLE ? ti++ : si++;
    }

    /* Scene::Scene, scene.cpp:153-181: one leaf per object, then constructBVH */
    if(d->n_objects == 0) {
        /* AABB::AABB(): NullObject leaf whose area the reference leaves uninitialised (bounding_box.cpp:13); every
         * query is a miss whatever the area holds (scene.cpp:210-219, object.cpp:52-54). */
        s->root = node_new(s);
        s->nodes[s->root].lo = v3_make(0.0F, 0.0F, 0.0F);
        s->nodes[s->root].hi = v3_make(0.0F, 0.0F, 0.0F);
        s->nodes[s->root].left = s->nodes[s->root].right = -1;
        s->nodes[s->root].obj = -2;
    }
    else {
        int *ids = (int *)malloc(sizeof(int) * d->n_objects);
        for(uint32_t i = 0; i < d->n_objects; i++) {
            int me = node_new(s);
            obj_bounds(s, (int)i, &s->nodes[me].lo, &s->nodes[me].hi);
            s->nodes[me].left = s->nodes[me].right = -1;
            s->nodes[me].obj = (int)i;
            ids[i] = me;
        }
        float *scratch = (float *)malloc(sizeof(float) * d->n_objects);
        s->root = bvh_build(s, ids, (int)d->n_objects, scratch);
        free(scratch);
        free(ids);
    }

    s->emissive = (int *)malloc(sizeof(int) * (d->n_objects + 1));
    s->cdf = (float *)malloc(sizeof(float) * (d->n_objects + 1));
    s->n_emissive = 0;
    register_emissive(s, s->root);

    /* scene.cpp:169-180 */
    float cumulative_probability = 0.0F;
    for(int i = 0; i < s->n_emissive; i++) {
        float probability = s->cdf[i];
        s->cdf[i] += cumulative_probability;
        cumulative_probability += probability;
    }
    for(int i = 0; i < s->n_emissive; i++) {
        s->cdf[i] /= cumulative_probability;
    }
    return s;
}

void oracle_scene_destroy(void *h) {
    scene_t *s = (scene_t *)h;
    if(s == NULL) {
        return;
    }
    free(s->kind);
    free(s->kidx);
    free(s->tri_pos);
    free(s->tri_nrm);
    free(s->tri_cull);
    free(s->tri_mat);
    free(s->sph);
    free(s->sph_mat);
    free(s->mat);
    free(s->light_pos);
    free(s->light_spec);
    free(s->nodes);
    free(s->emissive);
    free(s->cdf);
    pthread_mutex_destroy(&s->cnt_lock);
    free(s);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-5: impl::getChildIntersection + Scene::getIntersection, scene.cpp:104-150, 210-220                               */
/* ------------------------------------------------------------------------------------------------------------------ */

typedef struct {
    float t;
    int obj; /* -1: nullptr */
} hit_t;

static hit_t child_intersection(const scene_t *s, int node, ray_t ray, float t_max, oracle_counters *cnt) {
    const node_t *nd = &s->nodes[node];
    if(nd->left < 0) {
        hit_t h;
        h.t = obj_intersect(s, nd->obj, ray, cnt);
        h.obj = nd->obj;
        return h;
    }

    const float zero = 0.0F;
    cnt->aabb_tests += 2;
    float left_t = aabb_intersect(s->nodes[nd->left].lo, s->nodes[nd->left].hi, ray);
    float right_t = aabb_intersect(s->nodes[nd->right].lo, s->nodes[nd->right].hi, ray);

    float close_t = f_min(left_t, right_t);
    float far_t = f_max(left_t, right_t);
    int close = left_t < right_t ? nd->left : nd->right;
    int far = left_t < right_t ? nd->right : nd->left;

    hit_t close_intersection = {-1.0F, -1};
    if(close_t >= zero && close_t < t_max) {
        close_intersection = child_intersection(s, close, ray, t_max, cnt);
    }

    float close_intersection_t = close_intersection.t;
    if(close_intersection_t >= zero) {
        if(close_intersection_t < far_t) {
            return close_intersection;
        }
        t_max = f_min(t_max, close_intersection_t);
    }

    if(far_t >= zero && far_t < t_max) {
        hit_t far_intersection = child_intersection(s, far, ray, t_max, cnt);
        float far_intersection_t = far_intersection.t;
        if(far_intersection_t < zero || (close_intersection_t >= zero && close_intersection_t < far_intersection_t)) {
            return close_intersection;
        }
        return far_intersection;
    }
    return close_intersection;
}

static hit_t scene_intersect(const scene_t *s, ray_t ray, oracle_counters *cnt) {
    cnt->scene_queries++;
    if(s->n_obj == 0) {
        hit_t h = {-1.0F, -1};
        return h;
    }
    const node_t *root = &s->nodes[s->root];
    cnt->aabb_tests++;
    float t = aabb_intersect(root->lo, root->hi, ray);
    if(t >= 0.0F) {
        return child_intersection(s, s->root, ray, FLT_MAX, cnt);
    }
    hit_t h = {t, -1};
    return h;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-11: Scene::sampleLights, scene.cpp:222-289                                                                       */
/* ------------------------------------------------------------------------------------------------------------------ */

typedef struct {
    v3 pos;
    c4 spectrum;
    float pd;
} light_sample_t;

#define ORACLE_MAX_OBJECT_SAMPLES 16

static int scene_sample_lights(const scene_t *s, v3 pos, rng_t *re, light_sample_t *out, int cap) {
    int emissive_object_count = s->n_emissive;
    int object_sample_count = 2 + (int)log10((double)(emissive_object_count + 1));
    if(emissive_object_count < object_sample_count) {
        object_sample_count = emissive_object_count;
    }

    int n = 0;
    for(uint32_t i = 0; i < s->n_lights; i++) {
        if(n < cap) {
            out[n].pos = v3_ld(s->light_pos + 3 * i);
            out[n].spectrum = c4_ld(s->light_spec + 4 * i);
            out[n].pd = 1.0F; /* light.cpp:35-37 */
        }
        n++;
    }

    for(int i = 0; i < object_sample_count; i++) {
        float r = rng_uniform(re, 0.0F, 1.0F);

        /* std::lower_bound: first cdf[j] >= r */
        int lo = 0, len = s->n_emissive;
        while(len > 0) {
            int half = len >> 1;
            if(s->cdf[lo + half] < r) {
                lo = lo + half + 1;
                len = len - half - 1;
            }
            else {
                len = half;
            }
        }
        int object_index = lo;

        float selection_p = s->cdf[object_index];
        if(object_index > 0) {
            selection_p -= s->cdf[object_index - 1];
        }
        selection_p *= (float)object_sample_count;

        int obj = s->emissive[object_index];
        uint32_t k = s->kidx[obj];
        v3 surface_pos;
        float surface_p;
        int surface_cull;
        if(s->kind[obj] == PTO_OBJ_TRIANGLE) {
            tri_sample(s->tri_pos + 9 * (size_t)k, re, &surface_pos, &surface_p);
            surface_cull = s->tri_cull[k];
        }
        else {
            sphere_sample(s->sph + 4 * (size_t)k, re, &surface_pos, &surface_p);
            surface_cull = 0;
        }
        v3 surface_n = obj_normal(s, obj, surface_pos);

        v3 to_light = v3_sub(surface_pos, pos);
        v3 dir = v3_normalize(to_light);
        float abs_dot = fabsf(v3_dot(v3_neg(dir), surface_n));

        if(!(abs_dot > 0.0F)) {
            continue;
        }
        if(!(v3_len2(to_light) > 0.0F)) {
            continue;
        }
        if(surface_cull) {
            if(!(v3_dot(dir, surface_n) < 0.0F)) {
                continue;
            }
        }

        float conversion_factor = v3_len2(to_light) / abs_dot;
        const pto_material *m = obj_material(s, obj);
        if(n < cap) {
            out[n].pos = surface_pos;
            out[n].spectrum = c4_ld(m->emission);
            out[n].pd = selection_p * surface_p * conversion_factor;
        }
        n++;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-13..a-15: propagation.cpp                                                                                        */
/* ------------------------------------------------------------------------------------------------------------------ */

static const float PI_F = (float)M_PI;

/* propagation.cpp:11-21 */
static v3 importance_sample_cosine(float r1, float r2, float e, float *p) {
    float fac = sqrtf(1.0F - powf(r2, 2.0F / (e + 1)));
    float cos_theta = powf(r2, 1.0F / (e + 1));
    v3 vec = v3_make(fac * cosf(2.0F * PI_F * r1), fac * sinf(2.0F * PI_F * r1), cos_theta);
    *p = (e + 1) * powf(cos_theta, e) / (2.0F * PI_F);
    return vec;
}

/* propagation.cpp:24-62 */
static v3 local_to_global(v3 vec, v3 n) {
    v3 d;
    if(fabsf(n.e[0]) > 0.0F) {
        if(fabsf(n.e[1]) > 0.0F) {
            d = v3_make(0.0F, -n.e[0], n.e[1]);
        }
        else {
            d = v3_make(0.0F, -n.e[0], n.e[2]);
        }
    }
    else {
        if(fabsf(n.e[1]) > 0.0F) {
            d = v3_make(-n.e[1], n.e[2], 0.0F);
        }
        else {
            d = v3_make(1.0F, 0.0F, 0.0F);
        }
    }
    d = v3_normalize(d);
    v3 b1 = v3_normalize(v3_cross(d, n));
    v3 b2 = v3_normalize(v3_cross(b1, n));
    v3 vx = v3_make(b1.e[0], b2.e[0], n.e[0]);
    v3 vy = v3_make(b1.e[1], b2.e[1], n.e[1]);
    v3 vz = v3_make(b1.e[2], b2.e[2], n.e[2]);
    return v3_make(v3_dot(vx, vec), v3_dot(vy, vec), v3_dot(vz, vec));
}

/* propagation.cpp:64-83 */
static void fresnel_reflectance(float ray_dot, float ri_leaving, float ri_entering, float *reflectance, float *cos_theta_t_out) {
    float sin_theta_i = sqrtf(f_max(1.0F - ray_dot * ray_dot, 0.0F));
    float sin_theta_t = ri_leaving / ri_entering * sin_theta_i;
    if(sin_theta_t >= 1.0F) {
        *reflectance = 1.0F;
        *cos_theta_t_out = 0.0F;
        return;
    }
    float cos_theta_t = sqrtf(f_max(1.0F - sin_theta_t * sin_theta_t, 0.0F));
    float r_parallel = ((ri_entering * ray_dot) - (ri_leaving * cos_theta_t)) / ((ri_entering * ray_dot) + (ri_leaving * cos_theta_t));
    float r_perpendicular = ((ri_leaving * ray_dot) - (ri_entering * cos_theta_t)) / ((ri_leaving * ray_dot) + (ri_entering * cos_theta_t));
    *reflectance = (r_parallel * r_parallel + r_perpendicular * r_perpendicular) / 2.0F;
    *cos_theta_t_out = cos_theta_t;
}

static ray_t bsdf_propagate(int kind, int one_way, ray_t ray, v3 pos, v3 normal, float epsilon, rng_t *re, float refractive_index, float *factor,
                            float *pd) {
    ray_t out;
    if(kind == PTO_BSDF_LAMBERTIAN) {
        /* propagation.cpp:89-105; r1 is drawn first (clang evaluates call arguments left to right) */
        float r1 = rng_uniform(re, 0.0F, 1.0F);
        float r2 = rng_uniform(re, 0.0F, 1.0F);
        float p;
        v3 local_dir = importance_sample_cosine(r1, r2, 1.0F, &p);
        v3 dir = local_to_global(local_dir, normal);
        out.o = v3_add(pos, v3_scale(dir, epsilon));
        out.d = dir;
        *factor = 1.0F;
        *pd = p;
        return out;
    }
    if(kind == PTO_BSDF_GLASS) {
        /* propagation.cpp:120-160 */
        float ray_dot = -v3_dot(ray.d, normal);
        float ri_leaving = ray_dot >= 0 ? 1.0F : refractive_index;
        float ri_entering = ray_dot >= 0 ? refractive_index : 1.0F;
        float rat, cos_theta_t;
        fresnel_reflectance(fabsf(ray_dot), ri_leaving, ri_entering, &rat, &cos_theta_t);
        if(rng_bernoulli(re, (double)rat)) {
            v3 dir = v3_reflect(ray.d, v3_scale(normal, ray_dot < 0.0F ? -1.0F : 1.0F));
            out.o = v3_add(pos, v3_scale(dir, epsilon));
            out.d = dir;
            *factor = rat;
            *pd = rat;
            return out;
        }
        float ri_ratio = ri_leaving / ri_entering;
        v3 out_dir = v3_add(v3_scale(ray.d, ri_ratio),
                            v3_scale(v3_scale(normal, ri_ratio * fabsf(ray_dot) - cos_theta_t), ray_dot < 0.0F ? -1.0F : 1.0F));
        out_dir = v3_normalize(out_dir);
        float ri_fac = (ri_entering * ri_entering) / (ri_leaving * ri_leaving);
        out.o = v3_add(pos, v3_scale(out_dir, epsilon));
        out.d = out_dir;
        *factor = ri_fac * (1.0F - rat);
        *pd = 1.0F - rat;
        return out;
    }
    /* MirrorBRDF, propagation.cpp:180-204 */
    int unaligned = v3_dot(ray.d, normal) > 0.0F;
    if(one_way && unaligned) {
        out.o = v3_add(pos, v3_scale(ray.d, epsilon));
        out.d = ray.d;
        *factor = 1.0F;
        *pd = 1.0F;
        return out;
    }
    v3 normal_dir = normal;
    if(!one_way && unaligned) {
        normal_dir = v3_scale(normal_dir, -1.0F);
    }
    v3 dir = v3_reflect(ray.d, normal_dir);
    out.o = v3_add(pos, v3_scale(dir, epsilon));
    out.d = dir;
    *factor = 1.0F;
    *pd = 1.0F;
    return out;
}

static c4 bsdf_spectrum(int kind, int one_way, v3 from_dir, v3 to_dir, v3 normal, c4 light_spectrum, const float *diffuse, const float *specular,
                        int synthetic, float *shade, float *p) {
    if(kind == PTO_BSDF_LAMBERTIAN) {
        /* propagation.cpp:107-116 */
        *shade = f_max(v3_dot(normal, to_dir), 0.0F) / PI_F;
        *p = 1.0F;
        return c4_mul(c4_ld(diffuse), light_spectrum);
    }
    if(kind == PTO_BSDF_GLASS) {
        /* propagation.cpp:162-176 */
        c4 out = light_spectrum;
        if(v3_dot(from_dir, to_dir) <= 0.0F) {
            out = c4_mul(out, c4_ld(specular));
        }
        else {
            out = c4_mul(out, c4_ld(diffuse));
        }
        *shade = 1.0F;
        *p = synthetic ? 0.0F : 1.0F;
        return out;
    }
    /* propagation.cpp:206-217 */
    c4 out = light_spectrum;
    if(!one_way || (v3_dot(from_dir, to_dir) <= 0.0F)) {
        out = c4_mul(out, c4_ld(specular));
    }
    *shade = 1.0F;
    *p = synthetic ? 0.0F : 1.0F;
    return out;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-18: camera.cpp                                                                                                   */
/* ------------------------------------------------------------------------------------------------------------------ */

typedef struct {
    v3 origin, forward, up, right;
    float aperture_width_half, aperture_height_half;
    int aperture_kind;
    float hex_ratio;
    float focal_plane_dist;
} camera_t;

/* camera.cpp:53-76 */
static camera_t camera_make(const pto_camera_params *c) {
    camera_t cam;
    cam.origin = v3_ld(c->origin);
    v3 forward_dir = v3_normalize(v3_sub(v3_ld(c->look_at), cam.origin));
    cam.forward = v3_scale(forward_dir, c->focal_length);
    v3 up_dir = v3_normalize(v3_ld(c->up));
    float height_half = c->height / 2.0F;
    cam.up = v3_scale(up_dir, height_half);
    v3 right_dir = v3_normalize(v3_cross(cam.forward, cam.up));
    float width_half = height_half * c->aspect_ratio;
    cam.right = v3_scale(right_dir, width_half);
    cam.aperture_width_half = c->aperture_width / 2.0F;
    cam.aperture_height_half = c->aperture_height / 2.0F;
    cam.aperture_kind = c->aperture_kind;
    cam.hex_ratio = f_min(f_max(c->hex_ratio, 0.0F), 1.0F); /* camera.cpp:22-24 */
    cam.focal_plane_dist = c->focal_plane_dist;
    return cam;
}

/* camera.cpp:78-113 */
static ray_t camera_shoot(const camera_t *cam, float x, float y, float pixel_width, float pixel_height, rng_t *re) {
    float offset_x = rng_uniform(re, -pixel_width / 2.0F, pixel_width / 2.0F);
    float offset_y = rng_uniform(re, -pixel_height / 2.0F, pixel_height / 2.0F);
    float sensor_x = x + offset_x;
    float sensor_y = y + offset_y;
    v3 sensor_pos = v3_sub(v3_sub(v3_sub(cam->origin, cam->forward), v3_scale(cam->up, sensor_y)), v3_scale(cam->right, sensor_x));

    float aperture_offset_x = 0.0F;
    float aperture_offset_y = 0.0F;
    if(cam->aperture_kind == PTO_APERTURE_CIRCULAR) {
        /* camera.cpp:7-19 */
        float r = sqrtf(rng_uniform(re, 0.0F, 1.0F));
        float theta = 2 * PI_F * rng_uniform(re, 0.0F, 1.0F);
        float sx = r * cosf(theta);
        float sy = r * sinf(theta);
        aperture_offset_x = sx * cam->aperture_width_half;
        aperture_offset_y = sy * cam->aperture_height_half;
    }
    else if(cam->aperture_kind == PTO_APERTURE_HEXAGONAL) {
        /* camera.cpp:26-50 */
        float sx, sy;
        int in_polygon;
        do {
            sx = rng_uniform(re, 0.0F, 1.0F);
            sy = rng_uniform(re, 0.0F, 1.0F);
            float relative_x = sx - cam->hex_ratio;
            in_polygon = (relative_x <= 0.0F) || (relative_x / (1.0F - cam->hex_ratio)) >= sy;
        } while(!in_polygon);
        if(rng_bernoulli(re, 0.5)) {
            sx = -sx;
        }
        if(rng_bernoulli(re, 0.5)) {
            sy = -sy;
        }
        aperture_offset_x = sx * cam->aperture_width_half;
        aperture_offset_y = sy * cam->aperture_height_half;
    }
    ray_t ray;
    ray.o = v3_add(v3_add(cam->origin, v3_scale(cam->up, aperture_offset_x)), v3_scale(cam->right, aperture_offset_y));
    if(cam->focal_plane_dist > 0.0F) {
        v3 base_dir = v3_normalize(v3_sub(cam->origin, sensor_pos));
        v3 ray_target = v3_add(cam->origin, v3_scale(base_dir, cam->focal_plane_dist / v3_dot(cam->forward, base_dir)));
        ray.d = v3_normalize(v3_sub(ray_target, ray.o));
    }
    else {
        ray.d = v3_normalize(v3_sub(ray.o, sensor_pos));
    }
    return ray;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-12: impl::getSample, worker.cpp:26-146                                                                           */
/* ------------------------------------------------------------------------------------------------------------------ */

/* worker.cpp:12-14 */
static inline float get_contribution(c4 color) {
    return (color.e[0] + color.e[1] + color.e[2]) / 3.0F;
}

#define ORACLE_MAX_LIGHTS 64

static c4 get_sample(const scene_t *s, const camera_t *cam, const pto_options *opt, float x_camera, float y_camera, rng_t *re, int *collected,
                     oracle_counters *cnt) {
    float pixel_width = 1.0F / (float)opt->image_width;
    float pixel_height = 1.0F / (float)opt->image_height;
    float epsilon = opt->epsilon;
    cnt->samples++;

    ray_t ray = camera_shoot(cam, x_camera, y_camera, pixel_width, pixel_height, re);

    int sample_collected = 0;
    float contribution_unweighted = 1.0F;
    double sample_divisor = 1.0F;
    double sample_bounce_pd = 1.0;
    c4 sample_spectrum = c4_make(1.0F, 1.0F, 1.0F, 1.0F);
    c4 out_spectrum = c4_make(0.0F, 0.0F, 0.0F, 0.0F);
    int path_length = 0;
    for(;;) {
        hit_t hit = scene_intersect(s, ray, cnt);
        if(hit.t < 0.0F) {
            break;
        }
        path_length++;
        sample_collected = 1;
        cnt->vertices++;

        v3 pos = v3_add(ray.o, v3_scale(ray.d, hit.t));
        v3 n = obj_normal(s, hit.obj, pos);
        const pto_material *material = obj_material(s, hit.obj);

        c4 emission = c4_ld(material->emission);
        out_spectrum = c4_add(out_spectrum, c4_div(c4_mul(sample_spectrum, emission), (float)(sample_divisor * sample_bounce_pd)));

        float bounce_probability =
          path_length <= 4 ? 1.0F : 0.1F + 0.1F * f_min(contribution_unweighted * get_contribution(sample_spectrum), 1.0F);
        int do_bounce = rng_uniform(re, 0.0F, 1.0F) < bounce_probability;

        light_sample_t lights[ORACLE_MAX_LIGHTS];
        int n_lights = scene_sample_lights(s, pos, re, lights, ORACLE_MAX_LIGHTS);
        if(n_lights > ORACLE_MAX_LIGHTS) {
            n_lights = ORACLE_MAX_LIGHTS;
        }
        for(int li = 0; li < n_lights; li++) {
            v3 to_light = v3_sub(lights[li].pos, pos);
            v3 light_dir = v3_normalize(to_light);
            ray_t light_ray;
            light_ray.o = v3_add(pos, v3_scale(light_dir, epsilon));
            light_ray.d = light_dir;

            cnt->shadow_rays++;
            float light_t = scene_intersect(s, light_ray, cnt).t;
            if(light_t < 0.0F || (light_t >= v3_len(to_light) - epsilon)) {
                float shading_factor, shadow_ray_pd;
                c4 base_spectrum = bsdf_spectrum(material->bsdf, material->one_way, ray.d, light_ray.d, n, lights[li].spectrum, material->diffuse,
                                                 material->specular, 1, &shading_factor, &shadow_ray_pd);
                if(shadow_ray_pd > 0.0F) {
                    c4 combined_spectrum = c4_mul(c4_scale(base_spectrum, shading_factor), sample_spectrum);
                    c4 weighed_spectrum =
                      c4_div(combined_spectrum, (float)(sample_divisor * sample_bounce_pd * lights[li].pd * shadow_ray_pd));
                    out_spectrum = c4_add(out_spectrum, weighed_spectrum);
                }
            }
        }

        if(!do_bounce) {
            sample_bounce_pd *= 1.0F - bounce_probability;
            break;
        }
        sample_bounce_pd *= bounce_probability;
        if(sample_bounce_pd <= 1E-20) {
            break;
        }

        float ray_factor, ray_pd;
        ray_t next_ray = bsdf_propagate(material->bsdf, material->one_way, ray, pos, n, epsilon, re, material->ior, &ray_factor, &ray_pd);
        sample_divisor *= ray_pd;
        sample_divisor /= ray_factor;
        contribution_unweighted *= ray_factor;

        float shading_factor, shading_pd;
        c4 shaded_spectrum = bsdf_spectrum(material->bsdf, material->one_way, ray.d, next_ray.d, n, sample_spectrum, material->diffuse,
                                           material->specular, 0, &shading_factor, &shading_pd);
        sample_divisor *= shading_pd;
        sample_divisor /= shading_factor;
        contribution_unweighted *= shading_factor;
        sample_spectrum = shaded_spectrum;

        if(sample_divisor <= 1E-20) {
            break;
        }
        ray = next_ray;
    }

    out_spectrum.e[3] = sample_collected ? 1.0F : 0.0F;
    *collected = sample_collected;
    return out_spectrum;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a-19: processItem, worker.cpp:149-326                                                                              */
/* ------------------------------------------------------------------------------------------------------------------ */

static inline int i_min(int a, int b) {
    return (b < a) ? b : a;
}
static inline int i_max(int a, int b) {
    return (a < b) ? b : a;
}

typedef struct {
    c4 mean, m2;
    int count;
} candidate_t;

typedef struct {
    c4 color;
    float stddev;
} pixel_candidate_t;

static const c4 C4_ZERO = {{0.0F, 0.0F, 0.0F, 0.0F}};

static void process_item(const scene_t *s, const camera_t *cam, const pto_options *opt, const pto_stream *item, rng_t *re, float *out_image,
                         oracle_counters *cnt) {
    const float one_half = 1.0F / 2.0F;
    const int min_sc = opt->min_sample_count;
    const int max_sc = opt->max_sample_count;

    int stats_sample_count = i_min(i_max(min_sc / 4, 1), 64);
    int candidate_batch_count = i_max(i_max(min_sc, max_sc / 4) / stats_sample_count, 2);
    int check_sample_count = i_min(i_max(i_max(i_max(min_sc / 2, (max_sc - min_sc) / 8), 8), stats_sample_count), 1024) / stats_sample_count;

    size_t cand_cap = 16;
    candidate_t *candidates = (candidate_t *)malloc(sizeof(candidate_t) * cand_cap);
    pixel_candidate_t *pixel_candidates = (pixel_candidate_t *)malloc(sizeof(pixel_candidate_t) * cand_cap);

    for(int y = item->y; y < item->y + item->h; y++) {
        for(int x = item->x; x < item->x + item->w; x++) {
            float x_camera = 2 * (((float)x + one_half) / (float)opt->image_width - one_half);
            float y_camera = 2 * (((float)y + one_half) / (float)opt->image_height - one_half);
            y_camera = -y_camera;

            c4 pixel_value = C4_ZERO;
            int collected_sample_count = 0;
            c4 contribution_mean = C4_ZERO;
            c4 contribution_m2 = C4_ZERO;
            int contribution_count = 0;
            int stats_sample_index = 0;
            c4 sample_aggregate = C4_ZERO;
            size_t n_candidates = 0;
            c4 candidate_mean = C4_ZERO;
            c4 candidate_m2 = C4_ZERO;
            int candidate_count = 0;
            int remaining_checks = check_sample_count;
            int accepted_candidate = 0;

            for(int pixel_sample = 0; pixel_sample < max_sc; pixel_sample++) {
                int sample_collected;
                c4 color_contribution = get_sample(s, cam, opt, x_camera, y_camera, re, &sample_collected, cnt);
                if(!sample_collected) {
                    continue;
                }
                contribution_count++;
                stats_sample_index++;
                sample_aggregate = c4_add(sample_aggregate, color_contribution);

                if(stats_sample_index == stats_sample_count) {
                    sample_aggregate = c4_div(sample_aggregate, (float)stats_sample_count);

                    c4 delta = c4_sub(sample_aggregate, contribution_mean);
                    contribution_mean = c4_add(contribution_mean, c4_div(delta, (float)(contribution_count / stats_sample_count)));
                    c4 delta2 = c4_sub(sample_aggregate, contribution_mean);
                    contribution_m2 = c4_add(contribution_m2, c4_mul(delta, delta2));

                    if(candidate_count == candidate_batch_count) {
                        if(n_candidates + 1 >= cand_cap) {
                            cand_cap *= 2;
                            candidates = (candidate_t *)realloc(candidates, sizeof(candidate_t) * cand_cap);
                            pixel_candidates = (pixel_candidate_t *)realloc(pixel_candidates, sizeof(pixel_candidate_t) * cand_cap);
                        }
                        candidates[n_candidates].mean = candidate_mean;
                        candidates[n_candidates].m2 = candidate_m2;
                        candidates[n_candidates].count = candidate_count;
                        n_candidates++;
                        candidate_mean = C4_ZERO;
                        candidate_m2 = C4_ZERO;
                        candidate_count = 0;
                    }

                    candidate_count++;
                    c4 candidate_delta = c4_sub(sample_aggregate, candidate_mean);
                    candidate_mean = c4_add(candidate_mean, c4_div(candidate_delta, (float)candidate_count));
                    c4 candidate_delta2 = c4_sub(sample_aggregate, candidate_mean);
                    candidate_m2 = c4_add(candidate_m2, c4_mul(candidate_delta, candidate_delta2));

                    stats_sample_index = 0;
                    sample_aggregate = C4_ZERO;
                }

                pixel_value = c4_add(pixel_value, color_contribution);
                collected_sample_count++;

                if(stats_sample_index == 0 && collected_sample_count >= i_max(min_sc, 2)) {
                    int passed_check = 0;
                    if(contribution_count / stats_sample_count >= 2) {
                        c4 m2_weighted = c4_div(contribution_m2, (float)(contribution_count / stats_sample_count - 1));
                        float stddev = sqrtf(m2_weighted.e[0] + m2_weighted.e[1] + m2_weighted.e[2]);
                        /* worker.cpp:245: the ratio is evaluated in double because of the 1E-5 literal */
                        if(stddev < 1E-4F || (double)stddev / ((double)(3 * 3 * get_contribution(contribution_mean)) + 1E-5) < (double)0.2F) {
                            passed_check = 1;
                            remaining_checks--;
                            if(remaining_checks <= 0) {
                                accepted_candidate = 1;
                                break;
                            }
                        }
                    }
                    if(!passed_check) {
                        remaining_checks = check_sample_count;
                    }
                }
            }

            if(collected_sample_count > 0) {
                pixel_value = c4_scale(pixel_value, 1.0F / (float)collected_sample_count);
            }

            if(candidate_count > 0) {
                candidates[n_candidates].mean = candidate_mean;
                candidates[n_candidates].m2 = candidate_m2;
                candidates[n_candidates].count = candidate_count;
                n_candidates++;
            }

            if(!accepted_candidate) {
                size_t n_pc = 0;
                for(size_t i = 0; i < n_candidates; i++) {
                    if(candidates[i].count < i_max((candidate_batch_count * 3) / 4, 2)) {
                        continue;
                    }
                    c4 m2_weighted = c4_div(candidates[i].m2, (float)candidates[i].count);
                    float stddev = sqrtf(m2_weighted.e[0] + m2_weighted.e[1] + m2_weighted.e[2]);
                    pixel_candidates[n_pc].color = candidates[i].mean;
                    pixel_candidates[n_pc].stddev = stddev;
                    n_pc++;
                }

                if(n_pc > 0) {
                    /* std::sort on <= 16 elements is libstdc++'s insertion sort (bits/stl_algo.h __insertion_sort):
                     * an element smaller than the first is rotated to the front, otherwise it is inserted linearly. */
                    for(size_t i = 1; i < n_pc; i++) {
                        pixel_candidate_t val = pixel_candidates[i];
                        if(val.stddev < pixel_candidates[0].stddev) {
                            memmove(&pixel_candidates[1], &pixel_candidates[0], sizeof(pixel_candidate_t) * i);
                            pixel_candidates[0] = val;
                        }
                        else {
                            size_t j = i;
                            while(val.stddev < pixel_candidates[j - 1].stddev) {
                                pixel_candidates[j] = pixel_candidates[j - 1];
                                j--;
                            }
                            pixel_candidates[j] = val;
                        }
                    }

                    pixel_value = pixel_candidates[0].color;
                    float stddev = pixel_candidates[0].stddev;
                    for(size_t i = 1; i < n_pc; i++) {
                        float stddev_other = pixel_candidates[i].stddev;
                        c4 color_other = pixel_candidates[i].color;
                        if(stddev_other < f_max(stddev + 0.005F, stddev * 1.01F)) {
                            pixel_value = c4_add(pixel_value, c4_div(c4_sub(color_other, pixel_value), (float)(i + 1)));
                            stddev = stddev_other;
                        }
                        else {
                            break;
                        }
                    }
                }
            }

            float *o = out_image + 4 * ((size_t)y * (size_t)opt->image_width + (size_t)x);
            o[0] = pixel_value.e[0];
            o[1] = pixel_value.e[1];
            o[2] = pixel_value.e[2];
            o[3] = pixel_value.e[3];
        }
    }
    free(candidates);
    free(pixel_candidates);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* entry points                                                                                                       */
/* ------------------------------------------------------------------------------------------------------------------ */

static void counters_merge(scene_t *s, const oracle_counters *c) {
    pthread_mutex_lock(&s->cnt_lock);
    s->cnt.samples += c->samples;
    s->cnt.scene_queries += c->scene_queries;
    s->cnt.aabb_tests += c->aabb_tests;
    s->cnt.tri_tests += c->tri_tests;
    s->cnt.sphere_tests += c->sphere_tests;
    s->cnt.vertices += c->vertices;
    s->cnt.shadow_rays += c->shadow_rays;
    pthread_mutex_unlock(&s->cnt_lock);
}

void oracle_counters_reset(void *h) {
    scene_t *s = (scene_t *)h;
    pthread_mutex_lock(&s->cnt_lock);
    memset(&s->cnt, 0, sizeof(s->cnt));
    pthread_mutex_unlock(&s->cnt_lock);
}

void oracle_counters_get(void *h, oracle_counters *out) {
    scene_t *s = (scene_t *)h;
    pthread_mutex_lock(&s->cnt_lock);
    *out = s->cnt;
    pthread_mutex_unlock(&s->cnt_lock);
}

void oracle_rng_draws(uint64_t seed, uint64_t n, uint32_t *out) {
    rng_t re = {rng_seed_to_state(seed)};
    for(uint64_t i = 0; i < n; i++) {
        out[i] = rng_draw(&re);
    }
}

uint64_t oracle_rng_state_after(uint64_t seed, uint64_t n_draws) {
    rng_t re = {rng_seed_to_state(seed)};
    for(uint64_t i = 0; i < n_draws; i++) {
        rng_draw(&re);
    }
    return re.s;
}

void oracle_uniform_floats(uint64_t seed, float a, float b, uint64_t n, float *out) {
    rng_t re = {rng_seed_to_state(seed)};
    for(uint64_t i = 0; i < n; i++) {
        out[i] = rng_uniform(&re, a, b);
    }
}

uint64_t oracle_bernoulli(uint64_t seed, double p, uint64_t n, uint8_t *out_flags) {
    rng_t re = {rng_seed_to_state(seed)};
    for(uint64_t i = 0; i < n; i++) {
        out_flags[i] = rng_bernoulli(&re, p) ? 1 : 0;
    }
    return re.s;
}

static ray_t ray_ld(const float *r) {
    ray_t ray;
    ray.o = v3_ld(r);
    ray.d = v3_ld(r + 3);
    return ray;
}

static void v3_st(float *o, v3 v) {
    o[0] = v.e[0];
    o[1] = v.e[1];
    o[2] = v.e[2];
}

void oracle_aabb_intersect(uint64_t n, const float *boxes, const float *rays, float *out_t) {
    for(uint64_t i = 0; i < n; i++) {
        out_t[i] = aabb_intersect(v3_ld(boxes + 6 * i), v3_ld(boxes + 6 * i + 3), ray_ld(rays + 6 * i));
    }
}

void oracle_tri_intersect(uint64_t n, const float *tri, const uint8_t *cull, const float *rays, float *out_t) {
    for(uint64_t i = 0; i < n; i++) {
        out_t[i] = tri_intersect(tri + 9 * i, cull[i], ray_ld(rays + 6 * i));
    }
}

void oracle_tri_normal(uint64_t n, const float *tri, const float *nrm, const float *pos, float *out_n) {
    for(uint64_t i = 0; i < n; i++) {
        v3_st(out_n + 3 * i, tri_normal(tri + 9 * i, nrm + 9 * i, v3_ld(pos + 3 * i)));
    }
}

void oracle_tri_props(uint64_t n, const float *tri, float *out_area, float *out_box, float *out_face_normal) {
    for(uint64_t i = 0; i < n; i++) {
        out_area[i] = tri_area(tri + 9 * i);
        v3 lo, hi;
        tri_bounds(tri + 9 * i, &lo, &hi);
        v3_st(out_box + 6 * i, lo);
        v3_st(out_box + 6 * i + 3, hi);
        v3_st(out_face_normal + 3 * i, tri_face_normal(tri + 9 * i));
    }
}

void oracle_tri_sample(uint64_t n, const float *tri, const uint8_t *cull, const uint64_t *states, float *out_pos, float *out_p, uint8_t *out_cull,
                       uint64_t *out_states) {
    for(uint64_t i = 0; i < n; i++) {
        rng_t re = {states[i]};
        v3 pos;
        tri_sample(tri + 9 * i, &re, &pos, &out_p[i]);
        v3_st(out_pos + 3 * i, pos);
        out_cull[i] = cull[i] ? 1 : 0;
        out_states[i] = re.s;
    }
}

void oracle_sphere_intersect(uint64_t n, const float *sph, const float *rays, float *out_t) {
    for(uint64_t i = 0; i < n; i++) {
        out_t[i] = sphere_intersect(sph + 4 * i, ray_ld(rays + 6 * i));
    }
}

void oracle_sphere_normal(uint64_t n, const float *sph, const float *pos, float *out_n) {
    for(uint64_t i = 0; i < n; i++) {
        v3_st(out_n + 3 * i, sphere_normal(sph + 4 * i, v3_ld(pos + 3 * i)));
    }
}

void oracle_sphere_props(uint64_t n, const float *sph, float *out_area, float *out_box) {
    for(uint64_t i = 0; i < n; i++) {
        out_area[i] = sphere_area(sph + 4 * i);
        const float *sp = sph + 4 * i;
        v3 d = v3_make(sp[3], sp[3], sp[3]);
        v3_st(out_box + 6 * i, v3_sub(v3_ld(sp), d));
        v3_st(out_box + 6 * i + 3, v3_add(v3_ld(sp), d));
    }
}

void oracle_sphere_sample(uint64_t n, const float *sph, const uint64_t *states, float *out_pos, float *out_p, uint64_t *out_states) {
    for(uint64_t i = 0; i < n; i++) {
        rng_t re = {states[i]};
        v3 pos;
        sphere_sample(sph + 4 * i, &re, &pos, &out_p[i]);
        v3_st(out_pos + 3 * i, pos);
        out_states[i] = re.s;
    }
}

void oracle_bsdf_propagate(int kind, int one_way, uint64_t n, const float *rays, const float *pos, const float *nrm, float epsilon, const float *ior,
                           const uint64_t *states, float *out_ray, float *out_factor, float *out_pd, uint64_t *out_states) {
    for(uint64_t i = 0; i < n; i++) {
        rng_t re = {states[i]};
        ray_t out = bsdf_propagate(kind, one_way, ray_ld(rays + 6 * i), v3_ld(pos + 3 * i), v3_ld(nrm + 3 * i), epsilon, &re, ior[i],
                                   &out_factor[i], &out_pd[i]);
        v3_st(out_ray + 6 * i, out.o);
        v3_st(out_ray + 6 * i + 3, out.d);
        out_states[i] = re.s;
    }
}

void oracle_bsdf_spectrum(int kind, int one_way, uint64_t n, const float *from_dir, const float *to_dir, const float *nrm, const float *light_rgba,
                          const float *diffuse, const float *specular, int synthetic, float *out_rgba, float *out_shade, float *out_p) {
    for(uint64_t i = 0; i < n; i++) {
        c4 c = bsdf_spectrum(kind, one_way, v3_ld(from_dir + 3 * i), v3_ld(to_dir + 3 * i), v3_ld(nrm + 3 * i), c4_ld(light_rgba + 4 * i),
                             diffuse + 4 * i, specular + 4 * i, synthetic, &out_shade[i], &out_p[i]);
        memcpy(out_rgba + 4 * i, c.e, 16);
    }
}

void oracle_camera_shoot(const pto_camera_params *cp, uint64_t n, const float *xy, float pixel_width, float pixel_height, const uint64_t *states,
                         float *out_ray, uint64_t *out_states) {
    camera_t cam = camera_make(cp);
    for(uint64_t i = 0; i < n; i++) {
        rng_t re = {states[i]};
        ray_t r = camera_shoot(&cam, xy[2 * i], xy[2 * i + 1], pixel_width, pixel_height, &re);
        v3_st(out_ray + 6 * i, r.o);
        v3_st(out_ray + 6 * i + 3, r.d);
        out_states[i] = re.s;
    }
}

void oracle_scene_intersect(void *h, uint64_t n, const float *rays, float *out_t, int32_t *out_obj) {
    scene_t *s = (scene_t *)h;
    oracle_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    for(uint64_t i = 0; i < n; i++) {
        hit_t hit = scene_intersect(s, ray_ld(rays + 6 * i), &cnt);
        out_t[i] = hit.t;
        out_obj[i] = hit.obj;
    }
    counters_merge(s, &cnt);
}

void oracle_scene_sample_lights(void *h, uint64_t n, const float *pos, const uint64_t *states, int max_lights, int32_t *out_count, float *out_pos,
                                float *out_rgba, float *out_pd, uint64_t *out_states) {
    scene_t *s = (scene_t *)h;
    light_sample_t lights[ORACLE_MAX_LIGHTS];
    for(uint64_t i = 0; i < n; i++) {
        rng_t re = {states[i]};
        int cnt = scene_sample_lights(s, v3_ld(pos + 3 * i), &re, lights, ORACLE_MAX_LIGHTS);
        out_count[i] = cnt;
        for(int j = 0; j < cnt && j < max_lights && j < ORACLE_MAX_LIGHTS; j++) {
            size_t o = i * (size_t)max_lights + (size_t)j;
            v3_st(out_pos + 3 * o, lights[j].pos);
            memcpy(out_rgba + 4 * o, lights[j].spectrum.e, 16);
            out_pd[o] = lights[j].pd;
        }
        out_states[i] = re.s;
    }
}

static void dump_bvh(const scene_t *s, int node, int32_t *out_obj, float *out_box, size_t *pos) {
    size_t me = (*pos)++;
    const node_t *nd = &s->nodes[node];
    v3_st(out_box + 6 * me, nd->lo);
    v3_st(out_box + 6 * me + 3, nd->hi);
    if(nd->left < 0) {
        out_obj[me] = nd->obj;
    }
    else {
        out_obj[me] = -1;
        dump_bvh(s, nd->left, out_obj, out_box, pos);
        dump_bvh(s, nd->right, out_obj, out_box, pos);
    }
}

uint64_t oracle_bvh_dump(const pto_scene_desc *d, int32_t *out_obj, float *out_box) {
    if(d->n_objects == 0) {
        return 0;
    }
    scene_t *s = (scene_t *)oracle_scene_create(d);
    size_t pos = 0;
    dump_bvh(s, s->root, out_obj, out_box, &pos);
    oracle_scene_destroy(s);
    return pos;
}

void oracle_get_sample(void *h, const pto_camera_params *cp, const pto_options *op, uint64_t n, const float *xy_camera, const uint64_t *states,
                       float *out_rgba, uint8_t *out_collected, uint64_t *out_states) {
    scene_t *s = (scene_t *)h;
    camera_t cam = camera_make(cp);
    oracle_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    for(uint64_t i = 0; i < n; i++) {
        rng_t re = {states[i]};
        int collected;
        c4 c = get_sample(s, &cam, op, xy_camera[2 * i], xy_camera[2 * i + 1], &re, &collected, &cnt);
        memcpy(out_rgba + 4 * i, c.e, 16);
        out_collected[i] = collected ? 1 : 0;
        out_states[i] = re.s;
    }
    counters_merge(s, &cnt);
}

typedef struct {
    scene_t *s;
    camera_t cam;
    const pto_options *opt;
    const pto_stream *streams;
    uint64_t n;
    float *out_image;
    uint64_t *out_states;
    uint64_t next;
    pthread_mutex_t lock;
} render_job_t;

static void *render_worker(void *arg) {
    render_job_t *job = (render_job_t *)arg;
    oracle_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    for(;;) {
        pthread_mutex_lock(&job->lock);
        uint64_t begin = job->next;
        job->next += 64;
        pthread_mutex_unlock(&job->lock);
        if(begin >= job->n) {
            break;
        }
        uint64_t end = begin + 64 < job->n ? begin + 64 : job->n;
        for(uint64_t i = begin; i < end; i++) {
            rng_t re = {job->streams[i].rng_state};
            process_item(job->s, &job->cam, job->opt, &job->streams[i], &re, job->out_image, &cnt);
            if(job->out_states != NULL) {
                job->out_states[i] = re.s;
            }
        }
    }
    counters_merge(job->s, &cnt);
    return NULL;
}

void oracle_render_streams(void *h, const pto_camera_params *cp, const pto_options *op, const pto_stream *streams, uint64_t n, float *out_image,
                           uint64_t *out_states, int n_threads) {
    render_job_t job;
    job.s = (scene_t *)h;
    job.cam = camera_make(cp);
    job.opt = op;
    job.streams = streams;
    job.n = n;
    job.out_image = out_image;
    job.out_states = out_states;
    job.next = 0;
    pthread_mutex_init(&job.lock, NULL);
    if(n_threads <= 1) {
        render_worker(&job);
    }
    else {
        pthread_t *threads = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)(n_threads - 1));
        for(int t = 0; t < n_threads - 1; t++) {
            pthread_create(&threads[t], NULL, render_worker, &job);
        }
        render_worker(&job);
        for(int t = 0; t < n_threads - 1; t++) {
            pthread_join(threads[t], NULL);
        }
        free(threads);
    }
    pthread_mutex_destroy(&job.lock);
}

/* ---- post-processing: toneMap / gammaCorrect (src/post_processing.cpp:11-187), sequential restatement ------------------------------ */

static float post_peak(const float *c) { /* std::max({r, g, b}) */
    float m = c[0];
    m = (m < c[1]) ? c[1] : m;
    m = (m < c[2]) ? c[2] : m;
    return m;
}

static float post_heuristic(const float *c) { /* getBrightnessHeuristic, :27-30 */
    return c[3] * ((c[0] + c[1] + c[2]) / 3.0F + post_peak(c)) / 2.0F;
}

static int post_cmp(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

static float post_gaussian(float t, float mu, float sigma) { /* :11-20 */
    const float pi = (float)M_PI;
    const float fac = 1.0F / (sqrtf(2 * pi));
    const float exponent_part = (t - mu) / (sigma);
    return fac * expf(-(exponent_part * exponent_part) / 2.0F) / sigma;
}

static void post_tone_map(float *rgba, int width, int height) { /* :32-166 */
    const int pixel_count = width * height;
    if(pixel_count <= 0) {
        return;
    }
    float min_brightness = 0.0F, max_brightness = 1E-4F;
    float *values = (float *)malloc(sizeof(float) * (size_t)pixel_count);
    for(int i = 0; i < pixel_count; i++) {
        const float b = post_heuristic(rgba + 4 * (size_t)i);
        values[i] = b;
        min_brightness = (b < min_brightness) ? b : min_brightness;
        max_brightness = (max_brightness < b) ? b : max_brightness;
    }
    /* buckets of equal width in brightness, each sorted, concatenated (:49-88) == the sorted list */
    qsort(values, (size_t)pixel_count, sizeof(float), post_cmp);

    const int segments = pixel_count < 1024 ? pixel_count : 1024;
    float *weights = (float *)malloc(sizeof(float) * (size_t)segments);
    float *ceilings = (float *)malloc(sizeof(float) * (size_t)segments);
    float total = 0.0F;
    for(int i = 0; i < segments; i++) {
        float x = ((float)i + 0.5F) / (float)segments;
        x = 2.0F * (x - 0.5F);
        weights[i] = 0.1F + post_gaussian(x, 0.0F, 0.3F);
        total += weights[i];
    }
    int previous_index = 0;
    float missed = 0.0F;
    for(int i = 0; i < segments - 1; i++) {
        const int count = (int)roundf(weights[i] * (float)pixel_count / total + missed);
        if(count > 0) {
            int index = previous_index + count - 1;
            if(index > pixel_count - 1) {
                index = pixel_count - 1;
            }
            ceilings[i] = values[index];
            previous_index += count;
            missed = 0.0F;
        }
        else {
            ceilings[i] = i > 0 ? ceilings[i - 1] : min_brightness;
            missed += weights[i] * (float)pixel_count / total;
        }
    }
    ceilings[segments - 1] = max_brightness;

    const float tiny = 1.17549435e-38F;
    for(int i = 0; i < pixel_count; i++) {
        float *c = rgba + 4 * (size_t)i;
        const float peak = post_peak(c);
        const float brightness = (peak < tiny) ? tiny : peak;
        const float h = post_heuristic(c);
        int lo = 0, len = segments; /* std::lower_bound */
        while(len > 0) {
            const int half = len >> 1;
            if(ceilings[lo + half] < h) {
                lo = lo + half + 1;
                len = len - half - 1;
            }
            else {
                len = half;
            }
        }
        const int index = lo < segments ? lo : segments - 1;
        const float upper = ceilings[index];
        const float lower = index > 0 ? ceilings[index - 1] : min_brightness;
        const float diff = upper - lower;
        const float span = (diff < tiny) ? tiny : diff;
        const float value = (h - lower) / span;
        const float mapped_upper = (float)(index + 1) / (float)segments;
        const float mapped_lower = (float)index / (float)segments;
        const float mapped_span = mapped_upper - mapped_lower;
        const float mapped_value = mapped_lower + value * mapped_span;
        const float factor = mapped_value / brightness;
        c[0] *= factor;
        c[1] *= factor;
        c[2] *= factor;
    }
    free(values);
    free(weights);
    free(ceilings);
}

static void post_gamma(float *rgba, int width, int height, float gamma) { /* :171-182 */
    for(long long i = 0; i < (long long)width * height; i++) {
        float *c = rgba + 4 * (size_t)i;
        const float factor = powf(post_peak(c), 1.0F / gamma - 1.0F);
        c[0] *= factor;
        c[1] *= factor;
        c[2] *= factor;
    }
}

/* steps: 1 = toneMap, 2 = gammaCorrect(gamma), 3 = postProcess (gamma 1.8, post_processing.h:22) */
void oracle_post_process(float *rgba, int width, int height, int steps, float gamma) {
    if(steps == 3) {
        post_tone_map(rgba, width, height);
        post_gamma(rgba, width, height, 1.8F);
    }
    else if(steps == 1) {
        post_tone_map(rgba, width, height);
    }
    else if(steps == 2) {
        post_gamma(rgba, width, height, gamma);
    }
}
