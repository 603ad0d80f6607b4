/*
 * oracle/ref_shim.cpp -- C entry points around the UNMODIFIED reference sources.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is compiled together with /root/reference/src/{worker,camera}.cpp and
 * /root/reference/src/scene/{scene,object,bounding_box,light,material,propagation,mesh}.cpp by oracle/Makefile
 * (target oracle/_ref/libptref.so).  It contains no algorithm of its own: every function builds reference objects
 * through the reference's public constructors and forwards to the reference's own functions, so that
 * tests/golden/make_golden.py can record what the reference computes and bench.py can time it
 * (cpu_baseline.kind == "reference").  Nothing in the product loads this library.
 */
#include <PathTrace/post_processing.h>
#include <PathTrace/base.h>
#include <PathTrace/camera.h>
#include <PathTrace/worker.h>
#include <PathTrace/scene/scene.h>
#include <PathTrace/scene/object.h>
#include <PathTrace/scene/bounding_box.h>
#include <PathTrace/scene/light.h>
#include <PathTrace/scene/material.h>
#include <PathTrace/scene/mesh.h>
#include <PathTrace/scene/propagation.h>

#include "pt_desc.h"

#include <atomic>
#include <cstring>
#include <memory>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <vector>

// Functions with external linkage in the reference's named namespace `impl` (src/worker.cpp:26, src/scene/scene.cpp:12).
namespace impl {
    std::tuple<Spectrum, bool> getSample(const WorkItem &item, float x_camera, float y_camera, RandomEngine &re);
    AABB constructBVH(std::vector<AABB> &&bounding_boxes);
}

namespace {

    // A Material with every property constant, including the specular colour that ConstantMaterial cannot set.
    class ShimMaterial final : public Material {
      public:
        Color<float> diffuse, specular;
        float ior;
        Spectrum emission;
        ShimMaterial(const pto_material &m) :
          diffuse(m.diffuse[0], m.diffuse[1], m.diffuse[2], m.diffuse[3]), specular(m.specular[0], m.specular[1], m.specular[2], m.specular[3]), ior(m.ior),
          emission(Color<float>(m.emission[0], m.emission[1], m.emission[2], m.emission[3])) {}
        Color<float> getDiffuseColor(vec3<float>) const noexcept override { return diffuse; }
        Color<float> getSpecularColor(vec3<float>) const noexcept override { return specular; }
        float getRefractiveIndex(vec3<float>) const noexcept override { return ior; }
        Spectrum getEmission(Ray, vec3<float>) const noexcept override { return emission; }
        Spectrum probeEmission() const noexcept override { return emission; }
    };

    bool isWhite(const float *c) { return c[0] == 1.0F && c[1] == 1.0F && c[2] == 1.0F && c[3] == 1.0F; }

    std::shared_ptr<Material> makeMaterial(const pto_material &m) {
        if(isWhite(m.specular)) {
            return std::make_shared<ConstantMaterial>(Color<float>(m.diffuse[0], m.diffuse[1], m.diffuse[2], m.diffuse[3]), m.ior,
                                                      Spectrum(Color<float>(m.emission[0], m.emission[1], m.emission[2], m.emission[3])));
        }
        return std::make_shared<ShimMaterial>(m);
    }

    std::shared_ptr<BSDF> makeBSDF(int kind, int one_way) {
        switch(kind) {
            case PTO_BSDF_GLASS:
                return std::make_shared<GlassBDF>();
            case PTO_BSDF_MIRROR:
                return std::make_shared<MirrorBRDF>(one_way != 0);
            default:
                return std::make_shared<LambertianBRDF>();
        }
    }

    vec3<float> v3(const float *p) { return vec3<float>(p[0], p[1], p[2]); }

    Triangle makeTriangle(const float *pos, const float *nrm, bool cull) {
        Triangle t(v3(pos), v3(pos + 3), v3(pos + 6), cull);
        if(nrm != nullptr) {
            t.normal_a = v3(nrm);
            t.normal_b = v3(nrm + 3);
            t.normal_c = v3(nrm + 6);
        }
        return t;
    }

    std::vector<std::unique_ptr<Object>> buildObjects(const pto_scene_desc *d, std::unordered_map<const Object *, int> *index) {
        std::vector<std::shared_ptr<MaterialHandler>> handlers;
        for(uint32_t i = 0; i < d->n_materials; i++) {
            const auto &m = d->materials[i];
            handlers.push_back(std::make_shared<ConstantMaterialHandler>(makeMaterial(m), makeBSDF(m.bsdf, m.one_way)));
        }

        std::vector<std::unique_ptr<Object>> objects;
        objects.reserve(d->n_objects);
        uint32_t ti = 0;
        uint32_t si = 0;
        for(uint32_t i = 0; i < d->n_objects; i++) {
            std::unique_ptr<Object> obj;
            uint32_t mat;
            if(d->obj_kind[i] == PTO_OBJ_TRIANGLE) {
                obj = std::make_unique<Triangle>(
                  makeTriangle(d->tri_pos + 9 * static_cast<size_t>(ti), d->tri_nrm ? d->tri_nrm + 9 * static_cast<size_t>(ti) : nullptr, d->tri_cull[ti] != 0));
                mat = d->tri_material[ti];
                ti++;
            }
            else {
                const float *s = d->sph + 4 * static_cast<size_t>(si);
                obj = std::make_unique<Sphere>(v3(s), s[3]);
                mat = d->sph_material[si];
                si++;
            }
            // material index 0xFFFFFFFF keeps the reference's default handler (object.cpp:9-11,32)
            if(mat != 0xFFFFFFFFU) {
                obj->setMaterialHandler(handlers[mat]);
            }
            if(index != nullptr) {
                (*index)[obj.get()] = static_cast<int>(i);
            }
            objects.emplace_back(std::move(obj));
        }
        return objects;
    }

    std::vector<std::unique_ptr<LightSource>> buildLights(const pto_scene_desc *d) {
        std::vector<std::unique_ptr<LightSource>> lights;
        for(uint32_t i = 0; i < d->n_point_lights; i++) {
            const float *p = d->light_pos + 3 * i;
            const float *s = d->light_spectrum + 4 * i;
            lights.emplace_back(std::make_unique<PointLightSource>(v3(p), Spectrum(Color<float>(s[0], s[1], s[2], s[3]))));
        }
        return lights;
    }

    struct RefScene {
        std::unordered_map<const Object *, int> index;
        std::unique_ptr<Scene> scene;
    };

    Camera makeCamera(const pto_camera_params *c) {
        std::unique_ptr<ApertureSampler> sampler;
        if(c->aperture_kind == PTO_APERTURE_CIRCULAR) {
            sampler = std::make_unique<CircularApertureSampler>();
        }
        else if(c->aperture_kind == PTO_APERTURE_HEXAGONAL) {
            sampler = std::make_unique<HexagonalApertureSampler>(c->hex_ratio);
        }
        if(c->aperture_kind == PTO_APERTURE_NONE && c->focal_plane_dist == 0.0F && c->aperture_width == 0.0F && c->aperture_height == 0.0F) {
            return Camera(v3(c->origin), v3(c->look_at), v3(c->up), c->focal_length, c->height, c->aspect_ratio);
        }
        return Camera(v3(c->origin), v3(c->look_at), v3(c->up), c->focal_length, c->height, c->aspect_ratio, c->aperture_width, c->aperture_height,
                      std::move(sampler), c->focal_plane_dist);
    }

    RenderOptions makeOptions(const pto_options *o) {
        return RenderOptions{o->image_width, o->image_height, o->min_sample_count, o->max_sample_count, o->epsilon};
    }

    // RandomEngine(seed) has raw state seed ^ (~seed << 32) (base.h:26); invert that to start from a raw state.
    RandomEngine engineFromState(uint64_t state) {
        uint64_t lo = state & 0xFFFFFFFFULL;
        uint64_t hi = (state >> 32) ^ (~lo & 0xFFFFFFFFULL);
        uint64_t seed = (hi << 32) | lo;
        return RandomEngine(seed);
    }

    uint64_t engineState(const RandomEngine &re) {
        static_assert(sizeof(RandomEngine) == sizeof(uint64_t), "RandomEngine is one xorshift word");
        uint64_t s;
        std::memcpy(&s, &re, sizeof(s));
        return s;
    }

    void dumpBVH(const AABB &node, const std::unordered_map<const Object *, int> &index, int32_t *out_obj, float *out_box, size_t &pos) {
        size_t me = pos++;
        for(int k = 0; k < 3; k++) {
            out_box[me * 6 + k] = node.area.low[k];
            out_box[me * 6 + 3 + k] = node.area.high[k];
        }
        if(node.leaf) {
            auto it = index.find(node.child.get());
            out_obj[me] = it == index.end() ? -2 : it->second; // -2: NullObject leaf of an empty scene
        }
        else {
            out_obj[me] = -1;
            dumpBVH(*node.left, index, out_obj, out_box, pos);
            dumpBVH(*node.right, index, out_obj, out_box, pos);
        }
    }

}

extern "C" {

// ---- a-1..a-3: RNG and libstdc++ distributions ---------------------------------------------------------------------

void ref_rng_draws(uint64_t seed, uint64_t n, uint32_t *out) {
    RandomEngine re(seed);
    for(uint64_t i = 0; i < n; i++) {
        out[i] = re();
    }
}

uint64_t ref_rng_state_after(uint64_t seed, uint64_t n_draws) {
    RandomEngine re(seed);
    for(uint64_t i = 0; i < n_draws; i++) {
        re();
    }
    return engineState(re);
}

void ref_uniform_floats(uint64_t seed, float a, float b, uint64_t n, float *out) {
    RandomEngine re(seed);
    std::uniform_real_distribution<float> dist(a, b);
    for(uint64_t i = 0; i < n; i++) {
        out[i] = dist(re);
    }
}

// out_flags[i] = bernoulli(p)(re); returns the engine state after the n decisions
uint64_t ref_bernoulli(uint64_t seed, double p, uint64_t n, uint8_t *out_flags) {
    RandomEngine re(seed);
    std::bernoulli_distribution d(p);
    for(uint64_t i = 0; i < n; i++) {
        out_flags[i] = d(re) ? 1 : 0;
    }
    return engineState(re);
}

// ---- a-4: AABB slab test -------------------------------------------------------------------------------------------

void ref_aabb_intersect(uint64_t n, const float *boxes, const float *rays, float *out_t) {
    for(uint64_t i = 0; i < n; i++) {
        const float *b = boxes + 6 * i;
        const float *r = rays + 6 * i;
        AABB aabb(AABBArea{v3(b), v3(b + 3)}, std::make_unique<NullObject>());
        out_t[i] = aabb.getIntersection(Ray{v3(r), v3(r + 3)});
    }
}

// ---- a-6..a-8: primitives ------------------------------------------------------------------------------------------

void ref_tri_intersect(uint64_t n, const float *tri, const uint8_t *cull, const float *rays, float *out_t) {
    for(uint64_t i = 0; i < n; i++) {
        Triangle t = makeTriangle(tri + 9 * i, nullptr, cull[i] != 0);
        const float *r = rays + 6 * i;
        out_t[i] = t.getIntersection(Ray{v3(r), v3(r + 3)});
    }
}

void ref_tri_normal(uint64_t n, const float *tri, const float *nrm, const float *pos, float *out_n) {
    for(uint64_t i = 0; i < n; i++) {
        Triangle t = makeTriangle(tri + 9 * i, nrm + 9 * i, false);
        auto nn = t.getSurfaceNormal(v3(pos + 3 * i));
        out_n[3 * i + 0] = nn[0];
        out_n[3 * i + 1] = nn[1];
        out_n[3 * i + 2] = nn[2];
    }
}

void ref_tri_props(uint64_t n, const float *tri, float *out_area, float *out_box, float *out_face_normal) {
    for(uint64_t i = 0; i < n; i++) {
        Triangle t = makeTriangle(tri + 9 * i, nullptr, false);
        out_area[i] = t.getSurfaceArea();
        auto bv = t.getBoundingVolume();
        for(int k = 0; k < 3; k++) {
            out_box[6 * i + k] = bv.low[k];
            out_box[6 * i + 3 + k] = bv.high[k];
            out_face_normal[3 * i + k] = t.normal_a[k];
        }
    }
}

void ref_tri_sample(uint64_t n, const float *tri, const uint8_t *cull, const uint64_t *states, float *out_pos, float *out_p, uint8_t *out_cull,
                    uint64_t *out_states) {
    for(uint64_t i = 0; i < n; i++) {
        Triangle t = makeTriangle(tri + 9 * i, nullptr, cull[i] != 0);
        RandomEngine re = engineFromState(states[i]);
        auto [pos, p, c] = t.sampleSurface(re);
        for(int k = 0; k < 3; k++) {
            out_pos[3 * i + k] = pos[k];
        }
        out_p[i] = p;
        out_cull[i] = c ? 1 : 0;
        out_states[i] = engineState(re);
    }
}

void ref_sphere_intersect(uint64_t n, const float *sph, const float *rays, float *out_t) {
    for(uint64_t i = 0; i < n; i++) {
        Sphere s(v3(sph + 4 * i), sph[4 * i + 3]);
        const float *r = rays + 6 * i;
        out_t[i] = s.getIntersection(Ray{v3(r), v3(r + 3)});
    }
}

void ref_sphere_normal(uint64_t n, const float *sph, const float *pos, float *out_n) {
    for(uint64_t i = 0; i < n; i++) {
        Sphere s(v3(sph + 4 * i), sph[4 * i + 3]);
        auto nn = s.getSurfaceNormal(v3(pos + 3 * i));
        for(int k = 0; k < 3; k++) {
            out_n[3 * i + k] = nn[k];
        }
    }
}

void ref_sphere_props(uint64_t n, const float *sph, float *out_area, float *out_box) {
    for(uint64_t i = 0; i < n; i++) {
        Sphere s(v3(sph + 4 * i), sph[4 * i + 3]);
        out_area[i] = s.getSurfaceArea();
        auto bv = s.getBoundingVolume();
        for(int k = 0; k < 3; k++) {
            out_box[6 * i + k] = bv.low[k];
            out_box[6 * i + 3 + k] = bv.high[k];
        }
    }
}

void ref_sphere_sample(uint64_t n, const float *sph, const uint64_t *states, float *out_pos, float *out_p, uint64_t *out_states) {
    for(uint64_t i = 0; i < n; i++) {
        Sphere s(v3(sph + 4 * i), sph[4 * i + 3]);
        RandomEngine re = engineFromState(states[i]);
        auto [pos, p, c] = s.sampleSurface(re);
        (void)c;
        for(int k = 0; k < 3; k++) {
            out_pos[3 * i + k] = pos[k];
        }
        out_p[i] = p;
        out_states[i] = engineState(re);
    }
}

// ---- a-13..a-15: BSDFs ---------------------------------------------------------------------------------------------

void ref_bsdf_propagate(int kind, int one_way, uint64_t n, const float *rays, const float *pos, const float *nrm, float epsilon, const float *ior,
                        const uint64_t *states, float *out_ray, float *out_factor, float *out_pd, uint64_t *out_states) {
    auto bsdf = makeBSDF(kind, one_way);
    for(uint64_t i = 0; i < n; i++) {
        ConstantMaterial mat(Color<float>(1.0F, 1.0F, 1.0F, 1.0F), ior[i]);
        RandomEngine re = engineFromState(states[i]);
        const float *r = rays + 6 * i;
        auto [out, factor, pd] = bsdf->propagateRay(Ray{v3(r), v3(r + 3)}, v3(pos + 3 * i), v3(nrm + 3 * i), epsilon, re, &mat);
        for(int k = 0; k < 3; k++) {
            out_ray[6 * i + k] = out.origin[k];
            out_ray[6 * i + 3 + k] = out.dir[k];
        }
        out_factor[i] = factor;
        out_pd[i] = pd;
        out_states[i] = engineState(re);
    }
}

void ref_bsdf_spectrum(int kind, int one_way, uint64_t n, const float *from_dir, const float *to_dir, const float *nrm, const float *light_rgba,
                       const float *diffuse, const float *specular, int synthetic, float *out_rgba, float *out_shade, float *out_p) {
    auto bsdf = makeBSDF(kind, one_way);
    for(uint64_t i = 0; i < n; i++) {
        pto_material m{};
        std::memcpy(m.diffuse, diffuse + 4 * i, 16);
        std::memcpy(m.specular, specular + 4 * i, 16);
        m.ior = 1.0F;
        ShimMaterial mat(m);
        const float *l = light_rgba + 4 * i;
        vec3<float> zero(0.0F, 0.0F, 0.0F);
        auto [spec, shade, p] = bsdf->getSpectrum(Ray{zero, v3(from_dir + 3 * i)}, Ray{zero, v3(to_dir + 3 * i)}, zero, v3(nrm + 3 * i),
                                                  Spectrum(Color<float>(l[0], l[1], l[2], l[3])), &mat, synthetic != 0);
        auto c = spec.getColor();
        for(int k = 0; k < 4; k++) {
            out_rgba[4 * i + k] = c[k];
        }
        out_shade[i] = shade;
        out_p[i] = p;
    }
}

// ---- a-18: camera --------------------------------------------------------------------------------------------------

void ref_camera_shoot(const pto_camera_params *cp, uint64_t n, const float *xy, float pixel_width, float pixel_height, const uint64_t *states,
                      float *out_ray, uint64_t *out_states) {
    Camera camera = makeCamera(cp);
    for(uint64_t i = 0; i < n; i++) {
        RandomEngine re = engineFromState(states[i]);
        Ray r = camera.shootRay(xy[2 * i], xy[2 * i + 1], pixel_width, pixel_height, re);
        for(int k = 0; k < 3; k++) {
            out_ray[6 * i + k] = r.origin[k];
            out_ray[6 * i + 3 + k] = r.dir[k];
        }
        out_states[i] = engineState(re);
    }
}

// ---- a-5, a-10, a-11: scene ----------------------------------------------------------------------------------------

void *ref_scene_create(const pto_scene_desc *d) {
    auto *rs = new RefScene();
    auto objects = buildObjects(d, &rs->index);
    rs->scene = std::make_unique<Scene>(std::move(objects), buildLights(d));
    return rs;
}

void ref_scene_destroy(void *h) {
    delete static_cast<RefScene *>(h);
}

void ref_scene_intersect(void *h, uint64_t n, const float *rays, float *out_t, int32_t *out_obj) {
    auto *rs = static_cast<RefScene *>(h);
    for(uint64_t i = 0; i < n; i++) {
        const float *r = rays + 6 * i;
        auto [t, obj] = rs->scene->getIntersection(Ray{v3(r), v3(r + 3)});
        out_t[i] = t;
        if(obj == nullptr) {
            out_obj[i] = -1;
        }
        else {
            auto it = rs->index.find(obj);
            out_obj[i] = it == rs->index.end() ? -2 : it->second;
        }
    }
}

// Scene::sampleLights for n positions; at most max_lights tuples per call are stored.
void ref_scene_sample_lights(void *h, uint64_t n, const float *pos, const uint64_t *states, int max_lights, int32_t *out_count, float *out_pos,
                             float *out_rgba, float *out_pd, uint64_t *out_states) {
    auto *rs = static_cast<RefScene *>(h);
    for(uint64_t i = 0; i < n; i++) {
        RandomEngine re = engineFromState(states[i]);
        auto lights = rs->scene->sampleLights(v3(pos + 3 * i), vec3<float>(0.0F, 1.0F, 0.0F), re);
        out_count[i] = static_cast<int32_t>(lights.size());
        for(int j = 0; j < static_cast<int>(lights.size()) && j < max_lights; j++) {
            const auto &[lp, ls, lpd] = lights[j];
            auto c = ls.getColor();
            size_t o = i * max_lights + j;
            for(int k = 0; k < 3; k++) {
                out_pos[3 * o + k] = lp[k];
            }
            for(int k = 0; k < 4; k++) {
                out_rgba[4 * o + k] = c[k];
            }
            out_pd[o] = lpd;
        }
        out_states[i] = engineState(re);
    }
}

// Pre-order dump of impl::constructBVH over the same leaves Scene::Scene builds (scene.cpp:156-162).
// out_obj[i] = object index for a leaf, -1 for an inner node; out_box[i] = low xyz, high xyz.  Returns node count.
uint64_t ref_bvh_dump(const pto_scene_desc *d, int32_t *out_obj, float *out_box) {
    std::unordered_map<const Object *, int> index;
    auto objects = buildObjects(d, &index);
    std::vector<AABB> aabbs;
    aabbs.reserve(objects.size());
    for(auto &object : objects) {
        aabbs.emplace_back(object->getBoundingVolume(), std::move(object));
    }
    AABB root = impl::constructBVH(std::move(aabbs));
    size_t pos = 0;
    if(d->n_objects == 0) {
        return 0;
    }
    dumpBVH(root, index, out_obj, out_box, pos);
    return pos;
}

// ---- a-12: one path ------------------------------------------------------------------------------------------------

void ref_get_sample(void *h, const pto_camera_params *cp, const pto_options *op, uint64_t n, const float *xy_camera, const uint64_t *states,
                    float *out_rgba, uint8_t *out_collected, uint64_t *out_states) {
    auto *rs = static_cast<RefScene *>(h);
    Camera camera = makeCamera(cp);
    RenderOptions options = makeOptions(op);
    FrameRenderJob job{camera, *rs->scene, options};
    WorkItem item(&job, 0, 0, 1, 1);
    for(uint64_t i = 0; i < n; i++) {
        RandomEngine re = engineFromState(states[i]);
        auto [spec, collected] = impl::getSample(item, xy_camera[2 * i], xy_camera[2 * i + 1], re);
        auto c = spec.getColor();
        for(int k = 0; k < 4; k++) {
            out_rgba[4 * i + k] = c[k];
        }
        out_collected[i] = collected ? 1 : 0;
        out_states[i] = engineState(re);
    }
}

// ---- a-19, a-20: processItem on a list of streams --------------------------------------------------------------------
// Each stream is processItem(WorkItem(job, x, y, w, h), engine(rng_state)); tiles are written into the full
// row-major image exactly as doWork does (worker.cpp:348-352).  Streams are distributed over n_threads workers.
void ref_render_streams(void *h, const pto_camera_params *cp, const pto_options *op, const pto_stream *streams, uint64_t n, float *out_image,
                        uint64_t *out_states, int n_threads) {
    auto *rs = static_cast<RefScene *>(h);
    Camera camera = makeCamera(cp);
    RenderOptions options = makeOptions(op);
    FrameRenderJob job{camera, *rs->scene, options};
    const int width = options.image_width;

    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        for(;;) {
            // grab streams in chunks to keep the atomic off the critical path for 1x1 streams
            uint64_t begin = next.fetch_add(64, std::memory_order_relaxed);
            if(begin >= n) {
                break;
            }
            uint64_t end = std::min<uint64_t>(begin + 64, n);
            for(uint64_t i = begin; i < end; i++) {
                const auto &s = streams[i];
                RandomEngine re = engineFromState(s.rng_state);
                WorkItem item(&job, s.x, s.y, s.w, s.h);
                Image<> tile = processItem(item, re);
                for(int y = 0; y < s.h; y++) {
                    for(int x = 0; x < s.w; x++) {
                        auto c = tile(x, y);
                        float *o = out_image + 4 * (static_cast<size_t>(s.y + y) * width + (s.x + x));
                        o[0] = c[0];
                        o[1] = c[1];
                        o[2] = c[2];
                        o[3] = c[3];
                    }
                }
                if(out_states != nullptr) {
                    out_states[i] = engineState(re);
                }
            }
        }
    };

    if(n_threads <= 1) {
        work();
        return;
    }
    std::vector<std::thread> threads;
    for(int t = 0; t < n_threads - 1; t++) {
        threads.emplace_back(work);
    }
    work();
    for(auto &t : threads) {
        t.join();
    }
}

// ---- a-14 helpers exposed for unit fixtures ------------------------------------------------------------------------

// makePlane / makeBox / io::loadMesh pass-throughs so scenes in fixtures are built by the reference's own generators.
// Each returns the triangle count and fills pos/nrm (9 floats per triangle) up to `capacity` triangles.
static uint64_t storeTriangles(const std::vector<Triangle> &tris, uint64_t capacity, float *pos, float *nrm) {
    for(uint64_t i = 0; i < tris.size() && i < capacity; i++) {
        const Triangle &t = tris[i];
        for(int k = 0; k < 3; k++) {
            pos[9 * i + k] = t.a[k];
            pos[9 * i + 3 + k] = t.b[k];
            pos[9 * i + 6 + k] = t.c[k];
            nrm[9 * i + k] = t.normal_a[k];
            nrm[9 * i + 3 + k] = t.normal_b[k];
            nrm[9 * i + 6 + k] = t.normal_c[k];
        }
    }
    return tris.size();
}

uint64_t ref_make_plane(const float *a, const float *b, uint64_t capacity, float *pos, float *nrm) {
    return storeTriangles(makePlane(v3(a), v3(b), false), capacity, pos, nrm);
}

uint64_t ref_make_box(const float *a, const float *b, uint64_t capacity, float *pos, float *nrm) {
    return storeTriangles(makeBox(v3(a), v3(b), false), capacity, pos, nrm);
}

uint64_t ref_load_mesh(const char *obj_text, uint64_t len, const float *mat16, int smooth, uint64_t capacity, float *pos, float *nrm) {
    std::istringstream stream(std::string(obj_text, len));
    mat4<float> m{};
    for(int r = 0; r < 4; r++) {
        for(int c = 0; c < 4; c++) {
            m.rows[r][c] = mat16[4 * r + c];
        }
    }
    return storeTriangles(io::loadMesh(stream, m, false, smooth != 0), capacity, pos, nrm);
}

void ref_mat4_apply(const float *mat16, uint64_t n, const float *in, float *out) {
    mat4<float> m{};
    for(int r = 0; r < 4; r++) {
        for(int c = 0; c < 4; c++) {
            m.rows[r][c] = mat16[4 * r + c];
        }
    }
    for(uint64_t i = 0; i < n; i++) {
        auto v = m * v3(in + 3 * i);
        out[3 * i + 0] = v[0];
        out[3 * i + 1] = v[1];
        out[3 * i + 2] = v[2];
    }
}

// steps: 1 = toneMap, 2 = gammaCorrect(gamma), 3 = postProcess (which uses gammaCorrect's default gamma, post_processing.h:22)
void ref_post_process(float *rgba, int width, int height, int steps, float gamma) {
    Image<> image(width, height);
    for(int y = 0; y < height; y++) {
        for(int x = 0; x < width; x++) {
            const float *p = rgba + 4 * (static_cast<size_t>(y) * width + x);
            image(x, y) = Color<float>(p[0], p[1], p[2], p[3]);
        }
    }
    if(steps == 3) {
        postProcess(image);
    }
    else if(steps == 1) {
        toneMap(image);
    }
    else if(steps == 2) {
        gammaCorrect(image, gamma);
    }
    for(int y = 0; y < height; y++) {
        for(int x = 0; x < width; x++) {
            float *p = rgba + 4 * (static_cast<size_t>(y) * width + x);
            for(int c = 0; c < 4; c++) {
                p[c] = image(x, y)[c];
            }
        }
    }
}

}
