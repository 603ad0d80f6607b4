// src/host/scene.cpp -- Scene: flattens the caller's object graph into the arrays of include/pt_hip.h and owns the device scene.
#include <PathTrace/detail/world.h>

#include "../../include/pt_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <random>
#include <stdexcept>
#include <string>

namespace {

    void copy4(float *dst, const Color<float> &c) {
        for(int k = 0; k < 4; k++) {
            dst[k] = c[k];
        }
    }

    // index of the pt_material that describes `handler`, appending it on first use
    uint32_t materialIndex(const MaterialHandler *handler, std::map<const MaterialHandler *, uint32_t> &known, std::vector<pt_material> &materials) {
        auto found = known.find(handler);
        if(found != known.end()) {
            return found->second;
        }
        const auto *constant = dynamic_cast<const ConstantMaterialHandler *>(handler);
        if(constant == nullptr) {
            throw std::invalid_argument("PathTrace: only ConstantMaterialHandler can be evaluated on the device");
        }
        const vec3<float> anywhere{0.0F, 0.0F, 0.0F};
        const Material *material = constant->getMaterial(anywhere);
        const BSDF *bsdf = constant->getBSDF(anywhere);
        if(dynamic_cast<const ConstantMaterial *>(material) == nullptr) {
            throw std::invalid_argument("PathTrace: only ConstantMaterial can be evaluated on the device");
        }
        pt_material m{};
        copy4(m.diffuse, material->getDiffuseColor(anywhere));
        copy4(m.specular, material->getSpecularColor(anywhere));
        copy4(m.emission, material->probeEmission().getColor());
        m.ior = material->getRefractiveIndex(anywhere);
        if(dynamic_cast<const LambertianBRDF *>(bsdf) != nullptr) {
            m.bsdf = PT_BSDF_LAMBERTIAN;
        }
        else if(dynamic_cast<const GlassBDF *>(bsdf) != nullptr) {
            m.bsdf = PT_BSDF_GLASS;
        }
        else if(const auto *mirror = dynamic_cast<const MirrorBRDF *>(bsdf)) {
            m.bsdf = PT_BSDF_MIRROR;
            m.one_way = mirror->isOneWay() ? 1 : 0;
        }
        else {
            throw std::invalid_argument("PathTrace: only LambertianBRDF, GlassBDF and MirrorBRDF can be evaluated on the device");
        }
        const uint32_t index = static_cast<uint32_t>(materials.size());
        materials.push_back(m);
        known.emplace(handler, index);
        return index;
    }

} // namespace

Scene::Scene(std::vector<std::unique_ptr<Object>> &&objs, std::vector<std::unique_ptr<LightSource>> &&lights) :
  objects(std::move(objs)), light_sources(std::move(lights)) {
    std::vector<uint8_t> kinds;
    std::vector<float> tri_pos, tri_nrm, spheres, light_pos, light_spectrum;
    std::vector<uint8_t> tri_cull;
    std::vector<uint32_t> tri_material, sphere_material;
    std::vector<pt_material> materials;
    std::map<const MaterialHandler *, uint32_t> known;
    kinds.reserve(objects.size());

    for(const auto &object : objects) {
        const uint32_t material = materialIndex(object->getMaterialHandler(), known, materials);
        if(const auto *t = dynamic_cast<const Triangle *>(object.get())) {
            kinds.push_back(PT_OBJ_TRIANGLE);
            for(const vec3<float> *v : {&t->a, &t->b, &t->c}) {
                tri_pos.insert(tri_pos.end(), {(*v)[0], (*v)[1], (*v)[2]});
            }
            for(const vec3<float> *v : {&t->normal_a, &t->normal_b, &t->normal_c}) {
                tri_nrm.insert(tri_nrm.end(), {(*v)[0], (*v)[1], (*v)[2]});
            }
            tri_cull.push_back(t->cullsBackface() ? 1 : 0);
            tri_material.push_back(material);
        }
        else if(const auto *s = dynamic_cast<const Sphere *>(object.get())) {
            kinds.push_back(PT_OBJ_SPHERE);
            const auto o = s->getOrigin();
            spheres.insert(spheres.end(), {o[0], o[1], o[2], s->getRadius()});
            sphere_material.push_back(material);
        }
        else {
            throw std::invalid_argument("PathTrace: only Triangle and Sphere objects can be rendered on the device");
        }
    }
    for(const auto &light : light_sources) {
        const auto *point = dynamic_cast<const PointLightSource *>(light.get());
        if(point == nullptr) {
            throw std::invalid_argument("PathTrace: only PointLightSource lights can be rendered on the device");
        }
        const vec3<float> anywhere{0.0F, 0.0F, 0.0F};
        const auto [target, density] = point->importanceSample(anywhere);
        (void)density;
        const auto colour = point->getSpectrum(Ray{anywhere, vec3<float>{0.0F, 0.0F, 1.0F}}).getColor();
        light_pos.insert(light_pos.end(), {target[0], target[1], target[2]});
        light_spectrum.insert(light_spectrum.end(), {colour[0], colour[1], colour[2], colour[3]});
    }

    pt_scene_desc desc{};
    desc.n_objects = static_cast<uint32_t>(kinds.size());
    desc.obj_kind = kinds.data();
    desc.n_triangles = static_cast<uint32_t>(tri_cull.size());
    desc.tri_pos = tri_pos.data();
    desc.tri_nrm = tri_nrm.data();
    desc.tri_cull = tri_cull.data();
    desc.tri_material = tri_material.data();
    desc.n_spheres = static_cast<uint32_t>(sphere_material.size());
    desc.sph = spheres.data();
    desc.sph_material = sphere_material.data();
    desc.n_materials = static_cast<uint32_t>(materials.size());
    desc.materials = materials.data();
    desc.n_point_lights = static_cast<uint32_t>(light_sources.size());
    desc.light_pos = light_pos.data();
    desc.light_spectrum = light_spectrum.data();

    const char *device_env = std::getenv("PATHTRACE_DEVICE");
    const int device = device_env != nullptr ? std::atoi(device_env) : 0;
    if(pt_scene_create(device, &desc, &device_scene) != PT_OK) {
        throw std::runtime_error(std::string("PathTrace: cannot create the device scene: ") + pt_last_error());
    }

    // emissive objects and their cumulative selection probabilities, in the order the hierarchy registers them
    uint64_t n_emissive = 0;
    pt_scene_emissive(device_scene, nullptr, nullptr, 0, &n_emissive);
    std::vector<int32_t> indices(n_emissive);
    emissive_cdf.resize(n_emissive);
    pt_scene_emissive(device_scene, indices.data(), emissive_cdf.data(), n_emissive, nullptr);
    for(int32_t i : indices) {
        emissive.push_back(objects[static_cast<size_t>(i)].get());
    }
}

Scene::~Scene() {
    pt_scene_destroy(device_scene);
}

std::tuple<float, const Object *> Scene::getIntersection(const Ray &ray) const noexcept {
    const float packed[6] = {ray.origin[0], ray.origin[1], ray.origin[2], ray.dir[0], ray.dir[1], ray.dir[2]};
    float t = -1.0F;
    int32_t index = -1;
    if(pt_intersect_batch(device_scene, packed, 1, &t, &index) != PT_OK || index < 0) {
        return {t < 0.0F ? t : -1.0F, nullptr};
    }
    return {t, objects[static_cast<size_t>(index)].get()};
}

std::vector<std::tuple<vec3<float>, Spectrum, float>> Scene::sampleLights(vec3<float> pos, vec3<float> /*n*/, RandomEngine &re) const noexcept {
    std::uniform_real_distribution<float> dist(0, 1);
    const int n_emissive = static_cast<int>(emissive.size());
    const int n_object_samples = std::min(2 + static_cast<int>(std::log10(n_emissive + 1)), n_emissive);

    std::vector<std::tuple<vec3<float>, Spectrum, float>> samples;
    samples.reserve(light_sources.size() + static_cast<size_t>(n_object_samples));
    for(const auto &light : light_sources) {
        const auto [target, density] = light->importanceSample(pos);
        samples.emplace_back(target, light->getSpectrum(Ray{pos, (target - pos).normalize()}), density);
    }
    for(int i = 0; i < n_object_samples; i++) {
        const float pick = dist(re);
        const int chosen = static_cast<int>(std::lower_bound(emissive_cdf.begin(), emissive_cdf.end(), pick) - emissive_cdf.begin());
        float pick_probability = emissive_cdf[chosen];
        if(chosen > 0) {
            pick_probability -= emissive_cdf[chosen - 1];
        }
        pick_probability *= float(n_object_samples);

        const Object *object = emissive[chosen];
        const auto [point, point_density, front_only] = object->sampleSurface(re);
        const auto point_normal = object->getSurfaceNormal(point);
        const auto offset = point - pos;
        const auto dir = offset.normalize();
        const float cosine = std::abs(dot(-dir, point_normal));
        if(!(cosine > 0.0F) || !(offset.getLengthSquared() > 0.0F) || (front_only && !(dot(dir, point_normal) < 0.0F))) {
            continue;
        }
        const float area_to_solid_angle = offset.getLengthSquared() / cosine;
        const Material *material = object->getMaterialHandler()->getMaterial(point);
        samples.emplace_back(point, material->getEmission(Ray{pos, dir}, point), pick_probability * point_density * area_to_solid_angle);
    }
    return samples;
}
