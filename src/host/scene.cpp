// src/host/scene.cpp -- Scene: flattens the caller's object graph into the arrays of include/pt_hip.h and owns the device scene.
#include <PathTrace/detail/world.h>

#include "../../include/pt_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <typeinfo>

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
    std::vector<uint8_t> kinds(objects.size());
    std::vector<float> tri_pos, tri_nrm, spheres, light_pos, light_spectrum;
    std::vector<uint8_t> tri_cull;
    std::vector<uint32_t> tri_material, sphere_material;
    std::vector<pt_material> materials;
    std::map<const MaterialHandler *, uint32_t> known;

    auto onAllCores = [](size_t count, auto body) { // body(first, last) over contiguous slices of [0, count)
        const size_t workers = count < 65536 ? 1 : std::min<size_t>(std::max(1U, std::thread::hardware_concurrency()), 64);
        if(workers == 1) {
            body(size_t{0}, count);
            return;
        }
        std::vector<std::thread> pool;
        for(size_t w = 0; w < workers; w++) {
            pool.emplace_back(body, count * w / workers, count * (w + 1) / workers);
        }
        for(auto &worker : pool) {
            worker.join();
        }
    };

    // pass 1, on all cores (the objects are scattered over the heap): dynamic type and material handler of every object
    std::vector<const MaterialHandler *> handler_of(objects.size());
    std::atomic<bool> unknown_kind{false};
    onAllCores(objects.size(), [&](size_t first, size_t last) {
        for(size_t i = first; i < last; i++) {
            const Object &object = *objects[i];
            handler_of[i] = object.getMaterialHandler();
            if(typeid(object) == typeid(Triangle)) { // both classes are final: the dynamic type is the class itself or something else
                kinds[i] = PT_OBJ_TRIANGLE;
            }
            else if(typeid(object) == typeid(Sphere)) {
                kinds[i] = PT_OBJ_SPHERE;
            }
            else {
                unknown_kind.store(true);
            }
        }
    });
    if(unknown_kind.load()) {
        throw std::invalid_argument("PathTrace: only Triangle and Sphere objects can be rendered on the device");
    }
    // ... then, in order, its position among the objects of its kind and its material index (meshes share one handler, so the
    // last one seen is remembered before the map is asked)
    std::vector<uint32_t> typed_index(objects.size()), material_of(objects.size());
    uint32_t n_triangles = 0, n_spheres = 0;
    {
        const MaterialHandler *last_handler = nullptr;
        uint32_t last_material = 0;
        bool have_last = false;
        for(size_t i = 0; i < objects.size(); i++) {
            if(!have_last || handler_of[i] != last_handler) {
                last_material = materialIndex(handler_of[i], known, materials);
                last_handler = handler_of[i];
                have_last = true;
            }
            material_of[i] = last_material;
            typed_index[i] = kinds[i] == PT_OBJ_TRIANGLE ? n_triangles++ : n_spheres++;
        }
    }
    // pass 2, on all cores: the coordinate arrays
    tri_pos.resize(9 * static_cast<size_t>(n_triangles));
    tri_nrm.resize(9 * static_cast<size_t>(n_triangles));
    tri_cull.resize(n_triangles);
    tri_material.resize(n_triangles);
    spheres.resize(4 * static_cast<size_t>(n_spheres));
    sphere_material.resize(n_spheres);
    {
        auto fill = [&](size_t first, size_t last) {
            for(size_t i = first; i < last; i++) {
                const size_t k = typed_index[i];
                if(kinds[i] == PT_OBJ_TRIANGLE) {
                    const auto *t = static_cast<const Triangle *>(objects[i].get());
                    const vec3<float> *points[3] = {&t->a, &t->b, &t->c};
                    const vec3<float> *normals[3] = {&t->normal_a, &t->normal_b, &t->normal_c};
                    for(int v = 0; v < 3; v++) {
                        for(int c = 0; c < 3; c++) {
                            tri_pos[9 * k + 3 * static_cast<size_t>(v) + static_cast<size_t>(c)] = (*points[v])[static_cast<size_t>(c)];
                            tri_nrm[9 * k + 3 * static_cast<size_t>(v) + static_cast<size_t>(c)] = (*normals[v])[static_cast<size_t>(c)];
                        }
                    }
                    tri_cull[k] = t->cullsBackface() ? 1 : 0;
                    tri_material[k] = material_of[i];
                }
                else {
                    const auto *sphere = static_cast<const Sphere *>(objects[i].get());
                    const auto o = sphere->getOrigin();
                    spheres[4 * k + 0] = o[0];
                    spheres[4 * k + 1] = o[1];
                    spheres[4 * k + 2] = o[2];
                    spheres[4 * k + 3] = sphere->getRadius();
                    sphere_material[k] = material_of[i];
                }
            }
        };
        onAllCores(objects.size(), fill);
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
    int n_replicas = 1;
    if(const char *devices_env = std::getenv("PATHTRACE_DEVICES")) {
        n_replicas = std::string(devices_env) == "all" ? pt_device_count() - device : std::atoi(devices_env);
        n_replicas = std::max(n_replicas, 1);
    }
    // (tests on a one-GPU box: $PATHTRACE_REPLICAS_SHARE_DEVICE = k makes k replicas on the first device)
    int share = 0;
    if(const char *share_env = std::getenv("PATHTRACE_REPLICAS_SHARE_DEVICE")) {
        share = std::atoi(share_env);
        n_replicas = std::max(n_replicas, share);
    }
    for(int r = 0; r < n_replicas; r++) {
        pt_scene *replica = nullptr;
        if(pt_scene_create(share > 0 ? device : device + r, &desc, &replica) != PT_OK) {
            const std::string message = pt_last_error();
            for(pt_scene *made : replicas) {
                pt_scene_destroy(made);
            }
            replicas.clear();
            throw std::runtime_error("PathTrace: cannot create the device scene on device " + std::to_string(device + r) + ": " + message);
        }
        replicas.push_back(replica);
    }
    device_scene = replicas[0];

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
    for(pt_scene *replica : replicas) {
        pt_scene_destroy(replica);
    }
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
