// src/host/world.cpp -- host-side member functions of the scene-description classes (PathTrace/detail/world.h).
//
// These are the per-object functions of the public API -- what a caller gets when it asks ONE sphere, triangle, box or BSDF a
// question (the reference's unit tests do: AABB::getIntersection, Object::getBoundingVolume).  They follow the reference's
// definitions operation for operation (src/scene/{object,bounding_box,light,material,propagation}.cpp) because callers rely
// on those values when they build scenes.  Rendering does not pass through here: frames are produced by the device kernels
// behind include/pt_hip.h.
#include <PathTrace/detail/world.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <random>

namespace {

    constexpr float kPi = static_cast<float>(M_PI);

    const std::shared_ptr<Material> &defaultMaterial() {
        static const std::shared_ptr<Material> material = std::make_shared<ConstantMaterial>(Color<float>(1.0F, 1.0F, 1.0F, 1.0F));
        return material;
    }

    const std::shared_ptr<MaterialHandler> &defaultHandler() {
        static const std::shared_ptr<MaterialHandler> handler =
          std::make_shared<ConstantMaterialHandler>(defaultMaterial(), std::make_shared<LambertianBRDF>());
        return handler;
    }

    float uniform01(RandomEngine &re) {
        std::uniform_real_distribution<float> dist(0, 1);
        return dist(re);
    }

    // unit vector of the cosine-power lobe around +z: exponent e, two uniform numbers
    vec3<float> cosineLobe(float r1, float r2, float e, float &density) {
        const float sin_theta = std::sqrt(1.0F - std::pow(r2, 2.0F / (e + 1)));
        const float cos_theta = std::pow(r2, 1.0F / (e + 1));
        const float phi = 2.0F * kPi * r1;
        density = (e + 1) * std::pow(cos_theta, e) / (2.0F * kPi);
        return {sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta};
    }

    // rotates a vector given in the tangent frame of n into world space; the frame is derived from which components of n vanish
    vec3<float> tangentToWorld(vec3<float> v, vec3<float> n) {
        vec3<float> helper;
        if(std::abs(n[0]) > 0.0F) {
            helper = std::abs(n[1]) > 0.0F ? vec3<float>{0.0F, -n[0], n[1]} : vec3<float>{0.0F, -n[0], n[2]};
        }
        else {
            helper = std::abs(n[1]) > 0.0F ? vec3<float>{-n[1], n[2], 0.0F} : vec3<float>{1.0F, 0.0F, 0.0F};
        }
        helper = helper.normalize();
        const auto t1 = cross(helper, n).normalize();
        const auto t2 = cross(t1, n).normalize();
        const vec3<float> row_x = {t1[0], t2[0], n[0]};
        const vec3<float> row_y = {t1[1], t2[1], n[1]};
        const vec3<float> row_z = {t1[2], t2[2], n[2]};
        return {dot(row_x, v), dot(row_y, v), dot(row_z, v)};
    }

    // unpolarised Fresnel reflectance and the cosine of the refraction angle; total internal reflection gives (1, 0)
    std::tuple<float, float> fresnel(float cos_i, float n_from, float n_to) {
        const float sin_i = std::sqrt(std::max(1.0F - cos_i * cos_i, 0.0F));
        const float sin_t = n_from / n_to * sin_i;
        if(sin_t >= 1.0F) {
            return {1.0F, 0.0F};
        }
        const float cos_t = std::sqrt(std::max(1.0F - sin_t * sin_t, 0.0F));
        const float parallel = ((n_to * cos_i) - (n_from * cos_t)) / ((n_to * cos_i) + (n_from * cos_t));
        const float perpendicular = ((n_from * cos_i) - (n_to * cos_t)) / ((n_from * cos_i) + (n_to * cos_t));
        return {(parallel * parallel + perpendicular * perpendicular) / 2.0F, cos_t};
    }

} // namespace

// ---- lights and materials ---------------------------------------------------------------------------------------------------

std::tuple<vec3<float>, float> PointLightSource::importanceSample(vec3<float> /*from*/) const noexcept {
    return {pos, 1.0F};
}

Spectrum PointLightSource::getSpectrum(Ray /*ray*/) const noexcept {
    return spectrum;
}

Color<float> Material::getSpecularColor(vec3<float> /*pos*/) const noexcept {
    return Color<float>{1.0F, 1.0F, 1.0F, 1.0F};
}
float Material::getRefractiveIndex(vec3<float> /*pos*/) const noexcept {
    return 1.0F;
}
Spectrum Material::getEmission(Ray /*ray*/, vec3<float> /*pos*/) const noexcept {
    return {};
}
Spectrum Material::probeEmission() const noexcept {
    return {};
}

ConstantMaterial::ConstantMaterial(Color<float> diffuse_color, float refractive_index, Spectrum emission) noexcept :
  diffuse_color(diffuse_color), refractive_index(refractive_index), emission(emission) {}
Color<float> ConstantMaterial::getDiffuseColor(vec3<float> /*pos*/) const noexcept {
    return diffuse_color;
}
float ConstantMaterial::getRefractiveIndex(vec3<float> /*pos*/) const noexcept {
    return refractive_index;
}
Spectrum ConstantMaterial::getEmission(Ray /*ray*/, vec3<float> /*pos*/) const noexcept {
    return emission;
}
Spectrum ConstantMaterial::probeEmission() const noexcept {
    return emission;
}

// ---- BSDFs ------------------------------------------------------------------------------------------------------------------

LambertianBRDF::LambertianBRDF() noexcept = default;

std::tuple<Ray, float, float> LambertianBRDF::propagateRay(Ray /*ray*/, vec3<float> pos, vec3<float> normal, float epsilon, RandomEngine &re,
                                                           const Material * /*material*/) const noexcept {
    const float r1 = uniform01(re); // first draw: azimuth
    const float r2 = uniform01(re); // second draw: elevation
    float density;
    const vec3<float> dir = tangentToWorld(cosineLobe(r1, r2, 1.0F, density), normal);
    return {Ray{pos + dir * epsilon, dir}, 1.0F, density};
}

std::tuple<Spectrum, float, float> LambertianBRDF::getSpectrum(Ray /*from_camera*/, Ray to_light, vec3<float> pos, vec3<float> normal,
                                                               Spectrum light_spectrum, const Material *material, bool /*synthetic*/) const noexcept {
    const float shade = std::max(dot(normal, to_light.dir), 0.0F) / kPi;
    return {Spectrum{material->getDiffuseColor(pos)} * light_spectrum, shade, 1.0F};
}

GlassBDF::GlassBDF() noexcept = default;

std::tuple<Ray, float, float> GlassBDF::propagateRay(Ray ray, vec3<float> pos, vec3<float> normal, float epsilon, RandomEngine &re,
                                                     const Material *material) const noexcept {
    const float cos_signed = -dot(ray.dir, normal);
    const float ior = material->getRefractiveIndex(pos);
    const float n_from = cos_signed >= 0 ? 1.0F : ior;
    const float n_to = cos_signed >= 0 ? ior : 1.0F;
    auto [reflectance, cos_t] = fresnel(std::abs(cos_signed), n_from, n_to);

    std::bernoulli_distribution reflect_choice(reflectance);
    const float side = cos_signed < 0.0F ? -1.0F : 1.0F;
    if(reflect_choice(re)) {
        const vec3<float> dir = reflect(ray.dir, normal * side);
        return {Ray{pos + dir * epsilon, dir}, reflectance, reflectance};
    }
    const float eta = n_from / n_to;
    vec3<float> dir = ray.dir * eta + normal * (eta * std::abs(cos_signed) - cos_t) * side;
    dir = dir.normalize();
    const float radiance_scale = (n_to * n_to) / (n_from * n_from);
    return {Ray{pos + dir * epsilon, dir}, radiance_scale * (1.0F - reflectance), 1.0F - reflectance};
}

std::tuple<Spectrum, float, float> GlassBDF::getSpectrum(Ray from_camera, Ray to_light, vec3<float> pos, vec3<float> /*normal*/, Spectrum light_spectrum,
                                                         const Material *material, bool synthetic) const noexcept {
    const bool reflected = dot(from_camera.dir, to_light.dir) <= 0.0F;
    const Spectrum tint{reflected ? material->getSpecularColor(pos) : material->getDiffuseColor(pos)};
    return {light_spectrum * tint, 1.0F, synthetic ? 0.0F : 1.0F};
}

MirrorBRDF::MirrorBRDF(bool one_way) noexcept : one_way(one_way) {}

std::tuple<Ray, float, float> MirrorBRDF::propagateRay(Ray ray, vec3<float> pos, vec3<float> normal, float epsilon, RandomEngine & /*re*/,
                                                       const Material * /*material*/) const noexcept {
    const bool from_behind = dot(ray.dir, normal) > 0.0F;
    vec3<float> dir = ray.dir;
    if(!(one_way && from_behind)) {
        dir = reflect(ray.dir, from_behind ? vec3<float>(normal * -1.0F) : normal);
    }
    return {Ray{pos + dir * epsilon, dir}, 1.0F, 1.0F};
}

std::tuple<Spectrum, float, float> MirrorBRDF::getSpectrum(Ray from_camera, Ray to_light, vec3<float> pos, vec3<float> /*normal*/, Spectrum light_spectrum,
                                                           const Material *material, bool synthetic) const noexcept {
    Spectrum out = light_spectrum;
    if(!one_way || dot(from_camera.dir, to_light.dir) <= 0.0F) {
        out = out * Spectrum{material->getSpecularColor(pos)};
    }
    return {out, 1.0F, synthetic ? 0.0F : 1.0F};
}

// ---- material handlers and objects ------------------------------------------------------------------------------------------

const Material *MaterialHandler::probeMaterial() const noexcept {
    return defaultMaterial().get();
}

ConstantMaterialHandler::ConstantMaterialHandler(std::shared_ptr<Material> material, std::shared_ptr<BSDF> bsdf) :
  material(std::move(material)), bsdf(std::move(bsdf)) {}
const Material *ConstantMaterialHandler::probeMaterial() const noexcept {
    return material.get();
}
const Material *ConstantMaterialHandler::getMaterial(vec3<float> /*pos*/) const noexcept {
    return material.get();
}
const BSDF *ConstantMaterialHandler::getBSDF(vec3<float> /*pos*/) const noexcept {
    return bsdf.get();
}

Object::Object() : material_handler(defaultHandler()) {}
Object::Object(std::shared_ptr<MaterialHandler> material_handler) noexcept : material_handler(std::move(material_handler)) {}
const MaterialHandler *Object::getMaterialHandler() const noexcept {
    return material_handler.get();
}
void Object::setMaterialHandler(std::shared_ptr<MaterialHandler> handler) {
    material_handler = std::move(handler);
}
float Object::getSurfaceArea() const noexcept {
    return 0.0F;
}
std::tuple<vec3<float>, float, bool> Object::sampleSurface(RandomEngine & /*re*/) const noexcept {
    return {vec3<float>{}, 0.0F, false};
}

float NullObject::getIntersection(const Ray & /*ray*/) const noexcept {
    return -1.0F;
}
vec3<float> NullObject::getSurfaceNormal(vec3<float> /*pos*/) const noexcept {
    return {0.0F, 1.0F, 0.0F};
}
AABBArea NullObject::getBoundingVolume() const noexcept {
    return {};
}
float NullObject::getSurfaceArea() const noexcept {
    return 0.0F;
}

Sphere::Sphere(vec3<float> origin, float radius) : origin(origin), radius(radius), radius2(radius * radius) {}

float Sphere::getIntersection(const Ray &ray) const noexcept {
    const auto to_origin = ray.origin - origin;
    const float b = dot(ray.dir, to_origin);
    const float discriminant = b * b - to_origin.getLengthSquared() + radius2;
    return discriminant >= 0 ? -(b + std::sqrt(discriminant)) : -1.0F;
}
vec3<float> Sphere::getSurfaceNormal(vec3<float> pos) const noexcept {
    return (pos - origin).normalize();
}
AABBArea Sphere::getBoundingVolume() const noexcept {
    const vec3<float> r = {radius, radius, radius};
    return {origin - r, origin + r};
}
float Sphere::getSurfaceArea() const noexcept {
    return 4.0F * kPi * radius2;
}
std::tuple<vec3<float>, float, bool> Sphere::sampleSurface(RandomEngine &re) const noexcept {
    const float theta = 2.0F * kPi * uniform01(re);
    const float phi = std::acos(1.0F - 2.0F * uniform01(re));
    const float x = std::sin(phi) * std::cos(theta);
    const float y = std::sin(phi) * std::sin(theta);
    const float z = std::cos(phi);
    return {origin + vec3<float>{x, y, z} * radius, 1.0F / (4.0F * kPi * radius2), false};
}

Triangle::Triangle(vec3<float> a, vec3<float> b, vec3<float> c, bool cull_backface) : a(a), b(b), c(c), cull_backface(cull_backface) {
    normal_a = normal_b = normal_c = cross(b - a, c - a).normalize();
}

// Moeller-Trumbore; may return a negative distance (= miss)
float Triangle::getIntersection(const Ray &ray) const noexcept {
    constexpr float tiny = 1E-6F;
    const auto edge_ab = b - a;
    const auto edge_ac = c - a;
    const auto p = cross(ray.dir, edge_ac);
    const float det = dot(edge_ab, p);
    if(cull_backface ? det <= tiny : std::abs(det) <= tiny) {
        return -1.0F;
    }
    const float inv_det = 1.0F / det;
    const auto from_a = ray.origin - a;
    const float u = dot(from_a, p) * inv_det;
    if(u < 0 || u > 1) {
        return -1.0F;
    }
    const auto q = cross(from_a, edge_ab);
    const float v = dot(ray.dir, q) * inv_det;
    if(v < 0 || u + v > 1) {
        return -1.0F;
    }
    return dot(edge_ac, q) * inv_det;
}

vec3<float> Triangle::getSurfaceNormal(vec3<float> pos) const noexcept {
    const auto edge_ab = b - a;
    const auto edge_ac = c - a;
    const auto from_a = pos - a;
    const float bb = dot(edge_ab, edge_ab);
    const float bc = dot(edge_ab, edge_ac);
    const float cc = dot(edge_ac, edge_ac);
    const float pb = dot(from_a, edge_ab);
    const float pc = dot(from_a, edge_ac);
    const float inv = 1.0F / (bb * cc - bc * bc);
    const float weight_b = (cc * pb - bc * pc) * inv;
    const float weight_c = (bb * pc - bc * pb) * inv;
    const float weight_a = 1.0F - weight_b - weight_c;
    return (normal_a * weight_a + normal_b * weight_b + normal_c * weight_c).normalize();
}
AABBArea Triangle::getBoundingVolume() const noexcept {
    return {min(min(a, b), c), max(max(a, b), c)};
}
float Triangle::getSurfaceArea() const noexcept {
    return cross(b - a, c - a).getLength() / 2.0F;
}
std::tuple<vec3<float>, float, bool> Triangle::sampleSurface(RandomEngine &re) const noexcept {
    const float r1 = uniform01(re);
    const float r2 = uniform01(re);
    const float root = std::sqrt(r1);
    const vec3<float> point = a * (1.0F - root) + b * (root * (1.0F - r2)) + c * (root * r2);
    const float area = cross(b - a, c - a).getLength() / 2.0F;
    return {point, 1.0F / area, cull_backface};
}

// ---- bounding boxes ---------------------------------------------------------------------------------------------------------

AABB::AABB() : child(std::make_unique<NullObject>()), leaf(true) {}
AABB::AABB(AABB &&other) noexcept :
  area(other.area), left(std::move(other.left)), right(std::move(other.right)), child(std::move(other.child)), leaf(other.leaf) {}
AABB &AABB::operator=(AABB &&other) noexcept {
    area = other.area;
    left = std::move(other.left);
    right = std::move(other.right);
    child = std::move(other.child);
    leaf = other.leaf;
    return *this;
}
AABB::AABB(AABB &&l, AABB &&r) : area{min(l.area.low, r.area.low), max(l.area.high, r.area.high)}, leaf(false) {
    left = std::make_unique<AABB>(std::move(l));
    right = std::make_unique<AABB>(std::move(r));
}
AABB::AABB(AABBArea area, std::unique_ptr<Object> &&child) noexcept : area(area), child(std::move(child)), leaf(true) {}

float AABB::getIntersection(const Ray &ray) const noexcept {
    float inverse[3];
    for(int k = 0; k < 3; k++) {
        inverse[k] = std::abs(ray.dir[k]) > 0.0F ? 1.0F / ray.dir[k] : std::numeric_limits<float>::max();
    }
    const auto to_low = area.low - ray.origin;
    const auto to_high = area.high - ray.origin;
    const float x1 = to_low[0] * inverse[0], x2 = to_high[0] * inverse[0];
    const float y1 = to_low[1] * inverse[1], y2 = to_high[1] * inverse[1];
    const float z1 = to_low[2] * inverse[2], z2 = to_high[2] * inverse[2];
    const float enter = std::max(std::max(std::min(x1, x2), std::min(y1, y2)), std::min(z1, z2));
    const float leave = std::min(std::min(std::max(x1, x2), std::max(y1, y2)), std::max(z1, z2));
    if(leave < 0.0F || enter > leave) {
        return -1.0F;
    }
    return enter < 0.0F ? 0.0F : enter; // origin inside the box
}
