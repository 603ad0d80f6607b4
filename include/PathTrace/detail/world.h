// PathTrace/detail/world.h -- the scene-description classes of the PathTrace API: spectra, lights, materials, BSDFs, objects,
// bounding boxes and the Scene.
//
// Source-compatible with the reference's scene/*.h headers (same class names, constructors and virtual interfaces), so code
// that builds scenes for the reference builds them for this library unchanged.  What is different is underneath:
//   * Scene::Scene flattens its objects into the arrays of include/pt_hip.h and creates a device-resident scene
//     (pt_scene_create); it accepts the concrete classes declared here -- Triangle, Sphere, ConstantMaterialHandler with
//     ConstantMaterial, LambertianBRDF / GlassBDF / MirrorBRDF, PointLightSource -- and throws std::invalid_argument for
//     user-defined subclasses, which cannot be evaluated on the device;
//   * Scene::getIntersection and the render entry points of PathTrace/worker.h run on the GPU through the C ABI;
//   * the per-object virtual functions (Object::getIntersection, BSDF::propagateRay, ...) are ordinary host code kept for
//     callers that use them directly (the reference's own unit tests do); the renderer never calls them.
// The public headers PathTrace/scene/*.h only include this file.
#ifndef PATHTRACE_DETAIL_WORLD_H
#define PATHTRACE_DETAIL_WORLD_H

#include <PathTrace/detail/core.h>

#include <array>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

// ---- light.h -------------------------------------------------------------------------------------------------------------

// RGBA radiance / reflectance; arithmetic is per component
class Spectrum {
  public:
    Spectrum(Color<float> color = {0.0F, 0.0F, 0.0F, 0.0F}) noexcept : color(color) {}

    Color<float> getColor() const noexcept { return color; }
    Spectrum operator+(Spectrum o) const noexcept { return {Color<float>(color + o.color)}; }
    Spectrum operator*(Spectrum o) const noexcept { return {Color<float>(color * o.color)}; }
    Spectrum operator*(float f) const noexcept { return {Color<float>(color * f)}; }
    Spectrum operator/(float d) const noexcept { return {Color<float>(color / d)}; }

  private:
    Color<float> color;
};

class LightSource {
  public:
    virtual ~LightSource() = default;
    // (point to aim at, probability density of that choice)
    virtual std::tuple<vec3<float>, float> importanceSample(vec3<float> pos) const noexcept = 0;
    virtual Spectrum getSpectrum(Ray ray) const noexcept = 0;
};

// emits `spectrum` in every direction, without distance falloff
class PointLightSource final : public LightSource {
  public:
    PointLightSource(vec3<float> pos, Spectrum spectrum) noexcept : pos(pos), spectrum(spectrum) {}
    std::tuple<vec3<float>, float> importanceSample(vec3<float> from) const noexcept override;
    Spectrum getSpectrum(Ray ray) const noexcept override;

  private:
    vec3<float> pos;
    Spectrum spectrum;
};

// ---- material.h ----------------------------------------------------------------------------------------------------------

class Material {
  public:
    virtual ~Material() = default;
    virtual Color<float> getDiffuseColor(vec3<float> pos) const noexcept = 0;
    virtual Color<float> getSpecularColor(vec3<float> pos) const noexcept; // white
    virtual float getRefractiveIndex(vec3<float> pos) const noexcept;      // 1
    virtual Spectrum getEmission(Ray ray, vec3<float> pos) const noexcept; // none
    virtual Spectrum probeEmission() const noexcept;                       // none
};

class ConstantMaterial final : public Material {
  public:
    ConstantMaterial(Color<float> diffuse_color = Color<float>(1.0F, 1.0F, 1.0F, 1.0F), float refractive_index = 1.0F, Spectrum emission = {}) noexcept;
    Color<float> getDiffuseColor(vec3<float> pos) const noexcept override;
    float getRefractiveIndex(vec3<float> pos) const noexcept override;
    Spectrum getEmission(Ray ray, vec3<float> pos) const noexcept override;
    Spectrum probeEmission() const noexcept override;

  private:
    Color<float> diffuse_color;
    float refractive_index;
    Spectrum emission;
};

// ---- propagation.h -------------------------------------------------------------------------------------------------------

class BSDF {
  public:
    virtual ~BSDF() = default;
    // samples the continuation of `ray` at `pos`: (next ray offset by epsilon, throughput factor, probability density)
    virtual std::tuple<Ray, float, float> propagateRay(Ray ray, vec3<float> pos, vec3<float> normal, float epsilon, RandomEngine &re,
                                                       const Material *material) const noexcept = 0;
    // evaluates the pair (from_camera, to_light): (spectrum, shading factor, probability density of the pair)
    virtual std::tuple<Spectrum, float, float> getSpectrum(Ray from_camera, Ray to_light, vec3<float> pos, vec3<float> normal, Spectrum light_spectrum,
                                                           const Material *material, bool synthetic = false) const noexcept = 0;
};

#define PT_DECLARE_BSDF_INTERFACE                                                                                                             \
    std::tuple<Ray, float, float> propagateRay(Ray ray, vec3<float> pos, vec3<float> normal, float epsilon, RandomEngine &re,             \
                                               const Material *material) const noexcept override;                                          \
    std::tuple<Spectrum, float, float> getSpectrum(Ray from_camera, Ray to_light, vec3<float> pos, vec3<float> normal, Spectrum light_spectrum, \
                                                   const Material *material, bool synthetic = false) const noexcept override;

// cosine-weighted diffuse reflection around the surface normal as given
class LambertianBRDF : public BSDF {
  public:
    LambertianBRDF() noexcept;
    PT_DECLARE_BSDF_INTERFACE
};

// smooth dielectric: Fresnel-weighted choice between mirror reflection and refraction
class GlassBDF : public BSDF {
  public:
    GlassBDF() noexcept;
    PT_DECLARE_BSDF_INTERFACE
};

// perfect mirror; one_way lets rays through its back face
class MirrorBRDF : public BSDF {
  public:
    MirrorBRDF(bool one_way = false) noexcept;
    PT_DECLARE_BSDF_INTERFACE
    bool isOneWay() const noexcept { return one_way; }

  private:
    bool one_way;
};

#undef PT_DECLARE_BSDF_INTERFACE

// ---- object.h ------------------------------------------------------------------------------------------------------------

class MaterialHandler {
  public:
    virtual ~MaterialHandler() = default;
    virtual const Material *probeMaterial() const noexcept; // the default white material
    virtual const Material *getMaterial(vec3<float> pos) const noexcept = 0;
    virtual const BSDF *getBSDF(vec3<float> pos) const noexcept = 0;
};

// one material and one BSDF for the whole surface
class ConstantMaterialHandler final : public MaterialHandler {
  public:
    ConstantMaterialHandler(std::shared_ptr<Material> material, std::shared_ptr<BSDF> bsdf);
    const Material *probeMaterial() const noexcept override;
    const Material *getMaterial(vec3<float> pos) const noexcept override;
    const BSDF *getBSDF(vec3<float> pos) const noexcept override;

  private:
    std::shared_ptr<Material> material;
    std::shared_ptr<BSDF> bsdf;
};

struct AABBArea {
    vec3<float> low;
    vec3<float> high;
};

class Object {
  public:
    virtual ~Object() = default;
    Object(); // default handler: white Lambertian
    Object(std::shared_ptr<MaterialHandler> material_handler) noexcept;

    // distance along the ray to the surface, negative for a miss
    virtual float getIntersection(const Ray &ray) const noexcept = 0;
    virtual vec3<float> getSurfaceNormal(vec3<float> pos) const noexcept = 0;
    virtual AABBArea getBoundingVolume() const noexcept = 0;
    virtual float getSurfaceArea() const noexcept;
    // (uniformly sampled surface point, its density, whether only the front face emits)
    virtual std::tuple<vec3<float>, float, bool> sampleSurface(RandomEngine &re) const noexcept;

    const MaterialHandler *getMaterialHandler() const noexcept;
    void setMaterialHandler(std::shared_ptr<MaterialHandler> material_handler);

  private:
    std::shared_ptr<MaterialHandler> material_handler;
};

// what an empty scene holds: never hit
class NullObject final : public Object {
  public:
    NullObject() = default;
    float getIntersection(const Ray &ray) const noexcept override;
    vec3<float> getSurfaceNormal(vec3<float> pos) const noexcept override;
    AABBArea getBoundingVolume() const noexcept override;
    float getSurfaceArea() const noexcept override;
};

class Sphere final : public Object {
  public:
    Sphere(vec3<float> origin, float radius);
    float getIntersection(const Ray &ray) const noexcept override; // near root only: misses from inside
    vec3<float> getSurfaceNormal(vec3<float> pos) const noexcept override;
    AABBArea getBoundingVolume() const noexcept override;
    float getSurfaceArea() const noexcept override;
    std::tuple<vec3<float>, float, bool> sampleSurface(RandomEngine &re) const noexcept override;

    vec3<float> getOrigin() const noexcept { return origin; }
    float getRadius() const noexcept { return radius; }

  private:
    vec3<float> origin;
    float radius;
    float radius2;
};

class Triangle final : public Object {
  public:
    vec3<float> a;
    vec3<float> b;
    vec3<float> c;
    vec3<float> normal_a; // per-vertex shading normals; the constructor sets all three to the face normal
    vec3<float> normal_b;
    vec3<float> normal_c;

    Triangle(vec3<float> a, vec3<float> b, vec3<float> c, bool cull_backface = false);
    float getIntersection(const Ray &ray) const noexcept override;
    vec3<float> getSurfaceNormal(vec3<float> pos) const noexcept override; // barycentric blend of the vertex normals
    AABBArea getBoundingVolume() const noexcept override;
    float getSurfaceArea() const noexcept override;
    std::tuple<vec3<float>, float, bool> sampleSurface(RandomEngine &re) const noexcept override;

    bool cullsBackface() const noexcept { return cull_backface; }

  private:
    bool cull_backface;
};

// ---- bounding_box.h ------------------------------------------------------------------------------------------------------

// node of a binary bounding-volume hierarchy: two children, or one object
class AABB {
  public:
    AABBArea area;
    std::unique_ptr<AABB> left;
    std::unique_ptr<AABB> right;
    std::unique_ptr<Object> child;
    bool leaf;

    AABB(); // leaf around a NullObject
    AABB(AABB &&other) noexcept;
    AABB &operator=(AABB &&other) noexcept;
    AABB(AABB &&left, AABB &&right);                             // inner node, box = union
    AABB(AABBArea area, std::unique_ptr<Object> &&child) noexcept; // leaf

    // slab test: entry distance, 0 if the origin is inside, negative for a miss
    float getIntersection(const Ray &ray) const noexcept;
};

// ---- scene.h -------------------------------------------------------------------------------------------------------------

struct pt_scene;

class Scene {
  public:
    // takes ownership; builds the hierarchy and the device-resident copy (device = $PATHTRACE_DEVICE, default 0).  With
    // $PATHTRACE_DEVICES = N (or "all") the scene is replicated on N devices starting at that one and processJob deals its tiles
    // out to them (src/worker.cpp:364-387's worker threads become one persistent launch per device).
    Scene(std::vector<std::unique_ptr<Object>> &&objects, std::vector<std::unique_ptr<LightSource>> &&light_sources);
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    // closest object along the ray (t < 0: none) -- one-ray batch on the device
    std::tuple<float, const Object *> getIntersection(const Ray &ray) const noexcept;

    // light samples for a surface point: every LightSource, then min(2 + log10(E + 1), E) points on emissive objects
    std::vector<std::tuple<vec3<float>, Spectrum, float>> sampleLights(vec3<float> pos, vec3<float> n, RandomEngine &re) const noexcept;

    // the device scene behind the C ABI (include/pt_hip.h); used by processJob / processItem
    pt_scene *deviceScene() const noexcept { return device_scene; }
    const std::vector<pt_scene *> &deviceScenes() const noexcept { return replicas; }

  private:
    std::vector<std::unique_ptr<Object>> objects;
    std::vector<std::unique_ptr<LightSource>> light_sources;
    std::vector<const Object *> emissive;      // in registration order
    std::vector<float> emissive_cdf;           // normalised inclusive prefix sums
    pt_scene *device_scene = nullptr;     // = replicas[0]
    std::vector<pt_scene *> replicas;     // one per device
};

#endif
