// pt_device.h -- device functions of the hot path: fp32 vector algebra in the reference's evaluation order, the xorshift
// engine with libstdc++'s distributions, primitives, BSDFs, camera, light sampling.
//
// Everything here is compiled with -ffp-contract=off and without fast-math, so each source-level operation is one IEEE
// operation, exactly as in the reference's x86-64 build.  Citations are file:line under the reference tree.
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <float.h>

#include "pt_libm.h"
#include "pt_types.h"

#define PT_D __device__ __forceinline__

namespace ptd {

// ---- util/vector.h ---------------------------------------------------------------------------------------------------

struct V3 {
    float x, y, z;
};
struct C4 {
    float r, g, b, a;
};
struct Ray {
    V3 o, d;
};

PT_D V3 v3(float x, float y, float z) {
    V3 r = {x, y, z};
    return r;
}
PT_D V3 operator-(V3 a, V3 b) {
    return v3(a.x - b.x, a.y - b.y, a.z - b.z);
}
PT_D V3 operator+(V3 a, V3 b) {
    return v3(a.x + b.x, a.y + b.y, a.z + b.z);
}
PT_D V3 operator*(V3 a, float f) {
    return v3(a.x * f, a.y * f, a.z * f);
}
PT_D V3 neg(V3 a) {
    return v3(-a.x, -a.y, -a.z);
}
// vector.h:193-201: starts from 0 and accumulates left to right (0 + (-0) = +0 matters for signs)
PT_D float dot(V3 a, V3 b) {
    float d = 0.0f;
    d += a.x * b.x;
    d += a.y * b.y;
    d += a.z * b.z;
    return d;
}
PT_D float len2(V3 a) {
    return dot(a, a);
}
PT_D float len(V3 a) {
    return __builtin_sqrtf(len2(a));
}
// vector.h:161-167
PT_D V3 normalize(V3 a) {
    float inv = 1.0f / len(a);
    return a * inv;
}
PT_D V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vector.h:250-255: v - (n * 2) * d
PT_D V3 reflect(V3 v, V3 n) {
    float d = dot(v, n);
    return v - (n * 2.0f) * d;
}
// std::min / std::max
PT_D float fmin_std(float a, float b) {
    return (b < a) ? b : a;
}
PT_D float fmax_std(float a, float b) {
    return (a < b) ? b : a;
}

PT_D C4 c4(float r, float g, float b, float a) {
    C4 c = {r, g, b, a};
    return c;
}
PT_D C4 c4(float4 v) {
    return c4(v.x, v.y, v.z, v.w);
}
PT_D C4 operator+(C4 a, C4 b) {
    return c4(a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a);
}
PT_D C4 operator-(C4 a, C4 b) {
    return c4(a.r - b.r, a.g - b.g, a.b - b.b, a.a - b.a);
}
PT_D C4 operator*(C4 a, C4 b) {
    return c4(a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a);
}
PT_D C4 operator*(C4 a, float f) {
    return c4(a.r * f, a.g * f, a.b * f, a.a * f);
}
PT_D C4 operator/(C4 a, float d) {
    return c4(a.r / d, a.g / d, a.b / d, a.a / d);
}

// ---- base.h:24-38 + libstdc++ 11 <random> ------------------------------------------------------------------------------

PT_D uint32_t rng_draw(uint64_t &s) {
    uint64_t result = s * 0xD989BCACC137DCD5ULL;
    s ^= s >> 11;
    s ^= s << 31;
    s ^= s >> 18;
    return (uint32_t)(result >> 32);
}

// generate_canonical<float, 24>: one draw, float(draw) / 2^32, clamped below 1 (bits/random.tcc:3348-3382)
PT_D float rng_canonical(uint64_t &s) {
    float ret = (float)rng_draw(s) / 4294967296.0f;
    if(ret >= 1.0f) {
        ret = 0x1.fffffep-1f; // nextafterf(1, 0)
    }
    return ret;
}

// uniform_real_distribution<float>(a, b) (bits/random.h:1865-1871)
PT_D float rng_uniform(uint64_t &s, float a, float b) {
    return rng_canonical(s) * (b - a) + a;
}
PT_D float rng_uniform01(uint64_t &s) {
    return rng_canonical(s) * (1.0f - 0.0f) + 0.0f;
}

// bernoulli_distribution(p) (bits/random.h:3635-3644): two draws, low word first, compared in double
PT_D bool rng_bernoulli(uint64_t &s, double p) {
    double sum = (double)rng_draw(s);
    sum += (double)rng_draw(s) * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if(ret >= 1.0) {
        ret = 0x1.fffffffffffffp-1; // nextafter(1.0, 0.0)
    }
    return ret < p;
}

// ---- a-4: AABB::getIntersection, src/scene/bounding_box.cpp:38-73 -------------------------------------------------------

// inverse direction as the slab test forms it (:43-45)
PT_D V3 slab_inverse(V3 d) {
    return v3(__builtin_fabsf(d.x) > 0.0f ? 1.0f / d.x : FLT_MAX, __builtin_fabsf(d.y) > 0.0f ? 1.0f / d.y : FLT_MAX,
              __builtin_fabsf(d.z) > 0.0f ? 1.0f / d.z : FLT_MAX);
}

PT_D float slab_test(V3 lo, V3 hi, V3 o, V3 inv) {
    float t1 = (lo.x - o.x) * inv.x;
    float t2 = (hi.x - o.x) * inv.x;
    float t3 = (lo.y - o.y) * inv.y;
    float t4 = (hi.y - o.y) * inv.y;
    float t5 = (lo.z - o.z) * inv.z;
    float t6 = (hi.z - o.z) * inv.z;
    float t_min = fmax_std(fmax_std(fmin_std(t1, t2), fmin_std(t3, t4)), fmin_std(t5, t6));
    float t_max = fmin_std(fmin_std(fmax_std(t1, t2), fmax_std(t3, t4)), fmax_std(t5, t6));
    float t = t_min;
    if(t_max < 0.0f || t_min > t_max) {
        return -1.0f;
    }
    if(t_min < 0.0f && t_min <= t_max && t_max >= 0.0f) {
        t = 0.0f;
    }
    return t;
}

// ---- a-6: Triangle::getIntersection, src/scene/object.cpp:146-182 -------------------------------------------------------

PT_D float tri_intersect(V3 a, V3 ab, V3 ac, bool cull, V3 o, V3 d) {
    const float epsilon = 1E-6f;
    V3 pvec = cross(d, ac);
    float det = dot(ab, pvec);
    if(cull) {
        if(det <= epsilon) {
            return -1.0f;
        }
    }
    else {
        if(__builtin_fabsf(det) <= epsilon) {
            return -1.0f;
        }
    }
    float inv_det = 1.0f / det;
    V3 tvec = o - a;
    float u = dot(tvec, pvec) * inv_det;
    if(u < 0 || u > 1) {
        return -1.0f;
    }
    V3 qvec = cross(tvec, ab);
    float v = dot(d, qvec) * inv_det;
    if(v < 0 || u + v > 1) {
        return -1.0f;
    }
    return dot(ac, qvec) * inv_det;
}

// a-7: Triangle::getSurfaceNormal, object.cpp:126-144
PT_D V3 tri_normal(V3 a, V3 ab, V3 ac, V3 na, V3 nb, V3 nc, V3 pos) {
    V3 ap = pos - a;
    float d00 = dot(ab, ab);
    float d01 = dot(ab, ac);
    float d11 = dot(ac, ac);
    float d20 = dot(ap, ab);
    float d21 = dot(ap, ac);
    float inv_d = 1.0f / (d00 * d11 - d01 * d01);
    float v = (d11 * d20 - d01 * d21) * inv_d;
    float w = (d00 * d21 - d01 * d20) * inv_d;
    float u = 1.0f - v - w;
    return normalize((na * u + nb * v) + nc * w);
}

// a-8: Sphere::getIntersection, object.cpp:72-84 (near root only)
PT_D float sphere_intersect(V3 origin, float radius, V3 o, V3 d) {
    float radius2 = radius * radius;
    V3 co = o - origin;
    float dd = dot(d, co);
    float discriminant = dd * dd - len2(co) + radius2;
    if(discriminant >= 0) {
        return -(dd + __builtin_sqrtf(discriminant));
    }
    return -1.0f;
}

// ---- scene record access -------------------------------------------------------------------------------------------

struct TriRec {
    V3 a, ab, ac;
    uint32_t material;
    uint32_t obj_cull;
};

PT_D TriRec tri_unpack(float4 q0, float4 q1, float4 q2) {
    TriRec t;
    t.a = v3(q0.x, q0.y, q0.z);
    t.ab = v3(q0.w, q1.x, q1.y);
    t.ac = v3(q1.z, q1.w, q2.x);
    t.material = __float_as_uint(q2.y);
    t.obj_cull = __float_as_uint(q2.z);
    return t;
}

PT_D TriRec tri_load(const float4 *tris, uint32_t idx) {
    const float4 *p = tris + 3 * (size_t)idx;
    return tri_unpack(p[0], p[1], p[2]);
}

struct Material {
    C4 diffuse, specular, emission;
    float ior;
    int bsdf;
    int one_way;
};

// 16 bytes through a pointer of any address space (float4 in global memory, a 4-float vector in LDS)
template<typename F4Ptr>
PT_D float4 ldq(F4Ptr p, size_t i) {
    const auto v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

template<typename F4Ptr>
PT_D Material material_load(F4Ptr materials, uint32_t idx) {
    Material m;
    if(idx == 0xFFFFFFFFu) {
        // default handler: white Lambertian, no emission (object.cpp:9-11, material.cpp:3-17)
        m.diffuse = c4(1.0f, 1.0f, 1.0f, 1.0f);
        m.specular = c4(1.0f, 1.0f, 1.0f, 1.0f);
        m.emission = c4(0.0f, 0.0f, 0.0f, 0.0f);
        m.ior = 1.0f;
        m.bsdf = 0;
        m.one_way = 0;
        return m;
    }
    const F4Ptr p = materials + 4 * (size_t)idx;
    m.diffuse = c4(ldq(p, 0));
    m.specular = c4(ldq(p, 1));
    m.emission = c4(ldq(p, 2));
    float4 q = ldq(p, 3);
    m.ior = q.x;
    m.bsdf = (int)__float_as_uint(q.y);
    m.one_way = (int)__float_as_uint(q.z);
    return m;
}

// ---- a-13..a-15: src/scene/propagation.cpp ----------------------------------------------------------------------------

#define PT_PI_F 3.14159274101257324219f /* static_cast<float>(M_PI) */

// propagation.cpp:24-62
PT_D V3 local_to_global(V3 vec, V3 n) {
    V3 d;
    if(__builtin_fabsf(n.x) > 0.0f) {
        if(__builtin_fabsf(n.y) > 0.0f) {
            d = v3(0.0f, -n.x, n.y);
        }
        else {
            d = v3(0.0f, -n.x, n.z);
        }
    }
    else {
        if(__builtin_fabsf(n.y) > 0.0f) {
            d = v3(-n.y, n.z, 0.0f);
        }
        else {
            d = v3(1.0f, 0.0f, 0.0f);
        }
    }
    d = normalize(d);
    V3 b1 = normalize(cross(d, n));
    V3 b2 = normalize(cross(b1, n));
    V3 vx = v3(b1.x, b2.x, n.x);
    V3 vy = v3(b1.y, b2.y, n.y);
    V3 vz = v3(b1.z, b2.z, n.z);
    return v3(dot(vx, vec), dot(vy, vec), dot(vz, vec));
}

// BSDF::propagateRay for the three concrete BSDFs (propagation.cpp:89-105, 120-160, 180-204)
PT_D Ray bsdf_propagate(const Material &m, V3 ray_d, V3 pos, V3 normal, float epsilon, uint64_t &rng, float &factor, float &pd) {
    Ray out;
    if(m.bsdf == 0) {
        // importanceSampleCosine(dist(re), dist(re), 1.0F): clang evaluates the arguments left to right, r1 first.
        // With e = 1: pow(r2, 2/(e+1)) = pow(r2, 1) and pow(cos_theta, e) = pow(cos_theta, 1): the reference's build has these folded to
        // their first argument (and glibc's powf(x, 1) returns x for every x in [0, 1]: tests/test_abi_cpu.py), so no powf is evaluated
        // for them; pow(r2, 1/(e+1)) = powf(r2, 0.5f) is a real call there (and differs from sqrtf for 0.064 % of the inputs).
        float r1 = rng_uniform01(rng);
        float r2 = rng_uniform01(rng);
        float fac = __builtin_sqrtf(1.0f - r2);
        float cos_theta = ptm::powf_glibc(r2, 0.5f);
        float phi = 2.0f * PT_PI_F * r1;
        V3 local_dir = v3(fac * ptm::cosf_glibc(phi), fac * ptm::sinf_glibc(phi), cos_theta);
        float p = (1.0f + 1) * cos_theta / (2.0f * PT_PI_F);
        V3 dir = local_to_global(local_dir, normal);
        out.o = pos + dir * epsilon;
        out.d = dir;
        factor = 1.0f;
        pd = p;
        return out;
    }
    if(m.bsdf == 1) {
        float ray_dot = -dot(ray_d, normal);
        float ri_leaving = ray_dot >= 0 ? 1.0f : m.ior;
        float ri_entering = ray_dot >= 0 ? m.ior : 1.0f;
        // getFresnelReflectance(|ray_dot|, ri_leaving, ri_entering), propagation.cpp:64-83
        float rd = __builtin_fabsf(ray_dot);
        float sin_theta_i = __builtin_sqrtf(fmax_std(1.0f - rd * rd, 0.0f));
        float sin_theta_t = ri_leaving / ri_entering * sin_theta_i;
        float rat, cos_theta_t;
        if(sin_theta_t >= 1.0f) {
            rat = 1.0f;
            cos_theta_t = 0.0f;
        }
        else {
            cos_theta_t = __builtin_sqrtf(fmax_std(1.0f - sin_theta_t * sin_theta_t, 0.0f));
            float r_parallel = ((ri_entering * rd) - (ri_leaving * cos_theta_t)) / ((ri_entering * rd) + (ri_leaving * cos_theta_t));
            float r_perpendicular = ((ri_leaving * rd) - (ri_entering * cos_theta_t)) / ((ri_leaving * rd) + (ri_entering * cos_theta_t));
            rat = (r_parallel * r_parallel + r_perpendicular * r_perpendicular) / 2.0f;
        }
        if(rng_bernoulli(rng, (double)rat)) {
            V3 dir = reflect(ray_d, normal * (ray_dot < 0.0f ? -1.0f : 1.0f));
            out.o = pos + dir * epsilon;
            out.d = dir;
            factor = rat;
            pd = rat;
            return out;
        }
        float ri_ratio = ri_leaving / ri_entering;
        V3 out_dir = ray_d * ri_ratio + (normal * (ri_ratio * rd - cos_theta_t)) * (ray_dot < 0.0f ? -1.0f : 1.0f);
        out_dir = normalize(out_dir);
        float ri_fac = (ri_entering * ri_entering) / (ri_leaving * ri_leaving);
        out.o = pos + out_dir * epsilon;
        out.d = out_dir;
        factor = ri_fac * (1.0f - rat);
        pd = 1.0f - rat;
        return out;
    }
    bool unaligned = dot(ray_d, normal) > 0.0f;
    factor = 1.0f;
    pd = 1.0f;
    if(m.one_way && unaligned) {
        out.o = pos + ray_d * epsilon;
        out.d = ray_d;
        return out;
    }
    V3 normal_dir = normal;
    if(!m.one_way && unaligned) {
        normal_dir = normal_dir * -1.0f;
    }
    V3 dir = reflect(ray_d, normal_dir);
    out.o = pos + dir * epsilon;
    out.d = dir;
    return out;
}

// BSDF::getSpectrum (propagation.cpp:107-116, 162-176, 206-217)
PT_D C4 bsdf_spectrum(const Material &m, V3 from_dir, V3 to_dir, V3 normal, C4 light, bool synthetic, float &shade, float &p) {
    if(m.bsdf == 0) {
        shade = fmax_std(dot(normal, to_dir), 0.0f) / PT_PI_F;
        p = 1.0f;
        return m.diffuse * light;
    }
    p = synthetic ? 0.0f : 1.0f;
    shade = 1.0f;
    if(m.bsdf == 1) {
        if(dot(from_dir, to_dir) <= 0.0f) {
            return light * m.specular;
        }
        return light * m.diffuse;
    }
    if(!m.one_way || (dot(from_dir, to_dir) <= 0.0f)) {
        return light * m.specular;
    }
    return light;
}

// ---- a-18: Camera::shootRay, src/camera.cpp:78-113 ---------------------------------------------------------------------

PT_D V3 ld3(const float *p) {
    return v3(p[0], p[1], p[2]);
}

PT_D Ray camera_shoot(const PtDevCamera &cam, float x, float y, float pixel_width, float pixel_height, uint64_t &rng) {
    float offset_x = rng_uniform(rng, -pixel_width / 2.0f, pixel_width / 2.0f);
    float offset_y = rng_uniform(rng, -pixel_height / 2.0f, pixel_height / 2.0f);
    float sensor_x = x + offset_x;
    float sensor_y = y + offset_y;
    V3 origin = ld3(cam.origin), forward = ld3(cam.forward), up = ld3(cam.up), right = ld3(cam.right);
    V3 sensor_pos = ((origin - forward) - up * sensor_y) - right * sensor_x;

    float aperture_offset_x = 0.0f;
    float aperture_offset_y = 0.0f;
    if(cam.aperture_kind == 1) {
        // CircularApertureSampler, camera.cpp:7-19
        float r = __builtin_sqrtf(rng_uniform01(rng));
        float theta = 2.0f * PT_PI_F * rng_uniform01(rng);
        float sx = r * ptm::cosf_glibc(theta);
        float sy = r * ptm::sinf_glibc(theta);
        aperture_offset_x = sx * cam.aperture_width_half;
        aperture_offset_y = sy * cam.aperture_height_half;
    }
    else if(cam.aperture_kind == 2) {
        // HexagonalApertureSampler, camera.cpp:26-50 (rejection loop; each pass draws twice)
        float sx, sy;
        bool in_polygon;
        do {
            sx = rng_uniform01(rng);
            sy = rng_uniform01(rng);
            float relative_x = sx - cam.hex_ratio;
            in_polygon = (relative_x <= 0.0f) || (relative_x / (1.0f - cam.hex_ratio)) >= sy;
        } while(!in_polygon);
        if(rng_bernoulli(rng, 0.5)) {
            sx = -sx;
        }
        if(rng_bernoulli(rng, 0.5)) {
            sy = -sy;
        }
        aperture_offset_x = sx * cam.aperture_width_half;
        aperture_offset_y = sy * cam.aperture_height_half;
    }
    Ray ray;
    ray.o = (origin + up * aperture_offset_x) + right * aperture_offset_y;
    if(cam.focal_plane_dist > 0.0f) {
        V3 base_dir = normalize(origin - sensor_pos);
        V3 ray_target = origin + base_dir * (cam.focal_plane_dist / dot(forward, base_dir));
        ray.d = normalize(ray_target - ray.o);
    }
    else {
        ray.d = normalize(ray.o - sensor_pos);
    }
    return ray;
}

// ---- surface normal of a hit object -------------------------------------------------------------------------------------

// normal of a triangle from its shading record (six quarters), Triangle::getNormal (object.cpp:126-182)
template<typename F4Ptr>
PT_D V3 tri_shade_normal(F4Ptr rec, V3 pos, uint32_t &material) {
    const TriRec t = tri_unpack(ldq(rec, 0), ldq(rec, 1), ldq(rec, 2));
    const float4 n0 = ldq(rec, 3), n1 = ldq(rec, 4), n2 = ldq(rec, 5);
    material = t.material;
    return tri_normal(t.a, t.ab, t.ac, v3(n0.x, n0.y, n0.z), v3(n0.w, n1.x, n1.y), v3(n1.z, n1.w, n2.x), pos);
}

PT_D V3 object_normal(const PtDevScene &sc, uint32_t ref, V3 pos, uint32_t &material) {
    uint32_t idx = ref & PT_REF_INDEX; // (a hit carries the index of the object's record: pt_types.h)
    if(ref & PT_REF_SPHERE) {
        idx -= sc.n_tris + 1u;
        float4 s = sc.spheres[idx];
        material = sc.sph_meta[idx].x;
        return normalize(pos - v3(s.x, s.y, s.z)); // object.cpp:86-88
    }
    return tri_shade_normal(sc.tri_shade + 8 * (size_t)idx, pos, material); // one 128-byte line: geometry words, then the vertex normals
}

} // namespace ptd

#endif
