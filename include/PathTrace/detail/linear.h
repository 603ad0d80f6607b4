// PathTrace/detail/linear.h -- small fixed-size vectors, colours and matrices of the PathTrace API.
//
// API-compatible with the reference's util/vector.h, util/color.h and util/matrix.h (same names, same arithmetic: callers build
// their scenes with these types, so e.g. mat4 * vec3 must round exactly as the reference's does).  Written for this
// library; the public headers PathTrace/util/{vector,color,matrix}.h only include this file.
#ifndef PATHTRACE_DETAIL_LINEAR_H
#define PATHTRACE_DETAIL_LINEAR_H

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <utility>

namespace impl {

    // N scalars with element-wise arithmetic.  Reductions (dot, squared length) start from zero and add the terms in index order.
    template<typename TYPE, int SIZE>
    struct rt_vector {
        using value_type = TYPE;

        alignas(alignof(TYPE)) TYPE elements[SIZE];

        TYPE &operator[](std::size_t i) noexcept { return elements[i]; }
        constexpr TYPE operator[](std::size_t i) const noexcept { return elements[i]; }
        constexpr std::size_t size() const noexcept { return SIZE; }
        TYPE *data() noexcept { return elements; }
        constexpr const TYPE *data() const noexcept { return elements; }

        constexpr bool operator==(const rt_vector &o) const noexcept {
            for(int i = 0; i < SIZE; i++) {
                if(elements[i] != o.elements[i]) {
                    return false;
                }
            }
            return true;
        }
        constexpr bool operator!=(const rt_vector &o) const noexcept { return !(*this == o); }

#define PT_VEC_BINARY(OP)                                            \
    rt_vector operator OP(const rt_vector &o) const noexcept {      \
        rt_vector r;                                                 \
        for(int i = 0; i < SIZE; i++) {                              \
            r.elements[i] = elements[i] OP o.elements[i];            \
        }                                                            \
        return r;                                                    \
    }                                                                \
    rt_vector &operator OP##=(const rt_vector &o) noexcept {         \
        for(int i = 0; i < SIZE; i++) {                              \
            elements[i] OP## = o.elements[i];                        \
        }                                                            \
        return *this;                                                \
    }
        PT_VEC_BINARY(+)
        PT_VEC_BINARY(-)
#undef PT_VEC_BINARY

        // component-wise product
        rt_vector operator*(const rt_vector &o) const noexcept {
            rt_vector r;
            for(int i = 0; i < SIZE; i++) {
                r.elements[i] = elements[i] * o.elements[i];
            }
            return r;
        }
        rt_vector operator*(TYPE f) const noexcept {
            rt_vector r;
            for(int i = 0; i < SIZE; i++) {
                r.elements[i] = elements[i] * f;
            }
            return r;
        }
        rt_vector &operator*=(TYPE f) noexcept {
            for(int i = 0; i < SIZE; i++) {
                elements[i] *= f;
            }
            return *this;
        }
        // true division per component (not a multiplication by the reciprocal)
        rt_vector operator/(TYPE d) const noexcept {
            rt_vector r;
            for(int i = 0; i < SIZE; i++) {
                r.elements[i] = elements[i] / d;
            }
            return r;
        }
        rt_vector &operator/=(TYPE d) noexcept {
            for(int i = 0; i < SIZE; i++) {
                elements[i] /= d;
            }
            return *this;
        }
        rt_vector operator-() const noexcept {
            rt_vector r;
            for(int i = 0; i < SIZE; i++) {
                r.elements[i] = -elements[i];
            }
            return r;
        }

        TYPE getLengthSquared() const noexcept {
            TYPE sum = static_cast<TYPE>(0);
            for(int i = 0; i < SIZE; i++) {
                sum += elements[i] * elements[i];
            }
            return sum;
        }
        TYPE getLength() const noexcept { return std::sqrt(getLengthSquared()); }
        // scales by the reciprocal of the length; undefined for the zero vector
        rt_vector normalize() const noexcept {
            const TYPE inverse = static_cast<TYPE>(1) / getLength();
            return (*this) * inverse;
        }
        rt_vector normalizeSafely() noexcept { return std::abs(getLength()) > static_cast<TYPE>(0) ? normalize() : *this; }
    };

} // namespace impl

template<typename TYPE, int SIZE>
TYPE dot(const impl::rt_vector<TYPE, SIZE> &a, const impl::rt_vector<TYPE, SIZE> &b) noexcept {
    TYPE sum = static_cast<TYPE>(0);
    for(int i = 0; i < SIZE; i++) {
        sum += a[i] * b[i];
    }
    return sum;
}

template<typename TYPE, int SIZE>
impl::rt_vector<TYPE, SIZE> min(const impl::rt_vector<TYPE, SIZE> &a, const impl::rt_vector<TYPE, SIZE> &b) noexcept {
    impl::rt_vector<TYPE, SIZE> r;
    for(int i = 0; i < SIZE; i++) {
        r[i] = std::min(a[i], b[i]);
    }
    return r;
}

template<typename TYPE, int SIZE>
impl::rt_vector<TYPE, SIZE> max(const impl::rt_vector<TYPE, SIZE> &a, const impl::rt_vector<TYPE, SIZE> &b) noexcept {
    impl::rt_vector<TYPE, SIZE> r;
    for(int i = 0; i < SIZE; i++) {
        r[i] = std::max(a[i], b[i]);
    }
    return r;
}

template<typename TYPE>
impl::rt_vector<TYPE, 3> cross(const impl::rt_vector<TYPE, 3> &a, const impl::rt_vector<TYPE, 3> &b) noexcept {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}

// mirror image of v at a surface with unit normal n: v - (n * 2) * (v . n)
template<typename TYPE, int SIZE>
impl::rt_vector<TYPE, SIZE> reflect(const impl::rt_vector<TYPE, SIZE> &v, const impl::rt_vector<TYPE, SIZE> &n) noexcept {
    const TYPE d = dot(v, n);
    return v - n * 2 * d;
}

template<typename TYPE>
struct vec2 final : public impl::rt_vector<TYPE, 2> {
    using T = TYPE;
    static constexpr int SIZE = 2;
    vec2() noexcept = default;
    template<typename... A>
    vec2(A... a) noexcept : impl::rt_vector<TYPE, 2>{a...} {}
    vec2(const impl::rt_vector<TYPE, 2> &&v) noexcept : impl::rt_vector<TYPE, 2>(std::move(v)) {}
    T &x() noexcept { return this->elements[0]; }
    T &y() noexcept { return this->elements[1]; }
    T &u() noexcept { return this->elements[0]; }
    T &v() noexcept { return this->elements[1]; }
    constexpr T x() const noexcept { return this->elements[0]; }
    constexpr T y() const noexcept { return this->elements[1]; }
    constexpr T u() const noexcept { return this->elements[0]; }
    constexpr T v() const noexcept { return this->elements[1]; }
};

template<typename TYPE>
struct vec3 final : public impl::rt_vector<TYPE, 3> {
    using T = TYPE;
    static constexpr int SIZE = 3;
    vec3() noexcept = default;
    template<typename... A>
    vec3(A... a) noexcept : impl::rt_vector<TYPE, 3>{a...} {}
    vec3(const impl::rt_vector<TYPE, 3> &&v) noexcept : impl::rt_vector<TYPE, 3>(std::move(v)) {}
    T &x() noexcept { return this->elements[0]; }
    T &y() noexcept { return this->elements[1]; }
    T &z() noexcept { return this->elements[2]; }
    T &u() noexcept { return this->elements[0]; }
    T &v() noexcept { return this->elements[1]; }
    T &w() noexcept { return this->elements[2]; }
    constexpr T x() const noexcept { return this->elements[0]; }
    constexpr T y() const noexcept { return this->elements[1]; }
    constexpr T z() const noexcept { return this->elements[2]; }
    constexpr T u() const noexcept { return this->elements[0]; }
    constexpr T v() const noexcept { return this->elements[1]; }
    constexpr T w() const noexcept { return this->elements[2]; }
};

template<typename T>
using vec4 = impl::rt_vector<T, 4>;

// RGBA colour
template<typename TYPE>
struct Color : public impl::rt_vector<TYPE, 4> {
    using T = TYPE;
    static constexpr int SIZE = 4;
    Color() noexcept = default;
    template<typename... A>
    Color(A... a) : impl::rt_vector<TYPE, 4>{a...} {}
    Color(const impl::rt_vector<TYPE, 4> &&v) noexcept : impl::rt_vector<TYPE, 4>(std::move(v)) {}
    T &r() noexcept { return this->elements[0]; }
    T &g() noexcept { return this->elements[1]; }
    T &b() noexcept { return this->elements[2]; }
    T &a() noexcept { return this->elements[3]; }
    constexpr T r() const noexcept { return this->elements[0]; }
    constexpr T g() const noexcept { return this->elements[1]; }
    constexpr T b() const noexcept { return this->elements[2]; }
    constexpr T a() const noexcept { return this->elements[3]; }
};

namespace impl {

    template<typename TYPE, int WIDTH, int HEIGHT>
    struct matrix {
        alignas(alignof(TYPE)) rt_vector<TYPE, WIDTH> rows[HEIGHT];

        matrix operator*(TYPE f) const noexcept {
            matrix m;
            for(int i = 0; i < HEIGHT; i++) {
                m.rows[i] = rows[i] * f;
            }
            return m;
        }
        rt_vector<TYPE, HEIGHT> operator*(const rt_vector<TYPE, WIDTH> v) const noexcept {
            rt_vector<TYPE, HEIGHT> r;
            for(int i = 0; i < HEIGHT; i++) {
                r[i] = dot(rows[i], v);
            }
            return r;
        }
    };

} // namespace impl

template<typename T>
using mat3 = impl::matrix<T, 3, 3>;

// 4x4 matrix acting on points in affine coordinates: (x, y, z, 1) is transformed and divided by its w
template<typename T>
struct mat4 final : public impl::matrix<T, 4, 4> {
    impl::rt_vector<T, 3> operator*(const impl::rt_vector<T, 3> p) const noexcept {
        auto h = impl::matrix<T, 4, 4>::operator*(impl::rt_vector<T, 4>{p[0], p[1], p[2], static_cast<T>(1)});
        h = h * (static_cast<T>(1) / h[3]);
        return {h[0], h[1], h[2]};
    }
};

template<typename T>
const mat4<T> mat4_identity{vec4<float>{1.0F, 0.0F, 0.0F, 0.0F}, vec4<float>{0.0F, 1.0F, 0.0F, 0.0F}, vec4<float>{0.0F, 0.0F, 1.0F, 0.0F},
                            vec4<float>{0.0F, 0.0F, 0.0F, 1.0F}};

#endif
