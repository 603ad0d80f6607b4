// pt_libm.h -- the four glibc 2.35 libm functions on the hot path, restated for the device.
//
// The reference calls std::sin/std::cos/std::pow/std::acos on floats (src/scene/propagation.cpp:12-18, src/camera.cpp:12-16,
// src/scene/object.cpp:106-110), i.e. glibc's sinf, cosf, powf, acosf.  ROCm's ocml versions round differently in the last
// bit, and a one-ulp difference in a sampled direction changes the whole path, so the device evaluates glibc's published
// algorithms instead (glibc 2.35 = Ubuntu 22.04, the libm the reference links against in this image):
//   sinf/cosf  sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c   (double polynomial, |x| < 120 paths)
//   powf       sysdeps/ieee754/flt-32/e_powf.c, e_powf_log2_data.c, e_exp2f_data.c          (x >= 0 finite, no overflow paths)
//   acosf      sysdeps/ieee754/flt-32/e_acosf.c                                             (fdlibm, pure fp32)
// Provenance and licence.  The algorithms and the numerical tables below (polynomial coefficients, the 2^(i/32) and log2 tables,
// the 4/pi bits of the range reduction) are those of the files named above, written out again for this header: no glibc source
// text is included, but the constants are necessarily the same numbers.  glibc is distributed under the GNU Lesser General Public
// License, version 2.1 or later (Copyright (C) Free Software Foundation, Inc.; s_sinf.c / s_cosf.c / s_sincosf*.c / e_powf*.c /
// e_exp2f_data.c were contributed by Arm Ltd.); e_acosf.c derives from fdlibm ("Copyright (C) 1993 by Sun Microsystems, Inc.
// ... Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved.").
// Anyone redistributing this header in a product should treat it as a derived numerical restatement of LGPL-2.1+ / fdlibm code.
// They use only IEEE +,-,*,/,sqrt in fp32/fp64 and integer operations, which gfx950 executes bit-identically to x86-64
// as long as the compiler does not contract a*b+c into an fma (-ffp-contract=off; glibc's non-FMA build is the model).
// tests/test_libm_restated.py compiles this header for the host and compares it with the running glibc over the whole
// input domain the path can produce (sinf/cosf on [0, 7], powf(x, 0.5|1) on [2^-33, 1], acosf on [-1, 1]): zero mismatches.
#ifndef PT_LIBM_H
#define PT_LIBM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline
#else
#define PT_HD inline
#endif

namespace ptm {

PT_HD uint32_t as_u32(float f) {
    return __builtin_bit_cast(uint32_t, f);
}
PT_HD float as_f32(uint32_t u) {
    return __builtin_bit_cast(float, u);
}
PT_HD uint64_t as_u64(double f) {
    return __builtin_bit_cast(uint64_t, f);
}
PT_HD double as_f64(uint64_t u) {
    return __builtin_bit_cast(double, u);
}

// ---- sinf / cosf ---------------------------------------------------------------------------------------------------

// __sincosf_table[0] and [1] differ only in the sign of the cosine polynomial (index 1 evaluates -cos).
struct sincos_poly {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};

PT_HD float sincos_eval(double x, double x2, int negcos, int n) {
    const double sgn = negcos ? -1.0 : 1.0;
    if((n & 1) == 0) {
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        double x3 = x * x2;
        double t = s2 + x2 * s3;
        double x5 = x3 * x2;
        double s = x + x3 * s1;
        return (float)(s + x5 * t);
    }
    const double c0 = sgn * 0x1p0, c1 = sgn * -0x1.ffffffd0c621cp-2, c2 = sgn * 0x1.55553e1068f19p-5, c3 = sgn * -0x1.6c087e89a359dp-10,
                 c4 = sgn * 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double t2 = c3 + x2 * c4;
    double t1 = c0 + x2 * c1;
    double x6 = x4 * x2;
    double c = t1 + x4 * c2;
    return (float)(c + x6 * t2);
}

PT_HD uint32_t abstop12(float x) {
    return (as_u32(x) >> 20) & 0x7ff;
}

// reduce_fast without TOINT_INTRINSICS (x86-64): hpi_inv is prescaled by 2^24
PT_HD double sincos_reduce(double x, int *np) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * hpi;
}

// sincos_t::sign = {1, -1, -1, 1} indexed by the quadrant n & 3
PT_HD double sincos_sign(int n) {
    return (((n + 1) & 2) != 0) ? -1.0 : 1.0;
}

// valid for |y| < 120 (the path only produces [0, 2*pi])
PT_HD float sinf_glibc(float y) {
    double x = y;
    if(abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double s = x * x;
        if(abstop12(y) < abstop12(0x1p-12f)) {
            return y;
        }
        return sincos_eval(x, s, 0, 0);
    }
    int n;
    x = sincos_reduce(x, &n);
    double s = sincos_sign(n);
    return sincos_eval(x * s, x * x, (n & 2) != 0, n);
}

PT_HD float cosf_glibc(float y) {
    double x = y;
    if(abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double s = x * x;
        if(abstop12(y) < abstop12(0x1p-12f)) {
            return 1.0f;
        }
        return sincos_eval(x, s, 0, 1);
    }
    int n;
    x = sincos_reduce(x, &n);
    double s = sincos_sign(n);
    return sincos_eval(x * s, x * x, (n & 2) != 0, n ^ 1);
}

// ---- powf ----------------------------------------------------------------------------------------------------------

PT_HD double powf_log2(uint32_t ix) {
    // __powf_log2_data: 16 (invc, logc) pairs and the degree-5 polynomial
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
                             0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,
                             -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0,                0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
                             0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,  0x1.ce0a44eb17bccp-2};
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> (23 - 4)) % 16);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double z = (double)as_f32(iz);
    double r = z * invc[i] - 1;
    double y0 = logc[i] + (double)k;
    double r2 = r * r;
    double y = A0 * r + A1;
    double p = A2 * r + A3;
    double r4 = r2 * r2;
    double q = A4 * r + y0;
    q = p * r2 + q;
    y = y * r4 + q;
    return y;
}

PT_HD double powf_exp2(double xd) {
    // __exp2f_data.tab[i] = bits(2^(i/32)) - (i << 47)
    const uint64_t tab[32] = {0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
                              0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
                              0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
                              0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
                              0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
                              0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = xd + SHIFT;
    uint64_t ki = as_u64(kd);
    kd -= SHIFT;
    double r = xd - kd;
    uint64_t t = tab[ki % 32];
    t += ki << (52 - 5);
    double s = as_f64(t);
    double z = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1;
    y = z * r2 + y;
    y = y * s;
    return y;
}

// x is +0 or a positive normal float, y > 0 finite with |y * log2(x)| < 126 (the path calls it with x in [0, 1], y in {0.5, 1})
PT_HD float powf_glibc(float x, float y) {
    uint32_t ix = as_u32(x);
    if(ix == 0) {
        return x * x;
    }
    double logx = powf_log2(ix);
    double ylogx = y * logx;
    return (float)powf_exp2(ylogx);
}

// The whole of powf (glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c with TOINT_INTRINSICS == 0): every x, every y.  Used by the gamma
// correction (post_processing.cpp:171-182), whose base is any non-negative radiance and whose exponent 1 / gamma - 1 has either sign.
PT_HD int powf_checkint(uint32_t iy) {
    const int e = (int)(iy >> 23 & 0xff);
    if(e < 0x7f) {
        return 0;
    }
    if(e > 0x7f + 23) {
        return 2;
    }
    if(iy & ((1u << (0x7f + 23 - e)) - 1)) {
        return 0;
    }
    if(iy & (1u << (0x7f + 23 - e))) {
        return 1;
    }
    return 2;
}
PT_HD bool powf_zeroinfnan(uint32_t ix) {
    return 2 * ix - 1 >= 2u * 0x7f800000u - 1;
}
PT_HD float powf_glibc_full(float x, float y) {
    bool negate = false;
    uint32_t ix = as_u32(x), iy = as_u32(y);
    if(ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
        // either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan)
        if(powf_zeroinfnan(iy)) {
            if(2 * iy == 0) {
                return 1.0f; // (signalling NaNs are not distinguished: both operands are quiet on this path)
            }
            if(ix == 0x3f800000u) {
                return 1.0f;
            }
            if(2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) {
                return x + y;
            }
            if(2 * ix == 2 * 0x3f800000u) {
                return 1.0f;
            }
            if((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) {
                return 0.0f; // |x| < 1 && y == inf or |x| > 1 && y == -inf
            }
            return y * y;
        }
        if(powf_zeroinfnan(ix)) {
            float x2 = x * x;
            if((ix & 0x80000000u) && powf_checkint(iy) == 1) {
                x2 = -x2;
                negate = true;
            }
            if(2 * ix == 0 && (iy & 0x80000000u)) {
                return (negate ? -1.0f : 1.0f) / 0.0f;
            }
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        // x and y are non-zero finite
        if(ix & 0x80000000u) {
            const int yint = powf_checkint(iy);
            if(yint == 0) {
                return (x - x) / (x - x);
            }
            if(yint == 1) {
                negate = true;
            }
            ix &= 0x7fffffffu;
        }
        if(ix < 0x00800000u) {
            // normalise a subnormal x so that the exponent becomes negative
            ix = as_u32(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double logx = powf_log2(ix);
    const double ylogx = y * logx; // cannot overflow, y is single precision
    if((as_u64(ylogx) >> 47 & 0xffff) >= (as_u64(126.0) >> 47)) {
        // |y * log(x)| >= 126
        if(ylogx > 0x1.fffffffd1d571p+6) {
            const float big = negate ? -0x1p97f : 0x1p97f;
            return big * 0x1p97f;
        }
        if(ylogx <= -150.0) {
            const float tiny = negate ? -0x1p-95f : 0x1p-95f;
            return tiny * 0x1p-95f;
        }
    }
    const float r = (float)powf_exp2(ylogx);
    return negate ? -r : r;
}

// ---- acosf ---------------------------------------------------------------------------------------------------------

// correctly rounded on both sides: hipcc keeps -fhip-fp32-correctly-rounded-divide-sqrt on by default
#define PT_SQRTF(x) __builtin_sqrtf(x)

PT_HD float acosf_glibc(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f,
                pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
                qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    int32_t hx = (int32_t)as_u32(x);
    int32_t ix = hx & 0x7fffffff;
    if(ix == 0x3f800000) {
        if(hx > 0) {
            return 0.0f;
        }
        return pi + 2.0f * pio2_lo;
    }
    if(ix > 0x3f800000) {
        return (x - x) / (x - x);
    }
    if(ix < 0x3f000000) {
        if(ix <= 0x32800000) {
            return pio2_hi + pio2_lo;
        }
        float z = x * x;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if(hx < 0) {
        float z = (one + x) * 0.5f;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float s = PT_SQRTF(z);
        float r = p / q;
        float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    float z = (one - x) * 0.5f;
    float s = PT_SQRTF(z);
    float df = as_f32(as_u32(s) & 0xfffff000u);
    float c = (z - df * df) / (s + df);
    float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    float r = p / q;
    float w = r * s + c;
    return 2.0f * (df + w);
}

} // namespace ptm

#endif
