// ptk_math.h -- float32 device maths of the render kernels.
//
// GLSL leaves the precision of sin/cos/acos/atan/pow/normalize/inverse to the
// driver.  The kernels fix one definition: every routine here is a fixed
// sequence of IEEE-754 binary32 +,-,*,/ and sqrt operations, each rounded once
// (the file is compiled with -ffp-contract=off and correctly rounded
// divide/sqrt), plus exact bit manipulation.  Results therefore do not depend
// on scheduling and can be reproduced bit for bit by any IEEE implementation
// that follows the same operation order.  Polynomials are the classic
// single-precision Cephes kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PTK_DEV __device__ __forceinline__

struct f3 {
    float x, y, z;
};

PTK_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PTK_DEV f3 ld3(const float4& v) { return mk3(v.x, v.y, v.z); }
PTK_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTK_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTK_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PTK_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PTK_DEV f3 div3(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
// dot: products summed left to right.
PTK_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PTK_DEV f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// Correctly rounded: __builtin_sqrtf lowers to the IEEE sequence under hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt (the Makefile also passes it);
// __fsqrt_rn would be the 1-ulp native v_sqrt_f32 in this toolchain.
PTK_DEV float psqrt(float x) { return __builtin_sqrtf(x); }
PTK_DEV float length3(f3 a) { return psqrt(dot3(a, a)); }
// normalize(v) = v * (1 / sqrt(dot(v,v)))
PTK_DEV f3 normalize3(f3 a) {
    float inv = 1.0f / psqrt(dot3(a, a));
    return a * inv;
}
// reflect(I,N) = I - 2*dot(N,I)*N
PTK_DEV f3 reflect3(f3 i, f3 n) {
    float d = 2.0f * dot3(n, i);
    return i - n * d;
}
// refract(I,N,eta) per the GLSL specification
PTK_DEV f3 refract3(f3 i, f3 n, float eta) {
    float d = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float s = eta * d + psqrt(k);
    return i * eta - n * s;
}

PTK_DEV uint32_t fbits(float f) { return __float_as_uint(f); }
PTK_DEV float bitsf(uint32_t u) { return __uint_as_float(u); }

// float -> int, truncating; NaN -> 0; saturating.
PTK_DEV int32_t pf2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)x;
}
PTK_DEV float pmin(float x, float y) { return (y < x) ? y : x; }
PTK_DEV float pmax(float x, float y) { return (x < y) ? y : x; }
PTK_DEV float pabs(float x) { return bitsf(fbits(x) & 0x7fffffffu); }
PTK_DEV float psign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
PTK_DEV float pfract(float x) { return x - floorf(x); }
PTK_DEV float pmod(float x, float y) { return x - y * floorf(x / y); }

// ---- column-major mat4 (four float4 columns) -------------------------------
struct m44 {
    float4 c0, c1, c2, c3;
};
// (m * vec4(v, w)).xyz
PTK_DEV f3 mul_mv(const m44& m, f3 v, float w) {
    f3 r;
    r.x = ((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z) + m.c3.x * w;
    r.y = ((m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z) + m.c3.y * w;
    r.z = ((m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z) + m.c3.z * w;
    return r;
}
// (vec4(v, w) * m).xyz
PTK_DEV f3 mul_vm(f3 v, float w, const m44& m) {
    f3 r;
    r.x = ((v.x * m.c0.x + v.y * m.c0.y) + v.z * m.c0.z) + w * m.c0.w;
    r.y = ((v.x * m.c1.x + v.y * m.c1.y) + v.z * m.c1.z) + w * m.c1.w;
    r.z = ((v.x * m.c2.x + v.y * m.c2.y) + v.z * m.c2.z) + w * m.c2.w;
    return r;
}
// inverse(mat4): adjugate / determinant with 2x2 sub-determinants; a[i] is the
// column-major flat element col*4+row.
PTK_DEV m44 inverse44(const m44& m) {
    const float a0 = m.c0.x, a1 = m.c0.y, a2 = m.c0.z, a3 = m.c0.w;
    const float a4 = m.c1.x, a5 = m.c1.y, a6 = m.c1.z, a7 = m.c1.w;
    const float a8 = m.c2.x, a9 = m.c2.y, a10 = m.c2.z, a11 = m.c2.w;
    const float a12 = m.c3.x, a13 = m.c3.y, a14 = m.c3.z, a15 = m.c3.w;
    float s0 = a0 * a5 - a4 * a1;
    float s1 = a0 * a6 - a4 * a2;
    float s2 = a0 * a7 - a4 * a3;
    float s3 = a1 * a6 - a5 * a2;
    float s4 = a1 * a7 - a5 * a3;
    float s5 = a2 * a7 - a6 * a3;
    float c5 = a10 * a15 - a14 * a11;
    float c4 = a9 * a15 - a13 * a11;
    float c3 = a9 * a14 - a13 * a10;
    float c2 = a8 * a15 - a12 * a11;
    float c1 = a8 * a14 - a12 * a10;
    float c0 = a8 * a13 - a12 * a9;
    float det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    float inv = 1.0f / det;
    m44 r;
    r.c0.x = ((a5 * c5 - a6 * c4) + a7 * c3) * inv;
    r.c0.y = ((-a1 * c5 + a2 * c4) - a3 * c3) * inv;
    r.c0.z = ((a13 * s5 - a14 * s4) + a15 * s3) * inv;
    r.c0.w = ((-a9 * s5 + a10 * s4) - a11 * s3) * inv;
    r.c1.x = ((-a4 * c5 + a6 * c2) - a7 * c1) * inv;
    r.c1.y = ((a0 * c5 - a2 * c2) + a3 * c1) * inv;
    r.c1.z = ((-a12 * s5 + a14 * s2) - a15 * s1) * inv;
    r.c1.w = ((a8 * s5 - a10 * s2) + a11 * s1) * inv;
    r.c2.x = ((a4 * c4 - a5 * c2) + a7 * c0) * inv;
    r.c2.y = ((-a0 * c4 + a1 * c2) - a3 * c0) * inv;
    r.c2.z = ((a12 * s4 - a13 * s2) + a15 * s0) * inv;
    r.c2.w = ((-a8 * s4 + a9 * s2) - a11 * s0) * inv;
    r.c3.x = ((-a4 * c3 + a5 * c1) - a6 * c0) * inv;
    r.c3.y = ((a0 * c3 - a1 * c1) + a2 * c0) * inv;
    r.c3.z = ((-a12 * s3 + a13 * s1) - a14 * s0) * inv;
    r.c3.w = ((a8 * s3 - a9 * s1) + a10 * s0) * inv;
    return r;
}

// ---- transcendentals --------------------------------------------------------
// x = n*(pi/2) + r with a three-term Cody-Waite subtraction.
PTK_DEV float reduce_pio2(float x, int32_t& quadrant) {
    float fn = rintf(x * 0.63661977236758134308f);
    float r = x - fn * 1.5703125f;
    r = r - fn * 4.837512969970703125e-4f;
    r = r - fn * 7.54978995489188e-8f;
    quadrant = pf2i(fn);
    return r;
}
PTK_DEV float sin_poly(float r) {
    float z = r * r;
    float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
    p = p * z - 1.6666654611e-1f;
    return (p * z) * r + r;
}
PTK_DEV float cos_poly(float r) {
    float z = r * r;
    float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    return (p * z) * z + (1.0f - 0.5f * z);
}
PTK_DEV float psin(float x) {
    if (!(pabs(x) < 3.0e38f)) return x - x;
    int32_t q;
    float r = reduce_pio2(x, q);
    float s = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -s : s;
}
PTK_DEV float pcos(float x) {
    if (!(pabs(x) < 3.0e38f)) return x - x;
    int32_t q;
    float r = reduce_pio2(x, q);
    float c = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}
// sin and cos of the same argument (shares the reduction; same results as psin/pcos).
PTK_DEV void psincos(float x, float& s_out, float& c_out) {
    if (!(pabs(x) < 3.0e38f)) { s_out = x - x; c_out = x - x; return; }
    int32_t q;
    float r = reduce_pio2(x, q);
    float sp = sin_poly(r), cp = cos_poly(r);
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;
    s_out = (q & 2) ? -s : s;
    c_out = ((q + 1) & 2) ? -c : c;
}
PTK_DEV float asin_small(float x) {
    float z = x * x;
    float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    return (p * z) * x + x;
}
PTK_DEV float pacos(float x) {
    const float pi = 3.14159265358979323846f;
    const float pio2 = 1.57079632679489661923f;
    if (x != x || x > 1.0f || x < -1.0f) return bitsf(0x7fc00000u);
    if (x > 0.5f) {
        float s = psqrt(0.5f * (1.0f - x));
        return 2.0f * asin_small(s);
    }
    if (x < -0.5f) {
        float s = psqrt(0.5f * (1.0f + x));
        return pi - 2.0f * asin_small(s);
    }
    return pio2 - asin_small(x);
}
PTK_DEV float atan_pos(float x) {
    const float pio2 = 1.57079632679489661923f;
    const float pio4 = 0.78539816339744830962f;
    float y;
    if (x > 2.414213562373095f) {
        y = pio2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = pio4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    return y + ((p * z) * x + x);
}
PTK_DEV float patan2(float y, float x) {
    const float pi = 3.14159265358979323846f;
    const float pio2 = 1.57079632679489661923f;
    if (x != x || y != y) return bitsf(0x7fc00000u);
    if (x == 0.0f) {
        if (y > 0.0f) return pio2;
        if (y < 0.0f) return -pio2;
        return 0.0f;
    }
    float a = atan_pos(pabs(y / x));
    if (x < 0.0f) a = pi - a;
    return (y < 0.0f) ? -a : a;
}
PTK_DEV float plog2(float x) {
    if (x != x || x < 0.0f) return bitsf(0x7fc00000u);
    if (x == 0.0f) return bitsf(0xff800000u);
    uint32_t u = fbits(x);
    if (u == 0x7f800000u) return x;
    int32_t e = 0;
    if (u < 0x00800000u) { x = x * 16777216.0f; u = fbits(x); e = -24; }
    e += (int32_t)(u >> 23) - 127;
    float m = bitsf((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237309504880f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f * f - 1.1514610310e-1f;
    p = p * f + 1.1676998740e-1f;
    p = p * f - 1.2420140846e-1f;
    p = p * f + 1.4249322787e-1f;
    p = p * f - 1.6668057665e-1f;
    p = p * f + 2.0000714765e-1f;
    p = p * f - 2.4999993993e-1f;
    p = p * f + 3.3333331174e-1f;
    float ln = (f * z) * p - 0.5f * z;
    float r = ln * 0.44269504088896340735992f;
    r = r + f * 0.44269504088896340735992f;
    r = r + ln;
    r = r + f;
    return r + (float)e;
}
PTK_DEV float pexp2(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return bitsf(0x7f800000u);
    if (x < -126.0f) return 0.0f;
    float fn = rintf(x);
    float f = x - fn;
    float p = 1.535336188319500e-4f * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    int32_t n = pf2i(fn);
    return p * bitsf((uint32_t)(n + 127) << 23);
}
PTK_DEV float ppow(float x, float y) { return pexp2(y * plog2(x)); }
