/*
 * pto_math.h -- float32 scalar/vector maths of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it.
 *
 * GLSL leaves the precision of sin/cos/acos/atan/pow/normalize/inverse to the
 * driver, so the reference shader (shaders/ray_tracer.comp) has no single
 * bit-level answer for them.  The oracle fixes one: every function below is
 * built only from IEEE-754 binary32 +,-,*,/ and sqrt, each rounded once
 * (compile with -ffp-contract=off, no -ffast-math), plus exact bit
 * manipulation.  A second implementation that follows the same operation
 * order reproduces these results bit for bit on any IEEE machine.
 *
 * Polynomial coefficients are the classic single-precision Cephes ones
 * (sinf/cosf/asinf/atanf/logf/expf family).
 */
#ifndef PTO_MATH_H
#define PTO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { float c[4][4]; } m4; /* column-major: c[col][row], like GLSL */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int32_t f2i_bits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float i2f_bits(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

/* float -> int conversion, truncating, NaN -> 0, saturating (GLSL int(x) is
 * undefined there; this is the oracle's choice). */
static inline int32_t ptm_f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)x;
}

/* GLSL min/max/abs/sign/fract/mod as the spec writes them. */
static inline float ptm_min(float x, float y) { return (y < x) ? y : x; }
static inline float ptm_max(float x, float y) { return (x < y) ? y : x; }
static inline float ptm_abs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
static inline float ptm_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float ptm_fract(float x) { return x - floorf(x); }
static inline float ptm_mod(float x, float y) { return x - y * floorf(x / y); }

/* ---- vectors ---------------------------------------------------------- */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* dot: products summed left to right, one rounding per operation. */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
/* normalize(v) = v * (1 / sqrt(dot(v,v))) */
static inline v3 v3_normalize(v3 a) {
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_scale(a, inv);
}
/* reflect(I,N) = I - 2*dot(N,I)*N */
static inline v3 v3_reflect(v3 i, v3 n) {
    float d = 2.0f * v3_dot(n, i);
    return v3_sub(i, v3_scale(n, d));
}
/* refract(I,N,eta) per the GLSL specification. */
static inline v3 v3_refract(v3 i, v3 n, float eta) {
    float d = v3_dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
    float s = eta * d + sqrtf(k);
    return v3_sub(v3_scale(i, eta), v3_scale(n, s));
}

/* ---- mat4 -------------------------------------------------------------- */
/* (m * vec4(v, w)).xyz : row r = ((m[0][r]*x + m[1][r]*y) + m[2][r]*z) + m[3][r]*w */
static inline v3 m4_mul_v(const m4* m, v3 v, float w) {
    v3 r;
    r.x = ((m->c[0][0] * v.x + m->c[1][0] * v.y) + m->c[2][0] * v.z) + m->c[3][0] * w;
    r.y = ((m->c[0][1] * v.x + m->c[1][1] * v.y) + m->c[2][1] * v.z) + m->c[3][1] * w;
    r.z = ((m->c[0][2] * v.x + m->c[1][2] * v.y) + m->c[2][2] * v.z) + m->c[3][2] * w;
    return r;
}
/* (vec4(v, w) * m).xyz : column c = ((x*m[c][0] + y*m[c][1]) + z*m[c][2]) + w*m[c][3] */
static inline v3 v_mul_m4(v3 v, float w, const m4* m) {
    v3 r;
    r.x = ((v.x * m->c[0][0] + v.y * m->c[0][1]) + v.z * m->c[0][2]) + w * m->c[0][3];
    r.y = ((v.x * m->c[1][0] + v.y * m->c[1][1]) + v.z * m->c[1][2]) + w * m->c[1][3];
    r.z = ((v.x * m->c[2][0] + v.y * m->c[2][1]) + v.z * m->c[2][2]) + w * m->c[2][3];
    return r;
}
/* inverse(mat4): adjugate / determinant, 2x2 sub-determinants first.
 * a[i] is the matrix in column-major flat order (a[col*4+row]). */
static inline m4 m4_inverse(const m4* m) {
    const float* a = &m->c[0][0];
    float s0 = a[0] * a[5] - a[4] * a[1];
    float s1 = a[0] * a[6] - a[4] * a[2];
    float s2 = a[0] * a[7] - a[4] * a[3];
    float s3 = a[1] * a[6] - a[5] * a[2];
    float s4 = a[1] * a[7] - a[5] * a[3];
    float s5 = a[2] * a[7] - a[6] * a[3];
    float c5 = a[10] * a[15] - a[14] * a[11];
    float c4 = a[9] * a[15] - a[13] * a[11];
    float c3 = a[9] * a[14] - a[13] * a[10];
    float c2 = a[8] * a[15] - a[12] * a[11];
    float c1 = a[8] * a[14] - a[12] * a[10];
    float c0 = a[8] * a[13] - a[12] * a[9];
    float det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    float inv = 1.0f / det;
    m4 r;
    float* b = &r.c[0][0];
    b[0] = ((a[5] * c5 - a[6] * c4) + a[7] * c3) * inv;
    b[1] = ((-a[1] * c5 + a[2] * c4) - a[3] * c3) * inv;
    b[2] = ((a[13] * s5 - a[14] * s4) + a[15] * s3) * inv;
    b[3] = ((-a[9] * s5 + a[10] * s4) - a[11] * s3) * inv;
    b[4] = ((-a[4] * c5 + a[6] * c2) - a[7] * c1) * inv;
    b[5] = ((a[0] * c5 - a[2] * c2) + a[3] * c1) * inv;
    b[6] = ((-a[12] * s5 + a[14] * s2) - a[15] * s1) * inv;
    b[7] = ((a[8] * s5 - a[10] * s2) + a[11] * s1) * inv;
    b[8] = ((a[4] * c4 - a[5] * c2) + a[7] * c0) * inv;
    b[9] = ((-a[0] * c4 + a[1] * c2) - a[3] * c0) * inv;
    b[10] = ((a[12] * s4 - a[13] * s2) + a[15] * s0) * inv;
    b[11] = ((-a[8] * s4 + a[9] * s2) - a[11] * s0) * inv;
    b[12] = ((-a[4] * c3 + a[5] * c1) - a[6] * c0) * inv;
    b[13] = ((a[0] * c3 - a[1] * c1) + a[2] * c0) * inv;
    b[14] = ((-a[12] * s3 + a[13] * s1) - a[14] * s0) * inv;
    b[15] = ((a[8] * s3 - a[9] * s1) + a[10] * s0) * inv;
    return r;
}

/* ---- transcendentals ---------------------------------------------------- */

/* Quadrant reduction shared by sin and cos: x = n*(pi/2) + r, |r| <~ pi/4,
 * three-term Cody-Waite subtraction.  Valid for |x| < ~1e5 (the shader feeds
 * arguments below 200); beyond that the result is still deterministic. */
static inline float ptm_reduce_pio2(float x, int32_t* quadrant) {
    float fn = rintf(x * 0.63661977236758134308f);
    float r = x - fn * 1.5703125f;
    r = r - fn * 4.837512969970703125e-4f;
    r = r - fn * 7.54978995489188e-8f;
    *quadrant = ptm_f2i(fn);
    return r;
}
static inline float ptm_sin_poly(float r) {
    float z = r * r;
    float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
    p = p * z - 1.6666654611e-1f;
    return (p * z) * r + r;
}
static inline float ptm_cos_poly(float r) {
    float z = r * r;
    float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    return (p * z) * z + (1.0f - 0.5f * z);
}
static inline float ptm_sin(float x) {
    if (!(ptm_abs(x) < 3.0e38f)) return x - x; /* inf/NaN -> NaN */
    int32_t q;
    float r = ptm_reduce_pio2(x, &q);
    float s = (q & 1) ? ptm_cos_poly(r) : ptm_sin_poly(r);
    return (q & 2) ? -s : s;
}
static inline float ptm_cos(float x) {
    if (!(ptm_abs(x) < 3.0e38f)) return x - x;
    int32_t q;
    float r = ptm_reduce_pio2(x, &q);
    float c = (q & 1) ? ptm_sin_poly(r) : ptm_cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

/* asin on |x| <= 0.5 (Cephes asinf kernel). */
static inline float ptm_asin_small(float x) {
    float z = x * x;
    float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    return (p * z) * x + x;
}
static inline float ptm_acos(float x) {
    const float pi = 3.14159265358979323846f;
    const float pio2 = 1.57079632679489661923f;
    if (x != x || x > 1.0f || x < -1.0f) return u2f(0x7fc00000u);
    if (x > 0.5f) {
        float s = sqrtf(0.5f * (1.0f - x));
        return 2.0f * ptm_asin_small(s);
    }
    if (x < -0.5f) {
        float s = sqrtf(0.5f * (1.0f + x));
        return pi - 2.0f * ptm_asin_small(s);
    }
    return pio2 - ptm_asin_small(x);
}

/* atan on x >= 0 (Cephes atanf). */
static inline float ptm_atan_pos(float x) {
    const float pio2 = 1.57079632679489661923f;
    const float pio4 = 0.78539816339744830962f;
    float y;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y = pio2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
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
/* GLSL atan(y, x).  atan(0,0) is defined here as 0. */
static inline float ptm_atan2(float y, float x) {
    const float pi = 3.14159265358979323846f;
    const float pio2 = 1.57079632679489661923f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x == 0.0f) {
        if (y > 0.0f) return pio2;
        if (y < 0.0f) return -pio2;
        return 0.0f;
    }
    float a = ptm_atan_pos(ptm_abs(y / x));
    if (x < 0.0f) a = pi - a;
    return (y < 0.0f) ? -a : a;
}

/* log2(x), x > 0 finite normal; 0 -> -inf, negative -> NaN, inf -> inf.
 * Subnormals are scaled by 2^24 first. */
static inline float ptm_log2(float x) {
    if (x != x || x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0xff800000u);
    uint32_t u = f2u(x);
    if (u == 0x7f800000u) return x;
    int32_t e = 0;
    if (u < 0x00800000u) { x = x * 16777216.0f; u = f2u(x); e = -24; }
    e += (int32_t)(u >> 23) - 127;
    float m = u2f((u & 0x007fffffu) | 0x3f800000u); /* [1,2) */
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
    float ln = (f * z) * p - 0.5f * z; /* ln(m) - f */
    /* log2(m) = (f + ln) * log2(e), split to keep accuracy */
    float r = ln * 0.44269504088896340735992f;
    r = r + f * 0.44269504088896340735992f;
    r = r + ln;
    r = r + f;
    return r + (float)e;
}
/* exp2(x): -inf -> 0, overflow -> inf, underflow below 2^-126 -> 0. */
static inline float ptm_exp2(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return u2f(0x7f800000u);
    if (x < -126.0f) return 0.0f;
    float fn = rintf(x);
    float f = x - fn; /* [-0.5, 0.5] */
    float p = 1.535336188319500e-4f * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    int32_t n = ptm_f2i(fn);
    return p * u2f((uint32_t)(n + 127) << 23);
}
/* GLSL pow(x, y) = exp2(y * log2(x)); pow(0, y>0) = 0. */
static inline float ptm_pow(float x, float y) { return ptm_exp2(y * ptm_log2(x)); }

#endif /* PTO_MATH_H */
