/*
 * rt_det_math.h — deterministic fp32 primitives shared by host, device and oracle.
 *
 * Path tracing is chaotic: a 1-ulp difference in sin/cos/pow or one fused
 * multiply-add flips a hit/miss at a silhouette and the pixel diverges far
 * beyond the 1e-4 parity bar. So every arithmetic primitive that the GLSL
 * shader takes from the driver (GLSL 4.50 built-ins used by
 * shaders/raytrace.comp — call sites :187,292-293,318-319,358-364,419-423,
 * 467,478,547,552) is restated here ONCE, from the GLSL 4.50 spec formulas,
 * as plain IEEE-754 binary32 operations with a fixed evaluation order:
 *
 *   - only + - * / sqrt, comparisons and integer ops (all correctly rounded on
 *     x86-64 SSE and on gfx950 with hipcc's default
 *     -fhip-fp32-correctly-rounded-divide-sqrt);
 *   - no fused multiply-add: every translation unit that includes this header
 *     is compiled with -ffp-contract=off (the pragma below is a second guard
 *     under clang/hipcc), and rt_selftest_bits() detects a build that fused;
 *   - sin/cos/log2/exp2 are fixed polynomials (Cephes-style coefficients),
 *     so pow(x,y) = exp2(y*log2(x)) exactly as the GLSL spec derives it;
 *   - min/max have IEEE minNum/maxNum semantics (a NaN operand loses), which
 *     is what v_min_f32/v_max_f32 implement; GLSL leaves the NaN case
 *     undefined, so this is the defined choice for both sides.
 *
 * Nothing here is tuned for speed on either side; it is tuned to give the
 * same 32 bits on a Xeon/EPYC core and on a CDNA4 lane.
 */
#ifndef RT_DET_MATH_H
#define RT_DET_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ static __forceinline__
#else
#define RT_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define RT_PI 3.1415926535897932384f /* raytrace.comp:6 */
#define RT_INV_PI 0.3183098862f      /* raytrace.comp:7 */
#define RT_MISS_DST 99999999.0f      /* raytrace.comp:272,279 */

/* ---------------------------------------------------------------- bits */
RT_HD uint32_t rt_f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
RT_HD float rt_u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
RT_HD bool rt_isnan(float x) { return x != x; }
RT_HD bool rt_isinf(float x) { return (rt_f2u(x) & 0x7fffffffu) == 0x7f800000u; }
RT_HD float rt_abs(float x) { return rt_u2f(rt_f2u(x) & 0x7fffffffu); }

/* min/max: IEEE minNum/maxNum (the non-NaN operand wins). On gfx950 the
 * builtins lower to v_min_f32 / v_max_f32, which have exactly this rule; on
 * the host the same rule is spelled out. The sign of a zero result is not
 * observable anywhere in the path (results only feed comparisons, squares and
 * products that are added to non-negative sums). */
RT_HD float rt_min(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(a, b);
#else
    float m = (b < a) ? b : a;
    return (a != a) ? b : m;
#endif
}
RT_HD float rt_max(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(a, b);
#else
    float m = (a < b) ? b : a;
    return (a != a) ? b : m;
#endif
}
/* correctly rounded on both sides: x86 sqrtss, and on gfx950 the IEEE expansion
 * hipcc emits for sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt (its
 * default). HIP's __fsqrt_rn is NOT correctly rounded on gfx950 (1 ulp off
 * for ~15% of inputs, measured) and must not be used here. */
RT_HD float rt_sqrt(float x) { return __builtin_sqrtf(x); }
/* GLSL sign() */
RT_HD float rt_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
/* GLSL mix(x, y, a) = x*(1-a) + y*a */
RT_HD float rt_mix(float x, float y, float a) { return x * (1.f - a) + y * a; }
RT_HD float rt_clamp01(float x) { return rt_min(rt_max(x, 0.f), 1.f); }
/* GLSL smoothstep */
RT_HD float rt_smoothstep(float e0, float e1, float x) {
    float t = rt_clamp01((x - e0) / (e1 - e0));
    return t * t * (3.f - 2.f * t);
}
RT_HD float rt_radians(float deg) { return deg * 0.017453292519943295f; }

/* ---------------------------------------------------------------- sin / cos
 * Octant reduction with a three-part pi/4 (products with the small integer
 * octant are exact in fp32, so no FMA is needed), then degree-7 / degree-8
 * polynomials on [-pi/4, pi/4]. Valid for |x| < 8192; the path only calls it
 * with phi in [0, 2*pi] (raytrace.comp:410-413) and small camera angles. */
RT_HD void rt_sincos(float x, float* sn, float* cs) {
    float ax = rt_abs(x);
    int j = (int)(ax * 1.27323954473516f); /* 4/pi */
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = ax - y * 0.78515625f;
    r = r - y * 2.4187564849853515625e-4f;
    r = r - y * 3.77489497744594108e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    ps = ps * z * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z * z - 0.5f * z + 1.0f;
    int q = (j >> 1) & 3;
    float s = (q & 1) ? pc : ps;
    float c = (q & 1) ? ps : pc;
    if (q == 1 || q == 2) c = -c;
    if (q >= 2) s = -s;
    if (x < 0.f) s = -s;
    *sn = s;
    *cs = c;
}
RT_HD float rt_sin(float x) { float s, c; rt_sincos(x, &s, &c); return s; }
RT_HD float rt_cos(float x) { float s, c; rt_sincos(x, &s, &c); return c; }
RT_HD float rt_tan(float x) { float s, c; rt_sincos(x, &s, &c); return s / c; }

/* ---------------------------------------------------------------- log2 / exp2 / pow */
RT_HD float rt_log2(float x) {
    if (x != x) return x;
    if (x < 0.f) return rt_u2f(0x7fc00000u);
    if (x == 0.f) return rt_u2f(0xff800000u);
    uint32_t u = rt_f2u(x);
    if (u == 0x7f800000u) return x;
    int e = 0;
    if (u < 0x00800000u) { /* subnormal: scale by 2^24 (exact) */
        x = x * 16777216.f;
        u = rt_f2u(x);
        e = -24;
    }
    e += (int)(u >> 23) - 127;
    float m = rt_u2f((u & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m > 1.41421356237f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.f;
    float z = f * f;
    float p = 7.0376836292e-2f * f - 1.1514610310e-1f;
    p = p * f + 1.1676998740e-1f;
    p = p * f - 1.2420140846e-1f;
    p = p * f + 1.4249322787e-1f;
    p = p * f - 1.6668057665e-1f;
    p = p * f + 2.0000714765e-1f;
    p = p * f - 2.4999993993e-1f;
    p = p * f + 3.3333331174e-1f;
    float yv = p * f * z - 0.5f * z;
    float ln = f + yv;
    return ln * 1.44269504088896341f + (float)e;
}
RT_HD float rt_exp2(float x) {
    if (x != x) return x;
    if (x >= 128.f) return rt_u2f(0x7f800000u);
    if (x < -150.f) return 0.f;
    float t = x + 0.5f;
    int n = (int)t;
    if ((float)n > t) n -= 1; /* floor(x + 0.5) */
    float f = x - (float)n;   /* [-0.5, 0.5] */
    float p = 1.535336188319500e-4f * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    if (n >= -126) {
        if (n > 127) { /* only reachable for x in [127.5,128) */
            return p * rt_u2f((uint32_t)(127 + 127) << 23) * 2.f;
        }
        return p * rt_u2f((uint32_t)(n + 127) << 23);
    }
    return p * rt_u2f((uint32_t)(n + 24 + 127) << 23) * 5.9604644775390625e-8f;
}
/* GLSL 4.50 §8.2: pow(x,y) inherits its definition from exp2(y*log2(x)).
 * x < 0 gives NaN (undefined in GLSL); x == 0 with y > 0 gives 0. */
RT_HD float rt_pow(float x, float y) { return rt_exp2(y * rt_log2(x)); }

/* ---------------------------------------------------------------- vec3 */
typedef struct rt_vec3 { float x, y, z; } rt_vec3;

RT_HD rt_vec3 rt_v3(float x, float y, float z) { rt_vec3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD rt_vec3 rt_add(rt_vec3 a, rt_vec3 b) { return rt_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD rt_vec3 rt_sub(rt_vec3 a, rt_vec3 b) { return rt_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD rt_vec3 rt_mul(rt_vec3 a, rt_vec3 b) { return rt_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD rt_vec3 rt_scale(rt_vec3 a, float s) { return rt_v3(a.x * s, a.y * s, a.z * s); }
RT_HD rt_vec3 rt_neg(rt_vec3 a) { return rt_v3(-a.x, -a.y, -a.z); }
/* dot: ((ax*bx + ay*by) + az*bz), three roundings on the sums */
RT_HD float rt_dot(rt_vec3 a, rt_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD rt_vec3 rt_cross(rt_vec3 a, rt_vec3 b) {
    return rt_v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
/* normalize(v) = v * (1 / sqrt(dot(v,v))); normalize(0) = NaN as on GPUs
 * that compute 0 * inf. */
RT_HD rt_vec3 rt_normalize(rt_vec3 a) {
    float inv = 1.f / rt_sqrt(rt_dot(a, a));
    return rt_scale(a, inv);
}
/* reflect(I,N) = I - 2*dot(N,I)*N */
RT_HD rt_vec3 rt_reflect(rt_vec3 i, rt_vec3 n) {
    float k = 2.f * rt_dot(n, i);
    return rt_sub(i, rt_scale(n, k));
}
/* refract(I,N,eta): k = 1 - eta^2 (1 - dot(N,I)^2); k<0 -> 0 */
RT_HD rt_vec3 rt_refract(rt_vec3 i, rt_vec3 n, float eta) {
    float d = rt_dot(n, i);
    float k = 1.f - eta * eta * (1.f - d * d);
    if (k < 0.f) return rt_v3(0.f, 0.f, 0.f);
    float s = eta * d + rt_sqrt(k);
    return rt_sub(rt_scale(i, eta), rt_scale(n, s));
}

/* ---------------------------------------------------------------- mat4 (column-major, m[c*4+r]) */
/* (M * vec4(v, 0)).xyz */
RT_HD rt_vec3 rt_xform_dir(const float* m, rt_vec3 v) {
    return rt_v3((m[0] * v.x + m[4] * v.y) + m[8] * v.z,
                 (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
                 (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
/* (M * vec4(v, 1)).xyz */
RT_HD rt_vec3 rt_xform_point(const float* m, rt_vec3 v) {
    return rt_v3(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12],
                 ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13],
                 ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14]);
}

/* inverse(mat4) (raytrace.comp:292-293): adjugate / determinant through 2x2
 * sub-determinants. A pure function of the matrix, so host code evaluates it
 * once per object instead of twice per object per ray. */
RT_HD void rt_mat4_inverse(const float* m, float* o) {
    float a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3];
    float a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
    float a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11];
    float a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
    float b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10;
    float b02 = a00 * a13 - a03 * a10, b03 = a01 * a12 - a02 * a11;
    float b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
    float b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30;
    float b08 = a20 * a33 - a23 * a30, b09 = a21 * a32 - a22 * a31;
    float b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
    float det = ((((b00 * b11 - b01 * b10) + b02 * b09) + b03 * b08) - b04 * b07) + b05 * b06;
    float id = 1.f / det;
    o[0] = ((a11 * b11 - a12 * b10) + a13 * b09) * id;
    o[1] = ((a02 * b10 - a01 * b11) - a03 * b09) * id;
    o[2] = ((a31 * b05 - a32 * b04) + a33 * b03) * id;
    o[3] = ((a22 * b04 - a21 * b05) - a23 * b03) * id;
    o[4] = ((a12 * b08 - a10 * b11) - a13 * b07) * id;
    o[5] = ((a00 * b11 - a02 * b08) + a03 * b07) * id;
    o[6] = ((a32 * b02 - a30 * b05) - a33 * b01) * id;
    o[7] = ((a20 * b05 - a22 * b02) + a23 * b01) * id;
    o[8] = ((a10 * b10 - a11 * b08) + a13 * b06) * id;
    o[9] = ((a01 * b08 - a00 * b10) - a03 * b06) * id;
    o[10] = ((a30 * b04 - a31 * b02) + a33 * b00) * id;
    o[11] = ((a21 * b02 - a20 * b04) - a23 * b00) * id;
    o[12] = ((a11 * b07 - a10 * b09) - a12 * b06) * id;
    o[13] = ((a00 * b09 - a01 * b07) + a02 * b06) * id;
    o[14] = ((a31 * b01 - a30 * b03) - a32 * b00) * id;
    o[15] = ((a20 * b03 - a21 * b01) + a22 * b00) * id;
}


/* ---------------------------------------------------------------- textures
 * The snapshot's shader declares TextureBuffer[64] / TextureSampler[2] (raytrace.comp:122,148) and interpolates hit.uv
 * (:249-256) but never samples; the semantics below are this build's declared choice, from what the host sets up:
 * VK_FORMAT_R8G8B8A8_SRGB images (src/vk_engine.cpp:1158), VK_FILTER_NEAREST with sampler 0 = REPEAT and sampler 1 =
 * CLAMP_TO_EDGE (:525-531), selected by RenderObject.samplerIndex. */
/* texel index along one axis for VK_FILTER_NEAREST: floor(coord * size), wrapped (REPEAT) or clamped (CLAMP_TO_EDGE);
 * NaN and |coord * size| >= 2^30 give 0 */
RT_HD uint32_t rt_tex_index(float coord, uint32_t size, bool clampToEdge) {
    float f = coord * (float)size;
    if (!(rt_abs(f) < 1073741824.f)) return 0u;
    int i = (int)f;
    if ((float)i > f) i -= 1; /* floor */
    int n = (int)size;
    if (clampToEdge) return (uint32_t)(i < 0 ? 0 : (i > n - 1 ? n - 1 : i));
    int r = i % n;
    return (uint32_t)(r < 0 ? r + n : r);
}
/* one channel of an R8G8B8A8_SRGB texel -> linear (the sRGB transfer function, evaluated with rt_pow) */
RT_HD float rt_srgb8_to_linear(uint32_t byte) {
    float c = (float)byte / 255.f;
    return c <= 0.04045f ? c / 12.92f : rt_pow((c + 0.055f) / 1.055f, 2.4f);
}
/* The other three map slots of a material (src/vk_engine.cpp:1109-1141: map_Ks -> metalnessIndex, map_d -> alphaIndex,
 * map_bump -> bumpIndex; the snapshot's shader reads none of them). Declared semantics, all on the RED channel of
 * the texel at hit.uv, addressed like the albedo map:
 *   alpha      a triangle hit whose texel decodes below 0.5 is no hit (cut-out, inside calculateIntersections' triangle loop).
 *              rt_srgb8_to_linear is monotonic, (187) = 0.4969, (188) = 0.5029: the test is on the byte
 *   metalness  the texel's decoded value replaces material.reflectance (which the shader only compares with 0, raytrace.comp:509)
 *   bump       the texel's decoded value is a height; the interpolated normal is tilted by the height steps to the next texel
 *              of the row and of the column, along the triangle's dP/du and dP/dv (rt_bump_normal) */
#define RT_ALPHA_CUT_BYTE 188u
/* the next texel along one axis under the object's sampler */
RT_HD uint32_t rt_tex_next(uint32_t i, uint32_t size, bool clampToEdge) {
    return i + 1u < size ? i + 1u : (clampToEdge ? i : 0u);
}
/* n: interpolated normal before the front-face sign (object space); e1 = v1 - v0, e2 = v2 - v0 and (du1, dv1), (du2, dv2) the
 * same differences of the corners' uv; hx = h(x + 1, y) - h(x, y), hy = h(x, y + 1) - h(x, y) (rows run downwards, v upwards).
 * Where the height field is level, and on a triangle whose uv mapping is singular, the normal stays as it is. */
RT_HD rt_vec3 rt_bump_normal(rt_vec3 n, rt_vec3 e1, rt_vec3 e2, float du1, float dv1, float du2, float dv2, float hx, float hy) {
    float det = du1 * dv2 - du2 * dv1;
    if (!(rt_abs(det) > 0.f) || (hx == 0.f && hy == 0.f)) return n;
    float r = 1.f / det;
    rt_vec3 t = rt_scale(rt_sub(rt_scale(e1, dv2), rt_scale(e2, dv1)), r); /* dP/du */
    rt_vec3 b = rt_scale(rt_sub(rt_scale(e2, du1), rt_scale(e1, du2)), r); /* dP/dv */
    rt_vec3 g = rt_sub(rt_scale(rt_normalize(t), hx), rt_scale(rt_normalize(b), hy));
    return rt_sub(rt_normalize(n), g);
}

/* ---------------------------------------------------------------- RNG (raytrace.comp:158-163) */
RT_HD float rt_random(uint32_t* state) {
    uint32_t s = *state * 747796405u + 2891336453u;
    *state = s;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    r = (r >> 22) ^ r;
    /* 4294967295.f rounds to 2^32, so the division is an exact scaling */
    return (float)r / 4294967296.f;
}

/* ---------------------------------------------------------------- build self-test
 * Returns a word that differs if the compiler fused a*b+c, used a
 * non-IEEE division/sqrt, or flushed subnormals. Host, oracle and device
 * all must return RT_SELFTEST_EXPECT. */
#define RT_SELFTEST_EXPECT 0x0fu
RT_HD uint32_t rt_selftest_bits(volatile const float* in) {
    /* in = {1+2^-13, 1-2^-13, -1, 3, 1e-30, 1e-10, 2} supplied at run time */
    uint32_t ok = 0;
    float a = in[0], b = in[1], c = in[2];
    float p = a * b;      /* 1-2^-26 rounds to 1.0; an fma would keep -2^-26 */
    float s = p + c;
    if (s == 0.f) ok |= 1u;
    float q = in[0] / in[3];
    if (rt_f2u(q) == 0x3eaab000u) ok |= 2u; /* (1+2^-12)/3 correctly rounded */
    float d = in[4] * in[5]; /* 1e-40: subnormal must survive */
    if (d != 0.f && rt_f2u(d) == 0x000116c2u) ok |= 4u;
    float r = rt_sqrt(in[6]);
    if (rt_f2u(r) == 0x3fb504f3u) ok |= 8u;
    return ok;
}

#endif /* RT_DET_MATH_H */
