/*
 * rt_probe.h — raw values of every GLSL built-in the shader uses, for the tests.
 *
 * rt_det_math.h is compiled into the HIP path and into the oracle, so the GPU parity tests compare those functions with
 * themselves. This listing evaluates each of them once on caller-supplied inputs and returns the bare results, so that
 * tests/test_glsl_builtins.py can compare them — from the oracle's build on the CPU and from the device's build through
 * rt_device_math_probe — with restatements of the GLSL 4.50 formulas written independently in numpy float32.
 * It contains no arithmetic of its own.
 *
 * in[32]:  0-2 I   3-5 N   6 eta   7 x   8 e0   9 e1   10 a   11 y   12-14 V   15 byte (0..255, as float)   16-31 M (column-major mat4)
 * out[64]: 0-2 reflect(I,N)   3-5 refract(I,N,eta)   6 smoothstep(e0,e1,x)   7 mix(x,y,a)   8 sign(x)   9-11 normalize(I)
 *          12 tan(x)   13 radians(x)   14-16 cross(I,V)   17 dot(I,V)   18 min(x,y)   19 max(x,y)   20 sin(x)   21 cos(x)
 *          22 log2(|x|)   23 exp2(x)   24 pow(|x|,y)   25 sqrt(|x|)   26 abs(x)   27 isnan(x)   28 isinf(x)
 *          29 random(bits of x): the float   30 ... the new state (as float bits)   31 1/x (IEEE division)
 *          32-34 (M*vec4(V,0)).xyz   35-37 (M*vec4(V,1)).xyz   38-53 inverse(M)
 *          54 srgb8_to_linear(byte = in[15])   55 tex_index(x, 37, repeat)   56 tex_index(x, 37, clamp)   57-63 zero
 */
#ifndef RT_PROBE_H
#define RT_PROBE_H

#include "rt_det_math.h"

RT_HD void rt_math_probe(const float* in, float* out) {
    const rt_vec3 I = rt_v3(in[0], in[1], in[2]), N = rt_v3(in[3], in[4], in[5]), V = rt_v3(in[12], in[13], in[14]);
    const float eta = in[6], x = in[7], e0 = in[8], e1 = in[9], a = in[10], y = in[11];
    rt_vec3 r = rt_reflect(I, N);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    r = rt_refract(I, N, eta);
    out[3] = r.x; out[4] = r.y; out[5] = r.z;
    out[6] = rt_smoothstep(e0, e1, x);
    out[7] = rt_mix(x, y, a);
    out[8] = rt_sign(x);
    r = rt_normalize(I);
    out[9] = r.x; out[10] = r.y; out[11] = r.z;
    out[12] = rt_tan(x);
    out[13] = rt_radians(x);
    r = rt_cross(I, V);
    out[14] = r.x; out[15] = r.y; out[16] = r.z;
    out[17] = rt_dot(I, V);
    out[18] = rt_min(x, y);
    out[19] = rt_max(x, y);
    out[20] = rt_sin(x);
    out[21] = rt_cos(x);
    out[22] = rt_log2(rt_abs(x));
    out[23] = rt_exp2(x);
    out[24] = rt_pow(rt_abs(x), y);
    out[25] = rt_sqrt(rt_abs(x));
    out[26] = rt_abs(x);
    out[27] = rt_isnan(x) ? 1.f : 0.f;
    out[28] = rt_isinf(x) ? 1.f : 0.f;
    uint32_t st = rt_f2u(x);
    out[29] = rt_random(&st);
    out[30] = rt_u2f(st);
    out[31] = 1.f / x;
    r = rt_xform_dir(in + 16, V);
    out[32] = r.x; out[33] = r.y; out[34] = r.z;
    r = rt_xform_point(in + 16, V);
    out[35] = r.x; out[36] = r.y; out[37] = r.z;
    rt_mat4_inverse(in + 16, out + 38);
    out[54] = rt_srgb8_to_linear((uint32_t)in[15]);
    out[55] = (float)rt_tex_index(x, 37u, false);
    out[56] = (float)rt_tex_index(x, 37u, true);
    for (int k = 57; k < 64; k++) out[k] = 0.f;
}

#endif /* RT_PROBE_H */
