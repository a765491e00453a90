// diagnostic: which rt_det_math primitive differs between host and device?
#include "rt_det_math.h"
#include <cstdio>
#include <vector>
#define NV 26
__host__ __device__ inline void vals(float a, float b, float* o) {
    int k = 0; float s, c;
    rt_sincos(a * 6.2831855f, &s, &c);
    o[k++] = s; o[k++] = c; o[k++] = a / b; o[k++] = 1.f / (b + 0.25f);
    o[k++] = rt_sqrt(a); o[k++] = rt_pow(a, 5.f); o[k++] = rt_pow(a, 0.35f);
    o[k++] = rt_log2(b + 1e-3f); o[k++] = rt_exp2(a * 20.f - 10.f);
    rt_vec3 v = rt_normalize(rt_v3(a - 0.5f, b - 0.5f, a * b + 0.1f));
    o[k++] = v.x; o[k++] = v.y; o[k++] = v.z;
    o[k++] = rt_dot(v, rt_v3(b, a, 0.3f));
    rt_vec3 r = rt_refract(v, rt_normalize(rt_v3(0.1f, 1.f, b)), 0.5f + a);
    o[k++] = r.x; o[k++] = r.y; o[k++] = r.z;
    o[k++] = rt_smoothstep(0.f, 0.4f, a); o[k++] = rt_min(a, b); o[k++] = rt_max(a, b);
    o[k++] = a * b + b;
    uint32_t st = rt_f2u(a) ^ (rt_f2u(b) * 7u);
    o[k++] = rt_random(&st);
    o[k++] = rt_tan(a); o[k++] = (float)(int)(a * 1000.f); o[k++] = rt_mix(a, b, 0.3f); o[k++] = sqrtf(a); o[k++] = rt_abs(a - b);
}
__global__ void k(const float* a, const float* b, int n, float* o) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) vals(a[i], b[i], o + (size_t)i * NV); }
int main() {
    const int n = 4096; std::vector<float> a(n), b(n), h((size_t)n * NV), d((size_t)n * NV);
    uint32_t st = 12345u; for (int i = 0; i < n; i++) { a[i] = rt_random(&st); b[i] = rt_random(&st) + 1e-3f; }
    float *da, *db, *dd; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dd, (size_t)n * NV * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, n, dd); hipMemcpy(d.data(), dd, (size_t)n * NV * 4, hipMemcpyDeviceToHost);
    int bad[NV] = {0};
    for (int i = 0; i < n; i++) { vals(a[i], b[i], &h[(size_t)i * NV]);
        for (int j = 0; j < NV; j++) if (rt_f2u(h[(size_t)i * NV + j]) != rt_f2u(d[(size_t)i * NV + j])) { if (bad[j]++ < 2) printf("prim %d a=%.9g b=%.9g host=%.9g (%08x) dev=%.9g (%08x)\n", j, a[i], b[i], h[(size_t)i*NV+j], rt_f2u(h[(size_t)i*NV+j]), d[(size_t)i*NV+j], rt_f2u(d[(size_t)i*NV+j])); } }
    for (int j = 0; j < NV; j++) printf("prim %2d mismatches %d\n", j, bad[j]);
    return 0;
}
