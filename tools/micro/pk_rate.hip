// Issue rate of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 against their scalar forms on gfx950.
// hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s) {
    f2 a0 = {threadIdx.x * 1.0f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 m = {s, s * 1.0001f};
    for (int i = 0; i < N_IT; i++) {
        if (MODE == 0) {  // 16 scalar muls
#define S(v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v.x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v.y) : "v"(m.y));
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
        } else if (MODE == 1) {  // 8 packed muls = same math
#define P(v) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(m));
            P(a0) P(a1) P(a2) P(a3) P(a4) P(a5) P(a6) P(a7)
        } else if (MODE == 2) {
#define A(v) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(m));
            A(a0) A(a1) A(a2) A(a3) A(a4) A(a5) A(a6) A(a7)
        } else if (MODE == 3) {
#define F(v) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(m));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
        } else {
#define SA(v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v.x) : "v"(m.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v.y) : "v"(m.y));
            SA(a0) SA(a1) SA(a2) SA(a3) SA(a4) SA(a5) SA(a6) SA(a7)
        }
    }
    f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}
template <int MODE>
double run(float* d, const char* name, int instrPerIt) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;  // 8 blocks/CU = 8 waves/SIMD
    k<MODE><<<blocks, 256>>>(d, 1.0f);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(d, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)blocks * 4 * N_IT * instrPerIt;  // wave-instructions
    printf("%-14s %8.3f ms  %7.2f G wave-instr/s  (%.2f per clk per SIMD at 2.4 GHz)\n", name, ms, instr / ms / 1e6, instr / (ms * 1e-3) / (256 * 4 * 2.4e9));
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_mul_f32 x16", 16);
    run<1>(d, "v_pk_mul x8", 8);
    run<4>(d, "v_add_f32 x16", 16);
    run<2>(d, "v_pk_add x8", 8);
    run<3>(d, "v_pk_fma x8", 8);
    return 0;
}
