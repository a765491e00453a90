// Micro-benchmark: how fast does a CU's L1 serve 64-byte records gathered by
// random index? A: each lane reads its own record with 4 dwordx4 loads.
// B: the 4 lanes of a quad read ONE record per instruction (16 B each, 64 B
// contiguous), 4 instructions cover the quad's 4 records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void kA(const float4* __restrict__ tab, const uint32_t* __restrict__ idx, int iters, uint32_t mask, float* out) {
    uint32_t i = idx[blockIdx.x * blockDim.x + threadIdx.x];
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const float4* p = tab + 4 * (size_t)i;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        i = (__float_as_uint(d.w) ^ (i * 2654435761u)) & mask;  // dependent next index
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void kB(const float4* __restrict__ tab, const uint32_t* __restrict__ idx, int iters, uint32_t mask, float* out) {
    uint32_t i = idx[blockIdx.x * blockDim.x + threadIdx.x];
    const uint32_t j = threadIdx.x & 3u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float4 v[4];

        // quad broadcast of lane k's index: DPP quad_perm [k,k,k,k]
        uint32_t i0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i, 0x00, 0xf, 0xf, true);
        uint32_t i1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i, 0x55, 0xf, 0xf, true);
        uint32_t i2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i, 0xaa, 0xf, 0xf, true);
        uint32_t i3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i, 0xff, 0xf, 0xf, true);
        v[0] = tab[4 * (size_t)i0 + j]; v[1] = tab[4 * (size_t)i1 + j]; v[2] = tab[4 * (size_t)i2 + j]; v[3] = tab[4 * (size_t)i3 + j];
        // (no transpose here: only the memory behaviour is measured)
        acc += v[0].x + v[1].y + v[2].z + v[3].w;
        float w = j == 0 ? v[0].w : j == 1 ? v[1].w : j == 2 ? v[2].w : v[3].w;
        i = (__float_as_uint(w) ^ (i * 2654435761u)) & mask;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// A with only the first `active` lanes of every wave switched on: does the vector memory pipeline charge a load by the
// lanes that take part, or by the wave instruction? (trace_wave's interior step runs with ~35 of 64 lanes)
__global__ void kAm(const float4* __restrict__ tab, const uint32_t* __restrict__ idx, int iters, uint32_t mask, float* out, uint32_t active) {
    uint32_t i = idx[blockIdx.x * blockDim.x + threadIdx.x];
    float acc = 0.f;
    // active < 100: the first `active` lanes; 100 + k: every k-th lane (scattered over all quads); 200 + p: a pseudo-random p % of the lanes
    const uint32_t lane = threadIdx.x & 63u;
    const bool on = active < 100u ? lane < active : active < 200u ? (lane % (active - 100u)) == 0u : ((lane * 2654435761u + blockIdx.x * 40503u) >> 16) % 100u < active - 200u;
    if (on) {
        for (int it = 0; it < iters; it++) {
            const float4* p = tab + 4 * (size_t)i;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            i = (__float_as_uint(d.w) ^ (i * 2654435761u)) & mask;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// C: four lanes per record, ONE load instruction per record (lane j of a quad reads bytes 16 j .. 16 j + 15): the fetch of a
// "four lanes per ray" traversal. A wave instruction serves 16 records.
__global__ void kC(const float4* __restrict__ tab, const uint32_t* __restrict__ idx, int iters, uint32_t mask, float* out) {
    uint32_t i = idx[(blockIdx.x * blockDim.x + threadIdx.x) >> 2];
    const uint32_t j = threadIdx.x & 3u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const float4 v = tab[4 * (size_t)i + j];
        acc += v.x;
        const uint32_t w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v.w), 0xff, 0xf, 0xf, true);  // lane 3's word to the quad
        i = (w ^ (i * 2654435761u)) & mask;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const uint32_t nrec = 1u << 18;  // 16 MiB of 64-B records
    std::vector<float> h((size_t)nrec * 16);
    uint32_t s = 1;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; uint32_t u = s >> 8; memcpy(&x, &u, 4); }
    float4* tab; uint32_t* idx; float* out;
    const int blocks = 256 * 5, threads = 256, n = blocks * threads, iters = 200;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&idx, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> hi(n);
    for (auto& x : hi) { s = s * 1664525u + 1013904223u; x = (s >> 8) & (nrec - 1); }
    CHECK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t bits : {8u, 12u, 15u, 18u}) {  // table footprint: 16 KB (L1), 256 KB, 2 MB (L2), 16 MB
        uint32_t mask = (1u << bits) - 1;
        for (int rep = 0; rep < 2; rep++) {
            float msA, msB;
            hipEventRecord(e0); kA<<<blocks, threads>>>(tab, idx, iters, mask, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msA, e0, e1);
            hipEventRecord(e0); kB<<<blocks, threads>>>(tab, idx, iters, mask, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msB, e0, e1);
            double recs = (double)n * iters;
            if (rep) printf("footprint %8u B: A (lane-private 4x16B) %7.3f ms = %6.1f Grec/s %6.2f TB/s | B (quad-coalesced) %7.3f ms = %6.1f Grec/s %6.2f TB/s\n",
                            (mask + 1) * 64, msA, recs / msA / 1e6, recs * 64 / msA / 1e9, msB, recs / msB / 1e6, recs * 64 / msB / 1e9);
        }
    }
    // lanes taking part in A's loads, and the one-load-per-record quad fetch (table in L2: 2 MB)
    {
        const uint32_t mask = (1u << 15) - 1;
        for (uint32_t active : {64u, 48u, 32u, 16u, 8u, 102u, 104u, 108u, 275u, 250u, 225u}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0); kAm<<<blocks, threads>>>(tab, idx, iters, mask, out, active); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double winstr = (double)n / 64 * iters * 4;  // wave-level load instructions
            const double frac = active < 100u ? active / 64.0 : active < 200u ? 1.0 / (active - 100u) : (active - 200u) / 100.0;
            printf("A, lanes %3u (%s, %.0f %% active): %7.3f ms, %6.1f Grec/s, %5.2f ns per wave-level load on a CU (%.1f clocks at 2.4 GHz)\n", active,
                   active < 100u ? "first n" : active < 200u ? "every k-th" : "random", frac * 100, ms,
                   (double)n * frac * iters / ms / 1e6, ms * 1e6 / (winstr / 256), ms * 1e6 / (winstr / 256) * 2.4);
        }
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0); kC<<<blocks, threads>>>(tab, idx, iters, mask, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double winstr = (double)n / 64 * iters;
        printf("C, four lanes per record, one load each: %7.3f ms, %6.1f Grec/s, %5.2f ns per wave-level load on a CU (%.1f clocks)\n", ms,
               (double)n / 4 * iters / ms / 1e6, ms * 1e6 / (winstr / 256), ms * 1e6 / (winstr / 256) * 2.4);
    }
    return 0;
}
