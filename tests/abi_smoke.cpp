// abi_smoke.cpp — a C++ host that uses librt_amd.so exactly as INTEGRATION.md tells a maintainer of the reference engine
// to: no Python, no torch; include/rt_amd.h, the HIP host API for one device buffer, and the library.
// Sequence: prepare_storage_buffers -> copy_buffer x6 (rt_upload_scene) -> run_compute (rt_render) -> read back;
// update_buffer (rt_update_spheres) -> run_compute again; then the multi-GPU calls on a one-rank communicator
// (rt_comm_unique_id / rt_comm_init / rt_render into a strip / rt_gather_strips), whose gathered frame must equal the first.
// Writes the frames as raw fp32 to argv[2]; tests/test_abi_smoke.py compares them with the ctypes path bit for bit.
//
// build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/abi_smoke.cpp -L ray_tracer_amd -lrt_amd
//        -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/ray_tracer_amd -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_amd.h"

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != 0) {                                                                        \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? rt_last_error(ctx) : "-"); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: abi_smoke <asset dir> <out file>\n"); return 2; }
    rt_ctx* ctx = nullptr;
    rt_scene* scene = nullptr;
    if (rt_scene_create(&scene) != 0) return 1;
    if (rt_scene_prepare_default(scene, argv[1]) < 0) { std::fprintf(stderr, "%s\n", rt_scene_last_error(scene)); return 1; }
    const float p0[3] = {0.0f, 0.1f, -0.3f};
    if (rt_scene_set_sphere(scene, 0, p0, 0.4f, 5) < 0) return 1;
    RtSceneArrays a;
    if (rt_scene_get_arrays(scene, &a) < 0) return 1;

    CHECK(rt_create(0, &ctx));
    uint32_t bits = 0;
    CHECK(rt_device_selftest(ctx, &bits));
    if (bits != 0x0f) { std::fprintf(stderr, "self-test bits %x\n", bits); return 1; }
    CHECK(rt_upload_scene(ctx, &a));

    const uint32_t W = 96, H = 57;   // 57 rows: not a multiple of anything
    PushConstants pc;
    rt_push_constants_default(&pc, W, H);
    pc.rayTraceParams.singleRender = 1;
    pc.rayTraceParams.sampleLimit = 3;
    pc.rayTraceParams.sphereCount = a.sphereCount;
    pc.rayTraceParams.objectCount = a.objectCount;
    pc.frameCount = 0;

    std::vector<float> f1((size_t)W * H * 4), f2(f1.size()), f3(f1.size());
    CHECK(rt_render(ctx, &pc, W, H, 0, 1, H, nullptr));
    CHECK(rt_sync(ctx));
    CHECK(rt_read_rgba_f32(ctx, f1.data(), f1.size()));

    // update_buffer: the dielectric sphere becomes a mirror, in place
    const float p1[3] = {0.2f, 0.0f, -0.2f};
    if (rt_scene_set_sphere(scene, 0, p1, 0.35f, 4) < 0) return 1;
    if (rt_scene_get_arrays(scene, &a) < 0) return 1;
    CHECK(rt_update_spheres(ctx, a.spheres, a.sphereCount));
    CHECK(rt_render(ctx, &pc, W, H, 0, 1, H, nullptr));
    CHECK(rt_sync(ctx));
    CHECK(rt_read_rgba_f32(ctx, f2.data(), f2.size()));

    // an error is a status code and a message, never an abort (the reference: VK_CHECK -> abort)
    if (rt_render(ctx, &pc, W, H, 0, 1, H + 1, nullptr) == 0 || std::strlen(rt_last_error(ctx)) == 0) {
        std::fprintf(stderr, "rows beyond the image were accepted\n");
        return 1;
    }

    // several GPUs, rehearsed with one rank: strip -> gather -> frame
    unsigned char id[RT_COMM_ID_BYTES];
    if (rt_comm_unique_id(id) != 0) { std::fprintf(stderr, "rt_comm_unique_id failed (RCCL missing?)\n"); return 1; }
    CHECK(rt_comm_init(ctx, id, 1, 0));
    float *dStrip = nullptr, *dFrame = nullptr;
    if (hipMalloc((void**)&dStrip, f1.size() * 4) != hipSuccess || hipMalloc((void**)&dFrame, f1.size() * 4) != hipSuccess) return 1;
    (void)hipMemset(dStrip, 0, f1.size() * 4);
    CHECK(rt_render(ctx, &pc, W, H, /*row0 = rank*/ 0, /*rowStride = ranks*/ 1, H, dStrip));
    CHECK(rt_gather_strips(ctx, dStrip, W, H, 0, dFrame));
    CHECK(rt_sync(ctx));
    if (hipMemcpy(f3.data(), dFrame, f3.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (std::memcmp(f3.data(), f2.data(), f2.size() * 4) != 0) { std::fprintf(stderr, "gathered frame differs from the direct render\n"); return 1; }
    CHECK(rt_comm_destroy(ctx));
    (void)hipFree(dStrip); (void)hipFree(dFrame);

    RtCounters cnt;
    CHECK(rt_get_counters(ctx, &cnt));
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    std::fwrite(f1.data(), 4, f1.size(), f);
    std::fwrite(f2.data(), 4, f2.size(), f);
    std::fclose(f);
    std::printf("abi_smoke ok: %ux%u, %llu rays traced, %llu reference rays\n", W, H, (unsigned long long)cnt.raysTraced,
                (unsigned long long)cnt.raysReference);
    rt_destroy(ctx);
    rt_scene_destroy(scene);
    return 0;
}
