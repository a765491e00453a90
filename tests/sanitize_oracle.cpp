// The CPU oracle under AddressSanitizer + UBSan (test infrastructure checking test infrastructure): the default Cornell
// scene with three spheres and the klein bottle, a few samples per pixel with environment on and off, every debug mode,
// plus a batch of rays through oracle_trace_rays. Built and run by tests/test_sanitizers.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "raytrace_oracle.h"
#include "rt_amd.h"

int main(int argc, char** argv) {
    const std::string assets = argc > 1 ? argv[1] : "assets";
    rt_scene* s = nullptr;
    if (rt_scene_create(&s) || rt_scene_prepare_default(s, assets.c_str())) return 1;
    const float a[3] = {0.f, 0.1f, -0.3f}, b[3] = {0.5f, 0.1f, 0.f}, c[3] = {-0.5f, 0.1f, 0.f};
    rt_scene_set_sphere(s, 0, a, 0.4f, 5);
    rt_scene_set_sphere(s, 1, b, 0.4f, 4);
    rt_scene_set_sphere(s, 2, c, 0.4f, 0);
    RtPlacement pl;
    rt_placement_default(&pl);
    pl.scale[0] = pl.scale[1] = pl.scale[2] = 0.5f;
    pl.rotation[1] = 30.f;
    if (rt_scene_read_obj(s, (assets + "/klein_bottle.obj").c_str(), &pl, 5)) return 1;
    RtSceneArrays arr;
    if (rt_scene_get_arrays(s, &arr)) return 1;

    const uint32_t W = 48, H = 36;
    std::vector<float> img(W * H * 4, 0.f);
    double sum = 0;
    for (int env = 0; env < 2; ++env)
        for (int debug = -1; debug <= 2; ++debug) {
            PushConstants pc;
            rt_push_constants_default(&pc, W, H);
            pc.rayTraceParams.singleRender = 1;
            pc.rayTraceParams.sampleLimit = 2;
            pc.rayTraceParams.debug = debug;
            pc.environment.lightDir[3] = (float)env;
            pc.rayTraceParams.sphereCount = arr.sphereCount;
            pc.rayTraceParams.objectCount = arr.objectCount;
            OracleCounters cnt;
            memset(&cnt, 0, sizeof cnt);
            if (oracle_render(&arr, &pc, W, H, 0, 1, H, img.data(), &cnt, 2)) return 2;
            for (float v : img) sum += std::isfinite(v) ? v : 0.0;
        }
    std::vector<float> o(3 * 4096), d(3 * 4096);
    uint32_t st = 7;
    for (size_t i = 0; i < 4096; ++i) {
        for (int k = 0; k < 3; ++k) { o[3 * i + k] = oracle_random(&st) * 2.f - 1.f; d[3 * i + k] = oracle_random(&st) * 2.f - 1.f; }
        if (i % 97 == 0) d[3 * i] = d[3 * i + 1] = 0.f;      // axis-parallel rays: infinities in invDir
        if (i % 389 == 0) d[3 * i + 2] = 0.f;                // zero direction on those: NaNs everywhere
    }
    std::vector<RtHit> hits(4096);
    if (oracle_trace_rays(&arr, arr.sphereCount, arr.objectCount, 4096, o.data(), d.data(), hits.data())) return 3;
    printf("oracle under sanitizers ok (checksum %.3f, selftest %x)\n", sum, oracle_selftest());
    rt_scene_destroy(s);
    return 0;
}
