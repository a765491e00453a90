// Host scene surface under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on the pool).
//
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
//       -Iinclude -Iray_tracer_amd/csrc ray_tracer_amd/csrc/scene.cpp tools/sanitize_scene.cpp -o /tmp/sanitize_scene
//   /tmp/sanitize_scene assets 400
//
// 1. the default scene and every OBJ / MTL under the asset directory through rt_scene_read_obj / rt_scene_read_mtl;
// 2. N mutated copies of every small OBJ / MTL (truncated, characters deleted or replaced, indices made huge, zero or
//    negative, lines duplicated): the loader may accept or refuse each, it must not read or write out of bounds;
// 3. argument errors of the scene half of the ABI.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <sstream>
#include <string>
#include <unistd.h>
#include <vector>

#include "rt_amd.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd() {
    rng_state = rng_state * 747796405u + 2891336453u;
    uint32_t r = ((rng_state >> ((rng_state >> 28) + 4u)) ^ rng_state) * 277803737u;
    return (r >> 22) ^ r;
}

static std::string slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static std::vector<std::string> list(const std::string& dir, const char* ext) {
    std::vector<std::string> out;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            std::string n = e->d_name;
            if (n.size() > strlen(ext) && n.compare(n.size() - strlen(ext), strlen(ext), ext) == 0) out.push_back(dir + "/" + n);
        }
        closedir(d);
    }
    return out;
}

static std::string mutate(const std::string& src) {
    std::string s = src;
    const uint32_t kind = rnd() % 8u;
    if (s.empty()) return s;
    switch (kind) {
    case 0: s.resize(rnd() % s.size()); break;                                   // truncated anywhere
    case 1: for (int k = 0; k < 8; ++k) s.erase(rnd() % s.size(), 1 + rnd() % 3u); break;
    case 2: for (int k = 0; k < 8; ++k) s[rnd() % s.size()] = "0123456789/ -.\nfv#e"[rnd() % 19u]; break;
    case 3: {                                                                    // face indices out of range
        size_t p = s.find("\nf ");
        static const char* bad[] = {"\nf 0 0 0", "\nf -1 -2 -3", "\nf 4000000000 2 3", "\nf 1/99999999/1 2/2/2 3/3/3",
                                    "\nf 1//2147483647 2//2 3//3", "\nf 1 2", "\nf", "\nf 1/ 2/ 3/", "\nf a b c"};
        if (p != std::string::npos) s.insert(p, bad[rnd() % 9u]);
        break;
    }
    case 4: {                                                                    // a line many times over
        size_t a = rnd() % s.size(), b = s.find('\n', a);
        if (b != std::string::npos) { std::string ln = s.substr(a, b - a + 1); for (int k = 0; k < 50; ++k) s.insert(b + 1, ln); }
        break;
    }
    case 5: s.insert(rnd() % s.size(), "\nusemtl nothing_of_that_name\n"); break;
    case 6: s.insert(rnd() % s.size(), "\nv 1e39 -1e39 nan\nvn inf 0 0\nvt 1\n"); break;
    default: s.insert(rnd() % s.size(), std::string(1 + rnd() % 5000u, ' ')); break; // very long line of spaces
    }
    return s;
}

int main(int argc, char** argv) {
    const std::string assets = argc > 1 ? argv[1] : "assets";
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    RtPlacement pl;
    rt_placement_default(&pl);
    int loaded = 0, refused = 0;

    // 1. everything the repository ships
    {
        rt_scene* s = nullptr;
        if (rt_scene_create(&s)) return 1;
        if (rt_scene_prepare_default(s, assets.c_str())) { printf("default scene: %s\n", rt_scene_last_error(s)); return 1; }
        for (const std::string& dir : {assets, assets + "/bobadog", assets + "/sponza2"}) {
            for (const std::string& m : list(dir, ".mtl")) (rt_scene_read_mtl(s, m.c_str()) == 0 ? loaded : refused)++;
            for (const std::string& o : list(dir, ".obj")) {
                (rt_scene_read_obj(s, o.c_str(), &pl, 0) == 0 ? loaded : refused)++;
                (rt_scene_read_obj(s, o.c_str(), &pl, 1) == 0 ? loaded : refused)++;      // second time: the instancing branch
            }
        }
        RtSceneArrays a;
        if (rt_scene_get_arrays(s, &a)) return 1;
        printf("shipped assets: %d loads ok, %d refused; %u triangles, %u objects, %u nodes\n", loaded, refused, a.triangleCount,
               a.objectCount, a.bvhNodeCount);
        rt_scene_destroy(s);
    }

    // 2. mutated inputs
    char tmpl[] = "/tmp/rt_sanitize_XXXXXX";
    const char* tmp = mkdtemp(tmpl);
    if (!tmp) return 1;
    std::vector<std::string> seeds;
    for (const std::string& dir : {assets, assets + "/bobadog"})
        for (const char* ext : {".obj", ".mtl"})
            for (const std::string& f : list(dir, ext)) {
                std::string c = slurp(f);
                if (c.size() < 200000) seeds.push_back((ext[1] == 'o' ? "o" : "m") + c);
            }
    int ok = 0, bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const std::string& seed = seeds[rnd() % seeds.size()];
        const bool isObj = seed[0] == 'o';
        std::string body = mutate(seed.substr(1));
        if (rnd() % 4u == 0) body = mutate(body);
        const std::string path = std::string(tmp) + (isObj ? "/m.obj" : "/m.mtl");
        { std::ofstream f(path, std::ios::binary); f << body; }
        rt_scene* s = nullptr;
        if (rt_scene_create(&s)) return 1;
        const int rc = isObj ? rt_scene_read_obj(s, path.c_str(), &pl, (int)(rnd() % 3u)) : rt_scene_read_mtl(s, path.c_str());
        (rc == 0 ? ok : bad)++;
        RtSceneArrays a;
        if (rc == 0 && rt_scene_get_arrays(s, &a) == 0) {
            // what was accepted must be self-consistent: every node's triangle range and child index inside the arrays
            for (uint32_t i = 0; i < a.bvhNodeCount; ++i) {
                const BVHNode& n = a.bvhNodes[i];
                if (n.triCount ? (uint64_t)n.index + n.triCount > a.triangleCount : (uint64_t)n.index + 1 >= a.bvhNodeCount) {
                    printf("round %d: node %u out of range\n", r, i);
                    return 2;
                }
            }
            for (uint32_t i = 0; i < a.triangleCount; ++i)
                for (int k = 0; k < 3; ++k)
                    if ((&a.triangles[i].v0)[k] >= a.triPointCount) { printf("round %d: triangle %u corner out of range\n", r, i); return 2; }
        }
        rt_scene_destroy(s);
        unlink(path.c_str());
    }
    rmdir(tmp);
    printf("mutated inputs: %d accepted, %d refused, none out of bounds\n", ok, bad);

    // 3. argument errors
    {
        rt_scene* s = nullptr;
        rt_scene_create(&s);
        int e = 0;
        e += rt_scene_create(nullptr) != 0;
        e += rt_scene_read_obj(s, "/nonexistent/file.obj", &pl, 0) != 0;
        e += rt_scene_read_obj(s, nullptr, &pl, 0) != 0;
        e += rt_scene_read_mtl(s, "/nonexistent/file.mtl") != 0;
        e += rt_scene_add_mesh(s, "k", nullptr, nullptr, nullptr, 3, &pl, 0) != 0;
        const float p3[3] = {0, 0, 0};
        e += rt_scene_set_sphere(s, 10, p3, 1.f, 0) != 0;
        e += rt_scene_get_arrays(s, nullptr) != 0;
        printf("argument errors reported: %d of 7\n", e);
        rt_scene_destroy(s);
        if (e != 7) return 3;
    }
    return 0;
}
