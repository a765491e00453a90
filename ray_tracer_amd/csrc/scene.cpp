// scene.cpp — host scene surface of the MI355X path tracer.
//
// Keeps what the reference's VulkanEngine keeps on the CPU side
// (src/vk_engine.h:270-287): the AoS vectors spheres / rayMaterials /
// triPoints / triangles / objects / bvhNodes / centroids and the two caches
// loadedObjects / loadedMaterials, and fills them with the same semantics as
//   read_obj                 src/vk_engine.cpp:800-1037
//   read_mtl                 src/vk_engine.cpp:1060-1167
//   cornell_box              src/vk_engine.cpp:638-678
//   prepare_storage_buffers  src/vk_engine.cpp:680-758
//   build_bvh & friends      src/vk_engine.cpp:1169-1337
//   camera / constants       src/vk_engine.cpp:1631-1661
// No GPU call happens in this file; the device half (rt_device.hip) consumes
// the arrays through RtSceneArrays exactly like the reference's copy_buffer.
//
// Host matrix math restates the formulas glm 0.9.9.7 publishes for
// translate / rotate / scale / mat*mat / inverse; trig goes through
// rt_det_math.h so that scenes are bit-reproducible across libm versions.

#include "rt_amd.h"
#include "rt_det_math.h"
#include "scene_internal.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

namespace {

static_assert(sizeof(Sphere) == 32, "Sphere std140 size");
static_assert(offsetof(Sphere, radius) == 12 && offsetof(Sphere, materialIndex) == 16, "Sphere layout");
static_assert(sizeof(Triangle) == 48 && offsetof(Triangle, binormal) == 16 && offsetof(Triangle, tangent) == 32, "Triangle layout");
static_assert(sizeof(TrianglePoint) == 32, "TrianglePoint size");
static_assert(sizeof(RayMaterial) == 64 && offsetof(RayMaterial, emissionColor) == 16 &&
              offsetof(RayMaterial, emissionStrength) == 28 && offsetof(RayMaterial, reflectance) == 32 &&
              offsetof(RayMaterial, ior) == 36 && offsetof(RayMaterial, albedoIndex) == 40 &&
              offsetof(RayMaterial, bumpIndex) == 52, "RayMaterial layout");
static_assert(sizeof(RenderObject) == 80 && offsetof(RenderObject, smoothShade) == 64 &&
              offsetof(RenderObject, samplerIndex) == 76, "RenderObject layout");
static_assert(sizeof(BVHNode) == 32 && offsetof(BVHNode, index) == 24 && offsetof(BVHNode, triCount) == 28, "BVHNode layout");
static_assert(sizeof(CameraInfo) == 96 && offsetof(CameraInfo, pos) == 64 && offsetof(CameraInfo, nearPlane) == 76 &&
              offsetof(CameraInfo, aspectRatio) == 80 && offsetof(CameraInfo, fov) == 84, "CameraInfo layout");
static_assert(sizeof(EnvironmentData) == 64 && offsetof(EnvironmentData, lightDir) == 48, "EnvironmentData layout");
static_assert(sizeof(RayTracerData) == 40, "RayTracerData size");
static_assert(sizeof(PushConstants) == 208 && offsetof(PushConstants, environment) == 96 &&
              offsetof(PushConstants, rayTraceParams) == 160 && offsetof(PushConstants, frameCount) == 200, "PushConstants layout");

// ---- 4x4 column-major helpers (glm conventions) ---------------------------
struct Mat4 { float m[16]; };

Mat4 mat_identity() {
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
    return r;
}
// column c of (a*b) = ((a0*b[c][0] + a1*b[c][1]) + a2*b[c][2]) + a3*b[c][3]
Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float s = a.m[0 + row] * b.m[c * 4 + 0] + a.m[4 + row] * b.m[c * 4 + 1];
            s = s + a.m[8 + row] * b.m[c * 4 + 2];
            s = s + a.m[12 + row] * b.m[c * 4 + 3];
            r.m[c * 4 + row] = s;
        }
    return r;
}
Mat4 mat_translate(const float v[3]) {
    Mat4 r = mat_identity();
    r.m[12] = v[0]; r.m[13] = v[1]; r.m[14] = v[2];
    return r;
}
Mat4 mat_scale(const float v[3]) {
    Mat4 r = mat_identity();
    r.m[0] = v[0]; r.m[5] = v[1]; r.m[10] = v[2];
    return r;
}
// axis-angle rotation about a unit axis, the formula glm::rotate documents
Mat4 mat_rotate(float angleRad, float ax, float ay, float az) {
    float s, c;
    rt_sincos(angleRad, &s, &c);
    float t[3] = {(1.f - c) * ax, (1.f - c) * ay, (1.f - c) * az};
    Mat4 r = mat_identity();
    r.m[0] = c + t[0] * ax;       r.m[1] = t[0] * ay + s * az;  r.m[2] = t[0] * az - s * ay;
    r.m[4] = t[1] * ax - s * az;  r.m[5] = c + t[1] * ay;       r.m[6] = t[1] * az + s * ax;
    r.m[8] = t[2] * ax + s * ay;  r.m[9] = t[2] * ay - s * ax;  r.m[10] = c + t[2] * az;
    return r;
}

Mat4 placement_matrix(const RtPlacement& p) {
    Mat4 m = mat_translate(p.position);
    m = mat_mul(m, mat_rotate(rt_radians(p.rotation[0]), 1.f, 0.f, 0.f));
    m = mat_mul(m, mat_rotate(rt_radians(p.rotation[1]), 0.f, 1.f, 0.f));
    m = mat_mul(m, mat_rotate(rt_radians(p.rotation[2]), 0.f, 0.f, 1.f));
    m = mat_mul(m, mat_scale(p.scale));
    return m;
}

// ---- bounding boxes (src/vk_engine.h:81-115) ------------------------------
struct Box {
    float lo[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    float hi[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
    void grow_point(const float* p) {  // 3 components only
        for (int i = 0; i < 3; i++) {
            lo[i] = p[i] < lo[i] ? p[i] : lo[i];
            hi[i] = hi[i] < p[i] ? p[i] : hi[i];
        }
    }
    void grow_box(const Box& b) {  // all four components
        for (int i = 0; i < 4; i++) {
            lo[i] = (b.lo[i] < lo[i]) ? b.lo[i] : lo[i];
            hi[i] = (hi[i] < b.hi[i]) ? b.hi[i] : hi[i];
        }
    }
    float surface_area() const {  // "half area": xy + yz + zx
        float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

std::string dir_of(const std::string& path) { return path.substr(0, path.rfind('/') + 1); }

bool parse_float(const std::string& s, float& out) {
    const char* b = s.c_str();
    char* e = nullptr;
    out = strtof(b, &e);
    return e != b;
}
bool parse_int(const std::string& s, int& out) {
    const char* b = s.c_str();
    char* e = nullptr;
    long v = strtol(b, &e, 10);
    out = (int)v;
    return e != b;
}

}  // namespace

// ---------------------------------------------------------------- rt_scene
struct rt_scene : public RtSceneHost {
    std::string error;
    int fail(const std::string& msg) { error = msg; return -1; }

    // --- BVH -------------------------------------------------------------
    void update_bounds(uint32_t idx) {
        BVHNode& n = bvhNodes[idx];
        Box b;
        for (uint32_t i = 0; i < n.triCount; i++) {
            const Triangle& t = triangles[n.index + i];
            b.grow_point(triPoints[t.v0].position);
            b.grow_point(triPoints[t.v1].position);
            b.grow_point(triPoints[t.v2].position);
        }
        n.boundsX[0] = b.lo[0]; n.boundsX[1] = b.hi[0];
        n.boundsY[0] = b.lo[1]; n.boundsY[1] = b.hi[1];
        n.boundsZ[0] = b.lo[2]; n.boundsZ[1] = b.hi[2];
    }

    // binned SAH over centroids, 20 bins, first-best-wins in axis order x,y,z
    float find_split(const BVHNode& node, int& axis, float& splitPos) {
        const unsigned B = RT_BVH_BINS;
        float best = 1e30f;
        for (int a = 0; a < 3; a++) {
            float mn = 1e30f, mx = -1e30f;
            for (uint32_t i = 0; i < node.triCount; i++) {
                float c = centroids[node.index + i].v[a];
                mn = mn < c ? mn : c;
                mx = mx < c ? c : mx;
            }
            if (mn == mx) continue;

            Box binBox[RT_BVH_BINS];
            uint32_t binCount[RT_BVH_BINS] = {0};
            float scale = (float)B / (mx - mn);
            for (uint32_t i = 0; i < node.triCount; i++) {
                const Triangle& t = triangles[node.index + i];
                float f = floorf((centroids[node.index + i].v[a] - mn) * scale);
                float lim = (float)(B - 1);
                int bi = (int)(lim < f ? lim : f);
                binCount[bi]++;
                binBox[bi].grow_point(triPoints[t.v0].position);
                binBox[bi].grow_point(triPoints[t.v1].position);
                binBox[bi].grow_point(triPoints[t.v2].position);
            }

            float leftArea[RT_BVH_BINS - 1], rightArea[RT_BVH_BINS - 1];
            float leftCount[RT_BVH_BINS - 1], rightCount[RT_BVH_BINS - 1];
            Box leftBox, rightBox;
            int leftSum = 0, rightSum = 0;
            for (unsigned i = 0; i < B - 1; i++) {
                leftSum += binCount[i];
                leftCount[i] = (float)leftSum;
                leftBox.grow_box(binBox[i]);
                leftArea[i] = leftBox.surface_area();
                rightSum += binCount[B - 1 - i];
                rightCount[B - 2 - i] = (float)rightSum;
                rightBox.grow_box(binBox[B - 1 - i]);
                // The reference stores the running right-hand area at [i] first
                // and then at the mirrored slot (src/vk_engine.cpp:1321-1322):
                // slots >= 10 end up holding the area of the wrong suffix.
                // Kept, because it decides the tree topology.
                rightArea[i] = rightBox.surface_area();
                rightArea[B - 2 - i] = rightBox.surface_area();
            }

            scale = (mx - mn) / (float)B;
            for (unsigned i = 0; i < B - 1; i++) {
                float cost = leftCount[i] * leftArea[i] + rightCount[i] * rightArea[i];
                if (cost < best) {
                    axis = a;
                    splitPos = mn + scale * (float)(i + 1);
                    best = cost;
                }
            }
        }
        return best;
    }

    void leaf_stats(uint32_t depth, uint32_t triCount) {
        if (depth > statMaxDepth) statMaxDepth = depth;
        if (depth < statMinDepth) statMinDepth = depth;
        if (triCount > statMaxTri) statMaxTri = triCount;
    }

    void subdivide(uint32_t idx, uint32_t depth) {
        // bvhNodes was sized up front, so indices stay valid across recursion
        if (bvhNodes[idx].triCount <= 2 || depth >= 64) {
            leaf_stats(depth, bvhNodes[idx].triCount);
            return;
        }
        int axis = 0;
        float splitPos = 0.f;
        float best = find_split(bvhNodes[idx], axis, splitPos);

        Box parent;
        parent.lo[0] = bvhNodes[idx].boundsX[0]; parent.hi[0] = bvhNodes[idx].boundsX[1];
        parent.lo[1] = bvhNodes[idx].boundsY[0]; parent.hi[1] = bvhNodes[idx].boundsY[1];
        parent.lo[2] = bvhNodes[idx].boundsZ[0]; parent.hi[2] = bvhNodes[idx].boundsZ[1];
        float noSplit = (float)bvhNodes[idx].triCount * parent.surface_area();
        if (best >= noSplit) {
            leaf_stats(depth, bvhNodes[idx].triCount);
            return;
        }

        int i = (int)bvhNodes[idx].index;
        int j = i + (int)bvhNodes[idx].triCount - 1;
        while (i <= j) {
            if (centroids[i].v[axis] < splitPos) {
                i++;
            } else {
                std::swap(triangles[i], triangles[j]);
                std::swap(centroids[i], centroids[j]);
                j--;
            }
        }
        uint32_t first = bvhNodes[idx].index;
        uint32_t leftCount = (uint32_t)i - first;
        if (leftCount == 0 || leftCount == bvhNodes[idx].triCount) {
            leaf_stats(depth, bvhNodes[idx].triCount);
            return;
        }

        uint32_t child = nodesUsed;
        nodesUsed += 2;
        bvhNodes[child].index = first;
        bvhNodes[child].triCount = leftCount;
        bvhNodes[child + 1].index = (uint32_t)i;
        bvhNodes[child + 1].triCount = bvhNodes[idx].triCount - leftCount;
        bvhNodes[idx].index = child;
        bvhNodes[idx].triCount = 0;
        update_bounds(child);
        update_bounds(child + 1);
        subdivide(child, depth + 1);
        subdivide(child + 1, depth + 1);
    }

    RtBvhBuildHook bvhHook = nullptr;
    void* bvhHookUser = nullptr;

    int build_bvh(uint32_t size, uint32_t triIndex) {
        if (size == 0) return fail("build_bvh: a mesh group with 0 triangles (undefined in the reference)");
        if (bvhHook) {  // e.g. the GPU builder (rt_bvh_hook): same nodes, numbering and triangle order
            const uint32_t offset = (uint32_t)bvhNodes.size();
            const uint32_t cap = size * 2u - 1u;
            bvhNodes.resize((size_t)offset + cap, BVHNode{});
            uint32_t used = 0, st[3] = {0, 0, 0};
            const int rc = bvhHook(bvhHookUser, triPoints.data(), (uint32_t)triPoints.size(), triangles.data() + triIndex, &centroids[triIndex].v[0],
                                   size, triIndex, offset, bvhNodes.data() + offset, cap, &used, st);
            if (rc != 0 || used == 0 || used > cap) { bvhNodes.resize(offset); return fail("build_bvh: the BVH hook failed"); }
            bvhNodes.resize((size_t)offset + used);
            bvhNodes.shrink_to_fit();
            nodesUsed = offset + used;
            statNodeCount = used; statMaxDepth = st[0]; statMinDepth = st[1]; statMaxTri = st[2];
            return 0;
        }
        nodesUsed++;
        uint32_t offset = (uint32_t)bvhNodes.size();
        bvhNodes.resize(bvhNodes.size() + (size_t)size * 2 - 1, BVHNode{});
        bvhNodes[offset].index = triIndex;
        bvhNodes[offset].triCount = size;
        statMaxDepth = 0; statMinDepth = 0xffffffffu; statMaxTri = 0;
        update_bounds(offset);
        subdivide(offset, 0);
        bvhNodes.resize(nodesUsed);
        bvhNodes.shrink_to_fit();
        statNodeCount = nodesUsed - offset;
        return 0;
    }

    // --- objects -----------------------------------------------------------
    void push_object(const RtPlacement& p, uint32_t material, uint32_t smooth, uint32_t bvhIndex, uint32_t sampler) {
        RenderObject o{};
        Mat4 m = placement_matrix(p);
        memcpy(o.transformMatrix, m.m, sizeof(m.m));
        o.smoothShade = smooth;
        o.bvhIndex = bvhIndex;
        o.materialIndex = material;
        o.samplerIndex = sampler;
        objects.push_back(o);
        placements.push_back(p);
    }

    int flush_group(const std::string& cacheKey, const RtPlacement& p, int fallbackMaterial,
                    const std::string& currentMat, const std::string& mtlKeyPrefix, bool smooth,
                    uint32_t objectTriOffset, uint32_t sampler) {
        uint32_t mat;
        if (currentMat.empty()) {
            mat = (uint32_t)fallbackMaterial;
        } else {
            auto it = loadedMaterials.find(mtlKeyPrefix + "/" + currentMat);
            if (it == loadedMaterials.end())
                return fail("usemtl '" + currentMat + "' not found in " + mtlKeyPrefix);
            mat = (uint32_t)it->second;
        }
        uint32_t bvhIndex = (uint32_t)bvhNodes.size();
        push_object(p, mat, smooth ? 1u : 0u, bvhIndex, sampler);
        loadedObjects.emplace(cacheKey, (int)bvhIndex);
        return build_bvh((uint32_t)triangles.size() - objectTriOffset, objectTriOffset);
    }

    void push_triangle(const uint32_t pt[3], uint32_t frontOnly) {
        Triangle t{};
        t.v0 = pt[0]; t.v1 = pt[1]; t.v2 = pt[2];
        t.frontOnly = frontOnly;
        Vec3h c{{0.f, 0.f, 0.f}};
        for (int i = 0; i < 3; i++) {
            const float* p = triPoints[pt[i]].position;
            c.v[0] += p[0]; c.v[1] += p[1]; c.v[2] += p[2];
        }
        c.v[0] = c.v[0] / 3.f; c.v[1] = c.v[1] / 3.f; c.v[2] = c.v[2] / 3.f;
        triangles.push_back(t);
        centroids.push_back(c);
    }

    // --- OBJ ---------------------------------------------------------------
    int read_obj(const std::string& filePath, const RtPlacement& p, int material) {
        auto cached = loadedObjects.find(filePath);
        if (cached != loadedObjects.end()) {
            // instancing: only a RenderObject pointing at the cached BVH (:802-815)
            push_object(p, (uint32_t)material, 0u, (uint32_t)cached->second, 0u);
            return 0;
        }
        std::ifstream in(filePath);
        if (!in.is_open()) return 1;  // the reference returns silently (:834)

        uint32_t objectTriOffset = (uint32_t)triangles.size();
        bool smooth = false, includeUVs = false;
        std::string currentMat, materialFile, line;
        std::vector<Vec3h> positions, normals;
        std::vector<std::array<float, 2>> uvs;
        const std::string mtlDir = dir_of(filePath);
        long lineNo = 0;

        while (in) {
            line.clear();
            std::getline(in, line);
            lineNo++;
            auto bad = [&](const char* what) {
                return fail(filePath + ":" + std::to_string(lineNo) + ": " + what);
            };
            if (line.find("mtllib") != std::string::npos) {
                materialFile = line.size() >= 7 ? line.substr(7) : std::string();
                int rc = read_mtl(mtlDir + materialFile);
                if (rc < 0) return rc;
            }
            const std::string prefix = line.substr(0, line.find(' '));

            if (prefix == "v") {
                Vec3h pos;
                size_t idx = 2;
                for (int i = 0; i < 3; i++) {
                    size_t sp = line.find(' ', idx);
                    if (idx > line.size() || !parse_float(line.substr(idx, sp == std::string::npos ? sp : sp - idx), pos.v[i]))
                        return bad("bad 'v' line");
                    idx = sp == std::string::npos ? line.size() + 1 : sp + 1;
                }
                sceneBounds_grow(pos.v);
                positions.push_back(pos);
            } else if (prefix == "vt") {
                size_t s1 = line.find(' ', 2);
                size_t s2 = s1 == std::string::npos ? s1 : line.find(' ', s1 + 1);
                if (s1 == std::string::npos || s2 == std::string::npos) return bad("bad 'vt' line");
                std::array<float, 2> uv;
                if (!parse_float(line.substr(s1 + 1, s2 - s1 - 1), uv[0]) || !parse_float(line.substr(s2), uv[1]))
                    return bad("bad 'vt' line");
                uvs.push_back(uv);
            } else if (prefix == "vn") {
                Vec3h n;
                size_t idx = 3;
                for (int i = 0; i < 3; i++) {
                    size_t sp = line.find(' ', idx);
                    if (idx > line.size() || !parse_float(line.substr(idx, sp == std::string::npos ? sp : sp - idx), n.v[i]))
                        return bad("bad 'vn' line");
                    idx = sp == std::string::npos ? line.size() + 1 : sp + 1;
                }
                normals.push_back(n);
            } else if (prefix == "f") {
                // corner count = blanks on the line, the last character excluded,
                // so one trailing blank is tolerated (:881-884)
                int corners = 0;
                for (size_t i = 0; i + 1 < line.size(); i++)
                    if (line[i] == ' ') corners++;
                if (corners < 3 || corners > 4) return bad("faces need 3 corners (4 tolerated, 4th ignored)");

                std::vector<int> vi, ti, ni;
                size_t from = 0;
                for (int c = 0; c < corners; c++) {
                    size_t sp = line.find(' ', from);
                    size_t nx = line.find(' ', sp + 1);
                    std::string tok = (nx == std::string::npos)
                        ? line.substr(sp + 1)
                        : line.substr(sp + 1, nx - sp - (c == corners - 1 ? 0 : 1));
                    size_t s1 = tok.find('/');
                    size_t s2 = (s1 == std::string::npos) ? std::string::npos : tok.find('/', s1 + 1);
                    int v;
                    std::string vs = tok.substr(0, s1);
                    if (!vs.empty()) {
                        if (!parse_int(vs, v)) return bad("bad vertex index");
                        vi.push_back(v - 1);
                    }
                    // reference arithmetic on find() results: with no '/', the
                    // "uv" field is the whole token (npos + 1 wraps to 0)
                    std::string ts;
                    if (s1 == std::string::npos) ts = tok;
                    else if (s2 == std::string::npos) ts = tok.substr(s1 + 1);
                    else ts = tok.substr(s1 + 1, s2 - s1 - 1);
                    if (!ts.empty()) {
                        if (!parse_int(ts, v)) return bad("bad uv index");
                        ti.push_back(v - 1);
                        includeUVs = true;
                    }
                    // without any 'vn' so far the reference skips the cursor
                    // advance too, so every corner re-reads the first token (:906)
                    if (normals.empty()) continue;
                    std::string ns = (s2 == std::string::npos) ? tok : tok.substr(s2 + 1);
                    if (!ns.empty()) {
                        if (!parse_int(ns, v)) return bad("bad normal index");
                        ni.push_back(v - 1);
                    }
                    from = nx;
                }

                uint32_t pt[4] = {0, 0, 0, 0};
                for (int c = 0; c < corners; c++) {
                    TrianglePoint tp{};
                    float nrm[3] = {0.f, 0.f, 0.f};
                    if (!normals.empty()) {
                        if ((size_t)c >= ni.size() || ni[c] < 0 || (size_t)ni[c] >= normals.size()) return bad("normal index out of range");
                        memcpy(nrm, normals[ni[c]].v, 12);
                    }
                    float uv[2] = {0.f, 0.f};
                    if (includeUVs) {
                        if ((size_t)c >= ti.size() || ti[c] < 0 || (size_t)ti[c] >= uvs.size()) return bad("uv index out of range");
                        uv[0] = uvs[ti[c]][0]; uv[1] = uvs[ti[c]][1];
                    }
                    if ((size_t)c >= vi.size() || vi[c] < 0 || (size_t)vi[c] >= positions.size()) return bad("vertex index out of range");
                    memcpy(tp.position, positions[vi[c]].v, 12); tp.position[3] = uv[0];
                    memcpy(tp.normal, nrm, 12);                  tp.normal[3] = uv[1];
                    pt[c] = (uint32_t)triPoints.size();
                    triPoints.push_back(tp);  // one unshared point per corner (:929-934)
                }
                push_triangle(pt, p.frontOnly ? 1u : 0u);
            } else if (prefix == "usemtl") {
                size_t sp = line.find(' ');
                std::string mat = line.substr(sp + 1);
                if (currentMat.empty()) {  // first usemtl only names the group (:963-966)
                    currentMat = mat;
                    continue;
                }
                int rc = flush_group(filePath + "/" + currentMat, p, material, currentMat, mtlDir + materialFile,
                                     smooth, objectTriOffset, p.samplerIndex);
                if (rc < 0) return rc;
                currentMat = mat;
                objectTriOffset = (uint32_t)triangles.size();
                smooth = false;
            } else if (prefix == "s") {
                smooth = line.size() > 2 && (line[2] - '0') == 1;
            }
        }
        // last (or only) group; note the reference leaves samplerIndex at its
        // default 0 here (:1009-1019)
        return flush_group(filePath, p, material, currentMat, mtlDir + materialFile, smooth, objectTriOffset, 0u);
    }

    // --- MTL ---------------------------------------------------------------
    int read_mtl(const std::string& filePath) {
        std::ifstream in(filePath);
        if (!in.is_open()) return 1;  // "Could not open material file" and carry on (:1064-1067)
        std::string line, name;
        RayMaterial cur;
        rt_material_default(&cur);
        while (in) {
            line.clear();
            std::getline(in, line);
            if (line.find("newmtl") != std::string::npos) {
                if (!name.empty()) {
                    loadedMaterials.emplace(filePath + "/" + name, (int)rayMaterials.size());
                    rayMaterials.push_back(cur);
                    rt_material_default(&cur);
                }
                name = line.size() >= 7 ? line.substr(7) : std::string();
                continue;
            }
            std::string l;
            for (char ch : line) if (ch != '\t') l.push_back(ch);
            std::string prefix = l.substr(0, l.find(' '));
            if (prefix == "Ka" || prefix == "Kd") {
                size_t s1 = l.find(' ');
                size_t s2 = s1 == std::string::npos ? s1 : l.find(' ', s1 + 1);
                size_t s3 = s2 == std::string::npos ? s2 : l.find(' ', s2 + 1);
                float c[3];
                if (s3 == std::string::npos || !parse_float(l.substr(s1 + 1, s2 - s1 - 1), c[0]) ||
                    !parse_float(l.substr(s2 + 1, s3 - s2 - 1), c[1]) || !parse_float(l.substr(s3 + 1), c[2]))
                    return fail(filePath + ": bad " + prefix + " line");
                for (int i = 0; i < 3; i++) cur.albedo[i] *= c[i];  // Ka and Kd both multiply (:1090-1100)
            } else if (prefix == "map_Ka" || prefix == "map_Kd" || prefix == "map_Ks" || prefix == "map_d" || prefix == "map_bump") {
                // path = directory of the MTL file + the rest of the line (:1110-1113); the slot is claimed in file order
                // (map_bump is case-sensitive: Blender's map_Bump is skipped, as in the reference)
                if (texturesUsed >= (uint32_t)RT_MAX_TEXTURES) return fail(filePath + ": more than RT_MAX_TEXTURES texture slots");
                const size_t sp = l.find(' ');
                const std::string value = sp == std::string::npos ? std::string() : l.substr(sp + 1);
                texturePaths.push_back(filePath.substr(0, filePath.rfind('/') + 1) + value);
                int& slot = (prefix == "map_Ks") ? cur.metalnessIndex : (prefix == "map_d") ? cur.alphaIndex : (prefix == "map_bump") ? cur.bumpIndex : cur.albedoIndex;
                slot = (int)texturesUsed++;
            }
            // Ni, d are parsed and discarded; Ks, Ke, Ns, illum ignored (:1101-1108)
        }
        loadedMaterials.emplace(filePath + "/" + name, (int)rayMaterials.size());
        rayMaterials.push_back(cur);
        return 0;
    }

    void sceneBounds_grow(const float* p) {
        for (int i = 0; i < 3; i++) {
            sceneLo[i] = p[i] < sceneLo[i] ? p[i] : sceneLo[i];
            sceneHi[i] = sceneHi[i] < p[i] ? p[i] : sceneHi[i];
        }
    }
};

// ---------------------------------------------------------------- C ABI
extern "C" {

int rt_scene_create(rt_scene** out) {
    if (!out) return -1;
    *out = new rt_scene();
    return 0;
}
void rt_scene_destroy(rt_scene* s) { delete s; }
const char* rt_scene_last_error(const rt_scene* s) { return s ? s->error.c_str() : "null scene"; }

void rt_material_default(RayMaterial* m) {
    memset(m, 0, sizeof(*m));
    m->albedo[0] = m->albedo[1] = m->albedo[2] = 1.f;
    m->ior = -1.f;
    m->albedoIndex = m->metalnessIndex = m->alphaIndex = m->bumpIndex = -1;
}
void rt_placement_default(RtPlacement* p) {
    memset(p, 0, sizeof(*p));
    p->scale[0] = p->scale[1] = p->scale[2] = 1.f;
}

int rt_scene_add_material(rt_scene* s, const RayMaterial* m) {
    if (!s || !m) return -1;
    s->rayMaterials.push_back(*m);
    return (int)s->rayMaterials.size() - 1;
}

int rt_scene_set_sphere(rt_scene* s, uint32_t i, const float position[3], float radius, uint32_t materialIndex) {
    if (!s) return -1;
    if (i >= RT_MAX_SPHERES) return s->fail("sphere index >= MAX_SPHERES");
    if (s->spheres.size() < RT_MAX_SPHERES) s->spheres.resize(RT_MAX_SPHERES, Sphere{});
    Sphere& sp = s->spheres[i];
    memcpy(sp.position, position, 12);
    sp.radius = radius;
    sp.materialIndex = materialIndex;
    return 0;
}

int rt_scene_read_obj(rt_scene* s, const char* filePath, const RtPlacement* placement, int material) {
    if (!s || !filePath || !placement) return -1;
    return s->read_obj(filePath, *placement, material);
}
int rt_scene_read_mtl(rt_scene* s, const char* filePath) {
    if (!s || !filePath) return -1;
    return s->read_mtl(filePath);
}

int rt_scene_add_mesh(rt_scene* s, const char* key, const float* positions, const float* normals,
                      const float* uvs, uint32_t triCount, const RtPlacement* placement, int material) {
    if (!s || !key || !positions || !normals || !placement) return -1;
    std::string k(key);
    auto cached = s->loadedObjects.find(k);
    if (cached != s->loadedObjects.end()) {
        s->push_object(*placement, (uint32_t)material, 0u, (uint32_t)cached->second, 0u);
        return 0;
    }
    uint32_t triOffset = (uint32_t)s->triangles.size();
    for (uint32_t t = 0; t < triCount; t++) {
        uint32_t pt[3];
        for (int c = 0; c < 3; c++) {
            TrianglePoint tp{};
            memcpy(tp.position, positions + (size_t)t * 9 + c * 3, 12);
            memcpy(tp.normal, normals + (size_t)t * 9 + c * 3, 12);
            if (uvs) { tp.position[3] = uvs[(size_t)t * 6 + c * 2]; tp.normal[3] = uvs[(size_t)t * 6 + c * 2 + 1]; }
            s->sceneBounds_grow(tp.position);
            pt[c] = (uint32_t)s->triPoints.size();
            s->triPoints.push_back(tp);
        }
        s->push_triangle(pt, placement->frontOnly ? 1u : 0u);
    }
    return s->flush_group(k, *placement, material, std::string(), std::string(), false, triOffset, 0u);
}

int rt_scene_cornell_box(rt_scene* s, const char* assetDir) {
    if (!s || !assetDir) return -1;
    std::string d(assetDir);
    if (!d.empty() && d.back() != '/') d.push_back('/');
    auto load = [&](const char* file, const RtPlacement& p, int mat) {
        int rc = s->read_obj(d + file, p, mat);
        if (rc == 1) return s->fail("cannot open " + d + file);
        return rc;
    };
    RtPlacement light; rt_placement_default(&light);
    light.frontOnly = 1; light.position[1] = -1.5f;
    if (int rc = load("light2.obj", light, 3)) return rc;

    RtPlacement plane; rt_placement_default(&plane);
    plane.frontOnly = 1;
    plane.position[1] = 0.5f;                                           // bottom
    if (int rc = load("plane.obj", plane, 0)) return rc;
    plane.position[0] = -1.f; plane.position[1] = -0.5f; plane.position[2] = 0.f;
    plane.rotation[0] = 90.f; plane.rotation[1] = 0.f; plane.rotation[2] = 90.f;   // left, green
    if (int rc = load("plane.obj", plane, 2)) return rc;
    plane.position[0] = 1.f;
    plane.rotation[2] = -90.f;                                          // right, red
    if (int rc = load("plane.obj", plane, 1)) return rc;
    plane.position[0] = 0.f; plane.position[1] = -1.5f;
    plane.rotation[0] = plane.rotation[1] = plane.rotation[2] = 0.f;    // top (with the light's hole)
    if (int rc = load("ceiling.obj", plane, 0)) return rc;
    plane.position[1] = -0.5f; plane.position[2] = 1.f;
    plane.rotation[0] = 90.f;                                           // back
    if (int rc = load("plane.obj", plane, 0)) return rc;
    plane.position[2] = -1.f;
    plane.rotation[0] = -90.f;                                          // front
    if (int rc = load("plane.obj", plane, 0)) return rc;
    return 0;
}

int rt_scene_prepare_default(rt_scene* s, const char* assetDir) {
    if (!s || !assetDir) return -1;
    std::string d(assetDir);
    if (!d.empty() && d.back() != '/') d.push_back('/');
    s->spheres.assign(RT_MAX_SPHERES, Sphere{});

    RayMaterial white, red, green, li, mirror, dielectric;
    rt_material_default(&white);
    rt_material_default(&red);   red.albedo[1] = red.albedo[2] = 0.f;
    rt_material_default(&green); green.albedo[0] = green.albedo[2] = 0.f;
    rt_material_default(&li);
    li.albedo[0] = li.albedo[1] = li.albedo[2] = 0.f;
    li.emissionColor[0] = li.emissionColor[1] = li.emissionColor[2] = 1.f;
    li.emissionStrength = 2.4f;
    rt_material_default(&mirror); mirror.reflectance = 1.f;
    rt_material_default(&dielectric); dielectric.ior = 2.f;
    for (const RayMaterial* m : {&white, &red, &green, &li, &mirror, &dielectric}) s->rayMaterials.push_back(*m);

    RtPlacement cube; rt_placement_default(&cube);
    cube.samplerIndex = 1;
    cube.scale[0] = cube.scale[1] = cube.scale[2] = 0.25f;
    cube.rotation[1] = -30.f;
    cube.position[0] = -0.4f; cube.position[1] = 0.25f; cube.position[2] = -0.45f;
    int rc = s->read_obj(d + "cube.obj", cube, 0);
    if (rc == 1) return s->fail("cannot open " + d + "cube.obj");
    if (rc) return rc;
    cube.scale[0] = 0.3f; cube.scale[1] = 0.7f; cube.scale[2] = 0.3f;
    cube.rotation[1] = 30.f;
    cube.position[0] = 0.4f; cube.position[1] = -0.2f; cube.position[2] = 0.45f;
    if ((rc = s->read_obj(d + "cube.obj", cube, 0))) return rc;
    return rt_scene_cornell_box(s, assetDir);
}

int rt_scene_get_arrays(const rt_scene* s, RtSceneArrays* out) {
    if (!s || !out) return -1;
    out->spheres = s->spheres.data();        out->sphereCount = (uint32_t)s->spheres.size();
    out->materials = s->rayMaterials.data(); out->materialCount = (uint32_t)s->rayMaterials.size();
    out->triPoints = s->triPoints.data();    out->triPointCount = (uint32_t)s->triPoints.size();
    out->triangles = s->triangles.data();    out->triangleCount = (uint32_t)s->triangles.size();
    out->objects = s->objects.data();        out->objectCount = (uint32_t)s->objects.size();
    out->bvhNodes = s->bvhNodes.data();      out->bvhNodeCount = (uint32_t)s->bvhNodes.size();
    return 0;
}

uint32_t rt_scene_texture_count(const rt_scene* s) { return s ? (uint32_t)s->texturePaths.size() : 0u; }

const char* rt_scene_texture_path(const rt_scene* s, uint32_t i) {
    return (s && i < s->texturePaths.size()) ? s->texturePaths[i].c_str() : "";
}

int rt_scene_add_texture(rt_scene* s, const char* path) {
    if (!s || !path) return -1;
    if (s->texturesUsed >= (uint32_t)RT_MAX_TEXTURES) return s->fail("more than RT_MAX_TEXTURES texture slots");
    s->texturePaths.push_back(path);
    return (int)s->texturesUsed++;
}

int rt_scene_set_material(rt_scene* s, uint32_t i, const RayMaterial* m) {
    if (!s || !m) return -1;
    if (i >= s->rayMaterials.size()) return s->fail("material index out of range");
    s->rayMaterials[i] = *m;
    return 0;
}

int rt_scene_find_material(const rt_scene* s, const char* key) {
    if (!s || !key) return -1;
    auto it = s->loadedMaterials.find(key);
    return it == s->loadedMaterials.end() ? -1 : it->second;
}

int rt_scene_set_bvh_hook(rt_scene* s, RtBvhBuildHook hook, void* user) {
    if (!s) return -1;
    s->bvhHook = hook;
    s->bvhHookUser = user;
    return 0;
}

int rt_scene_last_bvh_stats(const rt_scene* s, uint32_t* nodeCount, uint32_t* maxDepth, uint32_t* minDepth, uint32_t* maxTri) {
    if (!s) return -1;
    if (nodeCount) *nodeCount = s->statNodeCount;
    if (maxDepth) *maxDepth = s->statMaxDepth;
    if (minDepth) *minDepth = s->statMinDepth;
    if (maxTri) *maxTri = s->statMaxTri;
    return 0;
}

void rt_transform_matrix(const RtPlacement* p, float outMat4[16]) {
    Mat4 m = placement_matrix(*p);
    memcpy(outMat4, m.m, sizeof(m.m));
}

// rotY * rotX * rotZ with the column constructors of src/vk_engine.cpp:1636-1653
void rt_camera_rotation(const float anglesDeg[3], float out[16]) {
    float sx, cx, sy, cy, sz, cz;
    rt_sincos(rt_radians(anglesDeg[0]), &sx, &cx);
    rt_sincos(rt_radians(anglesDeg[1]), &sy, &cy);
    rt_sincos(rt_radians(anglesDeg[2]), &sz, &cz);
    // column-major 3x3: m[c*3+r]
    const float rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    const float ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    const float rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    auto mul3 = [](const float* a, const float* b, float* r) {
        for (int c = 0; c < 3; c++)
            for (int row = 0; row < 3; row++) {
                float s = a[0 + row] * b[c * 3 + 0] + a[3 + row] * b[c * 3 + 1];
                r[c * 3 + row] = s + a[6 + row] * b[c * 3 + 2];
            }
    };
    float yx[9], yxz[9];
    mul3(ry, rx, yx);
    mul3(yx, rz, yxz);
    memset(out, 0, 64);
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) out[c * 4 + r] = yxz[c * 3 + r];
    out[15] = 1.f;
}

void rt_push_constants_default(PushConstants* pc, uint32_t width, uint32_t height) {
    memset(pc, 0, sizeof(*pc));
    const float angles[3] = {4.f, 0.f, 0.f};  // src/vk_engine.h:325
    rt_camera_rotation(angles, pc->camInfo.cameraRotation);
    pc->camInfo.pos[0] = 0.f; pc->camInfo.pos[1] = -0.5f; pc->camInfo.pos[2] = -3.5f;
    pc->camInfo.nearPlane = 0.1f;
    pc->camInfo.aspectRatio = (float)width / (float)height;
    pc->camInfo.fov = 50.f;
    EnvironmentData& e = pc->environment;
    e.horizonColor[0] = 0.986f; e.horizonColor[1] = 1.f; e.horizonColor[2] = 0.902f; e.horizonColor[3] = 1000.f;
    e.zenithColor[0] = 0.265f; e.zenithColor[1] = 0.595f; e.zenithColor[2] = 0.887f; e.zenithColor[3] = 10.f;
    e.groundColor[0] = e.groundColor[1] = e.groundColor[2] = 0.431f;
    rt_vec3 l = rt_normalize(rt_v3(2.f, 0.8f, -3.f));
    e.lightDir[0] = l.x; e.lightDir[1] = l.y; e.lightDir[2] = l.z; e.lightDir[3] = 0.f;
    RayTracerData& t = pc->rayTraceParams;
    t.progressive = 0; t.singleRender = 0; t.debug = -1;
    t.raysPerPixel = 1; t.bounceLimit = 8;
    t.sphereCount = RT_MAX_SPHERES; t.objectCount = 0;
    t.triangleCap = 50; t.boxCap = 200; t.sampleLimit = 10;
    pc->frameCount = 0;
}

uint32_t rt_host_selftest(void) {
    volatile float in[7] = {1.0001220703125f, 0.9998779296875f, -1.f, 3.f, 1e-30f, 1e-10f, 2.f};
    return rt_selftest_bits(in);
}

const char* rt_version(void) { return "ray_tracer_amd 0.1 (gfx950)"; }

}  // extern "C"
