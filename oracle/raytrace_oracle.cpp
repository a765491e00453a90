// raytrace_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar C++ restatement of the reference's per-pixel megakernel
// shaders/raytrace.comp, function by function, over the reference's own AoS
// buffers (the six SSBOs of raytrace.comp:124-146 and the 208-byte push
// constants). It is the checker for the HIP wavefront pipeline and the
// "port" CPU baseline of bench.py. Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load it; the product never does.
//
// PARITY STATUS: bit-level "parity unpinned": the reference has no tests or
// golden vectors for this path and its shader cannot be
// compiled here (no GLSL compiler, no Vulkan). Image-level it IS pinned, to
// the one output of the reference whose parameters are known — the screenshot
// renders/importance_sampling/0_1-NEE2.png (default Cornell scene, single
// render, 100 samples, bounce limit 5, all readable in its ImGui panel):
// tests/test_reference_render.py recovers the 1728x1117 render from it and
// finds the same silhouette and light edges to the pixel, the same radiometry
// per region and channel within 1.5 %, the same per-pixel noise level, and
// noise that follows this code's frameCount-0 RNG streams and no other seed.
// Besides that: (1) the RNG known-answer values derived from the formula of
// raytrace.comp:158-163, (2) the std140 layout sizes (A14),
// (3) line-by-line review against the shader (citations below), and
// (4) scene invariants of the default Cornell box (51 triangles, 153 points,
// 9 objects). GLSL built-ins come from include/rt_det_math.h (GLSL 4.50 spec
// formulas); their precision is implementation-defined in the reference.
//
// Build: g++ -O2 -ffp-contract=off (see oracle/Makefile). Threads split the
// rows for timing only; each pixel is computed by one thread, start to end.

#include "rt_amd.h"
#include "rt_det_math.h"
#include "rt_probe.h"
#include "raytrace_oracle.h"

#include <atomic>
#include <thread>
#include <vector>

namespace {

struct Ray { rt_vec3 origin, dir; };

// raytrace.comp:72-82
struct HitInfo {
    rt_vec3 hitPoint{0, 0, 0};
    rt_vec3 normal{0, 0, 0};
    float dst = 0.f;
    float uv[2] = {0.f, 0.f};    // raytrace.comp:80; only triangle hits set it (:249-256)
    uint32_t objectHitIndex = 0;
    uint32_t triHitIndex = 0;
    uint32_t materialIndex = 0;  // defined as 0 where the shader leaves it unset
    bool didHit = false;
    bool frontFace = false;
    bool isSphere = false;
};

// raytrace.comp:112-118
struct BxDFResult {
    rt_vec3 sampledDir, radiance, directLight;
    float originSign, cosineMisWeight;
};

struct Tally {  // one calculateIntersections call
    uint32_t box = 0, tri = 0;
};

struct Totals {
    uint64_t boxRef = 0, triRef = 0, raysRef = 0, hitsRef = 0;
    uint64_t boxUnique = 0, triUnique = 0, raysUnique = 0, hitsUnique = 0;
    uint64_t paths = 0, segments = 0;
    uint64_t stackOverflow = 0;
    uint64_t emitterTests = 0, lightQueryMismatch = 0;
};

struct Scene {
    RtSceneArrays a;
    uint32_t sphereCount, objectCount;
    std::vector<float> inv;  // inverse(object.transformMatrix), 16 floats per object
    // The wavefront pipeline's light queries (its definition of "executed work", see executed_light_query below):
    // every triangle of every object with an emissive material, the emissive spheres, and whether the shortcut applies.
    std::vector<std::pair<uint32_t, uint32_t>> emitTris;  // {object, triangle}
    uint32_t emitSphereMask = 0;
    bool emitMode = false;
};

// Textures. The snapshot's shader never samples one; the semantics are declared ones (include/rt_amd.h,
// rt_upload_textures). Parity unpinned. The table of the scene the next oracle_render calls use:
struct OracleTexture { uint32_t width, height; std::vector<uint8_t> rgba; };
std::vector<OracleTexture> g_textures;

// the map bound to a material slot (-1 or beyond the uploaded table: none)
const OracleTexture* mapOf(int32_t index) {
    return (index >= 0 && (size_t)index < g_textures.size()) ? &g_textures[(size_t)index] : nullptr;
}
// red byte of the texel at hit.uv (+ dx texels along the row, + dy along the column, under the object's sampler)
uint32_t mapRed8(const OracleTexture& t, const float uv[2], bool clampEdge, bool dx = false, bool dy = false) {
    uint32_t x = rt_tex_index(uv[0], t.width, clampEdge), y = rt_tex_index(1.f - uv[1], t.height, clampEdge);
    if (dx) x = rt_tex_next(x, t.width, clampEdge);
    if (dy) y = rt_tex_next(y, t.height, clampEdge);
    return t.rgba[((size_t)y * t.width + x) * 4];
}

// raytrace.comp:195-224
HitInfo sphereIntersection(const Sphere& sphere, const Ray& ray) {
    HitInfo h;
    h.isSphere = true;
    rt_vec3 center = rt_v3(sphere.position[0], sphere.position[1], sphere.position[2]);
    rt_vec3 oc = rt_sub(center, ray.origin);
    float a = rt_dot(ray.dir, ray.dir);
    float b = rt_dot(oc, ray.dir);
    float c = rt_dot(oc, oc) - sphere.radius * sphere.radius;
    float discriminant = b * b - a * c;
    if (discriminant >= 0.f) {
        float sqrtd = rt_sqrt(discriminant);
        float dst = (b - sqrtd) / a;
        h.frontFace = true;
        if (dst < 0.f) {
            dst = (b + sqrtd) / a;
            h.frontFace = false;
            if (dst < 0.f) return h;
        }
        h.didHit = true;
        h.dst = dst;
        h.hitPoint = rt_add(ray.origin, rt_scale(ray.dir, dst));
        h.normal = rt_scale(rt_normalize(rt_sub(h.hitPoint, center)), h.frontFace ? 1.f : -1.f);
        h.materialIndex = sphere.materialIndex;
    }
    return h;
}

// raytrace.comp:227-261
HitInfo triangleIntersection(const Ray& ray, const TrianglePoint& p0, const TrianglePoint& p1,
                             const TrianglePoint& p2, bool frontOnly) {
    rt_vec3 v0 = rt_v3(p0.position[0], p0.position[1], p0.position[2]);
    rt_vec3 v1 = rt_v3(p1.position[0], p1.position[1], p1.position[2]);
    rt_vec3 v2 = rt_v3(p2.position[0], p2.position[1], p2.position[2]);
    rt_vec3 v1v0 = rt_sub(v1, v0);
    rt_vec3 v2v0 = rt_sub(v2, v0);
    rt_vec3 rov0 = rt_sub(ray.origin, v0);
    rt_vec3 n = rt_cross(v1v0, v2v0);
    rt_vec3 q = rt_cross(rov0, ray.dir);
    float d0 = -rt_dot(ray.dir, n);
    float d = 1.f / d0;
    float dst = rt_dot(rov0, n) * d;
    float u = rt_dot(v2v0, q) * d;
    float v = -rt_dot(v1v0, q) * d;
    float w = 1.f - u - v;

    HitInfo hit;
    hit.frontFace = d0 >= 0.00000001f;
    hit.didHit = dst >= 0.f && u >= 0.f && v >= 0.f && w >= 0.f && !(!hit.frontFace && frontOnly);
    hit.hitPoint = rt_add(ray.origin, rt_scale(ray.dir, dst));
    hit.dst = dst;
    rt_vec3 n0 = rt_v3(p0.normal[0], p0.normal[1], p0.normal[2]);
    rt_vec3 n1 = rt_v3(p1.normal[0], p1.normal[1], p1.normal[2]);
    rt_vec3 n2 = rt_v3(p2.normal[0], p2.normal[1], p2.normal[2]);
    rt_vec3 ni = rt_add(rt_add(rt_scale(n0, w), rt_scale(n1, u)), rt_scale(n2, v));
    hit.normal = rt_scale(ni, hit.frontFace ? 1.f : -1.f);
    // hit.uv = w * v0uv + u * v1uv + v * v2uv; (0.5, 0.5) when two corners share their uv (:249-256)
    const float u0 = p0.position[3], v0u = p0.normal[3], u1 = p1.position[3], v1u = p1.normal[3], u2 = p2.position[3], v2u = p2.normal[3];
    hit.uv[0] = (w * u0 + u * u1) + v * u2;
    hit.uv[1] = (w * v0u + u * v1u) + v * v2u;
    if ((u0 == u1 && v0u == v1u) || (u1 == u2 && v1u == v2u) || (u2 == u0 && v2u == v0u)) { hit.uv[0] = 0.5f; hit.uv[1] = 0.5f; }
    return hit;
}

// raytrace.comp:263-274
float boxIntersection(const BVHNode& n, const rt_vec3& origin, const rt_vec3& invDir) {
    rt_vec3 tMin = rt_mul(rt_sub(rt_v3(n.boundsX[0], n.boundsY[0], n.boundsZ[0]), origin), invDir);
    rt_vec3 tMax = rt_mul(rt_sub(rt_v3(n.boundsX[1], n.boundsY[1], n.boundsZ[1]), origin), invDir);
    rt_vec3 t1 = rt_v3(rt_min(tMin.x, tMax.x), rt_min(tMin.y, tMax.y), rt_min(tMin.z, tMax.z));
    rt_vec3 t2 = rt_v3(rt_max(tMin.x, tMax.x), rt_max(tMin.y, tMax.y), rt_max(tMin.z, tMax.z));
    float tNear = rt_max(rt_max(t1.x, t1.y), t1.z);
    float tFar = rt_min(rt_min(t2.x, t2.y), t2.z);
    bool hit = tFar >= tNear && tFar > 0.f;
    return hit ? (tNear > 0.f ? tNear : 0.f) : RT_MISS_DST;
}

// raytrace.comp:276-353
// earlyT > 0 is not part of the shader: it is how the wavefront pipeline executes a light query (see
// executed_light_query) and only ever feeds the "executed work" counters, never a pixel.
HitInfo calculateIntersections(const Scene& sc, const Ray& ray, Tally& stats, uint64_t* overflow, float earlyT = 0.f) {
    HitInfo closestHit;
    closestHit.didHit = false;
    closestHit.dst = RT_MISS_DST;

    for (uint32_t i = 0; i < sc.sphereCount; i++) {
        HitInfo h = sphereIntersection(sc.a.spheres[i], ray);
        if (h.didHit && h.dst < closestHit.dst) {
            closestHit = h;
            closestHit.objectHitIndex = i;
        }
    }

    for (uint32_t i = 0; i < sc.objectCount; i++) {
        const RenderObject& object = sc.a.objects[i];
        const float* inv = &sc.inv[(size_t)i * 16];
        Ray tr;
        tr.dir = rt_xform_dir(inv, ray.dir);       // not renormalised: dst stays in world units
        tr.origin = rt_xform_point(inv, ray.origin);
        rt_vec3 invDir = rt_v3(1.f / tr.dir.x, 1.f / tr.dir.y, 1.f / tr.dir.z);
        const OracleTexture* alphaMap = g_textures.empty() ? nullptr : mapOf(sc.a.materials[object.materialIndex].alphaIndex);
        const OracleTexture* bumpMap = g_textures.empty() ? nullptr : mapOf(sc.a.materials[object.materialIndex].bumpIndex);
        const bool clampEdge = object.samplerIndex == 1u;

        uint32_t stack[128];
        uint32_t sp = 1;
        stack[0] = object.bvhIndex;
        while (sp > 0) {
            const BVHNode& node = sc.a.bvhNodes[stack[--sp]];
            if (node.triCount != 0) {
                stats.tri += node.triCount;
                for (uint32_t j = node.index; j < node.index + node.triCount; j++) {
                    const Triangle& tri = sc.a.triangles[j];
                    HitInfo h = triangleIntersection(tr, sc.a.triPoints[tri.v0], sc.a.triPoints[tri.v1],
                                                     sc.a.triPoints[tri.v2], tri.frontOnly != 0);
                    h.materialIndex = object.materialIndex;
                    // alpha map (declared, rt_det_math.h): the hit is cut out where the map's texel decodes below 0.5
                    if (alphaMap && h.didHit && h.dst < closestHit.dst && mapRed8(*alphaMap, h.uv, clampEdge) < RT_ALPHA_CUT_BYTE) h.didHit = false;
                    if (h.didHit && h.dst < closestHit.dst) {
                        closestHit = h;
                        if (bumpMap) {  // bump map (declared, rt_bump_normal): the object-space normal tilted by the height steps
                            const TrianglePoint &p0 = sc.a.triPoints[tri.v0], &p1 = sc.a.triPoints[tri.v1], &p2 = sc.a.triPoints[tri.v2];
                            const float sgn = h.frontFace ? 1.f : -1.f;
                            const float h0 = rt_srgb8_to_linear(mapRed8(*bumpMap, h.uv, clampEdge));
                            const float hx = rt_srgb8_to_linear(mapRed8(*bumpMap, h.uv, clampEdge, true, false)) - h0;
                            const float hy = rt_srgb8_to_linear(mapRed8(*bumpMap, h.uv, clampEdge, false, true)) - h0;
                            const rt_vec3 q0 = rt_v3(p0.position[0], p0.position[1], p0.position[2]);
                            const rt_vec3 e1 = rt_sub(rt_v3(p1.position[0], p1.position[1], p1.position[2]), q0);
                            const rt_vec3 e2 = rt_sub(rt_v3(p2.position[0], p2.position[1], p2.position[2]), q0);
                            const rt_vec3 nb = rt_bump_normal(rt_scale(h.normal, sgn), e1, e2, p1.position[3] - p0.position[3], p1.normal[3] - p0.normal[3],
                                                              p2.position[3] - p0.position[3], p2.normal[3] - p0.normal[3], hx, hy);
                            closestHit.normal = rt_scale(nb, sgn);
                        }
                        // forward matrix on the normal, not the inverse transpose (:318)
                        closestHit.normal = rt_normalize(rt_xform_dir(object.transformMatrix, closestHit.normal));
                        closestHit.hitPoint = rt_xform_point(object.transformMatrix, closestHit.hitPoint);
                        closestHit.triHitIndex = j;
                        closestHit.objectHitIndex = i;
                    }
                }
                if (closestHit.didHit && closestHit.dst < earlyT) {  // answered: nearer than the nearest emissive primitive
                    HitInfo none;
                    none.dst = RT_MISS_DST;
                    return none;
                }
            } else {
                const BVHNode& c1 = sc.a.bvhNodes[node.index];
                const BVHNode& c2 = sc.a.bvhNodes[node.index + 1];
                float dst1 = boxIntersection(c1, tr.origin, invDir);
                float dst2 = boxIntersection(c2, tr.origin, invDir);
                stats.box += 2;
                bool nearA = dst1 <= dst2;
                float dstNear = nearA ? dst1 : dst2;
                float dstFar = nearA ? dst2 : dst1;
                uint32_t idxNear = nearA ? node.index : node.index + 1;
                uint32_t idxFar = nearA ? node.index + 1 : node.index;
                if (dstFar < closestHit.dst) stack[sp++] = idxFar;
                if (dstNear < closestHit.dst) stack[sp++] = idxNear;
                if (sp > 64 && overflow) (*overflow)++;  // the shader's stack is uint[64] (:302)
                if (sp > 126) sp = 126;
            }
        }
    }
    return closestHit;
}

// raytrace.comp:177-181
float schlick(float cosine, float ri) {
    float r0 = (1.f - ri) / (1.f + ri);
    r0 = r0 * r0;
    return r0 + (1.f - r0) * rt_pow(1.f - cosine, 5.f);
}

// raytrace.comp:356-365
rt_vec3 getEnvironmentLight(const PushConstants& pc, const Ray& ray) {
    const EnvironmentData& env = pc.environment;
    if (!(env.lightDir[3] == 1.f)) return rt_v3(0.f, 0.f, 0.f);
    float skyT = rt_pow(rt_smoothstep(0.f, 0.4f, -ray.dir.y), 0.35f);
    rt_vec3 hor = rt_v3(env.horizonColor[0], env.horizonColor[1], env.horizonColor[2]);
    rt_vec3 zen = rt_v3(env.zenithColor[0], env.zenithColor[1], env.zenithColor[2]);
    rt_vec3 sky = rt_v3(rt_mix(hor.x, zen.x, skyT), rt_mix(hor.y, zen.y, skyT), rt_mix(hor.z, zen.z, skyT));
    rt_vec3 negL = rt_v3(-env.lightDir[0], -env.lightDir[1], -env.lightDir[2]);
    float sun = rt_pow(rt_max(0.f, rt_dot(ray.dir, negL)), env.horizonColor[3]) * env.zenithColor[3];
    float g2s = rt_smoothstep(-0.01f, 0.f, -ray.dir.y);
    float sunMask = g2s >= 1.f ? 1.f : 0.f;
    float add = sun * sunMask;
    return rt_v3(rt_mix(env.groundColor[0], sky.x, g2s) + add, rt_mix(env.groundColor[1], sky.y, g2s) + add,
                 rt_mix(env.groundColor[2], sky.z, g2s) + add);
}

// raytrace.comp:368-387 (hard-wired Cornell ceiling light)
rt_vec3 lightSampleDir(const rt_vec3& rayOrigin, uint32_t& state) {
    float x = rt_random(&state);
    float z = rt_random(&state);
    float randomX = rt_mix(-0.33333f, 0.33333f, x);
    float randomZ = rt_mix(-0.33333f, 0.33333f, z);
    rt_vec3 randomPoint = rt_v3(randomX, -1.5f, randomZ);
    return rt_normalize(rt_sub(randomPoint, rayOrigin));
}

// raytrace.comp:389-403, given the closest hit of the probe ray
float lightSamplePDF_fromHit(const Scene& sc, const HitInfo& hit, const rt_vec3& direction) {
    if (!hit.didHit || sc.a.materials[hit.materialIndex].emissionStrength == 0.f) return 0.f;
    float sqRadius = hit.dst * hit.dst;
    float cosTheta = rt_dot(rt_v3(0.f, -1.f, 0.f), direction);
    return sqRadius / (cosTheta * 0.4444444f);
}

// raytrace.comp:405-424
rt_vec3 cosineHemisphereDir(const rt_vec3& n, uint32_t& state) {
    float r1 = rt_random(&state);
    float r2 = rt_random(&state);
    float phi = (2.f * RT_PI) * r1;
    float sqrtR2 = rt_sqrt(r2);
    float sn, cs;
    rt_sincos(phi, &sn, &cs);
    float x = cs * sqrtR2;
    float y = sn * sqrtR2;
    float z = rt_sqrt(1.f - r2);
    rt_vec3 axis = rt_abs(rt_dot(n, rt_v3(1.f, 0.f, 0.f))) < 1.f ? rt_v3(1.f, 0.f, 0.f) : rt_v3(0.f, 0.f, 1.f);
    rt_vec3 t = rt_normalize(rt_cross(n, axis));
    rt_vec3 b = rt_cross(n, t);
    return rt_add(rt_add(rt_scale(t, x), rt_scale(b, y)), rt_scale(n, z));
}

// raytrace.comp:426-428
float cosineHemispherePDF(const rt_vec3& n, const rt_vec3& direction) {
    return rt_max(0.f, rt_dot(direction, n) * RT_INV_PI);
}

// ---- the wavefront pipeline's light queries (ray_tracer_amd/csrc/rt_kernels.hip.h: emitter_min_t, trace_wave's leaf step).
// The NEE ray and the cosine probe of a diffuse bounce only ask whether their closest hit is emissive and how far it is
// (raytrace.comp:389-403,443-460). The pipeline tests the emissive primitives directly (no boxes), which gives tE, the
// nearest of them on the ray; if there is none, or a sphere is nearer, it does not trace the ray; otherwise it traverses and
// stops at the first triangle hit nearer than tE. This function restates that, for the "executed work" counters the GPU
// tests compare, and checks the answer against the shader's own closest hit (lightQueryMismatch must stay 0).
float emitter_min_t(const Scene& sc, const Ray& ray, uint64_t& tested) {
    float tE = RT_MISS_DST;
    for (uint32_t i = 0; i < sc.sphereCount && i < 32; i++) {
        if (!((sc.emitSphereMask >> i) & 1u)) continue;
        HitInfo h = sphereIntersection(sc.a.spheres[i], ray);
        if (h.didHit && h.dst < tE) tE = h.dst;
        tested++;
    }
    for (const auto& e : sc.emitTris) {
        if (e.first >= sc.objectCount) continue;
        const float* inv = &sc.inv[(size_t)e.first * 16];
        Ray tr;
        tr.dir = rt_xform_dir(inv, ray.dir);
        tr.origin = rt_xform_point(inv, ray.origin);
        const Triangle& tri = sc.a.triangles[e.second];
        HitInfo h = triangleIntersection(tr, sc.a.triPoints[tri.v0], sc.a.triPoints[tri.v1], sc.a.triPoints[tri.v2], tri.frontOnly != 0);
        if (h.didHit && h.dst < tE) tE = h.dst;
        tested++;
    }
    return tE;
}

// albedo *= texel(material.albedoIndex, hit.uv), nearest filter, sampler by object.samplerIndex (0 repeat, 1 clamp to edge,
// src/vk_engine.cpp:525-531), R8G8B8A8_SRGB decoded to linear
bool g_cameraReuse = true;   // mirrors rt_set_tuning("camera_reuse", v): only the executed-work counters depend on it

rt_vec3 albedoTexel(const Scene& sc, const HitInfo& hit, const RayMaterial& m) {
    if (m.albedoIndex < 0 || (size_t)m.albedoIndex >= g_textures.size() || hit.isSphere) return rt_v3(1.f, 1.f, 1.f);
    const OracleTexture& t = g_textures[(size_t)m.albedoIndex];
    const bool clampEdge = sc.a.objects[hit.objectHitIndex].samplerIndex == 1u;
    const uint32_t x = rt_tex_index(hit.uv[0], t.width, clampEdge), y = rt_tex_index(1.f - hit.uv[1], t.height, clampEdge);
    const uint8_t* px = &t.rgba[((size_t)y * t.width + x) * 4];
    return rt_v3(rt_srgb8_to_linear(px[0]), rt_srgb8_to_linear(px[1]), rt_srgb8_to_linear(px[2]));
}

struct PathCtx {
    const Scene& sc;
    const PushConstants& pc;
    Totals& tot;
};

void tally_ref(PathCtx& c, const Tally& t, const HitInfo& h) {
    c.tot.boxRef += t.box; c.tot.triRef += t.tri; c.tot.raysRef++; c.tot.hitsRef += h.didHit;
}
void tally_unique(PathCtx& c, const Tally& t, const HitInfo& h) {
    c.tot.boxUnique += t.box; c.tot.triUnique += t.tri; c.tot.raysUnique++; c.tot.hitsUnique += h.didHit;
}

// What the pipeline executes for one light query, and whether its answer is the shader's (`full` = the shader's closest hit).
void executed_light_query(PathCtx& c, const Ray& ray, const Tally& fullTally, const HitInfo& full) {
    if (!c.sc.emitMode) { tally_unique(c, fullTally, full); return; }
    const float tE = emitter_min_t(c.sc, ray, c.tot.emitterTests);
    float sphereBest = RT_MISS_DST;
    for (uint32_t i = 0; i < c.sc.sphereCount; i++) {
        HitInfo h = sphereIntersection(c.sc.a.spheres[i], ray);
        if (h.didHit && h.dst < sphereBest) sphereBest = h.dst;
    }
    const bool fullEmissive = full.didHit && !(c.sc.a.materials[full.materialIndex].emissionStrength == 0.f);
    if (!(tE < RT_MISS_DST) || sphereBest < tE) {  // answered by the ray's creator: "not emissive"
        if (fullEmissive) c.tot.lightQueryMismatch++;
        return;
    }
    Tally t;
    HitInfo h = calculateIntersections(c.sc, ray, t, nullptr, tE);
    tally_unique(c, t, h);
    if (h.didHit) {  // ran to the end: the shader's closest hit itself
        if (!full.didHit || h.dst != full.dst || h.objectHitIndex != full.objectHitIndex || h.isSphere != full.isSphere) c.tot.lightQueryMismatch++;
    } else if (fullEmissive) {
        c.tot.lightQueryMismatch++;
    }
}

// raytrace.comp:430-464. `auxNeeded` says whether the wavefront pipeline
// would have traced the two probe rays (it skips them when the path ends at
// this bounce, because directLight/misWeight are then never read); the
// oracle always traces them, as the shader does, and only files the tallies
// under different counters.
BxDFResult diffuseBRDF(PathCtx& c, const HitInfo& prevHit, uint32_t& state, Tally aux[3], HitInfo auxHit[3], Ray auxRay[2]) {
    const RayMaterial& hitMaterial = c.sc.a.materials[prevHit.materialIndex];
    rt_vec3 albedo = rt_v3(hitMaterial.albedo[0], hitMaterial.albedo[1], hitMaterial.albedo[2]);
    if (!g_textures.empty() && hitMaterial.albedoIndex >= 0 && !prevHit.isSphere) albedo = rt_mul(albedo, albedoTexel(c.sc, prevHit, hitMaterial));
    rt_vec3 origin = rt_add(prevHit.hitPoint, rt_scale(prevHit.normal, 0.01f));

    rt_vec3 lightSample = lightSampleDir(origin, state);
    rt_vec3 cosineSample = cosineHemisphereDir(prevHit.normal, state);

    Ray lightRay{origin, lightSample};
    auxRay[0] = lightRay;
    HitInfo lightHit = calculateIntersections(c.sc, lightRay, aux[0], &c.tot.stackOverflow);  // :443
    auxHit[0] = lightHit;
    const RayMaterial& lightMaterial = c.sc.a.materials[lightHit.didHit ? lightHit.materialIndex : 0];

    HitInfo dup = calculateIntersections(c.sc, lightRay, aux[1], &c.tot.stackOverflow);      // :447 (same ray as :443)
    auxHit[1] = dup;
    float realLightPDF = lightSamplePDF_fromHit(c.sc, dup, lightSample);
    float cosinePDF = cosineHemispherePDF(prevHit.normal, lightSample);
    float misWeight1 = realLightPDF * realLightPDF / (realLightPDF * realLightPDF + cosinePDF * cosinePDF);
    if (rt_isnan(misWeight1)) misWeight1 = 0.f;

    Ray probe{origin, cosineSample};
    auxRay[1] = probe;
    HitInfo probeHit = calculateIntersections(c.sc, probe, aux[2], &c.tot.stackOverflow);    // :453
    auxHit[2] = probeHit;
    float lightPDF = lightSamplePDF_fromHit(c.sc, probeHit, cosineSample);
    float realCosinePDF = cosineHemispherePDF(prevHit.normal, cosineSample);
    float misWeight2 = realCosinePDF * realCosinePDF / (lightPDF * lightPDF + realCosinePDF * realCosinePDF);
    if (rt_isnan(misWeight2)) misWeight2 = 0.f;

    float nDotC = rt_dot(prevHit.normal, cosineSample);
    rt_vec3 radiance = rt_scale(rt_scale(albedo, RT_INV_PI), nDotC);
    radiance = rt_v3(radiance.x / realCosinePDF, radiance.y / realCosinePDF, radiance.z / realCosinePDF);
    rt_vec3 directLight = rt_scale(rt_v3(lightMaterial.emissionColor[0], lightMaterial.emissionColor[1],
                                         lightMaterial.emissionColor[2]), lightMaterial.emissionStrength);
    float k = (realLightPDF == 0.f) ? 0.f : misWeight1 / realLightPDF;
    rt_vec3 f = rt_scale(rt_scale(rt_scale(albedo, RT_INV_PI), rt_max(0.f, rt_dot(prevHit.normal, lightSample))), k);
    directLight = rt_mul(directLight, f);

    return BxDFResult{cosineSample, radiance, directLight, 1.f, misWeight2};
}

// raytrace.comp:466-469
BxDFResult specularBRDF(const rt_vec3& incoming, const HitInfo& prevHit) {
    return BxDFResult{rt_reflect(incoming, prevHit.normal), rt_v3(1.f, 1.f, 1.f), rt_v3(-1.f, -1.f, -1.f), 1.f, 1.f};
}

// raytrace.comp:471-481
BxDFResult dielectricBTDF(PathCtx& c, const rt_vec3& incoming, const HitInfo& prevHit, uint32_t& state) {
    const RayMaterial& m = c.sc.a.materials[prevHit.materialIndex];
    float ior = !prevHit.frontFace ? m.ior : 1.f / m.ior;
    float cosine = rt_dot(rt_neg(incoming), prevHit.normal);
    float sine = rt_sqrt(1.f - cosine * cosine);
    // GLSL || short-circuits: no draw on total internal reflection
    bool solution = (ior * sine) > 1.f || schlick(cosine, ior) > rt_random(&state);
    rt_vec3 dir = solution ? rt_reflect(incoming, prevHit.normal) : rt_refract(incoming, prevHit.normal, ior);
    float sgn = solution ? 1.f : rt_sign(rt_dot(prevHit.normal, incoming));
    return BxDFResult{dir, rt_v3(1.f, 1.f, 1.f), rt_v3(-1.f, -1.f, -1.f), sgn, 1.f};
}

// raytrace.comp:483-537
rt_vec3 trace(PathCtx& c, Ray ray, uint32_t& state, Tally& mainStats, uint32_t sampleIndex) {
    rt_vec3 totalColor = rt_v3(0.f, 0.f, 0.f);
    rt_vec3 attenuation = rt_v3(1.f, 1.f, 1.f);
    rt_vec3 directLight = rt_v3(0.f, 0.f, 0.f);
    float misWeight = 1.f;
    Ray newRay = ray;
    const uint32_t bounceLimit = c.pc.rayTraceParams.bounceLimit;
    c.tot.paths++;

    for (uint32_t j = 0; j <= bounceLimit; j++) {
        Tally t;
        HitInfo hit = calculateIntersections(c.sc, newRay, t, &c.tot.stackOverflow);
        mainStats.box += t.box; mainStats.tri += t.tri;
        tally_ref(c, t, hit);
        // the pipeline keeps the camera ray's hit from a pixel's first sample (every sample starts with the same ray, :541-557):
        // the first segment of the later samples is not traversed again (not so for the heat maps, which count per pixel)
        if (!(j == 0 && sampleIndex > 0 && g_cameraReuse && c.pc.rayTraceParams.debug < 0)) tally_unique(c, t, hit);
        c.tot.segments++;
        if (hit.didHit) {
            const RayMaterial& m = c.sc.a.materials[hit.materialIndex];
            rt_vec3 emission = rt_scale(rt_v3(m.emissionColor[0], m.emissionColor[1], m.emissionColor[2]), m.emissionStrength);
            emission = rt_v3(emission.x / misWeight, emission.y / misWeight, emission.z / misWeight);
            rt_vec3 finalLight = directLight.x == -1.f ? emission : directLight;
            totalColor = rt_add(totalColor, rt_mul(finalLight, attenuation));
            if (j == 0) totalColor = rt_add(totalColor, emission);
            if (rt_isnan(totalColor.x) || rt_isnan(totalColor.y) || rt_isnan(totalColor.z) || totalColor.x < 0.f ||
                totalColor.y < 0.f || totalColor.z < 0.f)
                return rt_v3(0.f, 0.f, 0.f);

            BxDFResult bxdf;
            bool diffuse = false;
            Tally aux[3];
            HitInfo auxHit[3];
            Ray auxRay[2];
            float reflectance = m.reflectance;
            if (!hit.isSphere) {  // metalness map (declared, rt_det_math.h): the texel's decoded red replaces the material's reflectance
                if (const OracleTexture* mt = mapOf(m.metalnessIndex))
                    reflectance = rt_srgb8_to_linear(mapRed8(*mt, hit.uv, c.sc.a.objects[hit.objectHitIndex].samplerIndex == 1u));
            }
            if (reflectance != 0.f) {
                bxdf = specularBRDF(newRay.dir, hit);
            } else if (m.ior != -1.f) {
                bxdf = dielectricBTDF(c, newRay.dir, hit, state);
            } else {
                bxdf = diffuseBRDF(c, hit, state, aux, auxHit, auxRay);
                diffuse = true;
            }
            attenuation = rt_mul(attenuation, bxdf.radiance);
            directLight = bxdf.directLight;

            float rrProb = rt_max(rt_max(attenuation.x, attenuation.y), attenuation.z);
            rrProb = rt_min(rrProb, 0.95f);
            rrProb = j <= 5 ? 1.f : rrProb;
            bool rrBreak = rt_random(&state) > rrProb;
            if (diffuse) {
                for (int k = 0; k < 3; k++) tally_ref(c, aux[k], auxHit[k]);
                // the pipeline traces the NEE ray and the cosine probe once each,
                // and only when a later segment can read their results
                if (!rrBreak && j < bounceLimit) {
                    executed_light_query(c, auxRay[0], aux[0], auxHit[0]);
                    executed_light_query(c, auxRay[1], aux[2], auxHit[2]);
                }
            }
            if (rrBreak) break;
            float invP = 1.f / rrProb;
            attenuation = rt_scale(attenuation, invP);

            misWeight = bxdf.cosineMisWeight;
            newRay.origin = rt_add(hit.hitPoint, rt_scale(rt_scale(hit.normal, bxdf.originSign), 0.00001f));
            newRay.dir = bxdf.sampledDir;
        } else {
            totalColor = rt_add(totalColor, rt_mul(attenuation, getEnvironmentLight(c.pc, newRay)));
            break;
        }
    }
    return totalColor;
}

// raytrace.comp:539-594 for one pixel
void pixel_main(PathCtx& c, uint32_t gx, uint32_t gy, uint32_t W, uint32_t H, float* rgba) {
    const CameraInfo& cam = c.pc.camInfo;
    float u = (float)gx / (float)W;
    float v = (float)gy / (float)H;
    float planeHeight = cam.nearPlane * rt_tan(rt_radians(cam.fov * 0.5f)) * 2.f;
    float planeWidth = planeHeight * cam.aspectRatio;
    rt_vec3 bottomLeft = rt_v3(-planeWidth / 2.f, -planeHeight / 2.f, 0.1f);
    rt_vec3 point = rt_add(bottomLeft, rt_v3(planeWidth * u, planeHeight * v, 0.f));
    rt_vec3 dir = rt_normalize(point);
    Ray ray;
    ray.dir = rt_xform_point(cam.cameraRotation, dir);  // vec4(dir, 1): w = 1 (:555)
    ray.origin = rt_v3(cam.pos[0], cam.pos[1], cam.pos[2]);

    uint32_t lol = c.pc.frameCount;
    uint32_t startingSeed = (uint32_t)(rt_random(&lol) * 23892183.f);
    uint32_t state = gy * W + gx + startingSeed;

    const RayTracerData& td = c.pc.rayTraceParams;
    Tally stats;
    rt_vec3 outColor = rt_v3(0.f, 0.f, 0.f);
    uint32_t samples = td.singleRender ? td.sampleLimit : td.raysPerPixel;
    for (uint32_t i = 0; i < samples; i++) outColor = rt_add(outColor, trace(c, ray, state, stats, i));
    float fs = (float)samples;
    outColor = rt_v3(outColor.x / fs, outColor.y / fs, outColor.z / fs);

    float weight = 1.f / ((float)c.pc.frameCount + 1.f);
    rt_vec3 old = rt_v3(rgba[0], rgba[1], rgba[2]);
    rt_vec3 blended = rt_add(rt_scale(old, 1.f - weight), rt_scale(outColor, weight));
    rt_vec3 finalColor = td.progressive ? blended : outColor;
    if (rt_isnan(finalColor.x) || rt_isnan(finalColor.y) || rt_isnan(finalColor.z) || rt_isinf(finalColor.x) ||
        rt_isinf(finalColor.y) || rt_isinf(finalColor.z))
        finalColor = rt_v3(1.f, 0.f, 1.f);

    float s0 = (float)stats.box, s1 = (float)stats.tri;
    float boxCap = (float)td.boxCap, triCap = (float)td.triangleCap;
    if (td.debug == 0) {
        finalColor = s0 > boxCap ? rt_v3(1.f, 0.f, 0.f) : rt_v3(s0 / boxCap, s0 / boxCap, s0 / boxCap);
    } else if (td.debug == 1) {
        finalColor = s1 > triCap ? rt_v3(1.f, 0.f, 0.f) : rt_v3(s1 / triCap, s1 / triCap, s1 / triCap);
    } else if (td.debug == 2) {
        finalColor = rt_v3(s0 / boxCap, 0.f, s1 / triCap);
    }
    rgba[0] = finalColor.x; rgba[1] = finalColor.y; rgba[2] = finalColor.z; rgba[3] = 1.f;
}

Scene make_scene(const RtSceneArrays* a, uint32_t sphereCount, uint32_t objectCount, bool lightQueries = true) {
    Scene sc;
    sc.a = *a;
    sc.sphereCount = sphereCount;
    sc.objectCount = objectCount;
    sc.inv.resize((size_t)objectCount * 16);
    for (uint32_t i = 0; i < objectCount; i++) rt_mat4_inverse(a->objects[i].transformMatrix, &sc.inv[(size_t)i * 16]);
    // the emitter list, by the rules of rt_device.hip's rebuild_emitters (over the UPLOADED scene: a->sphereCount / a->objectCount)
    auto emissive = [&](uint32_t m) { return m < a->materialCount && !(a->materials[m].emissionStrength == 0.f); };
    bool ok = lightQueries;
    for (uint32_t m = 0; m < a->materialCount && ok; m++)
        for (int k = 0; k < 3; k++) {
            float p = a->materials[m].emissionColor[k] * a->materials[m].emissionStrength;
            if (rt_isnan(p) || rt_isinf(p)) ok = false;
        }
    for (uint32_t i = 0; i < a->sphereCount && ok; i++)
        if (emissive(a->spheres[i].materialIndex)) {
            if (i < 32) sc.emitSphereMask |= 1u << i; else ok = false;
        }
    for (uint32_t i = 0; i < a->objectCount && ok; i++) {
        if (!emissive(a->objects[i].materialIndex)) continue;
        if (mapOf(a->materials[a->objects[i].materialIndex].alphaIndex)) { ok = false; break; }  // an emitter with holes: its list entries would need the map
        std::vector<uint32_t> st{a->objects[i].bvhIndex};
        uint64_t lo = ~0ull, hi = 0, sum = 0;
        while (!st.empty()) {
            const BVHNode& n = a->bvhNodes[st.back()];
            st.pop_back();
            if (n.triCount) {
                lo = lo < n.index ? lo : n.index;
                hi = hi > (uint64_t)n.index + n.triCount ? hi : (uint64_t)n.index + n.triCount;
                sum += n.triCount;
            } else {
                st.push_back(n.index);
                st.push_back(n.index + 1);
            }
            if (sum > RT_EMIT_MAX_TRIS) break;
        }
        if (sum != hi - lo || sc.emitTris.size() + sum > (size_t)RT_EMIT_MAX_TRIS) { ok = false; break; }
        for (uint64_t t = lo; t < hi; t++) sc.emitTris.emplace_back(i, (uint32_t)t);
    }
    sc.emitMode = ok;
    if (!ok) { sc.emitTris.clear(); sc.emitSphereMask = 0; }
    return sc;
}

void add_totals(OracleCounters& o, const Totals& t) {
    o.boxTestsReference += t.boxRef; o.triTestsReference += t.triRef;
    o.raysReference += t.raysRef; o.raysHitReference += t.hitsRef;
    o.boxTests += t.boxUnique; o.triTests += t.triUnique;
    o.raysTraced += t.raysUnique; o.raysHit += t.hitsUnique;
    o.paths += t.paths; o.segments += t.segments; o.stackOverflow += t.stackOverflow;
    o.emitterTests += t.emitterTests; o.lightQueryMismatch += t.lightQueryMismatch;
}

bool g_lightQueries = true;

}  // namespace

extern "C" {

// the texture table of the scene the next oracle_render calls use (rt_upload_textures); n = 0 removes it
void oracle_set_textures(const RtTexture* tex, uint32_t n) {
    g_textures.clear();
    for (uint32_t i = 0; i < n; i++) {
        OracleTexture t;
        t.width = tex[i].width; t.height = tex[i].height;
        t.rgba.assign(tex[i].rgba8, tex[i].rgba8 + (size_t)t.width * t.height * 4);
        g_textures.push_back(std::move(t));
    }
}

void oracle_set_camera_reuse(int on) { g_cameraReuse = on != 0; }

// mirrors rt_set_tuning("light_queries", v): which definition of "executed work" the counters follow (pixels never change)
void oracle_set_light_queries(int on) { g_lightQueries = on != 0; }

int oracle_render(const RtSceneArrays* scene, const PushConstants* pc, uint32_t width, uint32_t height, uint32_t row0,
                  uint32_t rowStride, uint32_t nRows, float* rgba, OracleCounters* counters, int threads) {
    if (!scene || !pc || !rgba || rowStride == 0) return -1;
    if (pc->rayTraceParams.sphereCount > scene->sphereCount || pc->rayTraceParams.objectCount > scene->objectCount) return -2;
    Scene sc = make_scene(scene, pc->rayTraceParams.sphereCount, pc->rayTraceParams.objectCount, g_lightQueries);
    if (threads < 1) threads = 1;
    std::vector<Totals> totals(threads);
    // work items are 64-pixel runs of a row so that many threads stay busy on few rows
    const uint32_t chunk = 64, chunksPerRow = (width + chunk - 1) / chunk;
    const uint64_t nChunks = (uint64_t)nRows * chunksPerRow;
    std::atomic<uint64_t> next{0};
    auto worker = [&](int tid) {
        PathCtx c{sc, *pc, totals[tid]};
        for (;;) {
            uint64_t w = next.fetch_add(1);
            if (w >= nChunks) break;
            uint32_t k = (uint32_t)(w / chunksPerRow), x0 = (uint32_t)(w % chunksPerRow) * chunk;
            uint32_t gy = row0 + k * rowStride;
            uint32_t x1 = x0 + chunk < width ? x0 + chunk : width;
            for (uint32_t gx = x0; gx < x1; gx++) pixel_main(c, gx, gy, width, height, rgba + ((size_t)k * width + gx) * 4);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        for (auto& t : totals) add_totals(*counters, t);
    }
    return 0;
}

int oracle_trace_rays(const RtSceneArrays* scene, uint32_t sphereCount, uint32_t objectCount, uint32_t n,
                      const float* origins, const float* dirs, RtHit* out) {
    if (!scene || !origins || !dirs || !out) return -1;
    if (sphereCount > scene->sphereCount || objectCount > scene->objectCount) return -2;
    Scene sc = make_scene(scene, sphereCount, objectCount);
    for (uint32_t i = 0; i < n; i++) {
        Ray r{rt_v3(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]), rt_v3(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2])};
        Tally t;
        HitInfo h = calculateIntersections(sc, r, t, nullptr);
        RtHit& o = out[i];
        memset(&o, 0, sizeof(o));
        o.dst = h.dst;
        o.didHit = h.didHit;
        o.boxTests = t.box;
        o.triTests = t.tri;
        if (h.didHit) {
            o.isSphere = h.isSphere;
            o.objectHitIndex = h.objectHitIndex;
            o.triHitIndex = h.isSphere ? 0 : h.triHitIndex;
            o.materialIndex = h.materialIndex;
            o.frontFace = h.frontFace;
            o.hitPoint[0] = h.hitPoint.x; o.hitPoint[1] = h.hitPoint.y; o.hitPoint[2] = h.hitPoint.z;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
        }
    }
    return 0;
}

// raytrace.comp:158-163, exposed for the known-answer tests
float oracle_random(uint32_t* state) { return rt_random(state); }

void oracle_math_probe(float x, float y, float out[8]) {
    float s, c;
    rt_sincos(x, &s, &c);
    out[0] = s; out[1] = c; out[2] = rt_tan(x); out[3] = rt_log2(x); out[4] = rt_exp2(x);
    out[5] = rt_pow(x, y); out[6] = rt_smoothstep(0.f, 0.4f, x); out[7] = rt_sqrt(x);
}

void oracle_mat4_inverse(const float m[16], float out[16]) { rt_mat4_inverse(m, out); }

// every GLSL built-in of SURVEY A12 on caller-supplied inputs (include/rt_probe.h): n x 32 floats in, n x 64 out
void oracle_glsl_probe(uint32_t n, const float* in, float* out) {
    for (uint32_t i = 0; i < n; i++) rt_math_probe(in + (size_t)i * 32, out + (size_t)i * 64);
}

uint32_t oracle_selftest(void) {
    volatile float in[7] = {1.0001220703125f, 0.9998779296875f, -1.f, 3.f, 1e-30f, 1e-10f, 2.f};
    return rt_selftest_bits(in);
}

unsigned oracle_hardware_threads(void) { return std::thread::hardware_concurrency(); }

}  // extern "C"
