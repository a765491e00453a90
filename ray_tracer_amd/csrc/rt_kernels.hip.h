// rt_kernels.hip.h — device data layout and the wavefront kernels for gfx950.
//
// The per-pixel megakernel shaders/raytrace.comp is split into four kernels
// that exchange work through device queues (indices into SoA path state):
//
//   k_raygen   raytrace.comp:539-564   camera ray, RNG seed, path reset
//   k_trace    raytrace.comp:276-353   closest hit: spheres + per-object BVH
//              (+ :195-274)            traversal with an LDS-resident stack
//   k_shade    raytrace.comp:483-537   emission/NEE add, BSDF sample
//              (+ :356-481)            (diffuse+MIS / mirror / dielectric),
//                                      Russian roulette, next rays, path
//                                      regeneration for the next sample
//   k_resolve  raytrace.comp:566-593   sample mean, progressive blend,
//                                      NaN->magenta, debug heat maps, store
//
// One path slot per pixel; a pixel's samples run one after another in its
// slot because the shader carries the RNG state serially across the samples
// of a pixel (:564,571-573). Every arithmetic step uses
// rt_det_math.h in the same order as the scalar oracle so that pixels and the
// box/triangle counters are bit-identical, not just within 1e-4.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"
#include "rt_det_math.h"
#include "rt_probe.h"

#define RT_WAVE 64
#ifndef RT_TE_REG
#define RT_MAP_METALNESS 1u
#define RT_MAP_ALPHA 2u
#define RT_MAP_BUMP 4u
#ifndef RT_QUEUE_ORDER
#define RT_QUEUE_ORDER 0   // k_shade: the ray queue's pieces are {main, NEE, cosine probes} (1: {main, cosine probes, NEE})
#endif
#define RT_TE_REG 1        // trace_wave<ROOMY>: a light query's tE rides in a register instead of being re-read from the hit record when a leaf step finds a hit
#endif
#ifndef RT_OBJTREE
#define RT_OBJTREE 1       // trace_wave, set-up step: the object hierarchy's block jumps compiled in (CULL kernels)
#endif
#ifndef RT_SHADE_IDENT
#define RT_SHADE_IDENT 1   // reconstruct_hit: no matrix loads / transforms for identity-transform objects hit by a plain ray
#endif
#define RT_BLOCK 256
#define RT_LEAF_BIT 0x80000000u
#define RT_HIT_NONE 0xffffffffu
#define RT_HIT_SPHERE 0x80000000u
#define RT_OBJTREE_LEVELS 8   // blocks of up to 256 objects in the object hierarchy (DevScene::objTree)

// ---------------------------------------------------------------- scene in HBM
// All arrays are read-only during a render.
//   nodes   : 2 x float4 per BVH node  {min.xyz, W0} {max.xyz, triCount}
//             W0 = index of the child pair (interior), or a ready-made leaf
//             reference LEAF|count<<28|firstTriangle (count <= 7), or
//             LEAF|nodeIndex for bigger leaves (first triangle in leafFirst[])
//             children of an interior node are adjacent and 64-B aligned, so
//             one interior visit is one 64-B fetch (the reference re-reads the
//             popped node and both children: 96 B, raytrace.comp:306,326-327)
//   triPos  : 3 x float4 per triangle   {v0.xyz, frontOnly} {v1.xyz,-} {v2.xyz,-}
//             in the reference's (builder-permuted) triangle order; hot
//   triNrm  : 3 x float4 per triangle   vertex normals; cold, read per hit only
//   objInv  : 3 x float4 per object     rows of inverse(transformMatrix)[0..2]
//   objFwd  : 3 x float4 per object     rows of transformMatrix[0..2]
//   objMeta : uint4 per object          {rootIndex|pairIndex, rootTriCount, materialIndex, flags (bit 0: identity transform, bit 1: objBox valid)}
//   objBox  : 2 x float4 per object     padded world-space box of a general-transform object
//   mats    : 3 x float4 per material   {albedo, reflectance} {emission, strength} {ior,-,-,-}
//   spheres : float4 {center, radius} + uint material
struct DevScene {
    const float4* nodes;
    const float4* nodesPk;   // child pairs interleaved for packed math, see rt_upload_scene
    const uint32_t* leafFirst;  // first triangle of a leaf, per node (read only for leaves with > 7 triangles)
    const float4* triPos;
    const float4* triNrm;
    const float4* objInv;
    const float4* objFwd;
    const uint4* objMeta;
    const float4* objBox;    // 2 x float4 per object: {lo.xyz, flags} {hi.xyz, root triangle count}; flags bit 0 identity transform,
                             // bit 1 padded world-space box of a general-transform object, bit 2 the box may clear the object's bit in a ray's object mask
                             // (identity object: its exact root box; general object: the padded world box)
    const float4* mats;
    const float4* spheres;
    const uint32_t* sphereMat;
    uint32_t sphereCount, objectCount, materialCount, nodeCount, triCount;
    uint32_t hotNodes;        // the device numbering puts the child pairs of the top levels of every mesh first (breadth first over all
                              // roots): node indices below this (an even number, at most 2 * RT_HOT_PAIRS) are the pairs a work-group of
                              // k_trace_pw<HOT> keeps in LDS
    uint32_t sphereTestMask;  // spheres sphere_seed has to test: all but those whose {center, radius} repeat an earlier sphere's bit for
                              // bit (the reference always uploads MAX_SPHERES = 10, src/vk_engine.cpp:682-686, zeroed when unused).
                              // A repeat computes the earlier sphere's very result and can never win the loop's strict `dst < best`
                              // (raytrace.comp:282-287), so leaving it out changes nothing
    const float4* maskBox;   // reachCount x 2 float4: the objects a ray's creator tests for the ray's object mask (reach_mask_from)
    const uint2* objSkipCost; // 33 entries: {box tests, triangle tests} the reference spends on objects [maskBase, maskBase + i) when a ray misses them all
    uint32_t reachCount;     // entries of maskBox
    // A hierarchy over the padded world boxes of consecutive general-transform objects, for scenes of many placed objects (the
    // reference walks its objects linearly, raytrace.comp:289-350: two box tests per object a ray misses; 256 separated instances
    // cost a ray 7 times the traversal of the same triangles in one mesh). Level k (1..objTreeLevels) holds, for every aligned
    // block of 2^k objects that are all general-transform objects with a padded box, the union of their boxes ({lo.xyz, 1}{hi.xyz, -};
    // w = 0: not such a block) at objTree[2 * (objTreeOff[k] + (object >> k))]. A ray that cannot reach a block's box before
    // its closest hit cannot reach any object in it, so the set-up step's skipping loop jumps the whole block at the reference's
    // cost for it (objCost: prefix sums over ALL objects of {box tests, triangle tests} a missed object is worth), in the
    // reference's object order. objTreeLevels = 0: no hierarchy (few placed objects: rt_update_objects).
    const float4* objTree;
    const uint2* objCost;
    uint32_t objTreeOff[RT_OBJTREE_LEVELS + 1];
    uint32_t objTreeLevels;
    uint32_t maskBase;       // a ray's object mask covers objects [maskBase, maskBase + 32): the window starts at the first object that can be
                             // ruled out at all (C5: 26 identity-transform groups, then sixteen placed dragons — all sixteen inside the window)
    // Light queries (the NEE ray and the cosine probe of a diffuse bounce, raytrace.comp:443-453) only ask "is the closest hit
    // emissive, and how far is it". emitTris lists every triangle of every object whose material is emissive ({object,
    // triangle}, sorted by object), emitSphereMask the emissive spheres; the creator of such a ray tests them all and knows the
    // nearest emissive primitive's distance tE before any traversal (emitter_min_t2). emitMode 0: too many emissive triangles
    // (or non-finite emission values): every light query is traversed in full, as round 1 did.
    const uint2* emitTris;
    const float4* emitPre;   // per listed triangle {v0, frontOnly} {v1 - v0} {v2 - v0} {cross(v1 - v0, v2 - v0)}: the part of tri_intersect that
                             // does not depend on the ray, computed once by k_emit_precompute with tri_intersect's own operations
    uint32_t emitCount, emitSphereMask, emitMode;
    // Textures: texels of all slots back to back (R8G8B8A8_SRGB, one uint32 per texel), texInfo[slot] =
    // {first texel, width, height, -}, triUV = 2 x float4 per triangle {u0 v0 u1 v1} {u2 v2 - -} (cold: read per shaded hit of a
    // textured material only). mats[3 m + 2].y carries albedoIndex, objMeta[].w bits 16.. the object's samplerIndex.
    const uint32_t* texels;
    const uint4* texInfo;
    const float4* triUV;
    uint32_t texCount;
    // The other three map slots (declared semantics: include/rt_det_math.h). mats[3 m + 2].z / .w carry metalnessIndex / bumpIndex,
    // objAlpha[object] the alpha map of the object's material and its sampler (slot | clamp << 8; 0xffffffff = none); mapFlags says
    // which kinds the scene binds at all (RT_MAP_*): the kernels that read them are separate ones (k_shade_maps, k_trace_pw_alpha)
    const uint32_t* objAlpha;
    uint32_t mapFlags;
    float cullOriginLimit;   // rays that start farther out than this (max |origin component|) skip nothing: the padding of the world-space
                             // boxes (1e-3 of an object's size and position) only dominates the slab tests' rounding, which grows with
                             // |origin|, while the origin is within 1e3 object scales (rt_update_objects)
};

// ---------------------------------------------------------------- path state (SoA, one slot per pixel)
enum { RAY_MAIN = 0, RAY_NEE = 1, RAY_PROBE = 2 };

// 16-byte records per slot: a path touches its state through a dozen dwordx4
// accesses instead of ~85 dword accesses (k_shade is bound by the number of
// memory instructions in flight, not by bytes).
// One base pointer and the array pitch instead of fifteen pointers: the kernels that shade and traverse in one
// (k_render_fused) are short of scalar registers, and an array's address is one scalar multiply-add away.
struct PathState {
    float4* base;
    uint32_t pitch;       // float4 elements between consecutive arrays (>= slots, 256-byte multiple)
    uint32_t pitchStat;   // uint32 elements between statBox and statTri
    __host__ __device__ __forceinline__ float4* arr(uint32_t k) const { return base + (size_t)k * pitch; }
    __host__ __device__ __forceinline__ float4* rayO() const { return arr(0); }        // main ray origin            | w: misWeight (raytrace.comp:487)
    __host__ __device__ __forceinline__ float4* rayD() const { return arr(1); }        // main ray direction         | w: RNG state bits (:564)
    __host__ __device__ __forceinline__ float4* auxO() const { return arr(2); }        // origin of the probe rays (read by the traversal only)
    __host__ __device__ __forceinline__ float4* auxDL() const { return arr(3); }       // NEE direction              | w: cosineHemispherePDF(n, lightSample)  (:448)
    __host__ __device__ __forceinline__ float4* auxDC() const { return arr(4); }       // cosine-sample direction    | w: cosineHemispherePDF(n, cosineSample) (:454)
    __host__ __device__ __forceinline__ float4* hit(uint32_t kind) const { return arr(5 + kind); }  // per ray kind. From the ray's creator: {closest sphere hit, its object bits, object mask, tE of a light query or 0}; from the traversal: {dst, object bits, triangle bits, 0}
    __host__ __device__ __forceinline__ float4* att() const { return arr(8); }         // attenuation                | w: bounce index j (28 bits), bit 31 = NEE results pending, bit 30 = the last bounce was specular (directLight = -1, :469,480), bits 29 / 28 = the NEE ray / the cosine probe was answered "not emissive" by its creator (no record written for it)
    __host__ __device__ __forceinline__ float4* total() const { return arr(9); }       // totalColor                 | w: samples finished for this pixel
    // (directLight has no record: between two segments it is either about to be recomputed from the probe results — bit 31 of
    // att.w — or one of two constants: -1 after a specular bounce, bit 30, and 0 at the start of a sample)
    __host__ __device__ __forceinline__ float4* pendAlbedo() const { return arr(10); } // albedo of the previous diffuse hit | w: max(0, dot(n, lightSample)) (:460)
    __host__ __device__ __forceinline__ float4* accum() const { return arr(11); }      // sum of trace() over the pixel's samples (:572)
    __host__ __device__ __forceinline__ float4* camHit() const { return arr(12); }     // the camera ray's hit record, kept from the pixel's first sample (FrameParams::camReuse)
    __host__ __device__ __forceinline__ float4* camDir() const { return arr(13); }     // the camera ray's direction (every sample of the pixel starts with it: no jitter, raytrace.comp:541-557)
    __host__ __device__ __forceinline__ uint32_t* statBox() const { return (uint32_t*)arr(14); }             // stats[0] of the pixel (main-path traversals only)
    __host__ __device__ __forceinline__ uint32_t* statTri() const { return (uint32_t*)arr(14) + pitchStat; } // stats[1]
};

struct Queues {
    uint32_t* active[2];  // path slots that have rays in flight
    uint32_t* rays[2];    // slot*4 + kind
    uint32_t* counts;     // [0..1] active counts, [2..3] ray counts
};

struct FrameParams {
    // camera, precomputed on the host with rt_det_math (raytrace.comp:547-550)
    float camRot[16];
    float camPos[3];
    float planeWidth, planeHeight;
    float bottomLeft[3];
    uint32_t width, height, row0, rowStride, nRows, nPixels;
    uint32_t tiled;         // 8x8 slot order (width % 8 == 0)
    uint32_t startingSeed;  // uint(random(frameCount) * 23892183)
    uint32_t samples, bounceLimit;
    uint32_t progressive, frameCount;
    uint32_t camReuse;      // every sample of a pixel starts with the same camera ray (no jitter, raytrace.comp:541-557,571-573): its hit is
                            // kept from the first sample and samples 2.. start from it without a traversal (off for the heat maps,
                            // which count every traversal's tests per pixel)
    uint32_t nFrames;       // > 1: this launch renders that many consecutive progressive frames of the tile at once (rt_render_frames).
                            // Slots then run over {64 tile slots} x {frames}: slot = (tile slot / 64) * 64 * nFrames + frame * 64 +
                            // tile slot % 64 (frame_slot), so that the waves in flight work on one region of the tile in all its
                            // frames (the frames' camera rays are identical: no jitter, raytrace.comp:541-557) rather than on the
                            // whole tile of one frame; frame f has frameCount + f and its own startingSeed
    int32_t debug;
    uint32_t boxCap, triCap;
    EnvironmentData env;
};

struct DevCounters {
    unsigned long long boxTests, triTests, raysTraced, raysHit, raysReference, paths, segments, emitterTests;
    unsigned long long skippedBoxTests;  // of boxTests: charged for objects rays were taken past (the reference tests the children of their roots), not executed
};

// ---------------------------------------------------------------- address spaces of the scene tables
// The shading code of k_render_fused reaches the scene through a pointer to the kernel-argument segment that the compiler
// cannot see through (opaque_kernarg), so a table pointer read from it is a generic pointer to the compiler and every load
// through it a FLAT instruction. Two statements of fact put that right (address-space inference does the rest after inlining):
//   rt_uniform(p): read-only during the kernel and indexed the same by every lane of a wave (the emitter list, the spheres,
//                  frame constants) -> constant address space -> scalar loads through the scalar cache, which keeps them off
//                  the vector memory pipeline that the traversal saturates;
//   rt_global(p):  read-only tables indexed per lane (materials, objects, triangles) -> global address space -> global_load.
#define RT_AS_CONSTANT __attribute__((address_space(4)))
#define RT_AS_GLOBAL __attribute__((address_space(1)))
#define RT_AS_LDS __attribute__((address_space(3)))
typedef float rt_f4v __attribute__((ext_vector_type(4)));
typedef float rt_f2v __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ const T* rt_uniform(const T* p) { return (const T*)(const RT_AS_CONSTANT T*)(unsigned long long)p; }
template <typename T> __device__ __forceinline__ const T* rt_global(const T* p) { return (const T*)(const RT_AS_GLOBAL T*)(unsigned long long)p; }

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ rt_vec3 f4xyz(float4 v) { return rt_v3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 mk4(rt_vec3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ float4 mk4u(rt_vec3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }

// (M * vec4(v,0)).xyz and (M * vec4(v,1)).xyz with M given as three rows
// {m0,m4,m8,m12},{m1,m5,m9,m13},{m2,m6,m10,m14}; same sums as rt_xform_*.
__device__ __forceinline__ rt_vec3 xform_dir_rows(float4 r0, float4 r1, float4 r2, rt_vec3 v) {
    return rt_v3((r0.x * v.x + r0.y * v.y) + r0.z * v.z, (r1.x * v.x + r1.y * v.y) + r1.z * v.z,
                 (r2.x * v.x + r2.y * v.y) + r2.z * v.z);
}
__device__ __forceinline__ rt_vec3 xform_point_rows(float4 r0, float4 r1, float4 r2, rt_vec3 v) {
    return rt_v3(((r0.x * v.x + r0.y * v.y) + r0.z * v.z) + r0.w, ((r1.x * v.x + r1.y * v.y) + r1.z * v.z) + r1.w,
                 ((r2.x * v.x + r2.y * v.y) + r2.z * v.z) + r2.w);
}

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RT_WAVE);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RT_WAVE);
    return v;
}

// ---------------------------------------------------------------- intersections
struct SphereHit { bool didHit, frontFace; float dst; };

// raytrace.comp:195-224 (distance part)
__device__ __forceinline__ SphereHit sphere_intersect(float4 s, rt_vec3 ro, rt_vec3 rd) {
    SphereHit h; h.didHit = false; h.frontFace = false; h.dst = 0.f;
    rt_vec3 oc = rt_sub(f4xyz(s), ro);
    float a = rt_dot(rd, rd);
    float b = rt_dot(oc, rd);
    float c = rt_dot(oc, oc) - s.w * s.w;
    float disc = b * b - a * c;
    if (disc >= 0.f) {
        float sq = rt_sqrt(disc);
        float dst = (b - sq) / a;
        h.frontFace = true;
        if (dst < 0.f) {
            dst = (b + sq) / a;
            h.frontFace = false;
            if (dst < 0.f) return h;
        }
        h.didHit = true;
        h.dst = dst;
    }
    return h;
}

// The sphere loop of calculateIntersections (raytrace.comp:282-287) for one ray. The kernels that
// CREATE rays (k_raygen, k_shade) run it, with all lanes busy, and leave the result in the ray's hit
// record as the starting "closest hit" of the traversal; k_trace_pw then only walks the objects.
__device__ __forceinline__ float box_intersect(float4 lo, float4 hi, rt_vec3 ro, rt_vec3 inv);

// A ray that may use an identity-transform object's space as is: finite and non-zero direction, finite origin, no
// negative zeros (multiplying by the identity matrix would turn -0 into +0 and change 1/dir's sign of infinity)
__device__ __forceinline__ bool ray_is_plain(rt_vec3 wo, rt_vec3 wd) {
    const uint32_t E = 0x7f800000u, M = 0x7fffffffu;
    return ((rt_f2u(wd.x) & M) - 1u < E - 1u) && ((rt_f2u(wd.y) & M) - 1u < E - 1u) && ((rt_f2u(wd.z) & M) - 1u < E - 1u) &&
           ((rt_f2u(wo.x) & M) < E) && ((rt_f2u(wo.y) & M) < E) && ((rt_f2u(wo.z) & M) < E) &&
           rt_f2u(wo.x) != 0x80000000u && rt_f2u(wo.y) != 0x80000000u && rt_f2u(wo.z) != 0x80000000u;
}

// A vector the identity matrix maps to itself bit for bit: every component finite and none a negative zero
__device__ __forceinline__ bool vec_is_plain(rt_vec3 v) {
    const uint32_t E = 0x7f800000u, M = 0x7fffffffu;
    return ((rt_f2u(v.x) & M) < E) && ((rt_f2u(v.y) & M) < E) && ((rt_f2u(v.z) & M) < E) &&
           rt_f2u(v.x) != 0x80000000u && rt_f2u(v.y) != 0x80000000u && rt_f2u(v.z) != 0x80000000u;
}

// What the creator of a ray hands to the traversal (hit record of the ray's kind): the closest sphere hit (the shader's
// sphere loop, raytrace.comp:282-287) and, in .z, the objects among the 32 of the mask's window (DevScene::maskBase) the ray
// has to enter at all. Bit i is cleared when object maskBase + i has an identity transform and an interior root and the ray misses the root's box (objBox holds it,
// exactly): both of the root's children are then missed as well, the reference does its two box tests on them and moves
// on, and so does the traversal — by adding 2 to the count (trace_wave: fetch_next_meta). On the Sponza stand-in a ray
// misses 16 of the 26 material groups' boxes on average. Same slab arithmetic as the traversal's (1/dir, box_intersect).
// boxes: DevScene::maskBox, wherever the caller keeps it (k_render_fused: in LDS): n objects {lo.xyz, object index - maskBase} {hi.xyz, -}
__device__ __forceinline__ bool origin_within(rt_vec3 o, float limit) {
    return rt_max(rt_max(rt_abs(o.x), rt_abs(o.y)), rt_abs(o.z)) <= limit;
}
__device__ __forceinline__ uint32_t reach_mask_from(const float4* boxes, uint32_t n, rt_vec3 ro, rt_vec3 rd, float originLimit) {
    uint32_t reach = 0xffffffffu;
    if (n && ray_is_plain(ro, rd) && origin_within(ro, originLimit)) {
        const rt_vec3 inv = rt_v3(1.f / rd.x, 1.f / rd.y, 1.f / rd.z);
        for (uint32_t k = 0; k < n; k++) {
            const float4 lo = boxes[2 * k], hi = boxes[2 * k + 1];
            if (box_intersect(lo, hi, ro, inv) == RT_MISS_DST) reach &= ~(1u << __float_as_uint(lo.w));
        }
    }
    return reach;
}
__device__ __forceinline__ uint32_t reach_mask(const DevScene& sc, rt_vec3 ro, rt_vec3 rd) {
    return reach_mask_from(sc.maskBox, sc.reachCount, ro, rd, sc.cullOriginLimit);
}

// withMask false: the caller adds the mask later (k_render_fused does, outside shade_path, where registers are not scarce)
__device__ __forceinline__ float4 sphere_seed(const DevScene& sc, rt_vec3 ro, rt_vec3 rd, bool withMask = true) {
    float best = RT_MISS_DST;
    uint32_t obj = RT_HIT_NONE;
    for (uint32_t m = sc.sphereTestMask; m; m &= m - 1u) {  // ascending, as the shader's loop; repeats of an earlier sphere left out
        const uint32_t i = (uint32_t)__ffs((int)m) - 1u;
        if (i >= sc.sphereCount) break;
        SphereHit h = sphere_intersect(rt_uniform(sc.spheres)[i], ro, rd);
        if (h.didHit && h.dst < best) { best = h.dst; obj = RT_HIT_SPHERE | i; }
    }
    for (uint32_t i = 32; i < sc.sphereCount; i++) {  // beyond the mask (the reference has ten spheres)
        SphereHit h = sphere_intersect(rt_uniform(sc.spheres)[i], ro, rd);
        if (h.didHit && h.dst < best) { best = h.dst; obj = RT_HIT_SPHERE | i; }
    }
    return make_float4(best, __uint_as_float(obj), __uint_as_float(withMask ? reach_mask(sc, ro, rd) : 0xffffffffu), 0.f);
}

struct TriHit { bool didHit, frontFace; float dst, u, v, w; };

// raytrace.comp:227-247
__device__ __forceinline__ TriHit tri_intersect(rt_vec3 ro, rt_vec3 rd, rt_vec3 v0, rt_vec3 v1, rt_vec3 v2, bool frontOnly) {
    rt_vec3 v1v0 = rt_sub(v1, v0);
    rt_vec3 v2v0 = rt_sub(v2, v0);
    rt_vec3 rov0 = rt_sub(ro, v0);
    rt_vec3 n = rt_cross(v1v0, v2v0);
    rt_vec3 q = rt_cross(rov0, rd);
    float d0 = -rt_dot(rd, n);
    float d = 1.f / d0;
    TriHit h;
    h.dst = rt_dot(rov0, n) * d;
    h.u = rt_dot(v2v0, q) * d;
    h.v = -rt_dot(v1v0, q) * d;
    h.w = 1.f - h.u - h.v;
    h.frontFace = d0 >= 0.00000001f;
    h.didHit = h.dst >= 0.f && h.u >= 0.f && h.v >= 0.f && h.w >= 0.f && !(!h.frontFace && frontOnly);
    return h;
}

// raytrace.comp:263-274
__device__ __forceinline__ float box_intersect(float4 lo, float4 hi, rt_vec3 ro, rt_vec3 inv) {
    float ax = (lo.x - ro.x) * inv.x, ay = (lo.y - ro.y) * inv.y, az = (lo.z - ro.z) * inv.z;
    float bx = (hi.x - ro.x) * inv.x, by = (hi.y - ro.y) * inv.y, bz = (hi.z - ro.z) * inv.z;
    float tNear = rt_max(rt_max(rt_min(ax, bx), rt_min(ay, by)), rt_min(az, bz));
    float tFar = rt_min(rt_min(rt_max(ax, bx), rt_max(ay, by)), rt_max(az, bz));
    bool hit = tFar >= tNear && tFar > 0.f;
    return hit ? (tNear > 0.f ? tNear : 0.f) : RT_MISS_DST;
}

// Both children of a pair from the interleaved layout (DevScene::nodesPk). Same operations and
// operand order per component as two box_intersect calls; the subtractions and multiplications are
// written on 2-vectors so that they become v_pk_add_f32 / v_pk_mul_f32: 12 instructions instead of
// 24. On gfx950 that saves issue slots, not ALU time: a packed fp32 op takes as long as its two
// scalar halves (tools/micro/pk_rate.hip: 0.21 vs 0.38 wave-instructions per clock and SIMD).
typedef float rt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void box_intersect_pair(float4 q0, float4 q1, float4 q2, rt_f2 oxy, rt_f2 ixy, rt_f2 zOI, float& d1, float& d2) {
    const rt_f2 ozz = zOI.xx, izz = zOI.yy;
    const rt_f2 aL = (rt_f2{q0.x, q0.y} - oxy) * ixy, bL = (rt_f2{q0.z, q0.w} - oxy) * ixy;
    const rt_f2 aR = (rt_f2{q1.x, q1.y} - oxy) * ixy, bR = (rt_f2{q1.z, q1.w} - oxy) * ixy;
    const rt_f2 zL = (rt_f2{q2.x, q2.y} - ozz) * izz, zR = (rt_f2{q2.z, q2.w} - ozz) * izz;
    const float nL = rt_max(rt_max(rt_min(aL.x, bL.x), rt_min(aL.y, bL.y)), rt_min(zL.x, zL.y));
    const float fL = rt_min(rt_min(rt_max(aL.x, bL.x), rt_max(aL.y, bL.y)), rt_max(zL.x, zL.y));
    const float nR = rt_max(rt_max(rt_min(aR.x, bR.x), rt_min(aR.y, bR.y)), rt_min(zR.x, zR.y));
    const float fR = rt_min(rt_min(rt_max(aR.x, bR.x), rt_max(aR.y, bR.y)), rt_max(zR.x, zR.y));
    d1 = (fL >= nL && fL > 0.f) ? (nL > 0.f ? nL : 0.f) : RT_MISS_DST;
    d2 = (fR >= nR && fR > 0.f) ? (nR > 0.f ? nR : 0.f) : RT_MISS_DST;
}

// Distance of the nearest emissive primitive a ray hits at all, RT_MISS_DST if it hits none, for the two light queries of a
// diffuse bounce at once (same origin: NEE direction rdA, cosine-sample direction rdB): the emissive spheres and every listed
// triangle, each tested with the operations the traversal would use (sphere_intersect; the object-space ray of
// reconstruct_hit and tri_intersect — whose ray-independent part, the edges and their cross product, comes from
// DevScene::emitPre, and whose origin-dependent part, ro - v0 and its dot with the normal, is shared by the two rays), no
// bounding boxes involved. Whatever calculateIntersections finds for such a ray, an emissive closest hit has exactly one of
// the distances minimised here, so:
//   * tE == RT_MISS_DST: the closest hit cannot be emissive -> lightSamplePDF is 0 and the NEE term is 0 (raytrace.comp:
//     389-403,443-460) whatever the ray hits; the ray is not traced at all;
//   * a hit nearer than tE (a sphere, found by the creator, or a triangle, found by the traversal) is not emissive and
//     the closest hit is at least as near: same conclusion, the traversal stops there (trace_wave, leaf step).
// Both are statements about computed values only (the minimum of the very numbers the traversal compares), not about
// geometry, so they hold bit for bit.
__device__ __forceinline__ void emitter_tri_test(rt_vec3 rov0, float rn, rt_vec3 rd, rt_vec3 e1, rt_vec3 e2, rt_vec3 n, bool frontOnly, float& tE) {
    // tri_intersect (raytrace.comp:227-247) from `q` on; rov0 = ro - v0, rn = dot(rov0, n), e1 = v1 - v0, e2 = v2 - v0, n = cross(e1, e2)
    const rt_vec3 q = rt_cross(rov0, rd);
    const float d0 = -rt_dot(rd, n);
    const float d = 1.f / d0;
    const float dst = rn * d;
    const float u = rt_dot(e2, q) * d;
    const float v = -rt_dot(e1, q) * d;
    const float w = 1.f - u - v;
    const bool frontFace = d0 >= 0.00000001f;
    const bool didHit = dst >= 0.f && u >= 0.f && v >= 0.f && w >= 0.f && !(!frontFace && frontOnly);
    if (didHit && dst < tE) tE = dst;
}
__device__ __forceinline__ void emitter_min_t2(const DevScene& sc, rt_vec3 ro, rt_vec3 rdA, rt_vec3 rdB, float& tA, float& tB, uint32_t& tested) {
    tA = RT_MISS_DST; tB = RT_MISS_DST;
    for (uint32_t m = sc.emitSphereMask; m; m &= m - 1u) {
        const uint32_t i = (uint32_t)__ffs((int)m) - 1u;
        if (i >= sc.sphereCount) break;
        const float4 s = rt_uniform(sc.spheres)[i];
        const SphereHit hA = sphere_intersect(s, ro, rdA), hB = sphere_intersect(s, ro, rdB);
        if (hA.didHit && hA.dst < tA) tA = hA.dst;
        if (hB.didHit && hB.dst < tB) tB = hB.dst;
        tested += 2u;
    }
    uint32_t curObj = 0xffffffffu;
    rt_vec3 tro = ro, trdA = rdA, trdB = rdB;
    for (uint32_t k = 0; k < sc.emitCount; k++) {
        const uint32_t o = rt_uniform(sc.emitTris)[k].x;
        if (o >= sc.objectCount) break;  // a dispatch with fewer objects than were uploaded (sorted by object)
        if (o != curObj) {
            curObj = o;
            const float4* inv = rt_uniform(sc.objInv) + 3 * curObj;
            const float4 i0 = inv[0], i1 = inv[1], i2 = inv[2];
            trdA = xform_dir_rows(i0, i1, i2, rdA);
            trdB = xform_dir_rows(i0, i1, i2, rdB);
            tro = xform_point_rows(i0, i1, i2, ro);
        }
        const float4* pre = rt_uniform(sc.emitPre) + 4 * k;
        const float4 p0 = pre[0], p1 = pre[1], p2 = pre[2], p3 = pre[3];
        const rt_vec3 rov0 = rt_sub(tro, f4xyz(p0)), n = f4xyz(p3);
        const float rn = rt_dot(rov0, n);
        const bool frontOnly = __float_as_uint(p0.w) != 0u;
        emitter_tri_test(rov0, rn, trdA, f4xyz(p1), f4xyz(p2), n, frontOnly, tA);
        emitter_tri_test(rov0, rn, trdB, f4xyz(p1), f4xyz(p2), n, frontOnly, tB);
        tested += 2u;
    }
}

// ---------------------------------------------------------------- the metalness / alpha / bump maps (declared: rt_det_math.h)
// hit.uv (raytrace.comp:249-256) from a triangle hit's barycentrics, and the uv differences of the triangle's corners
struct HitUV { float u, v, du1, dv1, du2, dv2; };
__device__ __forceinline__ HitUV hit_uv(const DevScene& sc, uint32_t tri, float bu, float bv, float bw) {
    const float4 a = rt_global(sc.triUV)[2 * (size_t)tri], b = rt_global(sc.triUV)[2 * (size_t)tri + 1];  // {u0 v0 u1 v1} {u2 v2}
    HitUV r;
    r.u = (bw * a.x + bu * a.z) + bv * b.x;
    r.v = (bw * a.y + bu * a.w) + bv * b.y;
    if ((a.x == a.z && a.y == a.w) || (a.z == b.x && a.w == b.y) || (b.x == a.x && b.y == a.y)) { r.u = 0.5f; r.v = 0.5f; }
    r.du1 = a.z - a.x; r.dv1 = a.w - a.y; r.du2 = b.x - a.x; r.dv2 = b.y - a.y;
    return r;
}
// red byte of map `slot`'s texel at (u, 1 - v), or of the next one along the row (dx) / the column (dy), under the object's sampler
__device__ __forceinline__ uint32_t map_red8(const DevScene& sc, uint32_t slot, bool clampEdge, float u, float v, bool dx, bool dy) {
    const uint4 ti = rt_global(sc.texInfo)[slot];
    uint32_t x = rt_tex_index(u, ti.y, clampEdge), y = rt_tex_index(1.f - v, ti.z, clampEdge);
    if (dx) x = rt_tex_next(x, ti.y, clampEdge);
    if (dy) y = rt_tex_next(y, ti.z, clampEdge);
    return rt_global(sc.texels)[(size_t)ti.x + (size_t)y * ti.y + x] & 0xffu;
}
// alpha map: is this triangle hit cut out? (trace_wave<ALPHA>, leaf step; objAlpha: DevScene)
__device__ __forceinline__ bool alpha_cut(const DevScene& sc, uint32_t obj, uint32_t tri, const TriHit& h) {
    const uint32_t oa = rt_global(sc.objAlpha)[obj];
    if ((oa & 0xffu) >= sc.texCount) return false;   // (0xffffffff: no map)
    const HitUV q = hit_uv(sc, tri, h.u, h.v, h.w);
    return map_red8(sc, oa & 0xffu, (oa & 0x100u) != 0u, q.u, q.v, false, false) < RT_ALPHA_CUT_BYTE;
}

// DevScene::emitPre from the listed triangles' positions: tri_intersect's first lines (raytrace.comp:228-231), same operations
__global__ void k_emit_precompute(const float4* __restrict__ triPos, const uint2* __restrict__ emitTris, uint32_t n, float4* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const size_t t = emitTris[k].y;
    const float4 a = triPos[3 * t], b = triPos[3 * t + 1], c = triPos[3 * t + 2];
    const rt_vec3 v0 = f4xyz(a), e1 = rt_sub(f4xyz(b), v0), e2 = rt_sub(f4xyz(c), v0), nn = rt_cross(e1, e2);
    out[4 * k] = a;
    out[4 * k + 1] = mk4(e1, 0.f);
    out[4 * k + 2] = mk4(e2, 0.f);
    out[4 * k + 3] = mk4(nn, 0.f);
}

// ---------------------------------------------------------------- leaf references
#define RT_LEAF_CNT_SHIFT 28
#define RT_LEAF_IDX_MASK 0x0fffffffu
__device__ __forceinline__ void leaf_from_ref(const DevScene& sc, uint32_t ref, uint32_t& first, uint32_t& cnt) {
    cnt = (ref >> RT_LEAF_CNT_SHIFT) & 7u;
    first = ref & RT_LEAF_IDX_MASK;
    if (cnt == 0) {  // big leaf: `first` is the node
        cnt = __float_as_uint(sc.nodes[2 * (size_t)first + 1].w);
        first = sc.leafFirst[first];
    }
}

// ---------------------------------------------------------------- k_trace
// One lane per ray. Each lane walks the object list and each object's BVH on
// its own (no wave-wide object loop), with the same visit order as the shader:
// far child pushed, near child taken next. The traversal stack lives in LDS,
// lane-interleaved (entry e of lane l at word e*64+l: conflict-free), and
// holds only far siblings, so STACK = deepest leaf depth is enough.
struct TraceArgs {
    const uint32_t* queue;    // slot*4 + kind, or NULL for identity (slot = gid, kind = MAIN)
    const uint32_t* count;    // device-side ray count
    uint32_t* perRayBox;      // optional per-ray stats (rt_trace_rays), indexed by gid
    uint32_t* perRayTri;
    DevCounters* counters;
    // the queue in three pieces (k_shade): `*count` main rays at queue[0 ..], `*countAux` rays of the second kind at
    // queue[auxOffset ..], `*countAux2` of the third at queue[2 * auxOffset ..]: a wave's rays are of one kind (coherent), and the
    // long ones are dealt first, so that a launch's tail is made of the short ones. NULL: one piece.
    const uint32_t* countAux;
    const uint32_t* countAux2;
    uint32_t auxOffset;
};
// queue position of the i-th ray of a launch
__device__ __forceinline__ uint32_t queue_pos(uint32_t i, uint32_t n0, uint32_t n1, uint32_t off) {
    return i < n0 ? i : (i < n0 + n1 ? i - n0 + off : i - n0 - n1 + 2u * off);
}

template <int STACK>
__global__ __launch_bounds__(RT_BLOCK) void k_trace(DevScene sc, PathState ps, TraceArgs ta) {
    __shared__ uint32_t s_stack[(RT_BLOCK / RT_WAVE) * STACK * RT_WAVE];
    const uint32_t nMain = *ta.count, nAux = ta.countAux ? *ta.countAux : 0u;
    const uint32_t n = nMain + nAux + (ta.countAux ? *ta.countAux2 : 0u);
    const uint32_t gid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t qpos = queue_pos(gid, nMain, nAux, ta.auxOffset);
    const bool live = gid < n && !(ta.queue && ta.queue[qpos] == 0xffffffffu);  // RT_QUEUE_HOLE
    if (__ballot(live) == 0ull) return;

    uint32_t* stack = s_stack + (threadIdx.x / RT_WAVE) * STACK * RT_WAVE + (threadIdx.x & (RT_WAVE - 1));

    uint32_t nBox = 0, nTri = 0;
    uint32_t didHit = 0;
    uint32_t slot = 0, kind = RAY_MAIN;
    if (live) {
        uint32_t id = ta.queue ? ta.queue[qpos] : (gid << 2);
        slot = id >> 2;
        kind = id & 3u;
        rt_vec3 ro, rd;
        if (kind == RAY_MAIN) { ro = f4xyz(ps.rayO()[slot]); rd = f4xyz(ps.rayD()[slot]); }
        else { ro = f4xyz(ps.auxO()[slot]); rd = f4xyz(kind == RAY_NEE ? ps.auxDL()[slot] : ps.auxDC()[slot]); }

        float best = RT_MISS_DST;
        uint32_t bestObj = RT_HIT_NONE, bestTri = 0;
        const float earlyT = ta.queue ? ps.hit(kind)[slot].w : 0.f;  // light queries: distance of the nearest emissive primitive

        for (uint32_t i = 0; i < sc.sphereCount; i++) {
            SphereHit h = sphere_intersect(sc.spheres[i], ro, rd);
            if (h.didHit && h.dst < best) { best = h.dst; bestObj = RT_HIT_SPHERE | i; }
        }

        uint32_t obj = 0, sp = 0;
        uint32_t curIdx = 0, curCnt = 0;
        bool have = false;
        rt_vec3 tro = ro, trd = rd, inv = rd;
        for (;;) {
            if (!have) {
                if (sp > 0) {
                    uint32_t ref = stack[(--sp) * RT_WAVE];
                    if (ref & RT_LEAF_BIT) {
                        leaf_from_ref(sc, ref, curIdx, curCnt);
                    } else {
                        curIdx = ref;
                        curCnt = 0;
                    }
                } else {
                    if (obj >= sc.objectCount) break;
                    float4 r0 = sc.objInv[3 * obj], r1 = sc.objInv[3 * obj + 1], r2 = sc.objInv[3 * obj + 2];
                    uint4 meta = sc.objMeta[obj];
                    trd = xform_dir_rows(r0, r1, r2, rd);
                    tro = xform_point_rows(r0, r1, r2, ro);
                    inv = rt_v3(1.f / trd.x, 1.f / trd.y, 1.f / trd.z);
                    curIdx = meta.x;
                    curCnt = meta.y;
                    if (curCnt) leaf_from_ref(sc, meta.x, curIdx, curCnt);
                    obj++;
                }
                have = true;
            }
            if (curCnt != 0) {
                nTri += curCnt;
                bool closer = false;
                for (uint32_t j = curIdx; j < curIdx + curCnt; j++) {
                    float4 a = sc.triPos[3 * j], b = sc.triPos[3 * j + 1], c = sc.triPos[3 * j + 2];
                    TriHit h = tri_intersect(tro, trd, f4xyz(a), f4xyz(b), f4xyz(c), __float_as_uint(a.w) != 0u);
                    if (h.didHit && h.dst < best) { best = h.dst; bestObj = obj - 1; bestTri = j; closer = true; }
                }
                have = false;
                // a light query that found something nearer than its nearest emissive primitive is answered (see trace_wave)
                if (closer && best < earlyT) { best = RT_MISS_DST; bestObj = RT_HIT_NONE; bestTri = 0; break; }
            } else {
                const float4* pr = sc.nodes + 2 * (size_t)curIdx;
                float4 lo1 = pr[0], hi1 = pr[1], lo2 = pr[2], hi2 = pr[3];
                float d1 = box_intersect(lo1, hi1, tro, inv);
                float d2 = box_intersect(lo2, hi2, tro, inv);
                nBox += 2;
                bool nearA = d1 <= d2;
                float dNear = nearA ? d1 : d2;
                float dFar = nearA ? d2 : d1;
                uint32_t nW = __float_as_uint(nearA ? lo1.w : lo2.w), nCnt = __float_as_uint(nearA ? hi1.w : hi2.w);
                uint32_t fW = __float_as_uint(nearA ? lo2.w : lo1.w);
                if (dFar < best) {
                    stack[sp * RT_WAVE] = fW;
                    sp++;
                }
                if (dNear < best) {
                    curIdx = nW; curCnt = nCnt;
                    if (nCnt) leaf_from_ref(sc, nW, curIdx, curCnt);
                } else have = false;
            }
        }

        ps.hit(kind)[slot] = make_float4(best, __uint_as_float(bestObj), __uint_as_float(bestTri), 0.f);
        if (kind == RAY_MAIN && ps.statBox()) { ps.statBox()[slot] += nBox; ps.statTri()[slot] += nTri; }
        if (ta.perRayBox) { ta.perRayBox[gid] = nBox; ta.perRayTri[gid] = nTri; }
        didHit = bestObj != RT_HIT_NONE;
    }

    // one atomic per wave and counter
    unsigned long long wb = wave_sum_u64(nBox), wt = wave_sum_u64(nTri);
    uint32_t wr = wave_sum_u32(live ? 1u : 0u), wh = wave_sum_u32(didHit);
    if (lane_id() == 0) {
        atomicAdd(&ta.counters->boxTests, wb);
        atomicAdd(&ta.counters->triTests, wt);
        atomicAdd(&ta.counters->raysTraced, (unsigned long long)wr);
        atomicAdd(&ta.counters->raysHit, (unsigned long long)wh);
    }
}

// ---------------------------------------------------------------- k_trace_pw (persistent waves)
// Same per-ray visit order and arithmetic as k_trace, organised for SIMD
// efficiency on 64-wide waves (k_trace measured 21 % active lanes per VALU
// instruction on Sponza):
//   * persistent waves pull rays from the queue with one atomic per refill and
//     re-arm idle lanes when at least `refill` lanes are idle, so a wave does
//     not drain down to its slowest ray;
//   * a lane's whole traversal state is one word `cur`: a child-pair index
//     (interior), a leaf reference, or a marker (needs a node / general
//     object setup pending / idle). Each round the wave votes (__ballot +
//     popcount, weighted) for interior, leaf or setup work and executes only
//     that step — one triangle per leaf step — and a common tail hands every
//     lane that ran out of work its next node: from its LDS stack, else the
//     root of the next object, else the ray is stored and the lane idles;
//   * leaf references (first triangle, count <= 7) are ready-made in the node
//     words, so pushes and pops move one word and need no memory read;
//   * objects whose transform is exactly the identity are entered in the tail
//     with register moves only: they reuse the world-space ray and 1/dir
//     (bit-identical to multiplying by the identity when no component is
//     zero, negative-zero or non-finite; other rays take the general path).
//     The next object's metadata is fetched one object ahead.
#define RT_META_LDS 128u           // objects whose {root word, flags} a work-group keeps in LDS (1 KB)
#define RT_HOT_PAIRS 192u          // child pairs of the meshes' top levels that come first in the device numbering; k_trace_pw<HOT> keeps the first HOT of them in LDS
#define RT_CUR_IDLE 0xffffffffu   // no ray
#define RT_CUR_NEED 0xfffffffeu   // wants its next node (tail)
#define RT_CUR_SETUP 0xfffffffdu  // entering object `obj`, which has a general transform
#define RT_CUR_WORLD 0xfffffffcu  // back to the world-space ray after a general-transform object
#define RT_CUR_INIT 0xfffffffbu   // new ray: load it, sphere tests, 1/dir
#define RT_CUR_DONE 0xfffffffau   // ray finished: store the result
#define RT_CUR_LEAF_MAX 0xfffffff0u  // leaf references are below the markers

struct TracePwArgs {
    const uint32_t* queue;
    const uint32_t* count;
    uint32_t* head;           // work counter, zeroed before the launch
    uint32_t refill;          // re-arm idle lanes when at least this many are idle
    uint32_t chunk;           // most queue entries a wave reserves per atomic (guided: fewer near the end)
    uint32_t wSetup, wLeaf;   // vote weights in eighths (interior = 8)
    uint32_t fastLanes;       // go straight to the interior step when at least this many lanes are at interior nodes
    uint32_t fastShare;       // ... but at most this many sixteenths of the lanes that hold a ray (0 = off)
    uint32_t* perRayBox;      // PIX only
    uint32_t* perRayTri;
    DevCounters* counters;
    unsigned long long* phaseStats;  // STATS only: [8] rounds and active lanes per phase
    unsigned long long* waveTimes;   // STATS only: wall_clock64() at start and end of every wave (2 per wave)
    uint32_t* overflow;       // OVF only: stack entries beyond STACK, (maxDepth - STACK) x resident lanes
    const uint32_t* countAux; // the queue's second and third piece (TraceArgs)
    const uint32_t* countAux2;
    uint32_t auxOffset;
};

// STATS: wait for every outstanding load, then read the clock (splits a step's time into "data on its way" and the rest)
#define RT_STAMP_AFTER_LOADS(acc, t0) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); (acc) += clock64() - (t0); } while (0)

// Counters a wave accumulates while it traces (reduced once per kernel).
struct WaveTotals {
    uint32_t totBox = 0, totTri = 0, totRays = 0, totHits = 0;
    uint32_t totSkipBox = 0;  // of totBox: box tests charged for objects the ray was taken past without entering them (CULL), not executed
    uint32_t dbgRounds[4] = {0, 0, 0, 0}, dbgLanes[4] = {0, 0, 0, 0};  // refill, setup, interior, leaf (STATS)
    unsigned long long dbgCycles[4] = {0, 0, 0, 0};                    // shader clocks spent in rounds of each kind (STATS)
    unsigned long long dbgLoad[4] = {0, 0, 0, 0};                      // ... of which: from the step's first load instruction to the arrival of its data (STATS)
    uint32_t dbgWait[4] = {0, 0, 0, 0};  // STATS: [3] trips of the set-up step's skipping loop; [0..2] lanes that sat out interior rounds at a leaf / in set-up states / without a ray
};

// OVF: the BVH is deeper than STACK; entries beyond the LDS part live in a global overflow buffer
// (rarely touched: the stack only holds far siblings), so deep trees keep the occupancy of shallow ones.
// LOCAL: the rays come from the wave's own list in LDS (k_render_fused) instead of the global queue.
// CULL: objects a ray cannot reach are skipped at the cost the reference has for them (the ray's object mask from its creator,
// and the set-up step's loop over general-transform objects); compiled in only for scenes with at least two general-transform
// objects: its mere presence costs an identity-only scene like Sponza 3 % (k_trace_pw) to 10 % (k_render_fused).
//
// Instruction issue, scalar and vector alike, is what bounds this loop (measured: ~110 VALU +
// ~100 SALU + 16 branches per round kept both issue ports ~70 % busy while the L1 and L2 idled;
// serving half of the node fetches from LDS changed nothing). So the hot round is written with as
// little control flow as possible: one ballot decides "enough lanes are at an interior node" and
// goes straight to the interior step; pushes and pops are unconditional LDS accesses with
// predicated pointer updates; the full vote, the refill and the leaf / setup steps live on a slow
// path that is only entered when fewer than `fastLanes` lanes are at interior nodes.
// ROOMY: the kernel is built for five work-groups per CU (96 registers per lane): a light query's tE rides in a register there
// (-2 % on the bench frame). With the 80 registers of six work-groups per CU, and in the fused kernel, one more live register
// means one more spill: measured there, it loses (Cornell + bunny +3 %, C5 at 4K +1 %).
// ALPHA: triangle hits are looked up in their object's alpha map before they count (alpha_cut; k_trace_pw_alpha only).
template <int STACK, bool OVF, bool PIX, bool STATS, bool LOCAL, bool CULL, int HOT = 0, bool ROOMY = false, bool ALPHA = false>
__device__ __forceinline__ void trace_wave(const DevScene& sc, const PathState& ps, const TracePwArgs& ta, uint32_t* stack, uint32_t* ovf,
                                           size_t ovfStride, const uint32_t* localList, uint32_t n, WaveTotals& wt, const uint2* metaLds,
                                           const float4* hotLds = nullptr) {
    uint32_t cur = RT_CUR_IDLE;
    uint32_t id = 0, qidx = 0;
    // ray in the current object's space and 1/dir (written by the setup step only); x and y are kept as
    // 2-vectors so that they sit in aligned register pairs for the packed slab test
    rt_vec3 trd = rt_v3(0, 0, 0);
    rt_f2 troXY = {0.f, 0.f}, invXY = {0.f, 0.f}, zOI = {0.f, 0.f};  // zOI = (origin.z, 1/dir.z)
    bool plain = false;    // ray eligible for the identity fast path
    bool atWorld = false;  // tro/trd/inv currently are the world-space ray
    float best = RT_MISS_DST;
    uint32_t bestObj = RT_HIT_NONE, bestTri = 0;
    uint32_t obj = 0, sp = 0;
    uint32_t nxW = 0, nxFlags = 0;   // objMeta of object `obj`, fetched ahead of its use
    uint32_t rayBox = 0, rayTri = 0;  // PIX: this ray's counters
    // wave-uniform refill state. LOCAL: the wave's own ray list is the whole 'queue', already reserved.
    bool exhausted = LOCAL;               // the queue has no entries left to reserve
    uint32_t resBase = 0, resCount = LOCAL ? n : 0u;  // reserved queue entries not yet dealt out
    uint32_t nextChunk = LOCAL ? 0u : min(ta.chunk, max(16u, n / (2u * gridDim.x * (RT_BLOCK / RT_WAVE))));

    uint32_t thr = ta.fastLanes;
    const uint32_t nMain = LOCAL ? 0u : *ta.count, nAux = (LOCAL || !ta.countAux) ? 0u : *ta.countAux;   // the queue's pieces (TraceArgs::countAux)

    uint32_t reach = 0xffffffffu;  // objects (of the mask's window) the ray has to enter, from its creator (sphere_seed)
    float earlyT = 0.f;            // light queries: tE, the distance of the nearest emissive primitive on the ray (0: any other ray)

    // Called right after `obj` moved past the object that is being entered: obj - 1 is the object under traversal. The
    // objects after it that the ray's mask rules out are jumped over here (two box tests each, nothing else), and how many
    // were is kept in nxFlags[15:8], so that the object under traversal is still obj - 1 - skipped (cur_object()).
    auto fetch_next_meta = [&]() {
        uint32_t skip = 0;
        if (CULL && obj - sc.maskBase < 32u) {
            const uint32_t w = obj - sc.maskBase;  // position in the mask's window
            const uint32_t m = reach >> w;
            skip = m ? (uint32_t)__ffs((int)m) - 1u : 32u - w;
            if (skip) {
                const uint2 c0 = sc.objSkipCost[w], c1 = sc.objSkipCost[w + skip];
                if (PIX) { rayBox += c1.x - c0.x; rayTri += c1.y - c0.y; } else { wt.totBox += c1.x - c0.x; wt.totTri += c1.y - c0.y; }
                wt.totSkipBox += c1.x - c0.x;
            }
            obj += skip;
        }
        uint32_t fl = 0;
        if (obj < sc.objectCount) {
            // {root word, flags} of the next object: from the work-group's LDS copy (metaLds: the first RT_META_LDS objects) — one
            // vector load per object and ray less on the memory pipeline that bounds the traversal (Sponza: 26 of ~400 per ray)
            if (obj < RT_META_LDS) {
                const uint2 m = metaLds[obj];
                nxW = m.x; fl = m.y;
            } else {
                const uint4 m = sc.objMeta[obj];
                nxW = m.x; fl = m.w & 0xffu;
            }
        }
        nxFlags = CULL ? (fl | (skip << 8)) : fl;
    };
    auto cur_object = [&]() { return CULL ? obj - 1u - (nxFlags >> 8) : obj - 1u; };

    for (;;) {
        const unsigned long long tRound = STATS ? clock64() : 0ull;
        int roundKind = 2;
        const unsigned long long mI = __ballot((int32_t)cur >= 0);
        uint32_t nI = __popcll(mI);
        bool runI = nI >= thr;
        if (!runI) {
            // ================= slow path: refill, full vote, leaf and setup steps =================
            // refill: hand queue entries to idle lanes (the ray itself is loaded by the setup step). The wave
            // reserves up to `chunk` entries per atomic (one hot counter saturates near 90 atomics/us) and
            // deals them out locally; reservations shrink near the end of the queue (guided scheduling).
            const unsigned long long mIdle = __ballot(cur == RT_CUR_IDLE);
            const uint32_t nIdle = __popcll(mIdle);
            if (nIdle == RT_WAVE && exhausted && resCount == 0) break;
            // the share of the live lanes that skips the vote, not a fixed count: while a list drains the wave keeps its fast path
            if (ta.fastShare) thr = min(ta.fastLanes, max(4u, ((RT_WAVE - nIdle) * ta.fastShare) >> 4));
            if (nIdle >= ta.refill && (resCount || !exhausted)) {
                if (resCount == 0) {
                    uint32_t base = 0;
                    if (lane_id() == 0) base = atomicAdd(ta.head, nextChunk);
                    base = __shfl(base, 0, RT_WAVE);
                    resBase = base;
                    resCount = base < n ? min(nextChunk, n - base) : 0u;
                    if (base + nextChunk >= n) exhausted = true;
                    const uint32_t left = base + nextChunk < n ? n - base - nextChunk : 0u;
                    nextChunk = min(ta.chunk, max(16u, left / (2u * gridDim.x * (RT_BLOCK / RT_WAVE))));
                }
                const uint32_t take = min(nIdle, resCount);
                if (STATS) { wt.dbgRounds[0]++; wt.dbgLanes[0] += take; }
                if (cur == RT_CUR_IDLE) {
                    const uint32_t rk = lanes_below(mIdle);
                    if (rk < take) {
                        qidx = resBase + rk;
                        id = LOCAL ? localList[qidx] : (ta.queue ? ta.queue[queue_pos(qidx, nMain, nAux, ta.auxOffset)] : (qidx << 2));
                        cur = (!LOCAL && id == 0xffffffffu) ? RT_CUR_IDLE : RT_CUR_INIT;  // RT_QUEUE_HOLE: no ray behind this entry (k_raygen)
                    }
                }
                if (STATS) RT_STAMP_AFTER_LOADS(wt.dbgLoad[0], tRound);
                resBase += take;
                resCount -= take;
            }

            // vote (weighted: a step that feeds lanes back into the interior state may run with fewer lanes)
            const uint32_t nL = __popcll(__ballot((int32_t)cur < 0 && cur < RT_CUR_LEAF_MAX));
            const uint32_t nS = __popcll(__ballot(cur - RT_CUR_INIT <= RT_CUR_SETUP - RT_CUR_INIT));
            const uint32_t scS = nS * ta.wSetup, scI = nI * 8u, scL = nL * ta.wLeaf;
            runI = nI && scI >= scL && scI >= scS;
            const bool runL = !runI && nL && scL >= scS;
            const bool runS = !runI && !runL && nS;
            if (STATS) {
                if (runL) { wt.dbgRounds[3]++; wt.dbgLanes[3] += nL; }
                else if (runS) { wt.dbgRounds[1]++; wt.dbgLanes[1] += nS; }
                roundKind = runI ? 2 : runL ? 3 : runS ? 1 : 0;
            }

            const unsigned long long tLeaf = STATS ? clock64() : 0ull;
            if (runL) {
                // ---------------- leaf step: up to two triangles (all of them for a leaf with > 7)
                if ((int32_t)cur < 0 && cur < RT_CUR_LEAF_MAX) {
                    const uint32_t cnt = (cur >> RT_LEAF_CNT_SHIFT) & 7u;
                    uint32_t j = cur & RT_LEAF_IDX_MASK, jEnd;
                    if (cnt == 0) {  // j is the node
                        jEnd = __float_as_uint(sc.nodes[2 * (size_t)j + 1].w);
                        j = sc.leafFirst[j];
                        jEnd += j;
                        cur = RT_CUR_NEED;
                    } else {
                        const uint32_t take = min(cnt, 2u);  // two triangles per step: their loads overlap; 1, 3 and 'all' measured slower
                        jEnd = j + take;
                        cur = (cnt > take) ? (cur + take - (take << RT_LEAF_CNT_SHIFT)) : RT_CUR_NEED;
                    }
                    if (PIX) rayTri += jEnd - j; else wt.totTri += jEnd - j;
                    bool closer = false;  // this step found a nearer hit
                    if (cnt != 0u) {
                        // one or two triangles: both fetched before either is tested
                        const uint32_t j1 = jEnd - 1u;
                        // Whole 16-byte vectors from the global address space: three aligned loads per triangle. (Through float4, whose
                        // padding words are never read, the compiler fetched a triangle's 44 bytes as 8 + 16 at offset 4 + 8 + 12: four
                        // instructions, one of them straddling two 16-byte slots.)
                        const RT_AS_GLOBAL rt_f4v* tp0 = (const RT_AS_GLOBAL rt_f4v*)sc.triPos + 3 * (size_t)j;
                        const RT_AS_GLOBAL rt_f4v* tp1 = (const RT_AS_GLOBAL rt_f4v*)sc.triPos + 3 * (size_t)j1;
                        const rt_f4v a0 = tp0[0], b0 = tp0[1], c0 = tp0[2];
                        const rt_f4v a1 = tp1[0], b1 = tp1[1], c1 = tp1[2];
                        if (STATS) RT_STAMP_AFTER_LOADS(wt.dbgLoad[3], tLeaf);
                        const rt_vec3 o = rt_v3(troXY.x, troXY.y, zOI.x);
                        const TriHit h0 = tri_intersect(o, trd, rt_v3(a0.x, a0.y, a0.z), rt_v3(b0.x, b0.y, b0.z), rt_v3(c0.x, c0.y, c0.z), __float_as_uint(a0.w) != 0u);
                        if (h0.didHit && h0.dst < best && !(ALPHA && alpha_cut(sc, cur_object(), j, h0))) { best = h0.dst; bestObj = cur_object(); bestTri = j; closer = true; }
                        if (j1 != j) {
                            const TriHit h1 = tri_intersect(o, trd, rt_v3(a1.x, a1.y, a1.z), rt_v3(b1.x, b1.y, b1.z), rt_v3(c1.x, c1.y, c1.z), __float_as_uint(a1.w) != 0u);
                            if (h1.didHit && h1.dst < best && !(ALPHA && alpha_cut(sc, cur_object(), j1, h1))) { best = h1.dst; bestObj = cur_object(); bestTri = j1; closer = true; }
                        }
                    } else {
                        for (; j < jEnd; j++) {
                            const float4 a = sc.triPos[3 * (size_t)j], b = sc.triPos[3 * (size_t)j + 1], c = sc.triPos[3 * (size_t)j + 2];
                            const TriHit h = tri_intersect(rt_v3(troXY.x, troXY.y, zOI.x), trd, f4xyz(a), f4xyz(b), f4xyz(c), __float_as_uint(a.w) != 0u);
                            if (h.didHit && h.dst < best && !(ALPHA && alpha_cut(sc, cur_object(), j, h))) { best = h.dst; bestObj = cur_object(); bestTri = j; closer = true; }
                        }
                    }
                    // Light queries carry tE, the distance of the nearest emissive primitive they hit at all (emitter_min_t2; 0 for
                    // every other ray). A hit nearer than tE is not emissive and the closest hit is no farther: the query's answer
                    // is "not emissive" whatever else the ray meets, so it ends here and reports no hit, which is what shade_path
                    // reads as "not emissive". tE is re-read from the ray's hit record on the rare step that finds a hit rather
                    // than held in a register through the loop. The leaf counts in full, as the shader counts it (:310).
                    if (closer && best < ((RT_TE_REG && ROOMY) ? earlyT : ps.hit(id & 3u)[id >> 2].w)) {
                        if ((int32_t)cur < 0 && cur < RT_CUR_LEAF_MAX) {  // triangles of this leaf not yet stepped through
                            const uint32_t rest = (cur >> RT_LEAF_CNT_SHIFT) & 7u;
                            if (PIX) rayTri += rest; else wt.totTri += rest;
                        }
                        if (CULL) {
                            // the objects after this one that the ray's mask rules out were charged when this one was entered
                            // (fetch_next_meta); the query ends before the reference's loop would have reached them
                            const uint32_t ahead = nxFlags >> 8;
                            if (ahead) {
                                const uint2 c0 = sc.objSkipCost[obj - sc.maskBase - ahead], c1 = sc.objSkipCost[obj - sc.maskBase];
                                if (PIX) { rayBox -= c1.x - c0.x; rayTri -= c1.y - c0.y; } else { wt.totBox -= c1.x - c0.x; wt.totTri -= c1.y - c0.y; }
                                wt.totSkipBox -= c1.x - c0.x;
                            }
                        }
                        best = RT_MISS_DST; bestObj = RT_HIT_NONE; bestTri = 0;
                        sp = 0; obj = sc.objectCount;
                        cur = RT_CUR_NEED;
                    }
                }
            } else if (runS) {
                // ---------------- setup step: the only place that writes tro/trd/inv
                //   INIT : new ray (world space, sphere tests)      -> NEED, or straight into object 0 if that needs a transform
                //   WORLD: world-space ray again                     -> NEED
                //   SETUP: into object `obj` with a general matrix   -> its root
                if (cur - RT_CUR_INIT <= RT_CUR_SETUP - RT_CUR_INIT) {
                    const uint32_t slot = id >> 2, kind = id & 3u;
                    rt_vec3 wo, wd;
                    if (kind == RAY_MAIN) { wo = f4xyz(ps.rayO()[slot]); wd = f4xyz(ps.rayD()[slot]); }
                    else { wo = f4xyz(ps.auxO()[slot]); wd = f4xyz(kind == RAY_NEE ? ps.auxDL()[slot] : ps.auxDC()[slot]); }
                    float4 seedPre = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (STATS) {  // the ray and its seed, all on their way at once, then the stamp
                        if (cur == RT_CUR_INIT) seedPre = ps.hit(kind)[slot];
                        RT_STAMP_AFTER_LOADS(wt.dbgLoad[1], tLeaf);
                    }
                    if (cur == RT_CUR_INIT) {
                        // the ray's creator already ran the sphere loop (sphere_seed)
                        const float4 seed = STATS ? seedPre : ps.hit(kind)[slot];
                        best = seed.x; bestObj = __float_as_uint(seed.y); bestTri = 0;
                        plain = ray_is_plain(wo, wd);
                        reach = __float_as_uint(seed.z);
                        if (RT_TE_REG && ROOMY) earlyT = seed.w;
                        obj = 0; sp = 0;
                        if (PIX) { rayBox = 0; rayTri = 0; }
                        wt.totRays++;
                        fetch_next_meta();
                    }
                    // a new ray whose first object has a general transform goes into it in this very step
                    bool general = cur == RT_CUR_SETUP || (cur == RT_CUR_INIT && sc.objectCount > 0u && !((nxFlags & 1u) && plain));
                    if (CULL && general && plain && origin_within(wo, sc.cullOriginLimit)) {
                        // General-transform objects the ray cannot reach before its closest hit so far are not entered: in
                        // the reference such an object costs the two box tests on its root's children (its root leaf's triangle
                        // tests) and nothing else; that is what is counted. (objBox: padded world box; plain: finite ray.)
                        const rt_vec3 iw = rt_v3(1.f / wd.x, 1.f / wd.y, 1.f / wd.z);
                        // two objects per trip, their boxes fetched together (lo.w = objMeta flags, hi.w = root triangle count)
                        while (obj < sc.objectCount) {
                            if (STATS) wt.dbgWait[2 + 1]++;   // (dbgWait[3]: trips of the skipping loop, summed over lanes)
                            if (RT_OBJTREE && sc.objTreeLevels) {
                                // the biggest aligned block of placed objects that starts here and is out of reach: jumped as a whole
                                bool block = false;
                                for (uint32_t k = obj ? min(sc.objTreeLevels, (uint32_t)__ffs((int)obj) - 1u) : sc.objTreeLevels; k >= 1u; k--) {
                                    const float4* t = sc.objTree + 2 * (size_t)(sc.objTreeOff[k] + (obj >> k));
                                    const float4 lo = t[0], hi = t[1];
                                    if (lo.w == 0.f || box_intersect(lo, hi, wo, iw) < best) continue;  // not a block of placed objects, or reachable: its first half next
                                    const uint2 c0 = sc.objCost[obj], c1 = sc.objCost[obj + (1u << k)];
                                    if (PIX) { rayBox += c1.x - c0.x; rayTri += c1.y - c0.y; } else { wt.totBox += c1.x - c0.x; wt.totTri += c1.y - c0.y; }
                                    wt.totSkipBox += c1.x - c0.x;
                                    obj += 1u << k;
                                    block = true;
                                    break;
                                }
                                if (block) continue;
                            }
                            const uint32_t i1 = min(obj + 1u, sc.objectCount - 1u);
                            const float4 a0 = sc.objBox[2 * obj], b0 = sc.objBox[2 * obj + 1], a1 = sc.objBox[2 * i1], b1 = sc.objBox[2 * i1 + 1];
                            if ((__float_as_uint(a0.w) & 3u) != 2u) break;  // identity, or no usable box
                            if (box_intersect(a0, b0, wo, iw) < best) break;
                            const uint32_t c0 = __float_as_uint(b0.w);
                            if (c0 == 0u) { if (PIX) rayBox += 2; else wt.totBox += 2; wt.totSkipBox += 2; }
                            else { if (PIX) rayTri += c0; else wt.totTri += c0; }
                            obj++;
                            if (obj >= sc.objectCount) break;
                            if (RT_OBJTREE && sc.objTreeLevels && !(obj & 1u)) continue;  // an even index again: the blocks that start here come first
                            if ((__float_as_uint(a1.w) & 3u) != 2u) break;
                            if (box_intersect(a1, b1, wo, iw) < best) break;
                            const uint32_t c1 = __float_as_uint(b1.w);
                            if (c1 == 0u) { if (PIX) rayBox += 2; else wt.totBox += 2; wt.totSkipBox += 2; }
                            else { if (PIX) rayTri += c1; else wt.totTri += c1; }
                            obj++;
                        }
                        fetch_next_meta();
                        general = obj < sc.objectCount && !(nxFlags & 1u);
                    }
                    if (general) {
                        const float4 r0 = sc.objInv[3 * obj], r1 = sc.objInv[3 * obj + 1], r2 = sc.objInv[3 * obj + 2];
                        trd = xform_dir_rows(r0, r1, r2, wd);
                        wo = xform_point_rows(r0, r1, r2, wo);
                    } else {
                        trd = wd;
                    }
                    troXY = rt_f2{wo.x, wo.y};
                    invXY = rt_f2{1.f / trd.x, 1.f / trd.y};
                    zOI = rt_f2{wo.z, 1.f / trd.z};
                    atWorld = !general;
                    if (general) {
                        cur = nxW;
                        obj++;
                        fetch_next_meta();
                    } else {
                        cur = RT_CUR_NEED;
                    }
                }
            }
        }

        const unsigned long long tStep = STATS ? clock64() : 0ull;
        if (runI) {
            // ================= interior step: both children of the pair `cur` =================
            if (STATS) {
                wt.dbgRounds[2]++; wt.dbgLanes[2] += nI;
                wt.dbgWait[0] += __popcll(__ballot((int32_t)cur < 0 && cur < RT_CUR_LEAF_MAX));
                wt.dbgWait[1] += __popcll(__ballot(cur >= RT_CUR_LEAF_MAX && cur != RT_CUR_IDLE));
                wt.dbgWait[2] += __popcll(__ballot(cur == RT_CUR_IDLE));
            }
            if ((int32_t)cur >= 0) {
                float4 q0, q1, q2;
                float2 lk;
                // (Left as written, the compiler merges the two branches into one set of loads through a generic pointer, FLAT
                // instructions that reach LDS through the vector memory pipeline. Naming the address spaces — ds_read_b128 on one side,
                // global_load on the other — measured 1.2 % SLOWER: both branches write the same registers, so the LDS reads are made
                // to wait for the other lanes' global loads before they are issued. profiles/README.md, "r03 small results".)
                if (HOT && cur < min(sc.hotNodes, 2u * (uint32_t)HOT)) {
                    // a child pair of a mesh's top levels: from the work-group's LDS copy
                    const float4* ph = hotLds + 2 * cur;
                    q0 = ph[0]; q1 = ph[1]; q2 = ph[2];
                    lk = *(const float2*)(ph + 3);
                } else {
                    const float4* pr = sc.nodesPk + 2 * (size_t)cur;
                    q0 = pr[0]; q1 = pr[1]; q2 = pr[2];
                    lk = *(const float2*)(pr + 3);
                }
                if (STATS) RT_STAMP_AFTER_LOADS(wt.dbgLoad[2], tStep);
                float d1, d2;
                box_intersect_pair(q0, q1, q2, troXY, invXY, zOI, d1, d2);
                if (PIX) rayBox += 2; else wt.totBox += 2;
                const bool nearA = d1 <= d2;
                const float dNear = nearA ? d1 : d2, dFar = nearA ? d2 : d1;
                const uint32_t nW = __float_as_uint(nearA ? lk.x : lk.y), fW = __float_as_uint(nearA ? lk.y : lk.x);
                // ready-made word (pair index or leaf reference); kept only if it qualifies
                if (OVF) {
                    stack[min(sp, (uint32_t)STACK) * RT_WAVE] = fW;
                    if (sp >= (uint32_t)STACK && dFar < best) ovf[(sp - STACK) * ovfStride] = fW;
                } else {
                    stack[sp * RT_WAVE] = fW;
                }
                sp += (dFar < best) ? 1u : 0u;
                cur = (dNear < best) ? nW : RT_CUR_NEED;
            }
        }

        // ================= tail: next node for every lane that ran out of work (predicated) =================
        {
            const bool need = cur == RT_CUR_NEED;
            const bool has = sp > 0;
            uint32_t top;
            if (OVF) {
                top = stack[min((has ? sp : 1u) - 1u, (uint32_t)STACK) * RT_WAVE];
                if (need && sp > (uint32_t)STACK) top = ((const RT_AS_GLOBAL uint32_t*)ovf)[(sp - 1u - STACK) * ovfStride];  // (named global: no FLAT load)
            } else {
                top = stack[((has ? sp : 1u) - 1u) * RT_WAVE];
            }
            const bool objLeft = obj < sc.objectCount;
            const bool ident = (nxFlags & 1u) && plain;  // identity transform: register moves only
            const uint32_t whenEmpty = objLeft ? (ident ? (atWorld ? nxW : RT_CUR_WORLD) : RT_CUR_SETUP) : RT_CUR_DONE;
            const bool enter = need && !has && objLeft && ident && atWorld;
            cur = need ? (has ? top : whenEmpty) : cur;
            sp -= (need && has) ? 1u : 0u;
            if (enter) {
                obj++;
                fetch_next_meta();
            }
            if (cur == RT_CUR_DONE) {
                const uint32_t slot = id >> 2, kind = id & 3u;
                ps.hit(kind)[slot] = make_float4(best, __uint_as_float(bestObj), __uint_as_float(bestTri), 0.f);
                if (PIX && kind == RAY_MAIN) { ps.statBox()[slot] += rayBox; ps.statTri()[slot] += rayTri; }
                if (PIX) {
                    if (ta.perRayBox) { ta.perRayBox[qidx] = rayBox; ta.perRayTri[qidx] = rayTri; }
                    wt.totBox += rayBox; wt.totTri += rayTri;
                }
                wt.totHits += (bestObj != RT_HIT_NONE) ? 1u : 0u;
                cur = RT_CUR_IDLE;
            }
        }
        if (STATS) wt.dbgCycles[roundKind] += clock64() - tRound;
    }

}

// the work-group's LDS copy of the first RT_META_LDS objects' {root word, flags} (fetch_next_meta)
__device__ __forceinline__ void fill_meta_lds(const DevScene& sc, uint2* s_meta) {
    if (threadIdx.x < RT_META_LDS) {
        uint2 m = make_uint2(0u, 0u);
        if (threadIdx.x < sc.objectCount) { const uint4 q = sc.objMeta[threadIdx.x]; m = make_uint2(q.x, q.w & 0xffu); }
        s_meta[threadIdx.x] = m;
    }
    __syncthreads();
}

// HOT > 0: the first HOT child pairs of the device numbering (the meshes' top levels, DevScene::hotNodes) are served from a copy
// in LDS (64 B each); BLOCKS = work-groups per CU the kernel is built for (what the LDS of stack + table leaves room for)
template <int STACK, bool OVF, bool PIX, bool STATS, bool CULL, int HOT, bool ROOMY, bool ALPHA>
__device__ __forceinline__ void trace_pw_block(const DevScene& sc, const PathState& ps, const TracePwArgs& ta) {
    __shared__ uint32_t s_stack[(RT_BLOCK / RT_WAVE) * (STACK + 1) * RT_WAVE];  // +1: pushes are unconditional
    __shared__ uint2 s_meta[RT_META_LDS];
    __shared__ float4 s_hot[HOT ? 4 * HOT : 1];
    if (HOT) for (uint32_t k = threadIdx.x; k < 2u * min(sc.hotNodes, 2u * (uint32_t)HOT); k += RT_BLOCK) s_hot[k] = sc.nodesPk[k];
    fill_meta_lds(sc, s_meta);
    uint32_t* stack = s_stack + (threadIdx.x / RT_WAVE) * (STACK + 1) * RT_WAVE + (threadIdx.x & (RT_WAVE - 1));
    // overflow entries of this lane: index k at ovf[k * ovfStride]
    uint32_t* ovf = OVF ? ta.overflow + (size_t)blockIdx.x * RT_BLOCK + threadIdx.x : nullptr;
    const size_t ovfStride = (size_t)gridDim.x * RT_BLOCK;
    WaveTotals wt;
    const unsigned long long tStart = STATS ? wall_clock64() : 0ull;
    trace_wave<STACK, OVF, PIX, STATS, false, CULL, HOT, ROOMY, ALPHA>(sc, ps, ta, stack, ovf, ovfStride, nullptr, *ta.count + (ta.countAux ? *ta.countAux + *ta.countAux2 : 0u), wt, s_meta, s_hot);

    const uint32_t skipTrips = STATS ? wave_sum_u32(wt.dbgWait[3]) : 0u;
    if (STATS && lane_id() == 0) {
        const size_t w = (size_t)blockIdx.x * (RT_BLOCK / RT_WAVE) + threadIdx.x / RT_WAVE;
        if (ta.waveTimes) {
            ta.waveTimes[2 * w] = tStart;
            ta.waveTimes[2 * w + 1] = wall_clock64();
        }
        for (int k = 0; k < 4; k++) {
            atomicAdd(&ta.phaseStats[k], (unsigned long long)wt.dbgRounds[k]);
            atomicAdd(&ta.phaseStats[4 + k], (unsigned long long)wt.dbgLanes[k]);
            atomicAdd(&ta.phaseStats[8 + k], wt.dbgCycles[k]);
            atomicAdd(&ta.phaseStats[16 + k], wt.dbgLoad[k]);
            if (k < 3) atomicAdd(&ta.phaseStats[12 + k], (unsigned long long)wt.dbgWait[k]);
            else atomicAdd(&ta.phaseStats[20], (unsigned long long)skipTrips);
        }
    }
    unsigned long long wb = wave_sum_u64(wt.totBox), wtri = wave_sum_u64(wt.totTri);
    uint32_t wr = wave_sum_u32(wt.totRays), wh = wave_sum_u32(wt.totHits);
    const unsigned long long wskip = CULL ? wave_sum_u64(wt.totSkipBox) : 0ull;
    if (lane_id() == 0 && wr) {
        if (CULL && wskip) atomicAdd(&ta.counters->skippedBoxTests, wskip);
        atomicAdd(&ta.counters->boxTests, wb);
        atomicAdd(&ta.counters->triTests, wtri);
        atomicAdd(&ta.counters->raysTraced, (unsigned long long)wr);
        atomicAdd(&ta.counters->raysHit, (unsigned long long)wh);
    }
}

template <int STACK, bool OVF, bool PIX, bool STATS, bool CULL, int HOT = 0, int BLOCKS = 6>
__global__ __launch_bounds__(RT_BLOCK, BLOCKS) void k_trace_pw(DevScene sc, PathState ps, TracePwArgs ta) {
    trace_pw_block<STACK, OVF, PIX, STATS, CULL, HOT, BLOCKS == 5, false>(sc, ps, ta);
}
// The traversal of a scene that binds an alpha map (DevScene::mapFlags & RT_MAP_ALPHA): one configuration for every such scene —
// 24 stack entries in LDS with the overflow buffer behind them, object culling compiled in, no top-level table, four work-groups
// per CU (128 registers: the texel look-up of a hit candidate sits in the leaf step)
template <bool PIX>
__global__ __launch_bounds__(RT_BLOCK, 4) void k_trace_pw_alpha(DevScene sc, PathState ps, TracePwArgs ta) {
    trace_pw_block<24, true, PIX, false, true, 0, true, true>(sc, ps, ta);
}

// ---------------------------------------------------------------- hit reconstruction
// The trace kernel stores only (dst, object, triangle). Everything else of the
// shader's HitInfo is recomputed here with the same operations the traversal
// used, so the values are the ones calculateIntersections would have stored
// (raytrace.comp:316-321 for triangles, :217-221 for spheres).
struct FullHit {
    rt_vec3 hitPoint, normal;
    uint32_t materialIndex;
    bool frontFace;
    float u, v;   // hit.uv of a triangle hit (reconstruct_hit<MAPS> only)
};

// hit.uv (raytrace.comp:249-256) and the albedo texel at (u, 1 - v); the declared texture semantics of include/rt_amd.h
// (the triangle test is run again for its u, v, w rather than carried through shade_path in three registers: textured hits are rare)
__device__ __forceinline__ rt_vec3 albedo_texel(const DevScene& sc, uint32_t slot, uint32_t tri, uint32_t obj, rt_vec3 ro, rt_vec3 rd) {
    float bu, bv, bw;
    {
        const float4 i0 = rt_global(sc.objInv)[3 * obj], i1 = rt_global(sc.objInv)[3 * obj + 1], i2 = rt_global(sc.objInv)[3 * obj + 2];
        const float4 p0 = rt_global(sc.triPos)[3 * (size_t)tri], p1 = rt_global(sc.triPos)[3 * (size_t)tri + 1], p2 = rt_global(sc.triPos)[3 * (size_t)tri + 2];
        const TriHit h = tri_intersect(xform_point_rows(i0, i1, i2, ro), xform_dir_rows(i0, i1, i2, rd), f4xyz(p0), f4xyz(p1), f4xyz(p2), __float_as_uint(p0.w) != 0u);
        bu = h.u; bv = h.v; bw = h.w;
    }
    const float4 a = rt_global(sc.triUV)[2 * (size_t)tri], b = rt_global(sc.triUV)[2 * (size_t)tri + 1];  // {u0 v0 u1 v1} {u2 v2}
    float u = (bw * a.x + bu * a.z) + bv * b.x, v = (bw * a.y + bu * a.w) + bv * b.y;
    const bool e01 = a.x == a.z && a.y == a.w, e12 = a.z == b.x && a.w == b.y, e20 = b.x == a.x && b.y == a.y;
    if (e01 || e12 || e20) { u = 0.5f; v = 0.5f; }
    const uint4 ti = rt_global(sc.texInfo)[slot];
    const bool clampEdge = ((rt_global(sc.objMeta)[obj].w >> 16) & 0xffffu) == 1u;
    const uint32_t x = rt_tex_index(u, ti.y, clampEdge), y = rt_tex_index(1.f - v, ti.z, clampEdge);
    const uint32_t t = rt_global(sc.texels)[(size_t)ti.x + (size_t)y * ti.y + x];
    return rt_v3(rt_srgb8_to_linear(t & 0xffu), rt_srgb8_to_linear((t >> 8) & 0xffu), rt_srgb8_to_linear((t >> 16) & 0xffu));
}

// MAPS (k_shade_maps): the object's bump map tilts the interpolated normal (rt_bump_normal), and the hit's uv is handed on
template <bool MAPS = false>
__device__ __forceinline__ FullHit reconstruct_hit(const DevScene& sc, rt_vec3 ro, rt_vec3 rd, uint32_t obj, uint32_t tri) {
    FullHit f;
    f.u = 0.f; f.v = 0.f;
    if (obj & RT_HIT_SPHERE) {
        uint32_t i = obj & ~RT_HIT_SPHERE;
        float4 s = rt_global(sc.spheres)[i];
        SphereHit h = sphere_intersect(s, ro, rd);
        f.hitPoint = rt_add(ro, rt_scale(rd, h.dst));
        f.normal = rt_scale(rt_normalize(rt_sub(f.hitPoint, f4xyz(s))), h.frontFace ? 1.f : -1.f);
        f.materialIndex = rt_global(sc.sphereMat)[i];
        f.frontFace = h.frontFace;
        return f;
    }
    // An object whose matrix and inverse are both exactly the identity (objMeta flag bit 3), hit by a plain ray: multiplying by
    // the identity returns the operand bit for bit while every component is finite and none is -0 ((1 * x + 0 * y) + 0 * z: a -0
    // would pick up the sign of 0 * y, an infinity would meet 0 * inf) — the condition the traversal's identity fast path tests on
    // the ray (ray_is_plain, trace_wave) and that is tested here on the interpolated normal and the object-space hit point as
    // well. Then neither matrix is fetched nor applied (Sponza: all 25 material groups; raytrace.comp:316-321).
    const uint4 meta = rt_global(sc.objMeta)[obj];
    const bool ident = RT_SHADE_IDENT && (meta.w & 8u) && ray_is_plain(ro, rd);
    rt_vec3 trd = rd, tro = ro;
    if (!ident) {
        const float4* inv = rt_global(sc.objInv) + 3 * obj;
        const float4 i0 = inv[0], i1 = inv[1], i2 = inv[2];
        trd = xform_dir_rows(i0, i1, i2, rd);
        tro = xform_point_rows(i0, i1, i2, ro);
    }
    const float4* tp = rt_global(sc.triPos) + 3 * (size_t)tri;
    float4 a = tp[0], b = tp[1], c = tp[2];
    TriHit h = tri_intersect(tro, trd, f4xyz(a), f4xyz(b), f4xyz(c), __float_as_uint(a.w) != 0u);
    const float4* tn = rt_global(sc.triNrm) + 3 * (size_t)tri;
    rt_vec3 n0 = f4xyz(tn[0]), n1 = f4xyz(tn[1]), n2 = f4xyz(tn[2]);
    rt_vec3 ni = rt_add(rt_add(rt_scale(n0, h.w), rt_scale(n1, h.u)), rt_scale(n2, h.v));
    if (MAPS) {
        const HitUV q = hit_uv(sc, tri, h.u, h.v, h.w);
        f.u = q.u; f.v = q.v;
        const uint32_t bumpSlot = __float_as_uint(rt_global(sc.mats)[3 * meta.z + 2].w);   // bumpIndex; 0xffffffff (-1) = none
        if (bumpSlot < sc.texCount) {
            const bool clampEdge = ((meta.w >> 16) & 0xffffu) == 1u;
            const float h0 = rt_srgb8_to_linear(map_red8(sc, bumpSlot, clampEdge, q.u, q.v, false, false));
            const float hx = rt_srgb8_to_linear(map_red8(sc, bumpSlot, clampEdge, q.u, q.v, true, false)) - h0;
            const float hy = rt_srgb8_to_linear(map_red8(sc, bumpSlot, clampEdge, q.u, q.v, false, true)) - h0;
            ni = rt_bump_normal(ni, rt_sub(f4xyz(b), f4xyz(a)), rt_sub(f4xyz(c), f4xyz(a)), q.du1, q.dv1, q.du2, q.dv2, hx, hy);
        }
    }
    ni = rt_scale(ni, h.frontFace ? 1.f : -1.f);
    rt_vec3 op = rt_add(tro, rt_scale(trd, h.dst));
    if (ident && vec_is_plain(ni) && vec_is_plain(op)) {
        f.normal = rt_normalize(ni);
        f.hitPoint = op;
    } else {
        const float4* fwd = rt_global(sc.objFwd) + 3 * obj;
        const float4 m0 = fwd[0], m1 = fwd[1], m2 = fwd[2];
        f.normal = rt_normalize(xform_dir_rows(m0, m1, m2, ni));
        f.hitPoint = xform_point_rows(m0, m1, m2, op);
    }
    f.materialIndex = meta.z;
    f.frontFace = h.frontFace;
    return f;
}

__device__ __forceinline__ uint32_t hit_material(const DevScene& sc, uint32_t obj) {
    return (obj & RT_HIT_SPHERE) ? rt_global(sc.sphereMat)[obj & ~RT_HIT_SPHERE] : rt_global(sc.objMeta)[obj].z;
}

// ---------------------------------------------------------------- shading pieces
// raytrace.comp:177-181
__device__ __forceinline__ float schlick(float cosine, float ri) {
    float r0 = (1.f - ri) / (1.f + ri);
    r0 = r0 * r0;
    return r0 + (1.f - r0) * rt_pow(1.f - cosine, 5.f);
}

// raytrace.comp:356-365
__device__ __forceinline__ rt_vec3 environment_light(const EnvironmentData& env, rt_vec3 d) {
    if (!(env.lightDir[3] == 1.f)) return rt_v3(0.f, 0.f, 0.f);
    float skyT = rt_pow(rt_smoothstep(0.f, 0.4f, -d.y), 0.35f);
    rt_vec3 sky = rt_v3(rt_mix(env.horizonColor[0], env.zenithColor[0], skyT), rt_mix(env.horizonColor[1], env.zenithColor[1], skyT),
                        rt_mix(env.horizonColor[2], env.zenithColor[2], skyT));
    rt_vec3 negL = rt_v3(-env.lightDir[0], -env.lightDir[1], -env.lightDir[2]);
    float sun = rt_pow(rt_max(0.f, rt_dot(d, negL)), env.horizonColor[3]) * env.zenithColor[3];
    float g2s = rt_smoothstep(-0.01f, 0.f, -d.y);
    float sunMask = g2s >= 1.f ? 1.f : 0.f;
    float add = sun * sunMask;
    return rt_v3(rt_mix(env.groundColor[0], sky.x, g2s) + add, rt_mix(env.groundColor[1], sky.y, g2s) + add,
                 rt_mix(env.groundColor[2], sky.z, g2s) + add);
}

// raytrace.comp:389-403 given the probe ray's closest hit
__device__ __forceinline__ float light_sample_pdf(const DevScene& sc, float t, uint32_t obj, rt_vec3 dir) {
    if (obj == RT_HIT_NONE) return 0.f;
    uint32_t m = hit_material(sc, obj);
    if (rt_global(sc.mats)[3 * m + 1].w == 0.f) return 0.f;
    float sq = t * t;
    float cosTheta = rt_dot(rt_v3(0.f, -1.f, 0.f), dir);
    return sq / (cosTheta * 0.4444444f);
}

// raytrace.comp:547-556 for one pixel
__device__ __forceinline__ rt_vec3 primary_dir(const FrameParams& fp, uint32_t gx, uint32_t gy) {
    float u = (float)gx / (float)fp.width;
    float v = (float)gy / (float)fp.height;
    rt_vec3 point = rt_add(rt_v3(fp.bottomLeft[0], fp.bottomLeft[1], fp.bottomLeft[2]),
                           rt_v3(fp.planeWidth * u, fp.planeHeight * v, 0.f));
    rt_vec3 dir = rt_normalize(point);
    return rt_xform_point(fp.camRot, dir);
}

// Slot -> pixel. Slots walk the tile's rows in 8x8 pixel blocks (one wave = one
// block, like the reference's 8x8 workgroups, raytrace.comp:4), so the rays of a
// wave start out coherent; a remainder of fewer than 8 rows, or a width that is
// not a multiple of 8, falls back to row-major order. `k` is the local row.
// several frames per launch: which frame a slot belongs to, its slot in the tile, and back
__device__ __forceinline__ uint32_t slot_frame(const FrameParams& fp, uint32_t slot) { return (slot >> 6) % fp.nFrames; }
__device__ __forceinline__ uint32_t slot_in_tile(const FrameParams& fp, uint32_t slot) { return ((slot >> 6) / fp.nFrames << 6) | (slot & 63u); }
__device__ __forceinline__ uint32_t frame_slot(const FrameParams& fp, uint32_t frame, uint32_t tileSlot) {
    return ((tileSlot >> 6) * fp.nFrames + frame) * 64u + (tileSlot & 63u);
}

__device__ __forceinline__ void slot_to_pixel(const FrameParams& fp, uint32_t slot, uint32_t& gx, uint32_t& gy, uint32_t& k) {
    if (fp.nFrames > 1u) slot = slot_in_tile(fp, slot);  // the same pixel in every frame of the launch
    const uint32_t tiledSlots = fp.tiled ? (fp.nRows & ~7u) * fp.width : 0u;
    if (slot < tiledSlots) {
        const uint32_t tile = slot >> 6, in = slot & 63u;
        const uint32_t tilesPerRow = fp.width >> 3;
        const uint32_t tyb = tile / tilesPerRow, txb = tile - tyb * tilesPerRow;
        gx = (txb << 3) + (in & 7u);
        k = (tyb << 3) + (in >> 3);
    } else {
        k = slot / fp.width;
        gx = slot - k * fp.width;
    }
    gy = fp.row0 + k * fp.rowStride;
}

// ---------------------------------------------------------------- k_raygen
// raytrace.comp:547-564 for the pixel of `slot`: camera ray, RNG seed, path reset, sphere seed
__device__ __forceinline__ void init_path(const DevScene& sc, const PathState& ps, const FrameParams& fp, uint32_t slot) {
    uint32_t gx, gy;
    uint32_t krow;
    slot_to_pixel(fp, slot, gx, gy, krow);
    ps.rayO()[slot] = make_float4(fp.camPos[0], fp.camPos[1], fp.camPos[2], 1.f);                          // misWeight = 1
    const rt_vec3 pd = primary_dir(fp, gx, gy);
    uint32_t startingSeed = fp.startingSeed;
    if (fp.nFrames > 1u) {  // frame f of the launch: uint(random(frameCount + f) * 23892183), as the host computes it for one frame
        uint32_t lol = fp.frameCount + slot_frame(fp, slot);
        startingSeed = (uint32_t)(rt_random(&lol) * 23892183.f);
    }
    ps.rayD()[slot] = mk4u(pd, gy * fp.width + gx + startingSeed);                                      // RNG seed (:564)
    ps.camDir()[slot] = mk4(pd, 0.f);
    ps.hit(RAY_MAIN)[slot] = sphere_seed(sc, rt_v3(fp.camPos[0], fp.camPos[1], fp.camPos[2]), pd);
    ps.att()[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(0u));                                   // j = 0
    ps.total()[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));                                 // sample 0
    ps.accum()[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    ps.statBox()[slot] = 0;
    ps.statTri()[slot] = 0;
}

#define RT_QUEUE_HOLE 0xffffffffu   // queue entry without a path (multi-frame dispatch: the padding of a tile that is not a multiple of 64 slots)
// slots [slotBegin, slotEnd) of the dispatch: one part of the multi-kernel pipeline (render_impl); the part's queues start with
// its slots in order (position = slot - slotBegin)
__global__ __launch_bounds__(RT_BLOCK) void k_raygen(DevScene sc, PathState ps, uint32_t* activeOut, uint32_t* raysOut, FrameParams fp,
                                                     uint32_t slotBegin, uint32_t slotEnd) {
    const uint32_t pos = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t slot = slotBegin + pos;
    // several frames per dispatch (rt_render_frames): the slots run over {64 tile slots} x {frames}, see FrameParams::nFrames
    if (slot >= slotEnd) return;
    if (fp.nFrames > 1u && slot_in_tile(fp, slot) >= fp.nPixels) {
        activeOut[pos] = RT_QUEUE_HOLE;
        raysOut[pos] = RT_QUEUE_HOLE;
        return;
    }
    init_path(sc, ps, fp, slot);
    activeOut[pos] = slot;
    raysOut[pos] = slot << 2;
}

// ---------------------------------------------------------------- k_shade
struct ShadeArgs {
    const uint32_t* inActive;
    const uint32_t* inCount;
    uint32_t* outActive;
    uint32_t* outRays;
    uint32_t* outActiveCount;
    uint32_t* outRayCount;
    DevCounters* counters;
    uint32_t* outAuxCount;    // the queue's second and third piece, outRays[auxOffset ..] and outRays[2 * auxOffset ..] (TraceArgs::countAux)
    uint32_t* outAuxCount2;
    uint32_t auxOffset;
};

// One path, one segment: trace()'s loop body (raytrace.comp:495-534) with diffuseBRDF split around the
// probe rays, plus main()'s sample loop (:571-573). Reads the hit records of the path's rays, writes
// its next rays. Outputs: alive (the path, or the pixel's next sample, goes on), auxMask (bit 0: the NEE ray, bit 1: the cosine probe
// of this diffuse bounce have to be traced — a light query that emitter_min_t2 answers is not; bit 2: the main ray needs no
// traversal, its hit record is already there: the kept camera hit), refRays (the shader's
// calculateIntersections calls for this segment), nPaths (1 if a sample finished), emitTests (primitives emitter_min_t2 tested).
// MAPS (k_shade_maps): the scene binds a metalness or a bump map
template <bool MAPS = false>
__device__ __forceinline__ void shade_path(const DevScene& sc, const PathState& ps, const FrameParams& fp, uint32_t slot, bool& alive,
                                           uint32_t& auxMask, uint32_t& refRays, uint32_t& nPaths, uint32_t& emitTests, bool withMask = true) {
    bool wantAux = false;  // a diffuse bounce whose MIS the next segment finishes (raytrace.comp:443-460)
    auxMask = 0;
    rt_vec3 auxOrigin = rt_v3(0, 0, 0), auxL = auxOrigin, auxC = auxOrigin;  // probe rays of this bounce (diffuse only)
    rt_vec3 auxAlbedo = auxOrigin;                                           // ... and what the next segment needs to finish its MIS
    float auxNdotL = 0.f, auxCosPdfL = 0.f, auxCosPdfC = 0.f;
    const float4 sO = ps.rayO()[slot], sD = ps.rayD()[slot], sA = ps.att()[slot], sT = ps.total()[slot];
    const float4 hM = ps.hit(RAY_MAIN)[slot];
    rt_vec3 ro = f4xyz(sO), rd = f4xyz(sD);
    rt_vec3 att = f4xyz(sA), total = f4xyz(sT);
    float misW = sO.w;
    uint32_t state = __float_as_uint(sD.w);
    uint32_t jraw = __float_as_uint(sA.w);
    uint32_t samplesDone = __float_as_uint(sT.w);
    uint32_t j = jraw & 0x0fffffffu;
    const bool pending = (jraw >> 31) != 0u;
    // directLight as the last segment left it (raytrace.comp:487,469,480,460): -1 after a specular bounce, 0 at the start of a
    // sample; after a diffuse bounce the block below recomputes it before it is read
    rt_vec3 direct = (jraw & 0x40000000u) ? rt_v3(-1.f, -1.f, -1.f) : rt_v3(0.f, 0.f, 0.f);
    bool specular = false;  // this segment's bounce is a mirror or dielectric one

    const uint32_t obj = __float_as_uint(hM.y);
    const uint32_t hitTriIdx = __float_as_uint(hM.z);
    refRays = 1;  // this segment's calculateIntersections (raytrace.comp:496)
    if (fp.camReuse && j == 0u && samplesDone == 0u) ps.camHit()[slot] = hM;  // the camera ray's hit, for the pixel's later samples

    bool done = false;       // this sample's trace() returned
    bool zeroed = false;     // ... through the NaN/negative early-out (:505)

    if (obj != RT_HIT_NONE) {
        if (pending) {
            // finish diffuseBRDF of the previous bounce (:443-460); its three
            // scene queries were the NEE ray (once for :443 and :447) and the
            // cosine probe (:453)
            // (a light query its creator answered left no record: it is the record of a ray that hit nothing)
            float4 hL = make_float4(RT_MISS_DST, __uint_as_float(RT_HIT_NONE), 0.f, 0.f), hC = hL;
            if (!(jraw & 0x20000000u)) hL = ps.hit(RAY_NEE)[slot];
            if (!(jraw & 0x10000000u)) hC = ps.hit(RAY_PROBE)[slot];
            const float4 aL = ps.auxDL()[slot], aC = ps.auxDC()[slot];
            float tL = hL.x, tC = hC.x;
            uint32_t oL = __float_as_uint(hL.y), oC = __float_as_uint(hC.y);
            rt_vec3 dL = f4xyz(aL), dC = f4xyz(aC);
            uint32_t lm = (oL == RT_HIT_NONE) ? 0u : hit_material(sc, oL);
            float4 lmE = rt_global(sc.mats)[3 * lm + 1];
            float realLightPDF = light_sample_pdf(sc, tL, oL, dL);
            float cosinePDF = aL.w;
            float misWeight1 = realLightPDF * realLightPDF / (realLightPDF * realLightPDF + cosinePDF * cosinePDF);
            if (rt_isnan(misWeight1)) misWeight1 = 0.f;
            float lightPDF = light_sample_pdf(sc, tC, oC, dC);
            float realCosinePDF = aC.w;
            float misWeight2 = realCosinePDF * realCosinePDF / (lightPDF * lightPDF + realCosinePDF * realCosinePDF);
            if (rt_isnan(misWeight2)) misWeight2 = 0.f;
            const float4 pA = ps.pendAlbedo()[slot];   // {albedo, max(0, dot(n, lightSample))}
            rt_vec3 albedo = f4xyz(pA);
            rt_vec3 dl = rt_scale(rt_v3(lmE.x, lmE.y, lmE.z), lmE.w);
            float k = (realLightPDF == 0.f) ? 0.f : misWeight1 / realLightPDF;
            rt_vec3 f = rt_scale(rt_scale(rt_scale(albedo, RT_INV_PI), pA.w), k);
            direct = rt_mul(dl, f);
            misW = misWeight2;
        }

        FullHit hit = reconstruct_hit<MAPS>(sc, ro, rd, obj, hitTriIdx);
        const float4* mp = rt_global(sc.mats) + 3 * hit.materialIndex;
        float4 mA = mp[0], mE = mp[1], mI = mp[2];
        if (MAPS) {  // metalness map: the texel's decoded red replaces the material's reflectance (triangle hits)
            const uint32_t metalSlot = __float_as_uint(mI.z);   // metalnessIndex; 0xffffffff (-1) = none
            if (metalSlot < sc.texCount && !(obj & RT_HIT_SPHERE))
                mA.w = rt_srgb8_to_linear(map_red8(sc, metalSlot, ((rt_global(sc.objMeta)[obj].w >> 16) & 0xffffu) == 1u, hit.u, hit.v, false, false));
        }

        // 0-1 NEE (:501-505)
        rt_vec3 emission = rt_scale(rt_v3(mE.x, mE.y, mE.z), mE.w);
        emission = rt_v3(emission.x / misW, emission.y / misW, emission.z / misW);
        rt_vec3 finalLight = direct.x == -1.f ? emission : direct;
        total = rt_add(total, rt_mul(finalLight, att));
        if (j == 0) total = rt_add(total, emission);
        if (rt_isnan(total.x) || rt_isnan(total.y) || rt_isnan(total.z) || total.x < 0.f || total.y < 0.f || total.z < 0.f) {
            done = true;
            zeroed = true;
        } else {
            rt_vec3 sampledDir, radiance;
            float originSign = 1.f;
            bool diffuse = false;
            if (mA.w != 0.f) {  // reflectance != 0: mirror (:466-469)
                sampledDir = rt_reflect(rd, hit.normal);
                radiance = rt_v3(1.f, 1.f, 1.f);
                specular = true;   // directLight = -1 (:469)
                misW = 1.f;
            } else if (mI.x != -1.f) {  // dielectric (:471-481)
                float ior = !hit.frontFace ? mI.x : 1.f / mI.x;
                float cosine = rt_dot(rt_neg(rd), hit.normal);
                float sine = rt_sqrt(1.f - cosine * cosine);
                bool solution = (ior * sine) > 1.f;
                if (!solution) solution = schlick(cosine, ior) > rt_random(&state);
                sampledDir = solution ? rt_reflect(rd, hit.normal) : rt_refract(rd, hit.normal, ior);
                originSign = solution ? 1.f : rt_sign(rt_dot(hit.normal, rd));
                radiance = rt_v3(1.f, 1.f, 1.f);
                specular = true;   // directLight = -1 (:480)
                misW = 1.f;
            } else {  // diffuse + NEE/MIS (:430-464), first half
                diffuse = true;
                refRays += 3;
                rt_vec3 albedo = rt_v3(mA.x, mA.y, mA.z);
                const uint32_t texSlot = __float_as_uint(mI.y);   // albedoIndex; 0xffffffff (-1) = none
                if (texSlot < sc.texCount && !(obj & RT_HIT_SPHERE)) albedo = rt_mul(albedo, albedo_texel(sc, texSlot, hitTriIdx, obj, ro, rd));
                rt_vec3 origin = rt_add(hit.hitPoint, rt_scale(hit.normal, 0.01f));
                // lightSampleDir (:368-387)
                float lx = rt_random(&state);
                float lz = rt_random(&state);
                rt_vec3 lp = rt_v3(rt_mix(-0.33333f, 0.33333f, lx), -1.5f, rt_mix(-0.33333f, 0.33333f, lz));
                rt_vec3 lightSample = rt_normalize(rt_sub(lp, origin));
                // cosineHemisphereDir (:405-424)
                float r1 = rt_random(&state);
                float r2 = rt_random(&state);
                float phi = (2.f * RT_PI) * r1;
                float sqrtR2 = rt_sqrt(r2);
                float sn, cs;
                rt_sincos(phi, &sn, &cs);
                float x = cs * sqrtR2, y = sn * sqrtR2, z = rt_sqrt(1.f - r2);
                rt_vec3 axis = rt_abs(rt_dot(hit.normal, rt_v3(1.f, 0.f, 0.f))) < 1.f ? rt_v3(1.f, 0.f, 0.f) : rt_v3(0.f, 0.f, 1.f);
                rt_vec3 tt = rt_normalize(rt_cross(hit.normal, axis));
                rt_vec3 bb = rt_cross(hit.normal, tt);
                rt_vec3 cosineSample = rt_add(rt_add(rt_scale(tt, x), rt_scale(bb, y)), rt_scale(hit.normal, z));

                float realCosinePDF = rt_max(0.f, rt_dot(cosineSample, hit.normal) * RT_INV_PI);
                float nDotC = rt_dot(hit.normal, cosineSample);
                radiance = rt_scale(rt_scale(albedo, RT_INV_PI), nDotC);
                radiance = rt_v3(radiance.x / realCosinePDF, radiance.y / realCosinePDF, radiance.z / realCosinePDF);
                sampledDir = cosineSample;

                // (the four records of the unfinished MIS are written at the end, once it is known that the path goes on and that
                // the next segment needs them)
                auxOrigin = origin; auxL = lightSample; auxC = cosineSample; auxAlbedo = albedo;
                auxNdotL = rt_max(0.f, rt_dot(hit.normal, lightSample));
                auxCosPdfL = rt_max(0.f, rt_dot(lightSample, hit.normal) * RT_INV_PI);
                auxCosPdfC = realCosinePDF;
            }
            att = rt_mul(att, radiance);

            // russian roulette (:520-524)
            float rrProb = rt_max(rt_max(att.x, att.y), att.z);
            rrProb = rt_min(rrProb, 0.95f);
            rrProb = j <= 5 ? 1.f : rrProb;
            if (rt_random(&state) > rrProb) {
                done = true;
            } else {
                float invP = 1.f / rrProb;
                att = rt_scale(att, invP);
                ro = rt_add(hit.hitPoint, rt_scale(rt_scale(hit.normal, originSign), 0.00001f));
                rd = sampledDir;
                j++;
                if (j > fp.bounceLimit) done = true;  // the for loop ends (:495)
                else wantAux = diffuse;
            }
        }
    } else {
        // miss (:532-533)
        total = rt_add(total, rt_mul(att, environment_light(fp.env, rd)));
        done = true;
    }

    if (done) {
        nPaths = 1;
        float4 acc = ps.accum()[slot];
        if (zeroed) total = rt_v3(0.f, 0.f, 0.f);
        const rt_vec3 sum = rt_add(f4xyz(acc), total);
        ps.accum()[slot] = mk4(sum, 0.f);
        samplesDone++;
        if (samplesDone < fp.samples) {
            // next sample of this pixel: same primary ray (kept by init_path rather than derived from the slot again: the
            // whole wave pays for this branch whenever one lane finishes a sample), RNG state runs on (:571-573)
            ro = rt_v3(fp.camPos[0], fp.camPos[1], fp.camPos[2]);
            rd = f4xyz(ps.camDir()[slot]);
            att = rt_v3(1.f, 1.f, 1.f);
            total = rt_v3(0.f, 0.f, 0.f);
            specular = false;  // directLight = 0 (:487)
            misW = 1.f;
            j = 0;
            alive = true;
            wantAux = false;
        }
    } else {
        alive = true;
    }

    if (alive) {
        if (done && fp.camReuse) {
            // a new sample of the same pixel: the same camera ray, whose hit is known — no traversal this round (bit 2)
            ps.hit(RAY_MAIN)[slot] = ps.camHit()[slot];
            auxMask |= 4u;
        } else {
            ps.hit(RAY_MAIN)[slot] = sphere_seed(sc, ro, rd, withMask);
        }
        if (wantAux) {
            float4 sL = sphere_seed(sc, auxOrigin, auxL, withMask), sC = sphere_seed(sc, auxOrigin, auxC, withMask);
            auxMask = 3u;
            if (sc.emitMode) {
                // Both are light queries: only "is the closest hit emissive, and at what distance" is read from them. With
                // tE the nearest emissive primitive on the ray: none at all, or a sphere nearer than it, answers "not
                // emissive" here and now (the record of a ray that hit nothing); otherwise the traversal carries tE in the
                // record's w and stops at the first hit nearer than it.
                float tL, tC;
                emitter_min_t2(sc, auxOrigin, auxL, auxC, tL, tC, emitTests);
                if (!(tL < RT_MISS_DST) || sL.x < tL) { sL.x = RT_MISS_DST; sL.y = __uint_as_float(RT_HIT_NONE); auxMask &= ~1u; }
                else sL.w = tL;
                if (!(tC < RT_MISS_DST) || sC.x < tC) { sC.x = RT_MISS_DST; sC.y = __uint_as_float(RT_HIT_NONE); auxMask &= ~2u; }
                else sC.w = tC;
            }
            const bool albedoFinite = !(rt_isnan(auxAlbedo.x) || rt_isnan(auxAlbedo.y) || rt_isnan(auxAlbedo.z) || rt_isinf(auxAlbedo.x) || rt_isinf(auxAlbedo.y) || rt_isinf(auxAlbedo.z));
            if ((auxMask & 3u) == 0u && albedoFinite) {
                // Both queries are answered "not emissive": the next segment's share of diffuseBRDF (:443-460) is known now.
                // realLightPDF = 0 gives k = 0, so directLight = emission * ((albedo / pi * nDotL) * 0) = +-0 (emitMode: every
                // material's emission is finite; the albedo is, checked above; nDotL = max(0, .) is) — and adding +-0 * attenuation
                // to totalColor is adding +0 * attenuation, which is what a segment without pending results does. lightPDF = 0
                // leaves cosineMisWeight = c^2 / (0 * 0 + c^2) with c = cosineHemispherePDF(n, cosineSample), the very expression of
                // the pending block. So this bounce is finished here: no records, nothing pending.
                float m2 = auxCosPdfC * auxCosPdfC / (0.f * 0.f + auxCosPdfC * auxCosPdfC);
                if (rt_isnan(m2)) m2 = 0.f;
                misW = m2;
                wantAux = false;
            } else {
                if (auxMask & 1u) ps.hit(RAY_NEE)[slot] = sL;     // the traversal's seed; an answered query needs no record (att.w bits 29 / 28)
                if (auxMask & 2u) ps.hit(RAY_PROBE)[slot] = sC;
                ps.auxO()[slot] = mk4(auxOrigin, 0.f);
                ps.auxDL()[slot] = mk4(auxL, auxCosPdfL);
                ps.auxDC()[slot] = mk4(auxC, auxCosPdfC);
                ps.pendAlbedo()[slot] = mk4(auxAlbedo, auxNdotL);
            }
        }
        const uint32_t answered = wantAux ? (((auxMask & 1u) ? 0u : 0x20000000u) | ((auxMask & 2u) ? 0u : 0x10000000u)) : 0u;
        ps.rayO()[slot] = mk4(ro, misW);
        ps.rayD()[slot] = mk4u(rd, state);
        ps.att()[slot] = mk4u(att, j | (wantAux ? 0x80000000u : 0u) | (specular ? 0x40000000u : 0u) | answered);
        ps.total()[slot] = mk4u(total, samplesDone);
    }
}

template <bool MAPS>
__device__ __forceinline__ void shade_block(const DevScene& sc, const PathState& ps, const ShadeArgs& sa, const FrameParams& fp) {
    __shared__ uint32_t s_cnt[RT_BLOCK / RT_WAVE][8];  // per wave: alive, main rays, refRays, paths, segments, emitter tests, NEE rays, cosine probes
    __shared__ uint32_t s_base[4];
    const uint32_t n = *sa.inCount;
    if (blockIdx.x * RT_BLOCK >= n) return;  // block-uniform
    const uint32_t gid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool live = gid < n;

    bool alive = false;    // path (or its successor sample) has a main ray for the next round
    uint32_t auxMask = 0;  // ... bit 0: and a NEE ray, bit 1: and a cosine probe
    uint32_t slot = 0;
    uint32_t refRays = 0, nPaths = 0, emitTests = 0;
    bool path = live;
    if (live) {
        slot = sa.inActive[gid];
        path = slot != 0xffffffffu;  // RT_QUEUE_HOLE
        if (path) shade_path<MAPS>(sc, ps, fp, slot, alive, auxMask, refRays, nPaths, emitTests);
    }

    // Queue compaction: ranks inside a wave from ballots, wave offsets through LDS, and ONE atomic
    // per block and queue (a single hot counter saturates near 90 atomics/us; per-wave atomics made
    // this kernel wait on them for half of its run time).
    const unsigned long long mAlive = __ballot(alive), mM = __ballot(alive && !(auxMask & 4u));
    const unsigned long long mL = __ballot(alive && (auxMask & 1u)), mC = __ballot(alive && (auxMask & 2u));
    const uint32_t nAlive = __popcll(mAlive), nM = __popcll(mM), nL = __popcll(mL), nC = __popcll(mC);
    const uint32_t wRef = wave_sum_u32(refRays), wPaths = wave_sum_u32(nPaths), wSeg = wave_sum_u32(path ? 1u : 0u), wEmit = wave_sum_u32(emitTests);
    const uint32_t wv = threadIdx.x / RT_WAVE;
    if (lane_id() == 0) {
        s_cnt[wv][0] = nAlive; s_cnt[wv][1] = nM; s_cnt[wv][2] = wRef; s_cnt[wv][3] = wPaths; s_cnt[wv][4] = wSeg; s_cnt[wv][5] = wEmit; s_cnt[wv][6] = nL; s_cnt[wv][7] = nC;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tA = 0, tR = 0, tRef = 0, tP = 0, tS = 0, tE = 0, tX = 0, tY = 0;
        for (int w = 0; w < RT_BLOCK / RT_WAVE; w++) { tA += s_cnt[w][0]; tR += s_cnt[w][1]; tRef += s_cnt[w][2]; tP += s_cnt[w][3]; tS += s_cnt[w][4]; tE += s_cnt[w][5]; tX += s_cnt[w][6]; tY += s_cnt[w][7]; }
        s_base[0] = tA ? atomicAdd(sa.outActiveCount, tA) : 0u;
        s_base[1] = tR ? atomicAdd(sa.outRayCount, tR) : 0u;
        s_base[2] = tX ? atomicAdd(RT_QUEUE_ORDER ? sa.outAuxCount2 : sa.outAuxCount, tX) : 0u;
        s_base[3] = tY ? atomicAdd(RT_QUEUE_ORDER ? sa.outAuxCount : sa.outAuxCount2, tY) : 0u;
        atomicAdd(&sa.counters->raysReference, (unsigned long long)tRef);
        atomicAdd(&sa.counters->paths, (unsigned long long)tP);
        atomicAdd(&sa.counters->segments, (unsigned long long)tS);
        if (tE) atomicAdd(&sa.counters->emitterTests, (unsigned long long)tE);
    }
    __syncthreads();
    if (alive) {
        uint32_t baseA = s_base[0], baseR = s_base[1];
        uint32_t baseX = (RT_QUEUE_ORDER ? 2u : 1u) * sa.auxOffset + s_base[2], baseY = (RT_QUEUE_ORDER ? 1u : 2u) * sa.auxOffset + s_base[3];
        for (uint32_t w = 0; w < wv; w++) { baseA += s_cnt[w][0]; baseR += s_cnt[w][1]; baseX += s_cnt[w][6]; baseY += s_cnt[w][7]; }
        sa.outActive[baseA + lanes_below(mAlive)] = slot;
        // the queue's pieces: main rays, NEE rays, cosine probes
        if (!(auxMask & 4u)) sa.outRays[baseR + lanes_below(mM)] = (slot << 2) | RAY_MAIN;
        if (auxMask & 1u) sa.outRays[baseX + lanes_below(mL)] = (slot << 2) | RAY_NEE;
        if (auxMask & 2u) sa.outRays[baseY + lanes_below(mC)] = (slot << 2) | RAY_PROBE;
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_shade(DevScene sc, PathState ps, ShadeArgs sa, FrameParams fp) { shade_block<false>(sc, ps, sa, fp); }
// the same for a scene that binds a metalness or a bump map (DevScene::mapFlags)
__global__ __launch_bounds__(RT_BLOCK) void k_shade_maps(DevScene sc, PathState ps, ShadeArgs sa, FrameParams fp) { shade_block<true>(sc, ps, sa, fp); }

// ---------------------------------------------------------------- k_resolve
// raytrace.comp:574-593. `rgba` holds the previous frame when progressive
// (kept in fp32 instead of the reference's 8-bit image).
// raytrace.comp:574-582 for one frame: sample mean, progressive blend with what the image holds, NaN/Inf -> magenta
__device__ __forceinline__ rt_vec3 blend_frame(const FrameParams& fp, uint32_t frameCount, float4 oldc, float4 accum) {
    rt_vec3 out = f4xyz(accum);
    float fs = (float)fp.samples;
    out = rt_v3(out.x / fs, out.y / fs, out.z / fs);
    float weight = 1.f / ((float)frameCount + 1.f);
    rt_vec3 blended = rt_add(rt_scale(rt_v3(oldc.x, oldc.y, oldc.z), 1.f - weight), rt_scale(out, weight));
    rt_vec3 fin = fp.progressive ? blended : out;
    if (rt_isnan(fin.x) || rt_isnan(fin.y) || rt_isnan(fin.z) || rt_isinf(fin.x) || rt_isinf(fin.y) || rt_isinf(fin.z))
        fin = rt_v3(1.f, 0.f, 1.f);
    return fin;
}

__device__ __forceinline__ void resolve_pixel(const PathState& ps, const FrameParams& fp, float4* rgba, uint32_t slot) {
    uint32_t gx, gy, krow;
    slot_to_pixel(fp, slot, gx, gy, krow);
    const size_t px = (size_t)krow * fp.width + gx;
    rt_vec3 fin = blend_frame(fp, fp.frameCount, rgba[px], ps.accum()[slot]);
    float s0 = (float)ps.statBox()[slot], s1 = (float)ps.statTri()[slot];
    float boxCap = (float)fp.boxCap, triCap = (float)fp.triCap;
    if (fp.debug == 0) {
        fin = s0 > boxCap ? rt_v3(1.f, 0.f, 0.f) : rt_v3(s0 / boxCap, s0 / boxCap, s0 / boxCap);
    } else if (fp.debug == 1) {
        fin = s1 > triCap ? rt_v3(1.f, 0.f, 0.f) : rt_v3(s1 / triCap, s1 / triCap, s1 / triCap);
    } else if (fp.debug == 2) {
        fin = rt_v3(s0 / boxCap, 0.f, s1 / triCap);
    }
    rgba[px] = make_float4(fin.x, fin.y, fin.z, 1.f);
}

// rt_render_frames: the frames of one launch blended into the image one after the other, in frame order, exactly as
// nFrames dispatches would have done it (the sample sums of frame f's pixels wait in ps.accum()[frame_slot(f, slot)])
__global__ __launch_bounds__(RT_BLOCK) void k_blend_frames(PathState ps, FrameParams fp, float4* rgba) {
    const uint32_t slot = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (slot >= fp.nPixels) return;  // `slot` is a slot of the tile here
    uint32_t gx, gy, krow;
    slot_to_pixel(fp, frame_slot(fp, 0u, slot), gx, gy, krow);
    const size_t px = (size_t)krow * fp.width + gx;
    float4 c = rgba[px];
    for (uint32_t f = 0; f < fp.nFrames; f++) {
        const rt_vec3 fin = blend_frame(fp, fp.frameCount + f, c, ps.accum()[frame_slot(fp, f, slot)]);
        c = make_float4(fin.x, fin.y, fin.z, 1.f);
    }
    rgba[px] = c;
}

__global__ __launch_bounds__(RT_BLOCK) void k_resolve(PathState ps, FrameParams fp, float4* rgba) {
    uint32_t slot = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (slot >= fp.nPixels) return;
    resolve_pixel(ps, fp, rgba, slot);
}

// ---------------------------------------------------------------- k_render_fused (wave-private pipeline)
// The same stages, run by each wave on its own pixels: init_path, then {trace_wave over the wave's ray list
// in LDS, shade_path, compaction of the next rays with __ballot ranks} until a pixel has finished all its
// samples, then resolve_pixel. Slots are reserved with one atomic per hand-out: a block of `batchPixels` <= 64
// consecutive slots when all lanes are free (pixelRefill = 64: a block at a time), or as many slots as there
// are free lanes as soon as `pixelRefill` of them are free. There is no device-wide barrier between the stages
// of different waves, no global ray queue and no per-round launch, so a small tile (one of 8 GPUs renders 1/8
// of the frame: ~340 k rays per round) does not wait for the slowest ray of the whole tile each round, as
// the multi-kernel pipeline does (measured: 44 % of its full-frame efficiency on a 1/8-height tile).
// Pixels are identical by construction: the per-pixel code is the same device functions.
struct FusedArgs {
    uint32_t* batchHead;   // zeroed before the launch; counts reserved pixel slots
    float4* rgba;
    DevCounters* counters;
    uint32_t* overflow;    // OVF only
    uint32_t refill, wSetup, wLeaf, fastLanes;
    uint32_t batchPixels;  // pixels per wave-private block, <= 64 (chosen by the host so the blocks fill the resident waves evenly)
    uint32_t scatter;      // g > 0: a block is made of chunks of g consecutive slots taken nBatches chunks apart
    uint32_t fastShare;
    unsigned long long* waveTimes;  // phase_stats only: wall_clock64() at the start and the end of every wave
    uint32_t pixelRefill;  // free lanes that make the wave reserve new pixels (>= batchPixels: only when all are free)
};

// The kernel-argument segment as memory the compiler knows nothing about: loads through the returned pointer
// cannot be hoisted above the call, so values that are only needed now and then are fetched (scalar loads,
// scalar cache) where they are used instead of occupying scalar registers across the traversal loop.
template <typename T>
__device__ __forceinline__ const T* opaque_kernarg() {
    unsigned long long v = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(v));
    return (const T*)(const RT_AS_CONSTANT T*)v;  // constant address space: what is read through it are scalar loads (a plain generic pointer made them FLAT vector loads)
}

struct FusedKernArgs {  // the whole kernel-argument segment, so that it can be addressed as memory
    DevScene sc;
    PathState ps;
    FrameParams fp;
    FusedArgs fa;
};

template <int STACK, bool OVF, bool PIX, bool CULL>
__global__ __launch_bounds__(RT_BLOCK, 5) void k_render_fused(FusedKernArgs ka) {
    const DevScene& sc = ka.sc;
    const PathState& ps = ka.ps;
    const FrameParams& fp = ka.fp;
    const FusedArgs& fa = ka.fa;
    __shared__ uint32_t s_stack[(RT_BLOCK / RT_WAVE) * (STACK + 1) * RT_WAVE];
    __shared__ uint32_t s_list[RT_BLOCK / RT_WAVE][3 * RT_WAVE];
    __shared__ float4 s_box[64];  // DevScene::maskBox (the objects of the mask's window that can be ruled out): the rays' object masks are computed from here (reach_mask_from)
    __shared__ uint2 s_meta[RT_META_LDS];
    const uint32_t nBox = CULL ? sc.reachCount : 0u;
    if (CULL && threadIdx.x < 2u * nBox) s_box[threadIdx.x] = sc.maskBox[threadIdx.x];
    fill_meta_lds(sc, s_meta);
    const uint32_t wv = threadIdx.x / RT_WAVE;
    uint32_t* stack = s_stack + wv * (STACK + 1) * RT_WAVE + (threadIdx.x & (RT_WAVE - 1));
    uint32_t* list = s_list[wv];
    uint32_t* ovf = OVF ? fa.overflow + (size_t)blockIdx.x * RT_BLOCK + threadIdx.x : nullptr;
    const size_t ovfStride = (size_t)gridDim.x * RT_BLOCK;
    const TracePwArgs ta{nullptr, nullptr, nullptr, fa.refill, 0u, fa.wSetup, fa.wLeaf, fa.fastLanes, fa.fastShare, nullptr, nullptr, fa.counters, nullptr, nullptr, fa.overflow};
    WaveTotals wt;
    uint32_t refTot = 0, pathTot = 0, segTot = 0, emitTot = 0;
    const unsigned long long tKernelStart = fa.waveTimes ? wall_clock64() : 0ull;
    // scatter = g > 0: batchPixels is a multiple of g and a block is batchPixels / g chunks of g slots, nBatches apart
    const uint32_t nBatches = fa.scatter ? ((fp.nPixels + fa.scatter - 1) / fa.scatter + fa.batchPixels / fa.scatter - 1) / (fa.batchPixels / fa.scatter)
                                         : ((fp.nFrames > 1u ? ((fp.nPixels + 63u) >> 6) * 64u * fp.nFrames : fp.nPixels) + fa.batchPixels - 1) / fa.batchPixels;

    // A lane keeps a pixel until all its samples are done, then resolves it and takes the next one: when `pixelRefill`
    // of the wave's lanes are free (or all of them), the wave reserves that many slots with one atomic. The wave stays
    // populated until the tile runs out, instead of draining to its slowest pixel once per block.
    const uint32_t nSlots = fp.nFrames > 1u ? ((fp.nPixels + 63u) >> 6) * 64u * fp.nFrames : fp.nPixels;  // nFrames > 1 comes with scatter = 0
    const uint32_t total = fa.scatter ? nBatches * fa.batchPixels : nSlots;
    const uint32_t refillAt = min(max(fa.pixelRefill, 1u), fa.batchPixels);
    uint32_t slot = 0;
    bool valid = false, alive = false, exhausted = false;
    uint32_t auxMask = 0;  // bit 0: the pixel's path has a NEE ray in flight, bit 1: a cosine probe
    for (;;) {
        const bool mine = lane_id() < fa.batchPixels && !alive;
        const unsigned long long mF = __ballot(mine);
        const uint32_t take = __popcll(mF);
        if (!exhausted && take >= refillAt) {
            uint32_t base = 0;
            if (lane_id() == 0) base = atomicAdd(fa.batchHead, take);
            base = __shfl(base, 0, RT_WAVE);
            if (base >= total) exhausted = true;
            else if (mine) {
                // Frame constants and the shading tables are re-read from the kernel-argument segment where they are used
                // (the asm makes the pointers opaque, so the loads cannot be hoisted): held across the traversal loop they
                // cost ~60 scalar registers of a kernel that has none to spare.
                const FusedKernArgs* kq = opaque_kernarg<FusedKernArgs>();
                const FrameParams* fq = &kq->fp;
                const DevScene* sq = &kq->sc;
                if (valid && fq->nFrames == 1u) resolve_pixel(ps, *fq, fa.rgba, slot);  // several frames: k_blend_frames, afterwards
                const uint32_t a = base + lanes_below(mF);
                uint32_t ns = a;
                if (fa.scatter) {
                    // a block's pixels are spread over the whole tile (chunks of `scatter` consecutive slots, nBatches chunks
                    // apart), so that all blocks cost about the same when every wave gets just one of them
                    const uint32_t ch = a / fa.scatter, perBlock = fa.batchPixels / fa.scatter;
                    ns = ((ch % perBlock) * nBatches + ch / perBlock) * fa.scatter + a % fa.scatter;
                }
                valid = a < total && ns < nSlots && (fq->nFrames == 1u || slot_in_tile(*fq, ns) < fq->nPixels);
                slot = ns;
                auxMask = 0;
                if (valid) {
                    init_path(*sq, ps, *fq, slot);
                    alive = fp.samples > 0;
                }
            }
        }
        const unsigned long long mA = __ballot(alive);
        if (mA == 0) {
            if (exhausted) break;
            continue;
        }
        const unsigned long long mM = __ballot(alive && !(auxMask & 4u));  // bit 2: the kept camera hit stands in for the main ray
        const unsigned long long mL = __ballot(alive && (auxMask & 1u)), mC = __ballot(alive && (auxMask & 2u));
        const uint32_t nM = __popcll(mM), nL = __popcll(mL), nC = __popcll(mC);
        if (alive) {
            if (!(auxMask & 4u)) list[lanes_below(mM)] = (slot << 2) | RAY_MAIN;
            if (auxMask & 1u) list[nM + lanes_below(mL)] = (slot << 2) | RAY_NEE;
            if (auxMask & 2u) list[nM + nL + lanes_below(mC)] = (slot << 2) | RAY_PROBE;
        }
        const uint32_t nRays = nM + nL + nC;
        __threadfence_block();  // the rays written by shade_path / init_path are read by other lanes of this wave
        trace_wave<STACK, OVF, PIX, false, true, CULL>(sc, ps, ta, stack, ovf, ovfStride, list, nRays, wt, s_meta);
        __threadfence_block();  // ... and so are the hit records
        if (alive) {
            bool nowAlive = false;
            uint32_t refRays = 0, nPaths = 0;
            const FusedKernArgs* kq = opaque_kernarg<FusedKernArgs>();
            const FrameParams* fq = &kq->fp;
            const DevScene* sq = &kq->sc;
            shade_path(*sq, ps, *fq, slot, nowAlive, auxMask, refRays, nPaths, emitTot, false);
            segTot++;
            if (CULL && nowAlive && nBox) {
                // the new rays' object masks (sphere_seed), here rather than inside shade_path: its registers are spilling already
                float4 sd;
                if (!(auxMask & 4u)) {
                    sd = ps.hit(RAY_MAIN)[slot];
                    sd.z = __uint_as_float(reach_mask_from(s_box, nBox, f4xyz(ps.rayO()[slot]), f4xyz(ps.rayD()[slot]), sc.cullOriginLimit));
                    ps.hit(RAY_MAIN)[slot] = sd;
                }
                if (auxMask & 1u) {
                    sd = ps.hit(RAY_NEE)[slot];
                    sd.z = __uint_as_float(reach_mask_from(s_box, nBox, f4xyz(ps.auxO()[slot]), f4xyz(ps.auxDL()[slot]), sc.cullOriginLimit));
                    ps.hit(RAY_NEE)[slot] = sd;
                }
                if (auxMask & 2u) {
                    sd = ps.hit(RAY_PROBE)[slot];
                    sd.z = __uint_as_float(reach_mask_from(s_box, nBox, f4xyz(ps.auxO()[slot]), f4xyz(ps.auxDC()[slot]), sc.cullOriginLimit));
                    ps.hit(RAY_PROBE)[slot] = sd;
                }
            }
            alive = nowAlive;
            auxMask = nowAlive ? auxMask : 0u;
            refTot += refRays;
            pathTot += nPaths;
        }
    }
    if (valid) {  // the pixels that finished after the tile ran out
        const FusedKernArgs* kq = opaque_kernarg<FusedKernArgs>();
        if (kq->fp.nFrames == 1u) resolve_pixel(ps, kq->fp, fa.rgba, slot);
    }

    if (fa.waveTimes && lane_id() == 0) {  // phase_stats: when did this wave run out of blocks?
        const size_t w = (size_t)blockIdx.x * (RT_BLOCK / RT_WAVE) + threadIdx.x / RT_WAVE;
        fa.waveTimes[2 * w] = tKernelStart;
        fa.waveTimes[2 * w + 1] = wall_clock64();
    }
    unsigned long long wb = wave_sum_u64(wt.totBox), wtri = wave_sum_u64(wt.totTri);
    uint32_t wr = wave_sum_u32(wt.totRays), wh = wave_sum_u32(wt.totHits);
    uint32_t wRef = wave_sum_u32(refTot), wP = wave_sum_u32(pathTot), wS = wave_sum_u32(segTot), wE = wave_sum_u32(emitTot);
    const unsigned long long wskip = CULL ? wave_sum_u64(wt.totSkipBox) : 0ull;
    if (lane_id() == 0 && (wr | wP | wS)) {
        if (CULL && wskip) atomicAdd(&fa.counters->skippedBoxTests, wskip);
        if (wE) atomicAdd(&fa.counters->emitterTests, (unsigned long long)wE);
        atomicAdd(&fa.counters->boxTests, wb);
        atomicAdd(&fa.counters->triTests, wtri);
        atomicAdd(&fa.counters->raysTraced, (unsigned long long)wr);
        atomicAdd(&fa.counters->raysHit, (unsigned long long)wh);
        atomicAdd(&fa.counters->raysReference, (unsigned long long)wRef);
        atomicAdd(&fa.counters->paths, (unsigned long long)wP);
        atomicAdd(&fa.counters->segments, (unsigned long long)wS);
    }
}

// ---------------------------------------------------------------- rt_trace_rays support
__global__ __launch_bounds__(RT_BLOCK) void k_hit_details(DevScene sc, PathState ps, uint32_t n, const uint32_t* perRayBox,
                                                          const uint32_t* perRayTri, RtHit* out) {
    uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    RtHit h;
    memset(&h, 0, sizeof(h));
    const float4 hm = ps.hit(RAY_MAIN)[i];
    h.dst = hm.x;
    uint32_t obj = __float_as_uint(hm.y);
    const uint32_t tri = __float_as_uint(hm.z);
    h.boxTests = perRayBox[i];
    h.triTests = perRayTri[i];
    if (obj != RT_HIT_NONE) {
        FullHit f = reconstruct_hit<true>(sc, f4xyz(ps.rayO()[i]), f4xyz(ps.rayD()[i]), obj, tri);
        h.didHit = 1;
        h.isSphere = (obj & RT_HIT_SPHERE) ? 1u : 0u;
        h.objectHitIndex = obj & ~RT_HIT_SPHERE;
        h.triHitIndex = h.isSphere ? 0u : tri;
        h.materialIndex = f.materialIndex;
        h.frontFace = f.frontFace;
        h.hitPoint[0] = f.hitPoint.x; h.hitPoint[1] = f.hitPoint.y; h.hitPoint[2] = f.hitPoint.z;
        h.normal[0] = f.normal.x; h.normal[1] = f.normal.y; h.normal[2] = f.normal.z;
    }
    out[i] = h;
}

__global__ __launch_bounds__(RT_BLOCK) void k_seed_rays(DevScene sc, PathState ps, uint32_t n) {
    uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < n) ps.hit(RAY_MAIN)[i] = sphere_seed(sc, f4xyz(ps.rayO()[i]), f4xyz(ps.rayD()[i]));
}

// ---------------------------------------------------------------- misc kernels
// start of a multi-kernel dispatch: n active paths and n rays in buffer 0, nothing in buffer 1, work counter 0
__global__ void k_init_counts(uint32_t* counts, uint32_t n) {
    if (threadIdx.x == 0) { counts[0] = n; counts[1] = 0; counts[2] = n; counts[3] = 0; counts[4] = 0; counts[5] = 0; counts[6] = 0; counts[7] = 0; counts[8] = 0; }
}
__global__ void k_zero_counts(uint32_t* a, uint32_t* b, uint32_t* c, uint32_t* d, uint32_t* e) {
    if (threadIdx.x == 0) { *a = 0; *b = 0; *c = 0; *d = 0; *e = 0; }
}

// Hash of the deterministic-math primitives over a fixed input table; the
// host computes the same hash with the same header (rt_device_selftest).
__device__ __host__ inline uint32_t selftest_mix(uint32_t h, float v) {
    h ^= rt_f2u(v);
    h *= 16777619u;
    return h;
}
__device__ __host__ inline uint32_t selftest_one(float a, float b) {
    uint32_t h = 2166136261u;
    float s, c;
    rt_sincos(a * 6.2831855f, &s, &c);
    h = selftest_mix(h, s); h = selftest_mix(h, c);
    h = selftest_mix(h, a / b); h = selftest_mix(h, 1.f / (b + 0.25f));
    h = selftest_mix(h, rt_sqrt(a)); h = selftest_mix(h, rt_pow(a, 5.f)); h = selftest_mix(h, rt_pow(a, 0.35f));
    h = selftest_mix(h, rt_log2(b + 1e-3f)); h = selftest_mix(h, rt_exp2(a * 20.f - 10.f));
    rt_vec3 v = rt_normalize(rt_v3(a - 0.5f, b - 0.5f, a * b + 0.1f));
    h = selftest_mix(h, v.x); h = selftest_mix(h, v.y); h = selftest_mix(h, v.z);
    h = selftest_mix(h, rt_dot(v, rt_v3(b, a, 0.3f)));
    rt_vec3 r = rt_refract(v, rt_normalize(rt_v3(0.1f, 1.f, b)), 0.5f + a);
    h = selftest_mix(h, r.x); h = selftest_mix(h, r.y); h = selftest_mix(h, r.z);
    h = selftest_mix(h, rt_smoothstep(0.f, 0.4f, a)); h = selftest_mix(h, rt_min(a, b)); h = selftest_mix(h, rt_max(a, b));
    h = selftest_mix(h, a * b + b);  // would differ if contracted to an fma
    uint32_t st = rt_f2u(a) ^ (rt_f2u(b) * 7u);
    h = selftest_mix(h, rt_random(&st));
    return h;
}
__global__ void k_selftest(const float* a, const float* b, uint32_t n, uint32_t* hashOut, uint32_t* bitsOut, const float* kat) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hashOut[i] = selftest_one(a[i], b[i]);
    if (i == 0) *bitsOut = rt_selftest_bits(kat);
}

// rt_gather_strips on the gathering rank: strips stored rank after rank (rank r: image rows r, r + n, ...) -> image rows
__global__ __launch_bounds__(RT_BLOCK) void k_deinterleave_rows(const float4* __restrict__ strips, float4* __restrict__ frame, uint32_t width,
                                                                uint32_t height, uint32_t nRanks) {
    const size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t r = y % nRanks, k = y / nRanks;
    // rows of the ranks before r: rank q has ceil((height - q) / nRanks) rows
    const uint32_t full = height / nRanks, extra = height % nRanks;
    const size_t before = (size_t)r * full + (r < extra ? r : extra);
    frame[i] = strips[(before + k) * width + x];
}

// include/rt_probe.h on the device: the raw values of every GLSL built-in, for tests/test_glsl_builtins.py
__global__ void k_math_probe(const float* in, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[32], o[64];
    for (int k = 0; k < 32; k++) a[k] = in[(size_t)i * 32 + k];
    rt_math_probe(a, o);
    for (int k = 0; k < 64; k++) out[(size_t)i * 64 + k] = o[k];
}

__global__ void k_copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}
