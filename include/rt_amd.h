/*
 * rt_amd.h — C ABI of the MI355X path tracer (librt_amd.so).
 *
 * Two halves:
 *
 *  (1) Scene surface (host only, no GPU touched). Keeps the reference's scene
 *      surface: the AoS structs of src/vk_engine.h:49-79,117-123,145-189 with
 *      the same field names and std140 / push-constant layouts,
 *      and the scene builders of src/vk_engine.cpp (read_obj :800-1037,
 *      read_mtl :1060-1167, cornell_box :638-678, prepare_storage_buffers
 *      :680-758, build_bvh :1169-1337, camera matrix of run_compute
 *      :1631-1661).
 *
 *  (2) Device half. Replaces the Vulkan seam of the reference:
 *        copy_buffer   (src/vk_engine.cpp:1401-1444)  -> rt_upload_scene
 *        update_buffer (src/vk_engine.cpp:1446-1475)  -> rt_update_*
 *        run_compute   (src/vk_engine.cpp:1623-1676)  -> rt_render
 *      The per-pixel megakernel shaders/raytrace.comp is replaced by a
 *      wavefront pipeline of HIP kernels for gfx950.
 *
 * All entry points are extern "C", take plain pointers and sizes, return an
 * int status (0 = ok, <0 = error; message via rt_last_error) and never abort
 * (the reference's VK_CHECK aborts, src/vk_engine.cpp:20-27).
 * A ctx is bound to one GPU and is not thread-safe; one process per GPU.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Scene structs — byte-compatible with the reference's host structs   */
/* ------------------------------------------------------------------ */

/* src/vk_engine.h:49-53 (std140, 32 B) */
typedef struct Sphere {
    float position[3];
    float radius;            /* @12 */
    uint32_t materialIndex;  /* @16 */
    uint32_t _pad[3];
} Sphere;

/* src/vk_engine.h:55-62 (48 B). binormal/tangent are never initialised by
 * the reference (calculate_binormal assigns nothing) and never read. */
typedef struct Triangle {
    uint32_t v0, v1, v2;
    uint32_t frontOnly;
    float binormal[3]; float _pad0; /* @16 */
    float tangent[3];  float _pad1; /* @32 */
} Triangle;

/* src/vk_engine.h:64-67 (32 B): uv.x is position.w, uv.y is normal.w */
typedef struct TrianglePoint {
    float position[4];
    float normal[4];
} TrianglePoint;

/* src/vk_engine.h:69-79 (64 B) */
typedef struct RayMaterial {
    float albedo[3]; float _pad0;  /* @0  default (1,1,1) */
    float emissionColor[3];        /* @16 default (0,0,0) */
    float emissionStrength;        /* @28 */
    float reflectance;             /* @32 */
    float ior;                     /* @36 default -1 */
    int32_t albedoIndex;           /* @40 default -1 */
    int32_t metalnessIndex;        /* @44 */
    int32_t alphaIndex;            /* @48 */
    int32_t bumpIndex;             /* @52 */
    uint32_t _pad1[2];
} RayMaterial;

/* src/vk_engine.h:117-123 (80 B); transformMatrix is column-major */
typedef struct RenderObject {
    float transformMatrix[16];
    uint32_t smoothShade;    /* @64 */
    uint32_t bvhIndex;       /* @68 */
    uint32_t materialIndex;  /* @72 */
    uint32_t samplerIndex;   /* @76 */
} RenderObject;

/* src/vk_engine.h:185-189 (32 B). triCount == 0: interior, index = left
 * child, right child = index + 1; else leaf, index = first triangle. */
typedef struct BVHNode {
    float boundsX[2], boundsY[2], boundsZ[2]; /* (min,max) per axis */
    uint32_t index, triCount;
} BVHNode;

/* src/vk_engine.h:145-151 (96 B) */
typedef struct CameraInfo {
    float cameraRotation[16];
    float pos[3];            /* @64 default (0,-0.5,-3.5) */
    float nearPlane;         /* @76 default 0.1 */
    float aspectRatio;       /* @80 */
    float fov;               /* @84 default 50 */
    float _pad[2];
} CameraInfo;

/* src/vk_engine.h:153-158 (64 B) */
typedef struct EnvironmentData {
    float horizonColor[4];   /* w = sun focus */
    float zenithColor[4];    /* w = sun intensity */
    float groundColor[3]; float _pad0;
    float lightDir[4];       /* w = environment on */
} EnvironmentData;

/* src/vk_engine.h:160-171 (40 B); the two bools occupy 4-byte slots */
typedef struct RayTracerData {
    uint32_t progressive;
    uint32_t singleRender;
    int32_t debug;           /* -1 off, 0 box heat map, 1 tri heat map, 2 both */
    uint32_t raysPerPixel;
    uint32_t bounceLimit;
    uint32_t sphereCount;
    uint32_t objectCount;
    uint32_t triangleCap;
    uint32_t boxCap;
    uint32_t sampleLimit;
} RayTracerData;

/* src/vk_engine.h:173-178 (208 B) */
typedef struct PushConstants {
    CameraInfo camInfo;          /* @0   */
    EnvironmentData environment; /* @96  */
    RayTracerData rayTraceParams;/* @160 */
    uint32_t frameCount;         /* @200 */
    uint32_t _pad;
} PushConstants;

/* src/vk_engine.h:125-132 (the placement half of ImGuiObject) */
typedef struct RtPlacement {
    float position[3];
    float rotation[3];       /* Euler degrees, applied T*Rx*Ry*Rz*S */
    float scale[3];
    uint32_t samplerIndex;
    uint32_t frontOnly;
} RtPlacement;

enum { RT_MAX_TEXTURES = 64, RT_MAX_MATERIALS = 10, RT_MAX_SPHERES = 10, RT_BVH_BINS = 20 };
/* Light queries (rt_set_tuning "light_queries") are answered from the list of emissive primitives while the scene's objects
 * with an emissive material have at most this many triangles between them; beyond it every query is traversed in full. */
enum { RT_EMIT_MAX_TRIS = 32 };

/* Borrowed views of the scene's host vectors (VulkanEngine members
 * spheres / rayMaterials / triPoints / triangles / objects / bvhNodes,
 * src/vk_engine.h:270-283). */
typedef struct RtSceneArrays {
    const Sphere* spheres;              uint32_t sphereCount;
    const RayMaterial* materials;       uint32_t materialCount;
    const TrianglePoint* triPoints;     uint32_t triPointCount;
    const Triangle* triangles;          uint32_t triangleCount;
    const RenderObject* objects;        uint32_t objectCount;
    const BVHNode* bvhNodes;            uint32_t bvhNodeCount;
} RtSceneArrays;

/* ------------------------------------------------------------------ */
/* (1) Scene surface — host only                                        */
/* ------------------------------------------------------------------ */
typedef struct rt_scene rt_scene;

int  rt_scene_create(rt_scene** out);
void rt_scene_destroy(rt_scene* s);
const char* rt_scene_last_error(const rt_scene* s);

/* rayMaterials.push_back (src/vk_engine.cpp:717-722); returns the index or <0 */
int  rt_scene_add_material(rt_scene* s, const RayMaterial* m);
/* a default-constructed RayMaterial (src/vk_engine.h:69-79) */
void rt_material_default(RayMaterial* m);
/* spheres[i] = ... (src/vk_engine.cpp:682-686); i < RT_MAX_SPHERES */
int  rt_scene_set_sphere(rt_scene* s, uint32_t i, const float position[3], float radius, uint32_t materialIndex);
void rt_placement_default(RtPlacement* p);

/* VulkanEngine::read_obj (src/vk_engine.cpp:800-1037). `material` is the
 * fallback material index for files without usemtl. A missing file is a
 * silent no-op returning 1 (the reference returns silently, :834). */
int  rt_scene_read_obj(rt_scene* s, const char* filePath, const RtPlacement* placement, int material);
/* VulkanEngine::read_mtl (src/vk_engine.cpp:1060-1167) without the image
 * uploads: map_* lines only claim texture slots (textures are never sampled
 * by the shader at this snapshot). */
int  rt_scene_read_mtl(rt_scene* s, const char* filePath);
/* A programmatic mesh goes through the same triangle/centroid/BVH path as an
 * OBJ group. positions/normals: 9 floats per triangle (corner-major);
 * uvs: 6 floats per triangle or NULL. */
int  rt_scene_add_mesh(rt_scene* s, const char* key, const float* positions, const float* normals,
                       const float* uvs, uint32_t triCount, const RtPlacement* placement, int material);
/* VulkanEngine::cornell_box (src/vk_engine.cpp:638-678); assetDir holds
 * light2.obj, plane.obj, ceiling.obj */
int  rt_scene_cornell_box(rt_scene* s, const char* assetDir);
/* VulkanEngine::prepare_storage_buffers (src/vk_engine.cpp:680-758): ten
 * zeroed spheres, the six built-in materials, the two cubes, cornell_box. */
int  rt_scene_prepare_default(rt_scene* s, const char* assetDir);
int  rt_scene_get_arrays(const rt_scene* s, RtSceneArrays* out);
/* The texture slots the MTL files claimed, in slot order (textures[texturesUsed], src/vk_engine.cpp:1109-1141): the host
 * decodes file i to RGBA8 (the reference: stb_image, STBI_rgb_alpha, src/vk_textures.cpp:103-113) and hands all of them to
 * rt_upload_textures. rt_scene_add_texture claims the next slot for a file no MTL names (bind it with a material's
 * albedoIndex); rt_scene_set_material overwrites rayMaterials[i] (the material editor's in-place edit, :1536-1545). */
uint32_t rt_scene_texture_count(const rt_scene* s);
const char* rt_scene_texture_path(const rt_scene* s, uint32_t i);
int  rt_scene_add_texture(rt_scene* s, const char* path);
int  rt_scene_set_material(rt_scene* s, uint32_t i, const RayMaterial* m);
/* index of a material loaded from an MTL file, key "<mtlpath>/<name>"; -1 if absent */
int  rt_scene_find_material(const rt_scene* s, const char* key);
/* BVH build statistics of the most recent build (printed by the reference, :1187-1193) */
int  rt_scene_last_bvh_stats(const rt_scene* s, uint32_t* nodeCount, uint32_t* maxDepth, uint32_t* minDepth, uint32_t* maxTri);

/* BVH builder plug. The scene half builds every mesh's BVH on the host, as the reference does
 * (build_bvh / subdivide / find_split_plane, src/vk_engine.cpp:1169-1337). A hook replaces that step for the meshes added
 * after it is set: it receives the mesh's triangles and centroids (the last `count` elements of the scene arrays, the
 * triangles' v0/v1/v2 indexing `points`), must reorder both the way the reference's partition loop does, write the
 * nodes in the reference's numbering (children of the node at nodeBase + i at nodeBase + nodes[i].index ... as absolute
 * indices, leaves with absolute triangle indices from triIndex0) and report {maxDepth, minDepth, maxTriCount}.
 * rt_bvh_hook is such a hook: the same tree, built on the GPU of the rt_ctx passed as `user` (rt_bvh_build). */
typedef int (*RtBvhBuildHook)(void* user, const TrianglePoint* points, uint32_t pointCount, Triangle* triangles, float* centroids,
                              uint32_t count, uint32_t triIndex0, uint32_t nodeBase, BVHNode* nodesOut, uint32_t nodeCapacity,
                              uint32_t* nodeCountOut, uint32_t statsOut[3]);
int  rt_scene_set_bvh_hook(rt_scene* s, RtBvhBuildHook hook, void* user);

/* Camera/constants half of run_compute (src/vk_engine.cpp:1631-1661):
 * cameraRotation = rotY * rotX * rotZ from Euler degrees. */
void rt_camera_rotation(const float anglesDeg[3], float outMat4[16]);
/* defaults of src/vk_engine.h:145-171,325 (angles {4,0,0}) with aspect = W/H */
void rt_push_constants_default(PushConstants* pc, uint32_t width, uint32_t height);
/* object.transformMatrix = T*Rx*Ry*Rz*S (src/vk_engine.cpp:1012-1016) */
void rt_transform_matrix(const RtPlacement* p, float outMat4[16]);

/* ------------------------------------------------------------------ */
/* (2) Device half                                                      */
/* ------------------------------------------------------------------ */
typedef struct rt_ctx rt_ctx;

/* Global counters summed over every closest-hit query the device executed
 * since the last rt_reset_counters. */
typedef struct RtCounters {
    uint64_t boxTests;      /* reference stats[0] semantics, raytrace.comp:338, all queries */
    uint64_t triTests;      /* reference stats[1] semantics, raytrace.comp:310, all queries */
    uint64_t raysTraced;    /* scene queries executed on the device: main rays, and the light queries (NEE ray, cosine probe) that
                             * their creators could not answer from the emitter list; boxTests / triTests count these queries,
                             * the light queries up to the hit that answers them */
    uint64_t raysHit;       /* of those, queries that reported a hit */
    uint64_t raysReference; /* queries the reference megakernel would have issued for the same paths */
    uint64_t paths;         /* trace() calls finished (pixel samples) */
    uint64_t segments;      /* path segments shaded */
    uint64_t traceLaunches; /* launches of the traversal kernel */
    uint64_t emitterTests;  /* emissive primitives tested directly by the creators of light queries (NEE ray, cosine probe) */
    uint64_t skippedBoxTests; /* of boxTests: the two tests the reference makes on the children of an object's root, charged for objects
                             * a ray was taken past without entering them (object masks, padded world boxes, the object hierarchy);
                             * boxTests - skippedBoxTests were executed. 0 for scenes without placed objects */
} RtCounters;

/* One closest-hit record of calculateIntersections (raytrace.comp:276-353) */
typedef struct RtHit {
    float dst;              /* RT_MISS_DST on miss */
    uint32_t didHit;
    uint32_t isSphere;
    uint32_t objectHitIndex;/* object index, or sphere index when isSphere */
    uint32_t triHitIndex;
    uint32_t materialIndex;
    uint32_t frontFace;
    float hitPoint[3];
    float normal[3];
    uint32_t boxTests;      /* this query's stats[0] */
    uint32_t triTests;      /* this query's stats[1] */
} RtHit;

int  rt_device_count(int* out);
int  rt_create(int device, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = ctx's own */
int  rt_set_stream(rt_ctx* ctx, void* hipStream);

/* copy_buffer x6 (src/vk_engine.cpp:686,753-757): takes the reference's AoS
 * host arrays, converts to the device layouts, uploads. Borrowed for the call. */
int  rt_upload_scene(rt_ctx* ctx, const RtSceneArrays* scene);
/* Textures (bindings raytrace.comp:122,148, upload src/vk_textures.cpp:103-200). Slot i of the scene's texture
 * table, as R8G8B8A8_SRGB texels, rows top to bottom (what stbi_load(..., STBI_rgb_alpha) returns). The snapshot's shader
 * computes hit.uv (raytrace.comp:249-256) and never samples; the semantics here are this build's declared choice
 * ("parity unpinned"): a triangle hit whose material has albedoIndex >= 0 multiplies the material's albedo by
 * the texel at (u, 1 - v) of hit.uv — OBJ's v runs upwards, the image's rows downwards; with the flip dread.obj shows its
 * plate, bolts and wheels where the reference's renders/dread_texture.png has them — nearest filter, sampler
 * RenderObject.samplerIndex (0 = repeat, 1 = clamp to edge, :525-531), sRGB -> linear. Spheres are not textured.
 * The other three slots a material carries (src/vk_engine.cpp:1109-1141: map_Ks -> metalnessIndex, map_d -> alphaIndex,
 * map_bump -> bumpIndex) are declared likewise, all on the red channel of the texel at hit.uv (rt_det_math.h, "the other three
 * map slots"): alpha — a triangle hit whose texel decodes below 0.5 is no hit, for every ray (inside calculateIntersections'
 * triangle loop, raytrace.comp:305-322); metalness — the decoded texel replaces RayMaterial.reflectance (which the shader only
 * compares with 0, :509); bump — the decoded texel is a height and the interpolated normal is tilted by the height steps to the
 * next texel of the row and of the column along the triangle's dP/du, dP/dv (rt_bump_normal). A slot < 0 or beyond the uploaded
 * table binds nothing. A scene that binds one of the three is rendered by the multi-kernel pipeline (k_shade_maps,
 * k_trace_pw_alpha) whatever "pipeline" says; one whose emissive material has an alpha map traces its light queries in full.
 * Borrowed for the call; n = 0 removes all textures. */
typedef struct RtTexture {
    uint32_t width, height;
    const uint8_t* rgba8;   /* width * height * 4 bytes */
} RtTexture;
int  rt_upload_textures(rt_ctx* ctx, const RtTexture* textures, uint32_t n);
/* update_buffer (src/vk_engine.cpp:1545,1572,1603) */
int  rt_update_materials(rt_ctx* ctx, const RayMaterial* m, uint32_t n);
int  rt_update_spheres(rt_ctx* ctx, const Sphere* s, uint32_t n);
int  rt_update_objects(rt_ctx* ctx, const RenderObject* o, uint32_t n);

/* run_compute: one dispatch of the path tracer over rows
 * y = row0 + k*rowStride, k in [0,nRows) of a width x height image, using the
 * global pixel index for the RNG seed so any tiling gives identical pixels.
 * samples per pixel = singleRender ? sampleLimit : raysPerPixel (:570).
 * d_rgba: device pointer to nRows*width*4 floats (RGBA fp32, before any 8-bit
 * step), or NULL to use the ctx's own framebuffer (rt_read_rgba_f32).
 * Asynchronous on the ctx stream; rt_sync waits. The call returns as soon as the dispatch is enqueued: the multi-kernel
 * pipeline enqueues every round it can need (each kernel reads its queue length on the device and leaves at once when
 * there is nothing left) in up to "lanes" parts on streams of their own, forked from and joined to the ctx stream, and
 * never waits for the device (tests/test_async_contract.py: the host is held for less than a tenth of the dispatch's
 * time on the bench frame in either pipeline). One exception: a context's first dispatch of >= 8 M pixel-samples of a
 * scene whose ray cost it has not measured yet is preceded by a small blocking probe dispatch (a few ms, rt_ray_cost;
 * "probe" 0 turns it off). */
int  rt_render(rt_ctx* ctx, const PushConstants* pc, uint32_t width, uint32_t height,
               uint32_t row0, uint32_t rowStride, uint32_t nRows, float* d_rgba);
/* nFrames consecutive progressive dispatches of the same tile — the same pixels, bit for bit, as nFrames calls of rt_render
 * with frameCount = pc->frameCount, pc->frameCount + 1, ... (the reference's frame loop, src/vk_engine.cpp:1782-1814, runs them
 * one after the other and waits for each). Frames are independent until they are blended (each has its own RNG seeds,
 * raytrace.comp:562-564), and a pixel's samples are serial, so what fills a GPU is paths: the frames' paths share one dispatch
 * (one launch of the fused kernel, or the queues of every round of the multi-kernel pipeline; the pipeline is chosen by the
 * paths of all the frames together) and the frames are blended into d_rgba in order afterwards. A 1/8-height tile of a
 * 1080p frame has fewer pixels than the GPU has resident lanes; eight frames of it run like a whole frame, and the
 * traversal gets cheaper per ray the more paths a dispatch holds. Not for the debug heat maps (rendered frame by frame then).
 * "frames_per_launch" (rt_set_tuning) caps the frames per dispatch (0 = as many as fit), "frames_max_mslots" the paths of
 * one dispatch in millions (default 24: 5.8 GB of path state); more frames than that go in several dispatches. */
int  rt_render_frames(rt_ctx* ctx, const PushConstants* pc, uint32_t width, uint32_t height,
                      uint32_t row0, uint32_t rowStride, uint32_t nRows, uint32_t nFrames, float* d_rgba);
int  rt_sync(rt_ctx* ctx);
/* forget the ctx's own framebuffer: the next rt_render(…, NULL) starts a new progressive history from a zeroed image */
int  rt_clear_framebuffer(rt_ctx* ctx);
/* copies the ctx's own framebuffer of the last rt_render(…, NULL) to host */
int  rt_read_rgba_f32(rt_ctx* ctx, float* hostOut, size_t nFloats);
/* 8-bit sRGB-encoded RGBA of the same framebuffer (the reference's display format) */
int  rt_read_rgba8_srgb(rt_ctx* ctx, uint8_t* hostOut, size_t nBytes);

/* calculateIntersections for n caller-supplied rays (host arrays, 3 floats
 * each), for kernel-level parity tests. Synchronous. */
int  rt_trace_rays(rt_ctx* ctx, uint32_t n, const float* origins, const float* dirs, RtHit* hitsOut);

int  rt_get_counters(rt_ctx* ctx, RtCounters* out);
int  rt_reset_counters(rt_ctx* ctx);
/* When enabled, every traversal-kernel launch is bracketed by HIP events on
 * the launch stream; rt_get_trace_time_ms returns their summed duration and
 * launch count since the last reset (synchronises). */
int  rt_set_profiling(rt_ctx* ctx, int enabled);
int  rt_get_trace_time_ms(rt_ctx* ctx, double* msOut, uint64_t* launchesOut);
/* The time during which at least one bracketed traversal launch was running since profiling was switched on (the union
 * of the launches' spans: a dispatch of the multi-kernel pipeline runs in parts on several streams, whose launches
 * overlap, so the summed durations of rt_get_trace_time_ms can exceed the wall time). Synchronises. */
int  rt_get_trace_busy_ms(rt_ctx* ctx, double* msOut);
/* Performance knobs; none of them changes a pixel or a counter.
 *   "light_queries"  1 (default): the NEE ray and the cosine probe of a diffuse bounce, which only ask whether their
 *                    closest hit is emissive and how far it is (raytrace.comp:389-403,443-460), are answered from the list of
 *                    emissive primitives where that is possible and stop at the first hit that answers them; 0: every one of
 *                    them is a full closest-hit traversal. Same pixels either way.
 *   "camera_reuse"   1 (default): the samples of a pixel all start with the same camera ray (the shader does not jitter,
 *                    raytrace.comp:541-557,571-573), so its hit is kept from the first sample and the later samples of the
 *                    dispatch start from it without a traversal; 0: traced every time. Off for the debug heat maps.
 *   "pipeline"       -1 (default) pick by tile size, 0 = multi-kernel wavefront pipeline
 *                    (k_trace_pw + k_shade per round), 1 = wave-private fused pipeline
 *                    (k_render_fused: every wave runs the stages on its own 8x8 pixel blocks)
 *   "fused_below_pixels"  paths of a dispatch (tile pixels x frames) below which -1 picks the fused pipeline
 *   "fused_below_box_tests"  ... and box tests per ray (measured on the context's earlier dispatches of
 *                    the scene, copied back without waiting) below which it does so at any size
 *   "lanes"          multi-kernel pipeline: a dispatch of at least "lanes_min_kslots" x 1024 paths is rendered in this many
 *                    independent parts (contiguous ranges of path slots, each with its own queues and its own stream, forked
 *                    from and joined to the ctx stream), so that one part's shading kernel and the draining tail of its
 *                    traversal launch run under the other parts' traversal; 1..4; 0 (default) = 3, except one for
 *                    dispatches of >= 8 M paths of scenes whose long rays walk into many placed objects (they lose by it)
 *   "lane_grid_pct"  ... each part's traversal launch taking this share of the resident work-groups (10..100; 0 = default: 50 —
 *                    three parts oversubscribe the GPU 1.5 times, so a part in its shading kernel leaves no traversal slot
 *                    empty — and 40 for parts of fewer than 1.2 M paths).
 *                    The parts only overlap while their streams sit on different hardware queues: ROCm deals a process's
 *                    streams onto GPU_MAX_HW_QUEUES queues (default 4); a host with many streams of its own should start
 *                    with that variable raised (bench.py sets 8)
 *   "trace_variant"  0 = one ray per lane (k_trace), 1 = persistent waves (k_trace_pw)
 *   "refill", "mk_refill", "chunk", "w_setup", "w_leaf", "fast_lanes", "lds_stack", "blocks_per_cu",
 *   "tile_slots", "phase_stats": traversal scheduling details, see DESIGN.md
 *   "fast_share"     sixteenths of the lanes that hold a ray which suffice to skip the vote (with "fast_lanes" as the
 *                    upper limit; 0 = "fast_lanes" only)
 *   "scatter"        fused pipeline: a block is made of chunks of this many consecutive slots taken from all
 *                    over the tile (0 = a block is neighbouring pixels; -1, default = 4 when a wave gets at
 *                    most two blocks, else 0)
 *   "pixel_refill"   fused pipeline: free lanes at which a wave resolves its finished pixels and reserves new ones,
 *                    1..64 (64 = a block at a time); 0 (default) = 8 when the scene's rays are long (the measure the
 *                    pipeline choice uses), else 64
 *   "batch_pixels"   fused pipeline: pixels per wave-private block, 1..64; 0 (default) = chosen per
 *                    launch so the blocks divide evenly over the resident waves ("batch_fixed" is
 *                    the fixed cost per block, in pixel units, that the chooser assumes) */
int  rt_set_tuning(rt_ctx* ctx, const char* key, int value);
/* The traversal kernel instantiation the last launch used, spelled as a demangler prints it ("k_trace_pw<20, false, false,
 * false, false, 144, 5>", "k_render_fused<24, false, false, true>"): tests/test_instantiations.py forces every instantiation in
 * the library and compares it with the oracle. The string lives in the context. */
const char* rt_last_kernel(const rt_ctx* ctx);
/* parts (streams) the last dispatch of the multi-kernel pipeline ran in ("lanes") */
int  rt_last_parts(const rt_ctx* ctx);
/* pipeline the last rt_render used (0 or 1) */
int  rt_last_pipeline(const rt_ctx* ctx);
/* box tests per executed ray of the uploaded scene as this context measured them (the figure the launch parameters
 * follow); < 0 while unknown. Measured from the counters of earlier dispatches, or — before the first dispatch of
 * >= 8 M pixel-samples of a scene — by a probe dispatch of eight rows at one sample ("probe", 0 turns it off). */
double rt_ray_cost(const rt_ctx* ctx);
/* The reference's BVH builder on the GPU (csrc/bvh_build.hip.h): same nodes, same numbering, same triangle order as
 * the host builder of the scene half (a zero bound may differ in sign). Blocking. `seconds` (optional) = device time. */
int  rt_bvh_build(rt_ctx* ctx, const TrianglePoint* points, uint32_t pointCount, Triangle* triangles, float* centroids,
                  uint32_t count, uint32_t triIndex0, uint32_t nodeBase, BVHNode* nodesOut, uint32_t nodeCapacity,
                  uint32_t* nodeCountOut, uint32_t statsOut[3]);
int  rt_bvh_hook(void* ctx, const TrianglePoint* points, uint32_t pointCount, Triangle* triangles, float* centroids,
                 uint32_t count, uint32_t triIndex0, uint32_t nodeBase, BVHNode* nodesOut, uint32_t nodeCapacity,
                 uint32_t* nodeCountOut, uint32_t statsOut[3]);
/* milliseconds the last rt_bvh_build spent between its first upload and its last download */
double rt_bvh_last_build_ms(const rt_ctx* ctx);

/* ------------------------------------------------------------------ */
/* Several GPUs of one node: one process (and one ctx) per GPU            */
/* ------------------------------------------------------------------ */
/* The frame shards by image rows with no exchange while it renders (a pixel depends on its global index and the read-only
 * scene, raytrace.comp:563-564): rank r of N calls rt_render / rt_render_frames with row0 = r, rowStride = N into a strip of
 * its own, and one gather at the end brings the strips to one rank. These four calls are that gather, over RCCL (xGMI):
 *   rank 0:      rt_comm_unique_id(id), hands the RT_COMM_ID_BYTES to the other processes by any means it has
 *   every rank:  rt_comm_init(ctx, id, N, r)
 *   every rank:  rt_gather_strips(ctx, d_strip, W, H, root, d_frame)  — asynchronous on the ctx stream, rt_sync waits
 * RCCL is loaded on the first of these calls; a single-GPU host never needs it. */
enum { RT_COMM_ID_BYTES = 128 };   /* sizeof(ncclUniqueId) */
int  rt_comm_unique_id(void* idOut);
int  rt_comm_init(rt_ctx* ctx, const void* id, int nRanks, int rank);
int  rt_comm_destroy(rt_ctx* ctx);
/* d_strip: this rank's rows rank, rank + N, ... of a width x height RGBA fp32 frame, in that order (what rt_render wrote);
 * d_frame: the whole frame, on `root` only (NULL elsewhere). Every rank of the communicator must call it. */
int  rt_gather_strips(rt_ctx* ctx, const float* d_strip, uint32_t width, uint32_t height, int root, float* d_frame);
/* The second half of rt_gather_strips on its own, for a host that moves the strips by other means: d_strips holds the strips
 * of ranks 0 .. nRanks - 1 one after the other (rank r: its ceil((height - r) / nRanks) rows in order), d_frame receives the
 * frame's rows. Asynchronous on the ctx stream. (tests: the row arithmetic for heights that nRanks does not divide) */
int  rt_deinterleave_strips(rt_ctx* ctx, const float* d_strips, uint32_t width, uint32_t height, int nRanks, float* d_frame);
/* ... the same with host buffers (copied in and out; blocking) */
int  rt_deinterleave_strips_host(rt_ctx* ctx, const float* strips, uint32_t width, uint32_t height, int nRanks, float* frame);

/* device self-test of the deterministic-math build (must equal RT_SELFTEST_EXPECT) */
int  rt_device_selftest(rt_ctx* ctx, uint32_t* bitsOut);
uint32_t rt_host_selftest(void);
/* Raw values of every GLSL built-in the shader uses (include/rt_probe.h), evaluated on the device: n inputs of 32
 * floats, 64 floats out each. The tests compare them with independent numpy restatements of the GLSL 4.50 formulas. */
int  rt_device_math_probe(rt_ctx* ctx, uint32_t n, const float* in, float* out);
/* streaming-copy ceiling measured on this GPU (GB/s), quoted beside the 8 TB/s nominal */
int  rt_measure_copy_bandwidth(rt_ctx* ctx, size_t bytes, int iters, double* gbpsOut);

const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
