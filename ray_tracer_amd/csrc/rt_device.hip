// rt_device.hip — device half of the C ABI: context, uploads, the wavefront
// render loop. Replaces the Vulkan seam of the reference (copy_buffer /
// update_buffer / run_compute, src/vk_engine.cpp:1401-1475,1623-1676) with
// hipMalloc'd buffers and HIP launches on one stream per context.
//
// There is no CPU fallback anywhere in this file: without a HIP device
// rt_create fails and nothing else can be called.

#include "rt_kernels.hip.h"
#include "bvh_build.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct EventPair { hipEvent_t a, b; };
// device-side {index,triCount} of a mesh root, keyed by its reference node index
#define RT_MAX_LANES 4
#define RT_FRAMES_MAX_SLOTS (24ull << 20)   // default for the paths of one multi-frame dispatch (ten 1080p frames or three 4K frames: 5.8 GB of path state)
struct RootInfo { uint32_t idx, cnt; float lo[3], hi[3]; uint32_t triFirst, triTotal; };  // triTotal = ~0u: the mesh's triangles are not one contiguous range

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;

    // scene
    DevScene sc{};
    std::vector<DevBuf> sceneBufs;
    DevBuf texelBuf, texInfoBuf, triUVBuf;
    DevBuf objTreeBuf, objCostBuf;
    int objTreeMin = 48;    // rt_set_tuning("object_tree_min"): general-transform objects from which the object hierarchy is built (0 = never)
    DevBuf matBuf, sphereBuf, sphereMatBuf, objInvBuf, objFwdBuf, objMetaBuf, objBoxBuf, objSkipBuf, maskBoxBuf, emitBuf, emitPreBuf;
    // host copies of what the emitter list is derived from (rebuild_emitters)
    std::vector<RayMaterial> hostMats;
    std::vector<uint32_t> hostSphereMat, hostObjMat, hostObjRoot, hostObjSampler;
    DevBuf objAlphaBuf;     // per object: its material's alpha map and sampler (refresh_maps)
    // multi-GPU (rt_comm_*): this rank's RCCL communicator, a staging buffer on the gathering rank
    ncclComm_t comm = nullptr;
    int commRanks = 0, commRank = 0;
    DevBuf gatherBuf;
    int framesPerLaunch = 0; // rt_render_frames: most frames of a tile rendered by one launch (0 = as many as fit)
    int cameraReuse = 1;    // rt_set_tuning("camera_reuse", 0): trace the camera ray of every sample
    int lightQueries = 1;   // rt_set_tuning("light_queries", 0): trace every NEE ray and cosine probe in full
    uint32_t maxLeafDepth = 0;
    std::vector<uint32_t> nodeRemap;          // reference node index -> device node index
    std::vector<RootInfo> rootOf;             // per reference node; idx = ~0u unless a mesh root

    // path state
    DevBuf stateBuf, queueBuf, fbBuf, counterBuf, scratchBuf, overflowBuf, overflowBufSide[RT_MAX_LANES - 1];
    // Multi-kernel pipeline in `lanes` independent parts (render_impl): the paths of a dispatch are split into contiguous slot
    // ranges, each with its own queues, counters and stream, so that one part's k_shade and the tail of its k_trace_pw launch
    // run under the other part's traversal. Part 0 is the ctx stream itself; the others fork from it and join it.
    hipStream_t sideStream[RT_MAX_LANES - 1] = {};
    hipEvent_t forkEvent = nullptr, joinEvent[RT_MAX_LANES - 1] = {}, pollEventSide[RT_MAX_LANES - 1] = {};
    bool lanesSet = false;                // "lanes" given explicitly: no automatic fall-back to one part
    int lanes = 3;                        // rt_set_tuning("lanes", 1..RT_MAX_LANES)
    uint32_t lanesMinSlots = 1u << 20;    // dispatches of fewer paths than this stay in one part
    hipStream_t curStream = nullptr;      // the stream and counters of the part whose launches are being built (launch_trace)
    uint32_t* curCounts = nullptr;
    int curLane = 0;
    int lastParts = 1;                    // parts the last multi-kernel dispatch ran in (rt_last_parts)
    int curGridPct = 100;                 // share of the resident work-groups a k_trace_pw launch of the current part takes
    int laneGridPct = 0;                  // rt_set_tuning("lane_grid_pct"): that share while a dispatch runs in several parts (0 = by the parts' size: 50, 40 below 1.2 M paths per part)
    uint32_t capacity = 0;  // pixels the state buffers hold
    PathState ps{};
    Queues q{};
    uint32_t* hostCounts = nullptr;  // pinned, 4 words
    uint32_t fbPixels = 0;
    bool fbValid = false;

    // profiling of the traversal kernel
    bool profiling = false;
    std::vector<EventPair> evPool;
    size_t evUsed = 0;
    double traceMs = 0.0;
    uint64_t traceLaunches = 0;
    hipEvent_t profBase = nullptr;                         // recorded when profiling is switched on: the origin of traceSpans
    std::vector<std::pair<float, float>> traceSpans;       // [start, end) of every bracketed launch, ms since profBase (rt_get_trace_busy_ms)
    uint64_t traceLaunchesTotal = 0;

    // tuning (rt_set_tuning)
    int traceVariant = 1;   // 0 = one-ray-per-lane k_trace, 1 = persistent waves k_trace_pw
    int pipeline = -1;      // 0 = multi-kernel wavefront pipeline, 1 = wave-private fused pipeline (k_render_fused), -1 = by tile size
    int lastPipeline = 0;   // what the last rt_render used
    char lastKernel[96] = "";  // the traversal kernel instantiation of the last launch, as a demangler prints it (rt_last_kernel)
    uint32_t fusedBelowPixels = 4000000;  // auto: dispatches of fewer paths than this use the fused pipeline — scaled down to 1.5 M as the rays get longer
                                          // (render_impl: sizeLimit). Sponza, 8 spp, ms per step with 1 / 2 / 4 / 10 frames of 1080p in one dispatch
                                          // (round 3, multi-kernel in three parts): fused 113.8 / 109.4 / 107.4 / 105, multi-kernel 103.4 / 90.7 / 84.7 / 79.5
    uint32_t fusedBelowBoxTests = 70;     // auto: ... and so do scenes whose rays are short (EXECUTED box tests per ray, measured), with one exception (render_impl)
    // box tests per ray of this scene, from counter snapshots copied back asynchronously after each dispatch
    DevCounters* snap = nullptr;          // pinned
    hipEvent_t snapEvent = nullptr;
    hipEvent_t pollEvent = nullptr;       // multi-kernel pipeline: the active-path count on its way back (never waited for)
    bool snapPending = false;
    unsigned long long snapBox = 0, snapRays = 0;  // counters at the previous snapshot
    double boxPerRay = -1.0;              // < 0: not measured yet
    double segPerPath = -1.0;             // path segments per pixel sample of this scene, from the same snapshots (< 0: not measured yet)
    unsigned long long snapSeg = 0, snapPaths = 0;
    int refill = 8;         // k_trace_pw: idle lanes that trigger a refill
    bool refillMkSet = false, wSetupSet = false;  // given explicitly (else by the scene's ray length, launch_pw_t)
    int refillMk = 16;      // the same for k_trace_pw over the global queue when set by hand ("mk_refill"); automatic: 12 for long rays, 16 otherwise (launch_pw_t)
    int chunk = 256;        // k_trace_pw: most queue entries reserved per atomic
    int ldsStackCap = 24;   // k_trace_pw: LDS stack entries per lane (8, 16 or 24); deeper BVHs use the overflow buffer
    int fastLanes = 32;     // k_trace_pw: lanes at interior nodes that skip the full vote (4K Sponza: 24 -> 32 is -2 %, 1080p: equal)
    int wSetup = 16, wLeaf = 16; // k_trace_pw: vote weights in eighths (interior = 8); the set-up weight when set by hand ("mk_w_setup"), automatic: 32 for long rays, 16 otherwise
    int blocksPerCU = 0;    // k_trace_pw: 0 = occupancy query
    int numCUs = 256;
    int phaseStats = 0;     // diagnostic: k_trace_pw counts rounds / active lanes per phase
    int tileSlots = 1;      // slots follow 8x8 pixel blocks instead of rows
    bool wLeafSet = false;
    bool fastLanesSet = false;  // fast_lanes given explicitly: it then also applies to the fused pipeline
    int wSetupFused = 16, wLeafFused = 24;  // vote weights of the fused pipeline (short private lists: leaves and set-ups sooner)
    double bvhBuildMs = 0.0; // last rt_bvh_build
    int fastShare = 10;     // sixteenths of the live lanes that suffice to skip the vote (0 = fixed count only): -1..-2 % everywhere
    bool cull = false;      // kernels with the object-skipping code (CULL) for this scene
    int maskIdentity = 0;   // identity-transform objects in the rays' object masks too (rt_update_objects reads it)
    int scatter = -1;       // fused pipeline: blocks made of chunks of this many slots from all over the tile; 0 = neighbouring pixels; -1 = auto
    uint64_t framesMaxSlots = RT_FRAMES_MAX_SLOTS;  // most paths of one multi-frame dispatch (rt_render_frames)
    int hotPairs = 2;       // k_trace_pw: child pairs of the meshes' top levels from LDS. 0 = off, 1 = as many as fit beside the stacks of
                            // six work-groups per CU, 2 = of five (Sponza, 21-entry stacks, ten frames in flight: 90.4 / 90.6 / 88.7 ms per step)
    int pixelRefill = 0;    // fused pipeline: free lanes at which a wave reserves new pixels (64 = a block at a time, 0 = by ray length)
    int batchPixels = 0;    // fused pipeline: pixels per wave-private block (0 = chosen per launch)
    int batchFixed = 80;    // ... and the fixed part of a block's cost in the chooser, in pixel units
    int lastBatchPixels = 0;
    DevBuf probeBuf;        // framebuffer of the ray-cost probe
    int probe = 1;          // measure an unknown scene with a small dispatch before its first big one
    bool inProbe = false;
    DevBuf waveTimeBuf;     // phase_stats: per-wave start/end clocks of the last k_trace_pw launch
    size_t waveTimesCount = 0;
    bool pixStats = false;  // this dispatch needs per-pixel box/triangle counts (debug heat maps)

    int fail(const std::string& m) { error = m; return -1; }
    int hip(hipError_t e, const char* what) {
        if (e == hipSuccess) return 0;
        error = std::string(what) + ": " + hipGetErrorString(e);
        return -(int)e - 1000;
    }
};

#define RT_HIP(ctx, call)                                    \
    do {                                                     \
        int _rc = (ctx)->hip((call), #call);                 \
        if (_rc) return _rc;                                 \
    } while (0)

namespace {

int dev_alloc(rt_ctx* c, DevBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return 0;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) bytes = 256;
    RT_HIP(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return 0;
}
void dev_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
}

int upload(rt_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = dev_alloc(c, b, bytes);
    if (rc) return rc;
    if (bytes) RT_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));  // src is borrowed for the call only
    return 0;
}

void pack_materials(const RayMaterial* m, uint32_t n, std::vector<float4>& out) {
    out.resize((size_t)std::max(n, 1u) * 3);
    for (uint32_t i = 0; i < n; i++) {
        out[3 * i + 0] = make_float4(m[i].albedo[0], m[i].albedo[1], m[i].albedo[2], m[i].reflectance);
        out[3 * i + 1] = make_float4(m[i].emissionColor[0], m[i].emissionColor[1], m[i].emissionColor[2], m[i].emissionStrength);
        float ai;
        memcpy(&ai, &m[i].albedoIndex, 4);   // bits of the int; -1 = no texture
        float mi, bi;
        memcpy(&mi, &m[i].metalnessIndex, 4);
        memcpy(&bi, &m[i].bumpIndex, 4);
        out[3 * i + 2] = make_float4(m[i].ior, ai, mi, bi);
    }
}

// rows 0..2 of a column-major mat4
void rows_of(const float* m, float4* out) {
    for (int r = 0; r < 3; r++) out[r] = make_float4(m[0 + r], m[4 + r], m[8 + r], m[12 + r]);
}

int ensure_state(rt_ctx* c, uint32_t nPixels) {
    if (c->capacity >= nPixels && c->stateBuf.p) return 0;
    const size_t stride4 = (((size_t)nPixels * 16) + 255) & ~(size_t)255;  // bytes per float4 array
    const size_t stride1 = (((size_t)nPixels * 4) + 255) & ~(size_t)255;
    const int nF4 = 14, nU1 = 2;
    int rc = dev_alloc(c, c->stateBuf, stride4 * nF4 + stride1 * nU1);
    if (rc) return rc;
    PathState& ps = c->ps;
    ps.base = (float4*)c->stateBuf.p;
    ps.pitch = (uint32_t)(stride4 / 16);
    ps.pitchStat = (uint32_t)(stride1 / 4);

    // queues: 2 x active (n) + 2 x rays (3n) + 4 counters
    const size_t qa = (((size_t)nPixels * 4) + 255) & ~(size_t)255;
    const size_t qr = (((size_t)nPixels * 3 * 4) + 255) & ~(size_t)255;
    rc = dev_alloc(c, c->queueBuf, 2 * qa + 2 * qr + 256);
    if (rc) return rc;
    char* qb = (char*)c->queueBuf.p;
    c->q.active[0] = (uint32_t*)qb;
    c->q.active[1] = (uint32_t*)(qb + qa);
    c->q.rays[0] = (uint32_t*)(qb + 2 * qa);
    c->q.rays[1] = (uint32_t*)(qb + 2 * qa + qr);
    c->q.counts = (uint32_t*)(qb + 2 * qa + 2 * qr);
    c->capacity = nPixels;
    return 0;
}

template <int STACK, bool OVF, bool CULL>
int launch_pw_t(rt_ctx* c, uint32_t maxRays, const TraceArgs& ta) {
    // per-ray counters are only needed for the pixel heat maps (debug >= 0) and rt_trace_rays
    const bool pix = c->pixStats || ta.perRayBox;
    // top-level pairs from LDS (k_trace_pw<HOT>): what 160 KB of LDS per CU leave beside the stacks. hot_pairs 1: six work-groups
    // per CU, 2: five (more pairs, no spills at 96 registers)
    // (the overflow-stack kernel with 16 entries in LDS keeps six work-groups AND 120 pairs: deep BVHs, see launch_trace)
    constexpr int HOT6 = OVF ? (STACK == 16 ? 120 : 0) : STACK == 8 ? 192 : STACK == 16 ? 136 : STACK == 20 ? 72 : 0;
    constexpr int HOT5 = OVF ? 0 : STACK == 24 ? 80 : STACK == 20 ? 144 : 192;
    int hotMode = (!pix && !c->phaseStats && c->sc.hotNodes > 0) ? c->hotPairs : 0;
    if (hotMode == 1 && HOT6 == 0) hotMode = 2;   // (a 24-entry stack leaves no room at six work-groups)
    if (hotMode == 2 && HOT5 == 0) hotMode = HOT6 ? 1 : 0;   // (overflow-stack instantiations: the table only beside 16-entry stacks)
    int perCU = c->blocksPerCU;
    if (perCU <= 0) {
        hipError_t e;
        e = hipErrorInvalidValue;
        // (if constexpr: an instantiation the tables can never select is not compiled — every kernel in the library can be
        // launched, and tests/test_instantiations.py launches every one of them against the oracle)
        if constexpr (HOT6 > 0) { if (hotMode == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_pw<STACK, OVF, false, false, CULL, HOT6, 6>, RT_BLOCK, 0); }
        if constexpr (HOT5 > 0) { if (hotMode == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_pw<STACK, OVF, false, false, CULL, HOT5, 5>, RT_BLOCK, 0); }
        if (hotMode == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_pw<STACK, OVF, false, false, CULL>, RT_BLOCK, 0);
        if (e != hipSuccess || perCU <= 0) perCU = 4;
    }
    uint32_t resident = (uint32_t)perCU * (uint32_t)c->numCUs;
    // a part of a dispatch that runs beside the other parts' launches takes its share of the resident work-groups (curGridPct)
    uint32_t blocks = std::min((maxRays + RT_BLOCK - 1) / RT_BLOCK, std::max(1u, (uint32_t)((uint64_t)resident * (uint32_t)c->curGridPct / 100u)));
    uint32_t* overflow = nullptr;
    hipStream_t stream = c->curStream ? c->curStream : c->stream;
    uint32_t* laneCounts = c->curCounts ? c->curCounts : c->q.counts;
    if (OVF) {
        const size_t need = (size_t)(c->maxLeafDepth - STACK) * resident * RT_BLOCK * 4;
        DevBuf& ob = c->curLane ? c->overflowBufSide[c->curLane - 1] : c->overflowBuf;   // launches of different parts run at the same time
        int rc = dev_alloc(c, ob, need);
        if (rc) return rc;
        overflow = (uint32_t*)ob.p;
    }
    unsigned long long* waveTimes = nullptr;
    if (c->phaseStats == 1 || (c->phaseStats >= 2 && c->traceLaunchesTotal == (uint64_t)(c->phaseStats - 2))) {  // 1: the last launch's waves; 2 + k: launch k's (after rt_reset_counters)
        c->waveTimesCount = (size_t)blocks * (RT_BLOCK / RT_WAVE);
        int rc = dev_alloc(c, c->waveTimeBuf, c->waveTimesCount * 16);
        if (rc) return rc;
        waveTimes = (unsigned long long*)c->waveTimeBuf.p;
    }
    // Long rays (the measure the pipeline choice uses) want new rays sooner and their set-up served later: idle lanes re-armed at 12
    // instead of 16, set-up steps voted in at weight 32 instead of 16 (Sponza 81.0 -> 79.4 ms per step, C5 115.4 -> 114.1;
    // Cornell + bunny / + dragon, short rays: +2.5 / +3.5 % with the same, so they keep 16 / 16). Knobs set by hand win.
    const bool longRays = c->boxPerRay >= (double)c->fusedBelowBoxTests;
    const uint32_t refillMk = c->refillMkSet ? (uint32_t)c->refillMk : (longRays ? 12u : 16u);
    const uint32_t wSetup = c->wSetupSet ? (uint32_t)c->wSetup : (longRays ? 32u : 16u);
    TracePwArgs pa{ta.queue, ta.count, laneCounts + 4, refillMk, (uint32_t)c->chunk, wSetup, (uint32_t)c->wLeaf, (uint32_t)c->fastLanes, (uint32_t)c->fastShare,
                   ta.perRayBox, ta.perRayTri, ta.counters, (unsigned long long*)((char*)c->counterBuf.p + sizeof(DevCounters)), waveTimes, overflow, ta.countAux, ta.countAux2, ta.auxOffset};
    {
        const bool stats = hotMode == 0 && c->phaseStats, px = hotMode == 0 && (c->phaseStats || pix);
        snprintf(c->lastKernel, sizeof c->lastKernel, "k_trace_pw<%d, %s, %s, %s, %s, %d, %d>", STACK, OVF ? "true" : "false", px ? "true" : "false",
                 stats ? "true" : "false", CULL ? "true" : "false", hotMode == 1 ? HOT6 : hotMode == 2 ? HOT5 : 0, hotMode == 2 ? 5 : 6);
    }
    if (hotMode == 1) { if constexpr (HOT6 > 0) hipLaunchKernelGGL((k_trace_pw<STACK, OVF, false, false, CULL, HOT6, 6>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa); }
    else if (hotMode == 2) { if constexpr (HOT5 > 0) hipLaunchKernelGGL((k_trace_pw<STACK, OVF, false, false, CULL, HOT5, 5>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa); }
    else if (c->phaseStats) hipLaunchKernelGGL((k_trace_pw<STACK, OVF, true, true, CULL>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa);
    else if (pix) hipLaunchKernelGGL((k_trace_pw<STACK, OVF, true, false, CULL>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa);
    else hipLaunchKernelGGL((k_trace_pw<STACK, OVF, false, false, CULL>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa);
    return 0;
}

// Pixels per wave-private block of k_render_fused. A wave finishes its block's samples one after the
// other (the reference's RNG runs on from sample to sample of a pixel), so a tile is done when the wave
// with the most blocks is: nPixels/64 blocks rarely divide evenly over the resident waves (a 1/8-height
// 1080p tile is 4050 blocks for 5120 waves), and a slightly smaller block that gives every wave the
// same number of blocks shortens that critical path. Measured block time ~ (80 + pixels) (drain of the
// longest ray and the shading step do not shrink with the block); beyond two blocks per wave the
// dynamic hand-out evens the waves out by itself and whole 8x8 blocks are best.
uint32_t fused_batch_pixels(const rt_ctx* c, uint32_t nPixels, uint32_t waves, uint32_t evenBelow) {
    if (c->batchPixels > 0) return (uint32_t)std::min(c->batchPixels, (int)RT_WAVE);
    if (((uint64_t)nPixels + RT_WAVE - 1) / RT_WAVE > (uint64_t)evenBelow * waves) return RT_WAVE;
    uint32_t best = RT_WAVE;
    uint64_t bestCost = ~0ull;
    for (uint32_t b = RT_WAVE; b >= 16; b--) {
        const uint64_t nb = (nPixels + b - 1) / b;
        const uint64_t rounds = (nb + waves - 1) / waves;
        const uint64_t cost = rounds * (uint64_t)((uint32_t)c->batchFixed + b);
        if (cost < bestCost) { bestCost = cost; best = b; }
    }
    return best;
}

template <int STACK, bool OVF, bool CULL>
int launch_fused_t(rt_ctx* c, const FrameParams& fp, float4* fb) {
    int perCU = c->blocksPerCU;
    if (perCU <= 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_render_fused<STACK, OVF, false, CULL>, RT_BLOCK, 0) != hipSuccess || perCU <= 0) perCU = 4;
    }
    const uint32_t resident = (uint32_t)perCU * (uint32_t)c->numCUs;
    const uint32_t nSlots = fp.nFrames > 1u ? ((fp.nPixels + 63u) / 64u) * 64u * fp.nFrames : fp.nPixels;  // rt_render_frames: frames are more slots of the same tile (rt_kernels.hip.h: frame_slot)
    // Pixels are replaced as they finish when rays are long (Sponza -7 %, its 1/2 and 1/4 tiles -8 % and -11 %: the wave no
    // longer drains to its slowest pixel once per block) and when a wave gets fewer than five blocks (Cornell + bunny /
    // + dragon, rank 0's rows of 2 GPUs -3 %, of 4 GPUs -13 %); with short rays and many blocks per wave a block at a
    // time is 4-7 % faster (the full 1080p frame of Cornell, + bunny, + dragon)
    // (scenes whose paths end early — open scenes, most samples leave after a bounce or two: fewer than 2.5 segments per sample
    // against ~4 in a closed box — empty a block's lanes unevenly; there replacing pays up to eight blocks per wave:
    // tools/heuristics_table.py, 256 bunnies on a floor under the sky, one 1080p frame: 19.1 against 19.7 ms)
    const uint64_t fewBelow = (c->segPerPath >= 0.0 && c->segPerPath < 2.5) ? 8ull : 5ull;
    const bool fewBlocks = ((uint64_t)nSlots + RT_WAVE - 1) / RT_WAVE < fewBelow * resident * (RT_BLOCK / RT_WAVE);
    const uint32_t pixelRefill = c->pixelRefill > 0 ? (uint32_t)c->pixelRefill
                               : ((c->boxPerRay >= (double)c->fusedBelowBoxTests || fewBlocks) ? 8u : (uint32_t)RT_WAVE);
    // a wave that replaces its pixels one by one evens out by itself as soon as there is more than one block per wave
    const uint32_t evenBelow = pixelRefill < RT_WAVE ? 1u : 2u;
    uint32_t batchPixels = fused_batch_pixels(c, nSlots, resident * (RT_BLOCK / RT_WAVE), evenBelow);
    // With one or two blocks per wave (one, when pixels are replaced as they finish) the tile is done when the most expensive block is: blocks made of 4-slot chunks from
    // all over the tile cost about the same (-6 % on a 1/8-height 1080p tile); with more blocks per wave the dynamic
    // hand-out balances by itself and neighbouring pixels (shared cache lines, coherent rays) are 3-8 % faster.
    const uint32_t wavesResident = resident * (RT_BLOCK / RT_WAVE);
    const uint32_t g = fp.nFrames > 1u ? 0u : c->scatter >= 0 ? (uint32_t)c->scatter : ((((uint64_t)nSlots + RT_WAVE - 1) / RT_WAVE <= (uint64_t)evenBelow * wavesResident) ? 4u : 0u);
    if (g) batchPixels = std::min((uint32_t)RT_WAVE, (batchPixels + g - 1) / g * g);
    const uint32_t nBatches = g ? ((nSlots + g - 1) / g + batchPixels / g - 1) / (batchPixels / g) : (nSlots + batchPixels - 1) / batchPixels;
    const uint32_t blocks = std::max(1u, std::min((nBatches + (RT_BLOCK / RT_WAVE) - 1) / (RT_BLOCK / RT_WAVE), resident));
    uint32_t* overflow = nullptr;
    if (OVF) {
        int rc = dev_alloc(c, c->overflowBuf, (size_t)(c->maxLeafDepth - STACK) * resident * RT_BLOCK * 4);
        if (rc) return rc;
        overflow = (uint32_t*)c->overflowBuf.p;
    }
    RT_HIP(c, hipMemsetAsync(c->q.counts + 5, 0, 4, c->stream));
    // lanes at interior nodes that make the wave skip the vote: long rays (Sponza: 157 box tests per ray) want the interior step
    // to wait for more lanes (40: -4 %); 24 for short rays and until the scene is measured
    const uint32_t fastLanes = c->fastLanesSet ? (uint32_t)c->fastLanes : (c->boxPerRay >= (double)c->fusedBelowBoxTests ? 40u : 24u);
    const uint32_t wLeaf = (uint32_t)c->wLeafFused;
    unsigned long long* waveTimes = nullptr;
    if (c->phaseStats) {
        c->waveTimesCount = (size_t)blocks * (RT_BLOCK / RT_WAVE);
        int rc = dev_alloc(c, c->waveTimeBuf, c->waveTimesCount * 16);
        if (rc) return rc;
        waveTimes = (unsigned long long*)c->waveTimeBuf.p;
    }
    FusedArgs fa{c->q.counts + 5, fb, (DevCounters*)c->counterBuf.p, overflow, (uint32_t)c->refill, (uint32_t)c->wSetupFused, wLeaf, fastLanes, batchPixels, g, (uint32_t)c->fastShare, waveTimes, pixelRefill};
    c->lastBatchPixels = (int)batchPixels;
    const FusedKernArgs ka{c->sc, c->ps, fp, fa};
    snprintf(c->lastKernel, sizeof c->lastKernel, "k_render_fused<%d, %s, %s, %s>", STACK, OVF ? "true" : "false", c->pixStats ? "true" : "false", CULL ? "true" : "false");
    if (c->pixStats) hipLaunchKernelGGL((k_render_fused<STACK, OVF, true, CULL>), dim3(blocks), dim3(RT_BLOCK), 0, c->stream, ka);
    else hipLaunchKernelGGL((k_render_fused<STACK, OVF, false, CULL>), dim3(blocks), dim3(RT_BLOCK), 0, c->stream, ka);
    RT_HIP(c, hipGetLastError());
    return 0;
}

int launch_fused(rt_ctx* c, const FrameParams& fp, float4* fb) {
    EventPair* ev = nullptr;
    if (c->profiling) {
        if (c->evUsed == c->evPool.size()) {
            EventPair p;
            RT_HIP(c, hipEventCreate(&p.a));
            RT_HIP(c, hipEventCreate(&p.b));
            c->evPool.push_back(p);
        }
        ev = &c->evPool[c->evUsed++];
        RT_HIP(c, hipEventRecord(ev->a, c->stream));
    }
    const uint32_t d = c->maxLeafDepth, cap = (uint32_t)c->ldsStackCap;
    int rc;
    if (c->cull) {
        if (d <= 8) rc = launch_fused_t<8, false, true>(c, fp, fb);
        else if (cap < 16) rc = launch_fused_t<8, true, true>(c, fp, fb);
        else if (d <= 16) rc = launch_fused_t<16, false, true>(c, fp, fb);
        else if (cap < 24) rc = launch_fused_t<16, true, true>(c, fp, fb);
        else if (d <= 24) rc = launch_fused_t<24, false, true>(c, fp, fb);
        else rc = launch_fused_t<24, true, true>(c, fp, fb);
    }
    else if (d <= 8) rc = launch_fused_t<8, false, false>(c, fp, fb);
    else if (cap < 16) rc = launch_fused_t<8, true, false>(c, fp, fb);
    else if (d <= 16) rc = launch_fused_t<16, false, false>(c, fp, fb);
    else if (cap < 24) rc = launch_fused_t<16, true, false>(c, fp, fb);
    else if (d <= 24) rc = launch_fused_t<24, false, false>(c, fp, fb);
    else rc = launch_fused_t<24, true, false>(c, fp, fb);
    if (rc) return rc;
    if (ev) RT_HIP(c, hipEventRecord(ev->b, c->stream));
    c->traceLaunchesTotal++;
    return 0;
}

// k_trace_pw_alpha<PIX>: 24 stack entries in LDS and the overflow buffer behind them, object culling compiled in, no top-level
// table — one kernel for every scene that binds an alpha map (launch_pw_t's launch, without its choices)
int launch_pw_alpha(rt_ctx* c, uint32_t maxRays, const TraceArgs& ta) {
    const bool pix = c->pixStats || ta.perRayBox;
    int perCU = c->blocksPerCU;
    if (perCU <= 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_pw_alpha<false>, RT_BLOCK, 0) != hipSuccess || perCU <= 0)) perCU = 4;
    const uint32_t resident = (uint32_t)perCU * (uint32_t)c->numCUs;
    const uint32_t blocks = std::min((maxRays + RT_BLOCK - 1) / RT_BLOCK, std::max(1u, (uint32_t)((uint64_t)resident * (uint32_t)c->curGridPct / 100u)));
    hipStream_t stream = c->curStream ? c->curStream : c->stream;
    uint32_t* laneCounts = c->curCounts ? c->curCounts : c->q.counts;
    uint32_t* overflow = nullptr;
    if (c->maxLeafDepth > 24u) {
        const size_t need = (size_t)(c->maxLeafDepth - 24u) * resident * RT_BLOCK * 4;
        DevBuf& ob = c->curLane ? c->overflowBufSide[c->curLane - 1] : c->overflowBuf;
        int rc = dev_alloc(c, ob, need);
        if (rc) return rc;
        overflow = (uint32_t*)ob.p;
    }
    const bool longRays = c->boxPerRay >= (double)c->fusedBelowBoxTests;
    const uint32_t refillMk = c->refillMkSet ? (uint32_t)c->refillMk : (longRays ? 12u : 16u);
    const uint32_t wSetup = c->wSetupSet ? (uint32_t)c->wSetup : (longRays ? 32u : 16u);
    TracePwArgs pa{ta.queue, ta.count, laneCounts + 4, refillMk, (uint32_t)c->chunk, wSetup, (uint32_t)c->wLeaf, (uint32_t)c->fastLanes, (uint32_t)c->fastShare,
                   ta.perRayBox, ta.perRayTri, ta.counters, (unsigned long long*)((char*)c->counterBuf.p + sizeof(DevCounters)), nullptr, overflow, ta.countAux, ta.countAux2, ta.auxOffset};
    snprintf(c->lastKernel, sizeof c->lastKernel, "k_trace_pw_alpha<%s>", pix ? "true" : "false");
    if (pix) hipLaunchKernelGGL((k_trace_pw_alpha<true>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa);
    else hipLaunchKernelGGL((k_trace_pw_alpha<false>), dim3(blocks), dim3(RT_BLOCK), 0, stream, c->sc, c->ps, pa);
    return 0;
}

template <int STACK>
void launch_v0_t(rt_ctx* c, uint32_t maxRays, const TraceArgs& ta) {
    uint32_t blocks = (maxRays + RT_BLOCK - 1) / RT_BLOCK;
    snprintf(c->lastKernel, sizeof c->lastKernel, "k_trace<%d>", STACK);
    hipLaunchKernelGGL((k_trace<STACK>), dim3(blocks), dim3(RT_BLOCK), 0, c->curStream ? c->curStream : c->stream, c->sc, c->ps, ta);
}

// the work counter (counts[4]) must be zero when this is called
int launch_trace(rt_ctx* c, uint32_t maxRays, const TraceArgs& ta) {
    if (maxRays == 0) return 0;
    hipStream_t stream = c->curStream ? c->curStream : c->stream;
    EventPair* ev = nullptr;
    if (c->profiling) {
        if (c->evUsed == c->evPool.size()) {
            EventPair p;
            RT_HIP(c, hipEventCreate(&p.a));
            RT_HIP(c, hipEventCreate(&p.b));
            c->evPool.push_back(p);
        }
        ev = &c->evPool[c->evUsed++];
        RT_HIP(c, hipEventRecord(ev->a, stream));
    }
    const uint32_t d = c->maxLeafDepth;
    int rc = 0;
    if (c->sc.mapFlags & RT_MAP_ALPHA) {  // a bound alpha map: the one traversal kernel that reads it (any depth, any objects)
        rc = launch_pw_alpha(c, maxRays, ta);
    } else if (c->traceVariant == 0) {  // one ray per lane, whole stack in LDS
        if (d <= 8) launch_v0_t<8>(c, maxRays, ta);
        else if (d <= 16) launch_v0_t<16>(c, maxRays, ta);
        else if (d <= 24) launch_v0_t<24>(c, maxRays, ta);
        else if (d <= 32) launch_v0_t<32>(c, maxRays, ta);
        else if (d <= 48) launch_v0_t<48>(c, maxRays, ta);
        else launch_v0_t<64>(c, maxRays, ta);
    } else {  // persistent waves; at most 24 entries in LDS, deeper ones in the overflow buffer
        // BVHs deeper than 24: 16 entries in LDS, the rest in the overflow buffer (the stack only holds far siblings and is rarely
        // that deep), which leaves room for 120 top-level pairs beside six work-groups per CU: C5 at 4K 474 -> 468 ms per step,
        // flattened 471 -> 462, 1080p 117.5 -> 115.6 (Cornell + dragon: level)
        const bool tableWanted = c->hotPairs && c->sc.hotNodes > 0 && !c->phaseStats && !(c->pixStats || ta.perRayBox);  // (launch_pw_t's condition)
        const uint32_t cap = (d > 24u && c->ldsStackCap >= 24 && tableWanted) ? 16u : (uint32_t)c->ldsStackCap;
        if (c->cull) {
            if (d <= 8) rc = launch_pw_t<8, false, true>(c, maxRays, ta);
            else if (cap < 16) rc = launch_pw_t<8, true, true>(c, maxRays, ta);
            else if (d <= 16) rc = launch_pw_t<16, false, true>(c, maxRays, ta);
            else if (cap < 24) rc = launch_pw_t<16, true, true>(c, maxRays, ta);
            else if (d <= 20) rc = launch_pw_t<20, false, true>(c, maxRays, ta);
            else if (d <= 24) rc = launch_pw_t<24, false, true>(c, maxRays, ta);
            else rc = launch_pw_t<24, true, true>(c, maxRays, ta);
        }
        else if (d <= 8) rc = launch_pw_t<8, false, false>(c, maxRays, ta);
        else if (cap < 16) rc = launch_pw_t<8, true, false>(c, maxRays, ta);
        else if (d <= 16) rc = launch_pw_t<16, false, false>(c, maxRays, ta);
        else if (cap < 24) rc = launch_pw_t<16, true, false>(c, maxRays, ta);
        else if (d <= 20) rc = launch_pw_t<20, false, false>(c, maxRays, ta);
        else if (d <= 24) rc = launch_pw_t<24, false, false>(c, maxRays, ta);
        else rc = launch_pw_t<24, true, false>(c, maxRays, ta);
    }
    if (rc) return rc;
    RT_HIP(c, hipGetLastError());
    if (ev) RT_HIP(c, hipEventRecord(ev->b, stream));
    c->traceLaunchesTotal++;
    return 0;
}

void poll_ray_cost(rt_ctx* c);

// sum finished event pairs; the stream must be idle
int harvest_events(rt_ctx* c) {
    for (size_t i = 0; i < c->evUsed; i++) {
        float ms = 0.f;
        RT_HIP(c, hipEventElapsedTime(&ms, c->evPool[i].a, c->evPool[i].b));
        c->traceMs += ms;
        c->traceLaunches++;
        float t0 = 0.f;
        if (c->profBase && hipEventElapsedTime(&t0, c->profBase, c->evPool[i].a) == hipSuccess) c->traceSpans.emplace_back(t0, t0 + ms);
    }
    c->evUsed = 0;
    return 0;
}

// The emitter list of the light queries (rt_kernels.hip.h: emitter_min_t2): every triangle of every object whose material is
// emissive, and the emissive spheres. "Emissive" is what lightSamplePDF asks (raytrace.comp:392): emissionStrength != 0.
// The shortcut is only taken when it is cheap (at most RT_EMIT_MAX_TRIS triangles) and exact: the NEE term of a query that
// is answered "not emissive" is emission * 0, which is 0 only while every material's emissionColor * emissionStrength is finite.
// The metalness, alpha and bump maps (declared semantics, include/rt_det_math.h): which of them the uploaded scene binds at all
// (DevScene::mapFlags — the kernels that read them are separate ones, picked by these bits), and every object's alpha map for the
// traversal. Follows the texture table, the materials and the objects: called whenever one of the three is replaced.
int refresh_maps(rt_ctx* c) {
    const uint32_t n = (uint32_t)c->hostObjMat.size();
    auto bound = [&](int32_t index) { return index >= 0 && (uint32_t)index < c->sc.texCount; };
    uint32_t flags = 0;
    std::vector<uint32_t> oa(std::max(n, 1u), 0xffffffffu);
    for (uint32_t i = 0; i < n; i++) {
        if (c->hostObjMat[i] >= c->hostMats.size()) continue;
        const RayMaterial& m = c->hostMats[c->hostObjMat[i]];
        if (bound(m.metalnessIndex)) flags |= RT_MAP_METALNESS;
        if (bound(m.bumpIndex)) flags |= RT_MAP_BUMP;
        if (bound(m.alphaIndex)) { flags |= RT_MAP_ALPHA; oa[i] = (uint32_t)m.alphaIndex | (c->hostObjSampler[i] == 1u ? 0x100u : 0u); }
    }
    int rc = upload(c, c->objAlphaBuf, oa.data(), oa.size() * 4);
    if (rc) return rc;
    c->sc.objAlpha = (const uint32_t*)c->objAlphaBuf.p;
    c->sc.mapFlags = flags;
    return 0;
}

int rebuild_emitters(rt_ctx* c) {
    c->sc.emitCount = 0; c->sc.emitSphereMask = 0; c->sc.emitMode = 0;
    { int rc = refresh_maps(c); if (rc) return rc; }
    if (!c->lightQueries || c->hostMats.empty()) return 0;
    auto emissive = [&](uint32_t m) { return m < c->hostMats.size() && !(c->hostMats[m].emissionStrength == 0.f); };
    for (const RayMaterial& m : c->hostMats)
        for (int k = 0; k < 3; k++)
            if (!std::isfinite(m.emissionColor[k] * m.emissionStrength)) return 0;
    uint32_t mask = 0;
    for (size_t i = 0; i < c->hostSphereMat.size() && i < 32; i++)
        if (emissive(c->hostSphereMat[i])) mask |= 1u << i;
    for (size_t i = 32; i < c->hostSphereMat.size(); i++)
        if (emissive(c->hostSphereMat[i])) return 0;  // beyond the mask (the reference has ten spheres)
    std::vector<uint2> list;
    for (size_t i = 0; i < c->hostObjMat.size(); i++) {
        if (!emissive(c->hostObjMat[i])) continue;
        { const int32_t ai = c->hostMats[c->hostObjMat[i]].alphaIndex; if (ai >= 0 && (uint32_t)ai < c->sc.texCount) return 0; }  // an emitter with holes: its list entries would need the map
        const RootInfo& r = c->rootOf[c->hostObjRoot[i]];
        if (r.triTotal == 0xffffffffu || list.size() + r.triTotal > (size_t)RT_EMIT_MAX_TRIS) return 0;
        for (uint32_t t = 0; t < r.triTotal; t++) list.push_back(make_uint2((uint32_t)i, r.triFirst + t));
    }
    if (!list.empty()) {
        int rc = upload(c, c->emitBuf, list.data(), list.size() * sizeof(uint2));
        if (rc) return rc;
    } else {
        int rc = dev_alloc(c, c->emitBuf, 256);
        if (rc) return rc;
    }
    c->sc.emitTris = (const uint2*)c->emitBuf.p;
    {   // the ray-independent part of every listed triangle's test, computed on the device with the traversal's own operations
        int rc = dev_alloc(c, c->emitPreBuf, std::max<size_t>(list.size(), 1) * 4 * sizeof(float4));
        if (rc) return rc;
        if (!list.empty()) {
            hipLaunchKernelGGL(k_emit_precompute, dim3(((uint32_t)list.size() + 63u) / 64u), dim3(64), 0, c->stream, c->sc.triPos, c->sc.emitTris, (uint32_t)list.size(), (float4*)c->emitPreBuf.p);
            RT_HIP(c, hipGetLastError());
            RT_HIP(c, hipStreamSynchronize(c->stream));  // set-up time; the renders may go to another stream later (rt_set_stream)
        }
        c->sc.emitPre = (const float4*)c->emitPreBuf.p;
    }
    c->sc.emitCount = (uint32_t)list.size();
    c->sc.emitSphereMask = mask;
    c->sc.emitMode = 1;
    return 0;
}

}  // namespace

extern "C" {

int rt_device_count(int* out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (out) *out = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? 0 : -1;
}

int rt_create(int device, rt_ctx** out) {
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -2;  // no GPU: fail loudly, no fallback
    if (device < 0 || device >= n) return -3;
    if (hipSetDevice(device) != hipSuccess) return -4;
    rt_ctx* c = new rt_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess) { c->ownStream = nullptr; rt_destroy(c); return -5; }
    c->stream = c->ownStream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->numCUs = prop.multiProcessorCount;
    }
    // every failure below goes through rt_destroy, which releases whatever exists by then (stream, pinned buffers, event)
    if (hipHostMalloc((void**)&c->hostCounts, 64, hipHostMallocDefault) != hipSuccess) { c->hostCounts = nullptr; rt_destroy(c); return -6; }
    if (hipHostMalloc((void**)&c->snap, sizeof(DevCounters), hipHostMallocDefault) != hipSuccess) { c->snap = nullptr; rt_destroy(c); return -6; }
    if (hipEventCreateWithFlags(&c->snapEvent, hipEventDisableTiming) != hipSuccess) { c->snapEvent = nullptr; rt_destroy(c); return -6; }
    if (hipEventCreateWithFlags(&c->pollEvent, hipEventDisableTiming) != hipSuccess) { c->pollEvent = nullptr; rt_destroy(c); return -6; }
    if (dev_alloc(c, c->counterBuf, sizeof(DevCounters) + 256) != 0) { rt_destroy(c); return -7; }
    (void)hipMemsetAsync(c->counterBuf.p, 0, sizeof(DevCounters) + 256, c->stream);
    (void)hipStreamSynchronize(c->stream);
    *out = c;
    return 0;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& b : c->sceneBufs) dev_free(b);
    for (DevBuf* b : {&c->matBuf, &c->sphereBuf, &c->sphereMatBuf, &c->objInvBuf, &c->objFwdBuf, &c->objMetaBuf, &c->objBoxBuf, &c->objSkipBuf, &c->maskBoxBuf, &c->emitBuf, &c->emitPreBuf, &c->texelBuf, &c->texInfoBuf, &c->triUVBuf, &c->objTreeBuf, &c->objCostBuf, &c->objAlphaBuf, &c->stateBuf,
                      &c->queueBuf, &c->fbBuf, &c->counterBuf, &c->scratchBuf, &c->overflowBuf, &c->waveTimeBuf, &c->probeBuf})
        dev_free(*b);
    (void)rt_comm_destroy(c);
    dev_free(c->gatherBuf);
    for (auto& e : c->evPool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->hostCounts) (void)hipHostFree(c->hostCounts);
    if (c->snap) (void)hipHostFree(c->snap);
    if (c->snapEvent) (void)hipEventDestroy(c->snapEvent);
    if (c->pollEvent) (void)hipEventDestroy(c->pollEvent);
    if (c->forkEvent) (void)hipEventDestroy(c->forkEvent);
    if (c->profBase) (void)hipEventDestroy(c->profBase);
    for (int l = 0; l < RT_MAX_LANES - 1; l++) {
        if (c->joinEvent[l]) (void)hipEventDestroy(c->joinEvent[l]);
        if (c->pollEventSide[l]) (void)hipEventDestroy(c->pollEventSide[l]);
        if (c->sideStream[l]) (void)hipStreamDestroy(c->sideStream[l]);
        dev_free(c->overflowBufSide[l]);
    }
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->error.c_str() : "null ctx"; }

int rt_set_stream(rt_ctx* c, void* s) {
    if (!c) return -1;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->ownStream;
    return 0;
}

int rt_upload_textures(rt_ctx* c, const RtTexture* tex, uint32_t n) {
    if (!c || (!tex && n)) return -1;
    if (n > (uint32_t)RT_MAX_TEXTURES) return c->fail("more than RT_MAX_TEXTURES textures");
    RT_HIP(c, hipSetDevice(c->device));
    std::vector<uint4> info(std::max(n, 1u), make_uint4(0u, 1u, 1u, 0u));
    size_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!tex[i].rgba8 || tex[i].width == 0 || tex[i].height == 0) return c->fail("texture " + std::to_string(i) + " is empty");
        if ((uint64_t)tex[i].width * tex[i].height > (1ull << 28) || total + (size_t)tex[i].width * tex[i].height > 0xffffffffull) return c->fail("textures too large");
        info[i] = make_uint4((uint32_t)total, tex[i].width, tex[i].height, 0u);
        total += (size_t)tex[i].width * tex[i].height;
    }
    std::vector<uint32_t> texels(std::max<size_t>(total, 1), 0u);
    for (uint32_t i = 0; i < n; i++) memcpy(&texels[info[i].x], tex[i].rgba8, (size_t)tex[i].width * tex[i].height * 4);
    int rc = upload(c, c->texelBuf, texels.data(), texels.size() * 4);
    if (rc) return rc;
    if ((rc = upload(c, c->texInfoBuf, info.data(), info.size() * sizeof(uint4)))) return rc;
    c->sc.texels = (const uint32_t*)c->texelBuf.p;
    c->sc.texInfo = (const uint4*)c->texInfoBuf.p;
    c->sc.texCount = n;
    return rebuild_emitters(c);   // (which maps are bound follows the table's size; an emitter with an alpha map leaves the emitter list)
}

int rt_update_materials(rt_ctx* c, const RayMaterial* m, uint32_t n) {
    if (!c || (!m && n)) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    std::vector<float4> packed;
    pack_materials(m, n, packed);
    int rc = upload(c, c->matBuf, packed.data(), packed.size() * sizeof(float4));
    if (rc) return rc;
    c->sc.mats = (const float4*)c->matBuf.p;
    c->sc.materialCount = n;
    c->hostMats.assign(m, m + n);
    return rebuild_emitters(c);
}

int rt_update_spheres(rt_ctx* c, const Sphere* s, uint32_t n) {
    if (!c || (!s && n)) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    std::vector<float4> sp(std::max(n, 1u));
    std::vector<uint32_t> sm(std::max(n, 1u));
    for (uint32_t i = 0; i < n; i++) {
        sp[i] = make_float4(s[i].position[0], s[i].position[1], s[i].position[2], s[i].radius);
        sm[i] = s[i].materialIndex;
    }
    int rc = upload(c, c->sphereBuf, sp.data(), sp.size() * sizeof(float4));
    if (rc) return rc;
    rc = upload(c, c->sphereMatBuf, sm.data(), sm.size() * 4);
    if (rc) return rc;
    c->sc.spheres = (const float4*)c->sphereBuf.p;
    c->sc.sphereMat = (const uint32_t*)c->sphereMatBuf.p;
    c->sc.sphereCount = n;
    // spheres whose {center, radius} repeat an earlier sphere's bit for bit are not tested by the rays' creators (DevScene::sphereTestMask)
    c->sc.sphereTestMask = 0;
    for (uint32_t i = 0; i < std::min(n, 32u); i++) {
        bool repeat = false;
        for (uint32_t k = 0; k < i && !repeat; k++) repeat = memcmp(&sp[i], &sp[k], sizeof(float4)) == 0;
        if (!repeat) c->sc.sphereTestMask |= 1u << i;
    }
    c->hostSphereMat.assign(sm.begin(), sm.begin() + n);
    return rebuild_emitters(c);
}

// objects: inverse computed once on the host with the shared
// rt_mat4_inverse, root metadata resolved against the uploaded BVH.
int rt_update_objects(rt_ctx* c, const RenderObject* o, uint32_t n) {
    if (!c || (!o && n)) return -1;
    if (c->rootOf.empty() && n) return c->fail("rt_update_objects before rt_upload_scene");
    RT_HIP(c, hipSetDevice(c->device));
    std::vector<float4> inv((size_t)std::max(n, 1u) * 3), fwd((size_t)std::max(n, 1u) * 3);
    std::vector<uint4> meta(std::max(n, 1u));
    std::vector<float4> wbox((size_t)std::max(n, 1u) * 2, make_float4(0.f, 0.f, 0.f, 0.f));
    uint32_t nGeneral = 0;
    double minScale = 1e300;  // smallest size-and-position scale among the padded boxes
    for (uint32_t i = 0; i < n; i++) {
        float im[16];
        rt_mat4_inverse(o[i].transformMatrix, im);
        rows_of(im, &inv[3 * (size_t)i]);
        rows_of(o[i].transformMatrix, &fwd[3 * (size_t)i]);
        if (o[i].bvhIndex >= c->rootOf.size()) return c->fail("object.bvhIndex out of range");
        const RootInfo& r = c->rootOf[o[i].bvhIndex];
        if (r.idx == 0xffffffffu) return c->fail("object.bvhIndex does not point at a mesh root of the uploaded BVH");
        if (o[i].materialIndex >= std::max(c->sc.materialCount, 1u)) return c->fail("object.materialIndex out of range");
        // exact identity inverse: the traversal may reuse the world-space ray (k_trace_pw)
        static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        bool isIdent = true;
        for (int r4 = 0; r4 < 3; r4++) {
            const float4& q = inv[3 * (size_t)i + r4];
            const float qq[4] = {q.x, q.y, q.z, q.w};
            for (int k4 = 0; k4 < 4; k4++) isIdent = isIdent && (qq[k4] == ident[r4 * 4 + k4]);
        }
        // World-space box of the object, padded: a ray that cannot reach it before its current closest hit cannot reach the
        // object's root box in object space either, so the object is worth exactly the reference's two box tests on the root's
        // children (or the root leaf's triangle tests) and no set-up. Padding 1e-3 of the box's size and position: four
        // orders of magnitude above what the fp32 inverse and the two slab tests can disagree by.
        uint32_t boxOk = 0;
        nGeneral += isIdent ? 0u : 1u;
        if (!isIdent) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool finite = true;
            for (int corner = 0; corner < 8; corner++) {
                const double p[3] = {(corner & 1) ? r.hi[0] : r.lo[0], (corner & 2) ? r.hi[1] : r.lo[1], (corner & 4) ? r.hi[2] : r.lo[2]};
                for (int d = 0; d < 3; d++) {
                    const float* m = o[i].transformMatrix;
                    const double w = (double)m[0 * 4 + d] * p[0] + (double)m[1 * 4 + d] * p[1] + (double)m[2 * 4 + d] * p[2] + (double)m[3 * 4 + d];
                    finite = finite && std::isfinite(w);
                    lo[d] = std::min(lo[d], w); hi[d] = std::max(hi[d], w);
                }
            }
            for (int k4 = 0; k4 < 12; k4++) finite = finite && std::isfinite((&inv[3 * (size_t)i].x)[k4]);
            if (finite) {
                double pad = 1e-6;
                for (int d = 0; d < 3; d++) pad = std::max(pad, 1e-3 * std::max(hi[d] - lo[d], std::max(std::fabs(lo[d]), std::fabs(hi[d]))));
                minScale = std::min(minScale, pad * 1e3);
                wbox[2 * (size_t)i] = make_float4((float)(lo[0] - pad), (float)(lo[1] - pad), (float)(lo[2] - pad), 0.f);
                wbox[2 * (size_t)i + 1] = make_float4((float)(hi[0] + pad), (float)(hi[1] + pad), (float)(hi[2] + pad), 0.f);
                boxOk = 6u;  // the rays' creators may rule the object out as well (bit 2)
            }
        }
        if (isIdent && c->maskIdentity) {  // exact root box: the rays' creators could rule the object out with the traversal's own
            wbox[2 * (size_t)i] = make_float4(r.lo[0], r.lo[1], r.lo[2], 0.f);       // slab test. Off by default: measured, the rounds
            wbox[2 * (size_t)i + 1] = make_float4(r.hi[0], r.hi[1], r.hi[2], 0.f);   // this saves are the cheap ones (profiles/README.md)
            boxOk = 4u;
        }
        // bit 3: the matrix itself is exactly the identity as well (reconstruct_hit then applies neither matrix)
        bool fwdIdent = isIdent;
        for (int r4 = 0; r4 < 3; r4++) {
            const float4& q = fwd[3 * (size_t)i + r4];
            const float qq[4] = {q.x, q.y, q.z, q.w};
            for (int k4 = 0; k4 < 4; k4++) fwdIdent = fwdIdent && (qq[k4] == ident[r4 * 4 + k4]);
        }
        meta[i] = make_uint4(r.idx, r.cnt, o[i].materialIndex, (isIdent ? 1u : 0u) | boxOk | (fwdIdent ? 8u : 0u) | ((o[i].samplerIndex & 0xffffu) << 16));
        {   // flags and root triangle count ride in the box's w components (one fetch per object in the skipping loop)
            const uint32_t fl = meta[i].w & 0xffffu, cn = r.cnt;
            memcpy(&wbox[2 * (size_t)i].w, &fl, 4);
            memcpy(&wbox[2 * (size_t)i + 1].w, &cn, 4);
        }
    }
    int rc = upload(c, c->objInvBuf, inv.data(), inv.size() * sizeof(float4));
    if (rc) return rc;
    if ((rc = upload(c, c->objFwdBuf, fwd.data(), fwd.size() * sizeof(float4)))) return rc;
    if ((rc = upload(c, c->objMetaBuf, meta.data(), meta.size() * sizeof(uint4)))) return rc;
    if ((rc = upload(c, c->objBoxBuf, wbox.data(), wbox.size() * sizeof(float4)))) return rc;
    c->sc.objBox = (const float4*)c->objBoxBuf.p;
    // the objects a ray's creator tests for the ray's object mask, compact: {lo.xyz, position in the window} {hi.xyz, -}. The mask
    // has 32 bits; its window starts at the first object that can be ruled out at all, so that a scene like C5 (26 identity
    // groups, then sixteen placed dragons) has all its placed objects under the mask
    std::vector<float4> maskBox(64, make_float4(0.f, 0.f, 0.f, 0.f));
    c->sc.reachCount = 0;
    uint32_t maskBase = 0;
    while (maskBase < n && !(meta[maskBase].w & 4u)) maskBase++;
    if (maskBase >= n) maskBase = 0;
    c->sc.maskBase = maskBase;
    for (uint32_t i = maskBase; i < std::min(n, maskBase + 32u); i++)
        if (meta[i].w & 4u) {
            const uint32_t k = c->sc.reachCount++, w = i - maskBase;
            maskBox[2 * k] = wbox[2 * (size_t)i]; maskBox[2 * k + 1] = wbox[2 * (size_t)i + 1];
            memcpy(&maskBox[2 * k].w, &w, 4);
        }
    // one general-transform object among identity ones (Sponza's emitter) does not pay for either mechanism: measured +3 % and
    // +8..19 % on that scene; from two on they do (Cornell + model: -5..-13 %)
    // The padding dominates the rounding of the world-space slab test and of the object-space ray only while the ray's
    // origin is not much farther out than the objects are big: both errors grow like 6e-8 * |origin|. Rays
    // that start beyond 1e3 object scales take the reference's own route through every object.
    c->sc.cullOriginLimit = minScale < 1e300 ? (float)(minScale * 1e3) : 0.f;
    c->cull = nGeneral >= 2 || (c->maskIdentity && c->sc.reachCount);
    if (!c->cull) c->sc.reachCount = 0;
    if ((rc = upload(c, c->maskBoxBuf, maskBox.data(), maskBox.size() * sizeof(float4)))) return rc;
    c->sc.maskBox = (const float4*)c->maskBoxBuf.p;
    // what the reference spends on objects [0, i) when a ray misses them all: two box tests per interior root, the root's
    // triangles per leaf root. A run of skipped objects costs the difference of two entries (trace_wave: fetch_next_meta).
    std::vector<uint2> skipCost(33, make_uint2(0u, 0u));  // over the mask's window
    for (uint32_t w = 0; w < 32u; w++) {
        const uint32_t i = maskBase + w;
        skipCost[w + 1] = skipCost[w];
        if (i < n) {
            if (meta[i].y == 0u) skipCost[w + 1].x += 2u;
            else skipCost[w + 1].y += meta[i].y;
        }
    }
    if ((rc = upload(c, c->objSkipBuf, skipCost.data(), skipCost.size() * sizeof(uint2)))) return rc;
    c->sc.objSkipCost = (const uint2*)c->objSkipBuf.p;
    {   // the hierarchy over runs of general-transform objects with a padded box (rt_kernels.hip.h: DevScene::objTree), for scenes with
        // many of them: measured on 256 separated bunny instances (tools/heuristics_table.py) against 7 % lost on C5's sixteen
        // overlapping dragons, which stay below the threshold (and inside the rays' object masks anyway)
        std::vector<float4> tree;
        std::vector<uint2> cost((size_t)n + 1, make_uint2(0u, 0u));
        for (uint32_t i = 0; i < n; i++) {
            cost[i + 1] = cost[i];
            if (meta[i].y == 0u) cost[i + 1].x += 2u; else cost[i + 1].y += meta[i].y;
        }
        uint32_t off[RT_OBJTREE_LEVELS + 1] = {};
        uint32_t levels = 0;
        if (c->objTreeMin > 0 && nGeneral >= (uint32_t)c->objTreeMin) {
            auto skippable = [&](uint32_t i) { return i < n && (meta[i].w & 3u) == 2u; };
            for (uint32_t k = 1; k <= (uint32_t)RT_OBJTREE_LEVELS; k++) {
                off[k] = (uint32_t)(tree.size() / 2);
                const uint32_t nb = (n + (1u << k) - 1) >> k;
                for (uint32_t b = 0; b < nb; b++) {
                    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                    bool ok = true;
                    for (uint32_t i = b << k; i < ((b + 1) << k); i++) {
                        if (!skippable(i)) { ok = false; break; }
                        const float4 &l = wbox[2 * (size_t)i], &h = wbox[2 * (size_t)i + 1];
                        lo[0] = std::min(lo[0], l.x); lo[1] = std::min(lo[1], l.y); lo[2] = std::min(lo[2], l.z);
                        hi[0] = std::max(hi[0], h.x); hi[1] = std::max(hi[1], h.y); hi[2] = std::max(hi[2], h.z);
                    }
                    tree.push_back(ok ? make_float4(lo[0], lo[1], lo[2], 1.f) : make_float4(0.f, 0.f, 0.f, 0.f));
                    tree.push_back(ok ? make_float4(hi[0], hi[1], hi[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
                }
            }
            levels = (uint32_t)RT_OBJTREE_LEVELS;
            if (getenv("RT_DEBUG_OBJTREE")) {
                for (uint32_t k = 1; k <= levels; k++) {
                    const uint32_t nb = (n + (1u << k) - 1) >> k;
                    uint32_t valid = 0;
                    for (uint32_t b = 0; b < nb; b++) valid += tree[2 * (size_t)(off[k] + b)].w != 0.f;
                    const float4 &l = tree[2 * (size_t)(off[k] + std::min(1u, nb - 1))], &h = tree[2 * (size_t)(off[k] + std::min(1u, nb - 1)) + 1];
                    fprintf(stderr, "[objtree] level %u: %u blocks, %u valid; block 1: (%g %g %g)-(%g %g %g)\n", k, nb, valid, l.x, l.y, l.z, h.x, h.y, h.z);
                }
                for (uint32_t i = 0; i < std::min(n, 6u); i++)
                    fprintf(stderr, "[objtree] object %u flags %x box (%g %g %g)-(%g %g %g)\n", i, meta[i].w, wbox[2 * i].x, wbox[2 * i].y, wbox[2 * i].z, wbox[2 * i + 1].x, wbox[2 * i + 1].y, wbox[2 * i + 1].z);
            }
        }
        if (tree.empty()) tree.assign(2, make_float4(0.f, 0.f, 0.f, 0.f));
        if ((rc = upload(c, c->objTreeBuf, tree.data(), tree.size() * sizeof(float4)))) return rc;
        if ((rc = upload(c, c->objCostBuf, cost.data(), cost.size() * sizeof(uint2)))) return rc;
        c->sc.objTree = (const float4*)c->objTreeBuf.p;
        c->sc.objCost = (const uint2*)c->objCostBuf.p;
        for (int k = 0; k <= RT_OBJTREE_LEVELS; k++) c->sc.objTreeOff[k] = off[k];
        c->sc.objTreeLevels = levels;
    }
    c->sc.objInv = (const float4*)c->objInvBuf.p;
    c->sc.objFwd = (const float4*)c->objFwdBuf.p;
    c->sc.objMeta = (const uint4*)c->objMetaBuf.p;
    c->sc.objectCount = n;
    c->hostObjMat.resize(n); c->hostObjRoot.resize(n); c->hostObjSampler.resize(n);
    for (uint32_t i = 0; i < n; i++) { c->hostObjMat[i] = o[i].materialIndex; c->hostObjRoot[i] = o[i].bvhIndex; c->hostObjSampler[i] = o[i].samplerIndex; }
    return rebuild_emitters(c);
}

}  // extern "C"

extern "C" {

int rt_upload_scene(rt_ctx* c, const RtSceneArrays* s) {
    if (!c || !s) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->snapPending) { c->snapBox = c->snap->boxTests; c->snapRays = c->snap->raysTraced; c->snapPending = false; }
    c->boxPerRay = -1.0;  // a new scene: its ray cost is not known yet
    c->hostObjMat.clear(); c->hostObjRoot.clear(); c->hostObjSampler.clear(); c->hostSphereMat.clear(); c->hostMats.clear();
    c->sc.emitMode = 0; c->sc.emitCount = 0; c->sc.emitSphereMask = 0;
    c->sc.mapFlags = 0;
    c->sc.texCount = 0;  // texture slots belong to the scene's materials: rt_upload_textures follows a new scene
    const uint32_t nNodes = s->bvhNodeCount, nTris = s->triangleCount;

    // ---- mesh segmentation: every distinct object.bvhIndex starts a mesh
    std::vector<uint32_t> roots;
    for (uint32_t i = 0; i < s->objectCount; i++) {
        if (s->objects[i].bvhIndex >= nNodes) return c->fail("object.bvhIndex out of range");
        roots.push_back(s->objects[i].bvhIndex);
    }
    std::sort(roots.begin(), roots.end());
    roots.erase(std::unique(roots.begin(), roots.end()), roots.end());

    // ---- device node numbering: shift each mesh so its child pairs (which
    // follow the root in twos) start on an even index = 64-byte boundary
    // The child pairs of the top levels of every mesh come first (breadth first over all roots, level by level, at most
    // RT_HOT_PAIRS of them): k_trace_pw<HOT> keeps exactly those in LDS (DevScene::hotNodes).
    c->nodeRemap.assign(nNodes, 0xffffffffu);
    uint32_t devCount = 0;
    {
        std::vector<uint32_t> level, next;
        for (uint32_t root : roots) level.push_back(root);
        uint32_t hot = 0;
        for (int depth = 0; depth < 8 && !level.empty() && hot < RT_HOT_PAIRS; depth++) {
            next.clear();
            for (uint32_t nidx : level) {
                const BVHNode& b = s->bvhNodes[nidx];
                if (b.triCount != 0 || hot >= RT_HOT_PAIRS) continue;
                if (b.index + 1 >= nNodes) return c->fail("BVH child index out of range");
                if (c->nodeRemap[b.index] != 0xffffffffu) continue;  // (a malformed BVH that shares children)
                c->nodeRemap[b.index] = 2 * hot;
                c->nodeRemap[b.index + 1] = 2 * hot + 1;
                hot++;
                next.push_back(b.index);
                next.push_back(b.index + 1);
            }
            level.swap(next);
        }
        c->sc.hotNodes = 2 * hot;
        uint32_t pos = 2 * hot;
        size_t r = 0;
        for (uint32_t nidx = 0; nidx < nNodes; nidx++) {
            const bool isRoot = r < roots.size() && roots[r] == nidx;
            if (isRoot) r++;
            if (c->nodeRemap[nidx] != 0xffffffffu) continue;  // a hot pair's node
            if (isRoot && (pos & 1u) == 0u) pos++;             // root on an odd slot: the pairs that follow it in twos start even
            c->nodeRemap[nidx] = pos++;
        }
        devCount = pos;
    }

    std::vector<float4> nodes((size_t)std::max(devCount, 2u) * 2, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint32_t> leafFirst(std::max(devCount, 1u), 0u);
    // W0 of a node: child-pair index (interior) or the leaf reference the kernels push and decode
    auto node_word = [&](uint32_t nidx) -> uint32_t {
        const BVHNode& b = s->bvhNodes[nidx];
        if (b.triCount == 0) return c->nodeRemap[b.index];
        if (b.triCount <= 7u && b.index + b.triCount <= 0x0ffffff0u) return RT_LEAF_BIT | (b.triCount << RT_LEAF_CNT_SHIFT) | b.index;
        return RT_LEAF_BIT | c->nodeRemap[nidx];
    };
    if (devCount > 0x0ffffff0u) return c->fail("BVH too large (node index needs more than 28 bits)");
    for (uint32_t nidx = 0; nidx < nNodes; nidx++) {
        const BVHNode& b = s->bvhNodes[nidx];
        if (b.triCount == 0) {
            if (b.index + 1 >= nNodes) return c->fail("BVH child index out of range");
            if (c->nodeRemap[b.index] & 1u) return c->fail("internal: child pair not 64-byte aligned (BVH not built in pairs)");
        } else if ((uint64_t)b.index + b.triCount > nTris) {
            return c->fail("BVH leaf triangle range out of bounds");
        }
        const uint32_t w0 = node_word(nidx);
        float4 lo = make_float4(b.boundsX[0], b.boundsY[0], b.boundsZ[0], 0.f);
        float4 hi = make_float4(b.boundsX[1], b.boundsY[1], b.boundsZ[1], 0.f);
        memcpy(&lo.w, &w0, 4);
        memcpy(&hi.w, &b.triCount, 4);
        nodes[2 * (size_t)c->nodeRemap[nidx]] = lo;
        nodes[2 * (size_t)c->nodeRemap[nidx] + 1] = hi;
        leafFirst[c->nodeRemap[nidx]] = b.triCount ? b.index : 0u;
    }

    // ---- the same child pairs, interleaved for packed fp32 math (k_trace_pw / k_render_fused):
    // pair at even node index p -> 4 float4 at 2p: {L.minx L.miny L.maxx L.maxy} {R.minx R.miny R.maxx R.maxy}
    // {L.minz L.maxz R.minz R.maxz} {L.W0 R.W0 - -}
    std::vector<float4> nodesPk(nodes.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t p = 0; p + 1 < (size_t)devCount; p += 2) {
        const float4 lo1 = nodes[2 * p], hi1 = nodes[2 * p + 1], lo2 = nodes[2 * p + 2], hi2 = nodes[2 * p + 3];
        nodesPk[2 * p] = make_float4(lo1.x, lo1.y, hi1.x, hi1.y);
        nodesPk[2 * p + 1] = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
        nodesPk[2 * p + 2] = make_float4(lo1.z, hi1.z, lo2.z, hi2.z);
        nodesPk[2 * p + 3] = make_float4(lo1.w, lo2.w, 0.f, 0.f);
    }

    // ---- deepest leaf per mesh decides the LDS stack size
    std::vector<RootInfo>& rootOf = c->rootOf;
    rootOf.assign(nNodes, RootInfo{0xffffffffu, 0, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0u, 0u});
    uint32_t maxDepth = 0;
    {
        std::vector<std::pair<uint32_t, uint32_t>> st;
        for (uint32_t root : roots) {
            const BVHNode& rb = s->bvhNodes[root];
            rootOf[root] = RootInfo{node_word(root), rb.triCount, {rb.boundsX[0], rb.boundsY[0], rb.boundsZ[0]}, {rb.boundsX[1], rb.boundsY[1], rb.boundsZ[1]}, 0u, 0u};
            st.clear();
            st.emplace_back(root, 0u);
            size_t visited = 0;
            uint64_t triLo = ~0ull, triHi = 0, triSum = 0;  // the mesh's triangles: the leaves' ranges, one contiguous run as the builder leaves them
            while (!st.empty()) {
                auto [nidx, d] = st.back();
                st.pop_back();
                if (++visited > (size_t)nNodes + 1) return c->fail("BVH has a cycle");
                const BVHNode& b = s->bvhNodes[nidx];
                if (b.triCount) {
                    maxDepth = std::max(maxDepth, d);
                    triLo = std::min<uint64_t>(triLo, b.index); triHi = std::max<uint64_t>(triHi, (uint64_t)b.index + b.triCount); triSum += b.triCount;
                    continue;
                }
                st.emplace_back(b.index, d + 1);
                st.emplace_back(b.index + 1, d + 1);
            }
            rootOf[root].triFirst = (uint32_t)triLo;
            rootOf[root].triTotal = (triSum == triHi - triLo) ? (uint32_t)triSum : 0xffffffffu;
        }
    }
    if (maxDepth > 64) return c->fail("BVH deeper than 64 levels (the reference's builder caps at 64)");
    c->maxLeafDepth = maxDepth;

    // ---- triangles: positions hot, normals cold, both in the reference's order
    std::vector<float4> tpos((size_t)std::max(nTris, 1u) * 3), tnrm((size_t)std::max(nTris, 1u) * 3);
    for (uint32_t t = 0; t < nTris; t++) {
        const Triangle& tr = s->triangles[t];
        const uint32_t vi[3] = {tr.v0, tr.v1, tr.v2};
        for (int k = 0; k < 3; k++) {
            if (vi[k] >= s->triPointCount) return c->fail("triangle point index out of range");
            const TrianglePoint& p = s->triPoints[vi[k]];
            float w = 0.f;
            if (k == 0) { uint32_t fo = tr.frontOnly ? 1u : 0u; memcpy(&w, &fo, 4); }
            tpos[3 * (size_t)t + k] = make_float4(p.position[0], p.position[1], p.position[2], w);
            tnrm[3 * (size_t)t + k] = make_float4(p.normal[0], p.normal[1], p.normal[2], 0.f);
        }
    }

    {   // vertex uvs (TrianglePoint: u in position.w, v in normal.w, src/vk_engine.h:64-67), 2 x float4 per triangle
        std::vector<float4> tuv((size_t)std::max(nTris, 1u) * 2, make_float4(0.f, 0.f, 0.f, 0.f));
        for (uint32_t t = 0; t < nTris; t++) {
            const Triangle& tr = s->triangles[t];
            const TrianglePoint &p0 = s->triPoints[tr.v0], &p1 = s->triPoints[tr.v1], &p2 = s->triPoints[tr.v2];
            tuv[2 * (size_t)t] = make_float4(p0.position[3], p0.normal[3], p1.position[3], p1.normal[3]);
            tuv[2 * (size_t)t + 1] = make_float4(p2.position[3], p2.normal[3], 0.f, 0.f);
        }
        int rcu = upload(c, c->triUVBuf, tuv.data(), tuv.size() * sizeof(float4));
        if (rcu) return rcu;
        c->sc.triUV = (const float4*)c->triUVBuf.p;
    }
    for (auto& b : c->sceneBufs) dev_free(b);
    c->sceneBufs.assign(5, DevBuf());
    int rc;
    if ((rc = upload(c, c->sceneBufs[0], nodes.data(), nodes.size() * sizeof(float4)))) return rc;
    if ((rc = upload(c, c->sceneBufs[1], tpos.data(), tpos.size() * sizeof(float4)))) return rc;
    if ((rc = upload(c, c->sceneBufs[2], tnrm.data(), tnrm.size() * sizeof(float4)))) return rc;
    c->sc.nodes = (const float4*)c->sceneBufs[0].p;
    c->sc.triPos = (const float4*)c->sceneBufs[1].p;
    c->sc.triNrm = (const float4*)c->sceneBufs[2].p;
    if ((rc = upload(c, c->sceneBufs[3], leafFirst.data(), leafFirst.size() * 4))) return rc;
    c->sc.leafFirst = (const uint32_t*)c->sceneBufs[3].p;
    if ((rc = upload(c, c->sceneBufs[4], nodesPk.data(), nodesPk.size() * sizeof(float4)))) return rc;
    c->sc.nodesPk = (const float4*)c->sceneBufs[4].p;
    c->sc.nodeCount = devCount;
    c->sc.triCount = nTris;

    if (s->materialCount == 0) return c->fail("scene needs at least one material");
    if ((rc = rt_update_materials(c, s->materials, s->materialCount))) return rc;
    for (uint32_t i = 0; i < s->sphereCount; i++)
        if (s->spheres[i].materialIndex >= s->materialCount) return c->fail("sphere.materialIndex out of range");
    if ((rc = rt_update_spheres(c, s->spheres, s->sphereCount))) return rc;
    return rt_update_objects(c, s->objects, s->objectCount);
}

int rt_sync(rt_ctx* c) {
    if (!c) return -1;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return harvest_events(c);
}

}  // extern "C"

namespace {
// Fold in the counter snapshot of an earlier dispatch if its copy has arrived (never waits).
void poll_ray_cost(rt_ctx* c) {
    if (!c->snapPending || hipEventQuery(c->snapEvent) != hipSuccess) return;
    c->snapPending = false;
    const unsigned long long box = c->snap->boxTests - c->snap->skippedBoxTests, rays = c->snap->raysTraced;  // executed tests: what a ray costs the GPU
    if (rays > c->snapRays && box >= c->snapBox && rays - c->snapRays > 100000ull)
        c->boxPerRay = (double)(box - c->snapBox) / (double)(rays - c->snapRays);
    const unsigned long long seg = c->snap->segments, paths = c->snap->paths;
    if (paths > c->snapPaths && seg >= c->snapSeg && paths - c->snapPaths > 100000ull) c->segPerPath = (double)(seg - c->snapSeg) / (double)(paths - c->snapPaths);
    c->snapBox = box;
    c->snapRays = rays;
    c->snapSeg = seg;
    c->snapPaths = paths;
}
// Queue the next snapshot behind the dispatch just enqueued.
void request_ray_cost(rt_ctx* c) {
    if (c->snapPending) return;
    if (hipMemcpyAsync(c->snap, c->counterBuf.p, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return;
    if (hipEventRecord(c->snapEvent, c->stream) != hipSuccess) return;
    c->snapPending = true;
}

// The launch parameters of both pipelines follow the scene's measured box tests per ray, which the first dispatch of a
// scene does not have — and a single-render job (the reference's singleRender mode: all samples in one dispatch) is
// nothing but a first dispatch. Before a big one, eight rows of the same tile are rendered once with one sample per
// pixel into a scratch image and the counters put back: a few ms, no trace in anything the caller can read.
int probe_ray_cost(rt_ctx* c, const PushConstants* pc, uint32_t width, uint32_t height, uint32_t row0, uint32_t rowStride, uint32_t nRows) {
    const uint32_t rows = std::min(nRows, 8u), skip = nRows / rows;
    PushConstants p = *pc;
    p.rayTraceParams.singleRender = 1;
    p.rayTraceParams.sampleLimit = 1;
    p.rayTraceParams.progressive = 0;
    p.rayTraceParams.debug = -1;
    int rc = dev_alloc(c, c->probeBuf, (size_t)rows * width * sizeof(float4));
    if (rc) return rc;
    DevCounters before, after;
    RT_HIP(c, hipMemcpyAsync(&before, c->counterBuf.p, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    const int pipeline = c->pipeline, lastPipeline = c->lastPipeline, lastBatch = c->lastBatchPixels;
    const bool profiling = c->profiling; const int phaseStats = c->phaseStats;
    const uint64_t launches = c->traceLaunchesTotal;
    c->pipeline = 1; c->profiling = false; c->phaseStats = 0; c->inProbe = true;
    rc = rt_render(c, &p, width, height, row0 + (skip / 2u) * rowStride, rowStride * skip, rows, (float*)c->probeBuf.p);
    c->pipeline = pipeline; c->profiling = profiling; c->phaseStats = phaseStats; c->inProbe = false;
    c->lastPipeline = lastPipeline; c->lastBatchPixels = lastBatch; c->traceLaunchesTotal = launches;
    if (rc) return rc;
    RT_HIP(c, hipMemcpyAsync(&after, c->counterBuf.p, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpyAsync(c->counterBuf.p, &before, sizeof(DevCounters), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->snapPending = false;  // the probe's own snapshot request: its copy has arrived, and it is not wanted
    if (after.raysTraced > before.raysTraced + 1000ull)
        c->boxPerRay = (double)((after.boxTests - after.skippedBoxTests) - (before.boxTests - before.skippedBoxTests)) / (double)(after.raysTraced - before.raysTraced);
    return 0;
}
}  // namespace

extern "C" {

}  // extern "C"

namespace {
// rt_render (nFrames = 1) and rt_render_frames
int render_impl(rt_ctx* c, const PushConstants* pc, uint32_t width, uint32_t height, uint32_t row0, uint32_t rowStride,
                uint32_t nRows, uint32_t nFrames, float* d_rgba) {
    if (!c || !pc) return -1;
    if (width == 0 || height == 0 || rowStride == 0) return c->fail("rt_render: bad image geometry");
    if (nRows && (uint64_t)row0 + (uint64_t)(nRows - 1) * rowStride >= height) return c->fail("rt_render: rows exceed the image");
    if (!c->sc.nodes) return c->fail("rt_render before rt_upload_scene");
    const RayTracerData& td = pc->rayTraceParams;
    if (td.sphereCount > c->sc.sphereCount) return c->fail("rayTraceParams.sphereCount exceeds the uploaded spheres");
    if (td.objectCount > c->sc.objectCount) return c->fail("rayTraceParams.objectCount exceeds the uploaded objects");
    if (td.bounceLimit >= (1u << 28) - 1u) return c->fail("rayTraceParams.bounceLimit needs more than 28 bits");
    const uint64_t np64 = (uint64_t)nRows * width;
    if ((np64 + 63) / 64 * 64 * nFrames >= (1ull << 30)) return c->fail("tile too large (slot ids are 30 bits)");
    const uint32_t nPixels = (uint32_t)np64;
    RT_HIP(c, hipSetDevice(c->device));
    if (nPixels == 0) return 0;

    int rc = ensure_state(c, nFrames > 1u ? (nPixels + 63u) / 64u * 64u * nFrames : nPixels);
    if (rc) return rc;
    float4* fb = (float4*)d_rgba;
    if (!fb) {
        const bool fresh = !c->fbBuf.p || c->fbPixels != nPixels;
        if ((rc = dev_alloc(c, c->fbBuf, (size_t)nPixels * sizeof(float4)))) return rc;
        if (fresh) RT_HIP(c, hipMemsetAsync(c->fbBuf.p, 0, (size_t)nPixels * sizeof(float4), c->stream));
        c->fbPixels = nPixels;
        fb = (float4*)c->fbBuf.p;
        c->fbValid = true;
    }

    // ---- per-frame constants (host side of raytrace.comp:547-564)
    FrameParams fp{};
    memcpy(fp.camRot, pc->camInfo.cameraRotation, 64);
    memcpy(fp.camPos, pc->camInfo.pos, 12);
    fp.planeHeight = pc->camInfo.nearPlane * rt_tan(rt_radians(pc->camInfo.fov * 0.5f)) * 2.f;
    fp.planeWidth = fp.planeHeight * pc->camInfo.aspectRatio;
    fp.bottomLeft[0] = -fp.planeWidth / 2.f;
    fp.bottomLeft[1] = -fp.planeHeight / 2.f;
    fp.bottomLeft[2] = 0.1f;
    fp.tiled = (c->tileSlots && (width % 8u) == 0u) ? 1u : 0u;
    fp.width = width; fp.height = height; fp.row0 = row0; fp.rowStride = rowStride; fp.nRows = nRows; fp.nPixels = nPixels;
    uint32_t lol = pc->frameCount;
    fp.startingSeed = (uint32_t)(rt_random(&lol) * 23892183.f);
    fp.samples = td.singleRender ? td.sampleLimit : td.raysPerPixel;
    fp.bounceLimit = td.bounceLimit;
    fp.progressive = td.progressive;
    fp.frameCount = pc->frameCount;
    fp.nFrames = nFrames;
    fp.camReuse = (c->cameraReuse && td.debug < 0) ? 1u : 0u;
    fp.debug = td.debug;
    fp.boxCap = td.boxCap;
    fp.triCap = td.triangleCap;
    fp.env = pc->environment;

    c->pixStats = td.debug >= 0;
    DevScene sc = c->sc;
    sc.sphereCount = td.sphereCount;
    sc.objectCount = td.objectCount;
    if (sc.objectCount < c->sc.objectCount) { sc.reachCount = 0; sc.objTreeLevels = 0; }  // a dispatch with fewer objects than were uploaded: no masks, no object hierarchy
    // c->sc holds this dispatch's counts while the launches are built; whatever way the function is left, the uploaded scene comes back
    struct SceneGuard {
        rt_ctx* c; DevScene saved;
        ~SceneGuard() { c->sc = saved; }
    } guard{c, c->sc};
    c->sc = sc;

    const uint32_t blocksPix = (nPixels + RT_BLOCK - 1) / RT_BLOCK;
    // several frames in one dispatch: the paths are {64 tile slots} x {frames} (FrameParams::nFrames), in either pipeline
    const uint32_t nSlots = nFrames > 1u ? (nPixels + 63u) / 64u * 64u * nFrames : nPixels;
    DevCounters* dc = (DevCounters*)c->counterBuf.p;
    uint32_t* counts = c->q.counts;

    // Both pipelines give the same bits; which one is faster depends on how much a wave has to do per pixel. Small tiles
    // and scenes with short rays (few box tests per ray, measured on this context's earlier dispatches) go to the fused one.
    poll_ray_cost(c);
    if (c->boxPerRay < 0.0 && c->probe && !c->inProbe && !c->sc.mapFlags && (uint64_t)nPixels * fp.samples >= 8000000ull && td.debug < 0) {
        c->sc = guard.saved;
        if ((rc = probe_ray_cost(c, pc, width, height, row0, rowStride, nRows))) return rc;
        c->sc = sc;
        c->pixStats = false;
    }
    const bool shortRays = c->boxPerRay >= 0.0 && c->boxPerRay < (double)c->fusedBelowBoxTests;
    // the longer the rays, the earlier the global queue of the multi-kernel pipeline pays — and with the dispatch in overlapping
    // parts earlier than it used to (tools/size_sweep.py, one 1080p frame = 2.07 M paths, fused / multi-kernel in parts: Sponza,
    // 153 executed box tests per ray, 113.8 / 103.4 ms; Sponza + 16 dragons 167.2 / 162.5; the klein bottle x 8, 84 tests,
    // 70.2 / 76.6; half a frame, 1.04 M paths: 61.5 / 66.3, 91.9 / 106.8, 41.5 / 57.4): 4 M paths up to 90 tests per ray, falling
    // to 1.5 M at 150
    double sizeLimit = (double)c->fusedBelowPixels;
    if (c->boxPerRay > 90.0) sizeLimit = std::max(0.375 * sizeLimit, sizeLimit - (c->boxPerRay - 90.0) * (0.625 / 60.0) * sizeLimit);
    // (the paths of all the frames of the dispatch count: four frames of a quarter of a 4K frame are a 4K frame's worth)
    // Short rays keep the fused pipeline at any size — unless the traversal misses the caches: a scene whose hot data (child pairs
    // and triangle positions) exceeds one XCD's 4 MB of L2 is bound by latency even with few tests per ray, and from 10 M paths the
    // multi-kernel pipeline's extra resident waves and overlapping parts win there too (Cornell + bunny, 33 box tests per ray, ten
    // 1080p frames: 36.1 against 38.0 ms per frame; + dragon 36.3 against 40.1; level between four and six frames:
    // tools/frames_sweep.py), while small scenes (bobadog, the 45-object scene)
    // and scenes of very short rays (fewer than 25 executed tests: 232 k loose triangles on a floor) stay fused (tools/heuristics_table.py)
    const bool bigScene = (uint64_t)c->sc.nodeCount * 32u + (uint64_t)c->sc.triCount * 48u > (4ull << 20);
    const bool shortButMissing = shortRays && bigScene && c->boxPerRay >= 25.0 && nSlots >= (10u << 20);
    c->lastPipeline = c->pipeline >= 0 ? c->pipeline : (((double)nSlots < sizeLimit || (shortRays && !shortButMissing)) ? 1 : 0);
    // a scene that binds a metalness, alpha or bump map: the kernels that read them belong to the multi-kernel pipeline
    // (k_shade_maps, k_trace_pw_alpha), whatever "pipeline" asks for
    if (c->sc.mapFlags) c->lastPipeline = 0;
    if (c->lastPipeline == 1) {  // wave-private fused pipeline: one launch for the whole dispatch
        rc = launch_fused(c, fp, fb);
        if (!rc && nFrames > 1u) {
            hipLaunchKernelGGL(k_blend_frames, dim3(blocksPix), dim3(RT_BLOCK), 0, c->stream, c->ps, fp, fb);
            rc = c->hip(hipGetLastError(), "k_blend_frames");
        }
        if (!rc) request_ray_cost(c);
        return rc;
    }
    // The dispatch in `nLanes` independent parts: contiguous slot ranges (whole 256-slot blocks of all the frames of a tile
    // block), each with its own region of the queues, its own counters and its own stream. A pixel's path depends on nothing but
    // its slot, so the parts give the same bits as the whole; what they buy is overlap: one part's k_shade (bound by the path
    // state it streams) and the draining tail of its k_trace_pw launch run under the other part's traversal (bound by latency).
    // Part 0 runs on the ctx stream, the others fork from it after the ray generation and join it before the image is written.
    int nLanes = (fp.samples > 0 && nSlots >= c->lanesMinSlots) ? std::max(1, std::min(c->lanes, (int)RT_MAX_LANES)) : 1;
    if (c->phaseStats) nLanes = 1;  // the diagnostic kernel's statistics are per launch
    // One scene shape loses by it (tools/lanes_table.py): long rays that walk into many placed objects (C5: sixteen instanced dragons,
    // ~190 executed box tests per ray; 116.1 ms per frame in one part against 121.2 in three at 1080p, 472 against 492 at 4K).
    // Every placed object a ray enters costs a set-up round that reloads the ray from its path state in HBM, so that traversal
    // competes with the other parts' k_shade for HBM instead of complementing it. Such scenes keep one part unless "lanes" was set.
    // (from 8 M paths on: a single 1080p frame of the same scene still gains 7 % from its parts, whose launches are short against their tails)
    if (!c->lanesSet && c->cull && c->boxPerRay >= 150.0 && nSlots >= (8u << 20)) nLanes = 1;
    for (int l = 1; l < nLanes; l++) {
        if (!c->sideStream[l - 1]) {
            if (hipStreamCreateWithFlags(&c->sideStream[l - 1], hipStreamNonBlocking) != hipSuccess) { c->sideStream[l - 1] = nullptr; nLanes = l; break; }
            if (hipEventCreateWithFlags(&c->joinEvent[l - 1], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->pollEventSide[l - 1], hipEventDisableTiming) != hipSuccess) { nLanes = l; break; }
        }
    }
    if (nLanes > 1 && !c->forkEvent && hipEventCreateWithFlags(&c->forkEvent, hipEventDisableTiming) != hipSuccess) { c->forkEvent = nullptr; nLanes = 1; }
    c->lastParts = nLanes;
    struct Lane {
        hipStream_t stream; uint32_t* counts; hipEvent_t poll; uint32_t begin, n, ubActive; int cur; bool pollPending, done;
    } lane[RT_MAX_LANES];
    {
        const uint32_t unit = 256u * std::max(1u, nFrames);  // whole blocks of k_shade, whole tile blocks of all their frames
        const uint32_t units = (nSlots + unit - 1) / unit;
        uint32_t at = 0;
        for (int l = 0; l < nLanes; l++) {
            const uint32_t u = units / (uint32_t)nLanes + ((uint32_t)l < units % (uint32_t)nLanes ? 1u : 0u);
            const uint32_t end = std::min(nSlots, at + u * unit);
            lane[l] = Lane{l ? c->sideStream[l - 1] : c->stream, counts + 16 * l, l ? c->pollEventSide[l - 1] : c->pollEvent, at, end - at, end - at, 0, false, end == at};
            at = end;
        }
    }
    for (int l = 0; l < nLanes; l++) {
        const Lane& L = lane[l];
        if (!L.n) continue;
        hipLaunchKernelGGL(k_raygen, dim3((L.n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->sc, c->ps,
                           c->q.active[0] + L.begin, c->q.rays[0] + 3 * (size_t)L.begin, fp, L.begin, L.begin + L.n);
        // counts: [0],[1] active paths of buffer 0/1; [2],[3] main rays of buffer 0/1; [4] the traversal's work counter; [5],[6] NEE rays, [7],[8] cosine probes of buffer 0/1
        if (fp.samples > 0) hipLaunchKernelGGL(k_init_counts, dim3(1), dim3(64), 0, c->stream, L.counts, L.n);
    }
    RT_HIP(c, hipGetLastError());

    if (fp.samples > 0) {
        if (nLanes > 1) {
            RT_HIP(c, hipEventRecord(c->forkEvent, c->stream));
            for (int l = 1; l < nLanes; l++) RT_HIP(c, hipStreamWaitEvent(lane[l].stream, c->forkEvent, 0));
        }
        // The whole dispatch is enqueued without waiting for the device (rt_amd.h: "asynchronous on the ctx stream"): every
        // kernel reads its queue length from device memory and leaves at once when the queue is empty, so the loop may
        // simply run to the most rounds a pixel can need, samples * (bounceLimit + 1). The active-path count is copied
        // back now and then without ever being waited for; a copy that has arrived shrinks the grids of the launches still
        // to be enqueued (active paths never increase, so a stale count is a valid upper bound) and ends the part at zero.
        const uint64_t maxRounds = (uint64_t)fp.samples * ((uint64_t)fp.bounceLimit + 1);
        struct LaneGuard {  // launch_trace builds its launches for the part named here
            rt_ctx* c;
            ~LaneGuard() { c->curStream = nullptr; c->curCounts = nullptr; c->curLane = 0; c->curGridPct = 100; }
        } laneGuard{c};
        // (small parts — one 1080p frame per dispatch is three parts of 0.69 M paths — run better on 40 % grids: 101.7 -> 99.5 ms per
        // frame; the bench's ten frames per dispatch, 6.9 M paths per part, on 50 %: 77.2 against 78.6)
        c->curGridPct = nLanes > 1 ? (c->laneGridPct > 0 ? c->laneGridPct : (lane[0].n < 1200000u ? 40 : 50)) : 100;
        for (uint64_t it = 0; it < maxRounds; it++) {
            bool any = false;
            for (int l = 0; l < nLanes; l++) {
                Lane& L = lane[l];
                if (L.done) continue;
                any = true;
                c->curStream = L.stream; c->curCounts = L.counts; c->curLane = l;
                const int cur = L.cur, nxt = cur ^ 1;
                uint32_t* const active[2] = {c->q.active[0] + L.begin, c->q.active[1] + L.begin};
                uint32_t* const rays[2] = {c->q.rays[0] + 3 * (size_t)L.begin, c->q.rays[1] + 3 * (size_t)L.begin};
                hipLaunchKernelGGL(k_zero_counts, dim3(1), dim3(64), 0, L.stream, L.counts + nxt, L.counts + 2 + nxt, L.counts + 4, L.counts + 5 + nxt, L.counts + 7 + nxt);
                TraceArgs ta{rays[cur], L.counts + 2 + cur, nullptr, nullptr, dc, L.counts + 5 + cur, L.counts + 7 + cur, L.n};
                const uint64_t ubRays = std::min<uint64_t>((uint64_t)L.ubActive * 3, (uint64_t)L.n * 3);
                if ((rc = launch_trace(c, (uint32_t)ubRays, ta))) break;
                ShadeArgs sa{active[cur], L.counts + cur, active[nxt], rays[nxt], L.counts + nxt, L.counts + 2 + nxt, dc, L.counts + 5 + nxt, L.counts + 7 + nxt, L.n};
                if (c->sc.mapFlags & (RT_MAP_METALNESS | RT_MAP_BUMP)) hipLaunchKernelGGL(k_shade_maps, dim3((L.ubActive + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, L.stream, c->sc, c->ps, sa, fp);
                else hipLaunchKernelGGL(k_shade, dim3((L.ubActive + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, L.stream, c->sc, c->ps, sa, fp);
                L.cur = nxt;
                if (L.pollPending && hipEventQuery(L.poll) == hipSuccess) {
                    L.pollPending = false;
                    L.ubActive = std::min(L.ubActive, c->hostCounts[8 + l]);
                    if (L.ubActive == 0) { L.done = true; continue; }
                }
                if (!L.pollPending && (it & 7u) == 7u) {
                    if (hipMemcpyAsync(c->hostCounts + 8 + l, L.counts + L.cur, 4, hipMemcpyDeviceToHost, L.stream) == hipSuccess &&
                        hipEventRecord(L.poll, L.stream) == hipSuccess)
                        L.pollPending = true;
                }
            }
            if (!any || rc) break;
        }
        // (a copy still in flight when the loop ends is harmless: the stream orders it before the next dispatch's own
        // copies, and the slot is only read after the event of the copy that filled it)
        for (int l = 1; l < nLanes; l++) {  // (also when a launch failed: nothing of this dispatch is left running beside the ctx stream)
            RT_HIP(c, hipEventRecord(c->joinEvent[l - 1], lane[l].stream));
            RT_HIP(c, hipStreamWaitEvent(c->stream, c->joinEvent[l - 1], 0));
        }
        if (rc) return rc;
    }
    if (nFrames > 1u) hipLaunchKernelGGL(k_blend_frames, dim3(blocksPix), dim3(RT_BLOCK), 0, c->stream, c->ps, fp, fb);
    else hipLaunchKernelGGL(k_resolve, dim3(blocksPix), dim3(RT_BLOCK), 0, c->stream, c->ps, fp, fb);
    RT_HIP(c, hipGetLastError());
    request_ray_cost(c);
    return 0;
}
}  // namespace

extern "C" {

int rt_render(rt_ctx* c, const PushConstants* pc, uint32_t width, uint32_t height, uint32_t row0, uint32_t rowStride,
              uint32_t nRows, float* d_rgba) {
    return render_impl(c, pc, width, height, row0, rowStride, nRows, 1u, d_rgba);
}

int rt_render_frames(rt_ctx* c, const PushConstants* pc, uint32_t width, uint32_t height, uint32_t row0, uint32_t rowStride,
                     uint32_t nRows, uint32_t nFrames, float* d_rgba) {
    if (!c || !pc) return -1;
    if (nFrames == 0) return 0;
    // The frames of one call are one dispatch: their paths share the launch (fused pipeline) or the queues of every round
    // (multi-kernel pipeline; render_impl picks the pipeline by the paths of all the frames together, so four frames of a
    // quarter of a 4K frame run like a whole 4K frame). Ordinary frames only (no heat maps: those read per-pixel counters at
    // resolve time), within the 30-bit slot ids and RT_FRAMES_MAX_SLOTS paths (3.9 GB of path state); more frames than that
    // go in several dispatches.
    const uint64_t np = (uint64_t)nRows * width;
    uint32_t per = 1;  // frames per dispatch
    if (nFrames > 1u && pc->rayTraceParams.debug < 0 && np > 0) {
        const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>(c->framesMaxSlots, np), (1ull << 30) - 1);
        per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nFrames, cap / ((np + 63) / 64 * 64)));
        if (c->framesPerLaunch > 0) per = std::min(per, (uint32_t)c->framesPerLaunch);
    }
    PushConstants p = *pc;
    for (uint32_t f = 0; f < nFrames; f += per) {
        const uint32_t n = std::min(per, nFrames - f);
        p.frameCount = pc->frameCount + f;
        int rc = render_impl(c, &p, width, height, row0, rowStride, nRows, n, d_rgba);
        if (rc) return rc;
    }
    return 0;
}

int rt_clear_framebuffer(rt_ctx* c) {
    if (!c) return -1;
    c->fbPixels = 0;  // the next rt_render(..., NULL) starts from a zeroed image
    c->fbValid = false;
    return 0;
}

int rt_read_rgba_f32(rt_ctx* c, float* out, size_t nFloats) {
    if (!c || !out) return -1;
    if (!c->fbValid) return c->fail("no ctx-owned framebuffer: rt_render was never called with d_rgba = NULL");
    if (nFloats != (size_t)c->fbPixels * 4) return c->fail("rt_read_rgba_f32: size mismatch");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemcpyAsync(out, c->fbBuf.p, nFloats * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int rt_read_rgba8_srgb(rt_ctx* c, uint8_t* out, size_t nBytes) {
    if (!c || !out) return -1;
    if (!c->fbValid || nBytes != (size_t)c->fbPixels * 4) return c->fail("rt_read_rgba8_srgb: size mismatch");
    std::vector<float> tmp((size_t)c->fbPixels * 4);
    int rc = rt_read_rgba_f32(c, tmp.data(), tmp.size());
    if (rc) return rc;
    // display encoding only; parity is defined on the fp32 buffer
    for (size_t i = 0; i < tmp.size(); i++) {
        float v = tmp[i];
        if (!(v > 0.f)) v = 0.f;
        if (v > 1.f) v = 1.f;
        if ((i & 3) != 3) v = v <= 0.0031308f ? 12.92f * v : 1.055f * rt_pow(v, 1.f / 2.4f) - 0.055f;
        out[i] = (uint8_t)(v * 255.f + 0.5f);
    }
    return 0;
}

int rt_trace_rays(rt_ctx* c, uint32_t n, const float* origins, const float* dirs, RtHit* hitsOut) {
    if (!c || !origins || !dirs || !hitsOut) return -1;
    if (!c->sc.nodes) return c->fail("rt_trace_rays before rt_upload_scene");
    if (n == 0) return 0;
    RT_HIP(c, hipSetDevice(c->device));
    int rc = ensure_state(c, n);
    if (rc) return rc;
    std::vector<float4> ho(n), hd(n);
    for (uint32_t i = 0; i < n; i++) {
        ho[i] = make_float4(origins[(size_t)i * 3], origins[(size_t)i * 3 + 1], origins[(size_t)i * 3 + 2], 0.f);
        hd[i] = make_float4(dirs[(size_t)i * 3], dirs[(size_t)i * 3 + 1], dirs[(size_t)i * 3 + 2], 0.f);
    }
    RT_HIP(c, hipMemcpyAsync(c->ps.rayO(), ho.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->ps.rayD(), hd.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    size_t need = (size_t)n * 8 + (size_t)n * sizeof(RtHit) + 256;
    if ((rc = dev_alloc(c, c->scratchBuf, need))) return rc;
    uint32_t* prb = (uint32_t*)c->scratchBuf.p;
    uint32_t* prt = prb + n;
    RtHit* dh = (RtHit*)((char*)c->scratchBuf.p + (((size_t)n * 8 + 255) & ~(size_t)255));
    c->hostCounts[0] = n; c->hostCounts[1] = 0; c->hostCounts[2] = 0; c->hostCounts[3] = 0; c->hostCounts[4] = 0;
    RT_HIP(c, hipMemcpyAsync(c->q.counts, c->hostCounts, 20, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemsetAsync(c->ps.statBox(), 0, (size_t)n * 4, c->stream));
    RT_HIP(c, hipMemsetAsync(c->ps.statTri(), 0, (size_t)n * 4, c->stream));
    hipLaunchKernelGGL(k_seed_rays, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->sc, c->ps, n);
    TraceArgs ta{nullptr, c->q.counts, prb, prt, (DevCounters*)c->counterBuf.p};
    if ((rc = launch_trace(c, n, ta))) return rc;
    hipLaunchKernelGGL(k_hit_details, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->sc, c->ps, n, prb, prt, dh);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(hitsOut, dh, (size_t)n * sizeof(RtHit), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return harvest_events(c);
}

int rt_get_counters(rt_ctx* c, RtCounters* out) {
    if (!c || !out) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    DevCounters h;
    RT_HIP(c, hipMemcpyAsync(&h, c->counterBuf.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    out->boxTests = h.boxTests; out->triTests = h.triTests; out->raysTraced = h.raysTraced; out->raysHit = h.raysHit;
    out->raysReference = h.raysReference; out->paths = h.paths; out->segments = h.segments;
    out->traceLaunches = c->traceLaunchesTotal;
    out->emitterTests = h.emitterTests;
    out->skippedBoxTests = h.skippedBoxTests;
    if (c->phaseStats) {
        unsigned long long ps[21];
        RT_HIP(c, hipMemcpy(ps, (char*)c->counterBuf.p + sizeof(DevCounters), sizeof(ps), hipMemcpyDeviceToHost));
        static const char* nm[4] = {"refill", "setup", "interior", "leaf"};
        for (int k = 0; k < 4; k++)
            fprintf(stderr, "[phase_stats] %-8s rounds %12llu lanes %14llu avg active %.1f  clocks/round %8.0f  share of wave time %.1f %%\n", nm[k], ps[k], ps[4 + k],
                    ps[k] ? (double)ps[4 + k] / ps[k] : 0.0, ps[k] ? (double)ps[8 + k] / ps[k] : 0.0,
                    100.0 * ps[8 + k] / std::max(1.0, (double)(ps[8] + ps[9] + ps[10] + ps[11])));
        fprintf(stderr, "[phase_stats] clocks per round from the step's first load to its data: refill (queue entry, incl. the atomic) %.0f, setup (ray, seed) %.0f, interior (child pair) %.0f, leaf (triangles) %.0f\n",
                ps[0] ? (double)ps[16] / ps[0] : 0.0, ps[1] ? (double)ps[17] / ps[1] : 0.0, ps[2] ? (double)ps[18] / ps[2] : 0.0, ps[3] ? (double)ps[19] / ps[3] : 0.0);
        fprintf(stderr, "[phase_stats] trips of the set-up step's object-skipping loop: %llu (%.1f per lane and set-up step)\n", ps[20], ps[5] ? (double)ps[20] / ps[5] : 0.0);
        fprintf(stderr, "[phase_stats] lanes sitting out interior rounds: %.1f at a leaf, %.1f in set-up states, %.1f without a ray (of 64, average)\n",
                ps[2] ? (double)ps[12] / ps[2] : 0.0, ps[2] ? (double)ps[13] / ps[2] : 0.0, ps[2] ? (double)ps[14] / ps[2] : 0.0);
        if (c->waveTimesCount && c->waveTimeBuf.p) {  // the last k_trace_pw launch: when did its waves finish?
            std::vector<unsigned long long> t(c->waveTimesCount * 2);
            RT_HIP(c, hipMemcpy(t.data(), c->waveTimeBuf.p, t.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t w = 0; w < c->waveTimesCount; w++) { t0 = std::min(t0, t[2 * w]); t1 = std::max(t1, t[2 * w + 1]); }
            std::vector<double> end(c->waveTimesCount);
            double busy = 0;
            const double span = (double)(t1 - t0);
            for (size_t w = 0; w < c->waveTimesCount; w++) { end[w] = (t[2 * w + 1] - t0) / span; busy += (t[2 * w + 1] - t[2 * w]) / span; }
            std::sort(end.begin(), end.end());
            auto q = [&](double f) { return end[std::min(end.size() - 1, (size_t)(f * end.size()))]; };
            fprintf(stderr, "[phase_stats] last launch: %zu waves, span %.3f ms (100 MHz clock), mean wave lifetime %.1f %% of it; waves finished by 10/25/50/75/90/99 %% : %.2f %.2f %.2f %.2f %.2f %.2f of the span\n",
                    c->waveTimesCount, span / 1e5, 100.0 * busy / c->waveTimesCount, q(0.10), q(0.25), q(0.50), q(0.75), q(0.90), q(0.99));
        }
    }
    return 0;
}

int rt_reset_counters(rt_ctx* c) {
    if (!c) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    poll_ray_cost(c);  // a snapshot of the last dispatch's counters has arrived by now: what it says about the scene's rays is kept
    RT_HIP(c, hipMemsetAsync(c->counterBuf.p, 0, sizeof(DevCounters) + 256, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->snapPending = false;
    c->snapBox = 0; c->snapRays = 0; c->snapSeg = 0; c->snapPaths = 0;
    int rc = harvest_events(c);
    c->traceMs = 0.0; c->traceLaunches = 0; c->traceLaunchesTotal = 0;
    c->traceSpans.clear();
    return rc;
}

int rt_set_profiling(rt_ctx* c, int on) {
    if (!c) return -1;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    int rc = harvest_events(c);
    c->profiling = on != 0;
    if (on) {
        if (!c->profBase) RT_HIP(c, hipEventCreate(&c->profBase));
        RT_HIP(c, hipEventRecord(c->profBase, c->stream));
        c->traceSpans.clear();
    }
    return rc;
}

int rt_get_trace_busy_ms(rt_ctx* c, double* ms) {
    if (!c || !ms) return -1;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    int rc = harvest_events(c);
    std::vector<std::pair<float, float>> v = c->traceSpans;
    std::sort(v.begin(), v.end());
    double busy = 0.0, end = -1e300;
    for (const auto& iv : v) {
        if ((double)iv.first > end) { busy += (double)iv.second - (double)iv.first; end = iv.second; }
        else if ((double)iv.second > end) { busy += (double)iv.second - end; end = iv.second; }
    }
    *ms = busy;
    return rc;
}

int rt_get_trace_time_ms(rt_ctx* c, double* ms, uint64_t* launches) {
    if (!c) return -1;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    int rc = harvest_events(c);
    if (ms) *ms = c->traceMs;
    if (launches) *launches = c->traceLaunches;
    return rc;
}

const char* rt_last_kernel(const rt_ctx* c) { return c ? c->lastKernel : ""; }
int rt_last_parts(const rt_ctx* c) { return c ? c->lastParts : 0; }

int rt_set_tuning(rt_ctx* c, const char* key, int value) {
    if (!c || !key) return -1;
    std::string k(key);
    if (k == "pipeline") { if (value < -1 || value > 1) return c->fail("pipeline: -1 (auto), 0 or 1"); c->pipeline = value; }
    else if (k == "probe") { c->probe = value ? 1 : 0; }
    else if (k == "frames_max_mslots") { if (value < 1 || value > 1000) return c->fail("frames_max_mslots: 1..1000 (millions of paths per multi-frame dispatch)"); c->framesMaxSlots = (uint64_t)value << 20; }
    else if (k == "frames_per_launch") { if (value < 0) return c->fail("frames_per_launch >= 0"); c->framesPerLaunch = value; }
    else if (k == "camera_reuse") { c->cameraReuse = value ? 1 : 0; }
    else if (k == "light_queries") { c->lightQueries = value ? 1 : 0; int rc = rebuild_emitters(c); if (rc) return rc; }
    else if (k == "fused_below_box_tests") { if (value < 0) return c->fail("fused_below_box_tests >= 0"); c->fusedBelowBoxTests = (uint32_t)value; }
    else if (k == "fused_below_pixels") { if (value < 0) return c->fail("fused_below_pixels >= 0"); c->fusedBelowPixels = (uint32_t)value; }
    else if (k == "trace_variant") { if (value < 0 || value > 1) return c->fail("trace_variant: 0 or 1"); c->traceVariant = value; }
    else if (k == "refill") { if (value < 1 || value > 64) return c->fail("refill: 1..64"); c->refill = value; c->refillMk = value; c->refillMkSet = true; }
    else if (k == "hot_pairs") { if (value < 0 || value > 2) return c->fail("hot_pairs: 0, 1 (six work-groups per CU) or 2 (five)"); c->hotPairs = value; }
    else if (k == "mk_refill") { if (value < 1 || value > 64) return c->fail("mk_refill: 1..64"); c->refillMk = value; c->refillMkSet = true; }
    else if (k == "lds_stack") { if (value != 8 && value != 16 && value != 24) return c->fail("lds_stack: 8, 16 or 24"); c->ldsStackCap = value; }
    else if (k == "fast_lanes") { if (value < 0 || value > 65) return c->fail("fast_lanes: 1..65 (0: back to the defaults)"); c->fastLanesSet = value != 0; c->fastLanes = value ? value : 32; }
    else if (k == "chunk") { if (value < 1 || value > 4096) return c->fail("chunk: 1..4096"); c->chunk = value; }
    else if (k == "w_setup") { if (value < 1 || value > 512) return c->fail("w_setup: 1..512"); c->wSetup = value; c->wSetupFused = value; c->wSetupSet = true; }
    else if (k == "w_leaf") { if (value < 1 || value > 512) return c->fail("w_leaf: 1..512"); c->wLeaf = value; c->wLeafFused = value; c->wLeafSet = true; }
    else if (k == "mk_w_setup") { if (value < 1 || value > 512) return c->fail("mk_w_setup: 1..512"); c->wSetup = value; c->wSetupSet = true; }
    else if (k == "mk_w_leaf") { if (value < 1 || value > 512) return c->fail("mk_w_leaf: 1..512"); c->wLeaf = value; }
    else if (k == "tile_slots") { c->tileSlots = value != 0; }
    else if (k == "mask_identity") { c->maskIdentity = value != 0; }
    else if (k == "fast_share") { if (value < 0 || value > 16) return c->fail("fast_share: 0..16"); c->fastShare = value; }
    else if (k == "scatter") { if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return c->fail("scatter: -1 (auto), 0, 1, 2, 4, 8 or 16"); c->scatter = value; }
    else if (k == "pixel_refill") { if (value < 0 || value > (int)RT_WAVE) return c->fail("pixel_refill must be 0 (by ray length) .. 64"); c->pixelRefill = value; }
    else if (k == "batch_pixels") { if (value < 0 || value > (int)RT_WAVE) return c->fail("batch_pixels must be 0 (auto) .. 64"); c->batchPixels = value; }
    else if (k == "batch_fixed") { if (value < 0 || value > 4096) return c->fail("batch_fixed out of range"); c->batchFixed = value; }
    else if (k == "phase_stats") { if (value < 0) return c->fail("phase_stats >= 0"); c->phaseStats = value; }
    else if (k == "object_tree_min") { if (value < 0) return c->fail("object_tree_min >= 0"); c->objTreeMin = value; }
    else if (k == "lanes") { if (value < 0 || value > RT_MAX_LANES) return c->fail("lanes: 1..4 (parts of a multi-kernel dispatch, each on its own stream), 0 = automatic"); c->lanes = value ? value : 3; c->lanesSet = value != 0; }
    else if (k == "lane_grid_pct") { if (value != 0 && (value < 10 || value > 100)) return c->fail("lane_grid_pct: 0 (by size) or 10..100"); c->laneGridPct = value; }
    else if (k == "lanes_min_kslots") { if (value < 0) return c->fail("lanes_min_kslots >= 0"); c->lanesMinSlots = (uint32_t)value << 10; }
    else if (k == "blocks_per_cu") { if (value < 0 || value > 8) return c->fail("blocks_per_cu: 0..8"); c->blocksPerCU = value; }
    else return c->fail("unknown tuning key " + k);
    return 0;
}

int rt_last_pipeline(const rt_ctx* c) { return c ? c->lastPipeline : -1; }
double rt_ray_cost(const rt_ctx* c) { return c ? c->boxPerRay : -1.0; }

// ---------------------------------------------------------------- GPU BVH build (bvh_build.hip.h)
int rt_bvh_build(rt_ctx* c, const TrianglePoint* points, uint32_t pointCount, Triangle* triangles, float* centroids, uint32_t count,
                 uint32_t triIndex0, uint32_t nodeBase, BVHNode* nodesOut, uint32_t nodeCapacity, uint32_t* nodeCountOut, uint32_t statsOut[3]) {
    if (!c || !points || !triangles || !centroids || !nodesOut || !nodeCountOut) return -1;
    if (count == 0) return c->fail("rt_bvh_build: a mesh with 0 triangles");
    if (nodeCapacity < 2u * count - 1u) return c->fail("rt_bvh_build: node capacity below 2 * count - 1");
    RT_HIP(c, hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    const bool dbg = getenv("RT_BVH_DEBUG") != nullptr;
    auto mark = [&](const char* what) { if (dbg) fprintf(stderr, "[rt_bvh_build] %-12s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); };
    std::vector<float> verts((size_t)count * 9);
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t vi[3] = {triangles[i].v0, triangles[i].v1, triangles[i].v2};
        for (int k = 0; k < 3; k++) {
            if (vi[k] >= pointCount) return c->fail("rt_bvh_build: triangle point index out of range");
            memcpy(&verts[(size_t)i * 9 + 3 * k], points[vi[k]].position, 12);
        }
    }
    const size_t nNodesMax = 2 * (size_t)count - 1;
    DevBuf bVerts, bCent, bPerm, bTmp, bHole, bNodes, bList, bCtr, bWide;
    auto release = [&]() { for (DevBuf* b : {&bVerts, &bCent, &bPerm, &bTmp, &bHole, &bNodes, &bList, &bCtr, &bWide}) dev_free(*b); };
    const uint32_t maxChunks = (count + RT_BVH_CHUNK - 1) / RT_BVH_CHUNK;
    int rc;
    if ((rc = upload(c, bVerts, verts.data(), verts.size() * 4)) || (rc = upload(c, bCent, centroids, (size_t)count * 12)) ||
        (rc = dev_alloc(c, bPerm, (size_t)count * 4)) || (rc = dev_alloc(c, bTmp, (size_t)count * 4)) || (rc = dev_alloc(c, bHole, (size_t)count * 4)) ||
        (rc = dev_alloc(c, bNodes, nNodesMax * sizeof(BNode))) || (rc = dev_alloc(c, bList, 2 * (size_t)(count + 1) * 4)) || (rc = dev_alloc(c, bCtr, 64)) ||
        (rc = dev_alloc(c, bWide, RT_BVH_WIDE_NODES * sizeof(WideAcc) + (size_t)RT_BVH_WIDE_NODES * maxChunks * 4))) {
        release();
        return rc;
    }
    mark("uploaded");
    BvhBuildArgs a{(const float*)bVerts.p, (const float*)bCent.p, (uint32_t*)bPerm.p, (uint32_t*)bTmp.p, (uint32_t*)bHole.p, (BNode*)bNodes.p, (uint32_t*)bCtr.p};
    uint32_t* lists[2] = {(uint32_t*)bList.p, (uint32_t*)bList.p + (count + 1)};
    uint32_t* nextCount = (uint32_t*)bCtr.p + 1;
    hipLaunchKernelGGL(k_bvh_root, dim3(1), dim3(RT_BVH_BLOCK), 0, c->stream, a, count);
    uint32_t zero = 0, nCur = 1;
    hipError_t e = hipMemcpyAsync(lists[0], &zero, 4, hipMemcpyHostToDevice, c->stream);  // the root is node 0
    int cur = 0;
    std::vector<uint32_t> levelStart{0u, 1u};  // arrival numbers of the nodes of level l: [levelStart[l], levelStart[l + 1])
    for (uint32_t level = 0; e == hipSuccess && nCur && level <= 64; level++) {
        e = hipMemsetAsync(nextCount, 0, 4, c->stream);
        if (e != hipSuccess) break;
        // threads per node by the size of the level's nodes: the whole mesh is spread over nCur of them
        if (nCur <= RT_BVH_WIDE_NODES && count >= 8192u) {  // big nodes: many work-groups per node (bvh_build.hip.h, wide path)
            const WideArgs w{a, lists[cur], (WideAcc*)bWide.p, (uint32_t*)((char*)bWide.p + RT_BVH_WIDE_NODES * sizeof(WideAcc)), maxChunks, lists[cur ^ 1], nextCount};
            const dim3 gc(maxChunks, nCur), gn(nCur);
            hipLaunchKernelGGL(w_init, gn, dim3(64), 0, c->stream, w);
            hipLaunchKernelGGL(w_minmax, gc, dim3(256), 0, c->stream, w);
            hipLaunchKernelGGL(w_bins, gc, dim3(256), 0, c->stream, w);
            hipLaunchKernelGGL(w_sweep, gn, dim3(64), 0, c->stream, w);
            hipLaunchKernelGGL(w_count, gc, dim3(256), 0, c->stream, w);
            hipLaunchKernelGGL(w_scan, gn, dim3(64), 0, c->stream, w);
            hipLaunchKernelGGL(w_left, gc, dim3(256), 0, c->stream, w);
            hipLaunchKernelGGL(w_right, gc, dim3(256), 0, c->stream, w);
            hipLaunchKernelGGL(w_commit, gc, dim3(256), 0, c->stream, w);
            hipLaunchKernelGGL(w_finish, gn, dim3(64), 0, c->stream, w);
        }
        else if (nCur <= 32) hipLaunchKernelGGL(k_bvh_level<1024>, dim3(nCur), dim3(1024), 0, c->stream, a, lists[cur], lists[cur ^ 1], nextCount);
        else if (count / nCur >= 128) hipLaunchKernelGGL(k_bvh_level<256>, dim3(nCur), dim3(256), 0, c->stream, a, lists[cur], lists[cur ^ 1], nextCount);
        else hipLaunchKernelGGL(k_bvh_level<64>, dim3(nCur), dim3(64), 0, c->stream, a, lists[cur], lists[cur ^ 1], nextCount);
        e = hipMemcpyAsync(&nCur, nextCount, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (nCur) levelStart.push_back(levelStart.back() + nCur);
        cur ^= 1;
    }
    if (e == hipSuccess) e = hipGetLastError();
    mark("levels done");
    uint32_t nNodes = 0;
    if (e == hipSuccess) e = hipMemcpy(&nNodes, bCtr.p, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && (nNodes == 0 || nNodes > nNodesMax || nNodes != levelStart.back())) { release(); return c->fail("rt_bvh_build: node counter out of range"); }

    // ---- the reference's numbering (bvh_build.hip.h): interior counts bottom-up, slots top-down, nodes written in place
    DevBuf bNum, bOut;
    uint32_t hstats[3] = {0u, 0xffffffffu, 0u};
    if (e == hipSuccess && ((rc = dev_alloc(c, bNum, 3 * (size_t)nNodes * 4 + 64)) || (rc = dev_alloc(c, bOut, (size_t)nNodes * sizeof(BVHNode))))) {
        release(); dev_free(bNum); dev_free(bOut);
        return rc;
    }
    std::vector<uint32_t> perm(count);
    if (e == hipSuccess) {
        uint32_t* nb = (uint32_t*)bNum.p;
        uint32_t* dstats = nb + 3 * (size_t)nNodes;
        BvhNumberArgs na{(const BNode*)bNodes.p, nb, nb + nNodes, nb + 2 * (size_t)nNodes, (BVHNode*)bOut.p, dstats, triIndex0, nodeBase};
        e = hipMemcpyAsync(dstats, hstats, 12, hipMemcpyHostToDevice, c->stream);
        const int nLevels = (int)levelStart.size() - 1;
        for (int l = nLevels - 1; l >= 0 && e == hipSuccess; l--) {
            const uint32_t b0 = levelStart[l], b1 = levelStart[l + 1];
            hipLaunchKernelGGL(k_bvh_count, dim3((b1 - b0 + 255) / 256), dim3(256), 0, c->stream, na, b0, b1);
        }
        for (int l = 0; l < nLevels && e == hipSuccess; l++) {
            const uint32_t b0 = levelStart[l], b1 = levelStart[l + 1];
            hipLaunchKernelGGL(k_bvh_number, dim3((b1 - b0 + 255) / 256), dim3(256), 0, c->stream, na, b0, b1);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(nodesOut, bOut.p, (size_t)nNodes * sizeof(BVHNode), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hstats, dstats, 12, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(perm.data(), bPerm.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
    }
    mark("numbered");
    release();
    dev_free(bNum); dev_free(bOut);
    if (e != hipSuccess) return c->fail(std::string("rt_bvh_build: ") + hipGetErrorString(e));
    *nodeCountOut = nNodes;
    if (statsOut) { statsOut[0] = hstats[0]; statsOut[1] = hstats[1]; statsOut[2] = hstats[2]; }

    // ---- triangles and centroids into the order the partition loops leave them in
    std::vector<Triangle> tOld(triangles, triangles + count);
    std::vector<float> cOld(centroids, centroids + (size_t)count * 3);
    for (uint32_t k2 = 0; k2 < count; k2++) {
        if (perm[k2] >= count) return c->fail("rt_bvh_build: bad permutation");
        triangles[k2] = tOld[perm[k2]];
        memcpy(centroids + 3 * (size_t)k2, &cOld[3 * (size_t)perm[k2]], 12);
    }
    mark("permuted");
    c->bvhBuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int rt_bvh_hook(void* ctx, const TrianglePoint* points, uint32_t pointCount, Triangle* triangles, float* centroids, uint32_t count,
                uint32_t triIndex0, uint32_t nodeBase, BVHNode* nodesOut, uint32_t nodeCapacity, uint32_t* nodeCountOut, uint32_t statsOut[3]) {
    return rt_bvh_build((rt_ctx*)ctx, points, pointCount, triangles, centroids, count, triIndex0, nodeBase, nodesOut, nodeCapacity, nodeCountOut, statsOut);
}

double rt_bvh_last_build_ms(const rt_ctx* c) { return c ? c->bvhBuildMs : 0.0; }

int rt_device_selftest(rt_ctx* c, uint32_t* bitsOut) {
    if (!c || !bitsOut) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    const uint32_t n = 4096;
    std::vector<float> a(n), b(n);
    uint32_t st = 12345u;
    for (uint32_t i = 0; i < n; i++) { a[i] = rt_random(&st); b[i] = rt_random(&st) + 1e-3f; }
    const float kat[7] = {1.0001220703125f, 0.9998779296875f, -1.f, 3.f, 1e-30f, 1e-10f, 2.f};
    size_t bytes = (size_t)n * 12 + 64 + 64;
    int rc = dev_alloc(c, c->scratchBuf, bytes);
    if (rc) return rc;
    float* da = (float*)c->scratchBuf.p;
    float* db = da + n;
    uint32_t* dh = (uint32_t*)(db + n);
    uint32_t* dbits = dh + n;
    float* dkat = (float*)(dbits + 8);
    RT_HIP(c, hipMemcpyAsync(da, a.data(), n * 4, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(db, b.data(), n * 4, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(dkat, kat, sizeof(kat), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_selftest, dim3(n / 256), dim3(256), 0, c->stream, da, db, n, dh, dbits, dkat);
    RT_HIP(c, hipGetLastError());
    std::vector<uint32_t> h(n);
    uint32_t bits = 0;
    RT_HIP(c, hipMemcpyAsync(h.data(), dh, n * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipMemcpyAsync(&bits, dbits, 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t mismatches = 0;
    for (uint32_t i = 0; i < n; i++)
        if (h[i] != selftest_one(a[i], b[i])) mismatches++;
    // bit 31 set: device and host disagree on some primitive
    *bitsOut = bits | (mismatches ? 0x80000000u : 0u);
    if (mismatches) return c->fail("device/host deterministic-math mismatch in " + std::to_string(mismatches) + " of 4096 probes");
    return 0;
}

// ---------------------------------------------------------------- multi-GPU: the final gather over RCCL
// One process per GPU, one rt_ctx per process. RCCL is loaded when the first rt_comm_* call needs it (a host that
// renders on one GPU never touches it); the typed function pointers keep the calls checked against <rccl/rccl.h>.
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("RCCL not found: ") + dlerror(); return false; }
        getUniqueId = (decltype(getUniqueId))dlsym(lib, "ncclGetUniqueId");
        commInitRank = (decltype(commInitRank))dlsym(lib, "ncclCommInitRank");
        commDestroy = (decltype(commDestroy))dlsym(lib, "ncclCommDestroy");
        groupStart = (decltype(groupStart))dlsym(lib, "ncclGroupStart");
        groupEnd = (decltype(groupEnd))dlsym(lib, "ncclGroupEnd");
        send = (decltype(send))dlsym(lib, "ncclSend");
        recv = (decltype(recv))dlsym(lib, "ncclRecv");
        errorString = (decltype(errorString))dlsym(lib, "ncclGetErrorString");
        if (!getUniqueId || !commInitRank || !commDestroy || !groupStart || !groupEnd || !send || !recv || !errorString) {
            err = "RCCL library lacks a required symbol";
            dlclose(lib); lib = nullptr;
            return false;
        }
        return true;
    }
} g_rccl;

int rccl_fail(rt_ctx* c, ncclResult_t r, const char* what) {
    c->error = std::string(what) + ": " + (g_rccl.errorString ? g_rccl.errorString(r) : "?");
    return -2000 - (int)r;
}
}  // namespace

static_assert(RT_COMM_ID_BYTES == sizeof(ncclUniqueId), "rt_amd.h: RT_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

int rt_comm_unique_id(void* idOut) {
    if (!idOut) return -1;
    if (!g_rccl.load()) return -2;
    ncclUniqueId id;
    if (g_rccl.getUniqueId(&id) != ncclSuccess) return -3;
    memcpy(idOut, &id, sizeof(id));
    return 0;
}

int rt_comm_init(rt_ctx* c, const void* id, int nRanks, int rank) {
    if (!c || !id) return -1;
    if (nRanks < 1 || rank < 0 || rank >= nRanks) return c->fail("rt_comm_init: rank out of range");
    if (c->comm) return c->fail("rt_comm_init: this context already has a communicator");
    if (!g_rccl.load()) return c->fail(g_rccl.err);
    RT_HIP(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.commInitRank(&c->comm, nRanks, uid, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return rccl_fail(c, r, "ncclCommInitRank"); }
    c->commRanks = nRanks; c->commRank = rank;
    return 0;
}

int rt_comm_destroy(rt_ctx* c) {
    if (!c) return -1;
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        g_rccl.commDestroy(c->comm);
        c->comm = nullptr;
    }
    c->commRanks = 0;
    return 0;
}

int rt_gather_strips(rt_ctx* c, const float* d_strip, uint32_t width, uint32_t height, int root, float* d_frame) {
    if (!c || !d_strip) return -1;
    if (!c->comm) return c->fail("rt_gather_strips before rt_comm_init");
    const int N = c->commRanks, me = c->commRank;
    if (root < 0 || root >= N) return c->fail("rt_gather_strips: root out of range");
    if (me == root && !d_frame) return c->fail("rt_gather_strips: the root needs d_frame");
    RT_HIP(c, hipSetDevice(c->device));
    auto rows_of = [&](int r) { return (uint32_t)r < height ? (height - (uint32_t)r + (uint32_t)N - 1u) / (uint32_t)N : 0u; };
    const size_t rowFloats = (size_t)width * 4;
    float* stage = nullptr;
    if (me == root) {
        int rc = dev_alloc(c, c->gatherBuf, (size_t)height * rowFloats * sizeof(float));
        if (rc) return rc;
        stage = (float*)c->gatherBuf.p;
    }
    // every rank's strip goes to the root over its own link: one send per rank, N receives on the root, one group
    ncclResult_t r = g_rccl.groupStart();
    if (r != ncclSuccess) return rccl_fail(c, r, "ncclGroupStart");
    if (rows_of(me)) r = g_rccl.send(d_strip, (size_t)rows_of(me) * rowFloats, ncclFloat, root, c->comm, c->stream);
    if (r == ncclSuccess && me == root) {
        size_t at = 0;
        for (int k = 0; k < N && r == ncclSuccess; k++) {
            if (rows_of(k)) r = g_rccl.recv(stage + at, (size_t)rows_of(k) * rowFloats, ncclFloat, k, c->comm, c->stream);
            at += (size_t)rows_of(k) * rowFloats;
        }
    }
    ncclResult_t e = g_rccl.groupEnd();
    if (r != ncclSuccess) return rccl_fail(c, r, "ncclSend/ncclRecv");
    if (e != ncclSuccess) return rccl_fail(c, e, "ncclGroupEnd");
    if (me == root) {  // strips (rank-major) -> image rows: row y came from rank y % N, its row y / N
        return rt_deinterleave_strips(c, stage, width, height, N, d_frame);
    }
    return 0;
}

int rt_deinterleave_strips(rt_ctx* c, const float* d_strips, uint32_t width, uint32_t height, int nRanks, float* d_frame) {
    if (!c || !d_strips || !d_frame) return -1;
    if (nRanks < 1 || width == 0 || height == 0) return c->fail("rt_deinterleave_strips: bad geometry");
    RT_HIP(c, hipSetDevice(c->device));
    const size_t n4 = (size_t)height * width;
    hipLaunchKernelGGL(k_deinterleave_rows, dim3((unsigned)((n4 + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, c->stream,
                       (const float4*)d_strips, (float4*)d_frame, width, height, (uint32_t)nRanks);
    RT_HIP(c, hipGetLastError());
    return 0;
}

int rt_deinterleave_strips_host(rt_ctx* c, const float* strips, uint32_t width, uint32_t height, int nRanks, float* frame) {
    if (!c || !strips || !frame) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)width * height * 16;
    int rc = dev_alloc(c, c->scratchBuf, 2 * bytes);
    if (rc) return rc;
    float* ds = (float*)c->scratchBuf.p;
    float* df = (float*)((char*)c->scratchBuf.p + bytes);
    RT_HIP(c, hipMemcpyAsync(ds, strips, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = rt_deinterleave_strips(c, ds, width, height, nRanks, df))) return rc;
    RT_HIP(c, hipMemcpyAsync(frame, df, bytes, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int rt_device_math_probe(rt_ctx* c, uint32_t n, const float* in, float* out) {
    if (!c || !in || !out) return -1;
    if (n == 0) return 0;
    RT_HIP(c, hipSetDevice(c->device));
    const size_t bi = (size_t)n * 32 * 4, bo = (size_t)n * 64 * 4;
    int rc = dev_alloc(c, c->scratchBuf, bi + bo);
    if (rc) return rc;
    float* di = (float*)c->scratchBuf.p;
    float* dout = di + (size_t)n * 32;
    RT_HIP(c, hipMemcpyAsync(di, in, bi, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_math_probe, dim3((n + 63) / 64), dim3(64), 0, c->stream, di, dout, n);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(out, dout, bo, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int rt_measure_copy_bandwidth(rt_ctx* c, size_t bytes, int iters, double* gbps) {
    if (!c || !gbps || iters <= 0) return -1;
    RT_HIP(c, hipSetDevice(c->device));
    bytes &= ~(size_t)255;
    if (bytes < 4096) return c->fail("copy size too small");
    void *src = nullptr, *dst = nullptr;
    RT_HIP(c, hipMalloc(&src, bytes));
    if (hipMalloc(&dst, bytes) != hipSuccess) { (void)hipFree(src); return c->fail("hipMalloc failed"); }
    (void)hipMemsetAsync(src, 1, bytes, c->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    size_t n = bytes / 16;
    hipLaunchKernelGGL(k_copy_f4, dim3(256 * 8), dim3(256), 0, c->stream, (const float4*)src, (float4*)dst, n);
    (void)hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(k_copy_f4, dim3(256 * 8), dim3(256), 0, c->stream, (const float4*)src, (float4*)dst, n);
    (void)hipEventRecord(e1, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    if (e != hipSuccess) return c->hip(e, "copy bandwidth");
    *gbps = (2.0 * (double)bytes * iters) / (ms * 1e-3) / 1e9;  // read + write
    return 0;
}

}  // extern "C"
