// bvh_build.hip.h — the reference's BVH builder on the GPU.
//
// Same tree as rt_scene::build_bvh (scene.cpp), i.e. as the reference's
// build_bvh / subdivide / find_split_plane (src/vk_engine.cpp:1169-1337): binned SAH
// over centroids with 20 bins, axes tried in x,y,z order with strict '<', the
// `rightArea` double store of :1321-1322, no split when the best cost is not below
// count * area(parent), leaves of <= 2 triangles, depth cap 64 — and the same
// triangle order inside every node, which is decided by the reference's in-place
// partition loop (:1246-1255). That loop is sequential; its result has a closed form
// that a work-group evaluates with two scans (see partition below), checked against
// the loop itself in tests/test_bvh_device.py.
//
// The tree is built level by level, one work-group per node of the level; nodes are
// allocated in arrival order and renumbered into the reference's depth-first order on
// the host (rt_device.hip: rt_bvh_build). All arithmetic that decides the topology is
// the same IEEE fp32 expressions as on the host (-ffp-contract=off); min/max
// reductions are exact in any order. The one thing that can differ from the host
// builder is the sign of a zero in a node bound (first-come on the host, -0 < +0
// here); no traversal result depends on it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"

#define RT_BVH_BINS 20
#define RT_BVH_BLOCK 256

struct BNode {
    float lo[3], hi[3];
    uint32_t first, count;  // triangle range (positions in `perm`)
    uint32_t child;         // first of the two children (arrival numbering), 0xffffffff = leaf
    uint32_t depth;
};

struct BvhBuildArgs {
    const float* verts;   // 9 floats per triangle (v0, v1, v2), original order
    const float* cent;    // 3 floats per triangle, original order
    uint32_t* perm;       // position -> original triangle
    uint32_t* tmp;        // scratch, same size
    uint32_t* hole;       // scratch, same size
    BNode* nodes;
    uint32_t* nodeCounter;
};

// ---- the host builder's Box, operation for operation (scene.cpp)
struct BBox {
    float lo[3] = {1e30f, 1e30f, 1e30f};
    float hi[3] = {-1e30f, -1e30f, -1e30f};
    __device__ void grow_box(const BBox& b) {
        for (int i = 0; i < 3; i++) {
            lo[i] = (b.lo[i] < lo[i]) ? b.lo[i] : lo[i];
            hi[i] = (hi[i] < b.hi[i]) ? b.hi[i] : hi[i];
        }
    }
    __device__ float surface_area() const {
        float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

__device__ __forceinline__ uint32_t bvh_key(float f) {  // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bvh_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ float wave_min_f(float v) {
    for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o, 64); v = v < w ? w : v; }
    return v;
}

// exclusive rank of `flag` among the block's threads in thread order, and the block total
template <int BLOCK>
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* s_w, uint32_t& total) {
    const unsigned long long m = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t inWave = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_w[w] = __popcll(m);
    __syncthreads();
    uint32_t base = 0;
    total = 0;
    for (uint32_t i = 0; i < BLOCK / 64; i++) {
        if (i < w) base += s_w[i];
        total += s_w[i];
    }
    return base + inWave;
}

// The 19-candidate sweep of find_split_plane over the three axes, one thread, in the host's order. cnt/lo/hi are
// [3][RT_BVH_BINS] and [3][RT_BVH_BINS][3] (keys). Returns whether the node is split.
__device__ bool bvh_sweep(const float* mnA, const float* mxA, const uint32_t* cnt, const uint32_t* lo, const uint32_t* hi, const BNode& nd,
                          int& axis, float& splitPos) {
    const unsigned B = RT_BVH_BINS;
    float best = 1e30f;
    axis = 0;
    splitPos = 0.f;
    for (int ax = 0; ax < 3; ax++) {
        const float mn = mnA[ax], mx = mxA[ax];
        if (mn == mx) continue;
        float leftArea[RT_BVH_BINS - 1], rightArea[RT_BVH_BINS - 1];
        float leftCount[RT_BVH_BINS - 1], rightCount[RT_BVH_BINS - 1];
        BBox leftBox, rightBox;
        int leftSum = 0, rightSum = 0;
        for (unsigned i = 0; i < B - 1; i++) {
            BBox bl, br;
            for (int d = 0; d < 3; d++) {
                bl.lo[d] = bvh_unkey(lo[(ax * B + i) * 3 + d]); bl.hi[d] = bvh_unkey(hi[(ax * B + i) * 3 + d]);
                br.lo[d] = bvh_unkey(lo[(ax * B + (B - 1 - i)) * 3 + d]); br.hi[d] = bvh_unkey(hi[(ax * B + (B - 1 - i)) * 3 + d]);
            }
            leftSum += (int)cnt[ax * B + i];
            leftCount[i] = (float)leftSum;
            leftBox.grow_box(bl);
            leftArea[i] = leftBox.surface_area();
            rightSum += (int)cnt[ax * B + (B - 1 - i)];
            rightCount[B - 2 - i] = (float)rightSum;
            rightBox.grow_box(br);
            rightArea[i] = rightBox.surface_area();          // the reference's double store (src/vk_engine.cpp:1321-1322)
            rightArea[B - 2 - i] = rightBox.surface_area();
        }
        const float scale = (mx - mn) / (float)B;
        for (unsigned i = 0; i < B - 1; i++) {
            const float cost = leftCount[i] * leftArea[i] + rightCount[i] * rightArea[i];
            if (cost < best) { axis = ax; splitPos = mn + scale * (float)(i + 1); best = cost; }
        }
    }
    BBox parent;
    for (int d = 0; d < 3; d++) { parent.lo[d] = nd.lo[d]; parent.hi[d] = nd.hi[d]; }
    const float noSplit = (float)nd.count * parent.surface_area();
    return !(best >= noSplit);
}

// BLOCK: 256 threads per node (64 for the small nodes of deep levels)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_bvh_level(BvhBuildArgs a, const uint32_t* cur, uint32_t* next, uint32_t* nextCount) {
    __shared__ uint32_t s_cnt[3][RT_BVH_BINS];
    __shared__ uint32_t s_lo[3][RT_BVH_BINS][3], s_hi[3][RT_BVH_BINS][3];
    __shared__ float s_red[12][BLOCK / 64];
    __shared__ float s_mn[3], s_mx[3];
    __shared__ float s_split;
    __shared__ int s_axis, s_do;
    __shared__ uint32_t s_w[BLOCK / 64];
    __shared__ uint32_t s_child;

    const uint32_t id = cur[blockIdx.x];
    const BNode nd = a.nodes[id];
    const uint32_t n = nd.count, first = nd.first;
    if (n <= 2u || nd.depth >= 64u) return;  // leaf
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;

    // ---- centroid range per axis (find_split: mn / mx)
    {
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll 4
        for (uint32_t i = tid; i < n; i += BLOCK) {
            const float* c = a.cent + 3 * (size_t)a.perm[first + i];
            for (int d = 0; d < 3; d++) { mn[d] = mn[d] < c[d] ? mn[d] : c[d]; mx[d] = mx[d] < c[d] ? c[d] : mx[d]; }
        }
        for (int d = 0; d < 3; d++) {
            const float lo = wave_min_f(mn[d]), hi = wave_max_f(mx[d]);
            if (lane == 0) { s_red[d][wv] = lo; s_red[3 + d][wv] = hi; }
        }
        __syncthreads();
        if (tid < 3) {
            float lo = s_red[tid][0], hi = s_red[3 + tid][0];
            for (uint32_t w = 1; w < BLOCK / 64; w++) { lo = s_red[tid][w] < lo ? s_red[tid][w] : lo; hi = hi < s_red[3 + tid][w] ? s_red[3 + tid][w] : hi; }
            s_mn[tid] = lo; s_mx[tid] = hi;
        }
        for (uint32_t i = tid; i < 3 * RT_BVH_BINS; i += BLOCK) {
            (&s_cnt[0][0])[i] = 0u;
            for (int d = 0; d < 3; d++) { (&s_lo[0][0][0])[3 * i + d] = bvh_key(1e30f); (&s_hi[0][0][0])[3 * i + d] = bvh_key(-1e30f); }
        }
        __syncthreads();
    }

    // ---- bins of the three axes in one pass (a triangle's three grow_point calls = its own box)
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += BLOCK) {
        const uint32_t t = a.perm[first + i];
        const float* v = a.verts + 9 * (size_t)t;
        const float* c = a.cent + 3 * (size_t)t;
        float tl[3], th[3];
        for (int d = 0; d < 3; d++) {
            float l = v[d], h = v[d];
            l = v[3 + d] < l ? v[3 + d] : l; h = h < v[3 + d] ? v[3 + d] : h;
            l = v[6 + d] < l ? v[6 + d] : l; h = h < v[6 + d] ? v[6 + d] : h;
            tl[d] = l; th[d] = h;
        }
        for (int ax = 0; ax < 3; ax++) {
            const float mn = s_mn[ax], mx = s_mx[ax];
            if (mn == mx) continue;
            const float scale = (float)RT_BVH_BINS / (mx - mn);
            const float f = floorf((c[ax] - mn) * scale);
            const float lim = (float)(RT_BVH_BINS - 1);
            const int bi = (int)(lim < f ? lim : f);
            atomicAdd(&s_cnt[ax][bi], 1u);
            for (int d = 0; d < 3; d++) { atomicMin(&s_lo[ax][bi][d], bvh_key(tl[d])); atomicMax(&s_hi[ax][bi][d], bvh_key(th[d])); }
        }
    }
    __syncthreads();

    // ---- the sweep, sequentially, as the host does it
    if (tid == 0) {
        int axis = 0;
        float splitPos = 0.f;
        s_do = bvh_sweep(s_mn, s_mx, &s_cnt[0][0], &s_lo[0][0][0], &s_hi[0][0][0], nd, axis, splitPos) ? 1 : 0;
        s_axis = axis;
        s_split = splitPos;
    }
    __syncthreads();
    if (!s_do) return;  // leaf, triangles untouched
    const int axis = s_axis;
    const float splitPos = s_split;
    auto isL = [&](uint32_t i) { return a.cent[3 * (size_t)a.perm[first + i] + axis] < splitPos; };

    // ---- the partition loop of the reference, in closed form.
    //   while (i <= j) { if (c[i] < split) i++; else swap(a[i], a[j--]); }
    // Let nL = #left elements, holes = positions p < nL holding a right element (ascending), hext = holes ++ [nL].
    //   p < nL, left element            : stays.
    //   q >= nL, left element           : goes to holes[c], c = #left elements in (q, n-1].
    //   pos >= nL receives              : hext[0] if pos == n-1; the element at pos+1 if that is a right element;
    //                                     else hext[c], c = #left elements in (pos, n-1].
    uint32_t nL = 0;
    {
        uint32_t cnt = 0;
#pragma unroll 4
        for (uint32_t i = tid; i < n; i += BLOCK) cnt += isL(i) ? 1u : 0u;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        __syncthreads();
        if (lane == 0) s_w[wv] = cnt;
        __syncthreads();
        for (uint32_t w = 0; w < BLOCK / 64; w++) nL += s_w[w];
    }
    uint32_t k = 0;  // holes
    for (uint32_t base = 0; base < nL; base += BLOCK) {
        const uint32_t p = base + tid;
        const bool in = p < nL;
        const bool l = in && isL(p);
        uint32_t tot;
        const uint32_t r = block_rank<BLOCK>(in && !l, s_w, tot);
        if (in) {
            if (l) a.tmp[first + p] = a.perm[first + p];
            else a.hole[first + k + r] = p;
        }
        k += tot;
    }
    __syncthreads();  // hole[] is read below by other threads
    {
        uint32_t cBase = 0;  // left elements seen so far, scanning from the right end
        const uint32_t nR = n - nL;
        for (uint32_t base = 0; base < nR; base += BLOCK) {
            const uint32_t t = base + tid;
            const bool in = t < nR;
            const uint32_t pos = n - 1u - (in ? t : 0u);
            const bool l = in && isL(pos);
            uint32_t tot;
            const uint32_t c = cBase + block_rank<BLOCK>(l, s_w, tot);
            if (in) {
                if (l) a.tmp[first + a.hole[first + c]] = a.perm[first + pos];
                uint32_t src;
                if (pos == n - 1u) src = k ? a.hole[first] : nL;
                else if (!isL(pos + 1u)) src = pos + 1u;
                else src = c < k ? a.hole[first + c] : nL;
                a.tmp[first + pos] = a.perm[first + src];
            }
            cBase += tot;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += BLOCK) a.perm[first + i] = a.tmp[first + i];
    __syncthreads();
    if (nL == 0u || nL == n) return;  // the reference gives up after the partition: leaf with the permuted order

    // ---- children: bounds of both ranges in one pass
    float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f}, cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += BLOCK) {
        const float* v = a.verts + 9 * (size_t)a.perm[first + i];
        for (int d = 0; d < 3; d++) {
            float l = v[d], h = v[d];
            l = v[3 + d] < l ? v[3 + d] : l; h = h < v[3 + d] ? v[3 + d] : h;
            l = v[6 + d] < l ? v[6 + d] : l; h = h < v[6 + d] ? v[6 + d] : h;
            if (i < nL) { bl[d] = l < bl[d] ? l : bl[d]; bh[d] = bh[d] < h ? h : bh[d]; }
            else { cl[d] = l < cl[d] ? l : cl[d]; ch[d] = ch[d] < h ? h : ch[d]; }
        }
    }
    for (int d = 0; d < 3; d++) {
        const float x0 = wave_min_f(bl[d]), x1 = wave_max_f(bh[d]), y0 = wave_min_f(cl[d]), y1 = wave_max_f(ch[d]);
        if (lane == 0) { s_red[d][wv] = x0; s_red[3 + d][wv] = x1; s_red[6 + d][wv] = y0; s_red[9 + d][wv] = y1; }
    }
    if (tid == 0) s_child = atomicAdd(a.nodeCounter, 2u);
    __syncthreads();
    if (tid == 0) {
        const uint32_t child = s_child;
        BNode L, R;
        for (int d = 0; d < 3; d++) {
            float x0 = s_red[d][0], x1 = s_red[3 + d][0], y0 = s_red[6 + d][0], y1 = s_red[9 + d][0];
            for (uint32_t w = 1; w < BLOCK / 64; w++) {
                x0 = s_red[d][w] < x0 ? s_red[d][w] : x0; x1 = x1 < s_red[3 + d][w] ? s_red[3 + d][w] : x1;
                y0 = s_red[6 + d][w] < y0 ? s_red[6 + d][w] : y0; y1 = y1 < s_red[9 + d][w] ? s_red[9 + d][w] : y1;
            }
            L.lo[d] = x0; L.hi[d] = x1; R.lo[d] = y0; R.hi[d] = y1;
        }
        L.first = first; L.count = nL; L.child = 0xffffffffu; L.depth = nd.depth + 1u;
        R.first = first + nL; R.count = n - nL; R.child = 0xffffffffu; R.depth = nd.depth + 1u;
        a.nodes[child] = L;
        a.nodes[child + 1u] = R;
        a.nodes[id].child = child;
        const uint32_t at = atomicAdd(nextCount, 2u);
        next[at] = child;
        next[at + 1u] = child + 1u;
    }
}

// bounds of the root (update_bounds of build_bvh): one block
__global__ __launch_bounds__(RT_BVH_BLOCK) void k_bvh_root(BvhBuildArgs a, uint32_t n) {
    __shared__ float s_red[6][RT_BVH_BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = tid; i < n; i += RT_BVH_BLOCK) {
        a.perm[i] = i;
        const float* v = a.verts + 9 * (size_t)i;
        for (int d = 0; d < 3; d++)
            for (int p = 0; p < 3; p++) { const float x = v[3 * p + d]; bl[d] = x < bl[d] ? x : bl[d]; bh[d] = bh[d] < x ? x : bh[d]; }
    }
    for (int d = 0; d < 3; d++) {
        const float x0 = wave_min_f(bl[d]), x1 = wave_max_f(bh[d]);
        if (lane == 0) { s_red[d][wv] = x0; s_red[3 + d][wv] = x1; }
    }
    __syncthreads();
    if (tid == 0) {
        BNode r;
        for (int d = 0; d < 3; d++) {
            float x0 = s_red[d][0], x1 = s_red[3 + d][0];
            for (uint32_t w = 1; w < RT_BVH_BLOCK / 64; w++) { x0 = s_red[d][w] < x0 ? s_red[d][w] : x0; x1 = x1 < s_red[3 + d][w] ? s_red[3 + d][w] : x1; }
            r.lo[d] = x0; r.hi[d] = x1;
        }
        r.first = 0; r.count = n; r.child = 0xffffffffu; r.depth = 0;
        a.nodes[0] = r;
        *a.nodeCounter = 1u;
    }
}

// ---------------------------------------------------------------- wide path: the first levels, many work-groups per node
// While a level has only a few nodes, each holding a large share of the mesh, one work-group per node leaves the GPU empty
// (the root level of an 871 k-triangle mesh took 17 of the builder's 50 ms). The same steps, split into kernels over chunks
// of RT_BVH_CHUNK positions (grid: chunks x nodes of the level), with the node's accumulators in global memory:
//   w_init -> w_minmax -> w_bins -> w_sweep -> w_count -> w_scan -> w_left -> w_right -> w_commit -> w_finish
// The partition's closed form only needs PL(p) = number of left elements before position p: per-chunk counts, one scan over
// the chunks, and a scan inside each chunk.
#define RT_BVH_CHUNK 2048u
#define RT_BVH_WIDE_NODES 32u   // levels with at most this many nodes take the wide path
struct WideAcc {
    uint32_t mn[3], mx[3];                                  // keys
    uint32_t cnt[3 * RT_BVH_BINS], lo[3 * RT_BVH_BINS * 3], hi[3 * RT_BVH_BINS * 3];
    uint32_t clo[3], chi[3], dlo[3], dhi[3];                // bounds of the left / right child, keys
    float split;
    int axis, doSplit;
    uint32_t nL, k;
};
struct WideArgs {
    BvhBuildArgs b;
    const uint32_t* cur;
    WideAcc* acc;        // one per node of the level
    uint32_t* chunkL;    // [nodes][maxChunks]: left elements per chunk, then their exclusive scan
    uint32_t maxChunks;
    uint32_t* next;
    uint32_t* nextCount;
};

__global__ __launch_bounds__(64) void w_init(WideArgs w) {
    WideAcc& A = w.acc[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < 3; i += 64) { A.mn[i] = bvh_key(1e30f); A.mx[i] = bvh_key(-1e30f); A.clo[i] = A.dlo[i] = bvh_key(1e30f); A.chi[i] = A.dhi[i] = bvh_key(-1e30f); }
    for (uint32_t i = threadIdx.x; i < 3 * RT_BVH_BINS; i += 64) A.cnt[i] = 0u;
    for (uint32_t i = threadIdx.x; i < 3 * RT_BVH_BINS * 3; i += 64) { A.lo[i] = bvh_key(1e30f); A.hi[i] = bvh_key(-1e30f); }
    if (threadIdx.x == 0) { A.split = 0.f; A.axis = 0; A.doSplit = 0; A.nL = 0u; A.k = 0u; }
}

__global__ __launch_bounds__(256) void w_minmax(WideArgs w) {
    const BNode nd = w.b.nodes[w.cur[blockIdx.y]];
    const uint32_t n = nd.count, first = nd.first, base = blockIdx.x * RT_BVH_CHUNK;
    if (n <= 2u || nd.depth >= 64u || base >= n) return;
    const uint32_t end = min(n, base + RT_BVH_CHUNK);
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) {
        const float* c = w.b.cent + 3 * (size_t)w.b.perm[first + i];
        for (int d = 0; d < 3; d++) { mn[d] = mn[d] < c[d] ? mn[d] : c[d]; mx[d] = mx[d] < c[d] ? c[d] : mx[d]; }
    }
    WideAcc& A = w.acc[blockIdx.y];
    for (int d = 0; d < 3; d++) {
        const float lo = wave_min_f(mn[d]), hi = wave_max_f(mx[d]);
        if ((threadIdx.x & 63u) == 0) { atomicMin(&A.mn[d], bvh_key(lo)); atomicMax(&A.mx[d], bvh_key(hi)); }
    }
}

__global__ __launch_bounds__(256) void w_bins(WideArgs w) {
    __shared__ uint32_t s_cnt[3 * RT_BVH_BINS], s_lo[3 * RT_BVH_BINS * 3], s_hi[3 * RT_BVH_BINS * 3];
    const BNode nd = w.b.nodes[w.cur[blockIdx.y]];
    const uint32_t n = nd.count, first = nd.first, base = blockIdx.x * RT_BVH_CHUNK;
    if (n <= 2u || nd.depth >= 64u || base >= n) return;
    const uint32_t end = min(n, base + RT_BVH_CHUNK);
    WideAcc& A = w.acc[blockIdx.y];
    for (uint32_t i = threadIdx.x; i < 3 * RT_BVH_BINS; i += 256) s_cnt[i] = 0u;
    for (uint32_t i = threadIdx.x; i < 3 * RT_BVH_BINS * 3; i += 256) { s_lo[i] = bvh_key(1e30f); s_hi[i] = bvh_key(-1e30f); }
    __syncthreads();
    float mnA[3], mxA[3];
    for (int d = 0; d < 3; d++) { mnA[d] = bvh_unkey(A.mn[d]); mxA[d] = bvh_unkey(A.mx[d]); }
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) {
        const uint32_t t = w.b.perm[first + i];
        const float* v = w.b.verts + 9 * (size_t)t;
        const float* c = w.b.cent + 3 * (size_t)t;
        float tl[3], th[3];
        for (int d = 0; d < 3; d++) {
            float l = v[d], h = v[d];
            l = v[3 + d] < l ? v[3 + d] : l; h = h < v[3 + d] ? v[3 + d] : h;
            l = v[6 + d] < l ? v[6 + d] : l; h = h < v[6 + d] ? v[6 + d] : h;
            tl[d] = l; th[d] = h;
        }
        for (int ax = 0; ax < 3; ax++) {
            const float mn = mnA[ax], mx = mxA[ax];
            if (mn == mx) continue;
            const float scale = (float)RT_BVH_BINS / (mx - mn);
            const float f = floorf((c[ax] - mn) * scale);
            const float lim = (float)(RT_BVH_BINS - 1);
            const int bi = (int)(lim < f ? lim : f);
            atomicAdd(&s_cnt[ax * RT_BVH_BINS + bi], 1u);
            for (int d = 0; d < 3; d++) { atomicMin(&s_lo[(ax * RT_BVH_BINS + bi) * 3 + d], bvh_key(tl[d])); atomicMax(&s_hi[(ax * RT_BVH_BINS + bi) * 3 + d], bvh_key(th[d])); }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3 * RT_BVH_BINS; i += 256) {
        if (s_cnt[i]) {
            atomicAdd(&A.cnt[i], s_cnt[i]);
            for (int d = 0; d < 3; d++) { atomicMin(&A.lo[3 * i + d], s_lo[3 * i + d]); atomicMax(&A.hi[3 * i + d], s_hi[3 * i + d]); }
        }
    }
}

__global__ __launch_bounds__(64) void w_sweep(WideArgs w) {
    if (threadIdx.x) return;
    const BNode nd = w.b.nodes[w.cur[blockIdx.x]];
    if (nd.count <= 2u || nd.depth >= 64u) return;
    WideAcc& A = w.acc[blockIdx.x];
    float mnA[3], mxA[3];
    for (int d = 0; d < 3; d++) { mnA[d] = bvh_unkey(A.mn[d]); mxA[d] = bvh_unkey(A.mx[d]); }
    int axis;
    float split;
    A.doSplit = bvh_sweep(mnA, mxA, A.cnt, A.lo, A.hi, nd, axis, split) ? 1 : 0;
    A.axis = axis;
    A.split = split;
}

// left elements per chunk
__global__ __launch_bounds__(256) void w_count(WideArgs w) {
    __shared__ uint32_t s_w[4];
    const WideAcc& A = w.acc[blockIdx.y];
    const BNode nd = w.b.nodes[w.cur[blockIdx.y]];
    const uint32_t n = nd.count, first = nd.first, base = blockIdx.x * RT_BVH_CHUNK;
    if (!A.doSplit || base >= n) return;
    const uint32_t end = min(n, base + RT_BVH_CHUNK);
    uint32_t cnt = 0;
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) cnt += (w.b.cent[3 * (size_t)w.b.perm[first + i] + A.axis] < A.split) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) w.chunkL[blockIdx.y * w.maxChunks + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of the chunk counts of one node (one work-group, sequential over at most a few hundred chunks per thread 0)
__global__ __launch_bounds__(64) void w_scan(WideArgs w) {
    if (threadIdx.x) return;
    WideAcc& A = w.acc[blockIdx.x];
    const BNode nd = w.b.nodes[w.cur[blockIdx.x]];
    if (!A.doSplit) return;
    const uint32_t chunks = (nd.count + RT_BVH_CHUNK - 1) / RT_BVH_CHUNK;
    uint32_t run = 0;
    uint32_t* c = w.chunkL + blockIdx.x * w.maxChunks;
    for (uint32_t i = 0; i < chunks; i++) { const uint32_t v = c[i]; c[i] = run; run += v; }
    A.nL = run;
}

// PL(p) for the positions of a chunk: lane order inside the work-group follows positions; returns via arrays in registers
//   left part: left elements stay, right elements register as holes (hole rank = p - PL(p))
__global__ __launch_bounds__(256) void w_left(WideArgs w) {
    __shared__ uint32_t s_w[4];
    WideAcc& A = w.acc[blockIdx.y];
    const BNode nd = w.b.nodes[w.cur[blockIdx.y]];
    const uint32_t n = nd.count, first = nd.first, base = blockIdx.x * RT_BVH_CHUNK;
    if (!A.doSplit || base >= n) return;
    const uint32_t nL = A.nL;
    uint32_t run = w.chunkL[blockIdx.y * w.maxChunks + blockIdx.x];  // PL(base)
    for (uint32_t off = 0; off < RT_BVH_CHUNK; off += 256) {
        const uint32_t p = base + off + threadIdx.x;
        const bool in = p < n;
        const bool l = in && (w.b.cent[3 * (size_t)w.b.perm[first + p] + A.axis] < A.split);
        uint32_t tot;
        const uint32_t pl = run + block_rank<256>(l, s_w, tot);  // PL(p)
        if (in && p < nL) {
            if (l) w.b.tmp[first + p] = w.b.perm[first + p];
            else w.b.hole[first + (p - pl)] = p;
        }
        if (in && nL > 0u && p == nL - 1u) A.k = nL - (pl + (l ? 1u : 0u));  // holes = right elements in [0, nL)
        run += tot;
        if (base + off + 256 >= n) break;
    }
}

//   right part (after every hole is known)
__global__ __launch_bounds__(256) void w_right(WideArgs w) {
    __shared__ uint32_t s_w[4];
    const WideAcc& A = w.acc[blockIdx.y];
    const BNode nd = w.b.nodes[w.cur[blockIdx.y]];
    const uint32_t n = nd.count, first = nd.first, base = blockIdx.x * RT_BVH_CHUNK;
    if (!A.doSplit || base >= n) return;
    const uint32_t nL = A.nL, k = A.k;
    uint32_t run = w.chunkL[blockIdx.y * w.maxChunks + blockIdx.x];
    for (uint32_t off = 0; off < RT_BVH_CHUNK; off += 256) {
        const uint32_t pos = base + off + threadIdx.x;
        const bool in = pos < n;
        const bool l = in && (w.b.cent[3 * (size_t)w.b.perm[first + pos] + A.axis] < A.split);
        uint32_t tot;
        const uint32_t pl = run + block_rank<256>(l, s_w, tot);
        if (in && pos >= nL) {
            const uint32_t c = nL - (pl + (l ? 1u : 0u));  // left elements in (pos, n-1]
            if (l) w.b.tmp[first + w.b.hole[first + c]] = w.b.perm[first + pos];
            uint32_t src;
            if (pos == n - 1u) src = k ? w.b.hole[first] : nL;
            else if (!(w.b.cent[3 * (size_t)w.b.perm[first + pos + 1u] + A.axis] < A.split)) src = pos + 1u;
            else src = c < k ? w.b.hole[first + c] : nL;
            w.b.tmp[first + pos] = w.b.perm[first + src];
        }
        run += tot;
        if (base + off + 256 >= n) break;
    }
}

//   the new order becomes the order; bounds of the two children
__global__ __launch_bounds__(256) void w_commit(WideArgs w) {
    WideAcc& A = w.acc[blockIdx.y];
    const BNode nd = w.b.nodes[w.cur[blockIdx.y]];
    const uint32_t n = nd.count, first = nd.first, base = blockIdx.x * RT_BVH_CHUNK;
    if (!A.doSplit || base >= n) return;
    const uint32_t end = min(n, base + RT_BVH_CHUNK), nL = A.nL;
    float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f}, cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) {
        const uint32_t t = w.b.tmp[first + i];
        w.b.perm[first + i] = t;
        const float* v = w.b.verts + 9 * (size_t)t;
        for (int d = 0; d < 3; d++) {
            float l = v[d], h = v[d];
            l = v[3 + d] < l ? v[3 + d] : l; h = h < v[3 + d] ? v[3 + d] : h;
            l = v[6 + d] < l ? v[6 + d] : l; h = h < v[6 + d] ? v[6 + d] : h;
            if (i < nL) { bl[d] = l < bl[d] ? l : bl[d]; bh[d] = bh[d] < h ? h : bh[d]; }
            else { cl[d] = l < cl[d] ? l : cl[d]; ch[d] = ch[d] < h ? h : ch[d]; }
        }
    }
    for (int d = 0; d < 3; d++) {
        const float x0 = wave_min_f(bl[d]), x1 = wave_max_f(bh[d]), y0 = wave_min_f(cl[d]), y1 = wave_max_f(ch[d]);
        if ((threadIdx.x & 63u) == 0) {
            atomicMin(&A.clo[d], bvh_key(x0)); atomicMax(&A.chi[d], bvh_key(x1));
            atomicMin(&A.dlo[d], bvh_key(y0)); atomicMax(&A.dhi[d], bvh_key(y1));
        }
    }
}

__global__ __launch_bounds__(64) void w_finish(WideArgs w) {
    if (threadIdx.x) return;
    const WideAcc& A = w.acc[blockIdx.x];
    const uint32_t id = w.cur[blockIdx.x];
    const BNode nd = w.b.nodes[id];
    if (!A.doSplit || A.nL == 0u || A.nL == nd.count) return;  // leaf (the second case keeps the permuted order, as the reference does)
    const uint32_t child = atomicAdd(w.b.nodeCounter, 2u);
    BNode L, R;
    for (int d = 0; d < 3; d++) {
        L.lo[d] = bvh_unkey(A.clo[d]); L.hi[d] = bvh_unkey(A.chi[d]);
        R.lo[d] = bvh_unkey(A.dlo[d]); R.hi[d] = bvh_unkey(A.dhi[d]);
    }
    L.first = nd.first; L.count = A.nL; L.child = 0xffffffffu; L.depth = nd.depth + 1u;
    R.first = nd.first + A.nL; R.count = nd.count - A.nL; R.child = 0xffffffffu; R.depth = nd.depth + 1u;
    w.b.nodes[child] = L;
    w.b.nodes[child + 1u] = R;
    w.b.nodes[id].child = child;
    const uint32_t at = atomicAdd(w.nextCount, 2u);
    w.next[at] = child;
    w.next[at + 1u] = child + 1u;
}

// ---- the reference's node numbering, on the device. A node's pair of children is allocated when the node is split and the
// left subtree is finished before the right one is touched, so the pair of an interior node v sits at 1 + 2 * (number of
// interior nodes before v in left-first pre-order). Levels occupy contiguous ranges of the arrival numbering.
//   k_bvh_count : bottom-up, interior nodes per subtree
//   k_bvh_number: top-down, pre-order rank and output slot, and the reference-layout node itself
struct BvhNumberArgs {
    const BNode* nodes;
    uint32_t* inner;   // interior nodes in the subtree
    uint32_t* rank;    // interior nodes before this one in pre-order (interior nodes only)
    uint32_t* slot;    // index of the node in the output
    BVHNode* out;
    uint32_t* stats;   // [0] max leaf depth [1] min leaf depth [2] max leaf size
    uint32_t triIndex0, nodeBase;
};

__global__ __launch_bounds__(256) void k_bvh_count(BvhNumberArgs a, uint32_t begin, uint32_t end) {
    const uint32_t v = begin + blockIdx.x * 256u + threadIdx.x;
    if (v >= end) return;
    const uint32_t c = a.nodes[v].child;
    a.inner[v] = c == 0xffffffffu ? 0u : 1u + a.inner[c] + a.inner[c + 1u];
}

__global__ __launch_bounds__(256) void k_bvh_number(BvhNumberArgs a, uint32_t begin, uint32_t end) {
    const uint32_t v = begin + blockIdx.x * 256u + threadIdx.x;
    if (v >= end) return;
    const BNode b = a.nodes[v];
    const uint32_t o = v == 0u ? 0u : a.slot[v];
    BVHNode n;
    n.boundsX[0] = b.lo[0]; n.boundsX[1] = b.hi[0];
    n.boundsY[0] = b.lo[1]; n.boundsY[1] = b.hi[1];
    n.boundsZ[0] = b.lo[2]; n.boundsZ[1] = b.hi[2];
    if (b.child == 0xffffffffu) {
        n.index = a.triIndex0 + b.first;
        n.triCount = b.count;
        atomicMax(&a.stats[0], b.depth);
        atomicMin(&a.stats[1], b.depth);
        atomicMax(&a.stats[2], b.count);
    } else {
        const uint32_t r = v == 0u ? 0u : a.rank[v];
        const uint32_t pair = 1u + 2u * r;
        n.index = a.nodeBase + pair;
        n.triCount = 0u;
        a.slot[b.child] = pair;
        a.slot[b.child + 1u] = pair + 1u;
        a.rank[b.child] = r + 1u;
        a.rank[b.child + 1u] = r + 1u + a.inner[b.child];
    }
    a.out[o] = n;
}
