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

// BLOCK: 256 threads per node, 1024 while a level has only a few (big) nodes
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
        const unsigned B = RT_BVH_BINS;
        float best = 1e30f, splitPos = 0.f;
        int axis = 0;
        for (int ax = 0; ax < 3; ax++) {
            const float mn = s_mn[ax], mx = s_mx[ax];
            if (mn == mx) continue;
            float leftArea[RT_BVH_BINS - 1], rightArea[RT_BVH_BINS - 1];
            float leftCount[RT_BVH_BINS - 1], rightCount[RT_BVH_BINS - 1];
            BBox leftBox, rightBox;
            int leftSum = 0, rightSum = 0;
            for (unsigned i = 0; i < B - 1; i++) {
                BBox bl, br;
                for (int d = 0; d < 3; d++) {
                    bl.lo[d] = bvh_unkey(s_lo[ax][i][d]); bl.hi[d] = bvh_unkey(s_hi[ax][i][d]);
                    br.lo[d] = bvh_unkey(s_lo[ax][B - 1 - i][d]); br.hi[d] = bvh_unkey(s_hi[ax][B - 1 - i][d]);
                }
                leftSum += (int)s_cnt[ax][i];
                leftCount[i] = (float)leftSum;
                leftBox.grow_box(bl);
                leftArea[i] = leftBox.surface_area();
                rightSum += (int)s_cnt[ax][B - 1 - i];
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
        const float noSplit = (float)n * parent.surface_area();
        s_do = best >= noSplit ? 0 : 1;
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
