// bvh_build_gpu.hip -- the acceleration structure of Scene::BuildAccelerationStructures (/root/reference/src/Scene.cpp:67-214)
// built ON the GPU: world-space triangle setup, 63-bit Morton codes, radix sort (rocPRIM), Karras-2012
// radix-tree hierarchy, bottom-up bounds, leaf formation (<= 4 triangles), depth-first node numbering, BVH2 emission, area-greedy
// 4-wide collapse and the per-triangle
// shading-attribute records -- the same outputs as the host builder (bvh_build.cpp), in device memory, without the host ever
// touching a triangle. The reference hands this job to the D3D12 driver (BLAS/TLAS build on the graphics queue); real-time modes
// rebuild the TLAS every frame (src/CommonRenderers.cpp:234-246), which is what a GPU build is for.
//
// Radiance does not depend on which builder ran: the hit definition of pt_device.h is BVH-independent (closest = min (t, inst, prim),
// candidates revisited through exclusive lower keys), boxes are padded conservatively with the host builder's rule, and the world
// transform / attribute unpacking use the same expression order with -ffp-contract=off.
#include "bvh_build_gpu.h"
#include "bvh_build.h"

#include <cstring>
using std::memset;   // rocprim/iterator/texture_cache_iterator.hpp calls an unqualified memset
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hrt {
namespace {

constexpr int kB = 256;
constexpr uint32_t kLeafBit = 0x80000000u;   // radix-tree child reference: leaf (sorted primitive index) when set

struct InstRec {            // per instance, host-prepared
    float world[16];
    uint32_t indexOffset, triBase, material, opaque;
};

struct BuildBuffers {
    // inputs
    const HrptVertexQuantized* vertices; const uint32_t* indices; const InstRec* inst; uint32_t instCount; uint32_t triCount;
    const float* boxesIn;       // box mode (prepare_boxes): 6 floats per primitive, min xyz then max xyz; no triangles, no attribute records
    uint32_t cubeMorton;        // Morton grid over the cube of the largest centroid extent instead of the per-axis box (trees over instances: flat layouts)
    // per unsorted triangle
    GpuTri* triU; float4* boxMinU; float4* boxMaxU;
    uint64_t* keyA; uint64_t* keyB; uint32_t* valA; uint32_t* valB;
    uint32_t* sceneBounds;      // 6 ordered-uint floats: centroid min xyz, max xyz
    uint32_t* flags;            // [0] non-finite vertex seen, [1] max depth of the kept BVH2, [2] max depth of the 4-wide tree
    // radix tree (n-1 internal nodes)
    uint32_t* childL; uint32_t* childR; uint32_t* rangeFirst; uint32_t* rangeLast; uint32_t* parentOfInternal; uint32_t* parentOfLeaf;
    uint32_t* visit; float4* nodeMin; float4* nodeMax;
    uint32_t* keep; uint32_t* newIndex;         // kept (range > 4) internal nodes -> dense BVH2 index
    uint32_t* keptParent; uint32_t* depth;      // per dense BVH2 node
    uint32_t* even; uint32_t* index4;           // per dense BVH2 node: root of a 4-wide node (flag) -> BVH4 node index
    uint32_t* depth4;                           // per dense BVH2 node that roots a 4-wide node: its depth in the 4-wide tree
    uint32_t* pre; uint32_t* keepPre; uint32_t* densePre;   // depth-first (pre-order) rank of every hierarchy node; keep flags / dense indices in that order
    // PLOC: node pool of 2n entries (0..n-1 leaves in Morton order, n + c = c-th merged node), cluster lists, per-iteration scratch
    float4* pMin; float4* pMax; uint32_t* pSize; uint32_t* pParent; uint32_t* pL; uint32_t* pR;
    uint32_t* clusterA; uint32_t* clusterB; uint32_t* nn; uint32_t* mergeFlag; uint32_t* validFlag; uint32_t* mergeIdx; uint32_t* validIdx;
    uint32_t* finalPos;
    uint32_t* plocState;        // [0] clusters left, [1] merged nodes created so far, [2] error flag
    // outputs
    GpuNode* nodes; GpuNode4* nodes4; GpuTri* tris; GpuTriAttr* attrs; GpuTriTangent* tangents;
};

__device__ __forceinline__ uint32_t ordered(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ void xform(const float* p, const float* M, float* o)
{   // mul(float4(p,1), M).xyz, left to right, no FMA (bvh_build.cpp transform_point)
    o[0] = ((p[0] * M[0] + p[1] * M[4]) + p[2] * M[8]) + M[12];
    o[1] = ((p[0] * M[1] + p[1] * M[5]) + p[2] * M[9]) + M[13];
    o[2] = ((p[0] * M[2] + p[1] * M[6]) + p[2] * M[10]) + M[14];
}

__device__ __forceinline__ uint32_t find_instance(const InstRec* inst, uint32_t n, uint32_t g)
{   // last instance whose triBase <= g (instances with zero triangles share a base with their successor and are skipped)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (inst[mid].triBase <= g) lo = mid; else hi = mid; }
    return lo;
}

// block reduction of the centroid bounds, one atomic pair per axis per block
__device__ __forceinline__ void reduce_centroid_bounds(const BuildBuffers& b, const float (&cmin)[3], const float (&cmax)[3])
{
    __shared__ float smin[3][kB / 64], smax[3][kB / 64];
    for (int k = 0; k < 3; ++k) {
        float lo = cmin[k], hi = cmax[k];
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); }
        if ((threadIdx.x & 63) == 0) { smin[k][threadIdx.x >> 6] = lo; smax[k][threadIdx.x >> 6] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float lo = smin[threadIdx.x][0], hi = smax[threadIdx.x][0];
        for (int w = 1; w < kB / 64; ++w) { lo = fminf(lo, smin[threadIdx.x][w]); hi = fmaxf(hi, smax[threadIdx.x][w]); }
        if (lo <= hi) { atomicMin(&b.sceneBounds[threadIdx.x], ordered(lo)); atomicMax(&b.sceneBounds[3 + threadIdx.x], ordered(hi)); }
    }
}
// ---- 1. world-space triangles, padded boxes, centroid bounds
__global__ __launch_bounds__(kB) void k_setup(BuildBuffers b)
{
    uint32_t g = blockIdx.x * kB + threadIdx.x;
    float cmin[3] = { 3e38f, 3e38f, 3e38f }, cmax[3] = { -3e38f, -3e38f, -3e38f };
    if (g < b.triCount) {
        uint32_t i = find_instance(b.inst, b.instCount, g);
        const InstRec& in = b.inst[i];
        uint32_t p = g - in.triBase;
        const uint32_t* ix = b.indices + in.indexOffset + 3 * (size_t)p;
        float v[3][3];
        for (int k = 0; k < 3; ++k) xform(b.vertices[ix[k]].m_Pos, in.world, v[k]);
        GpuTri t;
        for (int k = 0; k < 3; ++k) { t.p0[k] = v[0][k]; t.p1[k] = v[1][k]; t.p2[k] = v[2][k]; }
        t.inst = i; t.prim = p; t.flags = in.opaque;
        b.triU[g] = t;
        float mn[3], mx[3], c[3]; bool bad = false;
        float ext = 0.0f;
        for (int k = 0; k < 3; ++k) ext = fmaxf(ext, fmaxf(v[0][k], fmaxf(v[1][k], v[2][k])) - fminf(v[0][k], fminf(v[1][k], v[2][k])));
        const float extPad = 1e-6f * ext;                                 // bvh_build.cpp triangle_extent
        for (int k = 0; k < 3; ++k) {
            float lo = fminf(v[0][k], fminf(v[1][k], v[2][k])), hi = fmaxf(v[0][k], fmaxf(v[1][k], v[2][k]));
            bad = bad || !(v[0][k] == v[0][k]) || !(v[1][k] == v[1][k]) || !(v[2][k] == v[2][k]) || isinf(lo) || isinf(hi);
            float pad = 1e-5f * fmaxf(fabsf(lo), fabsf(hi)) + 1e-6f + extPad;      // the host builder's conservative padding
            mn[k] = lo - pad; mx[k] = hi + pad; c[k] = 0.5f * lo + 0.5f * hi;
            cmin[k] = c[k]; cmax[k] = c[k];
        }
        if (bad) atomicOr(&b.flags[0], 1u);
        b.boxMinU[g] = make_float4(mn[0], mn[1], mn[2], c[0]);
        b.boxMaxU[g] = make_float4(mx[0], mx[1], mx[2], c[1]);
        b.keyA[g] = (uint64_t)__float_as_uint(c[2]);          // parked: centroid z until k_morton replaces it
    }
    reduce_centroid_bounds(b, cmin, cmax);
}
// ---- 1'. box mode: the primitives are boxes already (the instances of the two-level structure, padded by the host: bvh_build.cpp
// rebuild_two_level_instances), taken as they are
__global__ __launch_bounds__(kB) void k_setup_boxes(BuildBuffers b)
{
    uint32_t g = blockIdx.x * kB + threadIdx.x;
    float cmin[3] = { 3e38f, 3e38f, 3e38f }, cmax[3] = { -3e38f, -3e38f, -3e38f };
    if (g < b.triCount) {
        const float* in = b.boxesIn + 6 * (size_t)g;
        float mn[3], mx[3], c[3]; bool bad = false;
        for (int k = 0; k < 3; ++k) {
            mn[k] = in[k]; mx[k] = in[3 + k];
            bad = bad || !(mn[k] <= mx[k]) || isinf(mn[k]) || isinf(mx[k]);
            c[k] = 0.5f * mn[k] + 0.5f * mx[k];
            cmin[k] = c[k]; cmax[k] = c[k];
        }
        if (bad) atomicOr(&b.flags[0], 1u);
        b.boxMinU[g] = make_float4(mn[0], mn[1], mn[2], c[0]);
        b.boxMaxU[g] = make_float4(mx[0], mx[1], mx[2], c[1]);
        b.keyA[g] = (uint64_t)__float_as_uint(c[2]);
    }
    reduce_centroid_bounds(b, cmin, cmax);
}

__device__ __forceinline__ uint64_t spread21(uint32_t x)
{   // 21 bits -> every third bit of 63
    uint64_t v = x & 0x1fffffu;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// ---- 2. 63-bit Morton code of the centroid inside the centroid bounds
__global__ __launch_bounds__(kB) void k_morton(BuildBuffers b)
{
    uint32_t g = blockIdx.x * kB + threadIdx.x;
    if (g >= b.triCount) return;
    float c[3] = { b.boxMinU[g].w, b.boxMaxU[g].w, __uint_as_float((uint32_t)b.keyA[g]) };
    uint32_t q[3];
    // instances usually stand on a ground plane: a per-axis grid would spend every third bit on their small height differences and tear
    // horizontal neighbours apart in the order; one cell size for all axes keeps the order 2-D until the cells are as small as the heights
    float cube = 0.0f;
    for (int k = 0; k < 3; ++k) cube = fmaxf(cube, unordered(b.sceneBounds[3 + k]) - unordered(b.sceneBounds[k]));
    for (int k = 0; k < 3; ++k) {
        float lo = unordered(b.sceneBounds[k]), hi = unordered(b.sceneBounds[3 + k]);
        float ext = b.cubeMorton ? cube : hi - lo;
        float t = ext > 0.0f ? (c[k] - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        uint32_t v = (uint32_t)(t * 2097151.0f);
        q[k] = v > 2097151u ? 2097151u : v;
    }
    b.keyA[g] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    b.valA[g] = g;
}

// ---- 3. Karras 2012: one thread per internal node of the binary radix tree over the sorted keys (ties broken by position)
// `shift` drops the low Morton bits: keys equal after the shift are split by position, i.e. by balanced halving of a run that is
// still in full-Morton (spatial) order -- this bounds the depth by (63 - shift) + log2(longest run).
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j, int shift)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i] >> shift, c = keys[j] >> shift;
    if (a == c) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ c));
}
__global__ __launch_bounds__(kB) void k_hierarchy(BuildBuffers b, int shift)
{
    const int n = (int)b.triCount;
    int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    const uint64_t* keys = b.keyB;
    int d = (delta(keys, n, i, i + 1, shift) - delta(keys, n, i, i - 1, shift)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d, shift);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d, shift) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, n, i, i + (l + t) * d, shift) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j, shift);
    int s = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d, shift) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int first = i < j ? i : j, last = i < j ? j : i;
    uint32_t cl = (first == gamma) ? ((uint32_t)gamma | kLeafBit) : (uint32_t)gamma;
    uint32_t cr = (last == gamma + 1) ? ((uint32_t)(gamma + 1) | kLeafBit) : (uint32_t)(gamma + 1);
    b.childL[i] = cl; b.childR[i] = cr; b.rangeFirst[i] = (uint32_t)first; b.rangeLast[i] = (uint32_t)last;
    if (cl & kLeafBit) b.parentOfLeaf[gamma] = (uint32_t)i; else b.parentOfInternal[gamma] = (uint32_t)i;
    if (cr & kLeafBit) b.parentOfLeaf[gamma + 1] = (uint32_t)i; else b.parentOfInternal[gamma + 1] = (uint32_t)i;
    if (i == 0) b.parentOfInternal[0] = 0xFFFFFFFFu;
}

// ---- 4. bottom-up bounds: the second thread to reach a node merges its children's boxes.
// The hand-off between the two threads crosses CUs and XCDs (private L1s, per-XCD L2s that are not coherent with each other). A
// __threadfence() pair per level would write back the whole XCD L2 every time (6.2 ms for 1.17 M triangles, 94 % of the build).
// Instead every box written here is stored write-through with agent scope (sc1), drained (vmcnt 0) before the arrival counter is
// bumped (agent-scope atomic), and the second arriver reads the sibling's box with agent-scope loads that bypass its L1: no fence.
typedef unsigned long long __attribute__((address_space(1))) gu64;
__device__ __forceinline__ void store_box_agent(float4* p, float4 v)
{
    gu64* q = (gu64*)p;
    __hip_atomic_store(q, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_box_agent(const float4* p)
{
    gu64* q = (gu64*)p;
    unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)c), 0.0f);
}
__global__ __launch_bounds__(kB) void k_fit(BuildBuffers b)
{
    uint32_t leaf = blockIdx.x * kB + threadIdx.x;
    if (leaf >= b.triCount) return;
    uint32_t node = b.parentOfLeaf[leaf];
    while (node != 0xFFFFFFFFu) {
        // arrival: everything this thread stored for the level below has been drained (end of the previous iteration)
        if (__hip_atomic_fetch_add(&b.visit[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;   // first arrival: the sibling subtree is not finished yet
        float4 mn[2], mx[2];
        uint32_t ch[2] = { b.childL[node], b.childR[node] };
        for (int k = 0; k < 2; ++k) {
            if (ch[k] & kLeafBit) { uint32_t g = b.valB[ch[k] & ~kLeafBit]; mn[k] = b.boxMinU[g]; mx[k] = b.boxMaxU[g]; }   // written by k_setup: plain loads
            else { mn[k] = load_box_agent(b.nodeMin + ch[k]); mx[k] = load_box_agent(b.nodeMax + ch[k]); }
        }
        store_box_agent(b.nodeMin + node, make_float4(fminf(mn[0].x, mn[1].x), fminf(mn[0].y, mn[1].y), fminf(mn[0].z, mn[1].z), 0.0f));
        store_box_agent(b.nodeMax + node, make_float4(fmaxf(mx[0].x, mx[1].x), fmaxf(mx[0].y, mx[1].y), fmaxf(mx[0].z, mx[1].z), 0.0f));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the write-through stores have reached memory before the next arrival is signalled
        node = b.parentOfInternal[node];
    }
}

// ---- 3b/4b. PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) instead of the radix tree: clusters in Morton order
// repeatedly merge with their nearest neighbour (smallest surface area of the joint box) inside a window of +-kPlocRadius positions
// when the choice is mutual. Quality close to a top-down SAH build at a small multiple of the LBVH cost.
constexpr int kPlocRadius = 16;
__device__ __forceinline__ float union_area(float4 amn, float4 amx, float4 bmn, float4 bmx)
{
    float dx = fmaxf(amx.x, bmx.x) - fminf(amn.x, bmn.x), dy = fmaxf(amx.y, bmx.y) - fminf(amn.y, bmn.y), dz = fmaxf(amx.z, bmx.z) - fminf(amn.z, bmn.z);
    return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(kB) void k_ploc_init(BuildBuffers b)
{
    uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= b.triCount) return;
    uint32_t g = b.valB[k];
    b.pMin[k] = b.boxMinU[g]; b.pMax[k] = b.boxMaxU[g]; b.pSize[k] = 1u; b.pParent[k] = 0xFFFFFFFFu;
    b.clusterA[k] = k;
}
// The iteration state lives on the device (plocState: [0] clusters left, [1] nodes created, [2] error), so that a batch of iterations runs
// without a host round trip; grids and scan lengths use the cluster count of the batch's start (an upper bound inside the batch).
__global__ __launch_bounds__(kB) void k_ploc_nn(BuildBuffers b, const uint32_t* __restrict__ cluster, int radius)
{
    const uint32_t count = b.plocState[0];
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= count) return;
    uint32_t me = cluster[i];
    float4 mn = b.pMin[me], mx = b.pMax[me];
    int lo = (int)i - radius < 0 ? 0 : (int)i - radius, hi = (int)i + radius >= (int)count ? (int)count - 1 : (int)i + radius;
    float best = 3e38f; uint32_t bestJ = i;
    for (int j = lo; j <= hi; ++j) {
        if (j == (int)i) continue;
        uint32_t o = cluster[j];
        float a = union_area(mn, mx, b.pMin[o], b.pMax[o]);
        if (a < best) { best = a; bestJ = (uint32_t)j; }       // ties keep the smaller index: makes mutual pairs well defined
    }
    b.nn[i] = bestJ;
}
__global__ __launch_bounds__(kB) void k_ploc_flags(BuildBuffers b, uint32_t bound)
{
    const uint32_t count = b.plocState[0];
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= count) { if (i < bound) { b.mergeFlag[i] = 0u; b.validFlag[i] = 0u; } return; }      // the scans run over `bound` entries
    uint32_t j = b.nn[i];
    bool mutual = j != i && b.nn[j] == i;
    b.mergeFlag[i] = (mutual && i < j) ? 1u : 0u;
    b.validFlag[i] = (mutual && i > j) ? 0u : 1u;
}
__global__ __launch_bounds__(kB) void k_ploc_apply(BuildBuffers b, const uint32_t* __restrict__ cluster, uint32_t* __restrict__ clusterOut)
{
    const uint32_t count = b.plocState[0], created = b.plocState[1];
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= count || !b.validFlag[i]) return;
    uint32_t pos = b.validIdx[i], me = cluster[i];
    if (b.mergeFlag[i]) {
        uint32_t other = cluster[b.nn[i]];
        uint32_t id = b.triCount + created + b.mergeIdx[i];            // deterministic: creation order = position order within the iteration
        float4 amn = b.pMin[me], amx = b.pMax[me], bmn = b.pMin[other], bmx = b.pMax[other];
        b.pMin[id] = make_float4(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z), 0.0f);
        b.pMax[id] = make_float4(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z), 0.0f);
        b.pL[id] = me; b.pR[id] = other; b.pSize[id] = b.pSize[me] + b.pSize[other]; b.pParent[id] = 0xFFFFFFFFu;
        b.pParent[me] = id; b.pParent[other] = id;
        clusterOut[pos] = id;
    } else clusterOut[pos] = me;
}
__global__ void k_ploc_advance(BuildBuffers b)
{
    const uint32_t count = b.plocState[0];
    if (count <= 1u) return;                                           // finished: the remaining iterations of the batch are no-ops
    const uint32_t merges = b.mergeIdx[count - 1u] + b.mergeFlag[count - 1u];
    if (merges == 0u || merges >= count) { b.plocState[2] = 1u; b.plocState[0] = 1u; return; }   // cannot happen with mutual nearest neighbours; never loop forever
    b.plocState[0] = count - merges; b.plocState[1] += merges;
}
// position of the first primitive of every pool node in depth-first order (left subtree first): sum of the left siblings' sizes on the way up
__global__ __launch_bounds__(kB) void k_ploc_offsets(BuildBuffers b)
{
    uint32_t v = blockIdx.x * kB + threadIdx.x;
    uint32_t total = 2u * b.triCount - 1u;
    if (v >= total) return;
    uint32_t off = 0, node = v;
    for (uint32_t p = b.pParent[node]; p != 0xFFFFFFFFu; p = b.pParent[p]) { if (b.pR[p] == node) off += b.pSize[b.pL[p]]; node = p; }
    b.finalPos[v] = off;
}
// hand the PLOC tree to the common back end: internal node c becomes radix-tree style node (n-2) - c (the root, created last, is 0)
__global__ __launch_bounds__(kB) void k_ploc_finish(BuildBuffers b, uint32_t* __restrict__ orderOut)
{
    uint32_t v = blockIdx.x * kB + threadIdx.x;
    const uint32_t n = b.triCount;
    if (v >= 2u * n - 1u) return;
    if (v < n) {                                                       // triangles in depth-first leaf order; the leaf's parent for refits (k_fit)
        orderOut[b.finalPos[v]] = b.valB[v];
        b.parentOfLeaf[b.finalPos[v]] = (n - 2u) - (b.pParent[v] - n);
        return;
    }
    uint32_t c = v - n, id = (n - 2u) - c;
    auto ref = [&](uint32_t child) { return child < n ? (b.finalPos[child] | kLeafBit) : (n - 2u) - (child - n); };
    b.childL[id] = ref(b.pL[v]); b.childR[id] = ref(b.pR[v]);
    b.nodeMin[id] = b.pMin[v]; b.nodeMax[id] = b.pMax[v];
    b.rangeFirst[id] = b.finalPos[v]; b.rangeLast[id] = b.finalPos[v] + b.pSize[v] - 1u;
    const uint32_t pp = b.pParent[v];
    b.parentOfInternal[id] = pp == 0xFFFFFFFFu ? 0xFFFFFFFFu : (n - 2u) - (pp - n);
}

// ---- 5. leaves of up to 4 triangles: an internal node stays inner iff its range holds more than kMaxLeafTris primitives
__global__ __launch_bounds__(kB) void k_classify(BuildBuffers b, uint32_t maxLeaf)
{
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i + 1 >= b.triCount) return;
    b.keep[i] = (b.rangeLast[i] - b.rangeFirst[i] + 1u > maxLeaf) ? 1u : 0u;
}

// ---- 5b. depth-first numbering. The dense node index decides where a node lives in memory; the traversal kernels walk the tree depth
// first, so nodes are numbered in pre-order: rank(v) = (leaves left of v's subtree) - (right turns on the path from the root) + depth(v),
// since a left sibling subtree with k leaves holds k - 1 inner nodes. The kept nodes (classify) are then compacted IN THAT ORDER.
__global__ __launch_bounds__(kB) void k_preorder(BuildBuffers b)
{
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i + 1 >= b.triCount) return;
    uint32_t rightTurns = 0, depth = 0, node = i;
    for (uint32_t p = b.parentOfInternal[node]; p != 0xFFFFFFFFu; p = b.parentOfInternal[p]) { if (b.childR[p] == node) ++rightTurns; ++depth; node = p; }
    const uint32_t r = b.rangeFirst[i] - rightTurns + depth;
    b.pre[i] = r; b.keepPre[r] = b.keep[i];
}
__global__ __launch_bounds__(kB) void k_preorder_gather(BuildBuffers b)
{
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i + 1 >= b.triCount) return;
    b.newIndex[i] = b.densePre[b.pre[i]];
}

__device__ __forceinline__ int32_t encode_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 2) | (count - 1u)); }

// child reference + box in the emitted BVH2
__device__ __forceinline__ void child_ref(const BuildBuffers& b, uint32_t ch, int32_t& ref, float4& mn, float4& mx)
{
    if (ch & kLeafBit) {
        uint32_t k = ch & ~kLeafBit, g = b.valB[k];
        ref = encode_leaf(k, 1u); mn = b.boxMinU[g]; mx = b.boxMaxU[g];
    } else {
        mn = b.nodeMin[ch]; mx = b.nodeMax[ch];
        ref = b.keep[ch] ? (int32_t)b.newIndex[ch] : encode_leaf(b.rangeFirst[ch], b.rangeLast[ch] - b.rangeFirst[ch] + 1u);
    }
}

// ---- 6. dense BVH2 in the traversal layout (child boxes live in the parent)
__global__ __launch_bounds__(kB) void k_emit2(BuildBuffers b)
{
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i + 1 >= b.triCount || !b.keep[i]) return;
    int32_t lr, rr; float4 lmn, lmx, rmn, rmx;
    child_ref(b, b.childL[i], lr, lmn, lmx);
    child_ref(b, b.childR[i], rr, rmn, rmx);
    uint32_t me = b.newIndex[i];
    GpuNode n;
    n.lmin[0] = lmn.x; n.lmin[1] = lmn.y; n.lmin[2] = lmn.z; n.left = lr;
    n.lmax[0] = lmx.x; n.lmax[1] = lmx.y; n.lmax[2] = lmx.z; n.right = rr;
    n.rmin[0] = rmn.x; n.rmin[1] = rmn.y; n.rmin[2] = rmn.z; n.pad0 = 0;
    n.rmax[0] = rmx.x; n.rmax[1] = rmx.y; n.rmax[2] = rmx.z; n.pad1 = 0;
    b.nodes[me] = n;
    if (lr >= 0) b.keptParent[lr] = me;
    if (rr >= 0) b.keptParent[rr] = me;
    if (i == 0) b.keptParent[0] = 0xFFFFFFFFu;
}

// ---- 7. depth of every dense node (walk to the root), max depth
__global__ __launch_bounds__(kB) void k_depth(BuildBuffers b, uint32_t nodeCount)
{
    uint32_t k = blockIdx.x * kB + threadIdx.x;
    uint32_t d = 0;
    if (k < nodeCount) {
        for (uint32_t p = b.keptParent[k]; p != 0xFFFFFFFFu; p = b.keptParent[p]) ++d;
        b.depth[k] = d;
    }
    uint32_t m = d;
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&b.flags[1], m);
}

// ---- 7b. SAH cost of the dense BVH2 (traversal cost 1, intersection cost 1 per triangle), relative to the root's surface area
__global__ __launch_bounds__(kB) void k_sah(BuildBuffers b, uint32_t nodeCount, float* __restrict__ costOut)
{
    uint32_t k = blockIdx.x * kB + threadIdx.x;
    float c = 0.0f;
    if (k < nodeCount) {
        GpuNode n = b.nodes[k];
        auto area = [](const float* mn, const float* mx) { float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return dx * dy + dy * dz + dz * dx; };
        float al = area(n.lmin, n.lmax), ar = area(n.rmin, n.rmax);
        c = al * (n.left >= 0 ? 1.0f : (float)(((uint32_t)~n.left & 3u) + 1u)) + ar * (n.right >= 0 ? 1.0f : (float)(((uint32_t)~n.right & 3u) + 1u));
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c != 0.0f) atomicAdd(costOut, c);
}

// ---- 8. 4-wide collapse, area-greedy like the host builder (bvh_build.cpp collapse4): the children of a 4-wide node start as the two
// children of its 2-wide root; the inner child with the largest box is replaced by its own two children until four slots are used.
// Which 2-wide nodes root a 4-wide node depends on the choices above them, so the flags are set top-down, one launch per 2-wide depth
// (a root at depth d only marks nodes at depths d+1..d+3); 4-wide indices are the prefix sum of the flags in dense = pre-order.
struct Wide4 { int32_t ref[4]; float mn[4][3], mx[4][3]; int cnt; };
__device__ __forceinline__ void greedy_children(const GpuNode* __restrict__ nodes, uint32_t k, bool greedy, Wide4& w)
{
    auto set = [&](int s, int32_t r, const float* a, const float* c) { w.ref[s] = r; for (int t = 0; t < 3; ++t) { w.mn[s][t] = a[t]; w.mx[s][t] = c[t]; } };
    GpuNode n = nodes[k];
    set(0, n.left, n.lmin, n.lmax); set(1, n.right, n.rmin, n.rmax); w.cnt = 2;
    if (!greedy) {          // two fixed 2-wide levels per 4-wide level: the 4-wide depth is exactly half the 2-wide depth
        const int32_t l = n.left, r = n.right;
        if (r >= 0) { GpuNode m = nodes[r]; set(1, m.left, m.lmin, m.lmax); set(2, m.right, m.rmin, m.rmax); w.cnt = 3; }
        if (l >= 0) { GpuNode m = nodes[l]; set(0, m.left, m.lmin, m.lmax); if (w.cnt == 3) set(3, m.right, m.rmin, m.rmax); else set(2, m.right, m.rmin, m.rmax); ++w.cnt; }
        return;
    }
    while (w.cnt < 4) {
        int best = -1; float bestArea = -1.0f;
        for (int i = 0; i < w.cnt; ++i)
            if (w.ref[i] >= 0) {
                float dx = w.mx[i][0] - w.mn[i][0], dy = w.mx[i][1] - w.mn[i][1], dz = w.mx[i][2] - w.mn[i][2];
                float a = dx * dy + dy * dz + dz * dx;
                if (a > bestArea) { bestArea = a; best = i; }
            }
        if (best < 0) break;
        GpuNode m = nodes[w.ref[best]];
        // static slot indices only (runtime-indexed private arrays go to scratch): best is 0..2, the new slot is cnt
#pragma unroll
        for (int i = 0; i < 3; ++i) if (i == best) set(i, m.left, m.lmin, m.lmax);
#pragma unroll
        for (int i = 2; i < 4; ++i) if (i == w.cnt) set(i, m.right, m.rmin, m.rmax);
        ++w.cnt;
    }
}
__global__ void k_mark4_init(BuildBuffers b) { b.even[0] = 1u; b.depth4[0] = 0u; }
__global__ __launch_bounds__(kB) void k_mark4(BuildBuffers b, uint32_t nodeCount, uint32_t depth, bool greedy)
{
    uint32_t k = blockIdx.x * kB + threadIdx.x;
    uint32_t m = 0;
    if (k < nodeCount && b.depth[k] == depth && b.even[k]) {
        Wide4 w; greedy_children(b.nodes, k, greedy, w);
        const uint32_t d4 = b.depth4[k] + 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < w.cnt && w.ref[i] >= 0) { b.even[w.ref[i]] = 1u; b.depth4[w.ref[i]] = d4; m = d4; }
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&b.flags[2], m);
}
__global__ __launch_bounds__(kB) void k_emit4(BuildBuffers b, uint32_t nodeCount, bool greedy)
{
    uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= nodeCount || !b.even[k]) return;
    Wide4 w; greedy_children(b.nodes, k, greedy, w);
    GpuNode4 o;
    float* px[6] = { &o.minx.x, &o.miny.x, &o.minz.x, &o.maxx.x, &o.maxy.x, &o.maxz.x };
    int32_t* pc = &o.child.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bool used = s < w.cnt;
        for (int a = 0; a < 3; ++a) { px[a][s] = used ? w.mn[s][a] : 1e30f; px[3 + a][s] = used ? w.mx[s][a] : 1e30f; }   // far degenerate box = never hit
        pc[s] = used ? (w.ref[s] >= 0 ? (int32_t)b.index4[w.ref[s]] : w.ref[s]) : 0x7fffffff;
    }
    o.pad = make_uint4(0, 0, 0, 0);
    b.nodes4[b.index4[k]] = o;
}

// ---- 9. triangles and shading attributes in leaf (sorted) order
__device__ __forceinline__ float half_bits(uint32_t h)
{   // f16tof32 (include/hobbyrt/detmath.h), exact
    uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) { if (m == 0) return __uint_as_float(s); float f = (float)m * __uint_as_float(0x33800000u); return s ? -f : f; }
    if (e == 31) return __uint_as_float(s | 0x7f800000u | (m << 13));
    return __uint_as_float(s | ((e + 112u) << 23) | (m << 13));
}
__device__ __forceinline__ void unpack_normal(const HrptVertexQuantized& q, float* n)
{   // UnpackVertex, MeshCommon.hlsli:9-22
    n[0] = (float)(q.m_Normal & 1023u) / 511.0f - 1.0f;
    n[1] = (float)((q.m_Normal >> 10) & 1023u) / 511.0f - 1.0f;
    n[2] = (float)((q.m_Normal >> 20) & 1023u) / 511.0f - 1.0f;
}
__device__ __forceinline__ float4 unpack_tangent(const HrptVertexQuantized& q)
{   // DecodeOct, Common.hlsli:174-181
    float ex = (float)(q.m_Tangent & 255u) / 127.0f - 1.0f, ey = (float)((q.m_Tangent >> 8) & 255u) / 127.0f - 1.0f;
    float v[3] = { ex, ey, (1.0f - fabsf(ex)) - fabsf(ey) };
    float neg = -v[2];
    float tt = (neg >= 0.0f) ? neg : 0.0f;
    v[0] += (v[0] >= 0.0f) ? -tt : tt;
    v[1] += (v[1] >= 0.0f) ? -tt : tt;
    float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    return make_float4(v[0] * inv, v[1] * inv, v[2] * inv, (q.m_Normal & (1u << 30)) != 0 ? -1.0f : 1.0f);
}
__global__ __launch_bounds__(kB) void k_attrs(BuildBuffers b)
{
    uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= b.triCount) return;
    uint32_t g = b.valB[k];
    GpuTri t = b.triU[g];
    b.tris[k] = t;
    uint32_t i = t.inst, p = t.prim;
    const InstRec& in = b.inst[i];
    const uint32_t* ix = b.indices + in.indexOffset + 3 * (size_t)p;
    HrptVertexQuantized q0 = b.vertices[ix[0]], q1 = b.vertices[ix[1]], q2 = b.vertices[ix[2]];
    float n0[3], n1[3], n2[3];
    unpack_normal(q0, n0); unpack_normal(q1, n1); unpack_normal(q2, n2);
    GpuTriAttr a;
    a.a = make_float4(n0[0], n0[1], n0[2], n1[0]);
    a.b = make_float4(n1[1], n1[2], n2[0], n2[1]);
    a.c = make_float4(n2[2], half_bits(q0.m_Uv & 0xFFFFu), half_bits(q0.m_Uv >> 16), half_bits(q1.m_Uv & 0xFFFFu));
    a.d = make_float4(half_bits(q1.m_Uv >> 16), half_bits(q2.m_Uv & 0xFFFFu), half_bits(q2.m_Uv >> 16), __uint_as_float(in.material));
    a.e = make_float4(__uint_as_float(i), __uint_as_float(p), 0.0f, 0.0f);
    b.attrs[k] = a;
    if (b.tangents) { GpuTriTangent tg; tg.t0 = unpack_tangent(q0); tg.t1 = unpack_tangent(q1); tg.t2 = unpack_tangent(q2); b.tangents[k] = tg; }
}

__global__ void k_init(uint32_t* sceneBounds, uint32_t* flags)
{
    if (threadIdx.x < 3) { sceneBounds[threadIdx.x] = 0xFFFFFFFFu; sceneBounds[3 + threadIdx.x] = 0u; }
    if (threadIdx.x < 4) flags[threadIdx.x] = 0u;
}

struct Arena {          // one temporary allocation carved into aligned pieces
    char* base = nullptr; size_t off = 0, cap = 0;
    template <class T> T* take(size_t n) { size_t o = off; off += (n * sizeof(T) + 255) & ~(size_t)255; return base ? reinterpret_cast<T*>(base + o) : nullptr; }
};

} // namespace

struct GpuBvhBuilder::Impl {
    BuildBuffers b0{};                 // pointers as carved at prepare(); build() works on a copy (the PLOC path swaps order buffers)
    void* scratch = nullptr; void* prim = nullptr;
    GpuTri* tris = nullptr; GpuTriAttr* attrs = nullptr; GpuTriTangent* tangents = nullptr;
    std::vector<InstRec> inst;         // static part filled at prepare(), world matrices per build()
    size_t sortBytes = 0, scanBytes = 0, bytes = 0;
    uint32_t n = 0;
    bool boxes = false;                // prepare_boxes(): the primitives are boxes (one per leaf), no triangle / attribute outputs
    uint32_t maxLeaf = 0;              // 0: the default leaf size (2, HRPT_GPU_BVH_MAX_LEAF); 1 in box mode
    int lastBudget = -1; bool lastUsePloc = false;   // hierarchy attempt that fitted the stacks at the previous build(): a rebuild starts there
    // what refit() needs of the last successful build: the buffer set as the back end saw it (final leaf order in valB), node count, 2-wide depth
    BuildBuffers last{}; bool haveTree = false; uint32_t lastNodeCount = 0, lastMaxDepth = 0, lastBits = 0; bool lastPloc = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~Impl()
    {
        if (scratch) (void)hipFree(scratch);
        if (tris) (void)hipFree(tris);
        if (attrs) (void)hipFree(attrs);
        if (tangents) (void)hipFree(tangents);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

GpuBvhBuilder::~GpuBvhBuilder() { delete p; }
size_t GpuBvhBuilder::deviceBytes() const { return p ? p->bytes : 0; }

hipError_t GpuBvhBuilder::prepare(const HrptSceneDesc& s, bool needTangents, hipStream_t stream, std::string& error)
{
    delete p; p = new Impl();
    Impl& m = *p;
    // instance table + triangle prefix (host, O(instances)); mesh / material / opacity of an instance never change between rebuilds
    m.inst.resize(s.instanceCount);
    uint64_t T = 0;
    for (uint32_t i = 0; i < s.instanceCount; ++i) {
        const HrptPerInstanceData& in = s.instances[i];
        const HrptMeshData& md = s.meshData[in.m_MeshDataIndex];
        m.inst[i].indexOffset = md.m_IndexOffsets[0]; m.inst[i].triBase = (uint32_t)T; m.inst[i].material = in.m_MaterialIndex;
        m.inst[i].opaque = triangle_flags_for_material(s.materials[in.m_MaterialIndex]);     // bit 0 opaque, bits 1-2 shading class
        T += md.m_IndexCounts[0] / 3;
    }
    if (T < 8 || T >= (1ull << 29)) { error = "triangle count outside the GPU builder's range"; return hipErrorInvalidValue; }
    return allocate((uint32_t)T, &s, needTangents, stream, error);
}

hipError_t GpuBvhBuilder::prepare_boxes(uint32_t count, hipStream_t stream, std::string& error)
{
    delete p; p = new Impl();
    p->boxes = true; p->maxLeaf = 1;
    if (count < 8 || count >= (1u << 25)) { error = "box count outside the GPU builder's range"; return hipErrorInvalidValue; }
    return allocate(count, nullptr, false, stream, error);
}

// every device buffer of a build over n primitives; `s` (triangle mode) also brings the geometry to the device
hipError_t GpuBvhBuilder::allocate(uint32_t n, const HrptSceneDesc* s, bool needTangents, hipStream_t stream, std::string& error)
{
    Impl& m = *p;
    hipError_t e;
    m.n = n;
    if (hipEventCreate(&m.ev0) != hipSuccess || hipEventCreate(&m.ev1) != hipSuccess) { error = "hipEventCreate"; return hipErrorUnknown; }

    // rocPRIM scratch sizes
    (void)rocprim::radix_sort_pairs(nullptr, m.sortBytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 63, stream);
    (void)rocprim::exclusive_scan(nullptr, m.scanBytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n, rocprim::plus<uint32_t>(), stream);

    BuildBuffers& b = m.b0;
    Arena A;
    for (int pass = 0; pass < 2; ++pass) {      // pass 0 sizes the arena, pass 1 hands out pointers
        A.off = 0;
        b.triU = A.take<GpuTri>(m.boxes ? 0 : n); b.boxMinU = A.take<float4>(n); b.boxMaxU = A.take<float4>(n);
        b.keyA = A.take<uint64_t>(n); b.keyB = A.take<uint64_t>(n); b.valA = A.take<uint32_t>(n); b.valB = A.take<uint32_t>(n);
        b.sceneBounds = A.take<uint32_t>(8); b.flags = A.take<uint32_t>(8); b.plocState = A.take<uint32_t>(8);
        b.childL = A.take<uint32_t>(n); b.childR = A.take<uint32_t>(n); b.rangeFirst = A.take<uint32_t>(n); b.rangeLast = A.take<uint32_t>(n);
        b.parentOfInternal = A.take<uint32_t>(n); b.parentOfLeaf = A.take<uint32_t>(n); b.visit = A.take<uint32_t>(n);
        b.nodeMin = A.take<float4>(n); b.nodeMax = A.take<float4>(n);
        b.keep = A.take<uint32_t>(n); b.newIndex = A.take<uint32_t>(n); b.keptParent = A.take<uint32_t>(n); b.depth = A.take<uint32_t>(n);
        b.even = A.take<uint32_t>(n); b.index4 = A.take<uint32_t>(n); b.depth4 = A.take<uint32_t>(n);
        b.pre = A.take<uint32_t>(n); b.keepPre = A.take<uint32_t>(n); b.densePre = A.take<uint32_t>(n);
        b.pMin = A.take<float4>(2 * (size_t)n); b.pMax = A.take<float4>(2 * (size_t)n); b.pSize = A.take<uint32_t>(2 * (size_t)n); b.pParent = A.take<uint32_t>(2 * (size_t)n);
        b.pL = A.take<uint32_t>(2 * (size_t)n); b.pR = A.take<uint32_t>(2 * (size_t)n); b.finalPos = A.take<uint32_t>(2 * (size_t)n);
        b.clusterA = A.take<uint32_t>(n); b.clusterB = A.take<uint32_t>(n); b.nn = A.take<uint32_t>(n); b.mergeFlag = A.take<uint32_t>(n); b.validFlag = A.take<uint32_t>(n);
        b.mergeIdx = A.take<uint32_t>(n); b.validIdx = A.take<uint32_t>(n);
        b.nodes = A.take<GpuNode>(n);            // the traversal kernels read the trees in place: a rebuild may change the node counts
        b.nodes4 = A.take<GpuNode4>(m.maxLeaf == 1 ? n : n / 2 + 1);      // (single-primitive leaves: up to n - 1 inner nodes survive the collapse of a degenerate tree)
        m.prim = A.take<char>(std::max(m.sortBytes, m.scanBytes));
        char* vtx = A.take<char>(s ? (size_t)s->vertexCount * sizeof(HrptVertexQuantized) : 0);
        char* idx = A.take<char>(s ? (size_t)s->indexCount * 4 : 0);
        char* ins = A.take<char>(m.inst.size() * sizeof(InstRec));
        b.boxesIn = A.take<float>(m.boxes ? 6 * (size_t)n : 0);
        b.vertices = reinterpret_cast<const HrptVertexQuantized*>(vtx); b.indices = reinterpret_cast<const uint32_t*>(idx); b.inst = reinterpret_cast<const InstRec*>(ins);
        if (pass == 0) {
            if ((e = hipMalloc(&m.scratch, A.off)) != hipSuccess) { error = "hipMalloc(GPU BVH build arena)"; return e; }
            A.base = static_cast<char*>(m.scratch); A.cap = A.off;
        }
    }
    b.instCount = (uint32_t)m.inst.size(); b.triCount = n; b.cubeMorton = m.boxes ? 1u : 0u;
    if (const char* e = getenv("HRPT_GPU_BVH_CUBE_MORTON")) b.cubeMorton = atoi(e) != 0;
    b.tris = nullptr; b.attrs = nullptr; b.tangents = nullptr;
    m.bytes = A.cap;
    if (m.boxes) return hipSuccess;
    m.bytes += (size_t)n * (sizeof(GpuTri) + sizeof(GpuTriAttr) + (needTangents ? sizeof(GpuTriTangent) : 0));
    if ((e = hipMalloc((void**)&m.tris, (size_t)n * sizeof(GpuTri))) != hipSuccess || (e = hipMalloc((void**)&m.attrs, (size_t)n * sizeof(GpuTriAttr))) != hipSuccess ||
        (needTangents && (e = hipMalloc((void**)&m.tangents, (size_t)n * sizeof(GpuTriTangent))) != hipSuccess)) { error = "hipMalloc(GPU BVH outputs)"; return hipErrorOutOfMemory; }
    b.tris = m.tris; b.attrs = m.attrs; b.tangents = m.tangents;
    if ((e = hipMemcpyAsync(const_cast<HrptVertexQuantized*>(b.vertices), s->vertices, (size_t)s->vertexCount * sizeof(HrptVertexQuantized), hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(const_cast<uint32_t*>(b.indices), s->indices, (size_t)s->indexCount * 4, hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess) { error = "hipMemcpyAsync(GPU BVH inputs)"; return e; }
    return hipSuccess;
}

hipError_t GpuBvhBuilder::build_boxes(const float* boxes, bool usePloc, uint32_t maxStackDepth, hipStream_t stream, GpuBuiltBvh& out, std::string& error)
{
    if (!p || !p->boxes) { error = "GPU BVH builder not prepared for boxes"; return hipErrorInvalidValue; }
    return build_any(nullptr, boxes, usePloc, maxStackDepth, stream, out, error);
}
hipError_t GpuBvhBuilder::build(const HrptPerInstanceData* instances, bool usePloc, uint32_t maxStackDepth, hipStream_t stream, GpuBuiltBvh& out, std::string& error)
{
    if (!p || p->boxes) { error = "GPU BVH builder not prepared"; return hipErrorInvalidValue; }
    return build_any(instances, nullptr, usePloc, maxStackDepth, stream, out, error);
}

hipError_t GpuBvhBuilder::build_any(const HrptPerInstanceData* instances, const float* boxes, bool usePloc, uint32_t maxStackDepth, hipStream_t stream, GpuBuiltBvh& out, std::string& error)
{
    out = GpuBuiltBvh();
    if (!p->scratch) { error = "GPU BVH builder not prepared"; return hipErrorInvalidValue; }
    Impl& m = *p;
    m.haveTree = false;
    hipError_t e;
    BuildBuffers b = m.b0;
    const uint32_t n = m.n;
    void* const prim = m.prim; size_t sortBytes = m.sortBytes, scanBytes = m.scanBytes;   // rocPRIM takes the size by non-const reference
    auto fail = [&](hipError_t err, const char* what) { error = what; (void)hipStreamSynchronize(stream); return err; };
    if (m.boxes) {
        if ((e = hipMemcpyAsync(const_cast<float*>(b.boxesIn), boxes, (size_t)n * 24, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e, "hipMemcpyAsync(GPU BVH boxes)");
    } else {
        for (size_t i = 0; i < m.inst.size(); ++i) std::memcpy(m.inst[i].world, instances[i].m_World, sizeof m.inst[i].world);
        if ((e = hipMemcpyAsync(const_cast<InstRec*>(b.inst), m.inst.data(), m.inst.size() * sizeof(InstRec), hipMemcpyHostToDevice, stream)) != hipSuccess)
            return fail(e, "hipMemcpyAsync(GPU BVH instances)");
    }
    hipEvent_t ev0 = m.ev0, ev1 = m.ev1;
    (void)hipEventRecord(ev0, stream);
    const dim3 gT((n + kB - 1) / kB), blk(kB);
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, stream, b.sceneBounds, b.flags);
    if (m.boxes) hipLaunchKernelGGL(k_setup_boxes, gT, blk, 0, stream, b);
    else hipLaunchKernelGGL(k_setup, gT, blk, 0, stream, b);
    hipLaunchKernelGGL(k_morton, gT, blk, 0, stream, b);
    if ((e = rocprim::radix_sort_pairs(prim, sortBytes, b.keyA, b.keyB, b.valA, b.valB, n, 0, 63, stream)) != hipSuccess) return fail(e, "rocprim::radix_sort_pairs");
    // Hierarchy attempts, best tree first: PLOC (when asked for), then the radix tree over the full 63-bit codes, then radix trees over
    // fewer Morton bits (no re-sort: the order stays the full-code order) until the depth fits the traversal stacks.
    uint32_t nodeCount = 0, maxDepthSeen = 0; int usedBits = 0; bool usedPloc = false;
    uint32_t maxLeafTris = 2;          // measured with the greedy collapse: 2 beats 4 on all three test scenes for both hierarchies (-2..-6 % frame time)
    if (const char* e = getenv("HRPT_GPU_BVH_MAX_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 4) maxLeafTris = (uint32_t)v; }
    if (m.maxLeaf >= 1 && m.maxLeaf <= 4) maxLeafTris = m.maxLeaf;
    int plocRadius = kPlocRadius;
    if (const char* e = getenv("HRPT_GPU_PLOC_RADIUS")) { int v = atoi(e); if (v >= 1 && v <= 256) plocRadius = v; }
    const int attempts[] = { 64, 63, 48, 39, 30, 21, 12, 0 };          // 64 = PLOC
    uint32_t* const mortonOrder = b.valB;                              // leaf k of the radix tree = triangle mortonOrder[k]
    if (m.lastUsePloc != usePloc) m.lastBudget = -1;
    m.lastUsePloc = usePloc;
    int fitted = -1;
    for (int budget : attempts) {
        if (m.lastBudget >= 0 && budget > m.lastBudget) continue;      // too deep last time: moving instances rarely changes that, and every attempt costs a host round trip
        if (budget == 64) {
            if (!usePloc) continue;
            hipLaunchKernelGGL(k_ploc_init, gT, blk, 0, stream, b);
            uint32_t count = n, created = 0; uint32_t* cur = b.clusterA; uint32_t* nxt = b.clusterB; bool ok = true;
            const uint32_t init[4] = { n, 0u, 0u, 0u };
            if ((e = hipMemcpyAsync(b.plocState, init, sizeof init, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e, "GPU BVH build (PLOC state)");
            for (int batch = 0; count > 1 && ok && batch < 64; ++batch) {
                const uint32_t bound = count;                          // no cluster list of this batch is longer
                const dim3 gC((bound + kB - 1) / kB);
                for (int it = 0; it < 6; ++it) {
                    hipLaunchKernelGGL(k_ploc_nn, gC, blk, 0, stream, b, cur, plocRadius);
                    hipLaunchKernelGGL(k_ploc_flags, gC, blk, 0, stream, b, bound);
                    if ((e = rocprim::exclusive_scan(prim, scanBytes, b.mergeFlag, b.mergeIdx, 0u, bound, rocprim::plus<uint32_t>(), stream)) != hipSuccess ||
                        (e = rocprim::exclusive_scan(prim, scanBytes, b.validFlag, b.validIdx, 0u, bound, rocprim::plus<uint32_t>(), stream)) != hipSuccess) return fail(e, "rocprim::exclusive_scan(PLOC)");
                    hipLaunchKernelGGL(k_ploc_apply, gC, blk, 0, stream, b, cur, nxt);
                    hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(1), 0, stream, b);
                    std::swap(cur, nxt);
                }
                uint32_t state[3] = { 0, 0, 0 };
                if ((e = hipMemcpyAsync(state, b.plocState, sizeof state, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
                    (e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "GPU BVH build (PLOC batch)");
                if (state[2] || state[0] == 0 || state[0] > count) { ok = false; break; }
                count = state[0]; created = state[1];
            }
            if (count > 1) ok = false;
            if (!ok || created != n - 1) continue;
            hipLaunchKernelGGL(k_ploc_offsets, dim3((2 * n - 1 + kB - 1) / kB), blk, 0, stream, b);
            hipLaunchKernelGGL(k_ploc_finish, dim3((2 * n - 1 + kB - 1) / kB), blk, 0, stream, b, b.valA);
            b.valB = b.valA;                                           // depth-first leaf order replaces the Morton order downstream
        } else {
            const int shift = 63 - budget;
            b.valB = mortonOrder;
            (void)hipMemsetAsync(b.visit, 0, (size_t)n * 4, stream);
            hipLaunchKernelGGL(k_hierarchy, gT, blk, 0, stream, b, shift);
            hipLaunchKernelGGL(k_fit, gT, blk, 0, stream, b);
        }
        (void)hipMemsetAsync(b.flags + 1, 0, 4, stream);
        hipLaunchKernelGGL(k_classify, gT, blk, 0, stream, b, maxLeafTris);
        hipLaunchKernelGGL(k_preorder, gT, blk, 0, stream, b);
        if ((e = rocprim::exclusive_scan(prim, scanBytes, b.keepPre, b.densePre, 0u, n - 1, rocprim::plus<uint32_t>(), stream)) != hipSuccess) return fail(e, "rocprim::exclusive_scan(keep)");
        hipLaunchKernelGGL(k_preorder_gather, gT, blk, 0, stream, b);
        hipLaunchKernelGGL(k_emit2, gT, blk, 0, stream, b);
        // dense node count = densePre[n-2] + keepPre[n-2]
        uint32_t tail[2] = { 0, 0 };
        if ((e = hipMemcpyAsync(&tail[0], b.densePre + (n - 2), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
            (e = hipMemcpyAsync(&tail[1], b.keepPre + (n - 2), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
            (e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "GPU BVH build (hierarchy)");
        nodeCount = tail[0] + tail[1];
        if (nodeCount == 0) return fail(hipErrorUnknown, "GPU BVH build produced no inner node");
        hipLaunchKernelGGL(k_depth, dim3((nodeCount + kB - 1) / kB), blk, 0, stream, b, nodeCount);
        uint32_t d = 0;
        if ((e = hipMemcpyAsync(&d, b.flags + 1, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess || (e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "GPU BVH build (depth)");
        maxDepthSeen = d; usedBits = budget == 64 ? 63 : budget; usedPloc = budget == 64;
        if (d + 2 <= maxStackDepth) { fitted = budget; break; }
    }
    m.lastBudget = fitted;
    const dim3 gN((nodeCount + kB - 1) / kB);
    // Area-greedy collapse (fuller nodes with tighter boxes: it is what makes a PLOC tree as fast as the host's SAH tree); the fixed
    // two-levels-at-a-time rule stays available for A/B runs (HRPT_GPU_BVH_COLLAPSE=fixed).
    bool greedy = true;
    if (const char* e = getenv("HRPT_GPU_BVH_COLLAPSE")) greedy = strcmp(e, "fixed") != 0;
    (void)hipMemsetAsync(b.even, 0, (size_t)nodeCount * 4, stream);
    hipLaunchKernelGGL(k_mark4_init, dim3(1), dim3(1), 0, stream, b);
    for (uint32_t d = 0; d <= maxDepthSeen; ++d) hipLaunchKernelGGL(k_mark4, gN, blk, 0, stream, b, nodeCount, d, greedy);
    if ((e = rocprim::exclusive_scan(prim, scanBytes, b.even, b.index4, 0u, nodeCount, rocprim::plus<uint32_t>(), stream)) != hipSuccess) return fail(e, "rocprim::exclusive_scan(even)");
    hipLaunchKernelGGL(k_emit4, gN, blk, 0, stream, b, nodeCount, greedy);
    float* sahDev = reinterpret_cast<float*>(b.flags + 4);
    (void)hipMemsetAsync(sahDev, 0, 4, stream);
    hipLaunchKernelGGL(k_sah, gN, blk, 0, stream, b, nodeCount, sahDev);
    if (!m.boxes) hipLaunchKernelGGL(k_attrs, gT, blk, 0, stream, b);
    (void)hipEventRecord(ev1, stream);
    uint32_t flags[4] = { 0, 0, 0, 0 }, tail4[2] = { 0, 0 }; float sahSum = 0.0f; GpuNode rootNode;
    if ((e = hipMemcpyAsync(flags, b.flags, sizeof flags, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&sahSum, sahDev, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&rootNode, b.nodes, sizeof rootNode, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&tail4[0], b.index4 + (nodeCount - 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&tail4[1], b.even + (nodeCount - 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "GPU BVH build (emit)");
    if (flags[0]) return fail(hipErrorInvalidValue, "non-finite vertex position");
    const uint32_t node4Count = tail4[0] + tail4[1];
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, ev0, ev1);
    out.nodes = b.nodes; out.nodeCount = nodeCount; out.nodes4 = b.nodes4; out.node4Count = node4Count;
    out.tris = b.tris; out.attrs = b.attrs; out.tangents = b.tangents; out.triCount = n; out.leafOrder = b.valB;
    {   // root box = union of the root's two child boxes
        float dx = std::max(rootNode.lmax[0], rootNode.rmax[0]) - std::min(rootNode.lmin[0], rootNode.rmin[0]);
        float dy = std::max(rootNode.lmax[1], rootNode.rmax[1]) - std::min(rootNode.lmin[1], rootNode.rmin[1]);
        float dz = std::max(rootNode.lmax[2], rootNode.rmax[2]) - std::min(rootNode.lmin[2], rootNode.rmin[2]);
        float ra = dx * dy + dy * dz + dz * dx;
        out.sahCost = ra > 0.0f ? 1.0f + sahSum / ra : 0.0f;
    }
    out.maxDepth = maxDepthSeen; out.maxDepth4 = flags[2]; out.mortonBits = (uint32_t)usedBits; out.ploc = usedPloc; out.deviceMs = ms;
    m.last = b; m.haveTree = true; m.lastNodeCount = nodeCount; m.lastMaxDepth = maxDepthSeen; m.lastBits = (uint32_t)usedBits; m.lastPloc = usedPloc;
    return hipSuccess;
}

// ---- refit: new boxes on the hierarchy of the last build (hrpt_refit_instances: small motions). Runs the primitive set-up, the bottom-up box fit
// over the stored parent links, the 2-wide emission, the 4-wide collapse (its greedy choices follow the new boxes, so the 4-wide node count
// may change) and the attribute records; skips the Morton codes, the sort, the hierarchy (the ~300 launches of PLOC) and the numbering.
bool GpuBvhBuilder::can_refit() const { return p && p->haveTree; }
hipError_t GpuBvhBuilder::refit(const HrptPerInstanceData* instances, hipStream_t stream, GpuBuiltBvh& out, std::string& error)
{
    if (!p || p->boxes || !p->haveTree) { error = "GPU BVH builder has no tree to refit"; return hipErrorInvalidValue; }
    return refit_any(instances, nullptr, stream, out, error);
}
hipError_t GpuBvhBuilder::refit_boxes(const float* boxes, hipStream_t stream, GpuBuiltBvh& out, std::string& error)
{
    if (!p || !p->boxes || !p->haveTree) { error = "GPU BVH builder has no tree to refit"; return hipErrorInvalidValue; }
    return refit_any(nullptr, boxes, stream, out, error);
}
hipError_t GpuBvhBuilder::refit_any(const HrptPerInstanceData* instances, const float* boxes, hipStream_t stream, GpuBuiltBvh& out, std::string& error)
{
    out = GpuBuiltBvh();
    Impl& m = *p;
    hipError_t e;
    const BuildBuffers b = m.last;
    const uint32_t n = m.n, nodeCount = m.lastNodeCount;
    void* const prim = m.prim; size_t scanBytes = m.scanBytes;
    auto fail = [&](hipError_t err, const char* what) { error = what; m.haveTree = false; (void)hipStreamSynchronize(stream); return err; };
    if (m.boxes) {
        if ((e = hipMemcpyAsync(const_cast<float*>(b.boxesIn), boxes, (size_t)n * 24, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e, "hipMemcpyAsync(GPU BVH boxes)");
    } else {
        for (size_t i = 0; i < m.inst.size(); ++i) std::memcpy(m.inst[i].world, instances[i].m_World, sizeof m.inst[i].world);
        if ((e = hipMemcpyAsync(const_cast<InstRec*>(b.inst), m.inst.data(), m.inst.size() * sizeof(InstRec), hipMemcpyHostToDevice, stream)) != hipSuccess)
            return fail(e, "hipMemcpyAsync(GPU BVH instances)");
    }
    (void)hipEventRecord(m.ev0, stream);
    const dim3 gT((n + kB - 1) / kB), gN((nodeCount + kB - 1) / kB), blk(kB);
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, stream, b.sceneBounds, b.flags);
    if (m.boxes) hipLaunchKernelGGL(k_setup_boxes, gT, blk, 0, stream, b);
    else hipLaunchKernelGGL(k_setup, gT, blk, 0, stream, b);
    (void)hipMemsetAsync(b.visit, 0, (size_t)n * 4, stream);
    hipLaunchKernelGGL(k_fit, gT, blk, 0, stream, b);
    hipLaunchKernelGGL(k_emit2, gT, blk, 0, stream, b);
    bool greedy = true;
    if (const char* env = getenv("HRPT_GPU_BVH_COLLAPSE")) greedy = strcmp(env, "fixed") != 0;
    (void)hipMemsetAsync(b.even, 0, (size_t)nodeCount * 4, stream);
    hipLaunchKernelGGL(k_mark4_init, dim3(1), dim3(1), 0, stream, b);
    for (uint32_t d = 0; d <= m.lastMaxDepth; ++d) hipLaunchKernelGGL(k_mark4, gN, blk, 0, stream, b, nodeCount, d, greedy);
    if ((e = rocprim::exclusive_scan(prim, scanBytes, b.even, b.index4, 0u, nodeCount, rocprim::plus<uint32_t>(), stream)) != hipSuccess) return fail(e, "rocprim::exclusive_scan(even)");
    hipLaunchKernelGGL(k_emit4, gN, blk, 0, stream, b, nodeCount, greedy);
    float* sahDev = reinterpret_cast<float*>(b.flags + 4);
    (void)hipMemsetAsync(sahDev, 0, 4, stream);
    hipLaunchKernelGGL(k_sah, gN, blk, 0, stream, b, nodeCount, sahDev);
    if (!m.boxes) hipLaunchKernelGGL(k_attrs, gT, blk, 0, stream, b);
    (void)hipEventRecord(m.ev1, stream);
    uint32_t flags[4] = { 0, 0, 0, 0 }, tail4[2] = { 0, 0 }; float sahSum = 0.0f; GpuNode rootNode;
    if ((e = hipMemcpyAsync(flags, b.flags, sizeof flags, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&sahSum, sahDev, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&rootNode, b.nodes, sizeof rootNode, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&tail4[0], b.index4 + (nodeCount - 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipMemcpyAsync(&tail4[1], b.even + (nodeCount - 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "GPU BVH refit");
    if (flags[0]) return fail(hipErrorInvalidValue, "non-finite vertex position");
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, m.ev0, m.ev1);
    out.nodes = b.nodes; out.nodeCount = nodeCount; out.nodes4 = b.nodes4; out.node4Count = tail4[0] + tail4[1];
    out.tris = b.tris; out.attrs = b.attrs; out.tangents = b.tangents; out.triCount = n; out.leafOrder = b.valB;
    {
        float dx = std::max(rootNode.lmax[0], rootNode.rmax[0]) - std::min(rootNode.lmin[0], rootNode.rmin[0]);
        float dy = std::max(rootNode.lmax[1], rootNode.rmax[1]) - std::min(rootNode.lmin[1], rootNode.rmin[1]);
        float dz = std::max(rootNode.lmax[2], rootNode.rmax[2]) - std::min(rootNode.lmin[2], rootNode.rmin[2]);
        float ra = dx * dy + dy * dz + dz * dx;
        out.sahCost = ra > 0.0f ? 1.0f + sahSum / ra : 0.0f;
    }
    out.maxDepth = m.lastMaxDepth; out.maxDepth4 = flags[2]; out.mortonBits = m.lastBits; out.ploc = m.lastPloc; out.deviceMs = ms; out.refitted = true;
    return hipSuccess;
}

// ---- the tree over the INSTANCES of the two-level structure (box mode): copies the nodes to the front of the scene's node array with every
// leaf -- position k of the final leaf order -- turned into the instance reference the two-level traversal expects, ~(instance << 2)
__global__ __launch_bounds__(256) void k_tlas_fixup(const GpuNode4* __restrict__ src, uint32_t count, const uint32_t* __restrict__ leafOrder, GpuNode4* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    GpuNode4 n = src[i];
    int32_t* ch = &n.child.x;
    for (int c = 0; c < 4; ++c) {
        if (ch[c] == 0x7fffffff || ch[c] >= 0) continue;
        const uint32_t first = ((uint32_t)~ch[c]) >> 2;
        ch[c] = ~(int32_t)(leafOrder[first] << 2);
    }
    dst[i] = n;
}
hipError_t launch_tlas_fixup(const GpuNode4* src, uint32_t count, const uint32_t* leafOrder, GpuNode4* dst, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tlas_fixup, dim3((count + 255) / 256), dim3(256), 0, stream, src, count, leafOrder, dst);
    return hipGetLastError();
}

} // namespace hrt
