// pt_wavefront.hip -- the production path: a persistent-threads wavefront path tracer for gfx950.
//
// One hrpt_render = all requested accumulation indices ("spp") of a tile at once:
//   wf_raygen -> per bounce { wf_extend -> wf_shade -> wf_shadow } -> wf_resolve
// replacing the single PathTracer_CSMain dispatch per index of the reference
// (/root/reference/src/shaders/PathTracer.hlsl:53-340, src/PathTracerRenderer.cpp:96-103).
//
// Data layout in HBM (one pool, SoA of float4, all streams coalesced 16 B/lane):
//   path queues A/B (ping-pong per bounce): rayO{o.xyz,tmin} rayD{d.xyz,rng} thr{T.xyz,sample} [med0{sigmaA,ior} med1{sigmaS,inVolume}]
//   hit records of the current queue:       hit{t,u,v,triangle}  (triangle = leaf-order index, 0xFFFFFFFF = miss)
//   NEE / shadow queue (dense):             sh0{origin,sample} sh1{N,roughness} sh2{V,metallic} sh3{baseColor,ior} sh4{T,count}
//                                           + per light sample shL{ux,uy,light}: direction, visibility and the BRDF x radiance
//                                           evaluation are done by wf_shadow, where every lane has NEE work and occluded
//                                           samples skip the evaluation
//   sampleRadiance[sample] (rgb): one slot per (pixel, accumulation index); emissive / NEE / sky are added in path order,
//                                 wf_resolve then folds the indices in order exactly like progressive accumulation (:332-339).
// Queues are SEGMENTED: a segment is 1024 consecutive samples owned by ONE wave at a time. The owning wave
// compacts survivors inside its segment with __ballot + popcount prefix (no global atomics, no block barriers,
// deterministic layout, pixel tiles stay together), and writes the segment's live count. Every kernel is a fixed
// grid of persistent waves striding over segments; each wave reaches its exit when the segment index runs out.
// The BVH (nodes + world-space triangles) is copied to LDS when it fits, and the traversal stack lives in LDS,
// one column per lane (conflict-free), sized from the builder's maximum depth.
#define HRPT_PHASE_TU 1
#include "pt_wavefront.h"

#include <mutex>
#include <unordered_map>
#include <vector>

#include "bvh_build.h"
#include "pt_path.h"

#ifdef HRPT_PHASE_PROFILE
__device__ unsigned long long g_phaseCounters[128];
#endif
#ifdef HRPT_SKY_DEBUG       // diagnostic build: what the miss branch of wf_shade fed into / got out of the sky lookup, first 2048 samples x 16 floats
__device__ float g_skyDebug[2048 * 16];
#endif

namespace hrt {

namespace {

constexpr uint32_t kMaxSegment = 1024;       // largest wave-owned segment (samples); the size is chosen per batch, 64..1024
constexpr uint32_t kBlock = 256;             // 4 waves
constexpr uint32_t kMaxLights = 8;
constexpr uint32_t kMaxSppPerBatch = 64;
constexpr size_t kLdsBudget = 64 * 1024;     // dynamic LDS per block: traversal stacks + BVH copy

struct WfBuffers {
    float4 *rayO[2], *rayD[2], *thr[2], *med0[2], *med1[2];
    uint32_t* pathCnt[2];
    float4* hit;
    uint32_t* hitInst;         // two-level structure only: CommittedInstanceIndex of the hit in `hit` (the record's triangle index is per mesh)
    float4 *sh0, *sh1, *sh2, *sh3, *sh4, *shL;
    uint32_t* shadowCnt;
    // shadow-ray stage of scenes with non-opaque geometry: one compacted ray per valid light sample {o, tmin} {d, tmax}, the (entry, light)
    // slot it belongs to, rays per segment, and the outcome of the opaque any-hit pass per (entry, light) slot (kVis*)
    float4 *sqO, *sqD; uint32_t* sqId; uint32_t* sqCnt; uint32_t* shVis;
    uint2* sqCand;             // kShadowCandidates (t, triangle) keys per (entry, light) slot: the non-opaque triangles a ray crossed, nearest first
    float4* radiance;
};

struct WfArgs {
    SceneView scene;
    WfBuffers b;
    uint32_t segSize;          // samples per segment (any size from 64 to kMaxSegment: chosen per batch so that the segments divide evenly among the waves of the grids)
    uint32_t numSegments;      // segments in this batch
    uint32_t numSamples;       // tilesX*tilesY*64*spp (padded to 8x8 pixel tiles)
    uint32_t pixelsPadded;     // tilesX*tilesY*64
    uint32_t tilesX, tilesY;
    TileRect rect;
    uint32_t imageWidth;
    uint32_t spp;              // accumulation indices in this batch
    uint32_t maxLights;        // per-path light-sample slots in the shadow queue
    uint32_t hasMedium;        // scene has thick transmissive materials (medium state travels with the path)
    uint32_t hasStochasticAlpha;
    uint32_t allOpaque;        // no ForceNonOpaque instance in the scene: closest-hit launches take the OPQ instantiations of wf_extend
    uint32_t refillMin;        // wf_extend refills its idle lanes once at least this many have finished their ray
    uint32_t streamSegments;   // wf_extend moves on to its next segment while rays of the previous one are still in flight
    uint32_t slimShadow;       // slim shadow-queue entries (scenes of the SIMPLE shade variant with one light): sh0{origin, input slot} sh1{N, material}
    uint32_t shadowParity;     // ... and the path queue those slots index (the input queue of the bounce's wf_shade)
    uint32_t nodeLoopMin;      // wf_extend leaves its node-descent loop once fewer lanes than this are still at inner nodes (0: never)
    uint32_t sortShade;        // wf_shade (general variants) shades the entries of a segment grouped by shading class
    const struct PrimaryArgs* primaryArgs;     // device copy (queue pool), valid when `primary`
    uint32_t primary;          // this launch works on bounce 0 of a batch whose primary rays are NOT in the path queue: slot == sample index, every
                               // wf_extend<PRIMARY> derives the ray of a slot from PrimaryArgs and leaves {direction, RNG seed} in rayD; origin = camera, throughput (1,1,1)
                               // (no wf_raygen launch; SIMPLE scenes)
    int32_t* spill[2];         // per-lane stack overflow columns of wf_extend / wf_shadow (they run concurrently), element k of thread g at [k * threads + g]
    DeviceCounters* counters;
};

enum : uint32_t { kVisNoRay = 0, kVisBlocked = 1, kVisClear = 2, kVisCandidates = 3 };   // shVis codes

struct JitterTable { float2 j[kMaxSppPerBatch]; };
// what the primary ray of a sample is computed from (PathTracer.hlsl:61-72, PathTracerRenderer.cpp:62-65); by value to the bounce-0 launches
struct PrimaryArgs { float clipToWorld[16]; float cam[3]; float invW, invH; uint32_t accumIndex; float2 j[kMaxSppPerBatch]; };

// Tile enumeration: row-major over the rectangle; with interleaved columns (stripeCount > 1) column-major, so that the consecutive tiles
// of a segment stay neighbours in the image (8 pixels apart vertically instead of 8 * stripeCount horizontally: -3 % per rank at 8 ranks).
HRT_DEV void tile_position(const WfArgs& a, uint32_t tile, uint32_t& tcol, uint32_t& trow)
{
    if (a.rect.stripeCount > 1u) { tcol = tile / a.tilesY; trow = tile - tcol * a.tilesY; }
    else { trow = tile / a.tilesX; tcol = tile - trow * a.tilesX; }
}

// Pixel and primary ray of sample `smp` of the batch (the body of wf_raygen; with a.primary the bounce-0 kernels call it instead of reading the
// path queue). false: the sample's pixel lies outside the rectangle (tiles are padded to 8 x 8).
HRT_DEV bool primary_ray(const WfArgs& a, const PrimaryArgs& pr, uint32_t smp, f3& o, f3& d, uint32_t& rng)
{
    const uint32_t k = smp / a.pixelsPadded, p = smp - k * a.pixelsPadded;
    const uint32_t tile = p >> 6, within = p & 63u;
    uint32_t tcol, trow; tile_position(a, tile, tcol, trow);
    const uint32_t px = a.rect.column_x(tcol) + (within & 7u), py = a.rect.y0 + trow * 8u + (within >> 3);
    if (!(px < a.rect.x1 && py < a.rect.y1)) return false;
    // init_path (pt_path.h) with the jitter / RNG stream of accumulation index first + k (PathTracerRenderer.cpp:62,:65)
    const float u = (((float)px + 0.5f) + pr.j[k].x) * pr.invW;
    const float v = (((float)py + 0.5f) + pr.j[k].y) * pr.invH;
    const float cx = u * 2.0f + -1.0f, cy = v * -2.0f + 1.0f;
    const float* M = pr.clipToWorld;
    const float ex = ((cx * M[0] + cy * M[4]) + 0.9f * M[8]) + 1.0f * M[12];
    const float ey = ((cx * M[1] + cy * M[5]) + 0.9f * M[9]) + 1.0f * M[13];
    const float ez = ((cx * M[2] + cy * M[6]) + 0.9f * M[10]) + 1.0f * M[14];
    const float ew = ((cx * M[3] + cy * M[7]) + 0.9f * M[11]) + 1.0f * M[15];
    const f3 end = mk3(ex / ew, ey / ew, ez / ew);
    o = mk3(pr.cam[0], pr.cam[1], pr.cam[2]);
    d = normalize(end - o);
    rng = hrt_rng_seed(px, py, pr.accumIndex + k);
    return true;
}
// entries of segment `seg` of the input path queue: counted by the producing kernel, or -- bounce 0 without a raygen pass -- every slot of the batch
HRT_DEV uint32_t path_count(const WfArgs& a, uint32_t in, uint32_t seg)
{
    (void)in;
    const uint32_t base = seg * a.segSize, left = a.numSamples - base, size = a.segSize; return left < size ? left : size;
}

// per-lane traversal stack in LDS: element (sp, lane-in-block) at base[sp * kBlock]
// DEPTH 64 = "deeper than 32": the first 32 entries stay in LDS, the (rarely reached) rest lives in a per-lane column of global memory.
// A 64-entry LDS stack is 64 KB per block, i.e. two blocks per CU: it cost 40 % on the scenes that needed it, although the worst case
// 3 * depth4 + 2 that forces the size is never approached by real rays.
// Measured (MI355X): a 64-entry LDS stack -> 32 + spill: -33 % frame time on the 1.17 M-triangle scene; closest-hit kernel 32 -> 16 LDS
// entries + spill: another -3 % there and on config 4 (occupancy); the shadow kernel is faster with 32 (+4 % with 16 on configs 4, 5).
constexpr int kExtendLdsStack = 16, kShadowLdsStack = 32;
constexpr uint32_t kMaxStackNeed = 128;        // deepest supported 4-wide stack need (3 * depth4 + 2)
template <int DEPTH, int LDSMAX>
struct LdsStack {
    static constexpr int kLdsStackMax = LDSMAX;
    static constexpr int kLds = DEPTH > kLdsStackMax ? kLdsStackMax : DEPTH;
    static constexpr int kRows = kLds;
    int32_t* base; int32_t* spill; uint32_t spillStride;
    HRT_DEV void push(int sp, int32_t v)
    {
        if (DEPTH <= kLdsStackMax || sp < kLds) base[(sp & (kLds - 1)) * kBlock] = v;
        else spill[(size_t)(sp - kLds) * spillStride] = v;
    }
    HRT_DEV int32_t pop(int sp)
    {
        if (DEPTH <= kLdsStackMax || sp < kLds) return base[(sp & (kLds - 1)) * kBlock];
        return spill[(size_t)(sp - kLds) * spillStride];
    }
};
// per-lane buffer of the K closest non-opaque shadow candidates: (t, triangle) of entry k at base[(k*2 + {0,1}) * kBlock];
// the barycentrics are recomputed from the triangle when the candidate is processed (same test => same bits)
constexpr int kShadowCandidates = 8;
struct LdsCandidates {
    int32_t* base;
    HRT_DEV void key(int k, float& t, uint32_t& tri) const { t = __int_as_float(base[(k * 2 + 0) * kBlock]); tri = (uint32_t)base[(k * 2 + 1) * kBlock]; }
    HRT_DEV void set(int k, float t, uint32_t tri) { base[(k * 2 + 0) * kBlock] = __float_as_int(t); base[(k * 2 + 1) * kBlock] = (int32_t)tri; }
    HRT_DEV void move(int dst, int src) { base[(dst * 2 + 0) * kBlock] = base[(src * 2 + 0) * kBlock]; base[(dst * 2 + 1) * kBlock] = base[(src * 2 + 1) * kBlock]; }
};
// ... with the instance next to the triangle (two-level structure: the triangle index is per mesh): entry k at base[(k*3 + {0,1,2}) * kBlock]
constexpr int kTwoLevelCandidates = 4;
struct LdsCandidates3 {
    int32_t* base;
    HRT_DEV void key(int k, float& t, uint32_t& tri, uint32_t& inst) const { t = __int_as_float(base[(k * 3 + 0) * kBlock]); tri = (uint32_t)base[(k * 3 + 1) * kBlock]; inst = (uint32_t)base[(k * 3 + 2) * kBlock]; }
    HRT_DEV void set(int k, float t, uint32_t tri, uint32_t inst) { base[(k * 3 + 0) * kBlock] = __float_as_int(t); base[(k * 3 + 1) * kBlock] = (int32_t)tri; base[(k * 3 + 2) * kBlock] = (int32_t)inst; }
    HRT_DEV void move(int dst, int src) { for (int w = 0; w < 3; ++w) base[(dst * 3 + w) * kBlock] = base[(src * 3 + w) * kBlock]; }
};
// candidate list of one shadow ray in global memory (written by wf_extend<ANYHIT> when the ray finishes, read by wf_shadow)
struct GlobalCandidates {
    const uint2* base;
    HRT_DEV void key(int k, float& t, uint32_t& tri) const { uint2 v = base[k]; t = __uint_as_float(v.x); tri = v.y; }
};
// BVH copy in LDS; W = node width (2: GpuNode, 4: GpuNode4)
// Stride of a 4-wide node in the LDS copy. At 128 bytes the rows of all even nodes start in the same four banks (and those of the odd nodes in four
// others): lanes at different nodes conflict 8-fold at worst. A multiple of 32 keeps the near ^ 16 = far addressing of inner_step.
#ifndef HRPT_LDS_NODE4_STRIDE
#define HRPT_LDS_NODE4_STRIDE 128
#endif
constexpr uint32_t kLdsNode4Stride = HRPT_LDS_NODE4_STRIDE;
static_assert(kLdsNode4Stride >= 128 && kLdsNode4Stride % 32 == 0, "LDS node stride: 128 bytes of node, near / far rows 32-byte aligned");
template <int W>
struct LdsBvh {
    static constexpr int kWidth = W == 5 ? 4 : W; static constexpr bool kTwoLevel = false; static constexpr bool kLds = true;
    const float4* nodes; const float4* tris;
    HRT_DEV void node(int i, float4& a, float4& b, float4& c, float4& d) const { const float4* p = nodes + 4 * i; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
    HRT_DEV void tri(uint32_t i, float4& a, float4& b, float4& c) const { const float4* p = tris + 3 * i; a = p[0]; b = p[1]; c = p[2]; }
    // rows by 32-bit LDS address (the copy starts 128-byte aligned: setup_lds, so near ^ 16 is the far row of the same axis)
    typedef __attribute__((address_space(3))) const char* LdsPtr;
    HRT_DEV uint32_t rowoff(int i, uint32_t byteOffset) const { return (uint32_t)(uintptr_t)(LdsPtr) reinterpret_cast<const char*>(nodes) + (uint32_t)i * kLdsNode4Stride + byteOffset; }
    HRT_DEV float4 load(uint32_t off) const { return *reinterpret_cast<const float4*>((const char*)(LdsPtr)(uintptr_t)off); }
};
template <int W> struct GlobalBvhOf;
template <> struct GlobalBvhOf<2> { using type = GlobalBvh; static HRT_DEV GlobalBvh make(const SceneView& s) { GlobalBvh g; g.nodes = s.nodes; g.tris = s.tris; return g; } };
constexpr int kTwoLevelTree = 44;     // GlobalBvhOf key of the two-level structure (4-wide nodes)
template <> struct GlobalBvhOf<kTwoLevelTree> { using type = GlobalBvhTl; static HRT_DEV GlobalBvhTl make(const SceneView& s) { GlobalBvhTl g; g.nodes = s.nodes4; g.tris = s.tris; g.instances = s.instances; return g; } };
template <> struct GlobalBvhOf<4> { using type = GlobalBvh4; static HRT_DEV GlobalBvh4 make(const SceneView& s) { GlobalBvh4 g; g.nodes = s.nodes4; g.tris = s.tris; return g; } };
constexpr int kQuantisedTree = 5;     // template width code of the 4-wide tree through its 64-byte quantised nodes (global memory only; pt_device.h GpuNodeQ)
template <> struct GlobalBvhOf<kQuantisedTree> { using type = GlobalBvhQ; static HRT_DEV GlobalBvhQ make(const SceneView& s) { GlobalBvhQ g; g.nodes = s.nodesQ; g.tris = s.tris; return g; } };

// Carves dynamic LDS: [stack: min(DEPTH, 32)*kBlock ints][bvh copy]; copies the BVH when LDS_BVH.
template <bool LDS_BVH, int DEPTH, int W, int LDSMAX>
HRT_DEV void setup_lds(char* smem, const SceneView& s, LdsStack<DEPTH, LDSMAX>& stack, LdsBvh<W>& lbvh, size_t extraBytes = 0)
{
    stack.base = reinterpret_cast<int32_t*>(smem) + threadIdx.x;
    stack.spill = nullptr; stack.spillStride = 0;
    if (LDS_BVH) {
        float4* dst = reinterpret_cast<float4*>(smem + (size_t)LdsStack<DEPTH, LDSMAX>::kRows * kBlock * 4 + extraBytes);
        const float4* srcN = W == 2 ? reinterpret_cast<const float4*>(s.nodes) : reinterpret_cast<const float4*>(s.nodes4);
        const float4* srcT = reinterpret_cast<const float4*>(s.tris);
        uint32_t nN = W == 2 ? s.nodeCount * 4 : s.node4Count * 8, nT = s.triCount * 3;
        uint32_t nNdst = nN;                                   // float4s the node copy occupies
        if (W == 2 || kLdsNode4Stride == 128u) { for (uint32_t i = threadIdx.x; i < nN; i += kBlock) dst[i] = srcN[i]; }
        else { nNdst = s.node4Count * (kLdsNode4Stride / 16u); for (uint32_t i = threadIdx.x; i < nN; i += kBlock) dst[(i >> 3) * (kLdsNode4Stride / 16u) + (i & 7u)] = srcN[i]; }
        for (uint32_t i = threadIdx.x; i < nT; i += kBlock) dst[nNdst + i] = srcT[i];
        lbvh.nodes = dst; lbvh.tris = dst + nNdst;
        __syncthreads();
    }
}

HRT_DEV uint32_t lane_id() { return threadIdx.x & 63u; }
// A value every lane of the wave holds identically (segment cursors, counts read from memory, the wave's index), moved to a scalar
// register: the compiler cannot prove uniformity of anything derived from threadIdx or from a load, and would otherwise keep such cursors
// in VGPRs and turn every test on them into an exec-mask branch.
HRT_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
HRT_DEV uint32_t prefix_rank(unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }

HRT_DEV unsigned long long wave_sum_u32(unsigned int v)
{
    unsigned long long x = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
// Statistics: one atomic per BLOCK on a counter shard (a returning-free add per wave on one word costs ~11 ns each and
// serialises at the kernel tail: 8192 waves = ~0.1 ms per launch).
HRT_DEV void block_count_add(unsigned long long* counterField0, size_t fieldOffsetWords, unsigned int perLane)
{
    __shared__ unsigned long long partial[kBlock / 64];
    unsigned long long w = wave_sum_u32(perLane);
    if (lane_id() == 0) partial[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t i = 0; i < kBlock / 64; ++i) t += partial[i];
        if (t) atomicAdd(counterField0 + (size_t)(blockIdx.x % kCounterShards) * (sizeof(DeviceCounters) / 8) + fieldOffsetWords, t);
    }
}
// The same for N wave-uniform counts (each wave passes its own totals; fields = word indices into DeviceCounters).
template <int N>
HRT_DEV void block_count_add_uniform(DeviceCounters* counters, const int (&fields)[N], const unsigned int (&perWave)[N])
{
    __shared__ unsigned int partialU[N][kBlock / 64];
    if (lane_id() == 0)
        for (int k = 0; k < N; ++k) partialU[k][threadIdx.x >> 6] = perWave[k];
    __syncthreads();
    if (threadIdx.x < (uint32_t)N) {
        unsigned long long t = 0;
        for (uint32_t i = 0; i < kBlock / 64; ++i) t += partialU[threadIdx.x][i];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(counters + blockIdx.x % kCounterShards) + fields[threadIdx.x], t);
    }
}

// PrimaryArgs into the queue pool (the kernels index its jitter table per lane: a by-value kernel argument would be copied to scratch for that)
__global__ void wf_store_primary(PrimaryArgs pr, PrimaryArgs* dst)
{
    if (threadIdx.x < 16) dst->clipToWorld[threadIdx.x] = pr.clipToWorld[threadIdx.x];
    if (threadIdx.x < 3) dst->cam[threadIdx.x] = pr.cam[threadIdx.x];
    if (threadIdx.x == 0) { dst->invW = pr.invW; dst->invH = pr.invH; dst->accumIndex = pr.accumIndex; }
    if (threadIdx.x < kMaxSppPerBatch) dst->j[threadIdx.x] = pr.j[threadIdx.x];
}

// ------------------------------------------------------------------ raygen
__global__ __launch_bounds__(kBlock) void wf_raygen(WfArgs a, HrptPathTracerConstants cb, JitterTable jt)
{
    const uint32_t wavesPerBlock = kBlock / 64, lane = lane_id();
    const uint32_t gw = uniform(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)), totalWaves = gridDim.x * wavesPerBlock;
    unsigned int nPaths = 0;
    for (uint32_t seg = gw; seg < a.numSegments; seg += totalWaves) {
        uint32_t segBase = seg * a.segSize, outCount = 0;
        for (uint32_t base = 0; base < a.segSize; base += 64) {
            uint32_t smp = segBase + base + lane;
            const bool mine = base + lane < a.segSize && smp < a.numSamples;      // (a segment need not be a multiple of 64 samples)
            bool active = mine;
            uint32_t k = 0, px = 0, py = 0;
            if (active) {
                k = smp / a.pixelsPadded;
                uint32_t p = smp - k * a.pixelsPadded;
                uint32_t tile = p >> 6, within = p & 63u;
                uint32_t tcol, trow; tile_position(a, tile, tcol, trow);
                px = a.rect.column_x(tcol) + (within & 7u);
                py = a.rect.y0 + trow * 8u + (within >> 3);
                active = px < a.rect.x1 && py < a.rect.y1;
            }
            if (mine) a.b.radiance[smp] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            unsigned long long m = __ballot(active);
            if (active) {
                PathState ps;
                {
                    // init_path (pt_path.h) with the jitter / RNG stream of accumulation index first+k (PathTracerRenderer.cpp:62,:65)
                    float u = (((float)px + 0.5f) + jt.j[k].x) * cb.m_View.m_ViewportSizeInv[0];
                    float v = (((float)py + 0.5f) + jt.j[k].y) * cb.m_View.m_ViewportSizeInv[1];
                    float cx = u * 2.0f + -1.0f, cy = v * -2.0f + 1.0f;
                    const float* M = cb.m_View.m_MatClipToWorldNoOffset;
                    float ex = ((cx * M[0] + cy * M[4]) + 0.9f * M[8]) + 1.0f * M[12];
                    float ey = ((cx * M[1] + cy * M[5]) + 0.9f * M[9]) + 1.0f * M[13];
                    float ez = ((cx * M[2] + cy * M[6]) + 0.9f * M[10]) + 1.0f * M[14];
                    float ew = ((cx * M[3] + cy * M[7]) + 0.9f * M[11]) + 1.0f * M[15];
                    f3 end = mk3(ex / ew, ey / ew, ez / ew);
                    ps.ray.o = mk3(cb.m_CameraPos[0], cb.m_CameraPos[1], cb.m_CameraPos[2]);
                    ps.ray.d = normalize(end - ps.ray.o);
                    ps.rng = hrt_rng_seed(px, py, cb.m_AccumulationIndex + k);
                }
                uint32_t o = segBase + outCount + prefix_rank(m);
                a.b.rayO[0][o] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, 0.0f);
                a.b.rayD[0][o] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, __uint_as_float(ps.rng));
                a.b.thr[0][o] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(smp));
                if (a.hasMedium) {
                    a.b.med0[0][o] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
                    a.b.med1[0][o] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                ++nPaths;
            }
            outCount += (uint32_t)__popcll(m);
        }
        if (lane == 0) a.b.pathCnt[0][seg] = outCount;
    }
    block_count_add(&a.counters->closestRays, 2, nPaths);
}

// ------------------------------------------------------------------ extend (closest hit)
// Persistent while-while traversal with lane refill: a wave keeps up to 64 rays of its segment in flight; whenever
// at least kRefillMin lanes have finished their ray, they fetch the next rays of the segment (ballot + prefix rank),
// so the traversal loop runs with full lanes instead of waiting for the slowest ray of a 64-ray batch.
// Waves per SIMD the kernels are compiled for (amdgpu_waves_per_eu; measured on MI355X, built without the SLP vectoriser -- csrc/Makefile):
//   wf_extend, tree in LDS      6   80 VGPRs without spills; the <PRIMARY> instantiation gives up 10 registers for it (-1.5 %)
//   wf_extend, tree in global   6   80-82 VGPRs since nodes and triangles are addressed by 32-bit offsets (84 and +2 % at 6 waves before): config 4 -2 %
//   wf_extend<TL>               5   100-117 VGPRs: a few spilled registers buy the fifth wave (-4 %); with the SLP vectoriser on the same
//                                   setting doubled the kernel's time (128 VGPRs wanted)
//   wf_shade, any variant       4   general single-light variant: 121-129 VGPRs wanted, -14 % against 3 waves. The same setting produced wrong sky
//                                   radiance while the SLP vectoriser was on (138 VGPRs wanted). Root cause (round 3, profiles/r03_wrong_sky_isa_evidence.txt):
//                                   a backend bug under register pressure -- the hoisted constant of mie_phase was kept alive through a live-range-split
//                                   copy placed in ONE arm of a divergent if/else, so the lanes of the other arm restored garbage. The guards: the random
//                                   trait scenes of the GPU suite (seed 1 fails on that build) and tests/test_kernel_resources.py (no scratch, <= 128
//                                   VGPRs). Re-check both whenever this kernel, the flags or the compiler change. 5 waves on <SIMPLE>: no gain
constexpr int kWavesExtendLds = 6, kWavesExtendGlobal = 6, kWavesExtendTwoLevel = 5, kWavesShade = 4;
#ifndef HRPT_WAVES_EXTEND_LDS_OPAQUE
#define HRPT_WAVES_EXTEND_LDS_OPAQUE 6
#endif
constexpr int kWavesExtendLdsOpaque = HRPT_WAVES_EXTEND_LDS_OPAQUE;
constexpr uint32_t kRefillMinDefault = 12;
constexpr uint32_t kShadeRing = 64;        // entries of wf_shade<SIMPLE>'s per-wave ring of parked specular-lobe paths
constexpr uint32_t kNoPathRecord = 0xFFFFFFFEu;     // hit-record code of a slot without a path (wf_extend<PRIMARY>; 0xFFFFFFFF = miss)

// ANYHIT: the same persistent loop over the shadow-ray queue (sqO / sqD / sqId, sqCnt rays per segment): the first hit on an opaque
// triangle ends the ray (kVisBlocked); otherwise the ray is clear or, if it crossed non-opaque triangles, left to wf_shadow's candidate
// pass (kVisCandidates). Shadow rays get the lane refill closest-hit rays have: 3.8 -> 8 Grays/s on the glass config.
// TL: the two-level structure of instanced scenes (pt_device.h "two-level traversal"): tree in global memory, closest hits only, every
// instance opaque; the hit's instance goes to its own stream (hitInst) next to the hit record.
// PRIMARY: bounce 0 of a batch without a raygen pass (WfArgs::primary): the refill derives the ray from the sample index. A separate instantiation:
// as a run-time branch the extra live state cost the kernel 6 VGPRs and 3 % on EVERY bounce.
// TL: 0 flat tree, 1 two-level structure with ForceOpaque instances only, 2 two-level with non-opaque instances (candidate re-trace compiled in)
// OPQ: no instance of the scene is ForceNonOpaque (closest-hit launches of the flat structure): the candidate rules of TraceRayStandard (lower
// bound of a re-trace, stochastic-alpha draws) are compiled out -- every hit is committed.
template <bool LDS_BVH, int DEPTH, int W, bool ANYHIT, int TL = 0, bool PRIMARY = false, bool OPQ = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TL ? kWavesExtendTwoLevel : (LDS_BVH ? (OPQ ? kWavesExtendLdsOpaque : kWavesExtendLds) : kWavesExtendGlobal)))) void wf_extend(WfArgs a, uint32_t parity)
{
    static_assert(!OPQ || (!ANYHIT && !TL), "OPQ: closest hits over the flat structure");
    static_assert(!TL || (!LDS_BVH && !ANYHIT && W == 4), "two-level traversal: global 4-wide tree, closest hit");
    static_assert(!PRIMARY || !ANYHIT, "primary rays are closest-hit rays");
    static_assert(W != kQuantisedTree || (!LDS_BVH && !TL), "quantised nodes: flat tree in global memory");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    LdsStack<DEPTH, kExtendLdsStack> stack; LdsBvh<W> lbvh;
    constexpr size_t candBytes = ANYHIT ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0;
    setup_lds<LDS_BVH, DEPTH, W>(smem, a.scene, stack, lbvh, candBytes);
    LdsCandidates cand; cand.base = reinterpret_cast<int32_t*>(smem + (size_t)LdsStack<DEPTH, kExtendLdsStack>::kRows * kBlock * 4) + threadIdx.x;
    if (DEPTH > kExtendLdsStack) { stack.spill = a.spill[ANYHIT ? 1 : 0] + blockIdx.x * kBlock + threadIdx.x; stack.spillStride = gridDim.x * kBlock; }
    typename GlobalBvhOf<(TL ? kTwoLevelTree : W)>::type gbvh = GlobalBvhOf<(TL ? kTwoLevelTree : W)>::make(a.scene);
    const SceneView& s = a.scene;

    const uint32_t wavesPerBlock = kBlock / 64;
    const uint32_t gw = uniform(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)), totalWaves = gridDim.x * wavesPerBlock;
    const float4* __restrict__ rayO = ANYHIT ? a.b.sqO : a.b.rayO[parity];
    float4* __restrict__ rayD = ANYHIT ? a.b.sqD : a.b.rayD[parity];
    const uint32_t* __restrict__ segCount = ANYHIT ? a.b.sqCnt : a.b.pathCnt[parity];
    const uint32_t slotsPerSample = ANYHIT ? a.maxLights : 1u;
    unsigned int nRays = 0;
    const bool emptyScene = s.nodeCount == 0 && s.rootLeaf == 0;

    // The wave streams through its segments (gw, gw + totalWaves, ...) without draining between them: when one segment has been
    // handed out completely the idle lanes refill from the next one, so only the very end of the wave's work runs with few lanes.
    {
        uint32_t seg = gw, cnt = 0, segBase = 0, next = 0;      // wave-uniform cursor: next ray of the current segment to hand out
        bool haveSeg = false;
        auto open_segment = [&]() {                             // first non-empty segment at or after `seg`
            haveSeg = false; cnt = 0; next = 0;
            for (; seg < a.numSegments; seg += totalWaves) {
                cnt = uniform(PRIMARY ? path_count(a, 0u, seg) : segCount[seg]);
                if (cnt) { segBase = (seg * a.segSize) * slotsPerSample; haveSeg = true; break; }
            }
        };
        open_segment();
        // per-lane traversal state
        bool active = false;
        Ray r; r.o = mk3(0.0f, 0.0f, 0.0f); r.d = mk3(0.0f, 0.0f, 1.0f); r.tmin = 0.0f; r.tmax = 1e10f;
        RayShear sh = make_shear(r.d); f3 inv = mk3(0.0f, 0.0f, 0.0f), noi = mk3(0.0f, 0.0f, 0.0f);
        TlCull tl; tl.inv = inv; tl.noi = noi; tl.noiF = noi; tl.inst = -1;      // TL only (dead otherwise)
        HitKey lower; lower.have = false; lower.t = 0.0f; lower.inst = 0; lower.prim = 0;
        Hit best; best.valid = false; best.t = 0.0f; best.inst = 0; best.prim = 0; best.u = 0.0f; best.v = 0.0f; best.opaque = 0; best.tri = 0;
        int32_t cur = kTraversalDone; int sp = 0; uint32_t slot = 0, rng = 0, rng0 = 0; float tlim = 0.0f;
        bool blocked = false, candOverflow = false; int candCount = 0;
        for (;;) {
            if (haveSeg && next >= cnt && (a.streamSegments || __ballot(active) == 0ull)) { seg += totalWaves; open_segment(); }
            // ---- refill idle lanes
            unsigned long long mIdle = __ballot(!active);
            uint32_t nIdle = (uint32_t)__popcll(mIdle);
            if (active) HRT_PHASE(ANYHIT ? PH_ANY_ITER : PH_EXT_ITER);
            if (haveSeg && (nIdle >= a.refillMin || nIdle == 64u)) {
                const uint32_t idx = next + prefix_rank(mIdle);
                bool take = !active && idx < cnt;
                if constexpr (PRIMARY) {                // bounce 0, rays from the sample index (false: padding pixel of an 8 x 8 tile)
                    if (take) {
                        slot = segBase + idx; take = primary_ray(a, *a.primaryArgs, slot, r.o, r.d, rng); r.tmin = 0.0f; r.tmax = 1e10f; rng0 = rng;
                        if (take) rayD[slot] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(rng));      // for wf_shade(0) / wf_shadow(0): direction + RNG seed
                        else a.b.hit[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kNoPathRecord));   // padding pixel of an 8 x 8 tile: wf_shade skips the slot
                    }
                }
                if (take) {
                    HRT_PHASE(ANYHIT ? PH_ANY_REFILL : PH_EXT_REFILL);
                    if constexpr (!PRIMARY) {
                        slot = segBase + idx;
                        float4 o = rayO[slot], d = rayD[slot];
                        r.o = mk3(o.x, o.y, o.z); r.d = mk3(d.x, d.y, d.z); r.tmin = o.w; r.tmax = ANYHIT ? d.w : 1e10f;
                        rng = __float_as_uint(d.w); rng0 = rng;
                    }
                    blocked = false; candOverflow = false; candCount = 0;
                    lower.have = false;
                    best.valid = false; tlim = r.tmax; sp = 0;
                    bool finite = (r.d.x == r.d.x && r.d.y == r.d.y && r.d.z == r.d.z);
                    sh = make_shear(r.d);
                    if constexpr (TL) tl_world(tl, r); else { inv = traversal_rcp(r.d); noi = slab_origin_term(r.o, inv); }
                    cur = (emptyScene || !finite) ? kTraversalDone : (s.nodeCount == 0 ? s.rootLeaf : 0);
                    active = true; ++nRays;
                }
                next += nIdle;
            }
            if (__ballot(active) == 0ull) { if (haveSeg) continue; break; }
            if (active) {
                // ---- descend inner nodes until this lane holds a leaf (or its stack ran out)
                // (the descent is cut short once fewer than nodeLoopMin lanes are still at inner nodes: the rest of the wave holds leaves and
                // would only wait; the lanes cut off keep their node and go on in the next round. Thresholded while-while.)
                if constexpr (TL) {
                    while (cur >= 0 && cur != kExitBlas) {
                        cur = inner_step(gbvh, cur, tl.noi, tl.noiF, tl.inv, r.tmin, tlim, stack, sp);
                        if ((uint32_t)__popcll(__ballot(cur >= 0 && cur != kExitBlas)) < a.nodeLoopMin) break;
                    }
                    // leaving an instance / entering one: the lane continues with the next tree in the next round
                    if (cur == kExitBlas || (cur < 0 && cur != kTraversalDone && tl.inst < 0)) cur = tl_switch(gbvh, cur, tl, r, stack, sp);
                    else if (cur < 0 && cur != kTraversalDone) {
                        const GpuInstance& I = s.instances[tl.inst];
                        const uint32_t enc = (uint32_t)(~cur), first = enc >> 2, count = (enc & 3u) + 1u;
                        const uint32_t inst = (uint32_t)tl.inst, iflags = I.flags & 7u;
                        for (uint32_t i = 0; i < count; ++i) {
                            float4 ta, tb, tc; gbvh.tri(first + i, ta, tb, tc);
                            f3 p0, p1, p2; tl_world_triangle(I, ta, tb, tc, p0, p1, p2);
                            float t, u, v;
                            const bool hitTri = tri_test(p0, p1, p2, r, sh, t, u, v);
                            const uint32_t prim = __float_as_uint(tb.w);
                            // Written as selects on one flag (every closest-hit loop of the library is). As `if (take) { best.t = t; ... }` this loop was
                            // compiled (ROCm 7.2, gfx950) to code that, on a tie in t won by (instance, primitive), took the new key but kept the OLD
                            // barycentrics: hits on an edge shared by two triangles then shaded with the other triangle's (u, v).
                            // tests/test_two_level_gpu.py::test_two_level_full_frame holds such a pixel; the select form is also 1 % faster.
                            const bool ok = TL != 2 || !lower.have || key_less(lower.t, lower.inst, lower.prim, t, inst, prim);       // behind a rejected non-opaque candidate
                            const bool take = hitTri && ok && (!best.valid || key_less(t, inst, prim, best.t, best.inst, best.prim));
                            best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                            best.inst = take ? inst : best.inst; best.prim = take ? prim : best.prim; best.tri = take ? first + i : best.tri;
                            best.opaque = take ? iflags : best.opaque;
                            best.valid = best.valid || take;
                            tlim = take ? t : tlim;
                        }
                        cur = stack.pop(--sp);      // the exit marker is below every entry pushed inside an instance
                    }
                } else {
                while (cur >= 0) {
                    HRT_PHASE(ANYHIT ? PH_ANY_NODE : PH_EXT_NODE);
                    if (LDS_BVH) cur = inner_step(lbvh, cur, noi, inv, r.tmin, tlim, stack, sp);
                    else cur = inner_step(gbvh, cur, noi, inv, r.tmin, tlim, stack, sp);
                    if ((uint32_t)__popcll(__ballot(cur >= 0)) < a.nodeLoopMin) break;
                }
                }
                // ---- intersect the leaf
                if (!TL && cur < 0 && cur != kTraversalDone) {
                    uint32_t enc = (uint32_t)(~cur);
                    uint32_t first = enc >> 2, count = (enc & 3u) + 1u;
                    HRT_PHASE(ANYHIT ? PH_ANY_LEAF : PH_EXT_LEAF);
                    for (uint32_t i = 0; i < count; ++i) {
                        HRT_PHASE(ANYHIT ? PH_ANY_TRI : PH_EXT_TRI);
                        float4 ta, tb, tc;
                        if (LDS_BVH) lbvh.tri(first + i, ta, tb, tc); else gbvh.tri(first + i, ta, tb, tc);
                        float t, u, v;
                        if (tri_test(mk3(ta.x, ta.y, ta.z), mk3(tb.x, tb.y, tb.z), mk3(tc.x, tc.y, tc.z), r, sh, t, u, v)) {
                            if (ANYHIT) {
                                if (__float_as_uint(tc.w) & 1u) { blocked = true; break; }     // opaque instance: committed, whatever lies in front of it
                                if (LDS_BVH) candidate_insert<kShadowCandidates>(lbvh, cand, candCount, candOverflow, t, first + i, __float_as_uint(ta.w), __float_as_uint(tb.w));
                                else candidate_insert<kShadowCandidates>(gbvh, cand, candCount, candOverflow, t, first + i, __float_as_uint(ta.w), __float_as_uint(tb.w));
                                continue;
                            }
                            uint32_t inst = __float_as_uint(ta.w), prim = __float_as_uint(tb.w);
                            const bool ok = OPQ || !lower.have || key_less(lower.t, lower.inst, lower.prim, t, inst, prim);
                            const bool take = ok && (!best.valid || key_less(t, inst, prim, best.t, best.inst, best.prim));      // selects: see the two-level loop above
                            best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                            best.inst = take ? inst : best.inst; best.prim = take ? prim : best.prim; best.tri = take ? first + i : best.tri;
                            best.opaque = take ? (__float_as_uint(tc.w) & 7u) : best.opaque;     // bit 0 opaque, bits 1-2 shading class
                            best.valid = best.valid || take;
                            tlim = take ? t : tlim;
                        }
                    }
                    cur = (blocked || sp == 0) ? kTraversalDone : stack.pop(--sp);
                }
                // ---- traversal finished: candidate resolution (TraceRayStandard) and hit record
                if (ANYHIT) {
                    if (cur == kTraversalDone) {
                        HRT_PHASE(PH_ANY_FINISH);
                        const uint32_t id = a.b.sqId[slot];
                        uint32_t code = blocked ? kVisBlocked : kVisClear;
                        if (!blocked && candCount > 0) {
                            uint2* out = a.b.sqCand + (size_t)id * kShadowCandidates;
                            for (int k = 0; k < candCount; ++k) { float ct; uint32_t ctri; cand.key(k, ct, ctri); out[k] = make_uint2(__float_as_uint(ct), ctri); }
                            code = kVisCandidates | ((uint32_t)candCount << 8) | (candOverflow ? 1u << 16 : 0u);
                        }
                        a.b.shVis[id] = code;
                        active = false;
                    }
                } else if (cur == kTraversalDone) {
                    bool done = true;
                    HRT_PHASE(PH_EXT_FINISH);
                    if (!OPQ && TL != 1 && best.valid && !(best.opaque & 1u)) HRT_PHASE(PH_EXT_CANDIDATE);
                    if (!OPQ && TL != 1 && best.valid && !(best.opaque & 1u) && !candidate_commits(s, best, rng)) {
                        // rejected non-opaque candidate: it becomes the exclusive lower bound of a new closest-hit query
                        lower.have = true; lower.t = best.t; lower.inst = best.inst; lower.prim = best.prim;
                        best.valid = false; tlim = r.tmax; sp = 0;
                        cur = s.nodeCount == 0 ? s.rootLeaf : 0;
                        if constexpr (TL) tl_world(tl, r);
                        done = false;
                    }
                    if (done) {
                        if (!OPQ && a.hasStochasticAlpha && rng != rng0) { float4 d = rayD[slot]; d.w = __uint_as_float(rng); rayD[slot] = d; }
                        // hit record: triangle (< 2^29: leaf references hold first << 2) | shading class << 29; 0xFFFFFFFF = miss
                        a.b.hit[slot] = make_float4(best.t, best.u, best.v, __uint_as_float(best.valid ? (best.tri | ((best.opaque >> 1) << 29)) : 0xFFFFFFFFu));
                        if constexpr (TL) a.b.hitInst[slot] = best.inst;
                        active = false;
                    }
                }
            }
        }
    }
    if (!ANYHIT) block_count_add(&a.counters->closestRays, 0, nRays);      // shadow rays are counted by wf_shadow
    if (PRIMARY) { __syncthreads(); block_count_add(&a.counters->closestRays, 2, nRays); }      // ... and the paths wf_raygen would have counted
}

// ------------------------------------------------------------------ stand-alone ray queries (hrpt_trace_rays) through the same persistent loop
// TraceRayStandard (RaytracingCommon.hlsli:138-198) / CalculateRTShadow<true> (CommonLighting.hlsli:380-496) for the reference's other inline-RT
// passes (DDGI ProbeTraceCS.hlsl:60-61,108-109, RT shadows, BrdfRayTracing.hlsl:138-140): a caller's array of HrptRay instead of the path queue,
// HrptRayHit records instead of hit records. A wave owns chunks of 256 consecutive rays and refills idle lanes exactly like wf_extend; the
// thread-per-ray kernel of round 1 (pt_megakernel.hip) was 1.8x slower on the same rays. SHADOW: the any-hit traversal gathers the non-opaque
// triangles a ray crosses in per-lane LDS columns; when the ray ends unblocked they are resolved front to back right there.
struct WfTraceArgs {
    SceneView scene; const HrptRay* rays; HrptRayHit* hits; uint64_t count;
    int32_t* spill; uint32_t refillMin, nodeLoopMin;
    bool quantised;         // host-side only: the launch picks the instantiation that walks SceneView::nodesQ
};
constexpr uint32_t kTraceChunkShift = 8;
template <bool LDS_BVH, int DEPTH, int W, bool SHADOW, bool TL = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(SHADOW ? 1 : (TL ? kWavesExtendTwoLevel : (LDS_BVH ? kWavesExtendLds : kWavesExtendGlobal))))) void wf_trace_rays(WfTraceArgs a)
{
    static_assert(!TL || (!LDS_BVH && W == 4), "two-level structure: global 4-wide trees (every instance opaque: a shadow ray ends at its first hit)");
    static_assert(W != kQuantisedTree || (!LDS_BVH && !TL), "quantised nodes: flat tree in global memory");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    LdsStack<DEPTH, kExtendLdsStack> stack; LdsBvh<W> lbvh;
    constexpr size_t candBytes = SHADOW ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0;
    setup_lds<LDS_BVH, DEPTH, W>(smem, a.scene, stack, lbvh, candBytes);
    LdsCandidates cand; cand.base = reinterpret_cast<int32_t*>(smem + (size_t)LdsStack<DEPTH, kExtendLdsStack>::kRows * kBlock * 4) + threadIdx.x;
    if (DEPTH > kExtendLdsStack) { stack.spill = a.spill + blockIdx.x * kBlock + threadIdx.x; stack.spillStride = gridDim.x * kBlock; }
    typename GlobalBvhOf<(TL ? kTwoLevelTree : W)>::type gbvh = GlobalBvhOf<(TL ? kTwoLevelTree : W)>::make(a.scene);
    const SceneView& s = a.scene;
    const uint32_t wavesPerBlock = kBlock / 64;
    const uint32_t gw = uniform(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)), totalWaves = gridDim.x * wavesPerBlock;
    TlCull tl; tl.inv = mk3(0.0f, 0.0f, 0.0f); tl.noi = tl.inv; tl.noiF = tl.inv; tl.inst = -1;      // TL only
    const uint64_t numChunks = (a.count + (1u << kTraceChunkShift) - 1) >> kTraceChunkShift;
    const bool emptyScene = s.nodeCount == 0 && s.rootLeaf == 0;

    uint64_t chunk = gw; uint32_t cnt = 0, next = 0; uint64_t chunkBase = 0; bool haveChunk = false;
    auto open_chunk = [&]() {
        haveChunk = chunk < numChunks; next = 0;
        if (haveChunk) { chunkBase = chunk << kTraceChunkShift; const uint64_t left = a.count - chunkBase; cnt = left < (1u << kTraceChunkShift) ? (uint32_t)left : (1u << kTraceChunkShift); }
    };
    open_chunk();
    bool active = false;
    Ray r; r.o = mk3(0.0f, 0.0f, 0.0f); r.d = mk3(0.0f, 0.0f, 1.0f); r.tmin = 0.0f; r.tmax = 1e10f;
    RayShear sh = make_shear(r.d); f3 inv = mk3(0.0f, 0.0f, 0.0f), noi = mk3(0.0f, 0.0f, 0.0f);
    HitKey lower; lower.have = false; lower.t = 0.0f; lower.inst = 0; lower.prim = 0;
    Hit best; best.valid = false; best.t = 0.0f; best.inst = 0; best.prim = 0; best.u = 0.0f; best.v = 0.0f; best.opaque = 0; best.tri = 0;
    int32_t cur = kTraversalDone; int sp = 0; uint64_t slot = 0; uint32_t rng = 0; float tlim = 0.0f;
    bool blocked = false, candOverflow = false, finite = true; int candCount = 0;
    for (;;) {
        if (haveChunk && next >= cnt) { chunk += totalWaves; open_chunk(); }
        const unsigned long long mIdle = __ballot(!active);
        const uint32_t nIdle = (uint32_t)__popcll(mIdle);
        if (haveChunk && (nIdle >= a.refillMin || nIdle == 64u)) {
            const uint32_t idx = next + prefix_rank(mIdle);
            if (!active && idx < cnt) {
                slot = chunkBase + idx;
                const float4* rp = reinterpret_cast<const float4*>(a.rays + slot);           // 48-byte records: three 16-byte loads
                const float4 q0 = rp[0], q1 = rp[1]; const uint32_t rr = reinterpret_cast<const uint32_t*>(rp + 2)[0];
                const f3 o = mk3(q0.x, q0.y, q0.z), d = mk3(q1.x, q1.y, q1.z);
                finite = d.x == d.x && d.y == d.y && d.z == d.z && o.x == o.x && o.y == o.y && o.z == o.z;
                if (SHADOW) r = shadow_ray(o, d, q1.w);                                        // the query applies its own bias; tmax = distance to the light
                else { r.o = o; r.d = d; r.tmin = q0.w; r.tmax = q1.w; }
                rng = rr; blocked = false; candOverflow = false; candCount = 0; lower.have = false; best.valid = false; tlim = r.tmax; sp = 0;
                sh = make_shear(r.d);
                if constexpr (TL) tl_world(tl, r); else { inv = traversal_rcp(r.d); noi = slab_origin_term(r.o, inv); }
                cur = (emptyScene || !finite) ? kTraversalDone : (s.nodeCount == 0 ? s.rootLeaf : 0);
                active = true;
            }
            next += nIdle;
        }
        if (__ballot(active) == 0ull) { if (haveChunk) continue; break; }
        if (active) {
            if constexpr (TL) {
                while (cur >= 0 && cur != kExitBlas) {
                    cur = inner_step(gbvh, cur, tl.noi, tl.noiF, tl.inv, r.tmin, tlim, stack, sp);
                    if ((uint32_t)__popcll(__ballot(cur >= 0 && cur != kExitBlas)) < a.nodeLoopMin) break;
                }
                if (cur == kExitBlas || (cur < 0 && cur != kTraversalDone && tl.inst < 0)) cur = tl_switch(gbvh, cur, tl, r, stack, sp);
                else if (cur < 0 && cur != kTraversalDone) {
                    const GpuInstance& I = s.instances[tl.inst];
                    const uint32_t enc = (uint32_t)(~cur), first = enc >> 2, count = (enc & 3u) + 1u, inst = (uint32_t)tl.inst;
                    for (uint32_t i = 0; i < count; ++i) {
                        float4 ta, tb, tc; gbvh.tri(first + i, ta, tb, tc);
                        f3 p0, p1, p2; tl_world_triangle(I, ta, tb, tc, p0, p1, p2);
                        float t, u, v;
                        const bool hitTri = tri_test(p0, p1, p2, r, sh, t, u, v);
                        if (SHADOW) { if (hitTri) { if (I.flags & 1u) { blocked = true; break; } candCount = 1; } continue; }      // candCount: "crossed a non-opaque instance"
                        const uint32_t prim = __float_as_uint(tb.w);
                        const bool ok = !lower.have || key_less(lower.t, lower.inst, lower.prim, t, inst, prim);
                        const bool take = hitTri && ok && (!best.valid || key_less(t, inst, prim, best.t, best.inst, best.prim));      // selects: see wf_extend<TL>
                        best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                        best.inst = take ? inst : best.inst; best.prim = take ? prim : best.prim; best.tri = take ? first + i : best.tri;
                        best.valid = best.valid || take; best.opaque = take ? (I.flags & 1u) : best.opaque;
                        tlim = take ? t : tlim;
                    }
                    cur = blocked ? kTraversalDone : stack.pop(--sp);
                }
            } else {
            while (cur >= 0) {
                if (LDS_BVH) cur = inner_step(lbvh, cur, noi, inv, r.tmin, tlim, stack, sp);
                else cur = inner_step(gbvh, cur, noi, inv, r.tmin, tlim, stack, sp);
                if ((uint32_t)__popcll(__ballot(cur >= 0)) < a.nodeLoopMin) break;
            }
            if (cur < 0 && cur != kTraversalDone) {
                const uint32_t enc = (uint32_t)(~cur), first = enc >> 2, count = (enc & 3u) + 1u;
                for (uint32_t i = 0; i < count; ++i) {
                    float4 ta, tb, tc;
                    if (LDS_BVH) lbvh.tri(first + i, ta, tb, tc); else gbvh.tri(first + i, ta, tb, tc);
                    float t, u, v;
                    if (tri_test(mk3(ta.x, ta.y, ta.z), mk3(tb.x, tb.y, tb.z), mk3(tc.x, tc.y, tc.z), r, sh, t, u, v)) {
                        if (SHADOW) {
                            if (__float_as_uint(tc.w) & 1u) { blocked = true; break; }
                            if (LDS_BVH) candidate_insert<kShadowCandidates>(lbvh, cand, candCount, candOverflow, t, first + i, __float_as_uint(ta.w), __float_as_uint(tb.w));
                            else candidate_insert<kShadowCandidates>(gbvh, cand, candCount, candOverflow, t, first + i, __float_as_uint(ta.w), __float_as_uint(tb.w));
                            continue;
                        }
                        const uint32_t inst = __float_as_uint(ta.w), prim = __float_as_uint(tb.w);
                        const bool ok = !lower.have || key_less(lower.t, lower.inst, lower.prim, t, inst, prim);
                        const bool take = ok && (!best.valid || key_less(t, inst, prim, best.t, best.inst, best.prim));      // selects: see wf_extend
                        best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                        best.inst = take ? inst : best.inst; best.prim = take ? prim : best.prim; best.tri = take ? first + i : best.tri;
                        best.opaque = take ? (__float_as_uint(tc.w) & 1u) : best.opaque;
                        best.valid = best.valid || take;
                        tlim = take ? t : tlim;
                    }
                }
                cur = (blocked || sp == 0) ? kTraversalDone : stack.pop(--sp);
            }
            }
            if (cur == kTraversalDone) {
                HrptRayHit out; out.t = 0.0f; out.u = 0.0f; out.v = 0.0f; out.instance = 0; out.primitive = 0; out.hit = 0; out.rng = rng; out.pad = 0;
                bool done = true;
                if (SHADOW) {
                    float vis = 1.0f;
                    if (finite) {
                        if (blocked) vis = 0.0f;
                        else if (candCount > 0) {
                            if constexpr (TL) vis = shadow_retrace_two_level(s, gbvh, r, stack);
                            else if (LDS_BVH) vis = shadow_resolve_candidates(s, lbvh, r, sh, candCount, candOverflow, cand, stack);
                            else vis = shadow_resolve_candidates(s, gbvh, r, sh, candCount, candOverflow, cand, stack);
                        }
                    }
                    out.t = vis; out.hit = vis < 1.0f ? 1u : 0u;
                } else if (best.valid && !best.opaque && !candidate_commits(s, best, rng)) {
                    lower.have = true; lower.t = best.t; lower.inst = best.inst; lower.prim = best.prim;
                    best.valid = false; tlim = r.tmax; sp = 0;
                    cur = s.nodeCount == 0 ? s.rootLeaf : 0;
                    if constexpr (TL) tl_world(tl, r);
                    done = false;
                } else if (best.valid) { out.t = best.t; out.u = best.u; out.v = best.v; out.instance = best.inst; out.primitive = best.prim; out.hit = 1u; out.rng = rng; }
                if (done) {
                    out.rng = rng;
                    float4* hp = reinterpret_cast<float4*>(a.hits + slot);
                    hp[0] = make_float4(out.t, out.u, out.v, __uint_as_float(out.instance));
                    hp[1] = make_float4(__uint_as_float(out.primitive), __uint_as_float(out.hit), __uint_as_float(out.rng), 0.0f);
                    active = false;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ shade (+ compaction, + NEE sample emission)
template <int MAXL>
struct NeeBuf { float ux[MAXL], uy[MAXL]; uint32_t light[MAXL]; };

// SIMPLE: scene traits proven at upload -- no textures, no transmissive / BLEND material, directional lights only --
// compile the corresponding branches out (the general variant is always correct).
// Occupancy: the multi-light variant is forced to 4 waves per SIMD (28 B of scratch per lane; -9 % on the glass config); the single-light
// variants stay at 3 (forcing 4 costs them +3 %: 44..108 B of spills on a kernel that is already latency-bound).
// MAXL = 0: any number of lights. The light loop runs twice: once to draw (it fixes the RNG state and tells whether the vertex has any light
// sample at all, which the compaction needs), and -- for the lanes that have one -- again from the saved RNG state, writing the samples
// straight into the entry's slots instead of buffering them per lane (AccumulateDirectLighting loops over all m_LightCount lights,
// CommonLighting.hlsli:877-908; the reference's UI does not bound them).
// Waves per SIMD: 4 for every variant (kWavesShade above). The SIMPLE variant's ring of parked paths is 64 entries so that four blocks fit a
// CU's LDS (-11 % shade time on config 2 against 3 waves with a 128-entry ring). The general single-light variant wants 129 VGPRs and runs
// 14 % faster on config 4 than at 3 waves. History: while the library was built with the SLP vectoriser this variant wanted 138 VGPRs, and forced
// to 128 it computed a WRONG sky radiance (tests/test_parity_gpu.py::test_random_material_subsets... seeds 1 and 10: every miss pixel ~10 % off;
// DESIGN.md section 4) -- the random trait scenes are the guard for this setting.
template <int MAXL, bool SIMPLE, bool PRIMARY = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kWavesShade, kWavesShade))) void wf_shade(WfArgs a, HrptPathTracerConstants cb, uint32_t parity, int bounce, int lastBounce)
{
    static_assert(!PRIMARY || SIMPLE, "no raygen pass: SIMPLE scenes only");
    const uint32_t wavesPerBlock = kBlock / 64, lane = lane_id();
    const uint32_t gw = uniform(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)), totalWaves = gridDim.x * wavesPerBlock;
    const uint32_t in = parity, out = parity ^ 1u;
    const SceneView& s = a.scene;
    // Deferred specular lobe (SIMPLE variant). On diffuse-dominated scenes ~10 % of the paths pick the GGX-VNDF lobe, a long branch that
    // used to run in every iteration with ~6 active lanes. Those lanes now park what the lobe needs in a per-wave LDS ring (SoA, 128
    // entries) and the wave runs the lobe for 64 of them at once (or for whatever is pending when a segment closes: their out-queue slots
    // belong to that segment). Per path the arithmetic is unchanged.
    extern __shared__ __attribute__((aligned(16))) char shadeSmem[];
    constexpr uint32_t kRing = kShadeRing, kRingFields = 23;
    float* const ring = reinterpret_cast<float*>(shadeSmem) + (size_t)uniform(threadIdx.x >> 6) * kRing * kRingFields;
    uint32_t ringHead = 0, pending = 0;
    unsigned int nEntriesOut = 0, nRadiance = 0;      // wave-uniform statistics (HrptStats queue-byte accounting)
    // Sort by shading class (general variants). wf_extend left the class of the hit triangle's material in the hit record (0 constants only,
    // 1 textured, 2 transmission branch; a miss counts as class 3: the sky lookup). When a segment is opened the wave counting-sorts its
    // entries by class into a permutation in LDS (two passes over the 4-byte class words, ballot + prefix rank) and then shades the entries
    // in that order, so that a 64-lane iteration holds one class (two at a class boundary) instead of a mix that runs every branch of
    // shade_surface_a / miss_sky with a fraction of the lanes. Per path nothing changes: every path carries its RNG state and sample index,
    // and survivors / NEE entries still compact into their own segment, only in another order.
    constexpr bool SORT = !SIMPLE;
    const uint32_t segSize = a.segSize;
    // two permutation tables per wave (the open segment A and its successor B, which fills the lanes A's last iteration leaves empty) + keys
    uint16_t* const permBase = reinterpret_cast<uint16_t*>(shadeSmem) + (size_t)uniform(threadIdx.x >> 6) * segSize * 2;
    uint8_t* const keys = reinterpret_cast<uint8_t*>(shadeSmem) + (size_t)wavesPerBlock * segSize * 4 + (size_t)uniform(threadIdx.x >> 6) * segSize;
    uint32_t permSel = 0;                      // which of the two tables belongs to segment A
    bool permuted = false, permutedB = false;
    // counting sort of the `n` entries of the segment at `base` by shading class into `perm`; false when the segment is uniform (queue order kept)
    auto sort_segment = [&](uint32_t base, uint32_t n, uint16_t* perm) -> bool {
        HRT_PHASE(PH_SHADE_SORT);
        uint32_t c0 = 0, c1 = 0, c2 = 0;
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t e = b + lane; uint32_t k = 4u;
            if (e < n) { k = __float_as_uint(a.b.hit[base + e].w) >> 29; k = k > 3u ? 3u : k; keys[e] = (uint8_t)k; }
            c0 += (uint32_t)__popcll(__ballot(k == 0u)); c1 += (uint32_t)__popcll(__ballot(k == 1u)); c2 += (uint32_t)__popcll(__ballot(k == 2u));
        }
        const uint32_t c3 = n - c0 - c1 - c2;
        if (c0 == n || c1 == n || c2 == n || c3 == n) return false;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t b0 = 0, b1 = c0, b2 = c0 + c1, b3 = c0 + c1 + c2;
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t e = b + lane; const uint32_t k = e < n ? keys[e] : 4u;
            const unsigned long long m0 = __ballot(k == 0u), m1 = __ballot(k == 1u), m2 = __ballot(k == 2u), m3 = __ballot(k == 3u);
            if (k == 0u) perm[b0 + prefix_rank(m0)] = (uint16_t)e;
            if (k == 1u) perm[b1 + prefix_rank(m1)] = (uint16_t)e;
            if (k == 2u) perm[b2 + prefix_rank(m2)] = (uint16_t)e;
            if (k == 3u) perm[b3 + prefix_rank(m3)] = (uint16_t)e;
            b0 += (uint32_t)__popcll(m0); b1 += (uint32_t)__popcll(m1); b2 += (uint32_t)__popcll(m2); b3 += (uint32_t)__popcll(m3);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        return true;
    };
    // The wave works through its segments (gw, gw + totalWaves, ...) as one stream of 64-lane iterations: when the open segment A
    // has fewer than 64 entries left, the remaining lanes take the first entries of the next non-empty segment B, so only the wave's
    // last iteration is partially filled (after compaction a 256-slot segment holds ~207 / 168 / 136 paths at bounces 1 / 2 / 3: one
    // partly empty iteration per segment otherwise). Survivors and NEE entries still compact into their OWN segment (front of the
    // out queue / shadow queue of A or B), so segments stay wave-owned and the layout deterministic.
    {
        uint32_t seg = gw, cnt = 0, segBase = 0, next = 0, outCount = 0, shCount = 0;   // segment A: cursor + output counters
        bool haveSeg = false;
        auto open_segment = [&]() {        // first non-empty segment at or after `seg`; empty ones get their (zero) counts written here
            haveSeg = false; cnt = 0; next = 0; outCount = 0; shCount = 0;
            for (; seg < a.numSegments; seg += totalWaves) {
                cnt = uniform(PRIMARY ? path_count(a, in, seg) : a.b.pathCnt[in][seg]);
                if (cnt) { segBase = seg * a.segSize; haveSeg = true; break; }
                if (lane == 0) { a.b.pathCnt[out][seg] = 0; a.b.shadowCnt[seg] = 0; }
            }
            if (SORT && a.sortShade && haveSeg) permuted = sort_segment(segBase, cnt, permBase + permSel * segSize);
        };
        open_segment();
        while (haveSeg) {
            // lanes [0, takeA) continue segment A; if A ends inside this iteration, lanes [takeA, takeA + takeB) start segment B
            const uint32_t takeA = cnt - next < 64u ? cnt - next : 64u;
            const uint32_t segA = seg, baseA = segBase, nextA = next;
            const bool endsA = nextA + takeA >= cnt;
            uint32_t takeB = 0, segB = 0, baseB = 0, cntB = 0;
            if (endsA && takeA < 64u) {
                uint32_t probe = seg + totalWaves;
                for (; probe < a.numSegments; probe += totalWaves) {
                    cntB = uniform(PRIMARY ? path_count(a, in, probe) : a.b.pathCnt[in][probe]);
                    if (cntB) break;
                    if (lane == 0) { a.b.pathCnt[out][probe] = 0; a.b.shadowCnt[probe] = 0; }
                }
                if (probe < a.numSegments) {
                    segB = probe; baseB = probe * a.segSize; takeB = cntB < 64u - takeA ? cntB : 64u - takeA;
                    if (SORT && a.sortShade) permutedB = sort_segment(baseB, cntB, permBase + (permSel ^ 1u) * segSize);
                } else segB = probe;          // no further segment: remembered so that the cursor below ends the loop
            }
            const bool inA = lane < takeA;
            uint32_t outCountB = 0, shCountB = 0;
            bool valid = lane < takeA + takeB, alive = false, wantDefer = false, wantRadiance = false;
            LobeDraw ld; ld.spec = false; ld.specProb = 0.0f; ld.root = 0.0f; ld.sp = 0.0f; ld.cp = 0.0f;
            uint32_t nNee = 0, smp = 0, slotIn = 0;
            PathState ps; f3 neeT = mk3(0.0f, 0.0f, 0.0f);
            constexpr bool STREAMED = MAXL == 0;
            NeeBuf<(STREAMED ? 1 : MAXL)> nee; SurfaceCarry carry;
            if (valid) {
                HRT_PHASE(PH_SHADE_ITER);
                uint32_t slot = inA ? baseA + ((SORT && permuted) ? (uint32_t)permBase[permSel * segSize + nextA + lane] : nextA + lane)
                                    : baseB + ((SORT && permutedB) ? (uint32_t)permBase[(permSel ^ 1u) * segSize + (lane - takeA)] : lane - takeA);
                slotIn = slot;
                float4 o, d, t;
                if constexpr (PRIMARY) {         // origin = camera, direction + seed as wf_extend<PRIMARY> left them, unit throughput, sample = slot
                    const PrimaryArgs& pr = *a.primaryArgs;
                    o = make_float4(pr.cam[0], pr.cam[1], pr.cam[2], 0.0f); d = a.b.rayD[in][slot];
                    t = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(slot));
                } else { o = a.b.rayO[in][slot]; d = a.b.rayD[in][slot]; t = a.b.thr[in][slot]; }
                float4 ha = a.b.hit[slot]; uint32_t tri = __float_as_uint(ha.w);
                const bool noPath = PRIMARY && tri == kNoPathRecord;        // padding pixel (only bounce 0 without a raygen pass has such slots)
                if (tri < kNoPathRecord) tri &= 0x1FFFFFFFu;          // bits 29-31: shading class (wf_extend)
                ps.ray.o = mk3(o.x, o.y, o.z); ps.ray.d = mk3(d.x, d.y, d.z); ps.ray.tmin = o.w; ps.ray.tmax = 1e10f;
                ps.rng = __float_as_uint(d.w);
                ps.throughput = mk3(t.x, t.y, t.z); smp = __float_as_uint(t.w);
                ps.radiance = mk3(0.0f, 0.0f, 0.0f);   // contributions of THIS stage; added to sampleRadiance below
                if (!SIMPLE && a.hasMedium) {
                    float4 m0 = a.b.med0[in][slot], m1 = a.b.med1[in][slot];
                    ps.sigmaA = mk3(m0.x, m0.y, m0.z); ps.interiorIOR = m0.w; ps.sigmaS = mk3(m1.x, m1.y, m1.z); ps.inVolume = m1.w != 0.0f;
                } else { ps.sigmaA = mk3(0.0f, 0.0f, 0.0f); ps.sigmaS = mk3(0.0f, 0.0f, 0.0f); ps.interiorIOR = 1.0f; ps.inVolume = false; }
                bool addRadiance = false; f3 add = mk3(0.0f, 0.0f, 0.0f);
                if (noPath) {
                } else if (tri != 0xFFFFFFFFu) {
                    HRT_PHASE(PH_SHADE_HIT);
                    Hit h; h.valid = true; h.t = ha.x; h.u = ha.y; h.v = ha.z; h.tri = tri; h.prim = 0; h.inst = s.instances ? a.b.hitInst[slot] : 0u; h.opaque = 1;
                    f3 emissiveTerm = mk3(0.0f, 0.0f, 0.0f);
                    SurfaceOutcome oc = shade_surface_a<!SIMPLE, !SIMPLE, SIMPLE>(s, cb, ps, h, carry, [&](uint32_t li, float ux, float uy) {
                        if (STREAMED) ++nNee;
                        else if (nNee < (uint32_t)MAXL) { nee.ux[nNee] = ux; nee.uy[nNee] = uy; nee.light[nNee] = li; ++nNee; }
                    });
                    if (oc == SURFACE_TRANSMITTED) alive = true;
                    else {
                        // ps.radiance now holds throughput*emissive of this hit (PathTracer.hlsl:258)
                        emissiveTerm = ps.radiance;
                        if (emissiveTerm.x != 0.0f || emissiveTerm.y != 0.0f || emissiveTerm.z != 0.0f) { addRadiance = true; add = emissiveTerm; }
                        neeT = ps.throughput;
                        // the sample drawn at the last bounce is never traced (the bounce loop ends, PathTracer.hlsl:90): nothing of it is observable
                        if (!lastBounce) {
                            if (SIMPLE) {
                                if (lobe_begin(ps, carry, bounce, ld)) {
                                    if (ld.spec) wantDefer = true;
                                    else alive = lobe_diffuse(ps, carry.worldPos, carry.N, carry.baseColor, carry.metallic, ld);
                                }
                            } else alive = shade_surface_b(ps, carry, bounce);
                        }
                    }
                } else {
                    miss_sky(s, cb, ps, bounce);   // ps.radiance = throughput * sky
                    addRadiance = true; add = ps.radiance;
#ifdef HRPT_SKY_DEBUG
                    if (smp < 2048u && bounce == 0) {
                        float* g = g_skyDebug + smp * 16u;
                        g[0] = ps.ray.o.x; g[1] = ps.ray.o.y; g[2] = ps.ray.o.z; g[3] = ps.ray.d.x; g[4] = ps.ray.d.y; g[5] = ps.ray.d.z;
                        g[6] = cb.m_SunDirection[0]; g[7] = cb.m_SunDirection[1]; g[8] = cb.m_SunDirection[2]; g[9] = s.lights[0].m_Intensity;
                        g[10] = ps.radiance.x; g[11] = ps.radiance.y; g[12] = ps.radiance.z; g[13] = ps.throughput.x; g[14] = ps.throughput.y; g[15] = ps.throughput.z;
                    }
#endif
                }
                if constexpr (PRIMARY) {
                    // first term of the sample (a padding slot gets a zero nobody reads): 0 + term, stored (no wf_raygen zeroed sampleRadiance; the addition keeps the sign of a -0 term as the
                    // read-modify-write below would)
                    a.b.radiance[smp] = addRadiance ? make_float4(0.0f + add.x, 0.0f + add.y, 0.0f + add.z, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else if (addRadiance) {
                    // radiance_total = radiance_total + term, in path order (x + 0 == x, so zero terms are skipped)
                    float4 r = a.b.radiance[smp];
                    r.x = r.x + add.x; r.y = r.y + add.y; r.z = r.z + add.z;
                    a.b.radiance[smp] = r;
                }
                if (lastBounce) alive = false;
                wantRadiance = addRadiance && !PRIMARY;
            }
            nRadiance += (unsigned int)__popcll(__ballot(wantRadiance));
            // ---- wave-local compaction of survivors into the out queue of their own segment
            const unsigned long long mA = __ballot(alive && inA), mB = __ballot(alive && !inA);
            if (alive) {
                HRT_PHASE(PH_SHADE_WRITE);
                uint32_t o = inA ? baseA + outCount + prefix_rank(mA) : baseB + prefix_rank(mB);
                a.b.rayO[out][o] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tmin);
                a.b.rayD[out][o] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, __uint_as_float(ps.rng));
                a.b.thr[out][o] = make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, __uint_as_float(smp));
                if (!SIMPLE && a.hasMedium) {
                    a.b.med0[out][o] = make_float4(ps.sigmaA.x, ps.sigmaA.y, ps.sigmaA.z, ps.interiorIOR);
                    a.b.med1[out][o] = make_float4(ps.sigmaS.x, ps.sigmaS.y, ps.sigmaS.z, ps.inVolume ? 1.0f : 0.0f);
                }
            }
            // ---- wave-local compaction of NEE work into the shadow queue of their own segment
            const unsigned long long msA = __ballot(nNee > 0 && inA), msB = __ballot(nNee > 0 && !inA);
            if (SIMPLE && MAXL == 1 && a.slimShadow && nNee > 0) {
                // Slim entry (32 instead of 96 bytes): in a scene of this variant nothing is drawn before the light loop and the throughput
                // is not touched before it, so wf_shadow takes V, the RNG state of the two draws, T and the sample index from the path's INPUT
                // record (slotIn), and roughness / metallic / base colour / IOR from the material constants (no texture can modify them).
                uint32_t e = inA ? baseA + shCount + prefix_rank(msA) : baseB + prefix_rank(msB);
                a.b.sh0[e] = make_float4(carry.worldPos.x, carry.worldPos.y, carry.worldPos.z, __uint_as_float(slotIn));
                a.b.sh1[e] = make_float4(carry.N.x, carry.N.y, carry.N.z, __uint_as_float(carry.material));
            } else if (nNee > 0) {
                uint32_t e = inA ? baseA + shCount + prefix_rank(msA) : baseB + prefix_rank(msB);
                a.b.sh0[e] = make_float4(carry.worldPos.x, carry.worldPos.y, carry.worldPos.z, __uint_as_float(smp));
                a.b.sh1[e] = make_float4(carry.N.x, carry.N.y, carry.N.z, carry.roughness);
                a.b.sh2[e] = make_float4(carry.V.x, carry.V.y, carry.V.z, carry.metallic);
                a.b.sh3[e] = make_float4(carry.baseColor.x, carry.baseColor.y, carry.baseColor.z, carry.ior);
                a.b.sh4[e] = make_float4(neeT.x, neeT.y, neeT.z, __uint_as_float(nNee));
                if (STREAMED) {
                    const f3 sunDir = mk3(cb.m_SunDirection[0], cb.m_SunDirection[1], cb.m_SunDirection[2]);
                    uint32_t rng = carry.rngBeforeLights, j = 0;
                    for (uint32_t i = 0; i < cb.m_LightCount; ++i) {
                        HrptGPULight l = load_light(s, i);
                        float ux, uy;
                        if (nee_draw<false>(l, carry.N, carry.worldPos, sunDir, rng, ux, uy)) a.b.shL[(size_t)e * a.maxLights + j++] = make_float4(ux, uy, __uint_as_float(i), 0.0f);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < (STREAMED ? 1 : MAXL); ++j)
                        if ((uint32_t)j < nNee) a.b.shL[(size_t)e * a.maxLights + j] = make_float4(nee.ux[j], nee.uy[j], __uint_as_float(nee.light[j]), 0.0f);
                }
            }
            outCount += (uint32_t)__popcll(mA); shCount += (uint32_t)__popcll(msA);
            outCountB = (uint32_t)__popcll(mB); shCountB = (uint32_t)__popcll(msB);
            nEntriesOut += (unsigned int)(__popcll(msA) + __popcll(msB));
            if (SIMPLE) {
                // ---- park the lanes that picked the specular lobe
                const unsigned long long md = __ballot(wantDefer);
                const uint32_t nNew = (uint32_t)__popcll(md);
                bool parked = false;
                for (int round = 0; round < 2 && !parked; ++round) {
                // (a ring of 64: when this iteration's lanes do not fit behind what is pending, the pending ones run first -- 58+ of 64 lanes, still dense)
                parked = pending + nNew <= kRing;
                if (parked && wantDefer) {
                    float* e = ring + ((ringHead + pending + prefix_rank(md)) & (kRing - 1u));
                    e[0 * kRing] = carry.N.x; e[1 * kRing] = carry.N.y; e[2 * kRing] = carry.N.z;
                    e[3 * kRing] = carry.V.x; e[4 * kRing] = carry.V.y; e[5 * kRing] = carry.V.z;
                    e[6 * kRing] = carry.F0.x; e[7 * kRing] = carry.F0.y; e[8 * kRing] = carry.F0.z;
                    e[9 * kRing] = carry.roughness; e[10 * kRing] = ld.root; e[11 * kRing] = ld.sp; e[12 * kRing] = ld.cp; e[13 * kRing] = ld.specProb;
                    e[14 * kRing] = carry.worldPos.x; e[15 * kRing] = carry.worldPos.y; e[16 * kRing] = carry.worldPos.z;
                    e[17 * kRing] = ps.throughput.x; e[18 * kRing] = ps.throughput.y; e[19 * kRing] = ps.throughput.z;
                    e[20 * kRing] = __uint_as_float(ps.rng); e[21 * kRing] = __uint_as_float(smp); e[22 * kRing] = inA ? 1.0f : 0.0f;
                }
                if (parked) pending += nNew;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                // ---- run the lobe for 64 parked paths at once; when segment A closes, for everything that is parked
                while (pending >= 64u || ((endsA || !parked) && pending)) {
                    const uint32_t n = pending < 64u ? pending : 64u;
                    bool alive2 = false, tagA = true; PathState q; uint32_t smp2 = 0;
                    if (lane < n) {
                        const float* e = ring + ((ringHead + lane) & (kRing - 1u));
                        f3 N = mk3(e[0 * kRing], e[1 * kRing], e[2 * kRing]), V = mk3(e[3 * kRing], e[4 * kRing], e[5 * kRing]), F0 = mk3(e[6 * kRing], e[7 * kRing], e[8 * kRing]);
                        LobeDraw d2; d2.spec = true; d2.root = e[10 * kRing]; d2.sp = e[11 * kRing]; d2.cp = e[12 * kRing]; d2.specProb = e[13 * kRing];
                        f3 wp = mk3(e[14 * kRing], e[15 * kRing], e[16 * kRing]);
                        q.throughput = mk3(e[17 * kRing], e[18 * kRing], e[19 * kRing]); q.rng = __float_as_uint(e[20 * kRing]); smp2 = __float_as_uint(e[21 * kRing]);
                        tagA = e[22 * kRing] != 0.0f;
                        alive2 = lobe_specular(q, wp, N, V, F0, e[9 * kRing], d2);
                    }
                    const unsigned long long m2A = __ballot(alive2 && tagA), m2B = __ballot(alive2 && !tagA);
                    if (alive2) {
                        uint32_t o = tagA ? baseA + outCount + prefix_rank(m2A) : baseB + outCountB + prefix_rank(m2B);
                        a.b.rayO[out][o] = make_float4(q.ray.o.x, q.ray.o.y, q.ray.o.z, q.ray.tmin);
                        a.b.rayD[out][o] = make_float4(q.ray.d.x, q.ray.d.y, q.ray.d.z, __uint_as_float(q.rng));
                        a.b.thr[out][o] = make_float4(q.throughput.x, q.throughput.y, q.throughput.z, __uint_as_float(smp2));
                    }
                    outCount += (uint32_t)__popcll(m2A); outCountB += (uint32_t)__popcll(m2B);
                    ringHead = (ringHead + n) & (kRing - 1u); pending -= n;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                }
            }
            // ---- advance the cursor
            if (!endsA) next = nextA + takeA;
            else {
                if (lane == 0) { a.b.pathCnt[out][segA] = outCount; a.b.shadowCnt[segA] = shCount; }
                if (takeB) {                 // B becomes the open segment, already takeB entries in
                    seg = segB; segBase = baseB; cnt = cntB; next = takeB;
                    outCount = outCountB; shCount = shCountB;
                    haveSeg = true;
                    if (SORT) { permSel ^= 1u; permuted = permutedB; permutedB = false; }
                    if (next >= cnt) {       // B was short enough to end in the same iteration
                        if (lane == 0) { a.b.pathCnt[out][seg] = outCount; a.b.shadowCnt[seg] = shCount; }
                        seg += totalWaves; open_segment();
                    }
                } else if (takeA < 64u) { seg = segB; haveSeg = false; }     // probed to the end: nothing left
                else { seg += totalWaves; open_segment(); }
            }
        }
    }
    block_count_add_uniform<2>(a.counters, { 3, 5 }, { nEntriesOut, nRadiance });
}

// ------------------------------------------------------------------ shadow rays (scenes with non-opaque geometry)
// One ray per valid light sample of every shadow-queue entry, compacted per segment, for the any-hit pass of wf_extend<ANYHIT>.
// nee_direction is evaluated here and again in wf_shadow (same inputs, same bits) rather than stored: 40 B per sample less traffic.
template <bool DIRONLY>
__global__ __launch_bounds__(kBlock) void wf_shadow_rays(WfArgs a, HrptPathTracerConstants cb)
{
    const SceneView& s = a.scene;
    const f3 sunDir = mk3(cb.m_SunDirection[0], cb.m_SunDirection[1], cb.m_SunDirection[2]);
    const uint32_t wavesPerBlock = kBlock / 64, lane = lane_id();
    const uint32_t gw = uniform(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)), totalWaves = gridDim.x * wavesPerBlock;
    for (uint32_t seg = gw; seg < a.numSegments; seg += totalWaves) {
        const uint32_t cnt = uniform(a.b.shadowCnt[seg]), segBase = seg * a.segSize, items = cnt * a.maxLights;
        uint32_t outCount = 0;
        for (uint32_t i0 = 0; i0 < items; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool valid = false; Ray ray; uint32_t id = 0;
            if (i < items) {
                HRT_PHASE(PH_SHADOW_RAYS_ITEM);
                const uint32_t el = i / a.maxLights, j = i - el * a.maxLights, e = segBase + el;
                id = e * a.maxLights + j;
                float4 h4 = a.b.sh4[e];
                if (j < __float_as_uint(h4.w)) {
                    float4 h0 = a.b.sh0[e], h1 = a.b.sh1[e], ls = a.b.shL[(size_t)e * a.maxLights + j];
                    HrptGPULight l = load_light(s, __float_as_uint(ls.z));
                    f3 L; float maxDist;
                    if (nee_direction<DIRONLY>(l, mk3(h1.x, h1.y, h1.z), mk3(h0.x, h0.y, h0.z), sunDir, cb.m_CosSunAngularRadius, ls.x, ls.y, L, maxDist)) {
                        ray = shadow_ray(mk3(h0.x, h0.y, h0.z), L, maxDist);
                        valid = true;
                    }
                }
                if (!valid) a.b.shVis[id] = kVisNoRay;
            }
            const unsigned long long m = __ballot(valid);
            if (valid) {
                const uint32_t o = segBase * a.maxLights + outCount + prefix_rank(m);
                a.b.sqO[o] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.tmin);
                a.b.sqD[o] = make_float4(ray.d.x, ray.d.y, ray.d.z, ray.tmax);
                a.b.sqId[o] = id;
            }
            outCount += (uint32_t)__popcll(m);
        }
        if (lane == 0) a.b.sqCnt[seg] = outCount;
    }
}

// ------------------------------------------------------------------ shadow (NEE visibility + accumulation)
// MODE kShadowOpaque: no ForceNonOpaque instance in the scene: plain any-hit query per light sample.
// MODE kShadowBuffered: non-opaque geometry, the kernel traverses itself: per-lane candidate buffer in LDS (after the stack) and the buffered query.
// MODE kShadowResolve: non-opaque geometry, visibility traversal already done by wf_shadow_rays + wf_extend<ANYHIT>: this kernel only walks
//   the recorded candidate lists (its stack serves the rare re-trace behind an overflowing list) and evaluates the contributions.
// MODE kShadowSlim: kShadowOpaque with directional lights only and the 32-byte entries of wf_shade<1, SIMPLE> (see there).
enum : int { kShadowOpaque = 0, kShadowBuffered = 1, kShadowResolve = 2, kShadowSlim = 3 };
// Waves per SIMD (built without the SLP vectoriser, csrc/Makefile): the buffered variant is held at 3 -- the gradient-sampled alpha test of
// mip-mapped MASK textures, a rare path, would otherwise cost every scene with alpha-tested geometry a wave (145-152 VGPRs; at 4 waves config 4's
// shadow stage is 6 % SLOWER: it traverses, and spills hurt its loops); the resolve variant runs at 4 (146-150 VGPRs wanted, 128 given: it waits
// on its queue reads for 71 % of its cycles and only walks candidate lists: glass config shadow stage -6 %); the two-level variants at 3
// (scenes with non-opaque instances -16 % shadow time). Forced budgets are re-checked with the random trait scenes (scripts/parity_campaign.sh:
// HRPT_WF_SHADOW_PATH=1 / 2 and the two-level scenes), because one has broken wf_shade's general variant before.
template <bool LDS_BVH, int DEPTH, int W, bool DIRONLY, int MODE, int TL = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MODE == 2 && TL == 0 ? 4 : ((MODE == 1 || TL != 0) ? 3 : 1)))) void wf_shadow(WfArgs a, HrptPathTracerConstants cb, int bounce)
{
    static_assert(!TL || (!LDS_BVH && W == 4 && (MODE == kShadowOpaque || MODE == kShadowSlim)), "two-level structure: opaque any-hit query over the global tree");
    static_assert(W != kQuantisedTree || (!LDS_BVH && !TL), "quantised nodes: flat tree in global memory");
    constexpr bool NONOPAQUE = MODE == kShadowBuffered || MODE == kShadowResolve;
    constexpr bool SLIM = MODE == kShadowSlim;
    constexpr int kLdsMax = MODE == kShadowResolve ? kExtendLdsStack : kShadowLdsStack;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    LdsStack<DEPTH, kLdsMax> stack; LdsBvh<W> lbvh;
    constexpr size_t candBytes = MODE == kShadowBuffered ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0;
    setup_lds<LDS_BVH, DEPTH, W>(smem, a.scene, stack, lbvh, candBytes);
    if (DEPTH > kLdsMax) { stack.spill = a.spill[1] + blockIdx.x * kBlock + threadIdx.x; stack.spillStride = gridDim.x * kBlock; }
    LdsCandidates cand; cand.base = reinterpret_cast<int32_t*>(smem + (size_t)LdsStack<DEPTH, kLdsMax>::kRows * kBlock * 4) + threadIdx.x;
    LdsCandidates3 cand3; cand3.base = cand.base;        // TL == 2: (t, mesh triangle, instance) columns in the same place (the launch adds their bytes)
    typename GlobalBvhOf<(TL ? kTwoLevelTree : W)>::type gbvh = GlobalBvhOf<(TL ? kTwoLevelTree : W)>::make(a.scene);
    const SceneView& s = a.scene;
    const f3 sunDir = mk3(cb.m_SunDirection[0], cb.m_SunDirection[1], cb.m_SunDirection[2]);
    const float sunIntensity = s.lights[0].m_Intensity;      // g_Lights[0], PathTracer.hlsl:137 (reference quirk kept)

    const uint32_t wavesPerBlock = kBlock / 64, lane = lane_id();
    const uint32_t gw = uniform(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)), totalWaves = gridDim.x * wavesPerBlock;
    unsigned int nRays = 0, nSamples = 0, nRadiance = 0, nSkipped16 = 0;
    // one shadow-queue entry: every light sample of one path vertex
    auto process = [&](uint32_t e) {
                HRT_PHASE(PH_SHADOW_ENTRY);
                if (SLIM) {
                    const float4 h0 = a.b.sh0[e], h1 = a.b.sh1[e];
                    const uint32_t slot = __float_as_uint(h0.w);
                    const float4 rd = a.b.rayD[a.shadowParity][slot];
                    const f3 origin = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z);
                    uint32_t rng = __float_as_uint(rd.w);
                    const float ux = hrt_rng_next(&rng), uy = hrt_rng_next(&rng);      // the two draws of nee_draw (CommonLighting.hlsli:730)
                    ++nSamples;
                    HrptGPULight l = load_light(s, 0u);
                    f3 L; float maxDist;
                    if (!nee_direction<true>(l, N, origin, sunDir, cb.m_CosSunAngularRadius, ux, uy, L, maxDist)) return;
                    float shadow;
                    if (LDS_BVH) shadow = shadow_query<true>(s, lbvh, origin, L, maxDist, stack);
                    else if constexpr (TL == 2) shadow = shadow_query_two_level_buffered<kTwoLevelCandidates>(s, gbvh, shadow_ray(origin, L, maxDist), stack, cand3);
                    else shadow = shadow_query<true>(s, gbvh, origin, L, maxDist, stack);
                    ++nRays;
                    if (shadow != 0.0f) {
                        const float4 th = a.primary ? make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(slot)) : a.b.thr[a.shadowParity][slot];
                        if (a.primary) ++nSkipped16;
                        const HrptMaterialConstants& mat = s.materials[__float_as_uint(h1.w)];
                        // GetPBRAttributes without textures (RaytracingCommon.hlsli:252-296): constants, roughness clamped at 0.04
                        f3 dif, spec;
                        nee_contribution<true>(s, l, nee_lighting(N, mk3(-rd.x, -rd.y, -rd.z), mk3(mat.m_BaseColor), hrt_max(mat.m_RoughnessMetallic[0], 0.04f),
                                                                  mat.m_RoughnessMetallic[1], mat.m_IOR), origin, sunDir, sunIntensity, L, dif, spec);
                        f3 totalDiffuse = mk3(0.0f, 0.0f, 0.0f) + dif * shadow, totalSpecular = mk3(0.0f, 0.0f, 0.0f);
                        if (bounce == 0) totalSpecular = totalSpecular + spec * shadow;
                        const f3 dsum = totalDiffuse + (bounce == 0 ? totalSpecular : mk3(0.0f, 0.0f, 0.0f));
                        if (dsum.x != 0.0f || dsum.y != 0.0f || dsum.z != 0.0f) {
                            const f3 term = mk3(th.x, th.y, th.z) * dsum;
                            const uint32_t smp = __float_as_uint(th.w);
                            float4 r = a.b.radiance[smp];
                            r.x = r.x + term.x; r.y = r.y + term.y; r.z = r.z + term.z;
                            a.b.radiance[smp] = r;
                            ++nRadiance;
                        }
                    }
                    return;
                }
                float4 h0 = a.b.sh0[e], h1 = a.b.sh1[e], h4 = a.b.sh4[e];
                f3 origin = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z), T = mk3(h4.x, h4.y, h4.z);
                uint32_t smp = __float_as_uint(h0.w), n = __float_as_uint(h4.w);
                nSamples += n;
                f3 totalDiffuse = mk3(0.0f, 0.0f, 0.0f), totalSpecular = mk3(0.0f, 0.0f, 0.0f);
                // Variants that traverse themselves request the first sample's record together with the entry (one dependent round trip less:
                // single-light scenes -2 % shadow time). Not the resolve variant: it is held at 128 VGPRs and loses 3 % to the extra live
                // registers; requesting sample j + 1 while j is worked on costs every variant more than it hides.
                float4 ls0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (MODE != kShadowResolve && n) ls0 = a.b.shL[(size_t)e * a.maxLights];
                for (uint32_t j = 0; j < n; ++j) {
                    HRT_PHASE(PH_SHADOW_SAMPLE);
                    uint32_t vis = kVisCandidates;
                    if (MODE == kShadowResolve) {
                        // outcome of the opaque any-hit pass (wf_shadow_rays + wf_extend<ANYHIT>): most rays are settled there
                        vis = a.b.shVis[e * a.maxLights + j];
                        if (vis == kVisNoRay) continue;                                  // nee_direction said no
                        if (vis == kVisBlocked) { ++nRays; continue; }                   // shadow factor 0: contributes +0
                    }
                    float4 ls = ls0;
                    if (MODE == kShadowResolve || j) ls = a.b.shL[(size_t)e * a.maxLights + j];
                    HrptGPULight l = load_light(s, __float_as_uint(ls.z));
                    f3 L; float maxDist;
                    if (!nee_direction<DIRONLY>(l, N, origin, sunDir, cb.m_CosSunAngularRadius, ls.x, ls.y, L, maxDist)) continue;
                    float shadow;
                    HRT_PHASE(PH_SHADOW_QUERY);
                    if (MODE == kShadowResolve) {
                        if (vis == kVisClear) shadow = 1.0f;                             // no triangle at all in (tmin, tmax)
                        else {                                                           // the non-opaque triangles it crossed, nearest first
                            Ray sr = shadow_ray(origin, L, maxDist);
                            GlobalCandidates gc; gc.base = a.b.sqCand + (size_t)(e * a.maxLights + j) * kShadowCandidates;
                            const int cnt = (int)((vis >> 8) & 0xFFu); const bool ovf = (vis >> 16) & 1u;
                            if (LDS_BVH) shadow = shadow_resolve_candidates(s, lbvh, sr, make_shear(sr.d), cnt, ovf, gc, stack);
                            else shadow = shadow_resolve_candidates(s, gbvh, sr, make_shear(sr.d), cnt, ovf, gc, stack);
                        }
                    } else if (MODE == kShadowBuffered) {
                        // (no descent threshold here: without lane refill it only costs: config 2 shadow +3 %, config 4 +2.5 %)
                        if (LDS_BVH) shadow = shadow_query_buffered<kShadowCandidates>(s, lbvh, origin, L, maxDist, stack, cand);
                        else shadow = shadow_query_buffered<kShadowCandidates>(s, gbvh, origin, L, maxDist, stack, cand);
                    } else {
                        if (LDS_BVH) shadow = shadow_query<true>(s, lbvh, origin, L, maxDist, stack);       // kShadowOpaque: no ForceNonOpaque instance in the scene
                        else if constexpr (TL == 2) shadow = shadow_query_two_level_buffered<kTwoLevelCandidates>(s, gbvh, shadow_ray(origin, L, maxDist), stack, cand3);
                    else shadow = shadow_query<true>(s, gbvh, origin, L, maxDist, stack);
                    }
                    ++nRays;
                    if (shadow != 0.0f) {   // an occluded sample contributes +0: its BRDF x radiance evaluation is skipped
                        HRT_PHASE(PH_SHADOW_CONTRIB);
                        float4 h2 = a.b.sh2[e], h3 = a.b.sh3[e];
                        f3 dif, spec;
                        nee_contribution<DIRONLY>(s, l, nee_lighting(N, mk3(h2.x, h2.y, h2.z), mk3(h3.x, h3.y, h3.z), h1.w, h2.w, h3.w), origin, sunDir,
                                                  sunIntensity, L, dif, spec);
                        totalDiffuse = totalDiffuse + dif * shadow;
                        if (bounce == 0) totalSpecular = totalSpecular + spec * shadow;
                    }
                }
                f3 dsum = totalDiffuse + (bounce == 0 ? totalSpecular : mk3(0.0f, 0.0f, 0.0f));
                if (dsum.x != 0.0f || dsum.y != 0.0f || dsum.z != 0.0f) {
                    f3 term = T * dsum;                                         // PathTracer.hlsl:261
                    float4 r = a.b.radiance[smp];
                    r.x = r.x + term.x; r.y = r.y + term.y; r.z = r.z + term.z;
                    a.b.radiance[smp] = r;
                    ++nRadiance;
                }
    };
    if (NONOPAQUE) {
        // Buffered variant (long, uneven entries): the wave hands out the entries of its segments (gw, gw + totalWaves, ...) as one
        // stream, an iteration whose segment runs out continues with the next one, so only the wave's last iteration is partly filled
        // (-4 % on configs 4 / 5). The short opaque any-hit variant below is faster with the plain per-segment loop (config 2).
        uint32_t seg = gw, cnt = 0, segBase = 0, next = 0; bool haveSeg = false;
        auto open_segment = [&]() {
            haveSeg = false; cnt = 0; next = 0;
            for (; seg < a.numSegments; seg += totalWaves) {
                cnt = uniform(a.b.shadowCnt[seg]);
                if (cnt) { segBase = seg * a.segSize; haveSeg = true; break; }
            }
        };
        open_segment();
        while (haveSeg) {
            uint32_t e = 0xFFFFFFFFu, filled = 0;
            while (filled < 64u && haveSeg) {
                uint32_t take = cnt - next < 64u - filled ? cnt - next : 64u - filled;
                if (lane >= filled && lane < filled + take) e = segBase + next + (lane - filled);
                next += take; filled += take;
                if (next >= cnt) { seg += totalWaves; open_segment(); }
            }
            if (e != 0xFFFFFFFFu) process(e);
        }
    } else {
        for (uint32_t seg = gw; seg < a.numSegments; seg += totalWaves) {
            const uint32_t cnt = uniform(a.b.shadowCnt[seg]), segBase = seg * a.segSize;
            for (uint32_t base = 0; base < cnt; base += 64) {
                uint32_t i = base + lane;
                if (i < cnt) process(segBase + i);
            }
        }
    }
    block_count_add(&a.counters->closestRays, 1, nRays);
    __syncthreads();
    block_count_add(&a.counters->closestRays, 4, nSamples);
    __syncthreads();
    block_count_add(&a.counters->closestRays, 6, nRadiance);
    if (SLIM && a.primary) { __syncthreads(); block_count_add(&a.counters->closestRays, 7, nSkipped16); }
}

// ------------------------------------------------------------------ resolve: fold the indices in order (:332-339)
__global__ __launch_bounds__(kBlock) void wf_resolve(WfArgs a, float4* __restrict__ accumulation, float4* __restrict__ output, uint32_t firstIndex)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (; p < a.pixelsPadded; p += stride) {
        uint32_t tile = p >> 6, within = p & 63u;
        uint32_t tcol, trow; tile_position(a, tile, tcol, trow);
        uint32_t px = a.rect.column_x(tcol) + (within & 7u), py = a.rect.y0 + trow * 8u + (within >> 3);
        if (px >= a.rect.x1 || py >= a.rect.y1) continue;
        size_t idx = (size_t)py * a.imageWidth + px;
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < a.spp; ++k) {
            float4 r = a.b.radiance[(size_t)k * a.pixelsPadded + p];
            float4 cur = make_float4(r.x, r.y, r.z, 1.0f);
            if (firstIndex + k > 0) {
                float4 prev = (k == 0) ? accumulation[idx] : acc;
                cur.x += prev.x; cur.y += prev.y; cur.z += prev.z; cur.w += prev.w;
            }
            acc = cur;
        }
        accumulation[idx] = acc;
        output[idx] = make_float4(acc.x / acc.w, acc.y / acc.w, acc.z / acc.w, 1.0f);
    }
}

// ------------------------------------------------------------------ host side
// The persistent kernels divide their segments evenly among the waves of the grid (seg = wave, wave + waves, ...), so a grid that is not a
// whole number of ROUNDS of what a CU holds of that kernel finishes late: with six resident blocks per CU a grid of 16 per CU takes three
// rounds of 1/16 of the work each, 12 takes two of 1/12 (measured: LDS-tree wf_extend +8 % at 16, global-tree wf_extend +20 % at 7 against 6).
// launch_rounds trims the requested grid to whole rounds of the kernel's own occupancy (hipOccupancyMaxActiveBlocksPerMultiprocessor: its
// registers, its LDS bytes), asked once per kernel and LDS size. Used for the closest-hit traversal kernels (wf_extend, wf_trace_rays), where
// the effect is large and consistent; the other kernels launch the grid asked for: trimmed, the any-hit pass of the glass config lost 12 %
// (one round of its five resident blocks instead of eight blocks per CU) and the buffered shadow kernel 3 %, the rest did not move.
static thread_local uint32_t tlCus = 0;        // compute units of the device the calling thread launches on (wavefront_render / wavefront_trace_rays set it)
static int resident_blocks_per_cu(const void* kernel, size_t ldsBytes)
{
    static std::mutex mu; static std::unordered_map<uint64_t, int> cache;
    const uint64_t key = (uint64_t)(uintptr_t)kernel * 0x9E3779B97F4A7C15ull ^ (uint64_t)ldsBytes;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, (int)kBlock, ldsBytes) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    cache.emplace(key, n);
    return n;
}
template <class K, class... Args> static void launch_rounds(K kernel, dim3 g, size_t ldsBytes, hipStream_t st, Args... args)
{
    const int resident = tlCus ? resident_blocks_per_cu(reinterpret_cast<const void*>(kernel), ldsBytes) : 0;
    const uint32_t perRound = tlCus * (uint32_t)(resident > 0 ? resident : 0);
    if (perRound && g.x > perRound) g.x = (g.x / perRound) * perRound;
    hipLaunchKernelGGL(kernel, g, dim3(kBlock), ldsBytes, st, args...);
}
struct Variant { bool lds; int depth; int width; size_t ldsBytes; bool twoLevel = false; bool twoLevelCandidates = false; bool quantised = false; };

template <bool L, int D, int W> void launch_extend_t(dim3 g, size_t sh, hipStream_t st, const WfArgs& a, uint32_t parity, bool anyHit)
{
    if constexpr (W >= 4) if (a.primary && !anyHit) {
        if (a.allOpaque) launch_rounds((wf_extend<L, D, W, false, 0, true, true>), g, sh, st, a, parity);
        else launch_rounds((wf_extend<L, D, W, false, 0, true>), g, sh, st, a, parity);
        return;
    }
    if (anyHit) hipLaunchKernelGGL((wf_extend<L, D, W, true>), g, dim3(kBlock), sh + (size_t)kShadowCandidates * 2 * kBlock * 4, st, a, parity);
    else if constexpr (W >= 4) { if (a.allOpaque) launch_rounds((wf_extend<L, D, W, false, 0, false, true>), g, sh, st, a, parity); else launch_rounds((wf_extend<L, D, W, false>), g, sh, st, a, parity); }
    else launch_rounds((wf_extend<L, D, W, false>), g, sh, st, a, parity);
}
// nonOpaque: 0 = opaque scene, 1 = buffered query inside wf_shadow, 2 = resolve only (after the any-hit pass)
template <bool L, int D, int W> void launch_shadow_t(dim3 g, size_t sh, hipStream_t st, const WfArgs& a, const HrptPathTracerConstants& cb, int bounce, bool dirOnly, int nonOpaque)
{
    if (nonOpaque == kShadowResolve) hipLaunchKernelGGL((wf_shadow<L, D, W, false, kShadowResolve>), g, dim3(kBlock), sh, st, a, cb, bounce);
    else if (nonOpaque == kShadowBuffered) {   // general variant (all light types) + candidate buffer
        // (not trimmed to whole rounds: measured, the buffered variant -- three resident blocks per CU, long uneven entries -- is 3 % slower
        // with 15 or 6 blocks per CU than with the 16 or 8 asked for)
        hipLaunchKernelGGL((wf_shadow<L, D, W, false, kShadowBuffered>), g, dim3(kBlock), sh + (size_t)kShadowCandidates * 2 * kBlock * 4, st, a, cb, bounce);
    } else if (nonOpaque == kShadowSlim) hipLaunchKernelGGL((wf_shadow<L, D, W, true, kShadowSlim>), g, dim3(kBlock), sh, st, a, cb, bounce);
    else if (dirOnly) hipLaunchKernelGGL((wf_shadow<L, D, W, true, kShadowOpaque>), g, dim3(kBlock), sh, st, a, cb, bounce);
    else hipLaunchKernelGGL((wf_shadow<L, D, W, false, kShadowOpaque>), g, dim3(kBlock), sh, st, a, cb, bounce);
}

// stack need classes: BVH2 8/16/32/64 (maxDepth + 2), BVH4 16/32/64 (3 * maxDepth4 + 2); class 64 = "deeper than the LDS part": the kernel keeps
// kExtendLdsStack / kShadowLdsStack entries in LDS and the rest in the overflow columns (LdsStack)
template <bool L> void launch_extend_l(Variant v, dim3 g, size_t sh, hipStream_t st, const WfArgs& a, uint32_t parity, bool anyHit)
{
    if (v.width == 2) {
        if (v.depth <= 8) launch_extend_t<L, 8, 2>(g, sh, st, a, parity, anyHit); else if (v.depth <= 16) launch_extend_t<L, 16, 2>(g, sh, st, a, parity, anyHit); else if (v.depth <= 32) launch_extend_t<L, 32, 2>(g, sh, st, a, parity, anyHit); else launch_extend_t<L, 64, 2>(g, sh, st, a, parity, anyHit);
    } else {
        if constexpr (!L) if (v.quantised) {
            if (v.depth <= 16) launch_extend_t<L, 16, kQuantisedTree>(g, sh, st, a, parity, anyHit); else if (v.depth <= 32) launch_extend_t<L, 32, kQuantisedTree>(g, sh, st, a, parity, anyHit); else launch_extend_t<L, 64, kQuantisedTree>(g, sh, st, a, parity, anyHit);
            return;
        }
        if (v.depth <= 16) launch_extend_t<L, 16, 4>(g, sh, st, a, parity, anyHit); else if (v.depth <= 32) launch_extend_t<L, 32, 4>(g, sh, st, a, parity, anyHit); else launch_extend_t<L, 64, 4>(g, sh, st, a, parity, anyHit);
    }
}
template <bool L> void launch_shadow_l(Variant v, dim3 g, size_t sh, hipStream_t st, const WfArgs& a, const HrptPathTracerConstants& cb, int bounce, bool dirOnly, int nonOpaque)
{
    if (v.width == 2) {
        if (v.depth <= 8) launch_shadow_t<L, 8, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else if (v.depth <= 16) launch_shadow_t<L, 16, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else if (v.depth <= 32) launch_shadow_t<L, 32, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else launch_shadow_t<L, 64, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque);
    } else {
        if constexpr (!L) if (v.quantised) {
            if (v.depth <= 16) launch_shadow_t<L, 16, kQuantisedTree>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else if (v.depth <= 32) launch_shadow_t<L, 32, kQuantisedTree>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else launch_shadow_t<L, 64, kQuantisedTree>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque);
            return;
        }
        if (v.depth <= 16) launch_shadow_t<L, 16, 4>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else if (v.depth <= 32) launch_shadow_t<L, 32, 4>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else launch_shadow_t<L, 64, 4>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque);
    }
}
template <int D, int TL> void launch_shadow_two_level(dim3 g, size_t sh, hipStream_t st, const WfArgs& a, const HrptPathTracerConstants& cb, int bounce, bool dirOnly, int mode)
{
    if (TL == 2) sh += (size_t)kTwoLevelCandidates * 3 * kBlock * 4;      // candidate columns of the buffered two-level shadow query
    if (mode == kShadowSlim) hipLaunchKernelGGL((wf_shadow<false, D, 4, true, kShadowSlim, TL>), g, dim3(kBlock), sh, st, a, cb, bounce);
    else if (dirOnly) hipLaunchKernelGGL((wf_shadow<false, D, 4, true, kShadowOpaque, TL>), g, dim3(kBlock), sh, st, a, cb, bounce);
    else hipLaunchKernelGGL((wf_shadow<false, D, 4, false, kShadowOpaque, TL>), g, dim3(kBlock), sh, st, a, cb, bounce);
}
template <int D, int TL> void launch_extend_two_level(dim3 g, size_t sh, hipStream_t st, const WfArgs& a, uint32_t parity)
{
    if (a.primary) launch_rounds((wf_extend<false, D, 4, false, TL, true>), g, sh, st, a, parity);
    else launch_rounds((wf_extend<false, D, 4, false, TL, false>), g, sh, st, a, parity);
}
void launch_extend(Variant v, dim3 g, size_t sh, hipStream_t st, const WfArgs& a, uint32_t parity, bool anyHit = false)
{
    if (v.twoLevel) {       // closest hits only (wavefront_render keeps the any-hit pass off for two-level scenes)
        if (v.twoLevelCandidates) {
            if (v.depth <= 16) launch_extend_two_level<16, 2>(g, sh, st, a, parity); else if (v.depth <= 32) launch_extend_two_level<32, 2>(g, sh, st, a, parity); else launch_extend_two_level<64, 2>(g, sh, st, a, parity);
        } else {
            if (v.depth <= 16) launch_extend_two_level<16, 1>(g, sh, st, a, parity); else if (v.depth <= 32) launch_extend_two_level<32, 1>(g, sh, st, a, parity); else launch_extend_two_level<64, 1>(g, sh, st, a, parity);
        }
        return;
    }
    if (v.lds) launch_extend_l<true>(v, g, sh, st, a, parity, anyHit); else launch_extend_l<false>(v, g, sh, st, a, parity, anyHit);
}
void launch_shadow(Variant v, dim3 g, size_t sh, hipStream_t st, const WfArgs& a, const HrptPathTracerConstants& cb, int bounce, bool dirOnly, int nonOpaque)
{
    if (v.twoLevel) {
        if (v.twoLevelCandidates) {
            if (v.depth <= 16) launch_shadow_two_level<16, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else if (v.depth <= 32) launch_shadow_two_level<32, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else launch_shadow_two_level<64, 2>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque);
        } else {
            if (v.depth <= 16) launch_shadow_two_level<16, 1>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else if (v.depth <= 32) launch_shadow_two_level<32, 1>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else launch_shadow_two_level<64, 1>(g, sh, st, a, cb, bounce, dirOnly, nonOpaque);
        }
        return;
    }
    if (v.lds) launch_shadow_l<true>(v, g, sh, st, a, cb, bounce, dirOnly, nonOpaque); else launch_shadow_l<false>(v, g, sh, st, a, cb, bounce, dirOnly, nonOpaque);
}

} // namespace

namespace {
template <bool L, int D, bool SH> void launch_trace_rays_t(dim3 g, size_t lds, hipStream_t st, const WfTraceArgs& a)
{
    if constexpr (!L) if (a.quantised) { launch_rounds((wf_trace_rays<L, D, kQuantisedTree, SH>), g, lds + (SH ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0), st, a); return; }
    launch_rounds((wf_trace_rays<L, D, 4, SH>), g, lds + (SH ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0), st, a);
}
template <bool L, bool SH> void launch_trace_rays_d(int depth, dim3 g, size_t lds, hipStream_t st, const WfTraceArgs& a)
{
    if (depth <= 16) launch_trace_rays_t<L, 16, SH>(g, lds, st, a); else if (depth <= 32) launch_trace_rays_t<L, 32, SH>(g, lds, st, a); else launch_trace_rays_t<L, 64, SH>(g, lds, st, a);
}
}



bool wavefront_trace_rays_supported(const SceneTraits& traits) { return (traits.twoLevelStackNeed ? traits.twoLevelStackNeed : 3 * traits.bvh4MaxDepth + 2) <= kMaxStackNeed; }

hipError_t wavefront_trace_rays(WavefrontState& st, const SceneView& scene, const SceneTraits& traits, const HrptRay* rays, HrptRayHit* hits, uint64_t count,
                                bool shadow, hipStream_t stream, std::string& error)
{
    if (count == 0) return hipSuccess;
    hipError_t e; int dev = 0; hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) { error = "hipGetDeviceProperties"; return e; }
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    tlCus = cus;
    const bool twoLevel = traits.twoLevelStackNeed != 0;
    const uint32_t need = twoLevel ? traits.twoLevelStackNeed : 3 * traits.bvh4MaxDepth + 2;
    const int depth = need <= 16 ? 16 : (need <= 32 ? 32 : 64);
    const size_t candBytes = shadow ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0;
    const size_t stackBytes = (size_t)(depth > kExtendLdsStack ? kExtendLdsStack : depth) * kBlock * 4;
    const size_t bvhBytes = (size_t)scene.node4Count * kLdsNode4Stride + (size_t)scene.triCount * 48;
    const bool lds = !twoLevel && bvhBytes > 0 && stackBytes + candBytes + bvhBytes <= kLdsBudget && !st.forceGlobalBvh;
    const uint32_t blocksPerCu = st.blocksPerCu ? st.blocksPerCu : 16;
    const uint64_t chunks = (count + 255) / 256, blocksNeeded = (chunks + 3) / 4;
    uint32_t grid = cus * blocksPerCu; if (grid > blocksNeeded) grid = (uint32_t)blocksNeeded;
    WfTraceArgs a{};
    a.scene = scene; a.rays = rays; a.hits = hits; a.count = count;
    a.refillMin = st.refillMin ? st.refillMin : kRefillMinDefault;
    a.nodeLoopMin = st.nodeLoopMin != ~0u ? st.nodeLoopMin : (lds ? 16u : 24u);
    a.quantised = !twoLevel && !lds && traits.quantisedNodes && scene.nodesQ != nullptr;
    if (depth > kExtendLdsStack) {
        // own overflow columns (a render may be in flight on the context's buffers only in stream order, but sizes differ)
        const uint32_t entries = need > (uint32_t)kExtendLdsStack ? need - kExtendLdsStack : 1u;
        const size_t bytes = (size_t)cus * blocksPerCu * kBlock * entries * 4;
        if (bytes > st.traceSpillBytes) {
            if (st.traceSpill) { (void)hipStreamSynchronize(stream); (void)hipFree(st.traceSpill); st.traceSpill = nullptr; st.traceSpillBytes = 0; }
            if ((e = hipMalloc(&st.traceSpill, bytes)) != hipSuccess) { error = "hipMalloc(traversal stack overflow)"; return e; }
            st.traceSpillBytes = bytes;
        }
        a.spill = static_cast<int32_t*>(st.traceSpill);
    }
    const size_t ldsBytes = stackBytes + (lds ? bvhBytes : 0);
    if (twoLevel) {
        const dim3 g(grid); const size_t sl = ldsBytes;     // (no candidate columns: visibility queries over non-opaque instances re-trace behind every candidate)
        if (shadow) { if (depth <= 16) launch_rounds((wf_trace_rays<false, 16, 4, true, true>), g, sl, stream, a); else if (depth <= 32) launch_rounds((wf_trace_rays<false, 32, 4, true, true>), g, sl, stream, a); else launch_rounds((wf_trace_rays<false, 64, 4, true, true>), g, sl, stream, a); }
        else { if (depth <= 16) launch_rounds((wf_trace_rays<false, 16, 4, false, true>), g, sl, stream, a); else if (depth <= 32) launch_rounds((wf_trace_rays<false, 32, 4, false, true>), g, sl, stream, a); else launch_rounds((wf_trace_rays<false, 64, 4, false, true>), g, sl, stream, a); }
    } else
    if (lds) { if (shadow) launch_trace_rays_d<true, true>(depth, dim3(grid), ldsBytes, stream, a); else launch_trace_rays_d<true, false>(depth, dim3(grid), ldsBytes, stream, a); }
    else { if (shadow) launch_trace_rays_d<false, true>(depth, dim3(grid), ldsBytes, stream, a); else launch_trace_rays_d<false, false>(depth, dim3(grid), ldsBytes, stream, a); }
    if ((e = hipGetLastError()) != hipSuccess) { error = "kernel launch"; return e; }
    return hipSuccess;
}

bool wavefront_supports(const SceneView& scene, const HrptPathTracerConstants& cb)
{
    (void)scene;
    return cb.m_MaxBounces >= 1;       // any number of lights: up to kMaxLights buffered per lane, beyond that streamed (wf_shade<0>)
}

void wavefront_release(WavefrontState& st)
{
    if (st.pool) (void)hipFree(st.pool);
    st.pool = nullptr; st.poolBytes = 0;
    if (st.spill) (void)hipFree(st.spill);
    st.spill = nullptr; st.spillBytes = 0;
    if (st.traceSpill) (void)hipFree(st.traceSpill);
    st.traceSpill = nullptr; st.traceSpillBytes = 0;
    for (hipEvent_t e : st.events) (void)hipEventDestroy(e);
    st.events.clear(); st.eventsUsed = 0;
    for (hipEvent_t e : st.forkEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : st.joinEvents) (void)hipEventDestroy(e);
    st.forkEvents.clear(); st.joinEvents.clear();
    if (st.auxStream) (void)hipStreamDestroy(st.auxStream);
    st.auxStream = nullptr;
}

void wavefront_collect_timing(WavefrontState& st)
{
    for (uint32_t i = 0; i + 1 < st.eventsUsed; i += 2) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, st.events[i], st.events[i + 1]) == hipSuccess) { st.kernelMs[st.kind[i / 2]] += t; st.kernelLaunches[st.kind[i / 2]]++; }
    }
    st.eventsUsed = 0;
}

#ifdef HRPT_SKY_DEBUG
extern "C" int hrpt_sky_debug_read(float* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_skyDebug), sizeof(float) * 2048 * 16) == hipSuccess ? 0 : -3; }
#endif
#ifdef HRPT_PHASE_PROFILE
// reads (and zeroes) the phase counters of this library build
extern "C" int hrpt_phase_profile_read(unsigned long long* out128)
{
    if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_phaseCounters), 128 * sizeof(unsigned long long)) != hipSuccess) return -3;
    static const unsigned long long zero[128] = { 0 };
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phaseCounters), zero, sizeof zero) != hipSuccess) return -3;
    return 0;
}
#endif

void wavefront_reset_timing(WavefrontState& st)
{
    st.eventsUsed = 0;
    for (int k = 0; k < 5; ++k) { st.kernelMs[k] = 0.0f; st.kernelLaunches[k] = 0; }
    st.raygenBytes = 0; st.resolveBytes = 0;
}

// Record sizes (the SoA streams of WfBuffers, 16 B per float4 stream):
//   path record   rayO + rayD + thr (+ med0 + med1)              48 (80) B       hit record  16 B
//   shadow entry  sh0 + sh1 + sh4 always read, sh2 + sh3 only for an entry with an unoccluded sample; + shL 16 B per light sample
//   radiance      one float4 read-modify-write                    32 B
// Any-hit schedule (kShadowResolve): per shadow ray sqO + sqD + sqId written and read (2 x 36 B), shVis written and read per light slot.
void wavefront_queue_bytes(const WavefrontState& st, const DeviceCounters& c, uint64_t& trace, uint64_t& shade, uint64_t& shadow)
{
    const uint64_t path = st.layout.pathRecordBytes, survivors = c.closestRays > c.paths ? c.closestRays - c.paths : 0;
    trace = c.closestRays * (32 + 16);
    shade = c.closestRays * (path + 16) + survivors * path + c.neeEntries * 80 + c.neeSamples * 16 + c.radianceShade * 32;
    shadow = c.neeEntries * 48 + c.neeSamples * 16 + c.radianceShadow * (32 + 32);
    if (st.layout.shadowMode == kShadowSlim) {     // 32-byte entries; wf_shadow re-reads rayD (always) and thr (unoccluded samples) of the path's input record
        shade = c.closestRays * (path + 16) + survivors * path + c.neeEntries * 32 + c.radianceShade * 32;
        shadow = c.neeEntries * (32 + 16) + c.radianceShadow * (16 + 32);
    }
    if (st.layout.shadowMode == kShadowResolve) shadow += c.neeEntries * 32 + c.neeSamples * 16 + c.shadowRays * 72 + c.neeEntries * st.layout.maxLights * 8;
    if (st.layout.fusedPrimary) {      // bounce 0 reads no path records: wf_extend only writes hits, wf_shade reads them and stores the first radiance term
        trace -= c.paths * 16;                               // no record read, {direction, seed} written
        shade = shade - c.paths * (path - 16) + c.paths * 16;     // 16 of the 48 record bytes read; first radiance term stored
        shadow -= c.skipped16 * 16;
    }
}

namespace {
// records an event; pairs are (begin, end) around one launch of kernel class `kind`
bool timing_mark(WavefrontState& st, hipStream_t stream, int kind, bool begin)
{
    if (st.eventsUsed >= 4096) return true;   // a single render never needs more; stop timing rather than grow unbounded
    if (st.eventsUsed + 1 > st.events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return false;
        st.events.push_back(e);
    }
    if (begin) { if (st.kind.size() < st.eventsUsed / 2 + 1) st.kind.resize(st.eventsUsed / 2 + 1); st.kind[st.eventsUsed / 2] = (uint8_t)kind; }
    (void)hipEventRecord(st.events[st.eventsUsed], stream);
    st.eventsUsed++;
    return true;
}
}

hipError_t wavefront_render(WavefrontState& st, const SceneView& scene, const SceneTraits& traits, const HrptPathTracerConstants& constants,
                            uint32_t accumCount, float4* accumulation, float4* output, uint32_t width, uint32_t height, TileRect rect,
                            DeviceCounters* counters, hipStream_t stream, std::string& error)
{
    (void)height;
    if (rect.x1 <= rect.x0 || rect.y1 <= rect.y0 || rect.columns() == 0) return hipSuccess;
    hipError_t e;
    const uint32_t tilesX = rect.columns(), tilesY = (rect.y1 - rect.y0 + 7) / 8;
    const uint64_t pixelsPadded = (uint64_t)tilesX * tilesY * 64;
    // batch the accumulation indices so that one batch stays below maxSamples AND its queue pool below a byte budget: the pool takes
    // 240 B per sample with one light and no medium, but ~1.2 KB with 8 lights and non-opaque geometry (shadow-ray queue + candidate lists)
    const uint32_t maxLights = constants.m_LightCount ? constants.m_LightCount : 1;
    const uint64_t bytesPerSample = 16ull * (2 * (3 + (traits.hasMedium ? 2 : 0)) + 1 + 5 + maxLights + 1) + (scene.instances ? 4 : 0) +
                                    ((traits.hasNonOpaque || maxLights > 1) ? (16ull + 16 + 4 + 4 + 8 * kShadowCandidates) * maxLights : 0);
    uint32_t sppPerBatch = accumCount < kMaxSppPerBatch ? accumCount : kMaxSppPerBatch;
    uint64_t maxSamples = st.maxSamplesPerBatch ? st.maxSamplesPerBatch : (64ull << 20);
    if (!st.maxSamplesPerBatch && pixelsPadded * sppPerBatch * bytesPerSample > st.poolBytes) {
        // the pool has to grow: keep it within half of what the device has free (other contexts -- a second frame in flight -- need theirs)
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const uint64_t budget = (uint64_t)(freeB + st.poolBytes) / 2;
            if (budget / bytesPerSample < maxSamples) maxSamples = budget / bytesPerSample;
        }
    }
    if (maxSamples * maxLights > 0xFFFFFFFFull) maxSamples = 0xFFFFFFFFull / maxLights;      // (entry, light) slot ids are 32-bit
    while (sppPerBatch > 1 && pixelsPadded * sppPerBatch > maxSamples) --sppPerBatch;
    // ---- pool layout (again with half the accumulation indices per batch when the device refuses the allocation)
    uint64_t capacity = 0; uint32_t segs = 0;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oRayO[2], oRayD[2], oThr[2], oMed0[2] = { 0, 0 }, oMed1[2] = { 0, 0 }, oCnt[2];
    size_t oHit = 0, oHitInst = 0, oPrimary = 0, oSh0 = 0, oSh1 = 0, oSh2 = 0, oSh3 = 0, oSh4 = 0, oShL = 0, oShCnt = 0, oRad = 0;
    size_t oSqO = 0, oSqD = 0, oSqId = 0, oSqCnt = 0, oShVis = 0, oSqCand = 0;
    for (;;) {
    capacity = ((pixelsPadded * sppPerBatch + kMaxSegment - 1) / kMaxSegment) * kMaxSegment;
    if (capacity >= (1ull << 31)) { error = "tile too large for one batch"; return hipErrorInvalidValue; }
    segs = (uint32_t)(capacity / 64);   // counter arrays sized for the smallest segment
    off = 0;
    for (int p = 0; p < 2; ++p) {
        oRayO[p] = carve(capacity * 16); oRayD[p] = carve(capacity * 16); oThr[p] = carve(capacity * 16);
        if (traits.hasMedium) { oMed0[p] = carve(capacity * 16); oMed1[p] = carve(capacity * 16); }
        oCnt[p] = carve((size_t)segs * 4);
    }
    oHit = carve(capacity * 16);
    oHitInst = scene.instances ? carve(capacity * 4) : 0;
    oPrimary = carve(sizeof(PrimaryArgs));
    oSh0 = carve(capacity * 16); oSh1 = carve(capacity * 16); oSh2 = carve(capacity * 16); oSh3 = carve(capacity * 16); oSh4 = carve(capacity * 16);
    oShL = carve(capacity * 16 * maxLights);
    oShCnt = carve((size_t)segs * 4); oRad = carve(capacity * 16);
    if (traits.hasNonOpaque || maxLights > 1) {     // (the mode is picked below; the arrays are small next to the path queues)
        oSqO = carve(capacity * 16 * maxLights); oSqD = carve(capacity * 16 * maxLights); oSqId = carve(capacity * 4 * maxLights);
        oSqCnt = carve((size_t)segs * 4); oShVis = carve(capacity * 4 * maxLights); oSqCand = carve(capacity * 8 * kShadowCandidates * maxLights);
    }
    if (off <= st.poolBytes) break;
    if (st.pool) { (void)hipStreamSynchronize(stream); if (st.auxStream) (void)hipStreamSynchronize(st.auxStream); (void)hipFree(st.pool); st.pool = nullptr; st.poolBytes = 0; }
    e = hipMalloc(&st.pool, off);
    if (e == hipSuccess) { st.poolBytes = off; break; }
    (void)hipGetLastError();
    if (sppPerBatch == 1 || st.maxSamplesPerBatch) {       // nothing left to halve (or the caller fixed the batch size)
        error = "hipMalloc(queue pool, " + std::to_string(off >> 20) + " MiB for " + std::to_string(capacity) + " samples; render a smaller tile)";
        return e;
    }
    sppPerBatch = (sppPerBatch + 1) / 2;
    }
    char* base = static_cast<char*>(st.pool);
    WfArgs a{};
    a.scene = scene;
    for (int p = 0; p < 2; ++p) {
        a.b.rayO[p] = (float4*)(base + oRayO[p]); a.b.rayD[p] = (float4*)(base + oRayD[p]); a.b.thr[p] = (float4*)(base + oThr[p]);
        a.b.med0[p] = (float4*)(base + oMed0[p]); a.b.med1[p] = (float4*)(base + oMed1[p]); a.b.pathCnt[p] = (uint32_t*)(base + oCnt[p]);
    }
    a.primaryArgs = (const PrimaryArgs*)(base + oPrimary);
    a.b.hit = (float4*)(base + oHit); a.b.hitInst = scene.instances ? (uint32_t*)(base + oHitInst) : nullptr;
    a.b.sh0 = (float4*)(base + oSh0); a.b.sh1 = (float4*)(base + oSh1); a.b.sh2 = (float4*)(base + oSh2); a.b.sh3 = (float4*)(base + oSh3);
    a.b.sh4 = (float4*)(base + oSh4); a.b.shL = (float4*)(base + oShL);
    a.b.shadowCnt = (uint32_t*)(base + oShCnt); a.b.radiance = (float4*)(base + oRad);
    a.b.sqO = (float4*)(base + oSqO); a.b.sqD = (float4*)(base + oSqD); a.b.sqId = (uint32_t*)(base + oSqId); a.b.sqCnt = (uint32_t*)(base + oSqCnt); a.b.shVis = (uint32_t*)(base + oShVis); a.b.sqCand = (uint2*)(base + oSqCand);
    a.tilesX = tilesX; a.tilesY = tilesY; a.rect = rect; a.imageWidth = width; a.pixelsPadded = (uint32_t)pixelsPadded;
    a.maxLights = maxLights; a.hasMedium = traits.hasMedium ? 1u : 0u; a.hasStochasticAlpha = traits.hasStochasticAlpha ? 1u : 0u; a.allOpaque = traits.hasNonOpaque ? 0u : 1u;
    st.layout.pathRecordBytes = traits.hasMedium ? 80u : 48u; st.layout.maxLights = maxLights;
    a.counters = counters;
    a.refillMin = st.refillMin ? st.refillMin : kRefillMinDefault;
    a.streamSegments = st.drainSegments ? 0u : 1u;
    // default: on for scenes that sample textures (the longest branch of shade_surface_a; Sponza-class config: same shade time, -15 % VALU
    // instructions), off otherwise (glass config: the sort costs 3.5 % of wf_shade, its classes are too few per segment to fill iterations)
    a.sortShade = st.shadeSort < 0 ? (traits.hasTextures ? 1u : 0u) : (uint32_t)st.shadeSort;
    a.nodeLoopMin = 0;      // set below, once the traversal variant is known

    // ---- kernel variants and grids
    int dev = 0; hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) { error = "hipGetDeviceProperties"; return e; }
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    tlCus = cus;
    // Node width per kernel class (measured, scripts/gpu_bvh4_ab.sh): the 4-wide tree wins for closest-hit queries everywhere
    // (fewer, fuller steps: -9..-11% extend time on configs 2/4/5) and for shadow queries that buffer non-opaque candidates or
    // read the BVH from global memory (-9..-14%); the small opaque any-hit kernel over an LDS-resident BVH is faster 2-wide
    // (the 4-wide step costs it 12 VGPRs = one wave of occupancy).
    const size_t candBytes = traits.hasNonOpaque ? (size_t)kShadowCandidates * 2 * kBlock * 4 : 0;
    auto pick = [&](int width, size_t extraBytes, int ldsStackMax) {
        Variant v; v.width = width;
        if (traits.twoLevelStackNeed) {      // two-level structure: 4-wide trees in global memory, its own kernels
            v.width = 4; v.twoLevel = true; v.lds = false; v.twoLevelCandidates = traits.hasNonOpaque;
            v.depth = traits.twoLevelStackNeed <= 16 ? 16 : (traits.twoLevelStackNeed <= 32 ? 32 : 64);
            v.ldsBytes = (size_t)(v.depth > ldsStackMax ? ldsStackMax : v.depth) * kBlock * 4 + st.padLdsBytes;
            return v;
        }
        if (v.width == 4 && 3 * traits.bvh4MaxDepth + 2 > kMaxStackNeed) v.width = 2;
        if (v.width == 2) v.depth = traits.bvhMaxDepth + 2 <= 8 ? 8 : (traits.bvhMaxDepth + 2 <= 16 ? 16 : (traits.bvhMaxDepth + 2 <= 32 ? 32 : 64));
        else v.depth = 3 * traits.bvh4MaxDepth + 2 <= 16 ? 16 : (3 * traits.bvh4MaxDepth + 2 <= 32 ? 32 : 64);
        const size_t bvhBytes = (v.width == 2 ? (size_t)scene.nodeCount * 64 : (size_t)scene.node4Count * kLdsNode4Stride) + (size_t)scene.triCount * 48;
        const size_t stackBytes = (size_t)(v.depth > ldsStackMax ? ldsStackMax : v.depth) * kBlock * 4;
        v.lds = bvhBytes > 0 && stackBytes + extraBytes + bvhBytes <= kLdsBudget && !st.forceGlobalBvh;
        v.ldsBytes = stackBytes + (v.lds ? bvhBytes : 0) + st.padLdsBytes;
        v.quantised = v.width == 4 && !v.lds && traits.quantisedNodes && scene.nodesQ != nullptr;
        return v;
    };
    const int forced = st.bvhWidth == 2 ? 2 : (st.bvhWidth == 4 ? 4 : 0);
    const Variant vE = pick(forced ? forced : 4, 0, kExtendLdsStack);
    Variant vS = pick(forced ? forced : 4, candBytes, kShadowLdsStack);
    if (!forced && vS.lds && !traits.hasNonOpaque) vS = pick(2, candBytes, kShadowLdsStack);
    // Shadow-ray schedule. Several lights per vertex, or glass (rays that cross many non-opaque triangles), make the per-ray cost very
    // uneven; when the tree is in global memory their visibility traversal runs in the refilling traversal kernel (wf_shadow_rays +
    // wf_extend<ANYHIT>, which also records the crossed non-opaque triangles) and wf_shadow only resolves: glass config 4.7 -> 4.2 ms per
    // bounce, an opaque 101 k-triangle scene with three lights 20.3 -> 18.4 ms per frame. With an LDS-resident tree the traversal is too
    // cheap to pay for the extra passes (Cornell box with three lights 11.3 vs 13.1 ms), and a single sun over alpha-tested foliage
    // (config 4) is faster with wf_shadow's own buffered query (0.96 vs 1.07 ms per bounce). HRPT_WF_SHADOW_PATH = 1 / 2 forces either.
    const bool unevenRays = maxLights > 1 || (traits.hasNonOpaque && traits.hasTransmissiveOrBlend);
    const int selfMode = traits.hasNonOpaque ? kShadowBuffered : kShadowOpaque;
    int shadowMode = (!vS.lds && unevenRays) ? kShadowResolve : selfMode;
    if (st.shadowPath == 1) shadowMode = selfMode;
    if (st.shadowPath == 2 && (traits.hasNonOpaque || maxLights > 1)) shadowMode = kShadowResolve;
    // two-level scenes: wf_shadow traverses itself (no any-hit pass over that structure); TL = 1 instantiations when every instance is opaque, TL = 2
    // (buffered candidates of non-opaque instances, shadow_query_two_level_buffered) otherwise -- launch_shadow picks by Variant::twoLevelCandidates
    if (traits.twoLevelStackNeed) shadowMode = kShadowOpaque;
    // any-hit pass over the shadow rays (same kernel family as vE). wf_extend<ANYHIT> always carves its candidate columns out of LDS
    // (launch_extend_t adds them to the launch), opaque scene or not, so the budget check must count them too.
    const Variant vA = pick(forced ? forced : 4, (size_t)kShadowCandidates * 2 * kBlock * 4, kExtendLdsStack);
    if (shadowMode == kShadowResolve) vS = pick(forced ? forced : 4, 0, kExtendLdsStack);
    const bool manyLightsEarly = maxLights > 1;
    const bool simpleEarly = !traits.hasTextures && !traits.hasTransmissiveOrBlend && traits.directionalLightsOnly && !st.forceGeneralShade;
    // slim shadow-queue entries: the SIMPLE single-light shade variant feeding the plain opaque any-hit query (HRPT_WF_SLIM_SHADOW=0 keeps the 96-byte entries)
    const bool slim = simpleEarly && !manyLightsEarly && shadowMode == kShadowOpaque && !st.noSlimShadow;
    if (slim) shadowMode = kShadowSlim;
    a.slimShadow = slim ? 1u : 0u;
    st.layout.shadowMode = shadowMode;
    // Thresholded while-while (measured, scripts/env_sweep.sh HRPT_WF_NODE_LOOP_MIN): 16 lanes for an LDS-resident tree (config 2 extend -3 %),
    // 24 for a tree in global memory (config 4 extend -11 %, glass config extend -24 % and its any-hit pass -14 %)
    a.nodeLoopMin = st.nodeLoopMin != ~0u ? st.nodeLoopMin : (vE.lds ? 16u : 24u);
    // more blocks than fit: the dispatcher back-fills CUs as blocks retire (scripts/knob_sweep.py). A context that is one lane of a
    // two-frames-in-flight loop (hrpt_set_shadow_overlap(ctx, 0)) and traverses a tree in global memory does better with half the grid:
    // its latency-bound kernels leave room for the other lane's (config 4 14.4 -> 14.0 ms, config 5 22.6 -> 21.9 ms per frame).
    const uint32_t blocksPerCu = st.blocksPerCu ? st.blocksPerCu : ((st.serialShadow && !vE.lds) ? 8 : 16);
    // wf_extend gets a grid of its own: a whole number of rounds of the six blocks a CU holds of it. Measured (scripts/env_sweep.sh
    // HRPT_WF_EXTEND_BLOCKS_PER_CU): tree in LDS 12 per CU (two rounds; 16 = 2.67 rounds: +8 % on config 2, the last round runs with four of six
    // slots filled), tree in global memory 6 (one persistent round: its waves are latency-bound and every further round re-pays the ramp:
    // config 4 extend -6 %, glass config -20 %; the two-level kernels hold five blocks per CU and launch_rounds trims the six to that: -6 / -10 %
    // on instanced scenes of opaque / non-opaque materials).
    const uint32_t extendBlocksPerCu = st.extendBlocksPerCu ? st.extendBlocksPerCu : (st.blocksPerCu ? st.blocksPerCu : (vE.lds ? 12u : 6u));
    const uint32_t maxBlocksPerCu = blocksPerCu > extendBlocksPerCu ? blocksPerCu : extendBlocksPerCu;
    if (vE.depth > kExtendLdsStack || vS.depth > kShadowLdsStack) {
        // stack overflow columns for trees whose worst-case stack need exceeds the LDS entries (see LdsStack); sized for the smaller LDS part
        const uint32_t worst = traits.twoLevelStackNeed ? traits.twoLevelStackNeed : (vE.width == 4 || vS.width == 4 ? 3 * traits.bvh4MaxDepth + 2 : traits.bvhMaxDepth + 2);
        const uint32_t entries = worst > (uint32_t)kExtendLdsStack ? worst - kExtendLdsStack : 1u;
        const size_t threads = (size_t)cus * maxBlocksPerCu * kBlock, bytes = 2 * threads * entries * 4;
        if (bytes > st.spillBytes) {
            if (st.spill) { (void)hipStreamSynchronize(stream); if (st.auxStream) (void)hipStreamSynchronize(st.auxStream); (void)hipFree(st.spill); st.spill = nullptr; st.spillBytes = 0; }
            if ((e = hipMalloc(&st.spill, bytes)) != hipSuccess) { error = "hipMalloc(traversal stack overflow)"; return e; }
            st.spillBytes = bytes;
        }
        a.spill[0] = static_cast<int32_t*>(st.spill); a.spill[1] = a.spill[0] + threads * entries;
    }

    const bool manyLights = maxLights > 1;
    const bool simpleScene = !traits.hasTextures && !traits.hasTransmissiveOrBlend && traits.directionalLightsOnly && !st.forceGeneralShade;
    for (uint32_t first = 0; first < accumCount; first += sppPerBatch) {
        const uint32_t spp = (accumCount - first) < sppPerBatch ? (accumCount - first) : sppPerBatch;
        a.spp = spp; a.numSamples = (uint32_t)(pixelsPadded * spp);
        // segment size: large segments amortise the partially filled last 64-lane iteration of every segment (after compaction a
        // segment holds ~80 % / 65 % / 53 % of its slots at bounces 1 / 2 / 3), small ones give every SIMD several waves when the batch is
        // small (tile-sharded multi-GPU runs) and balance uneven per-entry work (several lights per shadow entry). Measured on MI355X
        // (scripts/sweep_env.sh, scripts/seg_sweep.py): 512 wins for full-frame single-light batches (-3 % config 2, -4 % config 4),
        // 256 for a 135-row band (0.76 vs 0.89 ms) and for the three-light glass scene.
        // (512 only with an LDS-resident tree: rays through a big tree in global memory differ too much in length -- 256 is 2..4 % faster
        // there: config 4 14.4 -> 14.2 ms, 1.17 M triangles 20.1 -> 19.4 ms)
        const bool largeBatch = a.numSamples >= (8u << 20) && maxLights == 1 && vE.lds;
        uint32_t shift = st.segmentShift ? st.segmentShift : (largeBatch ? 9u : 8u);
        if (shift < 6) shift = 6;
        if (shift > 10) shift = 10;
        a.segSize = 1u << shift;
        if (st.segmentSize >= 64u && st.segmentSize <= kMaxSegment) a.segSize = st.segmentSize;       // HRPT_WF_SEGMENT_SIZE: any size (experiments)
        a.numSegments = (a.numSamples + a.segSize - 1) / a.segSize;
        const uint32_t wavesNeeded = a.numSegments, blocksNeeded = (wavesNeeded + 3) / 4;
        uint32_t grid = cus * blocksPerCu; if (grid > blocksNeeded) grid = blocksNeeded; if (grid == 0) grid = 1;
        uint32_t gridExtend = cus * extendBlocksPerCu; if (gridExtend > blocksNeeded) gridExtend = blocksNeeded; if (gridExtend == 0) gridExtend = 1;

        HrptPathTracerConstants cb = constants;
        cb.m_AccumulationIndex = constants.m_AccumulationIndex + first;
        JitterTable jt;
        for (uint32_t k = 0; k < spp; ++k) {   // PathTracerRenderer.cpp:65
            jt.j[k].x = hrpt_halton(cb.m_AccumulationIndex + k + 1, 2) - 0.5f;
            jt.j[k].y = hrpt_halton(cb.m_AccumulationIndex + k + 1, 3) - 0.5f;
        }
        // SIMPLE scenes: no raygen pass. wf_extend<PRIMARY> (the bounce-0 launch) derives the primary ray and the RNG seed of a slot (= sample index) from
        // PrimaryArgs in its refill and leaves {direction, seed} in rayD for wf_shade<PRIMARY> / wf_shadow, which take the camera position as origin and
        // (1, 1, 1) as throughput; wf_shade(0) stores the first radiance term instead of adding to a zeroed array; padding pixels of the 8 x 8 tiles get
        // a kNoPathRecord hit record. 96 B per sample less queue traffic and one launch less: config 2 -3 % one frame at a time, -4 % two in flight
        // (HRPT_WF_FUSED_PRIMARY=0 keeps wf_raygen). As run-time branches inside the ordinary kernels the same code cost every bounce 6 % (extend) and
        // 16 % (shade): the extra live values; and regenerating the ray in wf_shade instead of reading 16 bytes gave the saving back in instructions.
        const bool fusedPrimary = simpleScene && !manyLights && maxLights <= kMaxLights && vE.width == 4 && !traits.hasMedium && !traits.hasStochasticAlpha && !st.noFusedPrimary;
        PrimaryArgs pr{};
        if (fusedPrimary) {
            for (int i = 0; i < 16; ++i) pr.clipToWorld[i] = cb.m_View.m_MatClipToWorldNoOffset[i];
            for (int i = 0; i < 3; ++i) pr.cam[i] = cb.m_CameraPos[i];
            pr.invW = cb.m_View.m_ViewportSizeInv[0]; pr.invH = cb.m_View.m_ViewportSizeInv[1]; pr.accumIndex = cb.m_AccumulationIndex;
            for (uint32_t k = 0; k < spp; ++k) pr.j[k] = jt.j[k];
        }
        if (fusedPrimary) hipLaunchKernelGGL(wf_store_primary, dim3(1), dim3(64), 0, stream, pr, const_cast<PrimaryArgs*>(a.primaryArgs));
        st.layout.fusedPrimary = fusedPrimary;
        const bool timedEnds = st.profile && st.eventsUsed + 4 <= 4096;
        if (!fusedPrimary) {
            if (timedEnds) timing_mark(st, stream, 3, true);
            hipLaunchKernelGGL(wf_raygen, dim3(grid), dim3(kBlock), 0, stream, a, cb, jt);
            if (timedEnds) timing_mark(st, stream, 3, false);
        }
        {   // raygen: sampleRadiance zeroed + one path record per pixel of the rectangle and index; resolve: sampleRadiance read, Accumulation
            // read (when resuming) and written, Output written
            const uint64_t w = (uint64_t)rect.columns() * 8u, px = (w < rect.x1 - rect.x0 ? w : (uint64_t)rect.x1 - rect.x0) * (rect.y1 - rect.y0);
            if (!fusedPrimary) st.raygenBytes += (uint64_t)a.numSamples * 16 + px * spp * st.layout.pathRecordBytes;
            st.resolveBytes += px * ((uint64_t)spp * 16 + 32 + (cb.m_AccumulationIndex > 0 ? 16 : 0));
        }
        const int maxBounces = (int)cb.m_MaxBounces;
        // wf_shadow(b) only reads the shadow queue of shade(b) and adds into sampleRadiance; wf_extend(b+1) reads the path queue and
        // writes hit records: no shared buffer, so the two run concurrently (fork after shade(b), join before shade(b+1), which both
        // rewrites the shadow queue and adds the next radiance term -- the per-sample order of additions is unchanged).
        // Per-launch timing (HRPT_FRAME_PROFILE) needs stream order, so profiling renders serially.
        bool overlap = !st.profile && !st.serialShadow && maxBounces > 1;
        if (overlap) {
            if (!st.auxStream && hipStreamCreateWithFlags(&st.auxStream, hipStreamNonBlocking) != hipSuccess) { st.auxStream = nullptr; overlap = false; }
            while (overlap && st.forkEvents.size() < (size_t)maxBounces) {
                hipEvent_t f, j;
                if (hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess) { overlap = false; break; }
                if (hipEventCreateWithFlags(&j, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(f); overlap = false; break; }
                st.forkEvents.push_back(f); st.joinEvents.push_back(j);
            }
        }
        // NEE visibility + accumulation. With non-opaque geometry in the scene: ray generation, the opaque any-hit pass in the refilling
        // traversal kernel, then wf_shadow for the contributions and the (few) rays that crossed non-opaque triangles.
        auto shadow_stage = [&](hipStream_t sst, int bounce) {
            a.shadowParity = (uint32_t)bounce & 1u;
            if (shadowMode == kShadowResolve) {
                if (traits.directionalLightsOnly) hipLaunchKernelGGL((wf_shadow_rays<true>), dim3(grid), dim3(kBlock), 0, sst, a, cb);
                else hipLaunchKernelGGL((wf_shadow_rays<false>), dim3(grid), dim3(kBlock), 0, sst, a, cb);
                launch_extend(vA, dim3(grid), vA.ldsBytes, sst, a, 0u, true);
            }
            launch_shadow(vS, dim3(grid), vS.ldsBytes, sst, a, cb, bounce, traits.directionalLightsOnly, shadowMode);
        };
        bool pendingJoin = false;
        for (int bounce = 0; bounce < maxBounces; ++bounce) {
            const uint32_t parity = (uint32_t)bounce & 1u;
            const bool timed = st.profile && st.eventsUsed + 8 <= 4096;
            if (timed) timing_mark(st, stream, 0, true);
            a.primary = (fusedPrimary && bounce == 0) ? 1u : 0u;
            launch_extend(vE, dim3(gridExtend), vE.ldsBytes, stream, a, parity);
            if (timed) { timing_mark(st, stream, 0, false); timing_mark(st, stream, 1, true); }
            if (pendingJoin) { if ((e = hipStreamWaitEvent(stream, st.joinEvents[(size_t)bounce - 1], 0)) != hipSuccess) { error = "hipStreamWaitEvent(join)"; return e; } pendingJoin = false; }
            const int last = bounce + 1 == maxBounces ? 1 : 0;
            const size_t sortLds = (size_t)(kBlock / 64) * ((size_t)5 * a.segSize);      // per wave: two uint16 permutations (segments A, B) + uint8 class keys
            if (maxLights > kMaxLights) hipLaunchKernelGGL((wf_shade<0, false>), dim3(grid), dim3(kBlock), sortLds, stream, a, cb, parity, bounce, last);
            else if (manyLights) hipLaunchKernelGGL((wf_shade<(int)kMaxLights, false>), dim3(grid), dim3(kBlock), sortLds, stream, a, cb, parity, bounce, last);
            else if (simpleScene && a.primary) hipLaunchKernelGGL((wf_shade<1, true, true>), dim3(grid), dim3(kBlock), (kBlock / 64) * kShadeRing * 23 * 4, stream, a, cb, parity, bounce, last);
            else if (simpleScene) hipLaunchKernelGGL((wf_shade<1, true>), dim3(grid), dim3(kBlock), (kBlock / 64) * kShadeRing * 23 * 4, stream, a, cb, parity, bounce, last);
            else hipLaunchKernelGGL((wf_shade<1, false>), dim3(grid), dim3(kBlock), sortLds, stream, a, cb, parity, bounce, last);
            if (timed) { timing_mark(st, stream, 1, false); timing_mark(st, stream, 2, true); }
            if (overlap) {
                if ((e = hipEventRecord(st.forkEvents[(size_t)bounce], stream)) != hipSuccess || (e = hipStreamWaitEvent(st.auxStream, st.forkEvents[(size_t)bounce], 0)) != hipSuccess) { error = "fork to the shadow stream"; return e; }
                shadow_stage(st.auxStream, bounce);
                if ((e = hipEventRecord(st.joinEvents[(size_t)bounce], st.auxStream)) != hipSuccess) { error = "hipEventRecord(join)"; return e; }
                pendingJoin = true;
            } else {
                shadow_stage(stream, bounce);
            }
            if (timed) timing_mark(st, stream, 2, false);
        }
        if (pendingJoin && (e = hipStreamWaitEvent(stream, st.joinEvents[(size_t)maxBounces - 1], 0)) != hipSuccess) { error = "hipStreamWaitEvent(join)"; return e; }
        uint32_t rgrid = (uint32_t)((pixelsPadded + kBlock - 1) / kBlock); if (rgrid > cus * 8) rgrid = cus * 8;
        if (timedEnds) timing_mark(st, stream, 4, true);
        hipLaunchKernelGGL(wf_resolve, dim3(rgrid), dim3(kBlock), 0, stream, a, accumulation, output, cb.m_AccumulationIndex);
        if (timedEnds) timing_mark(st, stream, 4, false);
        if ((e = hipGetLastError()) != hipSuccess) { error = "kernel launch"; return e; }
    }
    return hipSuccess;
}

} // namespace hrt
