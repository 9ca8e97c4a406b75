// pt_capi.cpp -- implementation of the C ABI declared in include/hobbyrt_pt.h. Host code only:
// context/stream ownership, scene upload (copies), BVH build, per-dispatch constants exactly as
// PathTracerRenderer::Render fills them (/root/reference/src/PathTracerRenderer.cpp:58-75), launches.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <limits>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/hobbyrt_pt.h"
#include "bvh_build.h"
#include "bvh_build_gpu.h"
#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_wavefront.h"

using namespace hrt;

struct HrptContext {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t ownStream = nullptr;         // created by hrpt_create; `stream` may be redirected by hrpt_set_stream
    std::string err;
    // scene
    std::vector<void*> allocations;          // scene-lifetime device allocations
    std::vector<void*> bvhAllocations;       // acceleration structure of the host builder + per-instance records: replaced by hrpt_update_instances
    GpuNodeQ* nodesQ = nullptr; size_t nodesQCapacity = 0;    // quantised copy of the flat 4-wide tree (pt_device.h GpuNodeQ), kept across rebuilds
    GpuBvhBuilder* gpuBuilder = nullptr;     // GPU builders: geometry + build buffers stay on the device for rebuilds
    GpuBvhBuilder* tlasBuilder = nullptr; uint32_t tlasBuilderInstances = 0;   // two-level structure: the tree over the instances, built on the GPU (build_two_level)
    // host copy of what a rebuild needs (the reference's Scene keeps the same vectors: m_InstanceData, m_Vertices, m_Indices, m_MeshData)
    std::vector<HrptVertexQuantized> keptVertices; std::vector<uint32_t> keptIndices; std::vector<HrptMeshData> keptMeshData;
    std::vector<HrptPerInstanceData> keptInstances; std::vector<HrptMaterialConstants> keptMaterials; std::vector<HrptGPULight> keptLights;
    size_t lightCapacity = 0;                // entries the device light buffer can hold (hrpt_update_lights may grow it)
    SceneView view{};
    bool haveScene = false;
    uint32_t bvhNodes = 0, bvhTris = 0;
    // images
    uint32_t width = 0, height = 0;
    float4* dAccum = nullptr; float4* dOutput = nullptr; float4* dDisplay = nullptr;
    uint32_t* dHistogram = nullptr; float* dExposure = nullptr;   // persistent exposure buffer (HDRRenderer m_RG_ExposureBuffer)
    DeviceCounters* dCounters = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    bool timed = false;
    WavefrontState wf;
    SceneTraits traits;
    int bvhBuilder = HRPT_BVH_BUILDER_AUTO;       // hrpt_set_bvh_builder
    int accelStructure = HRPT_ACCEL_AUTO;         // hrpt_set_acceleration_structure
    BuiltTwoLevel* twoLevel = nullptr;            // two-level scenes: host copy (hrpt_update_instances rebuilds the instance tree from it)
    std::vector<void*> meshAllocations;           // ... and the device copies of the per-mesh arrays, which survive instance updates
    uint32_t megakernelFallbacks = 0;             // renders that wanted the wavefront pipeline but could not use it (HrptStats)
    HrptBuildInfo buildInfo{};
};

static std::mutex g_errMutex;
static std::string g_createError;

static int fail(HrptContext* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    else { std::lock_guard<std::mutex> l(g_errMutex); g_createError = msg; }
    return code;
}
#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? HRPT_ERR_OUT_OF_MEMORY : HRPT_ERR_HIP, \
                                          std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// DirectX::PackedVector::XMConvertFloatToHalf (round to nearest even), src/CommonResources.cpp:553
static uint16_t float_to_half(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? (0x200u | ((x >> 13) & 0x3ffu)) : 0u));
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        uint32_t shift = 126u - (x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t r = x + 0xfffu + ((x >> 13) & 1u);
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}

static void free_acceleration(HrptContext* c, bool keepGpuBuilder)
{
    for (void* p : c->bvhAllocations) (void)hipFree(p);
    c->bvhAllocations.clear();
    if (!keepGpuBuilder) {
        delete c->gpuBuilder; c->gpuBuilder = nullptr;
        delete c->tlasBuilder; c->tlasBuilder = nullptr; c->tlasBuilderInstances = 0;
        for (void* p : c->meshAllocations) (void)hipFree(p);
        c->meshAllocations.clear();
        delete c->twoLevel; c->twoLevel = nullptr;
    }
}

static void free_scene(HrptContext* c)
{
    free_acceleration(c, false);
    for (void* p : c->allocations) (void)hipFree(p);
    c->allocations.clear();
    if (c->nodesQ) { (void)hipFree(c->nodesQ); c->nodesQ = nullptr; c->nodesQCapacity = 0; }
    c->keptVertices.clear(); c->keptIndices.clear(); c->keptMeshData.clear(); c->keptInstances.clear(); c->keptMaterials.clear(); c->keptLights.clear(); c->lightCapacity = 0;
    c->haveScene = false;
    memset(&c->view, 0, sizeof c->view);
}

template <class T>
static int upload(HrptContext* c, const T* host, size_t count, const T** dev, std::vector<void*>* owner = nullptr)
{
    *dev = nullptr;
    size_t bytes = count * sizeof(T);
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, bytes ? bytes : 16));
    (owner ? *owner : c->allocations).push_back(p);
    if (bytes) HIP_TRY(c, hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, c->stream));
    *dev = static_cast<const T*>(p);
    return HRPT_OK;
}

extern "C" {

int hrpt_create(const HrptDeviceDesc* desc, HrptContext** out)
{
    if (!desc || !out) return fail(nullptr, HRPT_ERR_INVALID_ARGUMENT, "hrpt_create: null argument");
    *out = nullptr;
    if (desc->abiVersion != HRPT_ABI_VERSION) return fail(nullptr, HRPT_ERR_INVALID_ARGUMENT, "hrpt_create: ABI version mismatch");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, HRPT_ERR_NO_DEVICE, "hrpt_create: no HIP device available (the gfx950 kernels are the only backend)");
    if (desc->deviceOrdinal < 0 || desc->deviceOrdinal >= n) return fail(nullptr, HRPT_ERR_NO_DEVICE, "hrpt_create: device ordinal out of range");
    HrptContext* c = new HrptContext();
    c->device = desc->deviceOrdinal;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->evStart) != hipSuccess || hipEventCreate(&c->evStop) != hipSuccess ||
        hipMalloc((void**)&c->dCounters, sizeof(DeviceCounters) * kCounterShards) != hipSuccess ||
        hipMemset(c->dCounters, 0, sizeof(DeviceCounters) * kCounterShards) != hipSuccess) {
        int r = fail(nullptr, HRPT_ERR_HIP, "hrpt_create: stream/event/counter creation failed");
        delete c;
        return r;
    }
    c->ownStream = c->stream;
    if (const char* e = getenv("HRPT_WF_SEGMENT_SHIFT")) c->wf.segmentShift = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_WF_SEGMENT_SIZE")) c->wf.segmentSize = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_WF_BLOCKS_PER_CU")) c->wf.blocksPerCu = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_WF_EXTEND_BLOCKS_PER_CU")) c->wf.extendBlocksPerCu = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_WF_REFILL_MIN")) c->wf.refillMin = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_WF_BVH_WIDTH")) c->wf.bvhWidth = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_BVH_BUILDER")) c->bvhBuilder = (strcmp(e, "ploc") == 0 || strcmp(e, "2") == 0) ? HRPT_BVH_BUILDER_GPU_PLOC : ((strcmp(e, "gpu") == 0 || strcmp(e, "lbvh") == 0 || strcmp(e, "1") == 0) ? HRPT_BVH_BUILDER_GPU_LBVH : ((strcmp(e, "auto") == 0 || strcmp(e, "3") == 0) ? HRPT_BVH_BUILDER_AUTO : HRPT_BVH_BUILDER_HOST_SAH));
    if (const char* e = getenv("HRPT_WF_PAD_LDS")) c->wf.padLdsBytes = (uint32_t)atoi(e);
    if (const char* e = getenv("HRPT_WF_DRAIN_SEGMENTS")) c->wf.drainSegments = atoi(e) != 0;
    if (const char* e = getenv("HRPT_WF_SERIAL_SHADOW")) c->wf.serialShadow = atoi(e) != 0;
    if (const char* e = getenv("HRPT_WF_SHADOW_PATH")) c->wf.shadowPath = atoi(e);
    if (const char* e = getenv("HRPT_WF_SHADE_SORT")) c->wf.shadeSort = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("HRPT_WF_SLIM_SHADOW")) c->wf.noSlimShadow = atoi(e) == 0;
    if (const char* e = getenv("HRPT_WF_FUSED_PRIMARY")) c->wf.noFusedPrimary = atoi(e) == 0;
    if (const char* e = getenv("HRPT_WF_NODE_LOOP_MIN")) c->wf.nodeLoopMin = (uint32_t)atoi(e);
    *out = c;
    return HRPT_OK;
}

void hrpt_destroy(HrptContext* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    wavefront_release(c->wf);
    if (c->dAccum) (void)hipFree(c->dAccum);
    if (c->dOutput) (void)hipFree(c->dOutput);
    if (c->dDisplay) (void)hipFree(c->dDisplay);
    if (c->dHistogram) (void)hipFree(c->dHistogram);
    if (c->dExposure) (void)hipFree(c->dExposure);
    if (c->dCounters) (void)hipFree(c->dCounters);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

const char* hrpt_last_error(const HrptContext* c)
{
    if (c) return c->err.c_str();
    std::lock_guard<std::mutex> l(g_errMutex);
    static thread_local std::string copy;
    copy = g_createError;
    return copy.c_str();
}

// The acceleration structure + the records derived from instance transforms (Scene::BuildAccelerationStructures, src/Scene.cpp:67-214),
// written into `v`. First build of a scene or a rebuild after hrpt_update_instances (the GPU builder then keeps its device-resident
// geometry and buffers).
// Two-level structure: asked for, or (AUTO) large and heavily instanced
static bool two_level_wanted(const HrptContext* c, const HrptSceneDesc& s, uint64_t sceneTris)
{
    int want = c->accelStructure;
    if (const char* e = getenv("HRPT_ACCEL_STRUCTURE")) { const int v = atoi(e); if (v >= HRPT_ACCEL_AUTO && v <= HRPT_ACCEL_TWO_LEVEL) want = v; }
    if (want == HRPT_ACCEL_FLAT || s.instanceCount == 0) return false;
    if (want == HRPT_ACCEL_TWO_LEVEL) return true;
    std::vector<uint8_t> used(s.meshDataCount, 0); uint32_t distinct = 0;
    for (uint32_t i = 0; i < s.instanceCount; ++i) if (!used[s.instances[i].m_MeshDataIndex]) { used[s.instances[i].m_MeshDataIndex] = 1; ++distinct; }
    // scenes with non-opaque instances: measured cross-over against the flat structure at ~16 M world triangles (instanced alpha-tested + glass
    // meshes: 7.6 M triangles 35.5 vs 32.3 ms, 30 M 38 vs 46 ms; the two-level candidate buffer holds 4 entries with the instance next to the triangle)
    bool nonOpaque = false;
    for (uint32_t i = 0; i < s.instanceCount && !nonOpaque; ++i) nonOpaque = s.materials[s.instances[i].m_MaterialIndex].m_AlphaMode != HRPT_ALPHA_MODE_OPAQUE;
    return sceneTris >= (nonOpaque ? (16ull << 20) : (2ull << 20)) && (uint64_t)s.instanceCount >= 8ull * distinct;
}

// instancesOnly: the mesh trees of c->twoLevel are kept (hrpt_update_instances)
// kTwoLevelDoesNotFit: the scene cannot be held in this form (an instance with a singular world matrix -- a mesh flattened to a plane --, trees
// too deep): the caller builds the flat structure instead, which has no such limits
constexpr int kTwoLevelDoesNotFit = 1;
// The tree over the instances on the GPU (the reference rebuilds its TLAS on the GPU every frame, src/CommonRenderers.cpp:234-246): the
// builder's box mode over the instances' padded world boxes, then launch_tlas_fixup writes the nodes, leaves turned into instance references,
// to the front of the scene's node array. The builder and its buffers stay on the device: a rebuild (hrpt_update_instances) uploads 24 bytes
// per instance and runs the kernels. false: not built (a device error, a tree too deep): the caller builds the tree on the host instead.
static bool build_instance_tree_on_gpu(HrptContext* c, uint32_t instanceCount, const std::vector<float>& boxes, bool rebuild, bool refit, GpuNode4* dstNodes, uint32_t& depth4Levels)
{
    std::string gerr;
    if (!c->tlasBuilder || c->tlasBuilderInstances != instanceCount) {
        delete c->tlasBuilder; c->tlasBuilder = new GpuBvhBuilder(); c->tlasBuilderInstances = 0;
        if (c->tlasBuilder->prepare_boxes(instanceCount, c->stream, gerr) != hipSuccess) { delete c->tlasBuilder; c->tlasBuilder = nullptr; return false; }
        c->tlasBuilderInstances = instanceCount;
    }
    GpuBuiltBvh g;
    // Hierarchy: PLOC at upload, the Morton radix tree for rebuilds (hrpt_update_instances) unless a GPU builder was asked for by name. Measured on
    // 16 384 / 65 536 instances: the radix tree is built in 0.45 ms of device time against 1.9 / 2.1 ms and traverses 0 / 2 % slower, so a host that
    // moves instances every frame comes out ahead with it (update 1.8 / 4.0 ms against 3.0 / 5.5 ms), a static scene with PLOC.
    bool ploc = c->bvhBuilder == HRPT_BVH_BUILDER_GPU_PLOC || (c->bvhBuilder != HRPT_BVH_BUILDER_GPU_LBVH && !rebuild);
    if (const char* e = getenv("HRPT_TLAS_LBVH")) ploc = atoi(e) == 0;
    const hipError_t ge = (rebuild && refit && c->tlasBuilder->can_refit()) ? c->tlasBuilder->refit_boxes(boxes.data(), c->stream, g, gerr)      // hrpt_refit_instances
                                                                              : c->tlasBuilder->build_boxes(boxes.data(), ploc, kTraversalStackDepth, c->stream, g, gerr);
    if (ge != hipSuccess || g.maxDepth + 2 > kTraversalStackDepth || g.node4Count == 0 || g.node4Count > instanceCount) return false;
    if (launch_tlas_fixup(g.nodes4, g.node4Count, g.leafOrder, dstNodes, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return false;
    depth4Levels = g.maxDepth4 + 1;
    c->buildInfo.deviceBuildMs = g.deviceMs; c->buildInfo.usedBuilder = (g.ploc ? HRPT_BVH_BUILDER_GPU_PLOC : HRPT_BVH_BUILDER_GPU_LBVH) | (g.refitted ? HRPT_BVH_BUILDER_REFITTED : 0u);
    return true;
}

static int build_two_level(HrptContext* c, const HrptSceneDesc& s, SceneView& v, bool instancesOnly, bool refit)
{
    std::string berr; int r;
    const bool timing = getenv("HRPT_BUILD_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; (void)hipStreamSynchronize(c->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[two-level] %-22s %7.3f ms\n", what, std::chrono::duration<float, std::milli>(t - tp).count()); tp = t; };
    // who builds the tree over the instances: the GPU from 1024 instances on (host SAH: 2 / 6 / 16 ms for 4 096 / 16 384 / 65 536 instances, the GPU
    // ~1 ms), unless the host builder was asked for (hrpt_set_bvh_builder) or HRPT_TLAS_BUILDER says otherwise
    bool gpuTree = s.instanceCount >= 1024 && c->bvhBuilder != HRPT_BVH_BUILDER_HOST_SAH;
    if (const char* e = getenv("HRPT_TLAS_BUILDER")) gpuTree = s.instanceCount >= 8 && (strcmp(e, "gpu") == 0 || strcmp(e, "1") == 0);
    lap("(entry)");
    std::vector<float> boxes;
    std::vector<float>* wantBoxes = gpuTree ? &boxes : nullptr;      // set: the node range of the instance tree is reserved and left empty
    if (!instancesOnly) { delete c->twoLevel; c->twoLevel = new BuiltTwoLevel(); }
    if (!(instancesOnly ? rebuild_two_level_instances(s, *c->twoLevel, berr, wantBoxes) : build_scene_two_level(s, *c->twoLevel, berr, wantBoxes))) {
        if (berr.find("singular") != std::string::npos) return kTwoLevelDoesNotFit;
        return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: " + berr);
    }
    lap("host records");
    BuiltTwoLevel& b = *c->twoLevel;
    if (two_level_stack_need(b) > 128u) return kTwoLevelDoesNotFit;
    if (!instancesOnly) {
        const HostTri* dt; const HostTriAttr* da; const HostTriTangent* dtg;
        if ((r = upload(c, b.tris.data(), b.tris.size(), &dt, &c->meshAllocations)) != HRPT_OK) return r;
        if ((r = upload(c, b.attrs.data(), b.attrs.size(), &da, &c->meshAllocations)) != HRPT_OK) return r;
        v.tris = reinterpret_cast<const GpuTri*>(dt); v.triCount = (uint32_t)b.tris.size(); v.attrs = reinterpret_cast<const GpuTriAttr*>(da);
        v.tangents = nullptr;
        if (!b.tangents.empty()) {
            if ((r = upload(c, b.tangents.data(), b.tangents.size(), &dtg, &c->meshAllocations)) != HRPT_OK) return r;
            v.tangents = reinterpret_cast<const GpuTriTangent*>(dtg);
        }
    }
    const HostNode4* dn4; const HostInstance* di; const HostInstShade* dis;
    if (b.nodes4.size() >= kMaxStructureNodes) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: more than 2^25 nodes (32-bit node offsets in the traversal kernels)");
    if (gpuTree) {      // the reserved range at the front is written on the device (launch_tlas_fixup): only the mesh trees behind it cross PCIe
        void* p = nullptr;
        HIP_TRY(c, hipMalloc(&p, b.nodes4.size() * sizeof(HostNode4)));
        c->bvhAllocations.push_back(p);
        dn4 = static_cast<const HostNode4*>(p);
        HIP_TRY(c, hipMemcpyAsync(static_cast<HostNode4*>(p) + b.tlasNodeCount, b.nodes4.data() + b.tlasNodeCount, (b.nodes4.size() - b.tlasNodeCount) * sizeof(HostNode4), hipMemcpyHostToDevice, c->stream));
    } else if ((r = upload(c, b.nodes4.data(), b.nodes4.size(), &dn4, &c->bvhAllocations)) != HRPT_OK) return r;
    if ((r = upload(c, b.instances.data(), b.instances.size(), &di, &c->bvhAllocations)) != HRPT_OK) return r;
    if ((r = upload(c, b.instShade.data(), b.instShade.size(), &dis, &c->bvhAllocations)) != HRPT_OK) return r;
    lap("uploads");
    if (gpuTree) {
        uint32_t levels = 0;
        if (build_instance_tree_on_gpu(c, s.instanceCount, boxes, instancesOnly, refit, const_cast<GpuNode4*>(reinterpret_cast<const GpuNode4*>(dn4)), levels)) {
            b.maxDepth4Tlas = levels;
        } else {
            // the host builds it after all: same layout rules as ever (the reserved node range shrinks to the tree's size)
            for (void* p : c->bvhAllocations) (void)hipFree(p);
            c->bvhAllocations.clear();
            if (!rebuild_two_level_instances(s, b, berr)) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: " + berr);
            if ((r = upload(c, b.nodes4.data(), b.nodes4.size(), &dn4, &c->bvhAllocations)) != HRPT_OK) return r;
            if ((r = upload(c, b.instances.data(), b.instances.size(), &di, &c->bvhAllocations)) != HRPT_OK) return r;
            if ((r = upload(c, b.instShade.data(), b.instShade.size(), &dis, &c->bvhAllocations)) != HRPT_OK) return r;
            gpuTree = false;
        }
    }
    lap("instance tree (GPU)");
    if (two_level_stack_need(b) > 128u) return kTwoLevelDoesNotFit;
    v.nodes = nullptr; v.nodeCount = b.tlasNodeCount; v.rootLeaf = b.tlasRootLeaf;      // nodeCount != 0: the walk starts at node4 0 (the instance tree)
    v.nodes4 = reinterpret_cast<const GpuNode4*>(dn4); v.node4Count = (uint32_t)b.nodes4.size(); v.nodesQ = nullptr;
    v.instances = reinterpret_cast<const GpuInstance*>(di); v.instanceCount = (uint32_t)b.instances.size();
    v.instShade = reinterpret_cast<const GpuInstShade*>(dis);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!gpuTree) c->buildInfo.usedBuilder = HRPT_BVH_BUILDER_HOST_SAH;      // (the mesh trees are the host's either way; usedBuilder names who built the tree over the instances)
    c->buildInfo.structure = HRPT_ACCEL_TWO_LEVEL;
    c->buildInfo.instanceNodeCount = b.tlasNodeCount; c->buildInfo.distinctMeshes = b.distinctMeshes;
    c->buildInfo.triangleCount = v.triCount; c->buildInfo.nodeCount = 0; c->buildInfo.node4Count = v.node4Count;
    c->buildInfo.maxDepth = 0; c->buildInfo.maxDepth4 = b.maxDepth4Tlas + b.maxDepth4Blas;
    c->bvhNodes = v.node4Count; c->bvhTris = v.triCount;
    c->traits.bvhMaxDepth = 0; c->traits.bvh4MaxDepth = b.maxDepth4Tlas + b.maxDepth4Blas; c->traits.twoLevelStackNeed = two_level_stack_need(b); c->traits.quantisedNodes = false;
    return HRPT_OK;
}

// refit (hrpt_refit_instances): where a GPU builder holds the hierarchy of the previous build, its boxes are recomputed instead of the tree rebuilt
static int build_acceleration(HrptContext* c, const HrptSceneDesc& s, uint64_t sceneTris, SceneView& v, bool firstBuild, bool refit = false)
{
    const auto t0 = std::chrono::steady_clock::now();
    std::string berr;
    int r;
    c->buildInfo = HrptBuildInfo{};
    c->buildInfo.requestedBuilder = (uint32_t)c->bvhBuilder;
    c->buildInfo.structure = HRPT_ACCEL_FLAT;
    const bool keepMeshTrees = !firstBuild && c->twoLevel != nullptr;       // hrpt_update_instances on a two-level scene
    free_acceleration(c, !firstBuild);
    if (keepMeshTrees || (firstBuild && two_level_wanted(c, s, sceneTris))) {
        r = build_two_level(c, s, v, keepMeshTrees, refit);
        if (r != kTwoLevelDoesNotFit) {
            c->buildInfo.buildMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return r;
        }
        // flat after all: drop everything of the two-level form (a moved instance may have become singular: hrpt_update_instances ends up here too)
        free_acceleration(c, false);
        c->buildInfo.structure = HRPT_ACCEL_FLAT;
        firstBuild = true;
    }
    v.instances = nullptr; v.instanceCount = 0; c->traits.twoLevelStackNeed = 0;
    if (sceneTris >= kMaxStructureTriangles) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: too many triangles for the flat structure (2^32 / 48 = 89 M world-space triangles; instanced scenes can use HRPT_ACCEL_TWO_LEVEL)");
    uint32_t maxDepth = 0, maxDepth4 = 0;
    bool built = false;
    const int builder = c->bvhBuilder == HRPT_BVH_BUILDER_AUTO ? (sceneTris >= 65536 ? HRPT_BVH_BUILDER_GPU_PLOC : HRPT_BVH_BUILDER_HOST_SAH) : c->bvhBuilder;
    if (builder != HRPT_BVH_BUILDER_GPU_LBVH && builder != HRPT_BVH_BUILDER_GPU_PLOC) { delete c->gpuBuilder; c->gpuBuilder = nullptr; }
    if ((builder == HRPT_BVH_BUILDER_GPU_LBVH || builder == HRPT_BVH_BUILDER_GPU_PLOC) && sceneTris >= 8) {
        // the whole build runs on the device; only the per-instance adjugate rows (O(instances)) are prepared on the host
        GpuBuiltBvh g; std::string gerr;
        hipError_t ge = hipSuccess;
        if (!c->gpuBuilder) {
            c->gpuBuilder = new GpuBvhBuilder();
            ge = c->gpuBuilder->prepare(s, scene_needs_tangents(s), c->stream, gerr);
        }
        if (ge == hipSuccess) ge = (refit && !firstBuild && c->gpuBuilder->can_refit()) ? c->gpuBuilder->refit(s.instances, c->stream, g, gerr)
                                                                                          : c->gpuBuilder->build(s.instances, builder == HRPT_BVH_BUILDER_GPU_PLOC, kTraversalStackDepth, c->stream, g, gerr);
        if (ge == hipSuccess && g.maxDepth + 2 <= kTraversalStackDepth) {
            v.nodes = g.nodes; v.nodeCount = g.nodeCount; v.nodes4 = g.nodes4; v.node4Count = g.node4Count; v.tris = g.tris; v.triCount = g.triCount;
            v.rootLeaf = 0; v.attrs = g.attrs; v.tangents = g.tangents;
            maxDepth = g.maxDepth; maxDepth4 = g.maxDepth4; built = true;
            if (getenv("HRPT_GPU_BVH_HOST_COLLAPSE")) {     // experiment: the GPU-built 2-wide tree with the host's area-greedy, depth-first 4-wide collapse
                std::vector<HostNode> n2(g.nodeCount); std::vector<HostNode4> n4; uint32_t d4 = 0;
                HIP_TRY(c, hipMemcpy(n2.data(), g.nodes, n2.size() * sizeof(HostNode), hipMemcpyDeviceToHost));
                collapse_bvh2_on_host(n2, n4, d4);
                const HostNode4* dn4;
                if ((r = upload(c, n4.data(), n4.size(), &dn4, &c->bvhAllocations)) != HRPT_OK) return r;
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                v.nodes4 = reinterpret_cast<const GpuNode4*>(dn4); v.node4Count = (uint32_t)n4.size(); maxDepth4 = d4;
            }
            c->buildInfo.usedBuilder = g.ploc ? HRPT_BVH_BUILDER_GPU_PLOC : HRPT_BVH_BUILDER_GPU_LBVH; c->buildInfo.deviceBuildMs = g.deviceMs; c->buildInfo.mortonBits = g.mortonBits; c->buildInfo.sahCost = g.sahCost;
            if (g.refitted) c->buildInfo.usedBuilder |= HRPT_BVH_BUILDER_REFITTED;
        } else {
            // too deep for the traversal stacks (or a device error): drop the device-side builder and build on the host instead
            delete c->gpuBuilder; c->gpuBuilder = nullptr;
            if (ge == hipErrorInvalidValue && gerr == "non-finite vertex position") return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: " + gerr);
            if (ge == hipErrorOutOfMemory) return fail(c, HRPT_ERR_OUT_OF_MEMORY, "acceleration structure: " + gerr);
        }
        if (built) {
            std::vector<HostInstShade> shade; build_instance_shade(s, shade);
            const HostInstShade* dis;
            if ((r = upload(c, shade.data(), shade.size(), &dis, &c->bvhAllocations)) != HRPT_OK) return r;
            v.instShade = reinterpret_cast<const GpuInstShade*>(dis);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    if (!built) {
        BuiltBvh bvh;
        if (!build_scene_bvh(s, bvh, berr)) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: " + berr);
        std::vector<void*>* own = &c->bvhAllocations;
        const HostNode* dn; const HostTri* dt;
        if ((r = upload(c, bvh.nodes.data(), bvh.nodes.size(), &dn, own)) != HRPT_OK) return r;
        if ((r = upload(c, bvh.tris.data(), bvh.tris.size(), &dt, own)) != HRPT_OK) return r;
        v.nodes = reinterpret_cast<const GpuNode*>(dn); v.nodeCount = (uint32_t)bvh.nodes.size();
        v.tris = reinterpret_cast<const GpuTri*>(dt); v.triCount = (uint32_t)bvh.tris.size();
        v.rootLeaf = bvh.rootLeaf;
        const HostNode4* dn4;
        if ((r = upload(c, bvh.nodes4.data(), bvh.nodes4.size(), &dn4, own)) != HRPT_OK) return r;
        v.nodes4 = reinterpret_cast<const GpuNode4*>(dn4); v.node4Count = (uint32_t)bvh.nodes4.size();
        // The quantised vertex / index / mesh / instance buffers are consumed here: per-triangle attribute records and
        // per-instance adjugate rows replace the per-hit GetTriangleVertices + UnpackVertex + MakeAdjugateMatrix work.
        const HostTriAttr* da; const HostTriTangent* dtg; const HostInstShade* dis;
        if ((r = upload(c, bvh.attrs.data(), bvh.attrs.size(), &da, own)) != HRPT_OK) return r;
        if ((r = upload(c, bvh.instShade.data(), bvh.instShade.size(), &dis, own)) != HRPT_OK) return r;
        v.attrs = reinterpret_cast<const GpuTriAttr*>(da); v.instShade = reinterpret_cast<const GpuInstShade*>(dis);
        v.tangents = nullptr;
        if (!bvh.tangents.empty()) {
            if ((r = upload(c, bvh.tangents.data(), bvh.tangents.size(), &dtg, own)) != HRPT_OK) return r;
            v.tangents = reinterpret_cast<const GpuTriTangent*>(dtg);
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // the BuiltBvh staging vectors die at scope exit
        maxDepth = bvh.maxDepth; maxDepth4 = bvh.maxDepth4;
        c->buildInfo.usedBuilder = HRPT_BVH_BUILDER_HOST_SAH; c->buildInfo.sahCost = bvh.sahCost;
    }
    if (v.node4Count >= kMaxStructureNodes) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "acceleration structure: more than 2^25 nodes (32-bit node offsets in the traversal kernels)");
    // the 64-byte quantised form of the 4-wide tree, whichever builder made it (what the wavefront kernels may read when the tree is not in LDS)
    v.nodesQ = nullptr; bool quantisedNodes = false;
    if (v.node4Count) {
        if (c->nodesQCapacity < v.node4Count) {
            if (c->nodesQ) (void)hipFree(c->nodesQ);
            c->nodesQ = nullptr; c->nodesQCapacity = 0;
            const size_t cap = (size_t)v.node4Count + v.node4Count / 8 + 64;
            if (hipMalloc((void**)&c->nodesQ, cap * sizeof(GpuNodeQ)) != hipSuccess) return fail(c, HRPT_ERR_OUT_OF_MEMORY, "acceleration structure: quantised nodes");
            c->nodesQCapacity = cap;
        }
        double* dArea = reinterpret_cast<double*>(c->nodesQ + (c->nodesQCapacity - 1));       // the last (spare) record of the buffer: two doubles
        HIP_TRY(c, hipMemsetAsync(dArea, 0, 4 * sizeof(double), c->stream));
        HIP_TRY(c, launch_quantise_nodes(v.nodes4, v.node4Count, c->nodesQ, dArea, c->stream));
        double area[4] = { 0.0, 0.0, 0.0, 0.0 };
        HIP_TRY(c, hipMemcpyAsync(area, dArea, sizeof area, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        v.nodesQ = c->nodesQ;
        // Which nodes the kernels walk when the tree is in global memory. The quantised form saves three of seven 16-byte requests per lane and
        // step and pays in decode arithmetic and in looser boxes; what the looser LEAF boxes cost is triangle tests (three requests + a
        // watertight test each). Measured (MI355X, 1080p): Sponza-class scene: wf_extend 6.66 -> 6.18 ms, frame 13.8 -> 13.3 ms; glass scene
        // (18 k triangles of tessellated glass bodies): shadow-ray triangle tests x 2.4, closest-hit leaf visits + 43 %, frame +2 %.
        const float inflation = area[0] > 0.0 ? (float)(area[1] / area[0]) : 1.0f;
        if (getenv("HRPT_BVH_NODE_FORMAT_DEBUG")) fprintf(stderr, "quantised nodes: leaf area ratio %.4f (area-weighted), %.4f (mean over %.0f leaves)\n", inflation, area[3] > 0 ? area[2] / area[3] : 1.0, area[3]);
        int format = 0;
        if (const char* e = getenv("HRPT_BVH_NODE_FORMAT")) format = atoi(e);          // 1: fp32 nodes, 2: quantised nodes, else by the leaf-area ratio
        // (the leaf-area ratio is ~1.01 on BOTH scenes, so it does not tell them apart: on the glass scene it is the paths that bounce inside and between
        // the finely tessellated glass bodies that visit 40 % more leaves through the rounded boxes. Until that is understood the rule is empirical:
        // quantised nodes unless some instance is transmissive or BLEND.)
        bool glassy = false;
        for (uint32_t i = 0; i < s.instanceCount && !glassy; ++i) {
            const HrptMaterialConstants& m = s.materials[s.instances[i].m_MaterialIndex];
            glassy = m.m_TransmissionFactor > 0.0f || m.m_AlphaMode == HRPT_ALPHA_MODE_BLEND;
        }
        quantisedNodes = format == 2 || (format != 1 && inflation <= 1.10f && !glassy);
        c->buildInfo.leafAreaPermille = (uint32_t)(inflation * 1000.0f + 0.5f); c->buildInfo.nodeFormat = quantisedNodes ? 2u : 1u;
    }
    c->buildInfo.buildMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->buildInfo.triangleCount = v.triCount; c->buildInfo.nodeCount = v.nodeCount; c->buildInfo.node4Count = v.node4Count;
    c->buildInfo.maxDepth = maxDepth; c->buildInfo.maxDepth4 = maxDepth4;
    c->bvhNodes = v.nodeCount; c->bvhTris = v.triCount;
    c->traits.bvhMaxDepth = maxDepth; c->traits.bvh4MaxDepth = maxDepth4; c->traits.quantisedNodes = quantisedNodes;
    return HRPT_OK;
}

// What the kernels specialise on (SceneTraits), from the library's copy of instances / materials / lights; the tree depths are kept.
static void refresh_traits(HrptContext* c)
{
    SceneTraits t; t.bvhMaxDepth = c->traits.bvhMaxDepth; t.bvh4MaxDepth = c->traits.bvh4MaxDepth; t.twoLevelStackNeed = c->traits.twoLevelStackNeed; t.quantisedNodes = c->traits.quantisedNodes;
    for (const HrptPerInstanceData& in : c->keptInstances) {
        const HrptMaterialConstants& m = c->keptMaterials[in.m_MaterialIndex];
        // the transmission branch (PathTracer.hlsl:149-255) is entered for transmissive AND for BLEND materials (effective transmission
        // 1 - alpha), and a thick one switches the path's medium state there: that state then has to travel with the path
        if ((m.m_TransmissionFactor > 0.0f || m.m_AlphaMode == HRPT_ALPHA_MODE_BLEND) && m.m_IsThinSurface == 0) t.hasMedium = true;
        if (m.m_AlphaMode == HRPT_ALPHA_MODE_BLEND && !(m.m_TransmissionFactor > 0.0f)) t.hasStochasticAlpha = true;
        if (m.m_TextureFlags != 0) t.hasTextures = true;
        if (m.m_AlphaMode != HRPT_ALPHA_MODE_OPAQUE) t.hasNonOpaque = true;
        if (m.m_TransmissionFactor > 0.0f || m.m_AlphaMode == HRPT_ALPHA_MODE_BLEND) t.hasTransmissiveOrBlend = true;
    }
    for (const HrptGPULight& l : c->keptLights) if (l.m_Type != HRPT_LIGHT_DIRECTIONAL) t.directionalLightsOnly = false;
    c->traits = t;
}

static int upload_scene_impl(HrptContext* c, const HrptSceneDesc* s)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!s) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: null scene");
    if (!s->brunetonTransmittance || !s->brunetonScattering) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: Bruneton LUTs missing");
    if (s->textureCount && !s->textures) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: null texture table");
    HIP_TRY(c, hipSetDevice(c->device));
    std::string berr;
    uint64_t sceneTris = 0;
    if (!validate_scene(*s, sceneTris, berr, false)) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: " + berr);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    c->traits = SceneTraits();
    SceneView v{};
    int r;
    const bool timing = getenv("HRPT_BUILD_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; (void)hipStreamSynchronize(c->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[upload]    %-22s %7.3f ms\n", what, std::chrono::duration<float, std::milli>(t - tp).count()); tp = t; };
    if ((r = build_acceleration(c, *s, sceneTris, v, true)) != HRPT_OK) return r;
    lap("acceleration structure");
    if ((r = upload(c, s->materials, s->materialCount, &v.materials)) != HRPT_OK) return r;
    if ((r = upload(c, s->lights, s->lightCount, &v.lights)) != HRPT_OK) return r;
    v.lightCount = s->lightCount; c->lightCapacity = s->lightCount;

    std::vector<GpuTexture> table(s->textureCount);
    for (uint32_t i = 0; i < s->textureCount; ++i) {
        const HrptTextureDesc& td = s->textures[i];
        GpuTexture& g = table[i];
        memset(&g, 0, sizeof g);
        g.w = td.width; g.h = td.height; g.format = td.format; g.mipCount = td.mipCount ? td.mipCount : 1u;
        if (!td.texels) continue;
        if (td.width == 0 || td.height == 0) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: zero-sized texture");
        if (td.format > HRPT_TEXTURE_FORMAT_RGBA32_FLOAT) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: unknown texture format");
        if (g.mipCount > HRPT_TEXTURE_MAX_MIPS) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: more than HRPT_TEXTURE_MAX_MIPS mip levels");
        uint64_t texels = 0;
        for (uint32_t l = 0; l < g.mipCount; ++l) {
            if (l > 0 && (td.width >> l) == 0 && (td.height >> l) == 0) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: more mip levels than the texture size allows");
            g.mipOffset[l] = (uint32_t)texels;
            texels += (uint64_t)((td.width >> l) ? (td.width >> l) : 1u) * ((td.height >> l) ? (td.height >> l) : 1u);
        }
        if (texels > 0xFFFFFFFFull) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_upload_scene: texture too large");
        const size_t bpt = td.format <= HRPT_TEXTURE_FORMAT_RGBA8_SRGB ? 4 : (td.format == HRPT_TEXTURE_FORMAT_RGBA16_FLOAT ? 8 : 16);
        const uint8_t* d;
        if ((r = upload(c, static_cast<const uint8_t*>(td.texels), (size_t)texels * bpt, &d)) != HRPT_OK) return r;
        g.texels = d;
    }
    if ((r = upload(c, table.data(), table.size(), &v.textures)) != HRPT_OK) return r;
    v.textureCount = s->textureCount;
    lap("materials, textures");

    // Bruneton LUTs: float32 file layout -> RGBA16F (src/CommonResources.cpp:534-558)
    const size_t nT = 256u * 64u * 4u, nS = 256u * 128u * 32u * 4u;
    std::vector<uint16_t> hT(nT), hS(nS);
    for (size_t i = 0; i < nT; ++i) hT[i] = float_to_half(s->brunetonTransmittance[i]);
    {   // 4 M conversions: 9 ms of every upload on one thread
        const float* src = s->brunetonScattering; uint16_t* dst = hS.data();
        const unsigned threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        const size_t chunk = (nS + threads - 1) / threads;
        auto part = [src, dst, nS, chunk](size_t t) { for (size_t i = t * chunk, e = std::min(nS, i + chunk); i < e; ++i) dst[i] = float_to_half(src[i]); };
        std::vector<std::thread> pool;
        size_t started = 1;
        try { for (; started < threads; ++started) pool.emplace_back(part, started); } catch (const std::system_error&) {}
        part(0);
        for (std::thread& th : pool) th.join();
        for (size_t t = started; t < threads; ++t) part(t);       // (threads that could not be started)
    }
    if ((r = upload(c, hT.data(), nT, &v.lutTransmittance)) != HRPT_OK) return r;
    if ((r = upload(c, hS.data(), nS, &v.lutScattering)) != HRPT_OK) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // host staging vectors die at scope exit
    lap("atmosphere tables");

    c->view = v; c->haveScene = true;
    c->keptVertices.assign(s->vertices, s->vertices + s->vertexCount); c->keptIndices.assign(s->indices, s->indices + s->indexCount);
    c->keptMeshData.assign(s->meshData, s->meshData + s->meshDataCount); c->keptInstances.assign(s->instances, s->instances + s->instanceCount);
    c->keptMaterials.assign(s->materials, s->materials + s->materialCount);
    c->keptLights.assign(s->lights, s->lights + s->lightCount);
    refresh_traits(c);
    lap("kept copies, traits");
    return HRPT_OK;
}

// The scene description the rebuild paths hand to the builders, over the library's copies.
static HrptSceneDesc kept_scene_desc(HrptContext* c)
{
    HrptSceneDesc s{};
    s.vertices = c->keptVertices.data(); s.vertexCount = (uint32_t)c->keptVertices.size();
    s.indices = c->keptIndices.data(); s.indexCount = (uint32_t)c->keptIndices.size();
    s.meshData = c->keptMeshData.data(); s.meshDataCount = (uint32_t)c->keptMeshData.size();
    s.instances = c->keptInstances.data(); s.instanceCount = (uint32_t)c->keptInstances.size();
    s.materials = c->keptMaterials.data(); s.materialCount = (uint32_t)c->keptMaterials.size();
    static const HrptGPULight noLight{};                 // lights play no part in the build; validate_scene only wants the array to exist
    s.lights = &noLight; s.lightCount = 1;
    return s;
}
static uint64_t kept_triangle_count(const HrptContext* c)
{
    uint64_t n = 0;
    for (const HrptPerInstanceData& in : c->keptInstances) n += c->keptMeshData[in.m_MeshDataIndex].m_IndexCounts[0] / 3;
    return n;
}

static int update_lights_impl(HrptContext* c, const HrptGPULight* lights, uint32_t count)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!c->haveScene) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_lights: no scene uploaded");
    if (!lights || count == 0) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_lights: a scene needs at least one light (the reference guarantees a directional light, src/Scene.cpp:635-666)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));        // frames in flight still read the old buffer
    if (count > c->lightCapacity) {
        const HrptGPULight* d; int r;
        if ((r = upload(c, lights, count, &d)) != HRPT_OK) return r;      // the old, smaller buffer stays in the scene's allocation list
        c->view.lights = d; c->lightCapacity = count;
    } else {
        HIP_TRY(c, hipMemcpyAsync(const_cast<HrptGPULight*>(c->view.lights), lights, (size_t)count * sizeof(HrptGPULight), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->view.lightCount = count;
    c->keptLights.assign(lights, lights + count);
    refresh_traits(c);
    return HRPT_OK;
}

static int update_materials_impl(HrptContext* c, const HrptMaterialConstants* materials, uint32_t firstMaterial, uint32_t count)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!c->haveScene) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_materials: no scene uploaded");
    if (count == 0) return HRPT_OK;
    if (!materials) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_materials: null material array");
    if ((uint64_t)firstMaterial + count > c->keptMaterials.size()) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_materials: range exceeds the scene's material count");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // The acceleration structure caches two things of a material: whether its triangles are opaque (any-hit / candidate handling) and
    // whether any material needs tangent frames. A change of either needs a rebuild; everything else is a plain buffer write.
    HrptSceneDesc before = kept_scene_desc(c);
    const bool tangentsBefore = scene_needs_tangents(before);
    bool structural = false;
    for (uint32_t i = 0; i < count; ++i)
        if (triangle_flags_for_material(materials[i]) != triangle_flags_for_material(c->keptMaterials[firstMaterial + i])) structural = true;   // opacity or shading class
    std::memcpy(c->keptMaterials.data() + firstMaterial, materials, (size_t)count * sizeof(HrptMaterialConstants));
    HIP_TRY(c, hipMemcpyAsync(const_cast<HrptMaterialConstants*>(c->view.materials) + firstMaterial, materials, (size_t)count * sizeof(HrptMaterialConstants), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HrptSceneDesc s = kept_scene_desc(c);
    if (scene_needs_tangents(s) != tangentsBefore) structural = true;
    if (structural) {
        SceneView v = c->view;
        int r = build_acceleration(c, s, kept_triangle_count(c), v, true);     // from scratch: the GPU builder's resident instance table holds the opacity flags
        if (r != HRPT_OK) { c->haveScene = false; return r; }
        c->view = v;
    }
    refresh_traits(c);
    return HRPT_OK;
}

static int update_instances_impl(HrptContext* c, const HrptPerInstanceData* instances, uint32_t firstInstance, uint32_t count, bool refit)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!c->haveScene) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_instances: no scene uploaded");
    if (count == 0) return HRPT_OK;
    if (!instances) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_instances: null instance array");
    if ((uint64_t)firstInstance + count > c->keptInstances.size()) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_instances: range exceeds the scene's instance count");
    for (uint32_t i = 0; i < count; ++i) {
        const HrptPerInstanceData& now = instances[i]; const HrptPerInstanceData& was = c->keptInstances[firstInstance + i];
        if (now.m_MeshDataIndex != was.m_MeshDataIndex || now.m_MaterialIndex != was.m_MaterialIndex || now.m_LODIndex != was.m_LODIndex)
            return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_update_instances: mesh, material and LOD of an instance cannot change (upload the scene again)");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));        // frames in flight still traverse the old tree
    std::memcpy(c->keptInstances.data() + firstInstance, instances, (size_t)count * sizeof(HrptPerInstanceData));
    HrptSceneDesc s = kept_scene_desc(c);
    const uint64_t sceneTris = kept_triangle_count(c);
    SceneView v = c->view;
    int r = build_acceleration(c, s, sceneTris, v, false, refit);
    if (r != HRPT_OK) { c->haveScene = false; return r; }   // the old tree is gone: the scene has to be uploaded again
    c->view = v;
    return HRPT_OK;
}

int hrpt_resize(HrptContext* c, uint32_t width, uint32_t height)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (width == 0 || height == 0 || width > 65535u || height > 65535u)
        return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resize: size must be 1..65535 (RNG seed packs y*65536+x, RNG.hlsli:24)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->dAccum) { (void)hipFree(c->dAccum); c->dAccum = nullptr; }
    if (c->dOutput) { (void)hipFree(c->dOutput); c->dOutput = nullptr; }
    if (c->dDisplay) { (void)hipFree(c->dDisplay); c->dDisplay = nullptr; }
    size_t bytes = (size_t)width * height * sizeof(float4);
    HIP_TRY(c, hipMalloc((void**)&c->dAccum, bytes));
    HIP_TRY(c, hipMalloc((void**)&c->dOutput, bytes));
    HIP_TRY(c, hipMemsetAsync(c->dAccum, 0, bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dOutput, 0, bytes, c->stream));
    c->width = width; c->height = height;
    return HRPT_OK;
}

float hrpt_halton(uint32_t index, uint32_t base)   // src/Utilities.cpp:67-79
{
    float result = 0.0f;
    float f = 1.0f / (float)base;
    uint32_t i = index;
    while (i > 0) {
        result += f * (float)(i % base);
        i /= base;
        f /= (float)base;
    }
    return result;
}

static int render_impl(HrptContext* c, const HrptFrameParams* p)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!p) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: null params");
    if (!c->haveScene) return fail(c, HRPT_ERR_NO_SCENE, "hrpt_render: no scene uploaded");
    if (!c->dAccum) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: hrpt_resize not called");
    if (p->accumCount == 0) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: accumCount == 0");
    if (p->constants.m_MaxBounces > 64u) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: m_MaxBounces above 64 (the reference's UI stops at 12, src/ImGuiLayer.cpp:760; one kernel sequence is launched per bounce)");
    if (p->constants.m_LightCount > c->view.lightCount) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: m_LightCount exceeds the scene's light buffer");
    uint32_t vw = (uint32_t)p->constants.m_View.m_ViewportSize[0], vh = (uint32_t)p->constants.m_View.m_ViewportSize[1];
    if (vw != c->width || vh != c->height) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: m_ViewportSize does not match hrpt_resize");
    TileRect rect; rect.x0 = p->tileX0; rect.y0 = p->tileY0; rect.x1 = p->tileX1; rect.y1 = p->tileY1;
    rect.stripeCount = p->stripeCount ? p->stripeCount : 1u; rect.stripeIndex = p->stripeIndex;
    if (rect.stripeIndex >= rect.stripeCount) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: stripeIndex must be below stripeCount");
    if (rect.x0 == 0 && rect.y0 == 0 && rect.x1 == 0 && rect.y1 == 0) { rect.x1 = c->width; rect.y1 = c->height; }
    if (rect.x1 > c->width || rect.y1 > c->height || rect.x0 > rect.x1 || rect.y0 > rect.y1)
        return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: tile rectangle outside the image");
    HIP_TRY(c, hipSetDevice(c->device));

    bool wavefront = (p->flags & HRPT_FRAME_MEGAKERNEL) == 0 && wavefront_supports(c->view, p->constants);
    if (!wavefront && c->view.instances && c->traits.twoLevelStackNeed > 64u)
        return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_render: this two-level structure is deeper than the validation megakernel's 64-entry stack");
    if (!wavefront && (p->flags & HRPT_FRAME_MEGAKERNEL) == 0) c->megakernelFallbacks++;
    HIP_TRY(c, hipEventRecord(c->evStart, c->stream));
    if (wavefront) {
        std::string werr;
        c->wf.profile = (p->flags & HRPT_FRAME_PROFILE) != 0;
        hipError_t e = wavefront_render(c->wf, c->view, c->traits, p->constants, p->accumCount, c->dAccum, c->dOutput, c->width, c->height, rect,
                                        c->dCounters, c->stream, werr);
        if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? HRPT_ERR_OUT_OF_MEMORY : HRPT_ERR_HIP, "wavefront_render: " + werr + ": " + hipGetErrorString(e));
    } else {
        for (uint32_t k = 0; k < p->accumCount; ++k) {
            HrptPathTracerConstants cb = p->constants;
            cb.m_AccumulationIndex = p->constants.m_AccumulationIndex + k;                 // PathTracerRenderer.cpp:62,:105
            cb.m_Jitter[0] = hrpt_halton(cb.m_AccumulationIndex + 1, 2) - 0.5f;            // :65
            cb.m_Jitter[1] = hrpt_halton(cb.m_AccumulationIndex + 1, 3) - 0.5f;
            HIP_TRY(c, launch_megakernel(c->view, cb, c->dAccum, c->dOutput, c->width, rect, c->dCounters, c->stream));
        }
    }
    HIP_TRY(c, hipEventRecord(c->evStop, c->stream));
    c->timed = true;
    return HRPT_OK;
}

// No C++ exception crosses the C boundary: host-side allocation failures (std::bad_alloc on very large scenes) become status codes.
int hrpt_upload_scene(HrptContext* c, const HrptSceneDesc* s)
{
    try { return upload_scene_impl(c, s); }
    catch (const std::bad_alloc&) { return fail(c, HRPT_ERR_OUT_OF_MEMORY, "hrpt_upload_scene: host allocation failed"); }
    catch (const std::exception& e) { return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string("hrpt_upload_scene: ") + e.what()); }
}
int hrpt_refit_instances(HrptContext* c, const HrptPerInstanceData* instances, uint32_t firstInstance, uint32_t count)
{
    try { return update_instances_impl(c, instances, firstInstance, count, true); }
    catch (const std::bad_alloc&) { return fail(c, HRPT_ERR_OUT_OF_MEMORY, "hrpt_refit_instances: host allocation failed"); }
    catch (const std::exception& e) { return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string("hrpt_refit_instances: ") + e.what()); }
}
int hrpt_update_instances(HrptContext* c, const HrptPerInstanceData* instances, uint32_t firstInstance, uint32_t count)
{
    try { return update_instances_impl(c, instances, firstInstance, count, false); }
    catch (const std::bad_alloc&) { return fail(c, HRPT_ERR_OUT_OF_MEMORY, "hrpt_update_instances: host allocation failed"); }
    catch (const std::exception& e) { return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string("hrpt_update_instances: ") + e.what()); }
}
int hrpt_update_lights(HrptContext* c, const HrptGPULight* lights, uint32_t count)
{
    try { return update_lights_impl(c, lights, count); }
    catch (const std::bad_alloc&) { return fail(c, HRPT_ERR_OUT_OF_MEMORY, "hrpt_update_lights: host allocation failed"); }
    catch (const std::exception& e) { return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string("hrpt_update_lights: ") + e.what()); }
}
int hrpt_update_materials(HrptContext* c, const HrptMaterialConstants* materials, uint32_t firstMaterial, uint32_t count)
{
    try { return update_materials_impl(c, materials, firstMaterial, count); }
    catch (const std::bad_alloc&) { return fail(c, HRPT_ERR_OUT_OF_MEMORY, "hrpt_update_materials: host allocation failed"); }
    catch (const std::exception& e) { return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string("hrpt_update_materials: ") + e.what()); }
}
int hrpt_render(HrptContext* c, const HrptFrameParams* p)
{
    try { return render_impl(c, p); }
    catch (const std::bad_alloc&) { return fail(c, HRPT_ERR_OUT_OF_MEMORY, "hrpt_render: host allocation failed"); }
    catch (const std::exception& e) { return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string("hrpt_render: ") + e.what()); }
}

int hrpt_set_stream(HrptContext* c, void* hipStream, int useCallerStream)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = useCallerStream ? static_cast<hipStream_t>(hipStream) : c->ownStream;   // a NULL caller stream is the legacy default stream
    return HRPT_OK;
}

int hrpt_synchronize(HrptContext* c)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HRPT_OK;
}

int hrpt_get_device_images(HrptContext* c, void** accumulation, void** output)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!c->dAccum) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_get_device_images: hrpt_resize not called");
    if (accumulation) *accumulation = c->dAccum;
    if (output) *output = c->dOutput;
    return HRPT_OK;
}

static int read_image(HrptContext* c, const float4* src, float* dst, size_t bytes, const char* what)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!dst || !src || bytes != (size_t)c->width * c->height * 16) return fail(c, HRPT_ERR_INVALID_ARGUMENT, std::string(what) + ": bad buffer size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HRPT_OK;
}
int hrpt_read_accumulation(HrptContext* c, float* rgba, size_t bytes) { return read_image(c, c ? c->dAccum : nullptr, rgba, bytes, "hrpt_read_accumulation"); }
int hrpt_read_output(HrptContext* c, float* rgba, size_t bytes) { return read_image(c, c ? c->dOutput : nullptr, rgba, bytes, "hrpt_read_output"); }

int hrpt_write_accumulation(HrptContext* c, const float* rgba, size_t bytes)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!rgba || !c->dAccum || bytes != (size_t)c->width * c->height * 16) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_write_accumulation: bad buffer size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->dAccum, rgba, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HRPT_OK;
}

int hrpt_resolve_output(HrptContext* c)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!c->dAccum) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resolve_output: hrpt_resize not called");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_resolve(c->dAccum, c->dOutput, c->width * c->height, c->stream));
    return HRPT_OK;
}

int hrpt_resolve_device(HrptContext* c, const float* accumulationDevice, float* outputDevice, uint64_t pixelCount, void* stream)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!accumulationDevice || !outputDevice) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resolve_device: null image");
    if (pixelCount == 0) return HRPT_OK;
    if (pixelCount > 0xFFFFFFFFull) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resolve_device: image too large");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_resolve(reinterpret_cast<const float4*>(accumulationDevice), reinterpret_cast<float4*>(outputDevice), (uint32_t)pixelCount,
                              static_cast<hipStream_t>(stream)));
    return HRPT_OK;
}

int hrpt_allgather(HrptContext* const* ranks, int n)
{
    if (!ranks || n <= 0) return HRPT_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i) if (!ranks[i]) return HRPT_ERR_INVALID_ARGUMENT;
    HrptContext* c0 = ranks[0];
    const uint32_t W = c0->width, H = c0->height;
    if (!c0->dAccum || W == 0 || H == 0) return fail(c0, HRPT_ERR_INVALID_ARGUMENT, "hrpt_allgather: hrpt_resize not called");
    if (H % (uint32_t)n != 0) return fail(c0, HRPT_ERR_INVALID_ARGUMENT, "hrpt_allgather: image height must be a multiple of the number of ranks");
    for (int i = 0; i < n; ++i) {
        if (ranks[i]->width != W || ranks[i]->height != H || !ranks[i]->dAccum) return fail(c0, HRPT_ERR_INVALID_ARGUMENT, "hrpt_allgather: contexts differ in image size");
        for (int j = 0; j < i; ++j) if (ranks[j] == ranks[i]) return fail(c0, HRPT_ERR_INVALID_ARGUMENT, "hrpt_allgather: the same context appears twice");
    }
    const size_t rows = H / (uint32_t)n, bandBytes = rows * (size_t)W * sizeof(float4);
    std::vector<hipEvent_t> sent((size_t)n, nullptr);
    auto cleanup = [&]() { for (hipEvent_t e : sent) if (e) (void)hipEventDestroy(e); };
    // every rank pushes its band to all the others on its own stream, then marks the point where its sends are enqueued
    for (int i = 0; i < n; ++i) {
        HrptContext* src = ranks[i];
        hipError_t e = hipSetDevice(src->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sent[(size_t)i], hipEventDisableTiming);
        const size_t off = (size_t)i * rows * W;
        for (int j = 0; j < n && e == hipSuccess; ++j) {
            if (j == i) continue;
            HrptContext* dst = ranks[j];
            // the destination band must not be in use by the destination's earlier work (e.g. its previous resolve): order behind it
            hipEvent_t ready = nullptr;
            e = hipSetDevice(dst->device);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ready, dst->stream);
            if (e == hipSuccess) e = hipSetDevice(src->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(src->stream, ready, 0);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(dst->dAccum + off, dst->device, src->dAccum + off, src->device, bandBytes, src->stream);
            if (ready) (void)hipEventDestroy(ready);
        }
        if (e == hipSuccess) e = hipEventRecord(sent[(size_t)i], src->stream);
        if (e != hipSuccess) { cleanup(); return fail(c0, HRPT_ERR_HIP, std::string("hrpt_allgather (send): ") + hipGetErrorString(e)); }
    }
    // every rank waits for all senders, then resolves its now complete image
    for (int j = 0; j < n; ++j) {
        HrptContext* dst = ranks[j];
        hipError_t e = hipSetDevice(dst->device);
        for (int i = 0; i < n && e == hipSuccess; ++i) if (i != j) e = hipStreamWaitEvent(dst->stream, sent[(size_t)i], 0);
        if (e == hipSuccess) e = launch_resolve(dst->dAccum, dst->dOutput, W * H, dst->stream);
        if (e != hipSuccess) { cleanup(); return fail(c0, HRPT_ERR_HIP, std::string("hrpt_allgather (receive): ") + hipGetErrorString(e)); }
    }
    cleanup();      // destroying a recorded event is deferred by the runtime until the waits that reference it have run
    return HRPT_OK;
}

int hrpt_trace_rays(HrptContext* c, const HrptRay* rays, HrptRayHit* hits, uint64_t count, uint32_t flags)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!c->haveScene) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_trace_rays: no scene uploaded");
    if (count == 0) return HRPT_OK;
    if (!rays || !hits) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_trace_rays: null array");
    if ((flags & 0xFFu) > HRPT_RAYS_SHADOW || (flags & ~(0xFFu | HRPT_RAYS_DEVICE_POINTERS | HRPT_RAYS_THREAD_PER_RAY))) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_trace_rays: unknown flags");
    if (count > (1ull << 31)) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_trace_rays: too many rays in one call");
    if (c->view.instances && ((flags & HRPT_RAYS_THREAD_PER_RAY) || !wavefront_trace_rays_supported(c->traits)) && c->traits.twoLevelStackNeed > 64u)
        return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_trace_rays: this two-level structure is deeper than the thread-per-ray kernel's 64-entry stack");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool shadow = (flags & 0xFFu) == HRPT_RAYS_SHADOW;
    // the persistent refilling traversal kernel (pt_wavefront.hip wf_trace_rays); the thread-per-ray kernel stays as the fallback for trees
    // deeper than its stacks allow and as the cross-check (HRPT_RAYS_THREAD_PER_RAY)
    const bool persistent = !(flags & HRPT_RAYS_THREAD_PER_RAY) && wavefront_trace_rays_supported(c->traits) && c->view.node4Count > 0;
    auto trace = [&](const HrptRay* dr, HrptRayHit* dh) -> hipError_t {
        if (!persistent) return launch_trace_rays(c->view, dr, dh, count, shadow, c->stream);
        std::string werr;
        hipError_t te = wavefront_trace_rays(c->wf, c->view, c->traits, dr, dh, count, shadow, c->stream, werr);
        if (te != hipSuccess) c->err = "hrpt_trace_rays: " + werr;
        return te;
    };
    if (flags & HRPT_RAYS_DEVICE_POINTERS) {
        HIP_TRY(c, trace(rays, hits));
        return HRPT_OK;
    }
    HrptRay* dRays = nullptr; HrptRayHit* dHits = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dRays), count * sizeof(HrptRay));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dHits), count * sizeof(HrptRayHit));
    if (e == hipSuccess) e = hipMemcpyAsync(dRays, rays, count * sizeof(HrptRay), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = trace(dRays, dHits);
    if (e == hipSuccess) e = hipMemcpyAsync(hits, dHits, count * sizeof(HrptRayHit), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (dRays) (void)hipFree(dRays);
    if (dHits) (void)hipFree(dHits);
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? HRPT_ERR_OUT_OF_MEMORY : HRPT_ERR_HIP, std::string("hrpt_trace_rays: ") + hipGetErrorString(e));
    return HRPT_OK;
}

int hrpt_resolve_columns_device(HrptContext* c, const float* shardsDevice, float* accumulationDevice, float* outputDevice, uint32_t width, uint32_t height, uint32_t ranks, void* stream)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!shardsDevice || !outputDevice) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resolve_columns_device: null image");
    if (ranks == 0 || width == 0 || height == 0 || width % (8u * ranks) != 0) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resolve_columns_device: width must be a positive multiple of 8 * ranks");
    if ((uint64_t)width * height > 0xFFFFFFFFull) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_resolve_columns_device: image too large");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_resolve_columns(reinterpret_cast<const float4*>(shardsDevice), reinterpret_cast<float4*>(accumulationDevice), reinterpret_cast<float4*>(outputDevice),
                                      width, height, ranks, static_cast<hipStream_t>(stream)));
    return HRPT_OK;
}

int hrpt_set_shadow_overlap(HrptContext* c, int enabled)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    c->wf.serialShadow = enabled == 0;
    return HRPT_OK;
}

int hrpt_set_acceleration_structure(HrptContext* c, int structure)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (structure < HRPT_ACCEL_AUTO || structure > HRPT_ACCEL_TWO_LEVEL) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_set_acceleration_structure: unknown structure");
    c->accelStructure = structure;
    return HRPT_OK;
}

int hrpt_set_bvh_builder(HrptContext* c, int builder)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (builder != HRPT_BVH_BUILDER_HOST_SAH && builder != HRPT_BVH_BUILDER_GPU_LBVH && builder != HRPT_BVH_BUILDER_GPU_PLOC && builder != HRPT_BVH_BUILDER_AUTO) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_set_bvh_builder: unknown builder");
    c->bvhBuilder = builder;
    return HRPT_OK;
}

int hrpt_get_build_info(HrptContext* c, HrptBuildInfo* out)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!out) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_get_build_info: null out");
    if (!c->haveScene) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_get_build_info: no scene uploaded");
    *out = c->buildInfo;
    return HRPT_OK;
}

int hrpt_get_stats(HrptContext* c, HrptStats* out)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!out) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_get_stats: null out");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    DeviceCounters h[kCounterShards];
    HIP_TRY(c, hipMemcpy(h, c->dCounters, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    DeviceCounters total{};
    for (int i = 0; i < kCounterShards; ++i) {
        total.closestRays += h[i].closestRays; total.shadowRays += h[i].shadowRays; total.paths += h[i].paths; total.neeEntries += h[i].neeEntries;
        total.neeSamples += h[i].neeSamples; total.radianceShade += h[i].radianceShade; total.radianceShadow += h[i].radianceShadow; total.skipped16 += h[i].skipped16;
    }
    out->closestRays = total.closestRays; out->shadowRays = total.shadowRays; out->paths = total.paths;
    out->neeEntries = total.neeEntries; out->neeSamples = total.neeSamples;
    out->megakernelFallbacks = c->megakernelFallbacks; out->queuePoolBytes = c->wf.poolBytes;
    if (total.neeEntries || c->wf.raygenBytes || c->wf.resolveBytes) {     // the wavefront pipeline ran since the last reset
        wavefront_queue_bytes(c->wf, total, out->traceQueueBytes, out->shadeQueueBytes, out->shadowQueueBytes);
        out->raygenQueueBytes = c->wf.raygenBytes; out->resolveQueueBytes = c->wf.resolveBytes;
    }
    if (c->timed) { float ms = 0.0f; if (hipEventElapsedTime(&ms, c->evStart, c->evStop) == hipSuccess) out->lastRenderMs = ms; }
    wavefront_collect_timing(c->wf);
    out->traceKernelMs = c->wf.kernelMs[0]; out->traceKernelLaunches = c->wf.kernelLaunches[0];
    out->shadeKernelMs = c->wf.kernelMs[1]; out->shadeKernelLaunches = c->wf.kernelLaunches[1];
    out->shadowKernelMs = c->wf.kernelMs[2]; out->shadowKernelLaunches = c->wf.kernelLaunches[2];
    out->raygenKernelMs = c->wf.kernelMs[3]; out->raygenKernelLaunches = c->wf.kernelLaunches[3];
    out->resolveKernelMs = c->wf.kernelMs[4]; out->resolveKernelLaunches = c->wf.kernelLaunches[4];
    out->bvhNodeCount = c->bvhNodes; out->bvhTriangleCount = c->bvhTris; out->bvhMaxDepth = c->traits.bvhMaxDepth;
    return HRPT_OK;
}

int hrpt_post_process(HrptContext* c, const HrptPostParams* p)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!p) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_post_process: null params");
    if (!c->dOutput) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_post_process: hrpt_resize not called");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->dExposure) {
        HIP_TRY(c, hipMalloc((void**)&c->dExposure, 16));
        HIP_TRY(c, hipMalloc((void**)&c->dHistogram, 256 * sizeof(uint32_t)));
        const float one[4] = { 1.0f, 0.0f, 0.0f, 0.0f };
        HIP_TRY(c, hipMemcpy(c->dExposure, one, 16, hipMemcpyHostToDevice));
    }
    if (!c->dDisplay) HIP_TRY(c, hipMalloc((void**)&c->dDisplay, (size_t)c->width * c->height * sizeof(float4)));
    HIP_TRY(c, launch_post_chain(c->dOutput, c->dDisplay, c->width * c->height, *p, c->dHistogram, c->dExposure, c->stream));
    return HRPT_OK;
}

int hrpt_read_display(HrptContext* c, float* rgba, size_t bytes) { return read_image(c, c ? c->dDisplay : nullptr, rgba, bytes, "hrpt_read_display"); }

int hrpt_get_exposure(HrptContext* c, float* exposure, uint32_t histogram256[256])
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!exposure || !c->dExposure) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_get_exposure: no post pass has run");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(exposure, c->dExposure, sizeof(float), hipMemcpyDeviceToHost));
    if (histogram256) HIP_TRY(c, hipMemcpy(histogram256, c->dHistogram, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HRPT_OK;
}

int hrpt_set_exposure(HrptContext* c, float exposure)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->dExposure) {
        HIP_TRY(c, hipMalloc((void**)&c->dExposure, 16));
        HIP_TRY(c, hipMalloc((void**)&c->dHistogram, 256 * sizeof(uint32_t)));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->dExposure, &exposure, sizeof(float), hipMemcpyHostToDevice));
    return HRPT_OK;
}

int hrpt_selftest_f16_decode(HrptContext* c, float* out65536)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!out65536) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_selftest_f16_decode: null out");
    HIP_TRY(c, hipSetDevice(c->device));
    float* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, 65536 * sizeof(float)));
    hipError_t e = launch_f16_table(d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out65536, d, 65536 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, HRPT_ERR_HIP, std::string("hrpt_selftest_f16_decode: ") + hipGetErrorString(e));
    return HRPT_OK;
}

int hrpt_selftest_bvh(HrptContext* c, uint64_t* violations)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!violations) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_selftest_bvh: null out");
    if (!c->haveScene) return fail(c, HRPT_ERR_NO_SCENE, "hrpt_selftest_bvh: no scene uploaded");
    if (c->view.instances) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_selftest_bvh: not available on the two-level structure (hrpt_set_acceleration_structure)");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_bvh_check(c->view, d, c->stream);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, HRPT_ERR_HIP, std::string("hrpt_selftest_bvh: ") + hipGetErrorString(e));
    *violations = h;
    return HRPT_OK;
}

int hrpt_selftest_unorm8(HrptContext* c, float* out512)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    if (!out512) return fail(c, HRPT_ERR_INVALID_ARGUMENT, "hrpt_selftest_unorm8: null out");
    HIP_TRY(c, hipSetDevice(c->device));
    float* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, 512 * sizeof(float)));
    hipError_t e = launch_unorm8_table(d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out512, d, 512 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, HRPT_ERR_HIP, std::string("hrpt_selftest_unorm8: ") + hipGetErrorString(e));
    return HRPT_OK;
}

int hrpt_reset_stats(HrptContext* c)
{
    if (!c) return HRPT_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dCounters, 0, sizeof(DeviceCounters) * kCounterShards, c->stream));
    wavefront_reset_timing(c->wf);
    c->megakernelFallbacks = 0;
    return HRPT_OK;
}

} // extern "C"
