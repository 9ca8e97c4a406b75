// pt_wavefront.h -- host interface of the persistent wavefront pipeline (pt_wavefront.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/hobbyrt_pt.h"
#include "pt_kernels.h"

namespace hrt {

struct SceneView;

// Facts about the uploaded scene that select kernel variants / optional state streams.
struct SceneTraits {
    bool hasMedium = false;            // a thick (non-thin) transmissive material exists: interior IOR/sigma travel with the path
    bool hasStochasticAlpha = false;   // a non-transmissive BLEND material exists: TraceRayStandard draws RNG (RaytracingCommon.hlsli:181)
    bool hasTextures = false;          // some instanced material has m_TextureFlags != 0
    bool hasTransmissiveOrBlend = false;   // some instanced material takes the transmission branch (PathTracer.hlsl:149)
    bool directionalLightsOnly = true; // every GPULight is type 0
    bool hasNonOpaque = false;         // some instance is ForceNonOpaque (material alpha mode MASK or BLEND)
    uint32_t bvhMaxDepth = 0;
    uint32_t bvh4MaxDepth = 0;
    bool quantisedNodes = false;       // trees in global memory are walked through SceneView::nodesQ (64-byte nodes): chosen per scene at build time (pt_capi.cpp)
    uint32_t twoLevelStackNeed = 0;    // != 0: the scene holds the two-level structure (SceneView::instances); worst-case traversal stack entries
};

struct WavefrontState {
    void* pool = nullptr;              // one device allocation carved into the SoA queues
    size_t poolBytes = 0;
    void* spill = nullptr;             // traversal-stack overflow columns (only for trees deeper than the LDS stack)
    size_t spillBytes = 0;
    void* traceSpill = nullptr;        // the same for hrpt_trace_rays batches
    size_t traceSpillBytes = 0;
    // start/stop event pairs around every extend / shade / shadow launch; kind[i] = 0,1,2
    std::vector<hipEvent_t> events;
    std::vector<uint8_t> kind;
    uint32_t eventsUsed = 0;
    // kernel classes: 0 extend, 1 shade, 2 shadow stage, 3 raygen, 4 resolve
    float kernelMs[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };      // summed device time per kernel class since the last reset
    uint32_t kernelLaunches[5] = { 0, 0, 0, 0, 0 };
    // queue-byte accounting (HrptStats::*QueueBytes): what the host knows per render is summed here, the rest follows from the device
    // counters and the record layout of the last render
    uint64_t raygenBytes = 0, resolveBytes = 0;
    struct Layout { uint32_t pathRecordBytes = 48, maxLights = 1; int shadowMode = 0; bool fusedPrimary = false; } layout;
    // tuning knobs (0 = default)
    uint64_t maxSamplesPerBatch = 0;
    uint32_t blocksPerCu = 0;
    uint32_t extendBlocksPerCu = 0;    // HRPT_WF_EXTEND_BLOCKS_PER_CU: grid of wf_extend alone (0 = automatic: 12 / 6 per CU for a tree in LDS / global memory)
    bool profile = false;              // record HIP events around every extend / shade / shadow launch (HRPT_FRAME_PROFILE)
    uint32_t refillMin = 0;            // wf_extend lane-refill threshold (1..64); 0 = default
    uint32_t segmentShift = 0;         // log2 of the segment size (6..10); 0 = automatic
    uint32_t segmentSize = 0;          // a segment size that is not a power of two (64..1024); 0 = from segmentShift
    bool forceGlobalBvh = false;
    bool drainSegments = false;        // wf_extend finishes every ray of a segment before it opens the next one (A/B knob)
    bool serialShadow = false;         // true: wf_shadow runs in stream order instead of concurrently with the next wf_extend
    // second stream + fork/join events: wf_shadow(b) overlaps wf_extend(b+1) (they share no buffer)
    hipStream_t auxStream = nullptr;
    std::vector<hipEvent_t> forkEvents, joinEvents;
    uint32_t padLdsBytes = 0;          // experiment: extra dynamic LDS per trace block (lowers occupancy)
    uint32_t bvhWidth = 0;             // 2 or 4: node width the trace kernels traverse; 0 = default
    bool forceGeneralShade = false;
    uint32_t nodeLoopMin = ~0u;        // HRPT_WF_NODE_LOOP_MIN: the descent loops end when fewer lanes than this are at inner nodes (~0 = automatic, 0 = never)
    bool noFusedPrimary = false;       // HRPT_WF_FUSED_PRIMARY=0: SIMPLE scenes keep the wf_raygen pass (A/B knob)
    bool noSlimShadow = false;         // HRPT_WF_SLIM_SHADOW=0: the SIMPLE shade variant writes full 96-byte shadow-queue entries (A/B knob)
    int shadeSort = -1;                // HRPT_WF_SHADE_SORT = 0 / 1: general wf_shade variants shade in queue order / grouped by shading class (-1: automatic)
    int shadowPath = 0;                // scenes with non-opaque geometry: 0 = automatic, 1 = wf_shadow traverses itself (buffered query), 2 = any-hit pass + resolve
};

bool wavefront_supports(const SceneView& scene, const HrptPathTracerConstants& constants);
hipError_t wavefront_render(WavefrontState& st, const SceneView& scene, const SceneTraits& traits, const HrptPathTracerConstants& constants,
                            uint32_t accumCount, float4* accumulation, float4* output, uint32_t width, uint32_t height, TileRect rect,
                            DeviceCounters* counters, hipStream_t stream, std::string& error);
// hrpt_trace_rays over device arrays through the persistent refilling traversal kernel (4-wide tree); false when the tree is too deep for it
bool wavefront_trace_rays_supported(const SceneTraits& traits);
hipError_t wavefront_trace_rays(WavefrontState& st, const SceneView& scene, const SceneTraits& traits, const HrptRay* rays, HrptRayHit* hits, uint64_t count,
                                bool shadow, hipStream_t stream, std::string& error);
void wavefront_release(WavefrontState& st);
void wavefront_collect_timing(WavefrontState& st);   // call after the stream is synchronised; folds pending events into kernelMs
void wavefront_reset_timing(WavefrontState& st);
// Queue bytes per kernel class from the device counters (summed over the shards) and the record layout of the last render.
void wavefront_queue_bytes(const WavefrontState& st, const DeviceCounters& total, uint64_t& trace, uint64_t& shade, uint64_t& shadow);

} // namespace hrt
