// pt_wavefront.h -- host interface of the persistent wavefront pipeline (pt_wavefront.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/hobbyrt_pt.h"
#include "pt_kernels.h"

namespace hrt {

struct SceneView;

struct WavefrontState {
    void* pool = nullptr;            // one device allocation carved into the SoA queues
    size_t poolBytes = 0;
    uint32_t capacityPaths = 0;
    hipEvent_t evA = nullptr, evB = nullptr;
    float traceMs = 0.0f;            // summed device time of the closest-hit trace kernel in the last render
    uint32_t traceLaunches = 0;
};

bool wavefront_supports(const SceneView& scene, const HrptPathTracerConstants& constants);
hipError_t wavefront_render(WavefrontState& st, const SceneView& scene, const HrptPathTracerConstants& constants, uint32_t accumCount,
                            float4* accumulation, float4* output, uint32_t width, uint32_t height, TileRect rect,
                            DeviceCounters* counters, hipStream_t stream, std::string& error);
void wavefront_release(WavefrontState& st);
void wavefront_trace_timing(const WavefrontState& st, float* ms, uint32_t* launches);

} // namespace hrt
