// bvh_build_gpu.h -- GPU construction of the acceleration structure (see bvh_build_gpu.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <functional>
#include <string>

#include "../../include/hobbyrt_pt.h"
#include "pt_device.h"

namespace hrt {

struct GpuBuiltBvh {            // device pointers; nodes/nodes4/tris/attrs/tangents come from `sceneAlloc` (scene lifetime)
    GpuNode* nodes = nullptr; uint32_t nodeCount = 0;
    GpuNode4* nodes4 = nullptr; uint32_t node4Count = 0;
    GpuTri* tris = nullptr; GpuTriAttr* attrs = nullptr; GpuTriTangent* tangents = nullptr; uint32_t triCount = 0;
    uint32_t maxDepth = 0, maxDepth4 = 0;
    bool ploc = false;          // hierarchy by PLOC (nearest-neighbour clustering) rather than the Morton radix tree
    uint32_t mortonBits = 0;    // Morton bits the hierarchy used (63, or fewer when the full-code tree was too deep)
    float sahCost = 0.0f;       // surface-area-heuristic cost of the 2-wide tree (node cost 1, triangle cost 1), root area = 1
    float deviceMs = 0.0f;      // input copies excluded: first kernel .. last node copy
};

// `scene` must already be validated (bvh_build.h validate_scene). sceneAlloc(bytes) returns device memory owned by the caller
// (nullptr on failure). Triangle counts below 8 are not handled (the caller uses the host builder).
// The tree is rebuilt with fewer Morton bits until maxDepth + 2 <= maxStackDepth (the caller still checks the final depth).
hipError_t build_scene_bvh_gpu(const HrptSceneDesc& scene, bool needTangents, bool usePloc, uint32_t maxStackDepth, const std::function<void*(size_t)>& sceneAlloc,
                               hipStream_t stream, GpuBuiltBvh& out, std::string& error);

} // namespace hrt
