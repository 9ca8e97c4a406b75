// bvh_build_gpu.h -- GPU construction of the acceleration structure (see bvh_build_gpu.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hobbyrt_pt.h"
#include "pt_device.h"

namespace hrt {

struct GpuBuiltBvh {            // device pointers owned by the GpuBvhBuilder that produced them (valid until its next build() / destruction)
    GpuNode* nodes = nullptr; uint32_t nodeCount = 0;
    GpuNode4* nodes4 = nullptr; uint32_t node4Count = 0;
    GpuTri* tris = nullptr; GpuTriAttr* attrs = nullptr; GpuTriTangent* tangents = nullptr; uint32_t triCount = 0;
    uint32_t maxDepth = 0, maxDepth4 = 0;
    bool ploc = false;          // hierarchy by PLOC (nearest-neighbour clustering) rather than the Morton radix tree
    uint32_t mortonBits = 0;    // Morton bits the hierarchy used (63, or fewer when the full-code tree was too deep)
    float sahCost = 0.0f;       // surface-area-heuristic cost of the 2-wide tree (node cost 1, triangle cost 1), root area = 1
    float deviceMs = 0.0f;      // instance upload .. last kernel
};

// Keeps the scene's geometry (quantised vertices, indices) and every build buffer on the device, so that a rebuild after a transform
// change (hrpt_update_instances; the reference's per-frame TLAS rebuild, src/CommonRenderers.cpp:234-246) only uploads the instance
// table and runs the kernels: no allocation, no geometry traffic over PCIe.
class GpuBvhBuilder {
public:
    GpuBvhBuilder() = default;
    GpuBvhBuilder(const GpuBvhBuilder&) = delete;
    GpuBvhBuilder& operator=(const GpuBvhBuilder&) = delete;
    ~GpuBvhBuilder();
    // `scene` must already be validated (bvh_build.h validate_scene). Triangle counts below 8 are not handled (the caller uses the
    // host builder). Copies vertices / indices to the device and sizes all buffers for this scene's triangle count.
    hipError_t prepare(const HrptSceneDesc& scene, bool needTangents, hipStream_t stream, std::string& error);
    // (Re)builds from the world matrices of `instances` (as many as at prepare(), same mesh / material indices). The tree is rebuilt
    // with fewer Morton bits until maxDepth + 2 <= maxStackDepth (the caller still checks the final depth). Synchronises `stream`.
    hipError_t build(const HrptPerInstanceData* instances, bool usePloc, uint32_t maxStackDepth, hipStream_t stream, GpuBuiltBvh& out, std::string& error);
    size_t deviceBytes() const;
private:
    struct Impl;
    Impl* p = nullptr;
};

} // namespace hrt
