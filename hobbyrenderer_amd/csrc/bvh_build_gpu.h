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
    bool refitted = false;      // boxes refitted on the hierarchy of an earlier build (GpuBvhBuilder::refit)
    const uint32_t* leafOrder = nullptr;   // primitive (input index) at every position of the leaf order the leaf references count in
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
    // Box mode -- the tree over the instances of the two-level structure (pt_capi.cpp build_two_level): `count` boxes (6 floats each, min xyz
    // then max xyz, taken as they are), one per leaf, Morton grid over the cube of the largest extent. The result has nodes / nodes4 /
    // leafOrder only; a leaf reference ~(k << 2) names box leafOrder[k].
    hipError_t prepare_boxes(uint32_t count, hipStream_t stream, std::string& error);
    hipError_t build_boxes(const float* boxes, bool usePloc, uint32_t maxStackDepth, hipStream_t stream, GpuBuiltBvh& out, std::string& error);
    // New boxes on the hierarchy of the last successful build() / build_boxes() (same primitives, moved): no sort, no hierarchy construction.
    // The tree stays valid whatever the motion (every box is recomputed bottom-up); its quality is that of the old topology on the new positions.
    bool can_refit() const;
    hipError_t refit(const HrptPerInstanceData* instances, hipStream_t stream, GpuBuiltBvh& out, std::string& error);
    hipError_t refit_boxes(const float* boxes, hipStream_t stream, GpuBuiltBvh& out, std::string& error);
private:
    hipError_t refit_any(const HrptPerInstanceData* instances, const float* boxes, hipStream_t stream, GpuBuiltBvh& out, std::string& error);
    hipError_t allocate(uint32_t n, const HrptSceneDesc* scene, bool needTangents, hipStream_t stream, std::string& error);
    hipError_t build_any(const HrptPerInstanceData* instances, const float* boxes, bool usePloc, uint32_t maxStackDepth, hipStream_t stream, GpuBuiltBvh& out, std::string& error);
    struct Impl;
    Impl* p = nullptr;
};
// copies the `count` 4-wide nodes of a tree built in box mode to `dst`, leaves rewritten as two-level instance references ~(instance << 2)
hipError_t launch_tlas_fixup(const GpuNode4* src, uint32_t count, const uint32_t* leafOrder, GpuNode4* dst, hipStream_t stream);

} // namespace hrt
