// pt_kernels.h -- host-callable launchers of the gfx950 path-tracer kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hobbyrt_pt.h"

namespace hrt {

struct SceneView;   // pt_device.h
struct GpuNode4; struct GpuNodeQ;

constexpr int kCounterShards = 32;   // DeviceCounters[kCounterShards] per context; a block adds to shard blockIdx % kCounterShards
struct DeviceCounters {         // 64 B; summed over the shards by hrpt_get_stats. The wavefront kernels address the fields by word index.
    unsigned long long closestRays;       // 0
    unsigned long long shadowRays;        // 1
    unsigned long long paths;             // 2
    unsigned long long neeEntries;        // 3: shadow-queue entries wf_shade wrote (path vertices with at least one light sample)
    unsigned long long neeSamples;        // 4: light-sample records wf_shadow read
    unsigned long long radianceShade;     // 5: sampleRadiance read-modify-writes of wf_shade (emissive / sky terms)
    unsigned long long radianceShadow;    // 6: sampleRadiance read-modify-writes of wf_shadow (NEE terms)
    unsigned long long skipped16;         // 7: 16-byte path-record reads wf_shadow's slim mode did not make at bounce 0 of a batch without raygen pass
};

struct TileRect {
    uint32_t x0, y0, x1, y1;
    uint32_t stripeCount = 1, stripeIndex = 0;     // of the rectangle's 8-pixel columns, those with column % stripeCount == stripeIndex
    // number of 8-pixel columns this call covers / pixel column of its k-th one
    __host__ __device__ uint32_t columns() const { uint32_t all = (x1 - x0 + 7u) / 8u; return all > stripeIndex ? (all - stripeIndex + stripeCount - 1u) / stripeCount : 0u; }
    __host__ __device__ uint32_t column_x(uint32_t k) const { return x0 + (k * stripeCount + stripeIndex) * 8u; }
};

// One dispatch of the reference shader: one path per pixel of `rect` for constants.m_AccumulationIndex.
// Validation path: one thread per pixel, private traversal stack.
hipError_t launch_megakernel(const SceneView& scene, const HrptPathTracerConstants& constants, float4* accumulation,
                             float4* output, uint32_t imageWidth, TileRect rect, DeviceCounters* counters, hipStream_t stream);

// HDR post chain over `hdr` (W*H float4): histogram[256] + exposure[1] are context-owned device buffers.
hipError_t launch_post_chain(const float4* hdr, float4* display, uint32_t pixelCount, const HrptPostParams& params, uint32_t* histogram,
                             float* exposure, hipStream_t stream);

// Batch ray queries (hrpt_trace_rays): closest hit with the candidate rules of TraceRayStandard, or NEE-style visibility.
hipError_t launch_trace_rays(const SceneView& scene, const HrptRay* rays, HrptRayHit* hits, uint64_t count, bool shadow, hipStream_t stream);

// The flat 4-wide tree in its 64-byte quantised form (pt_device.h GpuNodeQ): out[i] from nodes4[i]. leafArea (two zeroed doubles, or null)
// receives the summed surface area of the leaf boxes before / after the rounding: what the looser boxes will cost in triangle tests.
hipError_t launch_quantise_nodes(const GpuNode4* nodes4, uint32_t count, GpuNodeQ* out, double* leafArea, hipStream_t stream);
// Self-test: counts child boxes of the 2-wide and 4-wide trees that do not contain their subtree's boxes / triangle vertices (0 = sound).
hipError_t launch_bvh_check(const SceneView& scene, unsigned long long* violations, hipStream_t stream);
// Self-test: out[i] = device decode of the binary16 pattern i, i in [0, 65536).
hipError_t launch_unorm8_table(float* out512, hipStream_t stream);
hipError_t launch_f16_table(float* out, hipStream_t stream);

// Output[xy] = accum.rgb / accum.a (PathTracer.hlsl:339) over the whole image.
hipError_t launch_resolve(const float4* accumulation, float4* output, uint32_t pixelCount, hipStream_t stream);
hipError_t launch_resolve_columns(const float4* shards, float4* accumulation, float4* output, uint32_t width, uint32_t height, uint32_t ranks, hipStream_t stream);

} // namespace hrt
