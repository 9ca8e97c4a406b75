// pt_megakernel.hip -- validation kernel: one thread per pixel runs the whole of PathTracer_CSMain
// (/root/reference/src/shaders/PathTracer.hlsl:53-340) for one accumulation index, exactly like one
// dispatch of the reference (8x8 groups -> here 64-lane waves over 8x8 pixel tiles). It exists to prove
// struct layouts, RNG streams, LUT sampling, the BVH and the shading stages bit-for-bit against the CPU
// oracle; the wavefront pipeline (pt_wavefront.hip) is the production path and reuses the same stages.
#include "pt_kernels.h"
#include "pt_path.h"

namespace hrt {

struct PrivateStack {
    int32_t e[64];          // kTraversalStackDepth (bvh_build.h)
    HRT_DEV void push(int sp, int32_t v) { e[sp & 63] = v; }
    HRT_DEV int32_t pop(int sp) { return e[sp & 63]; }
};

HRT_DEV unsigned long long wave_sum(unsigned int v)
{
    unsigned long long x = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__global__ __launch_bounds__(64) void pt_megakernel(SceneView s, HrptPathTracerConstants cb, float4* __restrict__ accumulation,
                                                    float4* __restrict__ output, uint32_t imageWidth, TileRect rect,
                                                    DeviceCounters* counters)
{
    // 8x8 pixel tile per 64-lane wave (same footprint as [numthreads(8,8,1)], PathTracer.hlsl:52)
    uint32_t lx = threadIdx.x & 7u, ly = threadIdx.x >> 3;
    uint32_t px = rect.column_x(blockIdx.x) + lx, py = rect.y0 + blockIdx.y * 8u + ly;
    bool active = px < rect.x1 && py < rect.y1;
    unsigned int nClosest = 0, nShadow = 0;

    if (active) {
        GlobalBvh bvh; bvh.nodes = s.nodes; bvh.tris = s.tris;
        PrivateStack stack;
        PathState ps; init_path(ps, cb, px, py);
        int maxBounces = (int)cb.m_MaxBounces;
        for (int bounce = 0; bounce < maxBounces; ++bounce) {
            Hit hit;
            ++nClosest;
            if (trace_standard(s, bvh, ps.ray, ps.rng, stack, hit)) {
                SurfaceCarry carry;
                f3 totalDiffuse = mk3(0.0f, 0.0f, 0.0f), totalSpecular = mk3(0.0f, 0.0f, 0.0f);
                f3 neeThroughput;
                const f3 sunDir = mk3(cb.m_SunDirection[0], cb.m_SunDirection[1], cb.m_SunDirection[2]);
                const float sunIntensity = s.lights[0].m_Intensity;                  // g_Lights[0], PathTracer.hlsl:137 (quirk kept)
                SurfaceOutcome oc = shade_surface_a<true, true, false>(s, cb, ps, hit, carry, [&](uint32_t li, float ux, float uy) {
                    HrptGPULight l = load_light(s, li);
                    f3 L; float maxDist;
                    if (!nee_direction<false>(l, carry.N, carry.worldPos, sunDir, cb.m_CosSunAngularRadius, ux, uy, L, maxDist)) return;
                    ++nShadow;
                    float shadow = shadow_query(s, bvh, carry.worldPos, L, maxDist, stack);
                    if (shadow != 0.0f) {                                              // an occluded sample contributes +0
                        f3 dif, spec;
                        nee_contribution<false>(s, l, nee_lighting(carry.N, carry.V, carry.baseColor, carry.roughness, carry.metallic, carry.ior),
                                                carry.worldPos, sunDir, sunIntensity, L, dif, spec);
                        totalDiffuse = totalDiffuse + dif * shadow;
                        totalSpecular = totalSpecular + spec * shadow;
                    }
                });
                if (oc == SURFACE_TRANSMITTED) continue;
                neeThroughput = ps.throughput;
                f3 dsum = totalDiffuse + (bounce == 0 ? totalSpecular : mk3(0.0f, 0.0f, 0.0f));
                ps.radiance = ps.radiance + neeThroughput * dsum;                   // PathTracer.hlsl:261
                if (!shade_surface_b(ps, carry, bounce)) break;
            } else {
                miss_sky(s, cb, ps, bounce);
                break;
            }
        }
        // accumulate + resolve, PathTracer.hlsl:332-339
        size_t idx = (size_t)py * imageWidth + px;
        float4 accum = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z, 1.0f);
        if (cb.m_AccumulationIndex > 0) {
            float4 prev = accumulation[idx];
            accum.x += prev.x; accum.y += prev.y; accum.z += prev.z; accum.w += prev.w;
        }
        accumulation[idx] = accum;
        output[idx] = make_float4(accum.x / accum.w, accum.y / accum.w, accum.z / accum.w, 1.0f);
    }
    unsigned long long c = wave_sum(nClosest), sh = wave_sum(nShadow), pa = wave_sum(active ? 1u : 0u);
    if (threadIdx.x == 0) {
        DeviceCounters* shard = counters + (blockIdx.x + blockIdx.y * gridDim.x) % kCounterShards;   // spread the tail atomics
        atomicAdd(&shard->closestRays, c);
        atomicAdd(&shard->shadowRays, sh);
        atomicAdd(&shard->paths, pa);
    }
}

__global__ void pt_resolve_kernel(const float4* __restrict__ accumulation, float4* __restrict__ output, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 a = accumulation[i];
        output[i] = make_float4(a.x / a.w, a.y / a.w, a.z / a.w, 1.0f);
    }
}

// Column shards of `ranks` ranks, rank-major as an all-gather delivers them ([rank][y][column of the rank][8 pixels]), straight to the
// resolved image (and, when asked for, the assembled accumulation image): one pass instead of re-assembly + resolve.
__global__ void pt_resolve_columns_kernel(const float4* __restrict__ shards, float4* __restrict__ accumulation, float4* __restrict__ output,
                                          uint32_t width, uint32_t height, uint32_t ranks)
{
    const uint32_t n = width * height, perRank = width / 8u / ranks;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t y = i / width, x = i - y * width, k = x >> 3, r = k % ranks, c = k / ranks;
        float4 a = shards[(((size_t)r * height + y) * perRank + c) * 8u + (x & 7u)];
        if (accumulation) accumulation[i] = a;
        output[i] = make_float4(a.x / a.w, a.y / a.w, a.z / a.w, 1.0f);
    }
}
hipError_t launch_resolve_columns(const float4* shards, float4* accumulation, float4* output, uint32_t width, uint32_t height, uint32_t ranks, hipStream_t stream)
{
    if (width == 0 || height == 0) return hipSuccess;
    uint32_t blocks = (width * height + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pt_resolve_columns_kernel, dim3(blocks), dim3(256), 0, stream, shards, accumulation, output, width, height, ranks);
    return hipGetLastError();
}

hipError_t launch_megakernel(const SceneView& scene, const HrptPathTracerConstants& constants, float4* accumulation, float4* output,
                             uint32_t imageWidth, TileRect rect, DeviceCounters* counters, hipStream_t stream)
{
    if (rect.x1 <= rect.x0 || rect.y1 <= rect.y0 || rect.columns() == 0) return hipSuccess;
    dim3 grid(rect.columns(), (rect.y1 - rect.y0 + 7) / 8, 1);
    hipLaunchKernelGGL(pt_megakernel, grid, dim3(64, 1, 1), 0, stream, scene, constants, accumulation, output, imageWidth, rect, counters);
    return hipGetLastError();
}

__global__ void pt_f16_table_kernel(float* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536u) out[i] = half_bits_to_float(i);
}
__global__ void pt_unorm8_table_kernel(float* __restrict__ out)
{
    uint32_t i = threadIdx.x;
    if (i < 256u) { out[i] = unorm8_to_float(i); out[256u + i] = (float)i / 255.0f; }      // fast path, and the division it replaces
}
hipError_t launch_unorm8_table(float* out512, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_unorm8_table_kernel, dim3(1), dim3(256), 0, stream, out512);
    return hipGetLastError();
}
hipError_t launch_f16_table(float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_f16_table_kernel, dim3(256), dim3(256), 0, stream, out);
    return hipGetLastError();
}

// Stand-alone ray queries over the scene's acceleration structure: TraceRayStandard (RaytracingCommon.hlsli:138-198) and
// CalculateRTShadow<true> (CommonLighting.hlsli:380-496) for callers other than the path tracer (the reference's DDGI probe trace,
// ray-traced shadows and BRDF ray tracing share exactly these two includes). One thread per ray, 2-wide tree, private stack.
template <bool SHADOW>
__global__ __launch_bounds__(256) void pt_trace_rays_kernel(SceneView s, const HrptRay* __restrict__ rays, HrptRayHit* __restrict__ hits, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    HrptRay r = rays[i];
    HrptRayHit out; out.t = 0.0f; out.u = 0.0f; out.v = 0.0f; out.instance = 0; out.primitive = 0; out.hit = 0; out.rng = r.rng; out.pad = 0;
    GlobalBvh bvh; bvh.nodes = s.nodes; bvh.tris = s.tris;
    PrivateStack stack;
    f3 o = mk3(r.origin[0], r.origin[1], r.origin[2]), d = mk3(r.direction[0], r.direction[1], r.direction[2]);
    const bool finite = d.x == d.x && d.y == d.y && d.z == d.z && o.x == o.x && o.y == o.y && o.z == o.z;
    if (SHADOW) {
        out.t = finite ? shadow_query(s, bvh, o, d, r.tmax, stack) : 1.0f;     // visibility in [0, 1]
        out.hit = out.t < 1.0f ? 1u : 0u;
    } else if (finite) {
        Ray ray; ray.o = o; ray.d = d; ray.tmin = r.tmin; ray.tmax = r.tmax;
        uint32_t rng = r.rng; Hit h;
        if (trace_standard(s, bvh, ray, rng, stack, h)) { out.t = h.t; out.u = h.u; out.v = h.v; out.instance = h.inst; out.primitive = h.prim; out.hit = 1; }
        out.rng = rng;
    }
    hits[i] = out;
}
hipError_t launch_trace_rays(const SceneView& scene, const HrptRay* rays, HrptRayHit* hits, uint64_t count, bool shadow, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    dim3 grid((unsigned)((count + 255) / 256));
    if (shadow) hipLaunchKernelGGL((pt_trace_rays_kernel<true>), grid, dim3(256), 0, stream, scene, rays, hits, count);
    else hipLaunchKernelGGL((pt_trace_rays_kernel<false>), grid, dim3(256), 0, stream, scene, rays, hits, count);
    return hipGetLastError();
}

hipError_t launch_resolve(const float4* accumulation, float4* output, uint32_t pixelCount, hipStream_t stream)
{
    if (pixelCount == 0) return hipSuccess;
    uint32_t blocks = (pixelCount + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pt_resolve_kernel, dim3(blocks), dim3(256), 0, stream, accumulation, output, pixelCount);
    return hipGetLastError();
}

} // namespace hrt
