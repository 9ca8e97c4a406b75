// pt_megakernel.hip -- validation kernel: one thread per pixel runs the whole of PathTracer_CSMain
// (/root/reference/src/shaders/PathTracer.hlsl:53-340) for one accumulation index, exactly like one
// dispatch of the reference (8x8 groups -> here 64-lane waves over 8x8 pixel tiles). It exists to prove
// struct layouts, RNG streams, LUT sampling, the BVH and the shading stages bit-for-bit against the CPU
// oracle; the wavefront pipeline (pt_wavefront.hip) is the production path and reuses the same stages.
#include <type_traits>

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

// TL: the scene holds the two-level structure (SceneView::instances): the same shader over closest_two_level / shadow_query_two_level
// (pt_device.h, pt_path.h), so that the two-level kernels of the wavefront pipeline keep the wavefront == megakernel cross-check every other
// path has. Private 64-entry stack: scenes whose two-level stack need is larger are refused by hrpt_render.
template <bool TL>
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
        typename std::conditional<TL, GlobalBvhTl, GlobalBvh>::type bvh;
        if constexpr (TL) { bvh.nodes = s.nodes4; bvh.tris = s.tris; bvh.instances = s.instances; } else { bvh.nodes = s.nodes; bvh.tris = s.tris; }
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
    if (scene.instances) hipLaunchKernelGGL(pt_megakernel<true>, grid, dim3(64, 1, 1), 0, stream, scene, constants, accumulation, output, imageWidth, rect, counters);
    else hipLaunchKernelGGL(pt_megakernel<false>, grid, dim3(64, 1, 1), 0, stream, scene, constants, accumulation, output, imageWidth, rect, counters);
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

// Self-test of the acceleration structure (hrpt_selftest_bvh): every child box stored in a node must contain what hangs below it -- the
// child's own child boxes when it is an inner node, the vertices of its triangles when it is a leaf. The GPU builder fits boxes bottom-up
// with relaxed agent-scope atomics and write-through stores (bvh_build_gpu.hip k_fit); a stale read there would give a parent box that is
// too small, i.e. silently missed hits, which image parity on a few scenes cannot rule out. One thread per node; violations are counted.
__device__ __forceinline__ bool box_in(const float* mn, const float* mx, const float* cmn, const float* cmx)
{
    return cmn[0] >= mn[0] && cmn[1] >= mn[1] && cmn[2] >= mn[2] && cmx[0] <= mx[0] && cmx[1] <= mx[1] && cmx[2] <= mx[2];
}
__device__ __forceinline__ bool leaf_in(const SceneView& s, int32_t ref, const float* mn, const float* mx)
{
    const uint32_t enc = (uint32_t)(~ref), first = enc >> 2, count = (enc & 3u) + 1u;
    bool ok = first + count <= s.triCount;
    for (uint32_t i = 0; ok && i < count; ++i) {
        const GpuTri& t = s.tris[first + i];
        for (int k = 0; k < 3; ++k)
            ok = ok && t.p0[k] >= mn[k] && t.p0[k] <= mx[k] && t.p1[k] >= mn[k] && t.p1[k] <= mx[k] && t.p2[k] >= mn[k] && t.p2[k] <= mx[k];
    }
    return ok;
}
__global__ void pt_bvh_check_kernel(SceneView s, unsigned long long* violations)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int bad = 0;
    if (i < s.nodeCount) {                              // 2-wide tree
        const GpuNode& n = s.nodes[i];
        const float* mn[2] = { n.lmin, n.rmin }; const float* mx[2] = { n.lmax, n.rmax }; const int32_t ref[2] = { n.left, n.right };
        for (int c = 0; c < 2; ++c) {
            if (ref[c] >= 0) {
                if ((uint32_t)ref[c] >= s.nodeCount) { ++bad; continue; }
                const GpuNode& m = s.nodes[ref[c]];
                if (!box_in(mn[c], mx[c], m.lmin, m.lmax) || !box_in(mn[c], mx[c], m.rmin, m.rmax)) ++bad;
            } else if (!leaf_in(s, ref[c], mn[c], mx[c])) ++bad;
        }
    }
    if (i < s.node4Count) {                             // its 4-wide collapse
        const GpuNode4& n = s.nodes4[i];
        const float* r[6] = { &n.minx.x, &n.miny.x, &n.minz.x, &n.maxx.x, &n.maxy.x, &n.maxz.x };
        const int32_t* ch = &n.child.x;
        for (int c = 0; c < 4; ++c) {
            if (ch[c] == 0x7fffffff) continue;          // unused slot
            const float mn[3] = { r[0][c], r[1][c], r[2][c] }, mx[3] = { r[3][c], r[4][c], r[5][c] };
            if (ch[c] >= 0) {
                if ((uint32_t)ch[c] >= s.node4Count) { ++bad; continue; }
                const GpuNode4& m = s.nodes4[ch[c]];
                const float* q[6] = { &m.minx.x, &m.miny.x, &m.minz.x, &m.maxx.x, &m.maxy.x, &m.maxz.x };
                const int32_t* mch = &m.child.x;
                for (int k = 0; k < 4; ++k) {
                    if (mch[k] == 0x7fffffff) continue;
                    const float cmn[3] = { q[0][k], q[1][k], q[2][k] }, cmx[3] = { q[3][k], q[4][k], q[5][k] };
                    if (!box_in(mn, mx, cmn, cmx)) ++bad;
                }
            } else if (!leaf_in(s, ch[c], mn, mx)) ++bad;
        }
    }
    if (s.nodesQ && i < s.node4Count) {                 // ... and its quantised form: every decoded box contains the fp32 box it was made from
        const GpuNode4& n = s.nodes4[i]; const GpuNodeQ& q = s.nodesQ[i];
        const float* r[6] = { &n.minx.x, &n.miny.x, &n.minz.x, &n.maxx.x, &n.maxy.x, &n.maxz.x };
        const float o[3] = { q.ox, q.oy, q.oz }, st[3] = { q.sx, q.sy, q.sz };
        const uint32_t lo[3] = { q.lox, q.loy, q.loz }, hi[3] = { q.hix, q.hiy, q.hiz };
        const int32_t* ch = &n.child.x;
        for (int c = 0; c < 4; ++c) {
            if (q.child[c] != ch[c]) ++bad;
            for (int a = 0; a < 3; ++a) {
                const float dlo = o[a] + (float)((lo[a] >> (8 * c)) & 255u) * st[a], dhi = o[a] + (float)((hi[a] >> (8 * c)) & 255u) * st[a];
                if (ch[c] == 0x7fffffff) { if (!(dlo > dhi)) ++bad; }                 // unused slot: an empty interval on every axis
                else if (!(dlo <= r[a][c] && dhi >= r[3 + a][c]) || !(st[a] > 0.0f)) ++bad;
            }
        }
    }
    if (bad) atomicAdd(violations, (unsigned long long)bad);
}
// GpuNode4 -> GpuNodeQ (see pt_device.h), one thread per node; run after every build / rebuild of the flat structure, whichever builder made it.
__global__ void pt_quantise_nodes_kernel(const GpuNode4* __restrict__ in, uint32_t count, GpuNodeQ* __restrict__ out, double* __restrict__ leafArea)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const GpuNode4 n = in[i];
    const float* r[6] = { &n.minx.x, &n.miny.x, &n.minz.x, &n.maxx.x, &n.maxy.x, &n.maxz.x };
    const int32_t* ch = &n.child.x;
    float o[3], st[3]; uint32_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        float mn = 3.0e38f, mx = -3.0e38f;
        for (int c = 0; c < 4; ++c) if (ch[c] != 0x7fffffff) { mn = fminf(mn, r[a][c]); mx = fmaxf(mx, r[3 + a][c]); }
        // 254 steps span the extent (the top plane may still need one step outward after rounding)
        float step = (mx - mn) * (1.0f / 254.0f) * 1.000001f;
        step = fmaxf(step, fmaxf(fabsf(mn), fabsf(mx)) * 2.4e-7f);       // never below two ulps of the coordinates: the slack must outweigh the rounding of o + q * s
        step = fmaxf(step, 1.0e-30f);
        o[a] = mn; st[a] = step; lo[a] = 0u; hi[a] = 0u;
        for (int c = 0; c < 4; ++c) {
            uint32_t ql = 255u, qh = 0u;
            if (ch[c] != 0x7fffffff) {
                float fl = floorf((r[a][c] - mn) / step), fh = ceilf((r[3 + a][c] - mn) / step);
                fl = fminf(fmaxf(fl, 0.0f), 255.0f); fh = fminf(fmaxf(fh, 0.0f), 255.0f);
                ql = (uint32_t)fl; qh = (uint32_t)fh;
                // exactly the decode the self-test uses: step outward until the decoded plane is on the right side (at most a step or two)
                while (ql > 0u && !(mn + (float)ql * step <= r[a][c])) --ql;
                while (qh < 255u && !(mn + (float)qh * step >= r[3 + a][c])) ++qh;
            }
            lo[a] |= ql << (8 * c); hi[a] |= qh << (8 * c);
        }
    }
    // what the rounding costs: surface area of the LEAF boxes before and after (the chance that a ray enters a box grows with its area, and
    // every entered leaf is three requests and a watertight test per triangle): leafArea[0] += fp32 area, leafArea[1] += decoded area
    if (leafArea) {
        double a0 = 0.0, a1 = 0.0, r0 = 0.0, r1 = 0.0;
        for (int c = 0; c < 4; ++c) {
            if (ch[c] == 0x7fffffff || ch[c] >= 0) continue;
            float e0[3], e1[3];
            for (int a = 0; a < 3; ++a) {
                e0[a] = r[3 + a][c] - r[a][c];
                e1[a] = (o[a] + (float)((hi[a] >> (8 * c)) & 255u) * st[a]) - (o[a] + (float)((lo[a] >> (8 * c)) & 255u) * st[a]);
            }
            const double s0 = (double)e0[0] * e0[1] + (double)e0[1] * e0[2] + (double)e0[2] * e0[0], s1 = (double)e1[0] * e1[1] + (double)e1[1] * e1[2] + (double)e1[2] * e1[0];
            a0 += s0; a1 += s1;
            if (s0 > 0.0) { r0 += s1 / s0; r1 += 1.0; }
        }
        if (a0 > 0.0 || a1 > 0.0) { atomicAdd(&leafArea[0], a0); atomicAdd(&leafArea[1], a1); atomicAdd(&leafArea[2], r0); atomicAdd(&leafArea[3], r1); }
    }
    GpuNodeQ q;
    q.ox = o[0]; q.oy = o[1]; q.oz = o[2]; q.sx = st[0]; q.sy = st[1]; q.sz = st[2];
    q.lox = lo[0]; q.hix = hi[0]; q.loy = lo[1]; q.hiy = hi[1]; q.loz = lo[2]; q.hiz = hi[2];
    for (int c = 0; c < 4; ++c) q.child[c] = ch[c];
    out[i] = q;
}
hipError_t launch_quantise_nodes(const GpuNode4* nodes4, uint32_t count, GpuNodeQ* out, double* leafArea, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_quantise_nodes_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, nodes4, count, out, leafArea);
    return hipGetLastError();
}
hipError_t launch_bvh_check(const SceneView& scene, unsigned long long* violations, hipStream_t stream)
{
    const uint32_t n = scene.nodeCount > scene.node4Count ? scene.nodeCount : scene.node4Count;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_bvh_check_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, scene, violations);
    return hipGetLastError();
}

// Stand-alone ray queries over the scene's acceleration structure: TraceRayStandard (RaytracingCommon.hlsli:138-198) and
// CalculateRTShadow<true> (CommonLighting.hlsli:380-496) for callers other than the path tracer (the reference's DDGI probe trace,
// ray-traced shadows and BRDF ray tracing share exactly these two includes). One thread per ray, 2-wide tree, private stack.
template <bool SHADOW, bool TL>
__global__ __launch_bounds__(256) void pt_trace_rays_kernel(SceneView s, const HrptRay* __restrict__ rays, HrptRayHit* __restrict__ hits, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    HrptRay r = rays[i];
    HrptRayHit out; out.t = 0.0f; out.u = 0.0f; out.v = 0.0f; out.instance = 0; out.primitive = 0; out.hit = 0; out.rng = r.rng; out.pad = 0;
    typename std::conditional<TL, GlobalBvhTl, GlobalBvh>::type bvh;
    if constexpr (TL) { bvh.nodes = s.nodes4; bvh.tris = s.tris; bvh.instances = s.instances; } else { bvh.nodes = s.nodes; bvh.tris = s.tris; }
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
    if (scene.instances) {
        if (shadow) hipLaunchKernelGGL((pt_trace_rays_kernel<true, true>), grid, dim3(256), 0, stream, scene, rays, hits, count);
        else hipLaunchKernelGGL((pt_trace_rays_kernel<false, true>), grid, dim3(256), 0, stream, scene, rays, hits, count);
    } else if (shadow) hipLaunchKernelGGL((pt_trace_rays_kernel<true, false>), grid, dim3(256), 0, stream, scene, rays, hits, count);
    else hipLaunchKernelGGL((pt_trace_rays_kernel<false, false>), grid, dim3(256), 0, stream, scene, rays, hits, count);
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
