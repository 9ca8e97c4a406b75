// pt_path.h -- the per-path stages of PathTracer_CSMain (/root/reference/src/shaders/PathTracer.hlsl:53-340)
// as device functions that both the validation megakernel and the wavefront kernels call:
//   primary_ray        :61-72     TraceRayStandard  RaytracingCommon.hlsli:138-198
//   shade_surface_a    :92-261    (attributes, transmission branch, emissive, NEE sample generation)
//   shade_surface_b    :264-313   (Russian roulette, BRDF lobe pick + sample, ray advance)
//   shadow_query       CommonLighting.hlsli:380-496 (CalculateRTShadow<true>)
//   miss_sky           :315-328
// NEE shadow rays draw no random numbers, so a stage may defer them: shade_surface_a hands every
// light sample to an `emit` callback (direction, distance, unshadowed diffuse/specular radiance).
#pragma once

#include "pt_device.h"

namespace hrt {

struct PathState {
    Ray ray;
    uint32_t rng;
    f3 throughput, radiance;
    // dielectric medium the current segment travels in (PathTracer.hlsl:78-81)
    bool inVolume; float interiorIOR; f3 sigmaA, sigmaS;
};

// what shade_surface_a needs to keep for shade_surface_b
struct SurfaceCarry {
    f3 N, V, worldPos, baseColor, F0; float Fr, roughness, metallic, ior;
    uint32_t rngBeforeLights;      // RNG state at the top of the light loop (:260): lets a schedule replay the loop's draws
    uint32_t material;             // index into the material constants
};

HRT_DEV void init_path(PathState& ps, const HrptPathTracerConstants& cb, uint32_t px, uint32_t py)
{
    // primary ray, PathTracer.hlsl:61-72; UVToClipXY Common.hlsli:50-53
    float u = (((float)px + 0.5f) + cb.m_Jitter[0]) * cb.m_View.m_ViewportSizeInv[0];
    float v = (((float)py + 0.5f) + cb.m_Jitter[1]) * cb.m_View.m_ViewportSizeInv[1];
    float cx = u * 2.0f + -1.0f, cy = v * -2.0f + 1.0f;
    const float* M = cb.m_View.m_MatClipToWorldNoOffset;
    float ex = ((cx * M[0] + cy * M[4]) + 0.9f * M[8]) + 1.0f * M[12];
    float ey = ((cx * M[1] + cy * M[5]) + 0.9f * M[9]) + 1.0f * M[13];
    float ez = ((cx * M[2] + cy * M[6]) + 0.9f * M[10]) + 1.0f * M[14];
    float ew = ((cx * M[3] + cy * M[7]) + 0.9f * M[11]) + 1.0f * M[15];
    f3 end = mk3(ex / ew, ey / ew, ez / ew);
    ps.ray.o = mk3(cb.m_CameraPos[0], cb.m_CameraPos[1], cb.m_CameraPos[2]);
    ps.ray.d = normalize(end - ps.ray.o);
    ps.ray.tmin = 0.0f; ps.ray.tmax = 1e10f;
    ps.rng = hrt_rng_seed(px, py, cb.m_AccumulationIndex);        // RNG.hlsli:21-27
    ps.throughput = mk3(1.0f, 1.0f, 1.0f); ps.radiance = mk3(0.0f, 0.0f, 0.0f);
    ps.inVolume = false; ps.interiorIOR = 1.0f; ps.sigmaA = mk3(0.0f, 0.0f, 0.0f); ps.sigmaS = mk3(0.0f, 0.0f, 0.0f);
}

// The candidate branch of TraceRayStandard (RaytracingCommon.hlsli:153-184) for a hit on a ForceNonOpaque instance:
// MASK -> AlphaTest (:91-110); BLEND -> transmissive materials always commit, others commit with probability alpha
// (one RNG draw, :181).
HRT_DEV bool candidate_commits(const SceneView& s, const Hit& h, uint32_t& rng)
{
    TriVerts tv = load_hit_attr(s, h);
    const HrptMaterialConstants& mat = s.materials[tv.material];
    uint32_t alphaMode = mat.m_AlphaMode;
    if (alphaMode == HRPT_ALPHA_MODE_MASK || (alphaMode == HRPT_ALPHA_MODE_BLEND && !(mat.m_TransmissionFactor > 0.0f))) {
        f2 uv = interpolated_uv(tv, h.u, h.v);
        float alpha = candidate_alpha(s, mat, uv);
        if (alphaMode == HRPT_ALPHA_MODE_MASK) return alpha >= mat.m_AlphaCutoff;
        return hrt_rng_next(&rng) < hrt_saturate(alpha);
    }
    return alphaMode == HRPT_ALPHA_MODE_BLEND;
}

// TraceRayStandard, RaytracingCommon.hlsli:138-198. Non-opaque candidates are visited front to back:
// a rejected candidate becomes the exclusive lower bound of the next closest-hit query.
template <class BVH, class STACK>
HRT_DEV bool trace_standard(const SceneView& s, const BVH& bvh, const Ray& ray, uint32_t& rng, STACK& stack, Hit& out)
{
    HitKey lower; lower.have = false; lower.t = 0.0f; lower.inst = 0; lower.prim = 0;
    for (;;) {
        Hit h;
        if constexpr (BVH::kTwoLevel) h = closest_two_level(bvh, s.rootLeaf, s.nodeCount, ray, lower, stack);
        else h = closest_any(bvh, s.rootLeaf, s.nodeCount, ray, lower, stack);
        if (!h.valid) return false;
        if ((h.opaque & 1u) || candidate_commits(s, h, rng)) { out = h; return true; }
        lower.have = true; lower.t = h.t; lower.inst = h.inst; lower.prim = h.prim;
    }
}

// ---- CalculateRTShadow<true>, CommonLighting.hlsli:380-496 ------------------------------------------------------
// State carried from candidate to candidate (:404-407) and the per-candidate body (:409-470); returns true when the
// candidate commits (the query then returns 0).
struct ShadowState { float transmission; bool inVolume; float inVolumeStartT; f3 sigmaT; };

HRT_DEV bool shadow_candidate(const SceneView& s, const Ray& ray, float hitT, uint32_t tri, float bu, float bv, ShadowState& st)
{
    TriVerts tv = load_tri_attr(s, tri);           // inst.m_LODIndex is 0 on this path (validated at upload), :423
    const HrptMaterialConstants& mat = s.materials[tv.material];
    f2 uv = interpolated_uv(tv, bu, bv);
    if (mat.m_AlphaMode == HRPT_ALPHA_MODE_MASK) {
        // AlphaTestGrad (RaytracingCommon.hlsli:112-130) with GetShadowRayGradients (:207-240); level 0 unless the texture has a mip chain
        return candidate_alpha_grad(s, mat, uv, tv, tri, bu, bv, ray.o) >= mat.m_AlphaCutoff;
    }
    if (mat.m_AlphaMode != HRPT_ALPHA_MODE_BLEND) return true;
    float alpha = candidate_alpha(s, mat, uv);
    float opacity = hrt_saturate(alpha * (1.0f - mat.m_TransmissionFactor));
    st.transmission *= (1.0f - opacity);
    if (mat.m_TransmissionFactor > 0.0f && mat.m_IsThinSurface == 0) {
        float w0 = (1.0f - bu) - bv;
        f3 ln = (tv.n0 * w0 + tv.n1 * bu) + tv.n2 * bv;
        f3 wn = normalize(transform_normal(ln, s.instShade[tv.inst]));
        bool front = dot(wn, ray.d) < 0.0f;
        if (front) { st.inVolume = true; st.inVolumeStartT = hitT; st.sigmaT = mk3(mat.m_SigmaA) + mk3(mat.m_SigmaS); }
        else if (st.inVolume) {
            float seg = hrt_max(0.0f, hitT - st.inVolumeStartT);
            f3 tr = mk3(hrt_exp(-st.sigmaT.x * seg), hrt_exp(-st.sigmaT.y * seg), hrt_exp(-st.sigmaT.z * seg));
            st.transmission *= dot(tr, mk3(0.2126f, 0.7152f, 0.0722f));
            st.inVolume = false;
        }
    }
    return st.transmission <= 1e-3f;
}
HRT_DEV float shadow_finish(const Ray& ray, ShadowState& st)                       // :477-495
{
    if (st.inVolume) {
        float seg = hrt_max(0.0f, ray.tmax - st.inVolumeStartT);
        f3 tr = mk3(hrt_exp(-st.sigmaT.x * seg), hrt_exp(-st.sigmaT.y * seg), hrt_exp(-st.sigmaT.z * seg));
        st.transmission *= dot(tr, mk3(0.2126f, 0.7152f, 0.0722f));
    }
    return hrt_saturate(st.transmission);
}
HRT_DEV Ray shadow_ray(f3 worldPos, f3 L, float maxDist)                           // :391-396
{
    const float kShadowBias = 0.01f;
    Ray ray; ray.o = worldPos; ray.d = L; ray.tmin = kShadowBias; ray.tmax = hrt_max(kShadowBias, maxDist - kShadowBias * 2.0f);
    return ray;
}

// A crossed non-opaque triangle of a two-level scene (the per-candidate body above; material and transform come from the hit's instance, the
// world-space vertices the mip-selecting alpha test wants from the instance's m_World).
HRT_DEV bool shadow_candidate_two_level(const SceneView& s, const Ray& ray, const Hit& h, ShadowState& st)
{
    TriVerts tv = load_hit_attr(s, h);
    const HrptMaterialConstants& mat = s.materials[tv.material];
    f2 uv = interpolated_uv(tv, h.u, h.v);
    if (mat.m_AlphaMode == HRPT_ALPHA_MODE_MASK) {
        float alpha = mat.m_BaseColor[3];
        if (mat.m_TextureFlags & HRPT_TEXFLAG_ALBEDO) {
            const uint32_t ti = mat.m_AlbedoTextureIndex;
            if (ti >= s.textureCount || !s.textures[ti].texels) alpha *= 0.0f;
            else {
                const GpuTexture& t = s.textures[ti];
                if (t.mipCount <= 1u) alpha *= sample_texture_level(t.texels, t.format, (int)t.w, (int)t.h, 0u, mat.m_AlbedoSamplerIndex, uv).w;
                else {      // GetShadowRayGradients (RaytracingCommon.hlsli:207-240) from the world-space triangle, as candidate_alpha_grad
                    const float4* tp = reinterpret_cast<const float4*>(s.tris + h.tri);
                    f3 p0, p1, p2; tl_world_triangle(s.instances[h.inst], tp[0], tp[1], tp[2], p0, p1, p2);
                    const float w0 = (1.0f - h.u) - h.v;
                    const f3 hitPos = (p0 * w0 + p1 * h.u) + p2 * h.v;
                    const float dist = length(hitPos - ray.o);
                    const float triangleArea = length(cross(p1 - p0, p2 - p0)) * 0.5f;
                    f2 uvRange;
                    uvRange.x = hrt_max(tv.uv0.x, hrt_max(tv.uv1.x, tv.uv2.x)) - hrt_min(tv.uv0.x, hrt_min(tv.uv1.x, tv.uv2.x));
                    uvRange.y = hrt_max(tv.uv0.y, hrt_max(tv.uv1.y, tv.uv2.y)) - hrt_min(tv.uv0.y, hrt_min(tv.uv1.y, tv.uv2.y));
                    const float gradientScale = triangleArea / hrt_max(dist, 0.1f);
                    f2 grad; grad.x = uvRange.x * gradientScale; grad.y = uvRange.y * gradientScale;
                    alpha *= sample_texture_grad(t, mat.m_AlbedoSamplerIndex, uv, grad, grad).w;
                }
            }
        }
        return alpha >= mat.m_AlphaCutoff;
    }
    if (mat.m_AlphaMode != HRPT_ALPHA_MODE_BLEND) return true;
    float alpha = candidate_alpha(s, mat, uv);
    float opacity = hrt_saturate(alpha * (1.0f - mat.m_TransmissionFactor));
    st.transmission *= (1.0f - opacity);
    if (mat.m_TransmissionFactor > 0.0f && mat.m_IsThinSurface == 0) {
        float w0 = (1.0f - h.u) - h.v;
        f3 ln = (tv.n0 * w0 + tv.n1 * h.u) + tv.n2 * h.v;
        f3 wn = normalize(transform_normal(ln, s.instShade[tv.inst]));
        bool front = dot(wn, ray.d) < 0.0f;
        if (front) { st.inVolume = true; st.inVolumeStartT = h.t; st.sigmaT = mk3(mat.m_SigmaA) + mk3(mat.m_SigmaS); }
        else if (st.inVolume) {
            float seg = hrt_max(0.0f, h.t - st.inVolumeStartT);
            f3 tr = mk3(hrt_exp(-st.sigmaT.x * seg), hrt_exp(-st.sigmaT.y * seg), hrt_exp(-st.sigmaT.z * seg));
            st.transmission *= dot(tr, mk3(0.2126f, 0.7152f, 0.0722f));
            st.inVolume = false;
        }
    }
    return st.transmission <= 1e-3f;
}
// CalculateRTShadow<true> over the two-level structure, re-trace form: an any-hit pass over the opaque instances, then -- only for a ray that
// crossed triangles of non-opaque instances -- one closest-hit query per candidate, front to back (the visiting order of every other form).
template <class STACK>
HRT_DEV float shadow_retrace_two_level(const SceneView& s, const GlobalBvhTl& bvh, const Ray& ray, STACK& stack)
{
    ShadowState st; st.transmission = 1.0f; st.inVolume = false; st.inVolumeStartT = 0.0f; st.sigmaT = mk3(0.0f, 0.0f, 0.0f);
    HitKey lower; lower.have = false; lower.t = 0.0f; lower.inst = 0; lower.prim = 0;
    for (;;) {
        Hit h = closest_two_level(bvh, s.rootLeaf, s.nodeCount, ray, lower, stack);
        if (!h.valid) break;
        if (h.opaque) return 0.0f;      // (cannot happen: the any-hit pass would have returned; kept for the symmetry with shadow_resolve_candidates)
        if (shadow_candidate_two_level(s, ray, h, st)) return 0.0f;
        lower.have = true; lower.t = h.t; lower.inst = h.inst; lower.prim = h.prim;
    }
    return shadow_finish(ray, st);
}
// Buffered form over the two-level structure (wf_shadow<TL = 2>): ONE traversal that returns at the first triangle of an opaque instance and keeps
// the K nearest triangles of non-opaque instances, sorted by (t, instance, primitive), in per-lane LDS columns CAND3 of (t, mesh triangle,
// instance); they are then visited front to back, and only a ray that crossed more than K continues with the re-trace loop behind the K-th.
// Same visiting order as shadow_query_two_level, hence the same result.
template <int K, class STACK, class CAND3>
HRT_DEV float shadow_query_two_level_buffered(const SceneView& s, const GlobalBvhTl& bvh, const Ray& ray, STACK& stack, CAND3& cand, uint32_t nodeLoopMin = 0)
{
    if (!(ray.d.x == ray.d.x && ray.d.y == ray.d.y && ray.d.z == ray.d.z)) return 1.0f;
    RayShear sh = make_shear(ray.d);
    TlCull c; tl_world(c, ray);
    int sp = 0, count = 0; bool overflow = false;
    int32_t cur;
    if (s.nodeCount == 0) { if (s.rootLeaf == 0) return 1.0f; cur = s.rootLeaf; } else cur = 0;
    auto prim_of = [&](uint32_t tri) { return reinterpret_cast<const uint32_t*>(bvh.tris + tri)[7]; };      // GpuTri::prim
    for (;;) {
        while (cur >= 0 && cur != kExitBlas) {
            cur = inner_step(bvh, cur, c.noi, c.noiF, c.inv, ray.tmin, ray.tmax, stack, sp);
            if ((uint32_t)__popcll(__ballot(cur >= 0 && cur != kExitBlas)) < nodeLoopMin) break;
        }
        if (cur == kTraversalDone) break;
        if (cur == kExitBlas || (cur < 0 && c.inst < 0)) { cur = tl_switch(bvh, cur, c, ray, stack, sp); continue; }
        if (cur >= 0) continue;
        const GpuInstance& I = bvh.instances[c.inst];
        const uint32_t enc = (uint32_t)(~cur), first = enc >> 2, n = (enc & 3u) + 1u, inst = (uint32_t)c.inst;
        for (uint32_t i = 0; i < n; ++i) {
            float4 ta, tb, tc; bvh.tri(first + i, ta, tb, tc);
            f3 p0, p1, p2; tl_world_triangle(I, ta, tb, tc, p0, p1, p2);
            float t, u, v;
            if (!tri_test(p0, p1, p2, ray, sh, t, u, v)) continue;
            if (I.flags & 1u) return 0.0f;                                        // opaque instance: committed
            const uint32_t prim = __float_as_uint(tb.w);
            int pos = count;
            if (count == K) {
                float lt; uint32_t ltri, linst; cand.key(K - 1, lt, ltri, linst);
                overflow = true;
                if (!key_less(t, inst, prim, lt, linst, prim_of(ltri))) continue;
                pos = K - 1;
            } else ++count;
            while (pos > 0) {
                float pt; uint32_t ptri, pinst; cand.key(pos - 1, pt, ptri, pinst);
                const bool less = t != pt ? t < pt : key_less(t, inst, prim, pt, pinst, prim_of(ptri));
                if (!less) break;
                cand.move(pos, pos - 1);
                --pos;
            }
            cand.set(pos, t, first + i, inst);
        }
        cur = stack.pop(--sp);
    }
    if (count == 0) return 1.0f;
    ShadowState st; st.transmission = 1.0f; st.inVolume = false; st.inVolumeStartT = 0.0f; st.sigmaT = mk3(0.0f, 0.0f, 0.0f);
    Hit h; h.valid = true; h.opaque = 0; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.inst = 0; h.prim = 0;
    for (int k = 0; k < count; ++k) {
        float kt; cand.key(k, kt, h.tri, h.inst);
        float4 ta, tb, tc; bvh.tri(h.tri, ta, tb, tc);
        f3 p0, p1, p2; tl_world_triangle(bvh.instances[h.inst], ta, tb, tc, p0, p1, p2);
        tri_test(p0, p1, p2, ray, sh, h.t, h.u, h.v);                               // recomputes (t, u, v) bit for bit
        h.prim = __float_as_uint(tb.w);
        if (shadow_candidate_two_level(s, ray, h, st)) return 0.0f;
    }
    if (overflow) {   // more than K candidates: continue behind the K-th with the re-trace loop
        HitKey lower; lower.have = true; lower.t = h.t; lower.inst = h.inst; lower.prim = h.prim;
        for (;;) {
            Hit n = closest_two_level(bvh, s.rootLeaf, s.nodeCount, ray, lower, stack);
            if (!n.valid) break;
            if (n.opaque) return 0.0f;
            if (shadow_candidate_two_level(s, ray, n, st)) return 0.0f;
            lower.t = n.t; lower.inst = n.inst; lower.prim = n.prim;
        }
    }
    return shadow_finish(ray, st);
}
template <class STACK>
HRT_DEV float shadow_query_two_level(const SceneView& s, const GlobalBvhTl& bvh, const Ray& ray, STACK& stack, uint32_t nodeLoopMin = 0)
{
    bool sawNonOpaque;
    if (any_hit_two_level(bvh, s.rootLeaf, s.nodeCount, ray, stack, sawNonOpaque, nodeLoopMin)) return 0.0f;
    return sawNonOpaque ? shadow_retrace_two_level(s, bvh, ray, stack) : 1.0f;
}

// Re-trace form: one closest-hit query per non-opaque candidate (validation megakernel).
// ALL_OPAQUE: the scene is known (upload-time trait) to hold ForceOpaque instances only: the candidate pass is compiled out.
template <bool ALL_OPAQUE = false, class BVH, class STACK>
HRT_DEV float shadow_query(const SceneView& s, const BVH& bvh, f3 worldPos, f3 L, float maxDist, STACK& stack, uint32_t nodeLoopMin = 0)
{
    Ray ray = shadow_ray(worldPos, L, maxDist);
    if constexpr (BVH::kTwoLevel) {
        if constexpr (ALL_OPAQUE) { bool none; return any_hit_two_level(bvh, s.rootLeaf, s.nodeCount, ray, stack, none, nodeLoopMin) ? 0.0f : 1.0f; }
        else return shadow_query_two_level(s, bvh, ray, stack, nodeLoopMin);
    }
    // Any hit on a ForceOpaque instance commits -> 0, whatever lies in front of it.
    bool sawNonOpaque;
    if (any_opaque(bvh, s.rootLeaf, s.nodeCount, ray, stack, sawNonOpaque, nodeLoopMin)) return 0.0f;
    if (ALL_OPAQUE || !sawNonOpaque) return 1.0f;
    ShadowState st; st.transmission = 1.0f; st.inVolume = false; st.inVolumeStartT = 0.0f; st.sigmaT = mk3(0.0f, 0.0f, 0.0f);
    HitKey lower; lower.have = false; lower.t = 0.0f; lower.inst = 0; lower.prim = 0;
    for (;;) {   // non-opaque candidates, front to back
        Hit h = closest_any(bvh, s.rootLeaf, s.nodeCount, ray, lower, stack);
        if (!h.valid) break;
        if (shadow_candidate(s, ray, h.t, h.tri, h.u, h.v, st)) return 0.0f;
        lower.have = true; lower.t = h.t; lower.inst = h.inst; lower.prim = h.prim;
    }
    return shadow_finish(ray, st);
}

// Buffered form (wavefront shadow stage): the any-hit pass over opaque triangles also collects the K closest non-opaque
// candidates, sorted by (t, inst, prim), into a per-lane buffer CAND (LDS columns of (t, triangle)); they are then processed front to
// back without further traversals. More than K candidates: the re-trace loop continues behind the K-th. Same visiting
// order as the re-trace form, hence the same result.
// (t, triangle) of a non-opaque hit into the per-lane buffer of the K smallest keys, kept sorted; `overflow` once a hit did not fit.
template <int K, class BVH, class CAND>
HRT_DEV void candidate_insert(const BVH& bvh, CAND& cand, int& count, bool& overflow, float t, uint32_t tri, uint32_t inst, uint32_t prim)
{
    int pos = count;
    if (count == K) {
        float lt; uint32_t ltri; cand.key(K - 1, lt, ltri);
        float4 la, lb, lc; bvh.tri(ltri, la, lb, lc);
        overflow = true;
        if (!key_less(t, inst, prim, lt, __float_as_uint(la.w), __float_as_uint(lb.w))) return;
        pos = K - 1;
    } else ++count;
    while (pos > 0) {
        float pt; uint32_t ptri; cand.key(pos - 1, pt, ptri);
        bool less;
        if (t != pt) less = t < pt;
        else { float4 pa, pb, pc; bvh.tri(ptri, pa, pb, pc); less = key_less(t, inst, prim, pt, __float_as_uint(pa.w), __float_as_uint(pb.w)); }
        if (!less) break;
        cand.move(pos, pos - 1);
        --pos;
    }
    cand.set(pos, t, tri);
}
// The candidates front to back (CalculateRTShadow's per-candidate body), then -- if more than K existed -- the re-trace loop behind the
// K-th, then the closing Beer-Lambert segment. `cand.key(k, t, tri)` yields the k-th smallest key.
template <class BVH, class STACK, class CAND>
HRT_DEV float shadow_resolve_candidates(const SceneView& s, const BVH& bvh, const Ray& ray, const RayShear& sh, int count, bool overflow, const CAND& cand, STACK& stack)
{
    if (count == 0) return 1.0f;
    ShadowState st; st.transmission = 1.0f; st.inVolume = false; st.inVolumeStartT = 0.0f; st.sigmaT = mk3(0.0f, 0.0f, 0.0f);
    float lastT = 0.0f; uint32_t lastTri = 0;
    for (int k = 0; k < count; ++k) {
        float t; uint32_t tri; cand.key(k, t, tri);
        float4 a, b, c; bvh.tri(tri, a, b, c);
        float t2, u, v;
        tri_test(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), ray, sh, t2, u, v);   // recomputes (t, u, v) bit for bit
        if (shadow_candidate(s, ray, t, tri, u, v, st)) return 0.0f;
        lastT = t; lastTri = tri;
    }
    if (overflow) {   // more than K candidates: continue behind the K-th with the re-trace loop
        float4 la, lb, lc; bvh.tri(lastTri, la, lb, lc);
        HitKey lower; lower.have = true; lower.t = lastT; lower.inst = __float_as_uint(la.w); lower.prim = __float_as_uint(lb.w);
        for (;;) {
            Hit h = closest_any(bvh, s.rootLeaf, s.nodeCount, ray, lower, stack);
            if (!h.valid) break;
            if (h.opaque) return 0.0f;
            if (shadow_candidate(s, ray, h.t, h.tri, h.u, h.v, st)) return 0.0f;
            lower.t = h.t; lower.inst = h.inst; lower.prim = h.prim;
        }
    }
    return shadow_finish(ray, st);
}
template <int K, class BVH, class STACK, class CAND>
HRT_DEV float shadow_query_buffered(const SceneView& s, const BVH& bvh, f3 worldPos, f3 L, float maxDist, STACK& stack, CAND& cand, uint32_t nodeLoopMin = 0)
{
    Ray ray = shadow_ray(worldPos, L, maxDist);
    if (!(ray.d.x == ray.d.x && ray.d.y == ray.d.y && ray.d.z == ray.d.z)) return 1.0f;
    RayShear sh = make_shear(ray.d);
    f3 inv = traversal_rcp(ray.d), noi = slab_origin_term(ray.o, inv);
    int sp = 0, count = 0; bool overflow = false;
    int32_t cur;
    if (s.nodeCount == 0) { if (s.rootLeaf == 0) return 1.0f; cur = s.rootLeaf; } else cur = 0;
    for (;;) {
        while (cur >= 0) {
            cur = inner_step(bvh, cur, noi, inv, ray.tmin, ray.tmax, stack, sp);
            if ((uint32_t)__popcll(__ballot(cur >= 0)) < nodeLoopMin) break;
        }
        if (cur == kTraversalDone) break;
        if (cur >= 0) continue;
        uint32_t enc = (uint32_t)(~cur);
        uint32_t first = enc >> 2, n = (enc & 3u) + 1u;
        for (uint32_t i = 0; i < n; ++i) {
            float4 a, b, c; bvh.tri(first + i, a, b, c);
            float t, u, v;
            if (!tri_test(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), ray, sh, t, u, v)) continue;
            if (__float_as_uint(c.w) & 1u) return 0.0f;                           // opaque instance: committed
            candidate_insert<K>(bvh, cand, count, overflow, t, first + i, __float_as_uint(a.w), __float_as_uint(b.w));
        }
        if (sp == 0) break;
        cur = stack.pop(--sp);
    }
    return shadow_resolve_candidates(s, bvh, ray, sh, count, overflow, cand, stack);
}

// One light of AccumulateDirectLighting (CommonLighting.hlsli:877-908) in three stages, so that a schedule may run the
// expensive parts where lanes are dense and skip them for occluded samples (a shadow factor of 0 contributes +0):
//   nee_draw          early-outs that precede the RNG draws (:723, :758-764, :815-833) + the two draws (:730, :776, :846)
//   nee_direction     jittered direction / distance from the draws; false when dot(N, L_s) <= 0 (no shadow ray, :731/:786/:856)
//   nee_contribution  radiance (sun: atmosphere LUT; point/spot: attenuation) + per-sample byproducts + EvaluateDirectLight
//                     without the shadow factor
// DIRONLY: the scene's light list is known to hold directional lights only (point/spot code compiled out).
template <bool DIRONLY>
HRT_DEV bool nee_draw(const HrptGPULight& l, f3 N, f3 worldPos, f3 sunDirection, uint32_t& rng, float& ux, float& uy)
{
    if (DIRONLY || l.m_Type == HRPT_LIGHT_DIRECTIONAL) {                        // :716-745
        if (dot(N, sunDirection) <= 0.0f) return false;
    } else if (!DIRONLY && (l.m_Type == HRPT_LIGHT_POINT || l.m_Type == HRPT_LIGHT_SPOT)) {   // :752-804, :809-874
        if (l.m_Intensity <= 0.0f) return false;
        f3 toLight = mk3(l.m_Position) - worldPos;
        float distSq = dot(toLight, toLight);
        if (l.m_Range > 0.0f && distSq > l.m_Range * l.m_Range) return false;
        if (l.m_Type == HRPT_LIGHT_SPOT) {
            float dist = hrt_sqrt(distSq);
            f3 Lc = toLight / dist;
            if (dot(N, Lc) <= 0.0f) return false;
            f3 lightDir = normalize(mk3(l.m_Direction));
            float cosTheta = dot(-Lc, lightDir);
            if (cosTheta < hrt_cos(l.m_SpotOuterConeAngle)) return false;
        }
    } else return false;
    ux = hrt_rng_next(&rng); uy = hrt_rng_next(&rng);
    return true;
}

template <bool DIRONLY>
HRT_DEV bool nee_direction(const HrptGPULight& l, f3 N, f3 worldPos, f3 sunDirection, float cosSun, float ux, float uy, f3& L, float& maxDist)
{
    if (DIRONLY || l.m_Type == HRPT_LIGHT_DIRECTIONAL) {
        L = sample_cone(sunDirection, cosSun, ux, uy);
        maxDist = 1e10f;
    } else {
        float cosT = 1.0f - 2.0f * ux;
        float sinT = hrt_sqrt(hrt_max(0.0f, 1.0f - cosT * cosT));
        float phi = 2.0f * HRT_PI * uy;
        float sp, cp; hrt_sincos(phi, &sp, &cp);
        f3 sphereDir = mk3(sinT * cp, cosT, sinT * sp);
        f3 samplePos = mk3(l.m_Position) + sphereDir * l.m_Radius;
        f3 toSample = samplePos - worldPos;
        float sampleDist = length(toSample);
        L = toSample / sampleDist;
        maxDist = sampleDist;
    }
    return !(dot(N, L) <= 0.0f);
}

template <bool DIRONLY>
HRT_DEV void nee_contribution(const SceneView& s, const HrptGPULight& l, Lighting in, f3 worldPos, f3 sunDirection, float sunIntensity, f3 L,
                              f3& diffuse, f3& specular)
{
    f3 radiance;
    if (DIRONLY || l.m_Type == HRPT_LIGHT_DIRECTIONAL) {
        // inputs.sunRadiance (PathTracer.hlsl:137): a pure function of the hit position, read only by directional lights
        radiance = atm::sun_radiance(s, atm::atmosphere_pos(worldPos), sunDirection, sunIntensity);
    } else {
        f3 toLight = mk3(l.m_Position) - worldPos;
        float distSq = dot(toLight, toLight);
        float dist = hrt_sqrt(distSq);
        float att = distance_attenuation(l, distSq, dist);
        f3 col = mk3(l.m_Color);
        if (l.m_Type == HRPT_LIGHT_SPOT) {
            f3 Lc = toLight / dist;
            f3 lightDir = normalize(mk3(l.m_Direction));
            float cosTheta = dot(-Lc, lightDir);
            float cosOuter = hrt_cos(l.m_SpotOuterConeAngle), cosInner = hrt_cos(l.m_SpotInnerConeAngle);
            float spotAtt = hrt_saturate((cosTheta - cosOuter) / (cosInner - cosOuter));
            radiance = ((col * l.m_Intensity) * spotAtt) * att;
        } else radiance = (col * l.m_Intensity) * att;
    }
    in.L = L;
    prepare_byproducts(in);
    evaluate_direct_unshadowed(in, radiance, diffuse, specular);
}

// The surface terms a deferred NEE evaluation needs (what LightingInputs carries into AccumulateDirectLighting).
HRT_DEV Lighting nee_lighting(f3 N, f3 V, f3 baseColor, float roughness, float metallic, float ior)
{
    Lighting in;
    in.N = N; in.V = V; in.L = mk3(0.0f, 0.0f, 0.0f); in.baseColor = baseColor; in.roughness = roughness; in.metallic = metallic; in.ior = ior;
    return in;
}

HRT_DEV HrptGPULight load_light(const SceneView& s, uint32_t i)
{
    if (i < s.lightCount) return s.lights[i];
    HrptGPULight l = HrptGPULight(); l.m_Type = 0;   // out-of-range structured-buffer read returns zeros on D3D12
    return l;
}

enum SurfaceOutcome { SURFACE_TRANSMITTED = 0, SURFACE_SCATTER = 1 };

// PathTracer.hlsl:92-261 up to the light loop. EMIT(lightIndex, ux, uy) receives the two random numbers of every light
// that passed nee_draw (carry is filled before the first call); the caller runs nee_direction, the shadow query and
// nee_contribution, and owes  radiance += throughput * (sum(diffuse_i*shadow_i) + (bounce==0 ? sum(specular_i*shadow_i) : 0)).
// Compile-time scene traits (all-true / DIRONLY=false is the general form and always correct):
//   TEX     some material samples a texture          (false: every m_TextureFlags is 0 -> sampling code compiled out)
//   TRANS   some material is transmissive or BLEND   (false: the transmission branch :149-255 is compiled out)
//   DIRONLY every light is directional               (true: point / spot code compiled out)
template <bool TEX, bool TRANS, bool DIRONLY, class EMIT>
HRT_DEV SurfaceOutcome shade_surface_a(const SceneView& s, const HrptPathTracerConstants& cb, PathState& ps, const Hit& hit,
                                       SurfaceCarry& carry, EMIT&& emit)
{
    TriVerts tv = load_hit_attr(s, hit);                                          // inst/mesh/vertex fetch :92-94,:104 (LOD 0, :103)
    GpuInstShade is = s.instShade[tv.inst];
    const HrptMaterialConstants& mat = s.materials[tv.material];
    uint32_t texFlags = TEX ? mat.m_TextureFlags : 0u;

    if (TRANS && ps.inVolume) {                                                            // Beer-Lambert :97-100
        f3 tr = mk3(hrt_exp(-(ps.sigmaA.x + ps.sigmaS.x) * hit.t), hrt_exp(-(ps.sigmaA.y + ps.sigmaS.y) * hit.t),
                    hrt_exp(-(ps.sigmaA.z + ps.sigmaS.z) * hit.t));
        ps.throughput = ps.throughput * tr;
    }
    SurfaceAttr attr = full_hit_attributes(s, hit, ps.ray, tv, is, (texFlags & HRPT_TEXFLAG_NORMAL) != 0);
    Pbr pbr = pbr_attributes(s, attr, mat, texFlags);

    f3 Ng = normalize(attr.worldNormal);
    f3 N = pbr.normal;
    f3 V = -ps.ray.d;
    bool isFrontFace = dot(Ng, ps.ray.d) < 0.0f;
    if (dot(N, V) < 0.0f) N = -N;

    f3 sunDir = mk3(cb.m_SunDirection[0], cb.m_SunDirection[1], cb.m_SunDirection[2]);

    Lighting in;
    in.N = N; in.V = V; in.L = mk3(0.0f, 0.0f, 0.0f); in.baseColor = pbr.baseColor;
    in.roughness = pbr.roughness; in.metallic = pbr.metallic; in.ior = mat.m_IOR;
    prepare_byproducts(in);                                                       // :142 (L = 0 => H = V)

    HRT_PHASE(PH_SHADE_ATTR);
    if (TRANS && (mat.m_TransmissionFactor > 0.0f || mat.m_AlphaMode == HRPT_ALPHA_MODE_BLEND)) {    // :149-255
        HRT_PHASE(PH_SHADE_TRANS);
        float effectiveAlpha = (mat.m_AlphaMode == HRPT_ALPHA_MODE_BLEND) ? pbr.alpha : 1.0f;
        float transmissionFactor = hrt_max(mat.m_TransmissionFactor, 1.0f - effectiveAlpha);
        float materialIOR = hrt_max(mat.m_IOR, 1.0001f);
        float outsideIOR = ps.inVolume ? ps.interiorIOR : 1.0f;
        float etaSurface = isFrontFace ? (outsideIOR / materialIOR) : (materialIOR / outsideIOR);
        float etaFresnel = etaSurface;
        float etaRefract = (mat.m_IsThinSurface != 0) ? 1.0f : etaFresnel;
        float cosT_geo;
        float F = fresnel_dielectric(etaFresnel, hrt_max(dot(N, V), 0.0f), cosT_geo);
        float probT = hrt_saturate((1.0f - F) * transmissionFactor);
        if (hrt_rng_next(&ps.rng) < probT) {
            f3 refractedDir, bsdfWeight;
            if (pbr.roughness <= 0.08f) {
                refractedDir = refract(ps.ray.d, N, etaRefract);
                if (dot(refractedDir, refractedDir) < 1e-8f) refractedDir = reflect(ps.ray.d, N);
                bsdfWeight = pbr.baseColor;
            } else {
                float ux = hrt_rng_next(&ps.rng), uy = hrt_rng_next(&ps.rng);
                f3 H = sample_ggx_vndf(ux, uy, N, V, pbr.roughness);
                float VdotH = hrt_saturate(dot(V, H));
                float cosT_mf; float F_mf = fresnel_dielectric(etaFresnel, VdotH, cosT_mf);
                float cosT_dir; fresnel_dielectric(etaRefract, VdotH, cosT_dir);
                refractedDir = H * (etaRefract * VdotH - cosT_dir) - V * etaRefract;
                if (dot(refractedDir, refractedDir) < 1e-8f) refractedDir = reflect(ps.ray.d, H);
                refractedDir = normalize(refractedDir);
                float alpha = pbr.roughness * pbr.roughness, alpha2 = alpha * alpha;
                float NdotL_t = hrt_abs(dot(N, refractedDir));
                float G1_t = (NdotL_t > HRT_K_EPSILON) ? 2.0f * NdotL_t / (NdotL_t + hrt_sqrt(alpha2 + (1.0f - alpha2) * NdotL_t * NdotL_t)) : 0.0f;
                bsdfWeight = ((pbr.baseColor * (1.0f - F_mf)) * G1_t) * NdotL_t;
            }
            ps.throughput = ps.throughput * bsdfWeight;
            if (mat.m_IsThinSurface == 0) {
                if (isFrontFace) { ps.inVolume = true; ps.interiorIOR = materialIOR; ps.sigmaA = mk3(mat.m_SigmaA); ps.sigmaS = mk3(mat.m_SigmaS); }
                else { ps.inVolume = false; ps.interiorIOR = 1.0f; ps.sigmaA = mk3(0.0f, 0.0f, 0.0f); ps.sigmaS = mk3(0.0f, 0.0f, 0.0f); }
            }
            ps.ray.o = attr.worldPos - N * 0.001f;
            ps.ray.d = normalize(refractedDir);
            ps.ray.tmin = 1e-4f; ps.ray.tmax = 1e10f;
            return SURFACE_TRANSMITTED;
        }
    }

    ps.radiance = ps.radiance + ps.throughput * pbr.emissive;                     // :258

    carry.N = N; carry.V = V; carry.worldPos = attr.worldPos; carry.baseColor = pbr.baseColor; carry.F0 = in.F0;
    carry.Fr = in.F.x; carry.roughness = pbr.roughness; carry.metallic = pbr.metallic; carry.ior = mat.m_IOR;
    carry.rngBeforeLights = ps.rng; carry.material = tv.material;
    HRT_PHASE(PH_SHADE_NEE);
    for (uint32_t i = 0; i < cb.m_LightCount; ++i) {                              // AccumulateDirectLighting :260
        HrptGPULight l = load_light(s, i);
        float ux, uy;
        if (nee_draw<DIRONLY>(l, N, attr.worldPos, sunDir, ps.rng, ux, uy)) emit(i, ux, uy);
    }
    return SURFACE_SCATTER;
}

// PathTracer.hlsl:264-313 in three pieces so that a wave can run the (rare, long) specular lobe for many paths at once:
//   lobe_begin     Russian roulette :264-270, lobe pick :275-280 and what both lobes share (two draws, one sincos, one sqrt)
//   lobe_diffuse   cosine-weighted sample + weight :295-304, ray advance :306-313
//   lobe_specular  GGX-VNDF sample + weight :281-294, ray advance :306-313
// shade_surface_b strings them together (validation megakernel, general wavefront variants). Values and operation order per path
// are those of the reference in every arrangement.
struct LobeDraw { float specProb, root, sp, cp; bool spec; };

HRT_DEV bool lobe_begin(PathState& ps, const SurfaceCarry& c, int bounce, LobeDraw& ld)
{
    HRT_PHASE(PH_SHADE_LOBE_BEGIN);
    if (bounce >= 2) {                                                            // Russian roulette :264-270
        float continuePr = hrt_saturate(maxcomp(ps.throughput));
        if (hrt_rng_next(&ps.rng) > continuePr) return false;
        ps.throughput = ps.throughput / continuePr;
    }
    ld.specProb = hrt_clamp(lerp(c.Fr * 0.5f + 0.5f * c.metallic, 1.0f, c.metallic), 0.1f, 0.9f);   // :275
    ld.spec = hrt_rng_next(&ps.rng) < ld.specProb;
    const float ux = hrt_rng_next(&ps.rng), uy = hrt_rng_next(&ps.rng);
    hrt_sincos(2.0f * HRT_PI * (ld.spec ? uy : ux), &ld.sp, &ld.cp);
    ld.root = hrt_sqrt(ld.spec ? ux : uy);
    return true;
}
HRT_DEV bool lobe_finish(PathState& ps, f3 worldPos, f3 N, f3 newDir, f3 numer, float denom)
{
    if (dot(N, newDir) <= 0.0f) return false;
    f3 brdfWeight = numer / denom;
    ps.throughput = ps.throughput * brdfWeight;
    if (maxcomp(ps.throughput) < 0.01f) return false;                             // :306
    ps.ray.o = worldPos; ps.ray.d = newDir; ps.ray.tmin = 1e-4f; ps.ray.tmax = 1e10f;   // :310-313
    return true;
}
HRT_DEV bool lobe_diffuse(PathState& ps, f3 worldPos, f3 N, f3 baseColor, float metallic, const LobeDraw& ld)
{
    HRT_PHASE(PH_SHADE_DIFFUSE);
    f3 T, B; tangent_frame(N, T, B);
    f3 newDir = sample_hemisphere_cosine_from(ld.root, ld.sp, ld.cp, T, B, N);
    return lobe_finish(ps, worldPos, N, newDir, baseColor * (1.0f - metallic), 1.0f - ld.specProb);
}
HRT_DEV bool lobe_specular(PathState& ps, f3 worldPos, f3 N, f3 V, f3 F0, float roughness, const LobeDraw& ld)
{
    HRT_PHASE(PH_SHADE_SPEC);
    f3 T, B; tangent_frame(N, T, B);
    f3 H = sample_ggx_vndf_from(ld.root, ld.sp, ld.cp, T, B, N, V, roughness);
    f3 newDir = reflect(-V, H);
    if (dot(N, newDir) <= 0.0f) return false;       // before the weight, as in the reference (:289)
    return lobe_finish(ps, worldPos, N, newDir, eval_ggx_vndf_weight(F0, N, V, newDir, H, roughness), ld.specProb);
}
HRT_DEV bool shade_surface_b(PathState& ps, const SurfaceCarry& c, int bounce)
{
    LobeDraw ld;
    if (!lobe_begin(ps, c, bounce, ld)) return false;
    return ld.spec ? lobe_specular(ps, c.worldPos, c.N, c.V, c.F0, c.roughness, ld) : lobe_diffuse(ps, c.worldPos, c.N, c.baseColor, c.metallic, ld);
}

// PathTracer.hlsl:315-328
HRT_DEV void miss_sky(const SceneView& s, const HrptPathTracerConstants& cb, PathState& ps, int bounce)
{
    HRT_PHASE(PH_SHADE_SKY);
    f3 sunDir = mk3(cb.m_SunDirection[0], cb.m_SunDirection[1], cb.m_SunDirection[2]);
    f3 sky = atm::sky_radiance(s, ps.ray.o, ps.ray.d, sunDir, s.lights[0].m_Intensity, bounce == 0);
    ps.radiance = ps.radiance + ps.throughput * sky;
}

} // namespace hrt
