/*
 * oracle/pt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C) of the reference path tracer
 * /root/reference/src/shaders/PathTracer.hlsl:53-340 and the includes it pulls in.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference snapshot holds no tests, golden
 * images or KATs for this path and its implementation
 * (HLSL SM 6.8 + DXR inline ray queries on D3D12) cannot be built or run here. The
 * oracle is pinned only by the KATs derivable from the reference text (RNG.hlsli,
 * Utilities.cpp Halton; tests/test_oracle_kat.py) and by analytic properties.
 *
 * The oracle takes the SAME input structs as the product C ABI (include/hobbyrt_pt.h
 * declares the reference's structured-buffer layouts); it shares no code with the
 * product except include/hobbyrt/detmath.h (the scalar intrinsic contract).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/hobbyrt_pt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrContext OrContext;

typedef struct OrStats {
    uint64_t closestRays;      /* TraceRayStandard calls */
    uint64_t shadowRays;       /* CalculateRTShadow<true> calls */
    uint64_t paths;
    uint64_t closestNodes;     /* BVH nodes visited by closest-hit queries (n of SURVEY 8d) */
    uint64_t closestTris;      /* triangles tested by closest-hit queries (t) */
    uint64_t shadowNodes;
    uint64_t shadowTris;
    uint64_t retraces;         /* extra traversals caused by rejected non-opaque candidates */
} OrStats;

/* Copies the scene, pre-transforms triangles to world space, builds the oracle's own BVH2,
 * converts LUTs to fp16. Returns NULL on invalid input (message via or_last_error). */
OrContext* or_create(const HrptSceneDesc* scene);
void or_destroy(OrContext* ctx);
const char* or_last_error(void);

/* One dispatch of PathTracer_CSMain over pixels [x0,x1) x [y0,y1) (0,0,0,0 = full viewport).
 * accumulation / output: width*height float4 images owned by the caller (u1 / u0).
 * bruteForce != 0 bypasses the BVH (tests the BVH-independence of the hit definition).
 * nthreads <= 0 selects all online cores. stats are ADDED to *stats when non-NULL. */
int or_render(OrContext* ctx, const HrptPathTracerConstants* cb,
              float* accumulation, float* output,
              uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
              int nthreads, int bruteForce, OrStats* stats);

/* Host logic of PathTracerRenderer::Render (src/PathTracerRenderer.cpp:58-75). */
float or_halton(uint32_t index, uint32_t base);                       /* src/Utilities.cpp:67-79 */
void  or_fill_constants(HrptPathTracerConstants* cb, const HrptPlanarViewConstants* view,
                        const float cameraPos[3], uint32_t lightCount, uint32_t accumulationIndex,
                        uint32_t frameIndex, uint32_t maxBounces, const float sunDirection[3],
                        float sunAngularSizeDeg);

/* HDR post chain: LuminanceHistogram.hlsl + ExposureAdaptation.hlsl (or manual exposure) +
 * Tonemap.hlsl over a W x H float4 HDR image. *exposure is the persistent exposure buffer (in/out). */
void  or_post_process(const float* hdr, uint32_t width, uint32_t height, const HrptPostParams* params, float* exposure,
                      uint32_t histogram[256], float* display);
float or_log2(float x); float or_exp2(float x); float or_pow(float x, float y);

/* Scalar probes for known-answer tests. */
uint32_t or_pcg_hash(uint32_t v);
uint32_t or_init_rng(uint32_t px, uint32_t py, uint32_t accumIndex);
float    or_next_float(uint32_t* state);
float    or_fresnel_dielectric(float eta, float cosThetaI, float* cosThetaT); /* PathTracer.hlsl:26-43 */
void     or_unpack_vertex(const HrptVertexQuantized* vq, float out12[12]);    /* MeshCommon.hlsli:9-22: pos3 normal3 uv2 tangent4 */
uint16_t or_float_to_half(float f);
float    or_half_to_float(uint16_t h);
float    or_sin(float x); float or_cos(float x); float or_exp(float x);
/* one closest-hit query against the context BVH (or brute force); returns 1 on hit */
int      or_trace_closest(OrContext* ctx, const float origin[3], const float dir[3], float tmin, float tmax,
                          int bruteForce, uint32_t* inst, uint32_t* prim, float bary[2], float* t);
/* TraceRayStandard / CalculateRTShadow<true> as stand-alone queries (checkers for hrpt_trace_rays); 1 on hit */
int      or_trace_standard(OrContext* ctx, const float origin[3], const float dir[3], float tmin, float tmax, uint32_t* rng,
                           uint32_t* inst, uint32_t* prim, float bary[2], float* t);
float    or_shadow_query(OrContext* ctx, const float worldPos[3], const float L[3], float maxDist);
/* GetAtmosphereSkyRadiance / GetAtmosphereSunRadiance (Atmosphere.hlsli:569-601) */
void     or_sky_radiance(OrContext* ctx, const float cameraPos[3], const float viewRay[3], const float sunDir[3],
                         float sunIntensity, int addSunDisk, float out[3]);
void     or_sun_radiance(OrContext* ctx, const float worldPos[3], const float sunDir[3], float sunIntensity, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
