/*
 * hobbyrt/detmath.h -- the numeric contract of the path-tracer boundary.
 *
 * The reference path tracer is an HLSL SM 6.8 compute shader
 * (/root/reference/src/shaders/PathTracer.hlsl:53-340). Its intrinsics (sin, cos,
 * exp, sqrt, saturate, min, max ...) are evaluated by whatever the D3D12 driver
 * emits, so there is no bit pattern to match. A stochastic path tracer however
 * branches on those values (Russian roulette PathTracer.hlsl:267, lobe pick :280,
 * transmission pick :169): one ulp flips a path. To make "same Scene + same RNG
 * stream => same radiance" a testable statement, every implementation of this
 * boundary (the HIP kernels and the CPU oracle) evaluates the HLSL intrinsics with
 * the functions below, which use only IEEE-754 binary32 + - * / sqrt, comparisons
 * and integer ops -- no FMA contraction (build with -ffp-contract=off), no libm,
 * no fast-math. They are bit-reproducible on x86-64 (gcc) and gfx950 (hipcc).
 *
 * This header is plain C99 and HIP-compatible; it holds scalar intrinsics only.
 * Vector algebra, shading, traversal and LUT sampling are written separately in
 * the product (hobbyrenderer_amd/csrc) and in the oracle (oracle/).
 */
#ifndef HOBBYRT_DETMATH_H
#define HOBBYRT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define HRT_FN __host__ __device__ static __inline__ __attribute__((always_inline))
#else
#define HRT_FN static inline
#endif

#define HRT_PI 3.14159265359f /* srrhi::CommonConsts::PI, shaders/Common.sr:48 */
#define HRT_K_EPSILON 1e-5f /* srrhi::CommonConsts::kEpsilon, shaders/Common.sr:50 ("general-purpose epsilon for division/sqrt guards"); PathTracer.hlsl:218 */

HRT_FN float hrt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
HRT_FN uint32_t hrt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

/* HLSL min/max: if one operand is NaN the other is returned. Select form so that
 * signed zeros are handled identically on every target. */
HRT_FN float hrt_min(float a, float b) { return (a <= b || b != b) ? a : b; }
HRT_FN float hrt_max(float a, float b) { return (a >= b || b != b) ? a : b; }
HRT_FN float hrt_clamp(float x, float lo, float hi) { return hrt_min(hrt_max(x, lo), hi); }
/* saturate(NaN) == 0 (DXIL Saturate). */
HRT_FN float hrt_saturate(float x) { return (x > 0.0f) ? ((x < 1.0f) ? x : 1.0f) : 0.0f; }
HRT_FN float hrt_abs(float x) { return hrt_u2f(hrt_f2u(x) & 0x7fffffffu); }

/* IEEE correctly rounded (sqrtss on x86; hipcc -fhip-fp32-correctly-rounded-divide-sqrt). */
HRT_FN float hrt_sqrt(float x) { return __builtin_sqrtf(x); }
/* HLSL rsqrt(x) is specified here as 1/sqrt(x) with two correctly rounded ops. */
HRT_FN float hrt_rsqrt(float x) { return 1.0f / __builtin_sqrtf(x); }
HRT_FN float hrt_floor(float x) { return __builtin_floorf(x); }

/* x^5 by repeated multiplication: pow(x, 5.0) in CommonLighting.hlsli:127,157,158. */
HRT_FN float hrt_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

/* Quadrant reduction by pi/2, Cody-Waite 3 terms. fn*DP1 and fn*DP2 are exact for
 * |fn| < 2^13, which covers every argument on the path (|x| <= 2*pi*(1+eps) and
 * light cone angles). */
HRT_FN float hrt_trig_reduce(float x, int* q)
{
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float DP1 = 1.5703125f;
    const float DP2 = 4.837512969970703125e-4f;
    const float DP3 = 7.54978995489188216e-8f;
    float fn = __builtin_floorf(x * TWO_OVER_PI + 0.5f);
    *q = (int)fn;
    return ((x - fn * DP1) - fn * DP2) - fn * DP3;
}
HRT_FN float hrt_sin_poly(float r)
{
    float z = r * r;
    return r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * (-1.9515295891e-4f)));
}
HRT_FN float hrt_cos_poly(float r)
{
    float z = r * r;
    return (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
}
HRT_FN float hrt_sin(float x)
{
    int q; float r = hrt_trig_reduce(x, &q);
    float s = hrt_sin_poly(r), c = hrt_cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
HRT_FN float hrt_cos(float x)
{
    int q; float r = hrt_trig_reduce(x, &q);
    float s = hrt_sin_poly(r), c = hrt_cos_poly(r);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}
/* sin and cos of the same angle with one reduction (phi = 2*pi*u sites). */
HRT_FN void hrt_sincos(float x, float* sn, float* cs)
{
    int q; float r = hrt_trig_reduce(x, &q);
    float s = hrt_sin_poly(r), c = hrt_cos_poly(r);
    float vs = (q & 1) ? c : s;
    float vc = (q & 1) ? s : c;
    *sn = (q & 2) ? -vs : vs;
    *cs = ((q + 1) & 2) ? -vc : vc;
}

/* exp(x), flushing results below 2^-126 to zero. */
HRT_FN float hrt_exp(float x)
{
    if (x != x) return x;
    if (x > 88.7228317f) return hrt_u2f(0x7f800000u);
    if (x < -87.3365448f) return 0.0f;
    float fn = __builtin_floorf(x * 1.44269504088896341f + 0.5f);
    float r = (x - fn * 0.693359375f) - fn * (-2.12194440e-4f);
    float z = r * r;
    float p = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r
                 + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f) * z + r + 1.0f;
    int n = (int)fn;                   /* -126 .. 128 */
    if (n > 127) { p = p * 2.0f; n = n - 1; }
    if (n < -126) return 0.0f;
    return p * hrt_u2f((uint32_t)(n + 127) << 23);
}

/* log2(x) (cephes log2f polynomial): -inf at 0, NaN below; subnormal inputs are scaled first. Used by the HDR post
 * chain (LuminanceHistogram.hlsl, ExposureAdaptation.hlsl, Tonemap.hlsl), not by the path tracer itself. */
HRT_FN float hrt_log2(float x)
{
    if (x != x || x < 0.0f) return hrt_u2f(0x7fc00000u);
    if (x == 0.0f) return hrt_u2f(0xff800000u);
    uint32_t u = hrt_f2u(x);
    if (u >= 0x7f800000u) return x;
    int e = 0;
    if (u < 0x00800000u) { x = x * 8388608.0f; u = hrt_f2u(x); e = -23; }
    e += (int)(u >> 23) - 126;                                      /* x = m * 2^e, m in [0.5, 1) */
    float m = hrt_u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                    + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;                                               /* ln(1+m) - m */
    const float LOG2EA = 0.44269504088896340736f;                   /* log2(e) - 1 */
    float r = y * LOG2EA;
    r = r + m * LOG2EA;
    r = r + y;
    r = r + m;
    return r + (float)e;
}
/* exp2(x) (cephes exp2f polynomial); results below 2^-126 flush to zero. */
HRT_FN float hrt_exp2(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return hrt_u2f(0x7f800000u);
    if (x < -126.0f) return 0.0f;
    float px = __builtin_floorf(x);
    int n = (int)px;
    float r = x - px;
    if (r > 0.5f) { n += 1; r = r - 1.0f; }
    float p = (((((1.535336188319500e-4f * r + 1.339887440266574e-3f) * r + 9.618437357674640e-3f) * r + 5.550332471162809e-2f) * r
                + 2.402264791363012e-1f) * r + 6.931472028550421e-1f) * r + 1.0f;
    if (n > 127) { p = p * 2.0f; n -= 1; }
    if (n < -126) return 0.0f;
    return p * hrt_u2f((uint32_t)(n + 127) << 23);
}
/* HLSL pow(x, y) = exp2(y * log2(x)) (x > 0; pow(0, y>0) = 0). */
HRT_FN float hrt_pow(float x, float y) { return hrt_exp2(y * hrt_log2(x)); }

/* HLSL f16tof32 of the low 16 bits (MeshCommon.hlsli:20; RGBA16F LUT texels). Exact. */
HRT_FN float hrt_f16tof32(uint32_t h)
{
    uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return hrt_u2f(s);
        /* subnormal half: m * 2^-24, exact in binary32 */
        float f = (float)m * hrt_u2f(0x33800000u);
        return (s != 0) ? -f : f;
    }
    if (e == 31) return hrt_u2f(s | 0x7f800000u | (m << 13));
    return hrt_u2f(s | ((e + 112u) << 23) | (m << 13));
}

/* RNG.hlsli:14-33 -- integer exact. */
HRT_FN uint32_t hrt_pcg_hash(uint32_t seed)
{
    uint32_t state = seed * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
HRT_FN uint32_t hrt_rng_seed(uint32_t px, uint32_t py, uint32_t accumIndex)
{
    return hrt_pcg_hash(px + py * 65536u + accumIndex * 6700417u);
}
/* float(state) * 2^-32; may return exactly 1.0f (state >= 0xFFFFFF80), kept as in the reference. */
HRT_FN float hrt_rng_next(uint32_t* state)
{
    *state = hrt_pcg_hash(*state);
    return (float)(*state) * (1.0f / 4294967296.0f);
}

#endif /* HOBBYRT_DETMATH_H */
