// pt_post.hip -- HDR post chain, the consumer of the path-tracer pass: three small HBM-bound kernels
// following /root/reference/src/shaders/{LuminanceHistogram.hlsl, ExposureAdaptation.hlsl, Tonemap.hlsl} as driven by
// HDRRenderer::Render (src/HDRRenderer.cpp:88-224). Integer histogram (LDS atomics per block, one global add per bin per
// block), a single 256-thread adaptation block that reduces with the reference's shared-memory tree (same order => same
// bits as the oracle), and a streaming tonemap (float4 in, float4 out).
#include "pt_kernels.h"
#include "pt_device.h"

namespace hrt {

namespace {
constexpr float kMinLog = -10.0f, kMaxLog = 20.0f;   // src/HDRRenderer.cpp:12-13

__global__ __launch_bounds__(256) void post_histogram(const float4* __restrict__ hdr, uint32_t n, uint32_t* __restrict__ histogram)
{
    __shared__ uint32_t local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 c = hdr[i];
        float lum = dot(mk3(c.x, c.y, c.z), mk3(0.2126f, 0.7152f, 0.0722f));       // GetLuminance
        uint32_t bin = 0;                                                             // ColorToBin
        if (!(lum < 0.0001f)) {
            float range = kMaxLog - kMinLog;
            float logLum = hrt_clamp((hrt_log2(lum) - kMinLog) / range, 0.0f, 1.0f);
            bin = (uint32_t)(logLum * 254.0f + 1.0f);
        }
        atomicAdd(&local[bin], 1u);
    }
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&histogram[threadIdx.x], local[threadIdx.x]);
}

__global__ __launch_bounds__(256) void post_adaptation(const uint32_t* __restrict__ histogram, HrptPostParams p, uint32_t numPixels, float* __restrict__ exposure)
{
    __shared__ float w[256];
    uint32_t t = threadIdx.x;
    float range = kMaxLog - kMinLog;
    float logLum = t == 0 ? kMinLog : (kMinLog + ((float)(t - 1) / 254.0f) * range);
    w[t] = (float)histogram[t] * logLum;
    __syncthreads();
    for (uint32_t i = 128; i > 0; i >>= 1) {
        if (t < i) w[t] += w[t + i];
        __syncthreads();
    }
    if (t == 0) {
        float avgLogLum = w[0] / hrt_max((float)numPixels, 1.0f);
        float avgLum = hrt_exp2(avgLogLum);
        float EV100 = hrt_log2(avgLum * 100.0f / 12.5f);
        EV100 = hrt_clamp(EV100, p.exposureValueMin, p.exposureValueMax);
        EV100 -= p.exposureCompensation;
        float target = 1.0f / (hrt_pow(2.0f, EV100) * 1.2f);
        float cur = exposure[0];
        exposure[0] = cur + (target - cur) * (1.0f - hrt_exp(-p.deltaTimeSeconds * p.adaptationSpeed));
    }
}

__global__ void post_set_exposure(float* exposure, float v) { if (threadIdx.x == 0 && blockIdx.x == 0) exposure[0] = v; }

HRT_DEV f3 pbr_neutral(f3 c)                                                          // Tonemap.hlsl:13-33
{
    const float startCompression = 0.8f - 0.04f, desaturation = 0.15f;
    float x = hrt_min(c.x, hrt_min(c.y, c.z));
    float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    c = mk3(c.x - offset, c.y - offset, c.z - offset);
    float peak = hrt_max(c.x, hrt_max(c.y, c.z));
    if (peak < startCompression) return c;
    const float d = 1.0f - startCompression;
    float newPeak = 1.0f - d * d / (peak + d - startCompression);
    float k = newPeak / peak;
    c = c * k;
    float g = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
    return mk3(lerp(c.x, newPeak * 1.0f, g), lerp(c.y, newPeak * 1.0f, g), lerp(c.z, newPeak * 1.0f, g));
}
HRT_DEV float srgb_oetf(float x)                                                      // Tonemap.hlsl:35-42
{
    float v = (x <= 0.0031308f) ? x * 12.92f : 1.055f * hrt_pow(x, 1.0f / 2.4f) - 0.055f;
    return hrt_saturate(v);
}
HRT_DEV f3 hdr_display_tonemap(f3 x, float maxNits)                                   // Tonemap.hlsl:72-92
{
    float maxSCRGB = maxNits / 80.0f;
    float lum = hrt_max(x.x, hrt_max(x.y, x.z));
    if (lum <= 1.0f) return x;
    float headroom = maxSCRGB - 1.0f, excess = lum - 1.0f;
    float compressed = excess * headroom / (excess + headroom);
    float newLum = 1.0f + compressed;
    float k = newLum / lum;
    return mk3(hrt_min(x.x * k, maxSCRGB), hrt_min(x.y * k, maxSCRGB), hrt_min(x.z * k, maxSCRGB));
}

__global__ __launch_bounds__(256) void post_tonemap(const float4* __restrict__ hdr, float4* __restrict__ display, uint32_t n,
                                                    const float* __restrict__ exposure, uint32_t hdrDisplay, float maxNits)
{
    float e = exposure[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 c4 = hdr[i];
        f3 c = mk3(c4.x, c4.y, c4.z) * e;
        f3 o;
        if (hdrDisplay) o = hdr_display_tonemap(c, maxNits);
        else { f3 t = pbr_neutral(c); o = mk3(srgb_oetf(t.x), srgb_oetf(t.y), srgb_oetf(t.z)); }
        display[i] = make_float4(o.x, o.y, o.z, 1.0f);
    }
}
} // namespace

hipError_t launch_post_chain(const float4* hdr, float4* display, uint32_t pixelCount, const HrptPostParams& p, uint32_t* histogram,
                             float* exposure, hipStream_t stream)
{
    if (pixelCount == 0) return hipSuccess;
    uint32_t blocks = (pixelCount + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipError_t e = hipMemsetAsync(histogram, 0, 256 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (p.autoExposure) {
        hipLaunchKernelGGL(post_histogram, dim3(blocks), dim3(256), 0, stream, hdr, pixelCount, histogram);
        hipLaunchKernelGGL(post_adaptation, dim3(1), dim3(256), 0, stream, histogram, p, pixelCount, exposure);
    } else hipLaunchKernelGGL(post_set_exposure, dim3(1), dim3(64), 0, stream, exposure, p.manualExposure);
    hipLaunchKernelGGL(post_tonemap, dim3(blocks), dim3(256), 0, stream, hdr, display, pixelCount, exposure, p.hdrDisplay, p.maxDisplayNits);
    return hipGetLastError();
}

} // namespace hrt
