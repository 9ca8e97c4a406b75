// atmosphere_precompute.cpp -- host-side producer of the three Bruneton LUTs the reference loads from
// bin/bruneton/{transmittance,scattering,irradiance}.dat (/root/reference/src/CommonResources.cpp:519-569).
// Those files are git-ignored build artefacts of the reference (.gitignore:46) and their producer is not in
// its tree, so they are Scene INPUTS here: this file generates stand-ins in the same raw-float32-RGBA layout
// from the constants of src/shaders/Atmosphere.hlsli:41-75, following the published precomputation of
// E. Bruneton, "Precomputed Atmospheric Scattering" (2017 implementation): transmittance, single
// Rayleigh+Mie scattering (combined texture, Mie.r in alpha) and direct ground irradiance = 0 + nothing
// (the irradiance LUT is not read on the path-tracer path). Higher scattering orders are NOT computed:
// parity of the LUT contents with the reference's files is unpinned; whatever is fed
// in is shared bit-identically by the HIP path and the oracle.
//
// Arithmetic: binary32 with include/hobbyrt/detmath.h so the tables are identical on every host.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/hobbyrt_pt.h"
#include "../../include/hobbyrt/detmath.h"

namespace {

constexpr float kBottom = 6360.0f, kTop = 6420.0f, kMuSMin = -0.207912f;
constexpr float kSunAngularRadius = 0.004675f;
constexpr float kSolar[3] = { 1.474000f, 1.850400f, 1.911980f };
constexpr float kRayleigh[3] = { 0.005802f, 0.013558f, 0.033100f };
constexpr float kMieScattering[3] = { 0.003996f, 0.003996f, 0.003996f };
constexpr float kMieExtinction[3] = { 0.004440f, 0.004440f, 0.004440f };
constexpr float kAbsorption[3] = { 0.000650f, 0.001881f, 0.000085f };
constexpr int TW = 256, TH = 64, SW = 256, SH = 128, SD = 32, NU = 8, MUS = 32, MU = 128, RS = 32;

struct Layer { float width, exp_term, exp_scale, linear_term, constant_term; };
constexpr Layer kRayleighLayer = { 0.0f, 1.0f, -1.0f / 8.0f, 0.0f, 0.0f };
constexpr Layer kMieLayer = { 0.0f, 1.0f, -1.0f / 1.2f, 0.0f, 0.0f };
constexpr Layer kOzone0 = { 25.0f, 0.0f, 0.0f, 1.0f / 15.0f, -2.0f / 3.0f };
constexpr Layer kOzone1 = { 0.0f, 0.0f, 0.0f, -1.0f / 15.0f, 8.0f / 3.0f };

inline float clampf(float x, float a, float b) { return hrt_clamp(x, a, b); }
inline float safe_sqrt(float a) { return hrt_sqrt(hrt_max(a, 0.0f)); }
inline float layer_density(const Layer& l, float alt) { return clampf(l.exp_term * hrt_exp(l.exp_scale * alt) + l.linear_term * alt + l.constant_term, 0.0f, 1.0f); }
inline float ozone_density(float alt) { return alt < kOzone0.width ? layer_density(kOzone0, alt) : layer_density(kOzone1, alt); }
inline float dist_top(float r, float mu) { return hrt_max(-r * mu + safe_sqrt(r * r * (mu * mu - 1.0f) + kTop * kTop), 0.0f); }
inline float dist_bottom(float r, float mu) { return hrt_max(-r * mu - safe_sqrt(r * r * (mu * mu - 1.0f) + kBottom * kBottom), 0.0f); }
inline bool hits_ground(float r, float mu) { return mu < 0.0f && r * r * (mu * mu - 1.0f) + kBottom * kBottom >= 0.0f; }
inline float texcoord(float x, int n) { return 0.5f / (float)n + x * (1.0f - 1.0f / (float)n); }
inline float unit_range(float u, int n) { return (u - 0.5f / (float)n) / (1.0f - 1.0f / (float)n); }

template <class D>
float optical_length(D density, float r, float mu)
{
    const int N = 500;
    float dx = dist_top(r, mu) / (float)N, result = 0.0f;
    for (int i = 0; i <= N; ++i) {
        float d = (float)i * dx;
        float ri = hrt_sqrt(d * d + 2.0f * r * mu * d + r * r);
        float y = density(ri - kBottom);
        result += y * ((i == 0 || i == N) ? 0.5f : 1.0f) * dx;
    }
    return result;
}

void compute_transmittance(float* T)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    for (int y = 0; y < TH; ++y)
        for (int x = 0; x < TW; ++x) {
            float x_mu = unit_range(((float)x + 0.5f) / (float)TW, TW), x_r = unit_range(((float)y + 0.5f) / (float)TH, TH);
            float rho = H * x_r;
            float r = hrt_sqrt(rho * rho + kBottom * kBottom);
            float d_min = kTop - r, d_max = rho + H;
            float d = d_min + x_mu * (d_max - d_min);
            float mu = d == 0.0f ? 1.0f : (H * H - rho * rho - d * d) / (2.0f * r * d);
            mu = clampf(mu, -1.0f, 1.0f);
            float olR = optical_length([](float a) { return layer_density(kRayleighLayer, a); }, r, mu);
            float olM = optical_length([](float a) { return layer_density(kMieLayer, a); }, r, mu);
            float olO = optical_length([](float a) { return ozone_density(a); }, r, mu);
            float* o = T + ((size_t)y * TW + x) * 4;
            for (int c = 0; c < 3; ++c) o[c] = hrt_exp(-(kRayleigh[c] * olR + kMieExtinction[c] * olM + kAbsorption[c] * olO));
            o[3] = 1.0f;
        }
}

struct Rgb { float v[3]; };

// bilinear fp32 lookup in the float transmittance table (clamp addressing)
Rgb lookup_transmittance(const float* T, float r, float mu)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    float rho = safe_sqrt(r * r - kBottom * kBottom);
    float d = dist_top(r, mu);
    float d_min = kTop - r, d_max = rho + H;
    float x_mu = (d - d_min) / (d_max - d_min), x_r = rho / H;
    float u = texcoord(x_mu, TW), v = texcoord(x_r, TH);
    float fx = u * (float)TW - 0.5f, fy = v * (float)TH - 0.5f;
    float ix = hrt_floor(fx), iy = hrt_floor(fy);
    float tx = fx - ix, ty = fy - iy;
    auto cl = [](int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); };
    int x0 = cl((int)ix, TW), x1 = cl((int)ix + 1, TW), y0 = cl((int)iy, TH), y1 = cl((int)iy + 1, TH);
    Rgb o;
    for (int c = 0; c < 3; ++c) {
        float a = T[((size_t)y0 * TW + x0) * 4 + c] * (1.0f - tx) + T[((size_t)y0 * TW + x1) * 4 + c] * tx;
        float b = T[((size_t)y1 * TW + x0) * 4 + c] * (1.0f - tx) + T[((size_t)y1 * TW + x1) * 4 + c] * tx;
        o.v[c] = a * (1.0f - ty) + b * ty;
    }
    return o;
}
Rgb get_transmittance(const float* T, float r, float mu, float d, bool ground)
{
    float r_d = clampf(hrt_sqrt(d * d + 2.0f * r * mu * d + r * r), kBottom, kTop);
    float mu_d = clampf((r * mu + d) / r_d, -1.0f, 1.0f);
    Rgb a, b, o;
    if (ground) { a = lookup_transmittance(T, r_d, -mu_d); b = lookup_transmittance(T, r, -mu); }
    else { a = lookup_transmittance(T, r, mu); b = lookup_transmittance(T, r_d, mu_d); }
    for (int c = 0; c < 3; ++c) o.v[c] = hrt_min(a.v[c] / b.v[c], 1.0f);
    return o;
}
Rgb transmittance_to_sun(const float* T, float r, float mu_s)
{
    float s = kBottom / r;
    float ch = -hrt_sqrt(hrt_max(1.0f - s * s, 0.0f));
    float e = s * kSunAngularRadius;
    float t = hrt_saturate(((mu_s - ch) + e) / (e + e));
    float f = (t * t) * (3.0f - 2.0f * t);
    Rgb o = lookup_transmittance(T, r, mu_s);
    for (int c = 0; c < 3; ++c) o.v[c] *= f;
    return o;
}

void scattering_slice(const float* T, float* S, int z)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    for (int y = 0; y < SH; ++y)
        for (int x = 0; x < SW; ++x) {
            // GetRMuMuSNuFromScatteringTextureFragCoord
            float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f, fcz = (float)z + 0.5f;
            float frag_nu = hrt_floor(fcx / (float)MUS);
            float frag_mu_s = fcx - frag_nu * (float)MUS;
            float uw_nu = frag_nu / (float)(NU - 1), uw_mus = frag_mu_s / (float)MUS, uw_mu = fcy / (float)MU, uw_r = fcz / (float)RS;
            float rho = H * unit_range(uw_r, RS);
            float r = hrt_sqrt(rho * rho + kBottom * kBottom);
            float mu; bool ground;
            if (uw_mu < 0.5f) {
                float d_min = r - kBottom, d_max = rho;
                float d = d_min + (d_max - d_min) * unit_range(1.0f - 2.0f * uw_mu, MU / 2);
                mu = d == 0.0f ? -1.0f : clampf(-(rho * rho + d * d) / (2.0f * r * d), -1.0f, 1.0f);
                ground = true;
            } else {
                float d_min = kTop - r, d_max = rho + H;
                float d = d_min + (d_max - d_min) * unit_range(2.0f * uw_mu - 1.0f, MU / 2);
                mu = d == 0.0f ? 1.0f : clampf((H * H - rho * rho - d * d) / (2.0f * r * d), -1.0f, 1.0f);
                ground = false;
            }
            float x_mu_s = unit_range(uw_mus, MUS);
            float d_min = kTop - kBottom, d_max = H;
            float D = dist_top(kBottom, kMuSMin);
            float A = (D - d_min) / (d_max - d_min);
            float a = (A - x_mu_s * A) / (1.0f + x_mu_s * A);
            float dd = d_min + hrt_min(a, A) * (d_max - d_min);
            float mu_s = dd == 0.0f ? 1.0f : clampf((H * H - dd * dd) / (2.0f * kBottom * dd), -1.0f, 1.0f);
            float nu = clampf(uw_nu * 2.0f - 1.0f, -1.0f, 1.0f);
            float k = hrt_sqrt((1.0f - mu * mu) * (1.0f - mu_s * mu_s));
            nu = clampf(nu, mu * mu_s - k, mu * mu_s + k);

            // ComputeSingleScattering
            const int N = 50;
            float dx = (ground ? dist_bottom(r, mu) : dist_top(r, mu)) / (float)N;
            float rs[3] = { 0, 0, 0 }, ms[3] = { 0, 0, 0 };
            for (int i = 0; i <= N; ++i) {
                float d = (float)i * dx;
                float r_d = clampf(hrt_sqrt(d * d + 2.0f * r * mu * d + r * r), kBottom, kTop);
                float mu_s_d = clampf((r * mu_s + d * nu) / r_d, -1.0f, 1.0f);
                Rgb t1 = get_transmittance(T, r, mu, d, ground);
                Rgb t2 = transmittance_to_sun(T, r_d, mu_s_d);
                float dr = layer_density(kRayleighLayer, r_d - kBottom), dm = layer_density(kMieLayer, r_d - kBottom);
                float w = (i == 0 || i == N) ? 0.5f : 1.0f;
                for (int c = 0; c < 3; ++c) { float t = t1.v[c] * t2.v[c]; rs[c] += t * dr * w; ms[c] += t * dm * w; }
            }
            float* o = S + (((size_t)z * SH + y) * SW + x) * 4;
            for (int c = 0; c < 3; ++c) o[c] = rs[c] * dx * kSolar[c] * kRayleigh[c];
            o[3] = ms[0] * dx * kSolar[0] * kMieScattering[0];
        }
}

} // namespace

extern "C" int hrpt_precompute_atmosphere(float* transmittance, float* scattering, float* irradiance, int nthreads)
{
    if (!transmittance || !scattering) return HRPT_ERR_INVALID_ARGUMENT;
    compute_transmittance(transmittance);
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > SD) nthreads = SD;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([=]() { for (int z = t; z < SD; z += nthreads) scattering_slice(transmittance, scattering, z); });
    for (auto& t : th) t.join();
    if (irradiance) memset(irradiance, 0, sizeof(float) * 64 * 16 * 4);
    return HRPT_OK;
}
