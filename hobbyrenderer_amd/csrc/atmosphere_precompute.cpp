// atmosphere_precompute.cpp -- producer of the three Bruneton LUTs the reference loads from
// bin/bruneton/{transmittance,scattering,irradiance}.dat (/root/reference/src/CommonResources.cpp:519-569).
// Those files are git-ignored build artefacts of the reference (.gitignore:46) and their producer is not in
// its tree, so they are Scene INPUTS here: this file generates them in the same raw-float32-RGBA layout from the
// constants of src/shaders/Atmosphere.hlsli:41-75 (Earth, ozone layer on, ground albedo 0.1), following the published
// precomputation of E. Bruneton, "Precomputed Atmospheric Scattering" (2017 implementation, functions.glsl / model.cc):
//   transmittance -> direct ground irradiance -> single Rayleigh + Mie scattering -> for every further order:
//   scattering density (512 directions per texel, previous order + light reflected by the ground), indirect ground
//   irradiance (1024 directions), multiple scattering (50 samples along the ray). scattering.rgb accumulates the Rayleigh-
//   phase-normalised sum of all orders, scattering.a the red single-Mie term (the combined-texture layout the shader reads,
//   Atmosphere.hlsli:305-341); irradiance accumulates the orders >= 2 (the direct term is added analytically at run time).
// `orders` = 1 stops after single scattering (round-1/2 behaviour: zero irradiance table); the default is 4, the value of
// Bruneton's demo. How many orders the reference's own files hold is unknown: parity of the LUT
// CONTENTS stays unpinned; whatever is fed in is shared bit-identically by the HIP path and the oracle.
//
// One source, two executors: every function below is __host__ __device__ and uses only include/hobbyrt/detmath.h
// arithmetic (binary32 + - * / sqrt, own exp / sin / cos, no contraction), so a table comes out bit-identical whether its
// texels are computed by host threads or by a HIP kernel with one thread per texel (tests/test_atmosphere.py). The GPU
// takes ~0.1 s for four orders, sixteen host threads about two minutes.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/hobbyrt_pt.h"
#include "../../include/hobbyrt/detmath.h"

namespace {

#define ATM_HD __host__ __device__ static inline

constexpr float kBottom = 6360.0f, kTop = 6420.0f, kMuSMin = -0.207912f, kMieG = 0.8f, kGroundAlbedo = 0.1f;
constexpr float kSunAngularRadius = 0.004675f;
constexpr int TW = 256, TH = 64, SW = 256, SH = 128, SD = 32, NU = 8, MUS = 32, MU = 128, RS = 32, IW = 64, IH = 16;

struct Rgb { float v[3]; };
struct Layer { float width, exp_term, exp_scale, linear_term, constant_term; };

ATM_HD Rgb solar() { return Rgb{ { 1.474000f, 1.850400f, 1.911980f } }; }
ATM_HD Rgb rayleigh_coeff() { return Rgb{ { 0.005802f, 0.013558f, 0.033100f } }; }
ATM_HD Rgb mie_scattering() { return Rgb{ { 0.003996f, 0.003996f, 0.003996f } }; }
ATM_HD Rgb mie_extinction() { return Rgb{ { 0.004440f, 0.004440f, 0.004440f } }; }
ATM_HD Rgb absorption() { return Rgb{ { 0.000650f, 0.001881f, 0.000085f } }; }

ATM_HD float clampf(float x, float a, float b) { return hrt_clamp(x, a, b); }
ATM_HD float safe_sqrt(float a) { return hrt_sqrt(hrt_max(a, 0.0f)); }
ATM_HD float layer_density(float width, float exp_term, float exp_scale, float linear_term, float constant_term, float alt)
{
    (void)width;
    return clampf(exp_term * hrt_exp(exp_scale * alt) + linear_term * alt + constant_term, 0.0f, 1.0f);
}
ATM_HD float rayleigh_density(float alt) { return layer_density(0.0f, 1.0f, -1.0f / 8.0f, 0.0f, 0.0f, alt); }
ATM_HD float mie_density(float alt) { return layer_density(0.0f, 1.0f, -1.0f / 1.2f, 0.0f, 0.0f, alt); }
ATM_HD float ozone_density(float alt)
{
    return alt < 25.0f ? layer_density(25.0f, 0.0f, 0.0f, 1.0f / 15.0f, -2.0f / 3.0f, alt) : layer_density(0.0f, 0.0f, 0.0f, -1.0f / 15.0f, 8.0f / 3.0f, alt);
}
ATM_HD float dist_top(float r, float mu) { return hrt_max(-r * mu + safe_sqrt(r * r * (mu * mu - 1.0f) + kTop * kTop), 0.0f); }
ATM_HD float dist_bottom(float r, float mu) { return hrt_max(-r * mu - safe_sqrt(r * r * (mu * mu - 1.0f) + kBottom * kBottom), 0.0f); }
ATM_HD bool hits_ground(float r, float mu) { return mu < 0.0f && r * r * (mu * mu - 1.0f) + kBottom * kBottom >= 0.0f; }
ATM_HD float texcoord(float x, int n) { return 0.5f / (float)n + x * (1.0f - 1.0f / (float)n); }
ATM_HD float unit_range(float u, int n) { return (u - 0.5f / (float)n) / (1.0f - 1.0f / (float)n); }
ATM_HD float rayleigh_phase(float nu) { const float k = 3.0f / (16.0f * HRT_PI); return k * (1.0f + nu * nu); }
ATM_HD float mie_phase(float g, float nu)
{
    const float k = 3.0f / (8.0f * HRT_PI) * (1.0f - g * g) / (2.0f + g * g);
    const float b = 1.0f + g * g - 2.0f * g * nu;
    return k * (1.0f + nu * nu) / (b * hrt_sqrt(b));
}
ATM_HD int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// ---------------------------------------------------------------- transmittance
template <int WHICH>       // 0 Rayleigh, 1 Mie, 2 ozone
ATM_HD float optical_length(float r, float mu)
{
    const int N = 500;
    const float dx = dist_top(r, mu) / (float)N;
    float result = 0.0f;
    for (int i = 0; i <= N; ++i) {
        const float d = (float)i * dx;
        const float ri = hrt_sqrt(d * d + 2.0f * r * mu * d + r * r);
        const float alt = ri - kBottom;
        const float y = WHICH == 0 ? rayleigh_density(alt) : (WHICH == 1 ? mie_density(alt) : ozone_density(alt));
        result += y * ((i == 0 || i == N) ? 0.5f : 1.0f) * dx;
    }
    return result;
}
ATM_HD void transmittance_texel(float* T, int x, int y)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    const float x_mu = unit_range(((float)x + 0.5f) / (float)TW, TW), x_r = unit_range(((float)y + 0.5f) / (float)TH, TH);
    const float rho = H * x_r;
    const float r = hrt_sqrt(rho * rho + kBottom * kBottom);
    const float d_min = kTop - r, d_max = rho + H;
    const float d = d_min + x_mu * (d_max - d_min);
    float mu = d == 0.0f ? 1.0f : (H * H - rho * rho - d * d) / (2.0f * r * d);
    mu = clampf(mu, -1.0f, 1.0f);
    const float olR = optical_length<0>(r, mu), olM = optical_length<1>(r, mu), olO = optical_length<2>(r, mu);
    float* o = T + ((size_t)y * TW + x) * 4;
    const Rgb kr = rayleigh_coeff(), km = mie_extinction(), ka = absorption();
    for (int c = 0; c < 3; ++c) o[c] = hrt_exp(-(kr.v[c] * olR + km.v[c] * olM + ka.v[c] * olO));
    o[3] = 1.0f;
}

// bilinear fp32 lookup in the float transmittance table (clamp addressing)
ATM_HD Rgb lookup_transmittance(const float* T, float r, float mu)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    const float rho = safe_sqrt(r * r - kBottom * kBottom);
    const float d = dist_top(r, mu);
    const float d_min = kTop - r, d_max = rho + H;
    const float x_mu = (d - d_min) / (d_max - d_min), x_r = rho / H;
    const float u = texcoord(x_mu, TW), v = texcoord(x_r, TH);
    const float fx = u * (float)TW - 0.5f, fy = v * (float)TH - 0.5f;
    const float ix = hrt_floor(fx), iy = hrt_floor(fy);
    const float tx = fx - ix, ty = fy - iy;
    const int x0 = clampi((int)ix, TW), x1 = clampi((int)ix + 1, TW), y0 = clampi((int)iy, TH), y1 = clampi((int)iy + 1, TH);
    Rgb o;
    for (int c = 0; c < 3; ++c) {
        const float a = T[((size_t)y0 * TW + x0) * 4 + c] * (1.0f - tx) + T[((size_t)y0 * TW + x1) * 4 + c] * tx;
        const float b = T[((size_t)y1 * TW + x0) * 4 + c] * (1.0f - tx) + T[((size_t)y1 * TW + x1) * 4 + c] * tx;
        o.v[c] = a * (1.0f - ty) + b * ty;
    }
    return o;
}
ATM_HD Rgb get_transmittance(const float* T, float r, float mu, float d, bool ground)
{
    const float r_d = clampf(hrt_sqrt(d * d + 2.0f * r * mu * d + r * r), kBottom, kTop);
    const float mu_d = clampf((r * mu + d) / r_d, -1.0f, 1.0f);
    Rgb a, b, o;
    if (ground) { a = lookup_transmittance(T, r_d, -mu_d); b = lookup_transmittance(T, r, -mu); }
    else { a = lookup_transmittance(T, r, mu); b = lookup_transmittance(T, r_d, mu_d); }
    for (int c = 0; c < 3; ++c) o.v[c] = hrt_min(a.v[c] / b.v[c], 1.0f);
    return o;
}
ATM_HD Rgb transmittance_to_sun(const float* T, float r, float mu_s)
{
    const float s = kBottom / r;
    const float ch = -hrt_sqrt(hrt_max(1.0f - s * s, 0.0f));
    const float e = s * kSunAngularRadius;
    const float t = hrt_saturate(((mu_s - ch) + e) / (e + e));
    const float f = (t * t) * (3.0f - 2.0f * t);
    Rgb o = lookup_transmittance(T, r, mu_s);
    for (int c = 0; c < 3; ++c) o.v[c] *= f;
    return o;
}

// ---------------------------------------------------------------- the 4D scattering parameterisation
struct RMuMuSNu { float r, mu, mu_s, nu; bool ground; };
// GetRMuMuSNuFromScatteringTextureFragCoord: the parameters of texel (x, y, z) of a 256 x 128 x 32 scattering table
ATM_HD RMuMuSNu texel_params(int x, int y, int z)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    const float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f, fcz = (float)z + 0.5f;
    const float frag_nu = hrt_floor(fcx / (float)MUS);
    const float frag_mu_s = fcx - frag_nu * (float)MUS;
    const float uw_nu = frag_nu / (float)(NU - 1), uw_mus = frag_mu_s / (float)MUS, uw_mu = fcy / (float)MU, uw_r = fcz / (float)RS;
    const float rho = H * unit_range(uw_r, RS);
    RMuMuSNu p;
    p.r = hrt_sqrt(rho * rho + kBottom * kBottom);
    if (uw_mu < 0.5f) {
        const float d_min = p.r - kBottom, d_max = rho;
        const float d = d_min + (d_max - d_min) * unit_range(1.0f - 2.0f * uw_mu, MU / 2);
        p.mu = d == 0.0f ? -1.0f : clampf(-(rho * rho + d * d) / (2.0f * p.r * d), -1.0f, 1.0f);
        p.ground = true;
    } else {
        const float d_min = kTop - p.r, d_max = rho + H;
        const float d = d_min + (d_max - d_min) * unit_range(2.0f * uw_mu - 1.0f, MU / 2);
        p.mu = d == 0.0f ? 1.0f : clampf((H * H - rho * rho - d * d) / (2.0f * p.r * d), -1.0f, 1.0f);
        p.ground = false;
    }
    const float x_mu_s = unit_range(uw_mus, MUS);
    const float d_min = kTop - kBottom, d_max = H;
    const float D = dist_top(kBottom, kMuSMin);
    const float A = (D - d_min) / (d_max - d_min);
    const float a = (A - x_mu_s * A) / (1.0f + x_mu_s * A);
    const float dd = d_min + hrt_min(a, A) * (d_max - d_min);
    p.mu_s = dd == 0.0f ? 1.0f : clampf((H * H - dd * dd) / (2.0f * kBottom * dd), -1.0f, 1.0f);
    p.nu = clampf(uw_nu * 2.0f - 1.0f, -1.0f, 1.0f);
    const float k = hrt_sqrt((1.0f - p.mu * p.mu) * (1.0f - p.mu_s * p.mu_s));
    p.nu = clampf(p.nu, p.mu * p.mu_s - k, p.mu * p.mu_s + k);
    return p;
}
// GetScatteringTextureUvwzFromRMuMuSNu (Atmosphere.hlsli:263-297)
ATM_HD void scattering_uvwz(float r, float mu, float mu_s, float nu, bool ground, float& u_nu, float& u_mu_s, float& u_mu, float& u_r)
{
    const float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    const float rho = safe_sqrt(r * r - kBottom * kBottom);
    u_r = texcoord(rho / H, RS);
    const float r_mu = r * mu;
    const float discriminant = r_mu * r_mu - r * r + kBottom * kBottom;
    if (ground) {
        const float d = -r_mu - safe_sqrt(discriminant);
        const float d_min = r - kBottom, d_max = rho;
        u_mu = 0.5f - 0.5f * texcoord(d_max == d_min ? 0.0f : (d - d_min) / (d_max - d_min), MU / 2);
    } else {
        const float d = -r_mu + safe_sqrt(discriminant + H * H);
        const float d_min = kTop - r, d_max = rho + H;
        u_mu = 0.5f + 0.5f * texcoord((d - d_min) / (d_max - d_min), MU / 2);
    }
    const float d = dist_top(kBottom, mu_s);
    const float d_min = kTop - kBottom, d_max = H;
    const float a = (d - d_min) / (d_max - d_min);
    const float D = dist_top(kBottom, kMuSMin);
    const float A = (D - d_min) / (d_max - d_min);
    u_mu_s = texcoord(hrt_max(1.0f - a / A, 0.0f) / (1.0f + a), MUS);
    u_nu = (nu + 1.0f) / 2.0f;
}
// trilinear fp32 lookup in a 256 x 128 x 32 table of `stride` floats per texel (first three used), clamp addressing
ATM_HD Rgb sample3d(const float* tab, int stride, float u, float v, float w)
{
    const float fx = u * (float)SW - 0.5f, fy = v * (float)SH - 0.5f, fz = w * (float)SD - 0.5f;
    const float ix = hrt_floor(fx), iy = hrt_floor(fy), iz = hrt_floor(fz);
    const float tx = fx - ix, ty = fy - iy, tz = fz - iz;
    const int x0 = clampi((int)ix, SW), x1 = clampi((int)ix + 1, SW), y0 = clampi((int)iy, SH), y1 = clampi((int)iy + 1, SH), z0 = clampi((int)iz, SD), z1 = clampi((int)iz + 1, SD);
    Rgb o;
    for (int c = 0; c < 3; ++c) {
        float s[2];
        for (int k = 0; k < 2; ++k) {
            const size_t zo = (size_t)(k ? z1 : z0) * SH;
            const float a = tab[((zo + y0) * SW + x0) * stride + c] * (1.0f - tx) + tab[((zo + y0) * SW + x1) * stride + c] * tx;
            const float b = tab[((zo + y1) * SW + x0) * stride + c] * (1.0f - tx) + tab[((zo + y1) * SW + x1) * stride + c] * tx;
            s[k] = a * (1.0f - ty) + b * ty;
        }
        o.v[c] = s[0] * (1.0f - tz) + s[1] * tz;
    }
    return o;
}
// GetScattering(texture, r, mu, mu_s, nu, ground): two 3D lookups, interpolated in nu
ATM_HD Rgb get_scattering(const float* tab, int stride, float r, float mu, float mu_s, float nu, bool ground)
{
    float u_nu, u_mu_s, u_mu, u_r;
    scattering_uvwz(r, mu, mu_s, nu, ground, u_nu, u_mu_s, u_mu, u_r);
    const float tex_coord_x = u_nu * (float)(NU - 1);
    const float tex_x = hrt_floor(tex_coord_x);
    const float lerp = tex_coord_x - tex_x;
    const float u0 = (tex_x + u_mu_s) / (float)NU, u1 = (tex_x + 1.0f + u_mu_s) / (float)NU;
    const Rgb a = sample3d(tab, stride, u0, u_mu, u_r), b = sample3d(tab, stride, u1, u_mu, u_r);
    Rgb o;
    for (int c = 0; c < 3; ++c) o.v[c] = a.v[c] * (1.0f - lerp) + b.v[c] * lerp;
    return o;
}
// bilinear lookup in a 64 x 16 irradiance table of 3 floats per texel (GetIrradiance, Atmosphere.hlsli:371-389)
ATM_HD Rgb get_irradiance(const float* tab, float r, float mu_s)
{
    const float x_r = (r - kBottom) / (kTop - kBottom), x_mu_s = mu_s * 0.5f + 0.5f;
    const float u = texcoord(x_mu_s, IW), v = texcoord(x_r, IH);
    const float fx = u * (float)IW - 0.5f, fy = v * (float)IH - 0.5f;
    const float ix = hrt_floor(fx), iy = hrt_floor(fy);
    const float tx = fx - ix, ty = fy - iy;
    const int x0 = clampi((int)ix, IW), x1 = clampi((int)ix + 1, IW), y0 = clampi((int)iy, IH), y1 = clampi((int)iy + 1, IH);
    Rgb o;
    for (int c = 0; c < 3; ++c) {
        const float a = tab[((size_t)y0 * IW + x0) * 3 + c] * (1.0f - tx) + tab[((size_t)y0 * IW + x1) * 3 + c] * tx;
        const float b = tab[((size_t)y1 * IW + x0) * 3 + c] * (1.0f - tx) + tab[((size_t)y1 * IW + x1) * 3 + c] * tx;
        o.v[c] = a * (1.0f - ty) + b * ty;
    }
    return o;
}

// ---------------------------------------------------------------- the passes, one texel each
struct Tables {
    const float* T;             // transmittance, 256 x 64 x 4
    float* deltaIrr;            // 64 x 16 x 3: direct irradiance, then the indirect irradiance of the order being computed
    float* irradiance;          // 64 x 16 x 4: accumulated (orders >= 2)
    float* deltaR; float* deltaM;        // single Rayleigh / Mie scattering without phase functions, 3 floats per texel
    float* deltaDensity;        // scattering density of the current order
    float* deltaMulti;          // scattering of the previous order >= 2 (with phase functions)
    float* scattering;          // 256 x 128 x 32 x 4: the table the shader reads
};

ATM_HD void direct_irradiance_texel(const Tables& t, int x, int y)
{
    const float x_mu_s = unit_range(((float)x + 0.5f) / (float)IW, IW), x_r = unit_range(((float)y + 0.5f) / (float)IH, IH);
    const float r = kBottom + x_r * (kTop - kBottom), mu_s = clampf(2.0f * x_mu_s - 1.0f, -1.0f, 1.0f);
    const float alpha = kSunAngularRadius;
    const float acf = mu_s < -alpha ? 0.0f : (mu_s > alpha ? mu_s : (mu_s + alpha) * (mu_s + alpha) / (4.0f * alpha));
    const Rgb tr = lookup_transmittance(t.T, r, mu_s), sol = solar();
    float* o = t.deltaIrr + ((size_t)y * IW + x) * 3;
    for (int c = 0; c < 3; ++c) o[c] = sol.v[c] * tr.v[c] * acf;
    float* acc = t.irradiance + ((size_t)y * IW + x) * 4;
    acc[0] = acc[1] = acc[2] = 0.0f; acc[3] = 1.0f;
}

ATM_HD void single_scattering_texel(const Tables& t, int x, int y, int z)
{
    const RMuMuSNu p = texel_params(x, y, z);
    const int N = 50;
    const float dx = (p.ground ? dist_bottom(p.r, p.mu) : dist_top(p.r, p.mu)) / (float)N;
    float rs[3] = { 0, 0, 0 }, ms[3] = { 0, 0, 0 };
    for (int i = 0; i <= N; ++i) {
        const float d = (float)i * dx;
        const float r_d = clampf(hrt_sqrt(d * d + 2.0f * p.r * p.mu * d + p.r * p.r), kBottom, kTop);
        const float mu_s_d = clampf((p.r * p.mu_s + d * p.nu) / r_d, -1.0f, 1.0f);
        const Rgb t1 = get_transmittance(t.T, p.r, p.mu, d, p.ground);
        const Rgb t2 = transmittance_to_sun(t.T, r_d, mu_s_d);
        const float dr = rayleigh_density(r_d - kBottom), dm = mie_density(r_d - kBottom);
        const float w = (i == 0 || i == N) ? 0.5f : 1.0f;
        for (int c = 0; c < 3; ++c) { const float tt = t1.v[c] * t2.v[c]; rs[c] += tt * dr * w; ms[c] += tt * dm * w; }
    }
    const size_t idx = ((size_t)z * SH + y) * SW + x;
    const Rgb sol = solar(), kr = rayleigh_coeff(), km = mie_scattering();
    float* o = t.scattering + idx * 4;
    for (int c = 0; c < 3; ++c) {
        const float ray = rs[c] * dx * sol.v[c] * kr.v[c], mie = ms[c] * dx * sol.v[c] * km.v[c];
        o[c] = ray;
        if (t.deltaR) { t.deltaR[idx * 3 + c] = ray; t.deltaM[idx * 3 + c] = mie; }
    }
    o[3] = ms[0] * dx * sol.v[0] * km.v[0];
}

// radiance of scattering order `order` arriving at (r, mu) from direction nu relative to the sun (GetScattering with the order switch)
ATM_HD Rgb scattering_of_order(const Tables& t, float r, float mu, float mu_s, float nu, bool ground, int order)
{
    if (order == 1) {
        const Rgb ray = get_scattering(t.deltaR, 3, r, mu, mu_s, nu, ground), mie = get_scattering(t.deltaM, 3, r, mu, mu_s, nu, ground);
        const float pr = rayleigh_phase(nu), pm = mie_phase(kMieG, nu);
        Rgb o;
        for (int c = 0; c < 3; ++c) o.v[c] = ray.v[c] * pr + mie.v[c] * pm;
        return o;
    }
    return get_scattering(t.deltaMulti, 3, r, mu, mu_s, nu, ground);
}

// ComputeScatteringDensity: light of order - 1 (and light reflected by the ground) scattered at (r, mu) towards the viewer
ATM_HD void scattering_density_texel(const Tables& t, int x, int y, int z, int order)
{
    const RMuMuSNu p = texel_params(x, y, z);
    const float ox = hrt_sqrt(1.0f - p.mu * p.mu), oz = p.mu;                      // omega = (ox, 0, oz), zenith = (0, 0, 1)
    const float sun_x = ox == 0.0f ? 0.0f : (p.nu - p.mu * p.mu_s) / ox;
    const float sun_y = hrt_sqrt(hrt_max(1.0f - sun_x * sun_x - p.mu_s * p.mu_s, 0.0f));
    const float sun_z = p.mu_s;
    const int N = 16;
    const float dphi = HRT_PI / (float)N, dtheta = HRT_PI / (float)N;
    const float rd = rayleigh_density(p.r - kBottom), md = mie_density(p.r - kBottom);
    const Rgb kr = rayleigh_coeff(), km = mie_scattering();
    float acc[3] = { 0, 0, 0 };
    for (int l = 0; l < N; ++l) {
        const float theta = ((float)l + 0.5f) * dtheta;
        float sin_theta, cos_theta; hrt_sincos(theta, &sin_theta, &cos_theta);
        const bool ground = hits_ground(p.r, cos_theta);
        float dist_ground = 0.0f; Rgb tr_ground = Rgb{ { 0, 0, 0 } }; float albedo = 0.0f;
        if (ground) { dist_ground = dist_bottom(p.r, cos_theta); tr_ground = get_transmittance(t.T, p.r, cos_theta, dist_ground, true); albedo = kGroundAlbedo; }
        for (int m = 0; m < 2 * N; ++m) {
            const float phi = ((float)m + 0.5f) * dphi;
            float sin_phi, cos_phi; hrt_sincos(phi, &sin_phi, &cos_phi);
            const float wx = cos_phi * sin_theta, wy = sin_phi * sin_theta, wz = cos_theta;
            const float domega = (dtheta * dphi) * sin_theta;
            const float nu1 = (sun_x * wx + sun_y * wy) + sun_z * wz;
            Rgb incident = scattering_of_order(t, p.r, wz, p.mu_s, nu1, ground, order - 1);
            // light reflected by the ground: normal of the ground point the direction reaches, irradiance of order - 2 arriving there
            const float gx = wx * dist_ground, gy = wy * dist_ground, gz = p.r + wz * dist_ground;
            const float gl = hrt_sqrt((gx * gx + gy * gy) + gz * gz);
            const float gdot = ((gx / gl) * sun_x + (gy / gl) * sun_y) + (gz / gl) * sun_z;
            const Rgb irr = get_irradiance(t.deltaIrr, kBottom, gdot);
            for (int c = 0; c < 3; ++c) incident.v[c] += tr_ground.v[c] * albedo * (1.0f / HRT_PI) * irr.v[c];
            const float nu2 = ox * wx + oz * wz;
            const float pr = rayleigh_phase(nu2), pm = mie_phase(kMieG, nu2);
            for (int c = 0; c < 3; ++c) acc[c] += incident.v[c] * (kr.v[c] * rd * pr + km.v[c] * md * pm) * domega;
        }
    }
    float* o = t.deltaDensity + (((size_t)z * SH + y) * SW + x) * 3;
    for (int c = 0; c < 3; ++c) o[c] = acc[c];
}

// ComputeIndirectIrradiance: sky light of scattering order `order` on a horizontal surface at (r, mu_s)
ATM_HD void indirect_irradiance_texel(const Tables& t, float* out3, int x, int y, int order)
{
    const float x_mu_s = unit_range(((float)x + 0.5f) / (float)IW, IW), x_r = unit_range(((float)y + 0.5f) / (float)IH, IH);
    const float r = kBottom + x_r * (kTop - kBottom), mu_s = clampf(2.0f * x_mu_s - 1.0f, -1.0f, 1.0f);
    const int N = 32;
    const float dphi = HRT_PI / (float)N, dtheta = HRT_PI / (float)N;
    const float sx = hrt_sqrt(1.0f - mu_s * mu_s), sz = mu_s;
    float acc[3] = { 0, 0, 0 };
    for (int j = 0; j < N / 2; ++j) {
        const float theta = ((float)j + 0.5f) * dtheta;
        float sin_theta, cos_theta; hrt_sincos(theta, &sin_theta, &cos_theta);
        for (int i = 0; i < 2 * N; ++i) {
            const float phi = ((float)i + 0.5f) * dphi;
            float sin_phi, cos_phi; hrt_sincos(phi, &sin_phi, &cos_phi);
            const float wx = cos_phi * sin_theta, wz = cos_theta;
            const float domega = (dtheta * dphi) * sin_theta;
            const float nu = wx * sx + wz * sz;
            const Rgb s = scattering_of_order(t, r, wz, mu_s, nu, false, order);
            for (int c = 0; c < 3; ++c) acc[c] += s.v[c] * wz * domega;
        }
    }
    float* o = out3 + ((size_t)y * IW + x) * 3;
    for (int c = 0; c < 3; ++c) o[c] = acc[c];
}

// ComputeMultipleScattering: the scattering density integrated along the view ray; also folds the result into the output table
ATM_HD void multiple_scattering_texel(const Tables& t, float* out3, int x, int y, int z)
{
    const RMuMuSNu p = texel_params(x, y, z);
    const int N = 50;
    const float dx = (p.ground ? dist_bottom(p.r, p.mu) : dist_top(p.r, p.mu)) / (float)N;
    float acc[3] = { 0, 0, 0 };
    for (int i = 0; i <= N; ++i) {
        const float d = (float)i * dx;
        const float r_i = clampf(hrt_sqrt(d * d + 2.0f * p.r * p.mu * d + p.r * p.r), kBottom, kTop);
        const float mu_i = clampf((p.r * p.mu + d) / r_i, -1.0f, 1.0f);
        const float mu_s_i = clampf((p.r * p.mu_s + d * p.nu) / r_i, -1.0f, 1.0f);
        const Rgb dens = get_scattering(t.deltaDensity, 3, r_i, mu_i, mu_s_i, p.nu, p.ground);
        const Rgb tr = get_transmittance(t.T, p.r, p.mu, d, p.ground);
        const float w = (i == 0 || i == N) ? 0.5f : 1.0f;
        for (int c = 0; c < 3; ++c) acc[c] += dens.v[c] * tr.v[c] * dx * w;
    }
    const size_t idx = ((size_t)z * SH + y) * SW + x;
    const float pr = rayleigh_phase(p.nu);
    for (int c = 0; c < 3; ++c) { out3[idx * 3 + c] = acc[c]; t.scattering[idx * 4 + c] += acc[c] / pr; }
}

// ---------------------------------------------------------------- executors
enum Pass { kTransmittance, kDirectIrradiance, kSingle, kDensity, kIndirect, kMultiple };
struct PassArgs { Tables t; float* Tw; float* out3; int order; };
ATM_HD void run_texel(int pass, const PassArgs& a, uint32_t i)
{
    if (pass == kTransmittance) transmittance_texel(a.Tw, (int)(i % TW), (int)(i / TW));
    else if (pass == kDirectIrradiance) direct_irradiance_texel(a.t, (int)(i % IW), (int)(i / IW));
    else if (pass == kIndirect) indirect_irradiance_texel(a.t, a.out3, (int)(i % IW), (int)(i / IW), a.order);
    else {
        const int x = (int)(i % SW), y = (int)((i / SW) % SH), z = (int)(i / (SW * SH));
        if (pass == kSingle) single_scattering_texel(a.t, x, y, z);
        else if (pass == kDensity) scattering_density_texel(a.t, x, y, z, a.order);
        else multiple_scattering_texel(a.t, a.out3, x, y, z);
    }
}
uint32_t pass_texels(int pass) { return pass == kTransmittance ? TW * TH : ((pass == kDirectIrradiance || pass == kIndirect) ? IW * IH : SW * SH * SD); }

__global__ __launch_bounds__(256) void atm_pass_kernel(int pass, PassArgs a, uint32_t first, uint32_t count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) run_texel(pass, a, first + i);
}

struct Executor {
    int device = -1, nthreads = 1;          // device < 0: host threads
    hipError_t run(int pass, const PassArgs& a, uint32_t first = 0, uint32_t count = 0) const
    {
        const uint32_t n = count ? count : pass_texels(pass);
        if (device >= 0) {
            hipLaunchKernelGGL(atm_pass_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, pass, a, first, n);
            hipError_t e = hipGetLastError();
            return e != hipSuccess ? e : hipDeviceSynchronize();
        }
        std::vector<std::thread> th;
        const int nt = nthreads < 1 ? 1 : nthreads;
        for (int k = 0; k < nt; ++k)
            th.emplace_back([=]() { for (uint32_t i = first + (uint32_t)k; i < first + n; i += (uint32_t)nt) run_texel(pass, a, i); });
        for (auto& t : th) t.join();
        return hipSuccess;
    }
};

// irradiance += delta (orders >= 2)
void accumulate_irradiance(float* irradiance4, const float* delta3) { for (int i = 0; i < IW * IH; ++i) for (int c = 0; c < 3; ++c) irradiance4[i * 4 + c] += delta3[i * 3 + c]; }

struct Buffers {            // all tables of a precomputation, in host or device memory
    bool onDevice = false;
    float *T = nullptr, *scat = nullptr, *irr = nullptr, *dIrr = nullptr, *dIrrNext = nullptr, *dR = nullptr, *dM = nullptr, *dDens = nullptr, *dMulti = nullptr;
    std::vector<float*> owned;
    hipError_t alloc(float** p, size_t floats)
    {
        if (onDevice) { hipError_t e = hipMalloc((void**)p, floats * 4); if (e != hipSuccess) return e; e = hipMemset(*p, 0, floats * 4); if (e != hipSuccess) return e; }
        else { *p = (float*)calloc(floats, 4); if (!*p) return hipErrorOutOfMemory; }
        owned.push_back(*p);
        return hipSuccess;
    }
    ~Buffers() { for (float* p : owned) { if (onDevice) (void)hipFree(p); else free(p); } }
};

int precompute(float* transmittance, float* scattering, float* irradiance, int orders, int nthreads, int device)
{
    if (!transmittance || !scattering) return HRPT_ERR_INVALID_ARGUMENT;
    if (orders < 1) orders = 1;
    if (orders > 8) orders = 8;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    Executor ex; ex.nthreads = nthreads;
    if (device == -2) {                                          // automatic: the current HIP device when there is one
        int n = 0; device = (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 0 : -1;
        if (device >= 0) { int cur = 0; if (hipGetDevice(&cur) == hipSuccess) device = cur; }
    }
    ex.device = device;
    Buffers b; b.onDevice = device >= 0;
    if (b.onDevice && hipSetDevice(device) != hipSuccess) return HRPT_ERR_NO_DEVICE;
    const size_t nS = (size_t)SW * SH * SD;
    auto bad = [](hipError_t e) { return e != hipSuccess; };
    if (b.onDevice) {
        if (bad(b.alloc(&b.T, (size_t)TW * TH * 4)) || bad(b.alloc(&b.scat, nS * 4)) || bad(b.alloc(&b.irr, (size_t)IW * IH * 4))) return HRPT_ERR_OUT_OF_MEMORY;
    } else { b.T = transmittance; b.scat = scattering; if (bad(b.alloc(&b.irr, (size_t)IW * IH * 4))) return HRPT_ERR_OUT_OF_MEMORY; }
    if (bad(b.alloc(&b.dIrr, (size_t)IW * IH * 3)) || bad(b.alloc(&b.dIrrNext, (size_t)IW * IH * 3))) return HRPT_ERR_OUT_OF_MEMORY;
    if (orders > 1 && (bad(b.alloc(&b.dR, nS * 3)) || bad(b.alloc(&b.dM, nS * 3)) || bad(b.alloc(&b.dDens, nS * 3)) || bad(b.alloc(&b.dMulti, nS * 3)))) return HRPT_ERR_OUT_OF_MEMORY;

    PassArgs a{};
    a.Tw = b.T;
    a.t.T = b.T; a.t.deltaIrr = b.dIrr; a.t.irradiance = b.irr; a.t.deltaR = b.dR; a.t.deltaM = b.dM; a.t.deltaDensity = b.dDens; a.t.deltaMulti = b.dMulti; a.t.scattering = b.scat;
    if (bad(ex.run(kTransmittance, a))) return HRPT_ERR_HIP;
    if (bad(ex.run(kDirectIrradiance, a))) return HRPT_ERR_HIP;
    if (bad(ex.run(kSingle, a))) return HRPT_ERR_HIP;
    std::vector<float> hostIrr((size_t)IW * IH * 4, 0.0f), hostDelta((size_t)IW * IH * 3);
    for (size_t i = 0; i < (size_t)IW * IH; ++i) hostIrr[i * 4 + 3] = 1.0f;
    for (int order = 2; order <= orders; ++order) {
        a.order = order;
        if (bad(ex.run(kDensity, a))) return HRPT_ERR_HIP;                      // reads deltaIrr of order - 2 (direct for order 2) and scattering of order - 1
        a.order = order - 1; a.out3 = b.dIrrNext;
        if (bad(ex.run(kIndirect, a))) return HRPT_ERR_HIP;                     // irradiance from the sky light of order - 1
        if (b.onDevice) { if (hipMemcpy(hostDelta.data(), b.dIrrNext, hostDelta.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return HRPT_ERR_HIP; }
        else memcpy(hostDelta.data(), b.dIrrNext, hostDelta.size() * 4);
        accumulate_irradiance(hostIrr.data(), hostDelta.data());
        std::swap(b.dIrr, b.dIrrNext); a.t.deltaIrr = b.dIrr;
        a.out3 = b.dMulti;
        if (bad(ex.run(kMultiple, a))) return HRPT_ERR_HIP;                     // scattering of this order; also added (without Rayleigh phase) to the output
    }
    if (b.onDevice) {
        if (hipMemcpy(transmittance, b.T, (size_t)TW * TH * 16, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(scattering, b.scat, nS * 16, hipMemcpyDeviceToHost) != hipSuccess) return HRPT_ERR_HIP;
    }
    if (irradiance) memcpy(irradiance, hostIrr.data(), hostIrr.size() * 4);
    return HRPT_OK;
}

} // namespace

// orders: 1 = single scattering only, 4 = Bruneton's demo (default of hrpt_precompute_atmosphere); device: -1 host threads, >= 0 that HIP device,
// -2 the current HIP device when there is one, else host threads. The tables are bit-identical whichever executor made them.
extern "C" int hrpt_precompute_atmosphere_ex(float* transmittance, float* scattering, float* irradiance, int orders, int nthreads, int device)
{
    return precompute(transmittance, scattering, irradiance, orders, nthreads, device);
}
extern "C" int hrpt_precompute_atmosphere(float* transmittance, float* scattering, float* irradiance, int nthreads)
{
    return precompute(transmittance, scattering, irradiance, 4, nthreads, -2);
}
// test hook: the texels [first, first + count) of one pass (3 = scattering density, 4 = indirect irradiance, 5 = multiple scattering) computed from
// tables given in HOST memory, by host threads (device < 0) or by the kernel on HIP device `device` (the tables are copied there and back):
// tests/test_atmosphere.py checks that both executors return the same bits. Every table is full size (3 floats per texel: 256 x 128 x 32 for the
// scattering tables, 64 x 16 for deltaIrr; transmittance / scattering4 4 floats per texel); out3 (full size too) receives the texels asked for.
extern "C" int hrpt_atmosphere_pass(int pass, int order, uint32_t first, uint32_t count, const float* transmittance, const float* deltaIrr, const float* deltaR, const float* deltaM,
                                     const float* deltaDensity, const float* deltaMulti, float* scattering4, float* out3, int nthreads, int device)
{
    if (pass != kDensity && pass != kMultiple && pass != kIndirect) return HRPT_ERR_INVALID_ARGUMENT;
    if (!transmittance || !deltaIrr || !deltaR || !deltaM || !deltaDensity || !deltaMulti || !scattering4 || !out3 || first + count > pass_texels(pass)) return HRPT_ERR_INVALID_ARGUMENT;
    const size_t nS = (size_t)SW * SH * SD, nI = (size_t)IW * IH, nOut = pass == kIndirect ? nI * 3 : nS * 3;
    Executor ex; ex.nthreads = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency(); ex.device = device;
    PassArgs a{}; a.order = order;
    if (device < 0) {
        a.t.T = transmittance; a.t.deltaIrr = const_cast<float*>(deltaIrr); a.t.deltaR = const_cast<float*>(deltaR); a.t.deltaM = const_cast<float*>(deltaM);
        a.t.deltaDensity = pass == kDensity ? out3 : const_cast<float*>(deltaDensity); a.t.deltaMulti = const_cast<float*>(deltaMulti); a.t.scattering = scattering4; a.out3 = out3;
        return ex.run(pass, a, first, count) == hipSuccess ? HRPT_OK : HRPT_ERR_HIP;
    }
    if (hipSetDevice(device) != hipSuccess) return HRPT_ERR_NO_DEVICE;
    Buffers b; b.onDevice = true;
    float *dOut = nullptr;
    auto up = [&](float** d, const float* h, size_t floats) { return b.alloc(d, floats) == hipSuccess && hipMemcpy(*d, h, floats * 4, hipMemcpyHostToDevice) == hipSuccess; };
    if (!up(&b.T, transmittance, (size_t)TW * TH * 4) || !up(&b.dIrr, deltaIrr, nI * 3) || !up(&b.dR, deltaR, nS * 3) || !up(&b.dM, deltaM, nS * 3) || !up(&b.dDens, deltaDensity, nS * 3) ||
        !up(&b.dMulti, deltaMulti, nS * 3) || !up(&b.scat, scattering4, nS * 4) || !up(&dOut, out3, nOut)) return HRPT_ERR_OUT_OF_MEMORY;
    a.t.T = b.T; a.t.deltaIrr = b.dIrr; a.t.deltaR = b.dR; a.t.deltaM = b.dM; a.t.deltaDensity = pass == kDensity ? dOut : b.dDens; a.t.deltaMulti = b.dMulti; a.t.scattering = b.scat; a.out3 = dOut;
    if (ex.run(pass, a, first, count) != hipSuccess) return HRPT_ERR_HIP;
    if (hipMemcpy(out3, dOut, nOut * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(scattering4, b.scat, nS * 16, hipMemcpyDeviceToHost) != hipSuccess) return HRPT_ERR_HIP;
    return HRPT_OK;
}
