// pt_device.h -- gfx950 device code shared by the path-tracer kernels: vector algebra, BVH
// traversal with a watertight triangle test, bindless/LUT sampling, Bruneton atmosphere
// lookups, PBR BRDF evaluation and sampling.
//
// Follows /root/reference/src/shaders/{PathTracer.hlsl, RaytracingCommon.hlsli,
// CommonLighting.hlsli, Atmosphere.hlsli, MeshCommon.hlsli, Common.hlsli, RNG.hlsli}; each
// function names the lines it implements. Scalar HLSL intrinsics come from the numeric
// contract include/hobbyrt/detmath.h; expression order is fixed (left to right, no FMA
// contraction: build with -ffp-contract=off) so that radiance is a pure function of
// (Scene, constants, pixel, accumulation index).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hobbyrt_pt.h"
#include "../../include/hobbyrt/detmath.h"
#include "../../include/hobbyrt/srgb_table.h"

#define HRT_DEV __device__ __forceinline__

// Phase profile (make PHASES=1 builds libhobbyrt_pt_phases.so): HRT_PHASE(k) counts, per wave, how often a code region runs and with how
// many lanes -- the lane utilisation of every stage of a kernel, which the PMC counters only give per kernel. One aggregated atomic per
// wave and execution: slow, a diagnostic build only (scripts/phase_profile.py prints the table).
#if defined(HRPT_PHASE_PROFILE) && defined(HRPT_PHASE_TU)       // only the wavefront translation unit is instrumented (no RDC)
extern __device__ unsigned long long g_phaseCounters[128];
#define HRT_PHASE(k)                                                                                                      \
    do {                                                                                                                  \
        const unsigned long long m_ = __ballot(true);                                                                     \
        if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)m_) - 1)) {                                              \
            atomicAdd(&g_phaseCounters[2 * (k)], 1ull); atomicAdd(&g_phaseCounters[2 * (k) + 1], (unsigned long long)__popcll(m_)); \
        }                                                                                                                 \
    } while (0)
#else
#define HRT_PHASE(k) do { } while (0)
#endif
// phase ids
enum {
    PH_EXT_ITER = 0, PH_EXT_REFILL, PH_EXT_NODE, PH_EXT_LEAF, PH_EXT_TRI, PH_EXT_FINISH, PH_EXT_CANDIDATE,
    PH_ANY_ITER = 8, PH_ANY_REFILL, PH_ANY_NODE, PH_ANY_LEAF, PH_ANY_TRI, PH_ANY_FINISH,
    PH_SHADE_ITER = 16, PH_SHADE_HIT, PH_SHADE_ATTR, PH_SHADE_TEX, PH_SHADE_TRANS, PH_SHADE_NEE, PH_SHADE_LOBE_BEGIN, PH_SHADE_DIFFUSE, PH_SHADE_SPEC,
    PH_SHADE_SKY, PH_SHADE_WRITE, PH_SHADE_SORT,
    PH_SHADOW_ENTRY = 32, PH_SHADOW_SAMPLE, PH_SHADOW_QUERY, PH_SHADOW_CONTRIB, PH_SHADOW_RAYS_ITEM,
    PH_COUNT = 40
};

namespace hrt {

// ------------------------------------------------------------------ vectors
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

HRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
HRT_DEV f3 mk3(const float* p) { return mk3(p[0], p[1], p[2]); }
HRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HRT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
HRT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
HRT_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
HRT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HRT_DEV float length(f3 a) { return hrt_sqrt(dot(a, a)); }
HRT_DEV f3 normalize(f3 a) { float inv = 1.0f / hrt_sqrt(dot(a, a)); return a * inv; }
HRT_DEV float lerp(float a, float b, float t) { return a + t * (b - a); }
HRT_DEV float comp(f3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
HRT_DEV float maxcomp(f3 a) { return hrt_max(a.x, hrt_max(a.y, a.z)); }
HRT_DEV f3 reflect(f3 i, f3 n) { float s = 2.0f * dot(i, n); return i - n * s; }
HRT_DEV f3 refract(f3 i, f3 n, float eta)
{
    float d = dot(n, i);
    float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float s = eta * d + hrt_sqrt(k);
    return i * eta - n * s;
}
HRT_DEV f4 lerp4(f4 a, f4 b, float t)
{
    f4 r; float w = 1.0f - t;
    r.x = a.x * w + b.x * t; r.y = a.y * w + b.y * t; r.z = a.z * w + b.z * t; r.w = a.w * w + b.w * t;
    return r;
}

// ------------------------------------------------------------------ device scene view
struct GpuNode {            // 64 B: both child boxes live in the parent -> one 64 B read per step
    float lmin[3]; int32_t left;    // child >= 0: inner node index; child < 0: leaf = ~((first << 2) | (count - 1))
    float lmax[3]; int32_t right;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};
struct GpuNode4 {           // 128 B: four child boxes in SoA + four child refs (bvh_build.h HostNode4); empty slot = far-away box
    float4 minx, maxx, miny, maxy, minz, maxz; int4 child; uint4 pad;     // rows at byte 0, 16, ..., 80; child refs at 96
};
// 64 B: the same four child boxes as a GpuNode4, quantised to 8 bits per plane relative to the node's own extent (origin = min corner of the
// union of the child boxes, one fp32 step per axis), rounded OUTWARD (and checked against the decode arithmetic), so a decoded box contains the fp32 box it came
// from (hrpt_selftest_bvh checks every node). What it buys: a lane's node fetch is 4 sixteen-byte requests instead of 7, and the L1 (TCP)
// of a CU looks up about ONE lane-request per cycle whatever the request's width -- the traversal kernels on trees in global memory are
// bound by exactly that rate (scripts/microbench/gather_nodes.hip; three more requests per step cost config 4's wf_extend +45 %).
// Culling only: the hit definition never depends on the boxes. Rows: {ox, oy, oz, sx} {lo_x, hi_x, lo_y, hi_y}
// {lo_z, hi_z, sy, sz} {child[4]}; byte c of a plane word belongs to child c; an unused slot has lo = 255 > hi = 0 on every axis.
struct GpuNodeQ {
    float ox, oy, oz, sx;
    uint32_t lox, hix, loy, hiy;
    uint32_t loz, hiz; float sy, sz;
    int32_t child[4];
};
static_assert(sizeof(GpuNodeQ) == 64, "four 16-byte rows");
struct GpuTri {             // 48 B world-space triangle (instance transform applied at upload)
    float p0[3]; uint32_t inst;
    float p1[3]; uint32_t prim;
    float p2[3]; uint32_t flags;    // bit 0: instance is ForceOpaque (src/Scene.cpp:150-154)
};
// Per-triangle shading attributes, unpacked once at upload (bvh_build.h HostTriAttr), and per-instance adjugate rows.
struct GpuTriAttr { float4 a, b, c, d, e; };   // a{n0,n1.x} b{n1.yz,n2.xy} c{n2.z,uv0,uv1.x} d{uv1.y,uv2,material} e{inst,prim,-,-}
struct GpuTriTangent { float4 t0, t1, t2; };
struct GpuInstShade { float4 adj0, adj1, adj2; };
// Two-level structure (bvh_build.h HostInstance): one record per instance, 128 B
struct GpuInstance {
    float world[12];            // m_World rows 0..3, xyz each
    float inv[12];              // inverse map, same layout (only the 3x3 part is read on the device)
    int32_t blasRoot; uint32_t flags, material; float boxEps;
    uint32_t mesh; float objMaxAbs, invNorm; uint32_t pad;
};
static_assert(sizeof(GpuInstance) == 128, "bvh_build.h HostInstance");
struct GpuTexture {         // decoded texels of all levels (HrptTextureDesc); level l starts mipOffset[l] TEXELS after `texels`
    const uint8_t* texels; uint32_t w, h, format, mipCount;
    uint32_t mipOffset[HRPT_TEXTURE_MAX_MIPS];
};

struct SceneView {
    const GpuNode* nodes; uint32_t nodeCount;
    const GpuNode4* nodes4; uint32_t node4Count;   // the same tree collapsed to 4-wide nodes (wavefront kernels); root = 0
    const GpuNodeQ* nodesQ;                         // nodes4 in the 64-byte quantised form, same indices (flat structure only; null otherwise)
    const GpuTri* tris; uint32_t triCount;
    int32_t rootLeaf;       // when the whole scene fits one leaf: encoded leaf, else 0
    const GpuTriAttr* attrs;            // parallel to tris
    const GpuTriTangent* tangents;      // parallel to tris, or null when no material samples a normal map
    const GpuInstShade* instShade;      // per instance
    const HrptMaterialConstants* materials;
    const HrptGPULight* lights; uint32_t lightCount;
    const GpuTexture* textures; uint32_t textureCount;
    const uint16_t* lutTransmittance;   // 256 x 64 RGBA16F
    const uint16_t* lutScattering;      // 256 x 128 x 32 RGBA16F
    // two-level structure (instanced scenes; null = the flat world-space tree above): nodes4 then holds the tree over the instances
    // (root 0, leaves ~(instance << 2)) followed by the object-space trees of the distinct meshes, tris / attrs / tangents are per MESH
    // triangle (object space), and a hit carries its instance next to the triangle index.
    const GpuInstance* instances; uint32_t instanceCount;
};

// ------------------------------------------------------------------ rays and hits
struct Ray { f3 o, d; float tmin, tmax; };
struct Hit { float t; uint32_t inst, prim; float u, v; uint32_t opaque; uint32_t tri; bool valid; };
struct HitKey { float t; uint32_t inst, prim; bool have; };

struct RayShear { int kx, ky, kz; float Sx, Sy, Sz; };

HRT_DEV RayShear make_shear(f3 d)
{
    RayShear s;
    int kz = 0;
    if (hrt_abs(d.y) > hrt_abs(comp(d, kz))) kz = 1;
    if (hrt_abs(d.z) > hrt_abs(comp(d, kz))) kz = 2;
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    if (comp(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    s.kx = kx; s.ky = ky; s.kz = kz;
    float dz = comp(d, kz);
    s.Sx = comp(d, kx) / dz; s.Sy = comp(d, ky) / dz; s.Sz = 1.0f / dz;
    return s;
}
HRT_DEV bool key_less(float t, uint32_t inst, uint32_t prim, float t2, uint32_t inst2, uint32_t prim2)
{
    if (t < t2) return true;
    if (t > t2) return false;
    if (inst < inst2) return true;
    if (inst > inst2) return false;
    return prim < prim2;
}
// Watertight ray/triangle test (Woop, Benthin, Wald 2013) in fp32; no culling; (u,v) = DXR barycentrics.
HRT_DEV bool tri_test(f3 p0, f3 p1, f3 p2, const Ray& r, const RayShear& s, float& t, float& u, float& v)
{
    f3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
    float Akz = comp(A, s.kz), Bkz = comp(B, s.kz), Ckz = comp(C, s.kz);
    float Ax = comp(A, s.kx) - s.Sx * Akz, Ay = comp(A, s.ky) - s.Sy * Akz;
    float Bx = comp(B, s.kx) - s.Sx * Bkz, By = comp(B, s.ky) - s.Sy * Bkz;
    float Cx = comp(C, s.kx) - s.Sx * Ckz, Cy = comp(C, s.ky) - s.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    float Az = s.Sz * Akz, Bz = s.Sz * Bkz, Cz = s.Sz * Ckz;
    float T = (U * Az + V * Bz) + W * Cz;
    float rcp = 1.0f / det;
    float tt = T * rcp;
    if (!(tt > r.tmin && tt < r.tmax)) return false;
    t = tt; u = V * rcp; v = W * rcp;
    return true;
}

// Node/triangle fetch policy: the BVH is read either from HBM/L2 (global) or from an LDS copy.
struct GlobalBvh {
    static constexpr int kWidth = 2; static constexpr bool kTwoLevel = false;
    const GpuNode* nodes; const GpuTri* tris;
    HRT_DEV void node(int i, float4& a, float4& b, float4& c, float4& d) const
    {
        const float4* p = reinterpret_cast<const float4*>(nodes + i);
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
    }
    HRT_DEV void tri(uint32_t i, float4& a, float4& b, float4& c) const
    {
        const float4* p = reinterpret_cast<const float4*>(tris + i);
        a = p[0]; b = p[1]; c = p[2];
    }
};

struct GlobalBvh4 {
    static constexpr int kWidth = 4; static constexpr bool kTwoLevel = false; static constexpr bool kLds = false;
    const GpuNode4* nodes; const GpuTri* tris;
    // Nodes and triangles are addressed as (array base, 32-bit byte offset): the base stays in scalar registers and a lane holds one VGPR per
    // address instead of a pair (global_load saddr + voffset form; 64-bit address arithmetic is two VALU operations per add on gfx950).
    // Hence the flat structure's limits: fewer than 2^25 4-wide nodes and 2^32 / 48 triangles (bvh_build.cpp validate_scene).
    HRT_DEV void tri(uint32_t i, float4& a, float4& b, float4& c) const
    {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tris) + (size_t)(i * 48u));
        a = p[0]; b = p[1]; c = p[2];
    }
    // one 16-byte row of node i: byte offset 0/16 = min/max x of the four children, 32/48 = y, 64/80 = z, 96 = child refs
    HRT_DEV uint32_t rowoff(int i, uint32_t byteOffset) const { return (uint32_t)i * 128u + byteOffset; }
    HRT_DEV float4 load(uint32_t off) const { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + (size_t)off); }
};

// The flat 4-wide tree through its quantised nodes (GpuNodeQ): what the wavefront kernels traverse when the tree is not LDS-resident.
struct GlobalBvhQ {
    static constexpr int kWidth = 4; static constexpr bool kTwoLevel = false; static constexpr bool kLds = false; static constexpr bool kQuantised = true;
    const GpuNodeQ* nodes; const GpuTri* tris;
    HRT_DEV void tri(uint32_t i, float4& a, float4& b, float4& c) const
    {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tris) + (size_t)(i * 48u));
        a = p[0]; b = p[1]; c = p[2];
    }
    HRT_DEV float4 row(int i, uint32_t r) const { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + (size_t)((uint32_t)i * 64u + r * 16u)); }
};
template <class BVH, class = void> struct IsQuantised { static constexpr bool value = false; };
template <class BVH> struct IsQuantised<BVH, decltype((void)BVH::kQuantised)> { static constexpr bool value = true; };

// Conservative slab test of one child box against [t0, t1]. Culling only: it never changes which hit is
// reported (the hit definition is BVH-independent), so it may use native min/max, the hardware reciprocal and
// FMA. A NaN plane distance (zero direction component with the origin exactly on a padded slab plane) culls the
// box, which is still conservative: padded planes lie strictly outside every triangle extent inside the box.
// Plane distances are one FMA each: t = plane * inv + noi with noi = -(o * inv) computed once per ray. Its error,
// ~6e-8 * |o * inv|, is covered either by the 2e-6 relative widening (when the plane is far from the origin compared with |o|)
// or by the box padding 1e-5 * |coord| + 1e-6 (when it is not). inv is finite (traversal_rcp caps it), so no NaN arises.
HRT_DEV f3 slab_origin_term(f3 o, f3 inv) { return mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)); }
HRT_DEV bool slab(float4 bmin, float4 bmax, f3 noi, f3 inv, float t0, float t1, float& tnear)
{
    float tx0 = __builtin_fmaf(bmin.x, inv.x, noi.x), tx1 = __builtin_fmaf(bmax.x, inv.x, noi.x);
    float ty0 = __builtin_fmaf(bmin.y, inv.y, noi.y), ty1 = __builtin_fmaf(bmax.y, inv.y, noi.y);
    float tz0 = __builtin_fmaf(bmin.z, inv.z, noi.z), tz1 = __builtin_fmaf(bmax.z, inv.z, noi.z);
    float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fmaxf(__builtin_fminf(tz0, tz1), t0));
    float hi = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fminf(__builtin_fmaxf(tz0, tz1), t1));
    tnear = lo;
    // widen by 2e-6 relative for the rounding of the plane distances
    return __builtin_fmaf(-__builtin_fabsf(lo), 2e-6f, lo) <= __builtin_fmaf(__builtin_fabsf(hi), 2e-6f, hi);
}

constexpr int32_t kTraversalDone = (int32_t)0x80000000;   // not a valid leaf encoding (first < 2^29)

// one box of a 4-wide node: entry distance, or +inf on a miss
HRT_DEV float slab1(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 noi, f3 inv, float t0, float t1)
{
    float tx0 = __builtin_fmaf(bminx, inv.x, noi.x), tx1 = __builtin_fmaf(bmaxx, inv.x, noi.x);
    float ty0 = __builtin_fmaf(bminy, inv.y, noi.y), ty1 = __builtin_fmaf(bmaxy, inv.y, noi.y);
    float tz0 = __builtin_fmaf(bminz, inv.z, noi.z), tz1 = __builtin_fmaf(bmaxz, inv.z, noi.z);
    float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fmaxf(__builtin_fminf(tz0, tz1), t0));
    float hi = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fminf(__builtin_fmaxf(tz0, tz1), t1));
    bool hit = __builtin_fmaf(-__builtin_fabsf(lo), 2e-6f, lo) <= __builtin_fmaf(__builtin_fabsf(hi), 2e-6f, hi);
    return hit ? lo : __builtin_inff();
}
// Compare-swap of two (entry distance, child) pairs: the pair with the smaller distance ends up in (ta, ra). One compare + four selects, all
// half-rate instructions on gfx950, and the five swaps of a node step are 25 of its 70 half-rate instructions -- but the integer form (mask
// (ib - ia) >> 31 of the bit patterns, masked XOR exchange: one shift + five full-rate integer / v_bitop3 operations) measured 4 % SLOWER on
// configs 2 and 4: full-rate integer and logic operations do not overlap with half-rate ones the way FMAs do.
HRT_DEV void cswap(float& ta, int32_t& ra, float& tb, int32_t& rb)
{
    bool sw = tb < ta;
    float t = sw ? tb : ta; tb = sw ? ta : tb; ta = t;
    int32_t r = sw ? rb : ra; rb = sw ? ra : rb; ra = r;
}

// One traversal step from inner node `cur`: tests its child boxes against [tmin, tlim], pushes the far hits (nearest on
// top) and returns the next node reference: the nearest hit child, else the popped stack top, else kTraversalDone.
// noi / noiF: the origin term of the near / far plane distances. They differ only inside an instance of the two-level structure, where
// every box is widened by a per-ray object-space slack (tl_enter); everywhere else the caller passes the same value twice.
template <class BVH, class STACK>
HRT_DEV int32_t inner_step(const BVH& bvh, int32_t cur, f3 noi, f3 noiF, f3 inv, float tmin, float tlim, STACK& stack, int& sp)
{
    if constexpr (BVH::kWidth == 2) {
        float4 a, b, c, d; bvh.node(cur, a, b, c, d);
        int32_t li = __float_as_int(a.w), ri = __float_as_int(b.w);
        float tl, tr;
        bool hl = slab(a, b, noi, inv, tmin, tlim, tl);
        bool hr = slab(c, d, noi, inv, tmin, tlim, tr);
        if (hl && hr) { bool leftFirst = tl <= tr; stack.push(sp++, leftFirst ? ri : li); return leftFirst ? li : ri; }
        if (hl) return li;
        if (hr) return ri;
        return (sp == 0) ? kTraversalDone : stack.pop(--sp);
    } else if constexpr (IsQuantised<BVH>::value) {
        // 64-byte quantised node: four requests. Plane distance of byte q of an axis: (o + q * s - ray.o) * inv = q * (s * inv) + (o * inv + noi);
        // the near / far plane WORDS of an axis are picked by the sign of the direction, the byte of child c by v_cvt_f32_ubyte<c>.
        const float4 r0 = bvh.row(cur, 0u), r1 = bvh.row(cur, 1u), r2 = bvh.row(cur, 2u), chf = bvh.row(cur, 3u);
        const float ax = r0.w * inv.x, ay = r2.z * inv.y, az = r2.w * inv.z;
        const float bx = __builtin_fmaf(r0.x, inv.x, noi.x), by = __builtin_fmaf(r0.y, inv.y, noi.y), bz = __builtin_fmaf(r0.z, inv.z, noi.z);
        const bool negx = inv.x < 0.0f, negy = inv.y < 0.0f, negz = inv.z < 0.0f;
        const uint32_t wnx = __float_as_uint(negx ? r1.y : r1.x), wfx = __float_as_uint(negx ? r1.x : r1.y);
        const uint32_t wny = __float_as_uint(negy ? r1.w : r1.z), wfy = __float_as_uint(negy ? r1.z : r1.w);
        const uint32_t wnz = __float_as_uint(negz ? r2.y : r2.x), wfz = __float_as_uint(negz ? r2.x : r2.y);
        auto boxq = [&](uint32_t sh) {
            const float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)((wnx >> sh) & 255u), ax, bx), __builtin_fmaf((float)((wny >> sh) & 255u), ay, by)),
                                                             __builtin_fmaf((float)((wnz >> sh) & 255u), az, bz)), tmin);
            const float far3 = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)((wfx >> sh) & 255u), ax, bx), __builtin_fmaf((float)((wfy >> sh) & 255u), ay, by)),
                                               __builtin_fmaf((float)((wfz >> sh) & 255u), az, bz));
            const float hi = __builtin_fminf(far3, tlim) * (1.0f + 4e-6f);       // as below: the scale after the clamp to tlim
            return lo <= hi ? lo : __builtin_inff();
        };
        float t0 = boxq(0u), t1 = boxq(8u), t2 = boxq(16u), t3 = boxq(24u);
        int32_t r0i = __float_as_int(chf.x), r1i = __float_as_int(chf.y), r2i = __float_as_int(chf.z), r3i = __float_as_int(chf.w);
        cswap(t0, r0i, t1, r1i); cswap(t2, r2i, t3, r3i); cswap(t0, r0i, t2, r2i); cswap(t1, r1i, t3, r3i); cswap(t1, r1i, t2, r2i);
        const float inf = __builtin_inff();
        if (t3 < inf) stack.push(sp++, r3i);
        if (t2 < inf) stack.push(sp++, r2i);
        if (t1 < inf) stack.push(sp++, r1i);
        if (t0 < inf) return r0i;
        return (sp == 0) ? kTraversalDone : stack.pop(--sp);
    } else {
        // The sign of the ray direction says which plane of every slab is the near one, so the near / far rows of the node are picked by
        // ADDRESS (min / max rows of an axis are 16 bytes apart: far = near ^ 16) instead of by six min/max per child box, and the 2e-6
        // relative widening of both ends is folded into one 4e-6 scale of the far end (hi > 0 whenever the test can pass: lo >= tmin >= 0).
        const uint32_t nx = inv.x < 0.0f ? 16u : 0u, ny = inv.y < 0.0f ? 48u : 32u, nz = inv.z < 0.0f ? 80u : 64u;
        // (the far address is derived from the NEAR ADDRESS, not from the offset: offsets are loop-invariant per ray and the compiler
        // would otherwise keep all six in registers -- +6 VGPRs cost the kernel a wave of occupancy; nodes are 128-byte aligned)
        const uint32_t onx = bvh.rowoff(cur, nx), ony = bvh.rowoff(cur, ny), onz = bvh.rowoff(cur, nz);
        const float4 anx = bvh.load(onx), any = bvh.load(ony), anz = bvh.load(onz);
        const float4 afx = bvh.load(onx ^ 16u), afy = bvh.load(ony ^ 16u), afz = bvh.load(onz ^ 16u);
        const float4 chf = bvh.load(bvh.rowoff(cur, 96u));
        auto box = [&](float bnx, float bny, float bnz, float bfx, float bfy, float bfz) {
            const float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(bnx, inv.x, noi.x), __builtin_fmaf(bny, inv.y, noi.y)), __builtin_fmaf(bnz, inv.z, noi.z)), tmin);
            const float far3 = __builtin_fminf(__builtin_fminf(__builtin_fmaf(bfx, inv.x, noiF.x), __builtin_fmaf(bfy, inv.y, noiF.y)), __builtin_fmaf(bfz, inv.z, noiF.z));
            // the scale comes AFTER the clamp to tlim: a second triangle at exactly the distance of the best hit so far (shared edges,
            // coplanar duplicates: the tie is decided by (instance, primitive)) must survive a near distance that rounds just above tlim
            const float hi = __builtin_fminf(far3, tlim) * (1.0f + 4e-6f);
            return lo <= hi ? lo : __builtin_inff();
        };
        float t0 = box(anx.x, any.x, anz.x, afx.x, afy.x, afz.x);
        float t1 = box(anx.y, any.y, anz.y, afx.y, afy.y, afz.y);
        float t2 = box(anx.z, any.z, anz.z, afx.z, afy.z, afz.z);
        float t3 = box(anx.w, any.w, anz.w, afx.w, afy.w, afz.w);
        int32_t r0 = __float_as_int(chf.x), r1 = __float_as_int(chf.y), r2 = __float_as_int(chf.z), r3 = __float_as_int(chf.w);
        cswap(t0, r0, t1, r1); cswap(t2, r2, t3, r3); cswap(t0, r0, t2, r2); cswap(t1, r1, t3, r3); cswap(t1, r1, t2, r2);
        const float inf = __builtin_inff();
        if (t3 < inf) stack.push(sp++, r3);
        if (t2 < inf) stack.push(sp++, r2);
        if (t1 < inf) stack.push(sp++, r1);
        if (t0 < inf) return r0;
        return (sp == 0) ? kTraversalDone : stack.pop(--sp);
    }
}

template <class BVH, class STACK>
HRT_DEV int32_t inner_step(const BVH& bvh, int32_t cur, f3 noi, f3 inv, float tmin, float tlim, STACK& stack, int& sp)
{
    return inner_step(bvh, cur, noi, noi, inv, tmin, tlim, stack, sp);
}

HRT_DEV f3 traversal_rcp(f3 d)
{
    // |1/d| is capped at 1e20 (a zero component would give inf, and inf * plane - inf * origin is NaN in the FMA slab form): over
    // t <= 1e10 such a ray moves less than 1e-10 along that axis, far inside the 1e-6 box padding, so the capped interval still
    // contains every t at which the ray is inside the box.
    auto one = [](float x) { return __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_rcpf(x), -1e20f), 1e20f); };
    return mk3(one(d.x), one(d.y), one(d.z));
}

// Closest triangle (opaque or not) with key strictly above `lower` (when lower.have) in (t, inst, prim) order.
// while-while traversal: all lanes of the wave first descend inner nodes, then intersect leaves together.
// STACK: per-lane traversal stack accessor (LDS column or private array).
// nodeLoopMin: the descent loop ends once fewer lanes than this are still at inner nodes (thresholded while-while; 0 = run to a leaf)
template <class BVH, class STACK>
HRT_DEV Hit closest_any(const BVH& bvh, int32_t rootLeaf, uint32_t nodeCount, const Ray& r, HitKey lower, STACK& stack, uint32_t nodeLoopMin = 0)
{
    Hit best; best.valid = false; best.t = r.tmax; best.inst = 0; best.prim = 0; best.u = 0; best.v = 0; best.opaque = 0; best.tri = 0;
    if (!(r.d.x == r.d.x && r.d.y == r.d.y && r.d.z == r.d.z)) return best;
    RayShear sh = make_shear(r.d);
    f3 inv = traversal_rcp(r.d), noi = slab_origin_term(r.o, inv);
    int sp = 0;
    int32_t cur;                       // current node reference: >= 0 inner, < 0 leaf
    if (nodeCount == 0) { if (rootLeaf == 0) return best; cur = rootLeaf; }
    else cur = 0;
    float tlim = r.tmax;               // == best.t once a hit exists
    for (;;) {
        while (cur >= 0) {
            cur = inner_step(bvh, cur, noi, inv, r.tmin, tlim, stack, sp);
            if ((uint32_t)__popcll(__ballot(cur >= 0)) < nodeLoopMin) break;
        }
        if (cur == kTraversalDone) break;
        if (cur < 0) {
            uint32_t enc = (uint32_t)(~cur);
            uint32_t first = enc >> 2, count = (enc & 3u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                float4 a, b, c; bvh.tri(first + i, a, b, c);
                float t, u, v;
                if (tri_test(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), r, sh, t, u, v)) {
                    uint32_t inst = __float_as_uint(a.w), prim = __float_as_uint(b.w);
                    bool ok = !lower.have || key_less(lower.t, lower.inst, lower.prim, t, inst, prim);
                    const bool take = ok && (!best.valid || key_less(t, inst, prim, best.t, best.inst, best.prim));      // selects on one flag: pt_wavefront.hip wf_extend<TL>
                    best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                    best.inst = take ? inst : best.inst; best.prim = take ? prim : best.prim; best.tri = take ? first + i : best.tri;
                    best.opaque = take ? (__float_as_uint(c.w) & 1u) : best.opaque;
                    best.valid = best.valid || take;
                    tlim = take ? t : tlim;
                }
            }
            if (sp == 0) break;
            cur = stack.pop(--sp);
        }
    }
    return best;
}

// Any opaque-instance triangle in (tmin, tmax)? Early exit. Non-opaque triangles are only noted.
template <class BVH, class STACK>
HRT_DEV bool any_opaque(const BVH& bvh, int32_t rootLeaf, uint32_t nodeCount, const Ray& r, STACK& stack, bool& sawNonOpaque, uint32_t nodeLoopMin = 0)
{
    sawNonOpaque = false;
    if (!(r.d.x == r.d.x && r.d.y == r.d.y && r.d.z == r.d.z)) return false;
    RayShear sh = make_shear(r.d);
    f3 inv = traversal_rcp(r.d), noi = slab_origin_term(r.o, inv);
    int sp = 0; int32_t cur;
    if (nodeCount == 0) { if (rootLeaf == 0) return false; cur = rootLeaf; }
    else cur = 0;
    for (;;) {
        while (cur >= 0) {
            cur = inner_step(bvh, cur, noi, inv, r.tmin, r.tmax, stack, sp);
            if ((uint32_t)__popcll(__ballot(cur >= 0)) < nodeLoopMin) break;
        }
        if (cur == kTraversalDone) break;
        if (cur < 0) {
            uint32_t enc = (uint32_t)(~cur);
            uint32_t first = enc >> 2, count = (enc & 3u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                float4 a, b, c; bvh.tri(first + i, a, b, c);
                float t, u, v;
                if (tri_test(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), r, sh, t, u, v)) {
                    if (__float_as_uint(c.w) & 1u) return true;
                    sawNonOpaque = true;
                }
            }
            if (sp == 0) break;
            cur = stack.pop(--sp);
        }
    }
    return false;
}

// ------------------------------------------------------------------ two-level traversal (instanced scenes)
// The tree over the instances is walked with the world-space ray. At an instance leaf the lane pushes kExitBlas, switches its CULLING
// state (inv, noi) to the ray in the instance's object space and continues in the mesh's tree; popping kExitBlas switches back. Triangles
// are never tested in object space: a leaf transforms its three object-space vertices with m_World exactly as the flat upload does
// (bvh_build.cpp transform_point: left to right, no FMA) and runs the same watertight test on the world ray, so the hit (t, u, v) and the
// (t, instance, primitive) order are those of the flat path bit for bit. The object-space ray keeps the world parametrisation (its
// direction is d * Minv, not normalised), so tlim needs no conversion.
constexpr int32_t kExitBlas = 0x7FFFFFFE;     // stack marker; never a node index (node4 counts stay below 2^29)
struct GlobalBvhTl : GlobalBvh4 {
    static constexpr bool kTwoLevel = true;
    const GpuInstance* instances;
};
struct TlCull { f3 inv, noi, noiF; int32_t inst; };     // inst < 0: in the tree over the instances (world space, noi == noiF)
HRT_DEV void tl_world(TlCull& c, const Ray& r)
{
    c.inv = traversal_rcp(r.d); c.noi = slab_origin_term(r.o, c.inv); c.noiF = c.noi; c.inst = -1;
}
// Culling state inside instance I. Everything that separates the computed object-space ray from the exact image of the world ray, and
// the object-space triangles from the (rounded) world triangles the test runs on, is bounded by `eps` and added to both sides of every
// slab of the mesh's tree:
//   - I.boxEps: the binary32 rounding of the transformed vertices, mapped back through |Minv| (host, bvh_build.cpp)
//   - the rounding of o - T (half an ulp of the larger operand) and of the three products / two sums per component of (o - T) * Minv,
//     d * Minv, and of Minv itself (rounded from binary64): 2.4e-7 relative to the operands' magnitudes times ||Minv||
//   - the direction error acts over the parameter range in which the ray can be inside the mesh's box: t * |d'| <= objMaxAbs + |o'|
// with a factor 4 on top. Over-estimating eps only costs culling.
HRT_DEV void tl_enter(TlCull& c, const GpuInstance& I, int32_t inst, const Ray& r)
{
    const float4* w = reinterpret_cast<const float4*>(I.world);
    const float4 w2 = w[2];                                        // {M22, T.x, T.y, T.z}
    const float4* q = reinterpret_cast<const float4*>(I.inv);
    const float4 q0 = q[0], q1 = q[1]; const float q22 = I.inv[8];    // rows of Minv: {a00 a01 a02 a10} {a11 a12 a20 a21} a22
    const f3 rel = mk3(r.o.x - w2.y, r.o.y - w2.z, r.o.z - w2.w);
    const f3 oo = mk3(__builtin_fmaf(rel.z, q1.z, __builtin_fmaf(rel.y, q0.w, rel.x * q0.x)),
                      __builtin_fmaf(rel.z, q1.w, __builtin_fmaf(rel.y, q1.x, rel.x * q0.y)),
                      __builtin_fmaf(rel.z, q22, __builtin_fmaf(rel.y, q1.y, rel.x * q0.z)));
    const f3 od = mk3(__builtin_fmaf(r.d.z, q1.z, __builtin_fmaf(r.d.y, q0.w, r.d.x * q0.x)),
                      __builtin_fmaf(r.d.z, q1.w, __builtin_fmaf(r.d.y, q1.x, r.d.x * q0.y)),
                      __builtin_fmaf(r.d.z, q22, __builtin_fmaf(r.d.y, q1.y, r.d.x * q0.z)));
    auto max3abs = [](f3 v) { return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z)); };
    const float oMax = __builtin_fmaxf(max3abs(r.o), __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(w2.y), __builtin_fabsf(w2.z)), __builtin_fabsf(w2.w)));
    const float dRatio = max3abs(r.d) * I.invNorm / __builtin_fmaxf(max3abs(od), 1e-30f);
    const float eps = I.boxEps + 4.0f * 2.4e-7f * (I.invNorm * (oMax + max3abs(rel)) + dRatio * (I.objMaxAbs + max3abs(oo)));
    c.inv = traversal_rcp(od);
    const f3 n = slab_origin_term(oo, c.inv);
    const f3 e = mk3(eps * __builtin_fabsf(c.inv.x), eps * __builtin_fabsf(c.inv.y), eps * __builtin_fabsf(c.inv.z));
    c.noi = mk3(n.x - e.x, n.y - e.y, n.z - e.z); c.noiF = mk3(n.x + e.x, n.y + e.y, n.z + e.z);
    c.inst = inst;
}
// world-space vertices of object-space triangle (a, b, c) of instance I: transform_point of bvh_build.cpp (mul(float4(p,1), M).xyz, left to right)
HRT_DEV void tl_world_triangle(const GpuInstance& I, float4 a, float4 b, float4 c, f3& p0, f3& p1, f3& p2)
{
    const float4* w = reinterpret_cast<const float4*>(I.world);
    const float4 w0 = w[0], w1 = w[1], w2 = w[2];      // {M00 M01 M02 M10} {M11 M12 M20 M21} {M22 Tx Ty Tz}
    auto xf = [&](float x, float y, float z) {
        return mk3(((x * w0.x + y * w0.w) + z * w1.z) + w2.y,
                   ((x * w0.y + y * w1.x) + z * w1.w) + w2.z,
                   ((x * w0.z + y * w1.y) + z * w2.x) + w2.w);
    };
    p0 = xf(a.x, a.y, a.z); p1 = xf(b.x, b.y, b.z); p2 = xf(c.x, c.y, c.z);
}
// One node reference of the two-level walk that is not an inner node of the current tree: the BLAS exit marker or an instance leaf.
// Returns the next reference (and updates the culling state); triangle leaves (c.inst >= 0, cur < 0) are the caller's.
template <class STACK>
HRT_DEV int32_t tl_switch(const GlobalBvhTl& bvh, int32_t cur, TlCull& c, const Ray& r, STACK& stack, int& sp)
{
    if (cur == kExitBlas) { tl_world(c, r); return sp == 0 ? kTraversalDone : stack.pop(--sp); }
    // instance leaf
    const int32_t inst = (int32_t)(((uint32_t)~cur) >> 2);
    const GpuInstance& I = bvh.instances[inst];
    const int32_t root = I.blasRoot;
    if (root == kTraversalDone) return sp == 0 ? kTraversalDone : stack.pop(--sp);      // mesh without triangles
    stack.push(sp++, kExitBlas);
    tl_enter(c, I, inst, r);
    return root;
}

// any_opaque over the two-level structure: true at the first triangle of a ForceOpaque instance; triangles of other instances are only noted
template <class STACK>
HRT_DEV bool any_hit_two_level(const GlobalBvhTl& bvh, int32_t rootLeaf, uint32_t nodeCount, const Ray& r, STACK& stack, bool& sawNonOpaque, uint32_t nodeLoopMin = 0)
{
    sawNonOpaque = false;
    if (!(r.d.x == r.d.x && r.d.y == r.d.y && r.d.z == r.d.z)) return false;
    RayShear sh = make_shear(r.d);
    TlCull c; tl_world(c, r);
    int sp = 0; int32_t cur;
    if (nodeCount == 0) { if (rootLeaf == 0) return false; cur = rootLeaf; }
    else cur = 0;
    for (;;) {
        while (cur >= 0 && cur != kExitBlas) {
            cur = inner_step(bvh, cur, c.noi, c.noiF, c.inv, r.tmin, r.tmax, stack, sp);
            if ((uint32_t)__popcll(__ballot(cur >= 0 && cur != kExitBlas)) < nodeLoopMin) break;
        }
        if (cur == kTraversalDone) break;
        if (cur == kExitBlas || (cur < 0 && c.inst < 0)) { cur = tl_switch(bvh, cur, c, r, stack, sp); continue; }
        if (cur < 0) {
            const GpuInstance& I = bvh.instances[c.inst];
            uint32_t enc = (uint32_t)(~cur);
            uint32_t first = enc >> 2, count = (enc & 3u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                float4 ta, tb, tc; bvh.tri(first + i, ta, tb, tc);
                f3 p0, p1, p2; tl_world_triangle(I, ta, tb, tc, p0, p1, p2);
                float t, u, v;
                if (tri_test(p0, p1, p2, r, sh, t, u, v)) { if (I.flags & 1u) return true; sawNonOpaque = true; }
            }
            cur = stack.pop(--sp);          // never empty inside an instance: the exit marker is below
        }
    }
    return false;
}
// closest_any over the two-level structure: the closest triangle with key strictly above `lower` (candidate re-trace of the shadow query)
template <class STACK>
HRT_DEV Hit closest_two_level(const GlobalBvhTl& bvh, int32_t rootLeaf, uint32_t nodeCount, const Ray& r, HitKey lower, STACK& stack)
{
    Hit best; best.valid = false; best.t = r.tmax; best.inst = 0; best.prim = 0; best.u = 0; best.v = 0; best.opaque = 0; best.tri = 0;
    if (!(r.d.x == r.d.x && r.d.y == r.d.y && r.d.z == r.d.z)) return best;
    RayShear sh = make_shear(r.d);
    TlCull c; tl_world(c, r);
    int sp = 0; int32_t cur;
    if (nodeCount == 0) { if (rootLeaf == 0) return best; cur = rootLeaf; }
    else cur = 0;
    float tlim = r.tmax;
    for (;;) {
        while (cur >= 0 && cur != kExitBlas) cur = inner_step(bvh, cur, c.noi, c.noiF, c.inv, r.tmin, tlim, stack, sp);
        if (cur == kTraversalDone) break;
        if (cur == kExitBlas || (cur < 0 && c.inst < 0)) { cur = tl_switch(bvh, cur, c, r, stack, sp); continue; }
        const GpuInstance& I = bvh.instances[c.inst];
        const uint32_t enc = (uint32_t)(~cur), first = enc >> 2, count = (enc & 3u) + 1u, inst = (uint32_t)c.inst;
        for (uint32_t i = 0; i < count; ++i) {
            float4 ta, tb, tc; bvh.tri(first + i, ta, tb, tc);
            f3 p0, p1, p2; tl_world_triangle(I, ta, tb, tc, p0, p1, p2);
            float t, u, v;
            const bool hitTri = tri_test(p0, p1, p2, r, sh, t, u, v);
            const uint32_t prim = __float_as_uint(tb.w);
            const bool ok = !lower.have || key_less(lower.t, lower.inst, lower.prim, t, inst, prim);
            const bool take = hitTri && ok && (!best.valid || key_less(t, inst, prim, best.t, best.inst, best.prim));
            best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
            best.inst = take ? inst : best.inst; best.prim = take ? prim : best.prim; best.tri = take ? first + i : best.tri;
            best.opaque = take ? (I.flags & 1u) : best.opaque;
            best.valid = best.valid || take;
            tlim = take ? t : tlim;
        }
        cur = stack.pop(--sp);
    }
    return best;
}

// ------------------------------------------------------------------ triangle attributes
// What GetTriangleVertices + UnpackVertex (RaytracingCommon.hlsli:33-50, MeshCommon.hlsli:9-22) yield for a hit,
// read from the per-triangle record built at upload (the quantised vertex / index buffers are consumed there).
struct TriVerts { f3 n0, n1, n2; f2 uv0, uv1, uv2; uint32_t material, inst; };
HRT_DEV TriVerts load_tri_attr(const SceneView& s, uint32_t tri)
{
    const float4* p = reinterpret_cast<const float4*>(s.attrs + tri);
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    TriVerts t;
    t.n0 = mk3(a.x, a.y, a.z); t.n1 = mk3(a.w, b.x, b.y); t.n2 = mk3(b.z, b.w, c.x);
    t.uv0.x = c.y; t.uv0.y = c.z; t.uv1.x = c.w; t.uv1.y = d.x; t.uv2.x = d.y; t.uv2.y = d.z;
    t.material = __float_as_uint(d.w); t.inst = __float_as_uint(e.x);
    return t;
}
// ... for a committed hit: with the two-level structure the record is per MESH triangle and the instance (hence the material) comes
// from the hit
HRT_DEV TriVerts load_hit_attr(const SceneView& s, const Hit& hit)
{
    TriVerts tv = load_tri_attr(s, hit.tri);
    if (s.instances) { tv.inst = hit.inst; tv.material = s.instances[hit.inst].material; }
    return tv;
}
// GetInterpolatedUV, RaytracingCommon.hlsli:79-89
HRT_DEV f2 interpolated_uv(const TriVerts& tv, float bx, float by)
{
    float w0 = (1.0f - bx) - by; f2 r;
    r.x = (tv.uv0.x * w0 + tv.uv1.x * bx) + tv.uv2.x * by;
    r.y = (tv.uv0.y * w0 + tv.uv1.y * bx) + tv.uv2.y * by;
    return r;
}
// TransformNormal, Common.hlsli:33-47
HRT_DEV f3 transform_normal(f3 n, const GpuInstShade& is)
{
    // adjugate rows cross(r1,r2), cross(r2,r0), cross(r0,r1) are precomputed per instance with the same arithmetic
    f3 a0 = mk3(is.adj0.x, is.adj0.y, is.adj0.z), a1 = mk3(is.adj1.x, is.adj1.y, is.adj1.z), a2 = mk3(is.adj2.x, is.adj2.y, is.adj2.z);
    f3 o = mk3((n.x * a0.x + n.y * a1.x) + n.z * a2.x,
               (n.x * a0.y + n.y * a1.y) + n.z * a2.y,
               (n.x * a0.z + n.y * a1.z) + n.z * a2.z);
    return normalize(o);
}

// ------------------------------------------------------------------ textures / LUTs
HRT_DEV int wrap_i(int i, int n, bool wrap)
{
    if (wrap) { int m = i % n; return m < 0 ? m + n : m; }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// byte / 255.0f, correctly rounded, without the IEEE division sequence: one Newton step on q = x * fl(1/255) with FMAs gives the
// correctly rounded quotient for every x in 0..255 (all 256 inputs are checked against the division on the device by
// tests/test_parity_gpu.py::test_device_unorm8_table; a textured hit decodes up to 80 channels, 3 instructions each instead of ~11).
HRT_DEV float unorm8_to_float(uint32_t byte)
{
    const float x = (float)byte, c = 0x1.010102p-8f;
    const float q = x * c;
    const float r = __builtin_fmaf(-q, 255.0f, x);
    return __builtin_fmaf(r, c, q);
}
// binary16 -> binary32 is exact for every input, so the hardware conversion (v_cvt_f32_f16, fp16 denormals enabled --
// the HIP default) gives the same bits as the integer decode of detmath.h (hrt_f16tof32) used on the host; checked for
// all 65536 encodings by tests/test_parity_gpu.py::test_device_f16_decode_table.
HRT_DEV float half_bits_to_float(uint32_t h)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}
static __device__ __constant__ const float kSrgbToLinear[256] = HRT_SRGB_TO_LINEAR_TABLE;
// One texel of a format other than RGBA8_UNORM: RGBA8_SRGB (r, g, b linearised BEFORE filtering, as a D3D12 sampler does for *_SRGB
// formats), RGBA16_FLOAT, RGBA32_FLOAT.
HRT_DEV f4 texel_fetch_other(const uint8_t* texels, uint32_t format, size_t idx)
{
    f4 r;
    if (format == HRPT_TEXTURE_FORMAT_RGBA8_SRGB) {
        const uint32_t p = reinterpret_cast<const uint32_t*>(texels)[idx];
        r.x = kSrgbToLinear[p & 255u]; r.y = kSrgbToLinear[(p >> 8) & 255u]; r.z = kSrgbToLinear[(p >> 16) & 255u]; r.w = unorm8_to_float(p >> 24);
    } else if (format == HRPT_TEXTURE_FORMAT_RGBA16_FLOAT) {
        const uint2 p = reinterpret_cast<const uint2*>(texels)[idx];
        r.x = half_bits_to_float(p.x & 0xffffu); r.y = half_bits_to_float(p.x >> 16); r.z = half_bits_to_float(p.y & 0xffffu); r.w = half_bits_to_float(p.y >> 16);
    } else {
        const float4 p = reinterpret_cast<const float4*>(texels)[idx];
        r.x = p.x; r.y = p.y; r.z = p.z; r.w = p.w;
    }
    return r;
}
HRT_DEV f4 texel_unorm8(const uint8_t* texels, size_t idx)
{
    const uint32_t p = reinterpret_cast<const uint32_t*>(texels)[idx];
    f4 r;
    r.x = unorm8_to_float(p & 255u); r.y = unorm8_to_float((p >> 8) & 255u); r.z = unorm8_to_float((p >> 16) & 255u); r.w = unorm8_to_float(p >> 24);
    return r;
}
// One level of a texture sampled like SampleLevel on the samplers of src/CommonResources.cpp:117-128 (0/1 anisotropic, 2/3 point, 4/5
// linear; odd = wrap, even = clamp), bilinear with exact fp32 weights: a(1 - t) + bt along x, then along y. levelOffset = first texel of
// the level (0 for level 0). RGBA8_UNORM -- the format of every stb-decoded image, the hot path of textured scenes -- keeps its straight-
// line code; the other formats share ONE copy of the fetch in a rolled loop over the four corners (inlined per corner and per texture slot
// they doubled wf_shade's memory instructions and cost wf_shadow a wave of occupancy: +18 % / +30 % on the Sponza-class config).
HRT_DEV f4 sample_texture_level(const uint8_t* texels, uint32_t format, int lw, int lh, uint32_t levelOffset, uint32_t samplerIndex, f2 uv)
{
    bool wrap = (samplerIndex <= 5u) ? ((samplerIndex & 1u) != 0) : false;
    bool point = (samplerIndex == 2u || samplerIndex == 3u);
    float fx = uv.x * (float)lw, fy = uv.y * (float)lh;
    if (!point) { fx = fx - 0.5f; fy = fy - 0.5f; }
    float ix = hrt_floor(fx), iy = hrt_floor(fy);
    float tx = point ? 0.0f : fx - ix, ty = point ? 0.0f : fy - iy;
    int x0 = wrap_i((int)ix, lw, wrap), x1 = wrap_i((int)ix + 1, lw, wrap);
    int y0 = wrap_i((int)iy, lh, wrap), y1 = wrap_i((int)iy + 1, lh, wrap);
    const size_t r0 = (size_t)levelOffset + (size_t)y0 * (size_t)lw, r1 = (size_t)levelOffset + (size_t)y1 * (size_t)lw;
    if (format == HRPT_TEXTURE_FORMAT_RGBA8_UNORM) {
        if (point) return texel_unorm8(texels, r0 + (size_t)x0);
        f4 a = lerp4(texel_unorm8(texels, r0 + (size_t)x0), texel_unorm8(texels, r0 + (size_t)x1), tx);
        f4 b = lerp4(texel_unorm8(texels, r1 + (size_t)x0), texel_unorm8(texels, r1 + (size_t)x1), tx);
        return lerp4(a, b, ty);
    }
    if (point) return texel_fetch_other(texels, format, r0 + (size_t)x0);
    f4 rowA, rowB; rowA.x = rowA.y = rowA.z = rowA.w = 0.0f; rowB = rowA;
#pragma nounroll
    for (uint32_t k = 0; k < 2u; ++k) {             // one copy of the two-texel row fetch
        const size_t r = k ? r1 : r0;
        const f4 v = lerp4(texel_fetch_other(texels, format, r + (size_t)x0), texel_fetch_other(texels, format, r + (size_t)x1), tx);
        if (k == 0u) rowA = v; else rowB = v;
    }
    return lerp4(rowA, rowB, ty);
}
// SampleBindlessTextureLevel(lod 0), Bindless.hlsli:118-123.
HRT_DEV f4 sample_texture(const SceneView& s, uint32_t texIndex, uint32_t samplerIndex, f2 uv)
{
    f4 zero; zero.x = zero.y = zero.z = zero.w = 0.0f;
    if (texIndex >= s.textureCount) return zero;
    const GpuTexture& t = s.textures[texIndex];
    const uint8_t* texels = t.texels;
    if (!texels) return zero;
    return sample_texture_level(texels, t.format, (int)t.w, (int)t.h, 0u, samplerIndex, uv);
}
// SampleBindlessTextureGrad, Bindless.hlsli:127-132 (tex.SampleGrad): D3D leaves the level-of-detail computation to the implementation
// within a tolerance; defined here (numeric contract) as the isotropic form of the D3D11.3 functional spec 7.18.11:
//   lod = log2(max(|ddx * size|, |ddy * size|)), clamped to [0, mipCount - 1]; linear / anisotropic samplers blend the two nearest levels
//   with the fractional part (fp32), point samplers take the nearest level. A single-level texture is a level-0 sample.
// The only caller passes ddx == ddy (GetShadowRayGradients, RaytracingCommon.hlsli:207-240), for which the anisotropic samplers'
// footprint is isotropic as well.
HRT_DEV f4 sample_texture_grad(const GpuTexture& t, uint32_t samplerIndex, f2 uv, f2 ddx, f2 ddy)
{
    const float ax = ddx.x * (float)t.w, ay = ddx.y * (float)t.h, bx = ddy.x * (float)t.w, by = ddy.y * (float)t.h;
    const float rho2 = hrt_max(ax * ax + ay * ay, bx * bx + by * by);
    float lod = rho2 > 0.0f ? 0.5f * hrt_log2(rho2) : 0.0f;
    lod = hrt_clamp(lod, 0.0f, (float)(t.mipCount - 1u));
    const bool point = (samplerIndex == 2u || samplerIndex == 3u);
    const float l0 = point ? hrt_floor(lod + 0.5f) : hrt_floor(lod), f = point ? 0.0f : lod - l0;
    const uint32_t i0 = (uint32_t)l0, i1 = i0 + 1u < t.mipCount ? i0 + 1u : i0;
    f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
    // (a loop over the one or two levels, not two inlined copies of the bilinear fetch)
    const uint32_t n = (f == 0.0f || i1 == i0) ? 1u : 2u;
#pragma nounroll
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t lvl = k ? i1 : i0;
        const int lw = (int)((t.w >> lvl) ? (t.w >> lvl) : 1u), lh = (int)((t.h >> lvl) ? (t.h >> lvl) : 1u);
        const f4 v = sample_texture_level(t.texels, t.format, lw, lh, t.mipOffset[lvl], samplerIndex, uv);
        if (k == 0) acc = v; else acc = lerp4(acc, v, f);
    }
    return acc;
}
HRT_DEV f4 lut_texel(const uint16_t* lut, size_t idx)
{
    uint2 p = reinterpret_cast<const uint2*>(lut)[idx];   // 4 halfs = 8 B, one coalescable load
    f4 r;
    r.x = half_bits_to_float(p.x & 0xffffu); r.y = half_bits_to_float(p.x >> 16);
    r.z = half_bits_to_float(p.y & 0xffffu); r.w = half_bits_to_float(p.y >> 16);
    return r;
}
HRT_DEV int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
HRT_DEV f4 sample_lut2d(const uint16_t* lut, int W, int H, float u, float v)
{
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float ix = hrt_floor(fx), iy = hrt_floor(fy);
    float tx = fx - ix, ty = fy - iy;
    int x0 = clampi((int)ix, W), x1 = clampi((int)ix + 1, W), y0 = clampi((int)iy, H), y1 = clampi((int)iy + 1, H);
    f4 a = lerp4(lut_texel(lut, (size_t)y0 * W + x0), lut_texel(lut, (size_t)y0 * W + x1), tx);
    f4 b = lerp4(lut_texel(lut, (size_t)y1 * W + x0), lut_texel(lut, (size_t)y1 * W + x1), tx);
    return lerp4(a, b, ty);
}
HRT_DEV f4 sample_lut3d(const uint16_t* lut, int W, int H, int D, float u, float v, float w)
{
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f, fz = w * (float)D - 0.5f;
    float ix = hrt_floor(fx), iy = hrt_floor(fy), iz = hrt_floor(fz);
    float tx = fx - ix, ty = fy - iy, tz = fz - iz;
    int x0 = clampi((int)ix, W), x1 = clampi((int)ix + 1, W), y0 = clampi((int)iy, H), y1 = clampi((int)iy + 1, H);
    int z0 = clampi((int)iz, D), z1 = clampi((int)iz + 1, D);
    size_t zo0 = (size_t)z0 * W * H, zo1 = (size_t)z1 * W * H;
    f4 a0 = lerp4(lut_texel(lut, zo0 + (size_t)y0 * W + x0), lut_texel(lut, zo0 + (size_t)y0 * W + x1), tx);
    f4 b0 = lerp4(lut_texel(lut, zo0 + (size_t)y1 * W + x0), lut_texel(lut, zo0 + (size_t)y1 * W + x1), tx);
    f4 s0 = lerp4(a0, b0, ty);
    f4 a1 = lerp4(lut_texel(lut, zo1 + (size_t)y0 * W + x0), lut_texel(lut, zo1 + (size_t)y0 * W + x1), tx);
    f4 b1 = lerp4(lut_texel(lut, zo1 + (size_t)y1 * W + x0), lut_texel(lut, zo1 + (size_t)y1 * W + x1), tx);
    f4 s1 = lerp4(a1, b1, ty);
    return lerp4(s0, s1, tz);
}

// ------------------------------------------------------------------ atmosphere (Atmosphere.hlsli)
namespace atm {
constexpr float kBottom = 6360.0f, kTop = 6420.0f;
constexpr float kSunAngularRadius = 0.004675f;      // 0.00935 / 2.0, Atmosphere.hlsli:42
constexpr float kMuSMin = -0.207912f, kMieG = 0.8f;

HRT_DEV f3 solar_irradiance() { return mk3(1.474000f, 1.850400f, 1.911980f); }
HRT_DEV float safe_sqrt(float a) { return hrt_sqrt(hrt_max(a, 0.0f)); }
HRT_DEV float dist_top(float r, float mu)                                   // :130-134
{
    float disc = r * r * (mu * mu - 1.0f) + kTop * kTop;
    return hrt_max(-r * mu + safe_sqrt(disc), 0.0f);
}
HRT_DEV bool ray_hits_ground(float r, float mu)                              // :157-160
{
    return mu < 0.0f && r * r * (mu * mu - 1.0f) + kBottom * kBottom >= 0.0f;
}
HRT_DEV float texcoord(float x, int size) { return 0.5f / (float)size + x * (1.0f - 1.0f / (float)size); }   // :176-179
HRT_DEV f3 transmittance_to_top(const SceneView& s, float r, float mu)      // :190-198, :207-211
{
    float rho = safe_sqrt(r * r - kBottom * kBottom);
    float d = dist_top(r, mu);
    float H = safe_sqrt(kTop * kTop - kBottom * kBottom);
    float x_mu = texcoord(d / (rho + H), 256);
    float x_r = texcoord(rho / H, 64);
    f4 t = sample_lut2d(s.lutTransmittance, 256, 64, x_mu, x_r);
    return mk3(t.x, t.y, t.z);
}
HRT_DEV f4 scattering_uvwz(float r, float mu, float mu_s, float nu, bool hitsGround)   // :263-297
{
    float H = hrt_sqrt(kTop * kTop - kBottom * kBottom);
    float rho = safe_sqrt(r * r - kBottom * kBottom);
    float u_r = texcoord(rho / H, 32);
    float r_mu = r * mu;
    float disc = r_mu * r_mu - r * r + kBottom * kBottom;
    float u_mu;
    if (hitsGround) {
        float d = -r_mu - safe_sqrt(disc);
        float d_min = r - kBottom, d_max = rho;
        u_mu = 0.5f - 0.5f * texcoord(d_max == d_min ? 0.0f : (d - d_min) / (d_max - d_min), 64);
    } else {
        float d = -r_mu + safe_sqrt(disc + H * H);
        float d_min = kTop - r, d_max = rho + H;
        u_mu = 0.5f + 0.5f * texcoord((d - d_min) / (d_max - d_min), 64);
    }
    float d = dist_top(kBottom, mu_s);
    float d_min = kTop - kBottom, d_max = H;
    float a = (d - d_min) / (d_max - d_min);
    float D = dist_top(kBottom, kMuSMin);
    float A = (D - d_min) / (d_max - d_min);
    float u_mu_s = texcoord(hrt_max(1.0f - a / A, 0.0f) / (1.0f + a), 32);
    f4 o; o.x = (nu + 1.0f) / 2.0f; o.y = u_mu_s; o.z = u_mu; o.w = u_r;
    return o;
}
HRT_DEV f3 combined_scattering(const SceneView& s, float r, float mu, float mu_s, float nu, bool hitsGround, f3& singleMie)  // :305-341
{
    f4 uvwz = scattering_uvwz(r, mu, mu_s, nu, hitsGround);
    float tex_coord_x = uvwz.x * 7.0f;
    float tex_x = hrt_floor(tex_coord_x);
    float lerp_val = tex_coord_x - tex_x;
    float u0 = (tex_x + uvwz.y) / 8.0f, u1 = (tex_x + 1.0f + uvwz.y) / 8.0f;
    f4 s0 = sample_lut3d(s.lutScattering, 256, 128, 32, u0, uvwz.z, uvwz.w);
    f4 s1 = sample_lut3d(s.lutScattering, 256, 128, 32, u1, uvwz.z, uvwz.w);
    f4 cs = lerp4(s0, s1, lerp_val);
    if (cs.x <= 0.0f) singleMie = mk3(0.0f, 0.0f, 0.0f);
    else {
        const f3 ray = mk3(0.005802f, 0.013558f, 0.033100f), mie = mk3(0.003996f, 0.003996f, 0.003996f);
        float k = ray.x / mie.x;
        f3 ratio = mk3(mie.x / ray.x, mie.y / ray.y, mie.z / ray.z);
        f3 t = mk3(cs.x * cs.w / cs.x * k, cs.y * cs.w / cs.x * k, cs.z * cs.w / cs.x * k);
        singleMie = t * ratio;
    }
    return mk3(cs.x, cs.y, cs.z);
}
HRT_DEV float rayleigh_phase(float nu) { float k = 3.0f / (16.0f * HRT_PI); return k * (1.0f + nu * nu); }   // :349-353
HRT_DEV float mie_phase(float g, float nu)                                                                   // :355-359
{
    float k = 3.0f / (8.0f * HRT_PI) * (1.0f - g * g) / (2.0f + g * g);
    float b = hrt_max(1.0f + g * g - 2.0f * g * nu, 0.0001f);
    return k * (1.0f + nu * nu) / (b * hrt_sqrt(b));
}
HRT_DEV float smoothstep(float a, float b, float x) { float t = hrt_saturate((x - a) / (b - a)); return (t * t) * (3.0f - 2.0f * t); }
HRT_DEV f3 transmittance_to_sun(const SceneView& s, float r, float mu_s)    // :414-420
{
    float sin_theta_h = kBottom / r;
    float cos_theta_h = -hrt_sqrt(hrt_max(1.0f - sin_theta_h * sin_theta_h, 0.0f));
    float e = sin_theta_h * kSunAngularRadius;
    float f = smoothstep(-e, e, mu_s - cos_theta_h);
    return transmittance_to_top(s, r, mu_s) * f;
}
HRT_DEV f3 atmosphere_pos(f3 worldPos) { return (worldPos - mk3(0.0f, -6360000.0f, 0.0f)) / 1000.0f; }   // :564-567
HRT_DEV f3 sun_radiance(const SceneView& s, f3 p_atmo, f3 sunDir, float sunIntensity)   // :569-574
{
    float r = length(p_atmo);
    float mu_s = dot(p_atmo, sunDir) / r;
    return (solar_irradiance() * transmittance_to_sun(s, r, mu_s)) * sunIntensity;
}
// GetAtmosphereSkyRadiance :583-601 (GetSkyRadiance :458-500 with shadow_length 0)
HRT_DEV f3 sky_radiance(const SceneView& s, f3 cameraPos, f3 viewRay, f3 sunDir, float sunIntensity, bool addSunDisk)
{
    f3 camera = atmosphere_pos(cameraPos);
    float r = length(camera);
    float rmu = dot(camera, viewRay);
    float dtop = -rmu - safe_sqrt(rmu * rmu - r * r + kTop * kTop);
    f3 transmittance, sky;
    bool outside = false;
    if (dtop > 0.0f) { camera = camera + viewRay * dtop; r = kTop; rmu += dtop; }
    else if (r > kTop) outside = true;
    if (outside) { transmittance = mk3(1.0f, 1.0f, 1.0f); sky = mk3(0.0f, 0.0f, 0.0f); }
    else {
        float mu = rmu / r;
        float mu_s = dot(camera, sunDir) / r;
        float nu = dot(viewRay, sunDir);
        bool hitsGround = ray_hits_ground(r, mu);
        transmittance = hitsGround ? mk3(0.0f, 0.0f, 0.0f) : transmittance_to_top(s, r, mu);
        f3 mie;
        f3 scat = combined_scattering(s, r, mu, mu_s, nu, hitsGround, mie);
        sky = scat * rayleigh_phase(nu) + mie * mie_phase(kMieG, nu);
    }
    if (addSunDisk) {
        float nu = dot(viewRay, sunDir);
        if (nu > hrt_cos(kSunAngularRadius)) {
            float den = HRT_PI * kSunAngularRadius * kSunAngularRadius;
            f3 si = solar_irradiance();
            sky = sky + mk3(si.x / den, si.y / den, si.z / den) * transmittance;
        }
    }
    return sky * sunIntensity;
}
} // namespace atm

// ------------------------------------------------------------------ BRDF (CommonLighting.hlsli)
struct Lighting {            // the part of LightingInputs the path tracer reads
    f3 N, V, L, baseColor; float roughness, metallic, ior;
    f3 F0, F; float kD;
    float NdotV, NdotL, NdotH, VdotH, LdotV, LdotH;
};
HRT_DEV f3 f_schlick(f3 spec, float VdotH)                                   // :123-129
{
    float Fc = hrt_pow5(1.0f - VdotH);
    float s = hrt_saturate(50.0f * spec.y) * Fc;
    float k = 1.0f - Fc;
    return mk3(s + k * spec.x, s + k * spec.y, s + k * spec.z);
}
HRT_DEV f3 compute_f0(f3 baseColor, float metallic, float ior)               // :64-68
{
    float q = (ior - 1.0f) / (ior + 1.0f);
    float d = q * q;
    return mk3(lerp(d, baseColor.x, metallic), lerp(d, baseColor.y, metallic), lerp(d, baseColor.z, metallic));
}
HRT_DEV void prepare_byproducts(Lighting& in)                                // :316-334
{
    in.NdotV = hrt_saturate(dot(in.N, in.V));
    in.NdotL = hrt_saturate(dot(in.N, in.L));
    f3 VpL = in.V + in.L;
    float len = dot(VpL, VpL);
    f3 H = (len > 1e-8f) ? VpL * hrt_rsqrt(len) : in.N;
    in.NdotH = hrt_saturate(dot(in.N, H));
    in.VdotH = hrt_saturate(dot(in.V, H));
    in.LdotV = hrt_saturate(dot(in.L, in.V));
    in.LdotH = hrt_saturate(dot(in.L, H));
    in.F0 = compute_f0(in.baseColor, in.metallic, in.ior);
    in.kD = 1.0f - in.metallic;
    in.F = f_schlick(in.F0, in.VdotH);
}
HRT_DEV float d_ggx(float NdotH, float roughness)                            // :80-86
{
    float alpha = roughness * roughness, alpha2 = alpha * alpha;
    float denom = NdotH * NdotH * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (HRT_PI * denom * denom);
}
HRT_DEV float burley_diffuse(float NdotL, float NdotV, float LdotH, float rough)   // :148-168
{
    if (NdotL <= 0.0f || NdotV <= 0.0f) return 0.0f;
    float rough2 = rough * rough;
    float FL = hrt_pow5(1.0f - NdotL), FV = hrt_pow5(1.0f - NdotV);
    float Fd90 = 0.5f + 2.0f * rough2 * LdotH * LdotH;
    float Fd = lerp(1.0f, Fd90, FL) * lerp(1.0f, Fd90, FV);
    return Fd * NdotL / HRT_PI;
}
// EvaluateDirectLight :360-375 with ComputeSpecularBRDF :345-358, before the shadow factor:
// diffuse = (diffuseTerm*kD*baseColor)*radiance, specular = (spec*NdotL)*radiance.
HRT_DEV void evaluate_direct_unshadowed(const Lighting& in, f3 radiance, f3& diffuse, f3& specular)
{
    float dt = burley_diffuse(in.NdotL, in.NdotV, in.LdotH, in.roughness);
    f3 kd = mk3(in.kD, in.kD, in.kD);
    f3 dif = (kd * dt) * in.baseColor;
    float alpha = in.roughness * in.roughness, alpha2 = alpha * alpha;
    float D = d_ggx(in.NdotH, in.roughness);
    float g1 = in.NdotV * hrt_sqrt(alpha2 + (1.0f - alpha2) * in.NdotL * in.NdotL);
    float g2 = in.NdotL * hrt_sqrt(alpha2 + (1.0f - alpha2) * in.NdotV * in.NdotV);
    float G2 = 0.5f / hrt_max(g1 + g2, 1e-6f);
    float k = D * G2;
    f3 spec = in.F * k;
    diffuse = dif * radiance;
    specular = (spec * in.NdotL) * radiance;
}
HRT_DEV void tangent_frame(f3 N, f3& T, f3& B)                                // :610-615, :176-178, :702-704
{
    f3 up = hrt_abs(N.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    T = normalize(cross(up, N));
    B = cross(N, T);
}
HRT_DEV f3 frame_combine(f3 T, f3 N, f3 B, f3 l) { return (T * l.x + N * l.y) + B * l.z; }
// The two BRDF lobes share their first steps -- two random numbers, sin/cos of 2*pi*u, one square root, the tangent frame of N --
// so the callers evaluate those once (shade_surface_b) and the lobes start from them: `root` = sqrt(uy) (cosine) or sqrt(ux) (VNDF),
// (sp, cp) = sincos(2*pi*ux) (cosine) or sincos(2*pi*uy) (VNDF). Same operations on the same values as the reference order.
HRT_DEV f3 sample_hemisphere_cosine_from(float root, float sp, float cp, f3 T, f3 B, f3 normal)   // :170-183
{
    float cosTheta = root;
    float sinTheta = hrt_sqrt(hrt_max(0.0f, 1.0f - cosTheta * cosTheta));
    return frame_combine(T, normal, B, mk3(sinTheta * cp, cosTheta, sinTheta * sp));
}
HRT_DEV f3 sample_hemisphere_cosine(float ux, float uy, f3 normal)
{
    float sp, cp; hrt_sincos(2.0f * HRT_PI * ux, &sp, &cp);
    f3 T, B; tangent_frame(normal, T, B);
    return sample_hemisphere_cosine_from(hrt_sqrt(uy), sp, cp, T, B, normal);
}
HRT_DEV f3 sample_ggx_vndf_from(float root, float sp, float cp, f3 T, f3 B, f3 N, f3 V, float roughness)   // :622-655
{
    float alpha = roughness * roughness;
    f3 Vl = mk3(dot(V, T), dot(V, N), dot(V, B));
    f3 Vh = normalize(mk3(alpha * Vl.x, Vl.y, alpha * Vl.z));
    float lensq = Vh.x * Vh.x + Vh.z * Vh.z;
    f3 T1 = lensq > 0.0f ? mk3(-Vh.z, 0.0f, Vh.x) / hrt_sqrt(lensq) : mk3(1.0f, 0.0f, 0.0f);
    f3 T2 = cross(Vh, T1);
    float r = root;
    float t1 = r * cp, t2 = r * sp;
    float s = 0.5f * (1.0f + Vh.y);
    t2 = lerp(hrt_sqrt(hrt_max(0.0f, 1.0f - t1 * t1)), t2, s);
    float nz = hrt_sqrt(hrt_max(0.0f, (1.0f - t1 * t1) - t2 * t2));
    f3 Nh = (T1 * t1 + T2 * t2) + Vh * nz;
    f3 Ne = normalize(mk3(alpha * Nh.x, hrt_max(0.0f, Nh.y), alpha * Nh.z));
    return frame_combine(T, N, B, Ne);
}
HRT_DEV f3 sample_ggx_vndf(float ux, float uy, f3 N, f3 V, float roughness)
{
    float sp, cp; hrt_sincos(2.0f * HRT_PI * uy, &sp, &cp);
    f3 T, B; tangent_frame(N, T, B);
    return sample_ggx_vndf_from(hrt_sqrt(ux), sp, cp, T, B, N, V, roughness);
}
HRT_DEV f3 eval_ggx_vndf_weight(f3 F0, f3 N, f3 V, f3 L, f3 H, float roughness)   // :671-688
{
    float alpha = roughness * roughness, alpha2 = alpha * alpha;
    float NdotV = hrt_saturate(dot(N, V)), NdotL = hrt_saturate(dot(N, L)), VdotH = hrt_saturate(dot(V, H));
    if (NdotV <= 0.0f || NdotL <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    f3 F = f_schlick(F0, VdotH);
    float G1L = 2.0f * NdotL / (NdotL + hrt_sqrt(alpha2 + (1.0f - alpha2) * NdotL * NdotL));
    return F * G1L;
}
HRT_DEV f3 sample_cone(f3 dir, float cosHalf, float ux, float uy)             // :693-708
{
    float cosTheta = 1.0f - ux * (1.0f - cosHalf);
    float sinTheta = hrt_sqrt(hrt_max(0.0f, 1.0f - cosTheta * cosTheta));
    float phi = 2.0f * HRT_PI * uy;
    float sp, cp; hrt_sincos(phi, &sp, &cp);
    f3 T, B; tangent_frame(dir, T, B);
    return frame_combine(T, dir, B, mk3(sinTheta * cp, cosTheta, sinTheta * sp));
}
// EvalFresnelDielectric, PathTracer.hlsl:26-43
HRT_DEV float fresnel_dielectric(float eta, float cosThetaI, float& cosThetaT)
{
    if (cosThetaI < 0.0f) { eta = 1.0f / eta; cosThetaI = -cosThetaI; }
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq >= 1.0f) { cosThetaT = 0.0f; return 1.0f; }
    cosThetaT = hrt_sqrt(hrt_max(0.0f, 1.0f - sinThetaTSq));
    float Rs = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    float Rp = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    return 0.5f * (Rs * Rs + Rp * Rp);
}

// candidate alpha = mat.m_BaseColor.w [* albedo.a] (RaytracingCommon.hlsli:97-103, :172-176)
HRT_DEV float candidate_alpha(const SceneView& s, const HrptMaterialConstants& mat, f2 uv)
{
    float alpha = mat.m_BaseColor[3];
    if (mat.m_TextureFlags & HRPT_TEXFLAG_ALBEDO) alpha *= sample_texture(s, mat.m_AlbedoTextureIndex, mat.m_AlbedoSamplerIndex, uv).w;
    return alpha;
}

// AlphaTestGrad's alpha (RaytracingCommon.hlsli:112-130) with the synthetic gradients of GetShadowRayGradients (:207-240): ddx = ddy =
// uvRange * triangleArea / max(dist, 0.1), from the world-space triangle, the hit's barycentrics and the ray origin. Only textures with a
// mip chain need them (a single-level texture is sampled at level 0 whatever the gradients are).
HRT_DEV float candidate_alpha_grad(const SceneView& s, const HrptMaterialConstants& mat, f2 uv, const TriVerts& tv, uint32_t tri, float bu, float bv, f3 rayOrigin)
{
    float alpha = mat.m_BaseColor[3];
    if (!(mat.m_TextureFlags & HRPT_TEXFLAG_ALBEDO)) return alpha;
    const uint32_t ti = mat.m_AlbedoTextureIndex;
    if (ti >= s.textureCount) return alpha * 0.0f;
    const GpuTexture& t = s.textures[ti];
    const uint8_t* texels = t.texels;
    if (!texels) return alpha * 0.0f;
    if (t.mipCount <= 1u) return alpha * sample_texture_level(texels, t.format, (int)t.w, (int)t.h, 0u, mat.m_AlbedoSamplerIndex, uv).w;
    const GpuTri& g = s.tris[tri];
    const f3 p0 = mk3(g.p0), p1 = mk3(g.p1), p2 = mk3(g.p2);
    const float w0 = (1.0f - bu) - bv;
    const f3 hitPos = (p0 * w0 + p1 * bu) + p2 * bv;
    const float dist = length(hitPos - rayOrigin);
    const float triangleArea = length(cross(p1 - p0, p2 - p0)) * 0.5f;
    f2 uvRange;
    uvRange.x = hrt_max(tv.uv0.x, hrt_max(tv.uv1.x, tv.uv2.x)) - hrt_min(tv.uv0.x, hrt_min(tv.uv1.x, tv.uv2.x));
    uvRange.y = hrt_max(tv.uv0.y, hrt_max(tv.uv1.y, tv.uv2.y)) - hrt_min(tv.uv0.y, hrt_min(tv.uv1.y, tv.uv2.y));
    const float gradientScale = triangleArea / hrt_max(dist, 0.1f);
    f2 grad; grad.x = uvRange.x * gradientScale; grad.y = uvRange.y * gradientScale;
    return alpha * sample_texture_grad(t, mat.m_AlbedoSamplerIndex, uv, grad, grad).w;
}

struct SurfaceAttr { f3 worldPos, worldNormal, worldTangent; float tangentSign; f2 uv; };
struct Pbr { f3 baseColor; float alpha, roughness, metallic; f3 emissive, normal; };

// GetFullHitAttributes, RaytracingCommon.hlsli:52-77 (LOD 0: PathTracer.hlsl:103)
HRT_DEV SurfaceAttr full_hit_attributes(const SceneView& s, const Hit& hit, const Ray& ray, const TriVerts& tv, const GpuInstShade& is, bool wantTangent)
{
    float bx = (1.0f - hit.u) - hit.v, by = hit.u, bz = hit.v;
    SurfaceAttr a;
    a.worldPos = ray.o + ray.d * hit.t;
    f3 ln = (tv.n0 * bx + tv.n1 * by) + tv.n2 * bz;
    a.worldNormal = transform_normal(ln, is);
    a.tangentSign = 1.0f;
    if (wantTangent && s.tangents) {
        GpuTriTangent tg = s.tangents[hit.tri];
        f3 t0 = mk3(tg.t0.x, tg.t0.y, tg.t0.z), t1 = mk3(tg.t1.x, tg.t1.y, tg.t1.z), t2 = mk3(tg.t2.x, tg.t2.y, tg.t2.z);
        f3 lt = (t0 * bx + t1 * by) + t2 * bz;
        a.worldTangent = transform_normal(lt, is);
        a.tangentSign = (tg.t0.w * bx + tg.t1.w * by) + tg.t2.w * bz;
    } else a.worldTangent = mk3(0.0f, 0.0f, 0.0f);
    a.uv.x = (tv.uv0.x * bx + tv.uv1.x * by) + tv.uv2.x * bz;
    a.uv.y = (tv.uv0.y * bx + tv.uv1.y * by) + tv.uv2.y * bz;
    return a;
}
// TransformNormalWithTBN, Common.hlsli:183-200
HRT_DEV f3 normal_with_tbn(float nx, float ny, f3 normal, f3 tangent, float tangentSign)
{
    float x = 2.0f * nx - 1.0f, y = 2.0f * ny - 1.0f;
    float z = hrt_sqrt(hrt_saturate(1.0f - (x * x + y * y)));
    f3 n_w = normalize(normal);
    f3 t_w = normalize(tangent);
    t_w = normalize(t_w - n_w * dot(t_w, n_w));
    f3 b_w = normalize(cross(n_w, t_w) * tangentSign);
    f3 o = mk3((x * t_w.x + y * b_w.x) + z * n_w.x, (x * t_w.y + y * b_w.y) + z * n_w.y, (x * t_w.z + y * b_w.z) + z * n_w.z);
    return normalize(o);
}
// GetPBRAttributes, RaytracingCommon.hlsli:252-296
// The four material textures of a hit fetched TOGETHER (GetPBRAttributes, RaytracingCommon.hlsli:252-296). Sampled one after the other, every texture
// is a chain of dependent round trips -- descriptor, then the four texels of its bilinear footprint, then (sRGB formats) the linearisation table --
// and a hit with three maps waits for nine of them in a row; wf_shade on textured scenes spends 69 % of its wave cycles waiting
// (profiles/r02_pmc_config4.txt). Here all descriptors are requested first, then all sixteen texels (unused slots read texel 0 of a valid
// texture: no divergent branch between the requests), and only then the filtering arithmetic runs, texture by texture, exactly as
// sample_texture_level does it: same operations, same order, same bits. Only for 8-bit formats (RGBA8_UNORM / RGBA8_SRGB, what stb-decoded and
// block-decoded images arrive as) at level 0; anything else takes the one-by-one path below.
#ifndef HRPT_NO_BATCHED_TEXTURES
struct TexSlot { const uint32_t* texels; uint32_t format, sampler; int w, h; bool on; };
HRT_DEV bool pbr_textures_batched(const SceneView& s, f2 uv, const HrptMaterialConstants& m, uint32_t texFlags, f4 (&out)[4])
{
    if (s.textureCount == 0u) return false;
    const uint32_t index[4] = { m.m_AlbedoTextureIndex, m.m_RoughnessMetallicTextureIndex, m.m_EmissiveTextureIndex, m.m_NormalTextureIndex };
    const uint32_t sampler[4] = { m.m_AlbedoSamplerIndex, m.m_RoughnessSamplerIndex, m.m_EmissiveSamplerIndex, m.m_NormalSamplerIndex };
    const uint32_t bit[4] = { HRPT_TEXFLAG_ALBEDO, HRPT_TEXFLAG_ROUGHNESS_METALLIC, HRPT_TEXFLAG_EMISSIVE, HRPT_TEXFLAG_NORMAL };
    TexSlot t[4];
    bool simple = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {             // round 1: the descriptors
        const bool want = (texFlags & bit[k]) != 0u && index[k] < s.textureCount;
        const GpuTexture& g = s.textures[want ? index[k] : 0u];
        t[k].texels = reinterpret_cast<const uint32_t*>(g.texels); t[k].format = g.format; t[k].w = (int)g.w; t[k].h = (int)g.h; t[k].sampler = sampler[k];
        t[k].on = want && g.texels != nullptr;
        // (a flagged texture whose index is out of range or whose texels are missing samples as zero, as sample_texture does)
        simple = simple && (!t[k].on || g.format == HRPT_TEXTURE_FORMAT_RGBA8_UNORM || g.format == HRPT_TEXTURE_FORMAT_RGBA8_SRGB);
    }
    if (!simple) return false;
    uint32_t word[4][4]; float tx[4], ty[4]; bool point[4];
    const uint32_t* safe = reinterpret_cast<const uint32_t*>(s.textures);      // readable, whatever it holds: the slot's result is discarded
#pragma unroll
    for (int k = 0; k < 4; ++k) {             // round 2: the footprints (sample_texture_level's address arithmetic) and the sixteen texel requests
        const bool wrap = (t[k].sampler <= 5u) ? ((t[k].sampler & 1u) != 0) : false;
        point[k] = (t[k].sampler == 2u || t[k].sampler == 3u);
        const int lw = t[k].on ? t[k].w : 1, lh = t[k].on ? t[k].h : 1;
        float fx = uv.x * (float)lw, fy = uv.y * (float)lh;
        if (!point[k]) { fx = fx - 0.5f; fy = fy - 0.5f; }
        const float ix = hrt_floor(fx), iy = hrt_floor(fy);
        tx[k] = point[k] ? 0.0f : fx - ix; ty[k] = point[k] ? 0.0f : fy - iy;
        const int x0 = wrap_i((int)ix, lw, wrap), x1 = wrap_i((int)ix + 1, lw, wrap), y0 = wrap_i((int)iy, lh, wrap), y1 = wrap_i((int)iy + 1, lh, wrap);
        const uint32_t* base = t[k].on ? t[k].texels : safe;
        const uint32_t r0 = (uint32_t)y0 * (uint32_t)lw, r1 = (uint32_t)y1 * (uint32_t)lw;
        word[k][0] = base[t[k].on ? r0 + (uint32_t)x0 : 0u]; word[k][1] = base[t[k].on ? r0 + (uint32_t)x1 : 0u];
        word[k][2] = base[t[k].on ? r1 + (uint32_t)x0 : 0u]; word[k][3] = base[t[k].on ? r1 + (uint32_t)x1 : 0u];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {             // round 3: decode + filter, texture by texture
        f4 c[4];
        const bool srgb = t[k].format == HRPT_TEXTURE_FORMAT_RGBA8_SRGB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = word[k][j];
            if (srgb) { c[j].x = kSrgbToLinear[p & 255u]; c[j].y = kSrgbToLinear[(p >> 8) & 255u]; c[j].z = kSrgbToLinear[(p >> 16) & 255u]; }
            else { c[j].x = unorm8_to_float(p & 255u); c[j].y = unorm8_to_float((p >> 8) & 255u); c[j].z = unorm8_to_float((p >> 16) & 255u); }
            c[j].w = unorm8_to_float(p >> 24);
        }
        f4 r;
        if (point[k]) r = c[0];
        else { const f4 a = lerp4(c[0], c[1], tx[k]), b = lerp4(c[2], c[3], tx[k]); r = lerp4(a, b, ty[k]); }
        if (!t[k].on) r.x = r.y = r.z = r.w = 0.0f;
        out[k] = r;
    }
    return true;
}
#endif
HRT_DEV Pbr pbr_attributes(const SceneView& s, const SurfaceAttr& a, const HrptMaterialConstants& m, uint32_t texFlags)
{
    Pbr p;
#ifndef HRPT_NO_BATCHED_TEXTURES
    f4 tex[4];
    if (texFlags != 0u && pbr_textures_batched(s, a.uv, m, texFlags, tex)) {
        HRT_PHASE(PH_SHADE_TEX);
        p.baseColor = mk3(m.m_BaseColor); p.alpha = m.m_BaseColor[3];
        if (texFlags & HRPT_TEXFLAG_ALBEDO) { p.baseColor = p.baseColor * mk3(tex[0].x, tex[0].y, tex[0].z); p.alpha *= tex[0].w; }
        p.roughness = m.m_RoughnessMetallic[0]; p.metallic = m.m_RoughnessMetallic[1];
        if (texFlags & HRPT_TEXFLAG_ROUGHNESS_METALLIC) { p.roughness = tex[1].y; p.metallic = tex[1].z; }
        p.roughness = hrt_max(p.roughness, 0.04f);
        p.emissive = mk3(m.m_EmissiveFactor);
        if (texFlags & HRPT_TEXFLAG_EMISSIVE) p.emissive = p.emissive * mk3(tex[2].x, tex[2].y, tex[2].z);
        if (texFlags & HRPT_TEXFLAG_NORMAL) p.normal = normal_with_tbn(tex[3].x, tex[3].y, a.worldNormal, a.worldTangent, a.tangentSign);
        else p.normal = normalize(a.worldNormal);
        return p;
    }
#endif
    p.baseColor = mk3(m.m_BaseColor); p.alpha = m.m_BaseColor[3];
    if (texFlags & HRPT_TEXFLAG_ALBEDO) {
        HRT_PHASE(PH_SHADE_TEX);
        f4 t = sample_texture(s, m.m_AlbedoTextureIndex, m.m_AlbedoSamplerIndex, a.uv);
        p.baseColor = p.baseColor * mk3(t.x, t.y, t.z); p.alpha *= t.w;
    }
    p.roughness = m.m_RoughnessMetallic[0];
    p.metallic = m.m_RoughnessMetallic[1];
    if (texFlags & HRPT_TEXFLAG_ROUGHNESS_METALLIC) {
        f4 t = sample_texture(s, m.m_RoughnessMetallicTextureIndex, m.m_RoughnessSamplerIndex, a.uv);
        p.roughness = t.y; p.metallic = t.z;
    }
    p.roughness = hrt_max(p.roughness, 0.04f);
    p.emissive = mk3(m.m_EmissiveFactor);
    if (texFlags & HRPT_TEXFLAG_EMISSIVE) {
        f4 t = sample_texture(s, m.m_EmissiveTextureIndex, m.m_EmissiveSamplerIndex, a.uv);
        p.emissive = p.emissive * mk3(t.x, t.y, t.z);
    }
    if (texFlags & HRPT_TEXFLAG_NORMAL) {
        f4 t = sample_texture(s, m.m_NormalTextureIndex, m.m_NormalSamplerIndex, a.uv);
        p.normal = normal_with_tbn(t.x, t.y, a.worldNormal, a.worldTangent, a.tangentSign);
    } else p.normal = normalize(a.worldNormal);
    return p;
}

// Light distance attenuation, CommonLighting.hlsli:766-769 / :835-837 (pow(x,4), pow(x,2) by multiplication)
HRT_DEV float distance_attenuation(const HrptGPULight& l, float distSq, float dist)
{
    float a = 1.0f / (distSq + 1.0f);
    if (l.m_Range > 0.0f) {
        float q = dist / l.m_Range; float q2 = q * q; float q4 = q2 * q2;
        float sa = hrt_saturate(1.0f - q4);
        a *= sa * sa;
    }
    return a;
}

} // namespace hrt
