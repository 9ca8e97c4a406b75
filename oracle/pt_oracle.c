/*
 * oracle/pt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT. See pt_oracle.h.
 *
 * Plain-C restatement of the reference path tracer. Every function cites the
 * reference lines it follows (paths relative to /root/reference/src/shaders unless
 * they start with src/). PARITY UNPINNED by the reference (no goldens exist there).
 *
 * Two things are NOT in the reference source and are defined here:
 *  (1) ray/triangle intersection: Woop-Benthin-Wald watertight test in fp32 on
 *      world-space triangles (instance transform applied once at load, in the
 *      row-vector convention of Common.hlsli:18-21);
 *  (2) the closest hit is the minimum of (t, instanceIndex, primitiveIndex) in
 *      lexicographic order over all triangles with tmin < t < tmax, and non-opaque
 *      candidates (instances flagged ForceNonOpaque, src/Scene.cpp:150-154) are
 *      visited FRONT TO BACK in that same order. DXR leaves candidate order
 *      undefined, so this is one legal order; it makes the result independent of
 *      the acceleration structure (brute force == BVH, bit for bit).
 *
 * Expression order conventions (needed for bit reproducibility, no FMA):
 *   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z ; normalize(v) = v * (1/sqrt(dot(v,v)))
 *   lerp(a,b,t) = a + t*(b-a) ; mul(v,M)_j = ((v.x*M0j + v.y*M1j) + v.z*M2j) + v.w*M3j
 *   a*s1 + b*s2 + c*s3 is evaluated left to right.
 */
#define _GNU_SOURCE
#include "pt_oracle.h"
#include "../include/hobbyrt/detmath.h"
#include "../include/hobbyrt/srgb_table.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { float x, y; } v2;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float length3(v3 a) { return hrt_sqrt(dot3(a, a)); }
static inline v3 normalize3(v3 a) { float inv = 1.0f / hrt_sqrt(dot3(a, a)); return scale3(a, inv); }
static inline float lerp1(float a, float b, float t) { return a + t * (b - a); }
static inline float max3c(v3 a) { return hrt_max(a.x, hrt_max(a.y, a.z)); }
/* HLSL reflect(i,n) = i - 2*n*dot(i,n) */
static inline v3 reflect3(v3 i, v3 n) { float s = 2.0f * dot3(i, n); return sub3(i, scale3(n, s)); }
/* HLSL refract: zero vector on total internal reflection (relied on at PathTracer.hlsl:180) */
static inline v3 refract3(v3 i, v3 n, float eta)
{
    float d = dot3(n, i);
    float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return V3(0, 0, 0);
    float s = eta * d + hrt_sqrt(k);
    return sub3(scale3(i, eta), scale3(n, s));
}
/* mul(float4(p,w), M) with row-major M[16] -- Common.hlsli:18-21 */
static inline v4 mul_v4_m(float x, float y, float z, float w, const float* M)
{
    v4 r;
    r.x = ((x * M[0] + y * M[4]) + z * M[8]) + w * M[12];
    r.y = ((x * M[1] + y * M[5]) + z * M[9]) + w * M[13];
    r.z = ((x * M[2] + y * M[6]) + z * M[10]) + w * M[14];
    r.w = ((x * M[3] + y * M[7]) + z * M[11]) + w * M[15];
    return r;
}
static inline v3 transform_point(v3 p, const float* M)
{
    /* mul(float4(p,1), M).xyz ; 1*M3j == M3j exactly */
    return V3(((p.x * M[0] + p.y * M[4]) + p.z * M[8]) + M[12],
              ((p.x * M[1] + p.y * M[5]) + p.z * M[9]) + M[13],
              ((p.x * M[2] + p.y * M[6]) + p.z * M[10]) + M[14]);
}
/* TransformNormal, Common.hlsli:33-47: normalize(mul(n, adjugate(world3x3))) */
static inline v3 transform_normal(v3 n, const float* M)
{
    v3 r0 = V3(M[0], M[1], M[2]), r1 = V3(M[4], M[5], M[6]), r2 = V3(M[8], M[9], M[10]);
    v3 a0 = cross3(r1, r2), a1 = cross3(r2, r0), a2 = cross3(r0, r1);
    v3 o = V3((n.x * a0.x + n.y * a1.x) + n.z * a2.x,
              (n.x * a0.y + n.y * a1.y) + n.z * a2.y,
              (n.x * a0.z + n.y * a1.z) + n.z * a2.z);
    return normalize3(o);
}

/* ------------------------------------------------------------------ half conversion */
/* DirectX::PackedVector::XMConvertFloatToHalf (round to nearest even), src/CommonResources.cpp:553 */
uint16_t or_float_to_half(float f)
{
    uint32_t x = hrt_f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) /* >= 65536, inf or NaN */
        return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? (0x200u | ((x >> 13) & 0x3ffu)) : 0u));
    if (x < 0x38800000u) { /* subnormal half or zero */
        if (x < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (ties at exactly 2^-25 round to even = 0) */
        uint32_t e = x >> 23;
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - e; /* 14..24 */
        uint32_t half = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t r = x + 0xfffu + ((x >> 13) & 1u);
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
float or_half_to_float(uint16_t h) { return hrt_f16tof32(h); }

/* ------------------------------------------------------------------ context */
typedef struct { v3 p0, p1, p2; uint32_t inst, prim; uint32_t opaque; } WTri;
typedef struct { float bmin[3], bmax[3]; int32_t left, right; uint32_t first, count; } BNode;
typedef struct { uint8_t* texels; uint32_t w, h, format, mipCount; uint32_t mipOffset[HRPT_TEXTURE_MAX_MIPS]; } OTex;   /* HrptTextureDesc: all levels, decoded texels */

struct OrContext {
    HrptVertexQuantized* vertices; uint32_t vertexCount;
    uint32_t* indices; uint32_t indexCount;
    HrptMeshData* meshData; uint32_t meshDataCount;
    HrptPerInstanceData* instances; uint32_t instanceCount;
    HrptMaterialConstants* materials; uint32_t materialCount;
    HrptGPULight* lights; uint32_t lightCount;
    OTex* textures; uint32_t textureCount;
    uint16_t* lutTransmittance;   /* 256x64 RGBA16F */
    uint16_t* lutScattering;      /* 256x128x32 RGBA16F */
    WTri* tris; uint32_t triCount;
    uint32_t* triOrder;           /* BVH leaf order -> tris index */
    BNode* nodes; uint32_t nodeCount;
};

static __thread char g_err[256];
static char g_err_shared[256];
const char* or_last_error(void) { return g_err_shared; }
static void set_err(const char* m) { snprintf(g_err, sizeof g_err, "%s", m); snprintf(g_err_shared, sizeof g_err_shared, "%s", m); }

static void* dup_mem(const void* p, size_t n) { void* r = malloc(n ? n : 1); if (r && n) memcpy(r, p, n); return r; }

/* ---- BVH2 build (oracle-owned; median split on largest centroid extent) ---- */
typedef struct { OrContext* c; float* cent; } BuildCtx;
static int g_axis; static float* g_cent;
static int cmp_axis(const void* a, const void* b)
{
    uint32_t ia = *(const uint32_t*)a, ib = *(const uint32_t*)b;
    float ca = g_cent[ia * 3 + g_axis], cb = g_cent[ib * 3 + g_axis];
    if (ca < cb) return -1; if (ca > cb) return 1;
    return (ia < ib) ? -1 : (ia > ib);
}
static void tri_bounds(const WTri* t, float* mn, float* mx)
{
    const v3* p = &t->p0;
    for (int k = 0; k < 3; k++) {
        float a = (&p[0].x)[k], b = (&p[1].x)[k], c = (&p[2].x)[k];
        mn[k] = hrt_min(a, hrt_min(b, c)); mx[k] = hrt_max(a, hrt_max(b, c));
    }
}
static int32_t build_node(OrContext* c, float* cent, uint32_t first, uint32_t count)
{
    int32_t id = (int32_t)c->nodeCount++;
    BNode* n = &c->nodes[id];
    float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
    float cmn[3] = { 1e30f, 1e30f, 1e30f }, cmx[3] = { -1e30f, -1e30f, -1e30f };
    for (uint32_t i = first; i < first + count; i++) {
        float a[3], b[3]; tri_bounds(&c->tris[c->triOrder[i]], a, b);
        for (int k = 0; k < 3; k++) {
            mn[k] = hrt_min(mn[k], a[k]); mx[k] = hrt_max(mx[k], b[k]);
            float ce = cent[c->triOrder[i] * 3 + k];
            cmn[k] = hrt_min(cmn[k], ce); cmx[k] = hrt_max(cmx[k], ce);
        }
    }
    /* conservative padding: the fp32 triangle test may accept points a few ulp outside -- ulps of the coordinates and, for a hit next to the
       ray origin, of the triangle's size (the shear works on vertex - origin): 1e-6 of the node's largest side covers every triangle inside */
    float ext = hrt_max(mx[0] - mn[0], hrt_max(mx[1] - mn[1], mx[2] - mn[2]));
    for (int k = 0; k < 3; k++) {
        float m = hrt_max(hrt_abs(mn[k]), hrt_abs(mx[k]));
        float pad = 1e-5f * m + 1e-6f + 1e-6f * ext;
        n->bmin[k] = mn[k] - pad; n->bmax[k] = mx[k] + pad;
    }
    n->left = n->right = -1; n->first = first; n->count = count;
    if (count <= 2) return id;
    /* full-sweep SAH over the three axes (oracle-owned builder; quality matters only for the n/t counters) */
    int bestAxis = -1; uint32_t bestSplit = count / 2; float bestCost = 1e30f;
    float* rarea = (float*)malloc(count * sizeof(float));
    for (int axis = 0; axis < 3; axis++) {
        if (!(cmx[axis] - cmn[axis] > 0.0f)) continue;
        g_axis = axis; g_cent = cent;
        qsort(c->triOrder + first, count, sizeof(uint32_t), cmp_axis);
        float bmn[3] = { 1e30f, 1e30f, 1e30f }, bmx[3] = { -1e30f, -1e30f, -1e30f };
        for (uint32_t i = count; i-- > 1;) {
            float a[3], b[3]; tri_bounds(&c->tris[c->triOrder[first + i]], a, b);
            for (int k = 0; k < 3; k++) { bmn[k] = hrt_min(bmn[k], a[k]); bmx[k] = hrt_max(bmx[k], b[k]); }
            float dx = bmx[0] - bmn[0], dy = bmx[1] - bmn[1], dz = bmx[2] - bmn[2];
            rarea[i] = dx * dy + dy * dz + dz * dx;
        }
        bmn[0] = bmn[1] = bmn[2] = 1e30f; bmx[0] = bmx[1] = bmx[2] = -1e30f;
        for (uint32_t i = 0; i + 1 < count; i++) {
            float a[3], b[3]; tri_bounds(&c->tris[c->triOrder[first + i]], a, b);
            for (int k = 0; k < 3; k++) { bmn[k] = hrt_min(bmn[k], a[k]); bmx[k] = hrt_max(bmx[k], b[k]); }
            float dx = bmx[0] - bmn[0], dy = bmx[1] - bmn[1], dz = bmx[2] - bmn[2];
            float cost = (dx * dy + dy * dz + dz * dx) * (float)(i + 1) + rarea[i + 1] * (float)(count - i - 1);
            if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = i + 1; }
        }
    }
    free(rarea);
    if (bestAxis < 0) { bestAxis = 0; bestSplit = count / 2; }
    g_axis = bestAxis; g_cent = cent;
    qsort(c->triOrder + first, count, sizeof(uint32_t), cmp_axis);
    uint32_t half = bestSplit;
    int32_t l = build_node(c, cent, first, half);
    int32_t r = build_node(c, cent, first + half, count - half);
    n = &c->nodes[id];
    n->left = l; n->right = r; n->count = 0;
    return id;
}

void or_destroy(OrContext* c)
{
    if (!c) return;
    free(c->vertices); free(c->indices); free(c->meshData); free(c->instances); free(c->materials); free(c->lights);
    if (c->textures) { for (uint32_t i = 0; i < c->textureCount; i++) free(c->textures[i].texels); free(c->textures); }
    free(c->lutTransmittance); free(c->lutScattering); free(c->tris); free(c->triOrder); free(c->nodes);
    free(c);
}

OrContext* or_create(const HrptSceneDesc* s)
{
    if (!s || (!s->vertices && s->vertexCount) || (!s->indices && s->indexCount) || (!s->meshData && s->meshDataCount) ||
        (!s->instances && s->instanceCount) || (!s->materials && s->materialCount) || !s->lights ||
        !s->brunetonTransmittance || !s->brunetonScattering || s->lightCount == 0) { set_err("null scene array"); return NULL; }
    OrContext* c = (OrContext*)calloc(1, sizeof *c);
    c->vertices = dup_mem(s->vertices, (size_t)s->vertexCount * sizeof *s->vertices); c->vertexCount = s->vertexCount;
    c->indices = dup_mem(s->indices, (size_t)s->indexCount * 4); c->indexCount = s->indexCount;
    c->meshData = dup_mem(s->meshData, (size_t)s->meshDataCount * sizeof *s->meshData); c->meshDataCount = s->meshDataCount;
    c->instances = dup_mem(s->instances, (size_t)s->instanceCount * sizeof *s->instances); c->instanceCount = s->instanceCount;
    c->materials = dup_mem(s->materials, (size_t)s->materialCount * sizeof *s->materials); c->materialCount = s->materialCount;
    c->lights = dup_mem(s->lights, (size_t)s->lightCount * sizeof *s->lights); c->lightCount = s->lightCount;
    c->textureCount = s->textureCount;
    c->textures = (OTex*)calloc(s->textureCount ? s->textureCount : 1, sizeof(OTex));
    for (uint32_t i = 0; i < s->textureCount; i++) {
        if (s->textures[i].texels) {
            OTex* t = &c->textures[i];
            t->w = s->textures[i].width; t->h = s->textures[i].height; t->format = s->textures[i].format;
            t->mipCount = s->textures[i].mipCount ? s->textures[i].mipCount : 1u;
            if (t->w == 0 || t->h == 0 || t->format > HRPT_TEXTURE_FORMAT_RGBA32_FLOAT || t->mipCount > HRPT_TEXTURE_MAX_MIPS) { set_err("bad texture description"); or_destroy(c); return NULL; }
            size_t texels = 0;
            for (uint32_t l = 0; l < t->mipCount; l++) {
                t->mipOffset[l] = (uint32_t)texels;
                texels += (size_t)((t->w >> l) ? (t->w >> l) : 1u) * ((t->h >> l) ? (t->h >> l) : 1u);
            }
            size_t bpt = t->format <= HRPT_TEXTURE_FORMAT_RGBA8_SRGB ? 4 : (t->format == HRPT_TEXTURE_FORMAT_RGBA16_FLOAT ? 8 : 16);
            t->texels = dup_mem(s->textures[i].texels, texels * bpt);
        }
    }
    /* LUTs: float32 file format -> RGBA16F, src/CommonResources.cpp:534-558 */
    size_t nT = 256u * 64u * 4u, nS = 256u * 128u * 32u * 4u;
    c->lutTransmittance = (uint16_t*)malloc(nT * 2); c->lutScattering = (uint16_t*)malloc(nS * 2);
    for (size_t i = 0; i < nT; i++) c->lutTransmittance[i] = or_float_to_half(s->brunetonTransmittance[i]);
    for (size_t i = 0; i < nS; i++) c->lutScattering[i] = or_float_to_half(s->brunetonScattering[i]);

    /* validate + world-space triangles. Triangle p of instance i = indices[idxOff[0]+3p+k]. */
    uint64_t triCount = 0;
    for (uint32_t i = 0; i < c->instanceCount; i++) {
        const HrptPerInstanceData* in = &c->instances[i];
        if (in->m_MeshDataIndex >= c->meshDataCount || in->m_MaterialIndex >= c->materialCount || in->m_LODIndex >= 8) { set_err("instance index out of range"); or_destroy(c); return NULL; }
        const HrptMeshData* md = &c->meshData[in->m_MeshDataIndex];
        if ((uint64_t)md->m_IndexOffsets[0] + md->m_IndexCounts[0] > c->indexCount || md->m_IndexCounts[0] % 3) { set_err("mesh index range"); or_destroy(c); return NULL; }
        triCount += md->m_IndexCounts[0] / 3;
    }
    for (uint32_t i = 0; i < c->indexCount; i++) if (c->indices[i] >= c->vertexCount) { set_err("vertex index out of range"); or_destroy(c); return NULL; }
    c->triCount = (uint32_t)triCount;
    c->tris = (WTri*)malloc((triCount ? triCount : 1) * sizeof(WTri));
    uint32_t k = 0;
    for (uint32_t i = 0; i < c->instanceCount; i++) {
        const HrptPerInstanceData* in = &c->instances[i];
        const HrptMeshData* md = &c->meshData[in->m_MeshDataIndex];
        uint32_t opaque = c->materials[in->m_MaterialIndex].m_AlphaMode == HRPT_ALPHA_MODE_OPAQUE; /* src/Scene.cpp:135,151 */
        for (uint32_t p = 0; p < md->m_IndexCounts[0] / 3; p++) {
            WTri* t = &c->tris[k++];
            const uint32_t* ix = &c->indices[md->m_IndexOffsets[0] + 3 * p];
            const float* a = c->vertices[ix[0]].m_Pos; const float* b = c->vertices[ix[1]].m_Pos; const float* d = c->vertices[ix[2]].m_Pos;
            t->p0 = transform_point(V3(a[0], a[1], a[2]), in->m_World);
            t->p1 = transform_point(V3(b[0], b[1], b[2]), in->m_World);
            t->p2 = transform_point(V3(d[0], d[1], d[2]), in->m_World);
            t->inst = i; t->prim = p; t->opaque = opaque;
        }
    }
    c->triOrder = (uint32_t*)malloc((triCount ? triCount : 1) * 4);
    c->nodes = (BNode*)malloc((2 * triCount + 2) * sizeof(BNode));
    float* cent = (float*)malloc((triCount ? triCount : 1) * 12);
    for (uint32_t i = 0; i < c->triCount; i++) {
        c->triOrder[i] = i;
        float a[3], b[3]; tri_bounds(&c->tris[i], a, b);
        for (int q = 0; q < 3; q++) cent[i * 3 + q] = 0.5f * a[q] + 0.5f * b[q];
    }
    c->nodeCount = 0;
    if (c->triCount) build_node(c, cent, 0, c->triCount);
    free(cent);
    return c;
}

/* ------------------------------------------------------------------ intersection */
typedef struct { v3 o, d; float tmin, tmax; } Ray;
typedef struct { int kx, ky, kz; float Sx, Sy, Sz; } RayShear;
typedef struct { float t; uint32_t inst, prim; float u, v; uint32_t opaque; int valid; } Hit;
typedef struct { uint64_t nodes, tris; } TravCount;

static inline RayShear make_shear(v3 d)
{
    RayShear s; const float* dd = &d.x;
    int kz = 0;
    if (hrt_abs(dd[1]) > hrt_abs(dd[kz])) kz = 1;
    if (hrt_abs(dd[2]) > hrt_abs(dd[kz])) kz = 2;
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    if (dd[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
    s.kx = kx; s.ky = ky; s.kz = kz;
    s.Sx = dd[kx] / dd[kz]; s.Sy = dd[ky] / dd[kz]; s.Sz = 1.0f / dd[kz];
    return s;
}
/* key order: (t, inst, prim) */
static inline int key_less(float t, uint32_t inst, uint32_t prim, float t2, uint32_t inst2, uint32_t prim2)
{
    if (t < t2) return 1; if (t > t2) return 0;
    if (inst < inst2) return 1; if (inst > inst2) return 0;
    return prim < prim2;
}
/* Watertight ray/triangle (Woop, Benthin, Wald 2013), fp32 only, no backface culling.
 * Edge-on zeros count as inside. Barycentrics follow DXR: (u,v) = weights of v1, v2. */
static inline int tri_test(const WTri* tr, const Ray* r, const RayShear* s, float* t, float* u, float* v)
{
    v3 A = sub3(tr->p0, r->o), B = sub3(tr->p1, r->o), C = sub3(tr->p2, r->o);
    const float* a = &A.x; const float* b = &B.x; const float* c = &C.x;
    float Ax = a[s->kx] - s->Sx * a[s->kz], Ay = a[s->ky] - s->Sy * a[s->kz];
    float Bx = b[s->kx] - s->Sx * b[s->kz], By = b[s->ky] - s->Sy * b[s->kz];
    float Cx = c[s->kx] - s->Sx * c[s->kz], Cy = c[s->ky] - s->Sy * c[s->kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = (U + V) + W;
    if (det == 0.0f) return 0;
    float Az = s->Sz * a[s->kz], Bz = s->Sz * b[s->kz], Cz = s->Sz * c[s->kz];
    float T = (U * Az + V * Bz) + W * Cz;
    float rcp = 1.0f / det;
    float tt = T * rcp;
    if (!(tt > r->tmin && tt < r->tmax)) return 0;
    *t = tt; *u = V * rcp; *v = W * rcp;
    return 1;
}

/* Closest triangle of any kind with key strictly greater than (lt, linst, lprim) when haveLower. */
static void consider(const OrContext* c, uint32_t ti, const Ray* r, const RayShear* s, int haveLower,
                     float lt, uint32_t linst, uint32_t lprim, Hit* best)
{
    const WTri* tr = &c->tris[ti];
    float t, u, v;
    if (!tri_test(tr, r, s, &t, &u, &v)) return;
    if (haveLower && !key_less(lt, linst, lprim, t, tr->inst, tr->prim)) return;
    if (best->valid && !key_less(t, tr->inst, tr->prim, best->t, best->inst, best->prim)) return;
    best->valid = 1; best->t = t; best->inst = tr->inst; best->prim = tr->prim; best->u = u; best->v = v; best->opaque = tr->opaque;
}

static Hit closest_any(const OrContext* c, const Ray* r, int haveLower, float lt, uint32_t linst, uint32_t lprim,
                       int brute, TravCount* tc)
{
    Hit best; memset(&best, 0, sizeof best);
    if (c->triCount == 0) return best;
    if (!(r->d.x == r->d.x && r->d.y == r->d.y && r->d.z == r->d.z)) return best; /* NaN direction: no hits */
    RayShear s = make_shear(r->d);
    if (brute) {
        for (uint32_t i = 0; i < c->triCount; i++) { tc->tris++; consider(c, i, r, &s, haveLower, lt, linst, lprim, &best); }
        return best;
    }
    /* Ordered traversal: a node is pushed only after its own box was hit; n counts AABB tests
     * (one AABB 24 B + 8 B link each -- the traffic any BVH2 traversal must do). */
    struct { int32_t node; float tnear; } stack[128]; int sp = 0;
    const float* o = &r->o.x; const float* d = &r->d.x;
    float inv[3]; for (int k = 0; k < 3; k++) inv[k] = 1.0f / d[k];
    #define OR_BOX_TEST(NODE, TLIM, HIT, TNEAR) do { \
        const BNode* bn_ = (NODE); float t0_ = r->tmin, t1_ = (TLIM); int miss_ = 0; tc->nodes++; \
        for (int k = 0; k < 3 && !miss_; k++) { \
            if (d[k] == 0.0f) { if (o[k] < bn_->bmin[k] || o[k] > bn_->bmax[k]) miss_ = 1; continue; } \
            float ta_ = (bn_->bmin[k] - o[k]) * inv[k], tb_ = (bn_->bmax[k] - o[k]) * inv[k]; \
            float lo_ = hrt_min(ta_, tb_), hi_ = hrt_max(ta_, tb_); \
            lo_ = lo_ - hrt_abs(lo_) * 1e-6f; hi_ = hi_ + hrt_abs(hi_) * 1e-6f; \
            t0_ = hrt_max(t0_, lo_); t1_ = hrt_min(t1_, hi_); if (t0_ > t1_) miss_ = 1; } \
        (HIT) = !miss_; (TNEAR) = t0_; } while (0)
    int rootHit; float rootNear;
    OR_BOX_TEST(&c->nodes[0], r->tmax, rootHit, rootNear);
    if (rootHit) { stack[sp].node = 0; stack[sp].tnear = rootNear; sp++; }
    while (sp) {
        sp--;
        const BNode* n = &c->nodes[stack[sp].node];
        if (best.valid && stack[sp].tnear > best.t) continue;
        if (n->left < 0) {
            for (uint32_t i = n->first; i < n->first + n->count; i++) { tc->tris++; consider(c, c->triOrder[i], r, &s, haveLower, lt, linst, lprim, &best); }
        } else {
            float tlim = best.valid ? best.t : r->tmax;
            int hl, hr; float tl, tr;
            OR_BOX_TEST(&c->nodes[n->left], tlim, hl, tl);
            OR_BOX_TEST(&c->nodes[n->right], tlim, hr, tr);
            if (sp + 2 > 128) abort();
            if (hl && hr) {
                int leftFirst = tl <= tr;
                stack[sp].node = leftFirst ? n->right : n->left; stack[sp].tnear = leftFirst ? tr : tl; sp++;
                stack[sp].node = leftFirst ? n->left : n->right; stack[sp].tnear = leftFirst ? tl : tr; sp++;
            } else if (hl) { stack[sp].node = n->left; stack[sp].tnear = tl; sp++; }
            else if (hr) { stack[sp].node = n->right; stack[sp].tnear = tr; sp++; }
        }
    }
    #undef OR_BOX_TEST
    return best;
}

/* ------------------------------------------------------------------ vertex fetch */
typedef struct { v3 pos, normal; v2 uv; v4 tangent; } Vtx;

/* DecodeOct, Common.hlsli:174-181 */
static v3 decode_oct(float ex, float ey)
{
    v3 v = V3(ex, ey, (1.0f - hrt_abs(ex)) - hrt_abs(ey));
    float t = hrt_max(-v.z, 0.0f);
    v.x += (v.x >= 0.0f) ? -t : t;
    v.y += (v.y >= 0.0f) ? -t : t;
    return normalize3(v);
}
/* UnpackVertex, MeshCommon.hlsli:9-22 */
static Vtx unpack_vertex(const HrptVertexQuantized* q)
{
    Vtx v;
    v.pos = V3(q->m_Pos[0], q->m_Pos[1], q->m_Pos[2]);
    v.normal.x = (float)(q->m_Normal & 1023u) / 511.0f - 1.0f;
    v.normal.y = (float)((q->m_Normal >> 10) & 1023u) / 511.0f - 1.0f;
    v.normal.z = (float)((q->m_Normal >> 20) & 1023u) / 511.0f - 1.0f;
    float ox = (float)(q->m_Tangent & 255u) / 127.0f - 1.0f;
    float oy = (float)((q->m_Tangent >> 8) & 255u) / 127.0f - 1.0f;
    v3 tg = decode_oct(ox, oy);
    v.tangent.x = tg.x; v.tangent.y = tg.y; v.tangent.z = tg.z;
    v.tangent.w = (q->m_Normal & (1u << 30)) != 0 ? -1.0f : 1.0f;
    v.uv.x = hrt_f16tof32(q->m_Uv & 0xFFFFu); v.uv.y = hrt_f16tof32(q->m_Uv >> 16);
    return v;
}
void or_unpack_vertex(const HrptVertexQuantized* vq, float o[12])
{
    Vtx v = unpack_vertex(vq);
    o[0] = v.pos.x; o[1] = v.pos.y; o[2] = v.pos.z; o[3] = v.normal.x; o[4] = v.normal.y; o[5] = v.normal.z;
    o[6] = v.uv.x; o[7] = v.uv.y; o[8] = v.tangent.x; o[9] = v.tangent.y; o[10] = v.tangent.z; o[11] = v.tangent.w;
}
/* GetTriangleVertices, RaytracingCommon.hlsli:33-50 */
static void get_triangle_vertices(const OrContext* c, uint32_t prim, uint32_t lod, const HrptMeshData* mesh, Vtx tv[3])
{
    uint32_t base = mesh->m_IndexOffsets[lod];
    for (int k = 0; k < 3; k++) {
        uint32_t ii = base + 3 * prim + (uint32_t)k;
        uint32_t vi = (ii < c->indexCount) ? c->indices[ii] : 0u;   /* D3D structured-buffer OOB reads return 0 */
        tv[k] = unpack_vertex(&c->vertices[vi]);
    }
}
/* GetInterpolatedUV, RaytracingCommon.hlsli:79-89 */
static v2 interpolated_uv(const Vtx tv[3], float bx, float by)
{
    float w0 = (1.0f - bx) - by; v2 r;
    r.x = (tv[0].uv.x * w0 + tv[1].uv.x * bx) + tv[2].uv.x * by;
    r.y = (tv[0].uv.y * w0 + tv[1].uv.y * bx) + tv[2].uv.y * by;
    return r;
}

/* ------------------------------------------------------------------ textures */
/* SampleLevel(lod 0) of an RGBA8_UNORM single-mip texture with exact fp32 bilinear.
 * Global sampler table src/CommonResources.cpp:117-128: 0 aniso clamp, 1 aniso wrap,
 * 2 point clamp, 3 point wrap, 4 linear clamp, 5 linear wrap; others -> linear clamp. */
static int wrap_i(int i, int n, int wrap)
{
    if (wrap) { int m = i % n; return m < 0 ? m + n : m; }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
static const float kSrgbToLinear[256] = HRT_SRGB_TO_LINEAR_TABLE;
/* one texel of level `level` (lw texels wide), decoded per HRPT_TEXTURE_FORMAT_*; *_SRGB is linearised before filtering */
static v4 texel_fetch(const OTex* t, uint32_t level, int lw, int x, int y)
{
    size_t idx = (size_t)t->mipOffset[level] + (size_t)y * (size_t)lw + (size_t)x;
    v4 r;
    if (t->format <= HRPT_TEXTURE_FORMAT_RGBA8_SRGB) {
        const uint8_t* p = t->texels + idx * 4;
        if (t->format == HRPT_TEXTURE_FORMAT_RGBA8_SRGB) { r.x = kSrgbToLinear[p[0]]; r.y = kSrgbToLinear[p[1]]; r.z = kSrgbToLinear[p[2]]; }
        else { r.x = (float)p[0] / 255.0f; r.y = (float)p[1] / 255.0f; r.z = (float)p[2] / 255.0f; }
        r.w = (float)p[3] / 255.0f;
    } else if (t->format == HRPT_TEXTURE_FORMAT_RGBA16_FLOAT) {
        const uint16_t* p = (const uint16_t*)t->texels + idx * 4;
        r.x = hrt_f16tof32(p[0]); r.y = hrt_f16tof32(p[1]); r.z = hrt_f16tof32(p[2]); r.w = hrt_f16tof32(p[3]);
    } else {
        const float* p = (const float*)t->texels + idx * 4;
        r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    }
    return r;
}
static v4 lerp4(v4 a, v4 b, float t)
{
    v4 r = { a.x * (1.0f - t) + b.x * t, a.y * (1.0f - t) + b.y * t, a.z * (1.0f - t) + b.z * t, a.w * (1.0f - t) + b.w * t };
    return r;
}
static v4 sample_texture_level(const OTex* t, uint32_t samplerIndex, v2 uv, uint32_t level)
{
    int lw = (int)((t->w >> level) ? (t->w >> level) : 1u), lh = (int)((t->h >> level) ? (t->h >> level) : 1u);
    int wrap = (samplerIndex <= 5u) ? (int)(samplerIndex & 1u) : 0;
    int point = (samplerIndex == 2u || samplerIndex == 3u);
    float fx = uv.x * (float)lw, fy = uv.y * (float)lh;
    if (point) {
        int x = wrap_i((int)hrt_floor(fx), lw, wrap), y = wrap_i((int)hrt_floor(fy), lh, wrap);
        return texel_fetch(t, level, lw, x, y);
    }
    fx = fx - 0.5f; fy = fy - 0.5f;
    float ix = hrt_floor(fx), iy = hrt_floor(fy);
    float tx = fx - ix, ty = fy - iy;
    int x0 = wrap_i((int)ix, lw, wrap), x1 = wrap_i((int)ix + 1, lw, wrap);
    int y0 = wrap_i((int)iy, lh, wrap), y1 = wrap_i((int)iy + 1, lh, wrap);
    v4 a = lerp4(texel_fetch(t, level, lw, x0, y0), texel_fetch(t, level, lw, x1, y0), tx);
    v4 b = lerp4(texel_fetch(t, level, lw, x0, y1), texel_fetch(t, level, lw, x1, y1), tx);
    return lerp4(a, b, ty);
}
static v4 sample_texture(const OrContext* c, uint32_t texIndex, uint32_t samplerIndex, v2 uv)
{
    v4 zero = { 0, 0, 0, 0 };
    if (texIndex >= c->textureCount || !c->textures[texIndex].texels) return zero; /* unbound descriptor reads 0 */
    return sample_texture_level(&c->textures[texIndex], samplerIndex, uv, 0u);
}
/* tex.SampleGrad (Bindless.hlsli:127-132). Level of detail: the isotropic form of the D3D11.3 functional spec, fixed by the numeric
 * contract (see pt_device.h sample_texture_grad): lod = log2(max(|ddx * size|, |ddy * size|)) clamped to the chain; linear and
 * anisotropic samplers blend the two nearest levels, point samplers take the nearest. */
static v4 sample_texture_grad(const OrContext* c, uint32_t texIndex, uint32_t samplerIndex, v2 uv, v2 ddx, v2 ddy)
{
    v4 zero = { 0, 0, 0, 0 };
    if (texIndex >= c->textureCount || !c->textures[texIndex].texels) return zero;
    const OTex* t = &c->textures[texIndex];
    if (t->mipCount <= 1u) return sample_texture_level(t, samplerIndex, uv, 0u);
    float ax = ddx.x * (float)t->w, ay = ddx.y * (float)t->h, bx = ddy.x * (float)t->w, by = ddy.y * (float)t->h;
    float rho2 = hrt_max(ax * ax + ay * ay, bx * bx + by * by);
    float lod = rho2 > 0.0f ? 0.5f * hrt_log2(rho2) : 0.0f;
    lod = hrt_clamp(lod, 0.0f, (float)(t->mipCount - 1u));
    int point = (samplerIndex == 2u || samplerIndex == 3u);
    if (point) return sample_texture_level(t, samplerIndex, uv, (uint32_t)hrt_floor(lod + 0.5f));
    float l0 = hrt_floor(lod), f = lod - l0;
    uint32_t i0 = (uint32_t)l0, i1 = i0 + 1u < t->mipCount ? i0 + 1u : i0;
    v4 a = sample_texture_level(t, samplerIndex, uv, i0);
    if (f == 0.0f || i1 == i0) return a;
    return lerp4(a, sample_texture_level(t, samplerIndex, uv, i1), f);
}

/* RGBA16F LUTs, linear-clamp sampler (index 4). */
static v4 lut_texel(const uint16_t* lut, size_t idx)
{
    const uint16_t* p = lut + idx * 4;
    v4 r = { hrt_f16tof32(p[0]), hrt_f16tof32(p[1]), hrt_f16tof32(p[2]), hrt_f16tof32(p[3]) };
    return r;
}
static int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
static v4 sample_lut2d(const uint16_t* lut, int W, int H, float u, float v)
{
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float ix = hrt_floor(fx), iy = hrt_floor(fy);
    float tx = fx - ix, ty = fy - iy;
    int x0 = clampi((int)ix, W), x1 = clampi((int)ix + 1, W), y0 = clampi((int)iy, H), y1 = clampi((int)iy + 1, H);
    v4 a = lerp4(lut_texel(lut, (size_t)y0 * W + x0), lut_texel(lut, (size_t)y0 * W + x1), tx);
    v4 b = lerp4(lut_texel(lut, (size_t)y1 * W + x0), lut_texel(lut, (size_t)y1 * W + x1), tx);
    return lerp4(a, b, ty);
}
static v4 sample_lut3d(const uint16_t* lut, int W, int H, int D, float u, float v, float w)
{
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f, fz = w * (float)D - 0.5f;
    float ix = hrt_floor(fx), iy = hrt_floor(fy), iz = hrt_floor(fz);
    float tx = fx - ix, ty = fy - iy, tz = fz - iz;
    int x0 = clampi((int)ix, W), x1 = clampi((int)ix + 1, W), y0 = clampi((int)iy, H), y1 = clampi((int)iy + 1, H);
    int z0 = clampi((int)iz, D), z1 = clampi((int)iz + 1, D);
    v4 s[2];
    for (int q = 0; q < 2; q++) {
        size_t zo = (size_t)(q ? z1 : z0) * W * H;
        v4 a = lerp4(lut_texel(lut, zo + (size_t)y0 * W + x0), lut_texel(lut, zo + (size_t)y0 * W + x1), tx);
        v4 b = lerp4(lut_texel(lut, zo + (size_t)y1 * W + x0), lut_texel(lut, zo + (size_t)y1 * W + x1), tx);
        s[q] = lerp4(a, b, ty);
    }
    return lerp4(s[0], s[1], tz);
}

/* ------------------------------------------------------------------ atmosphere (Atmosphere.hlsli) */
#define ATM_BOTTOM 6360.0f
#define ATM_TOP 6420.0f
#define ATM_SUN_ANGULAR_RADIUS 0.004675f   /* 0.00935 / 2.0, :42 */
#define ATM_MU_S_MIN (-0.207912f)
static const v3 ATM_SOLAR_IRRADIANCE = { 1.474000f, 1.850400f, 1.911980f };
static const v3 ATM_RAYLEIGH_SCATTERING = { 0.005802f, 0.013558f, 0.033100f };
static const v3 ATM_MIE_SCATTERING = { 0.003996f, 0.003996f, 0.003996f };
#define ATM_MIE_G 0.8f

static float atm_clamp_distance(float d) { return hrt_max(d, 0.0f); }                 /* :106-109 */
static float atm_safe_sqrt(float a) { return hrt_sqrt(hrt_max(a, 0.0f)); }            /* :116-119 */
static float atm_dist_top(float r, float mu)                                          /* :130-134 */
{
    float disc = r * r * (mu * mu - 1.0f) + ATM_TOP * ATM_TOP;
    return atm_clamp_distance(-r * mu + atm_safe_sqrt(disc));
}
static int atm_ray_hits_ground(float r, float mu)                                     /* :157-160 */
{
    return mu < 0.0f && r * r * (mu * mu - 1.0f) + ATM_BOTTOM * ATM_BOTTOM >= 0.0f;
}
static float atm_texcoord(float x, int size) { return 0.5f / (float)size + x * (1.0f - 1.0f / (float)size); } /* :176-179 */
/* GetTransmittanceTextureUvFromRMu :190-198 + GetTransmittanceToTopAtmosphereBoundary :207-211 */
static v3 atm_transmittance_to_top(const OrContext* c, float r, float mu)
{
    float rho = atm_safe_sqrt(r * r - ATM_BOTTOM * ATM_BOTTOM);
    float d = atm_dist_top(r, mu);
    float H = atm_safe_sqrt(ATM_TOP * ATM_TOP - ATM_BOTTOM * ATM_BOTTOM);
    float x_mu = atm_texcoord(d / (rho + H), 256);
    float x_r = atm_texcoord(rho / H, 64);
    v4 s = sample_lut2d(c->lutTransmittance, 256, 64, x_mu, x_r);
    return V3(s.x, s.y, s.z);
}
/* GetScatteringTextureUvwzFromRMuMuSNu :263-297 */
static v4 atm_scattering_uvwz(float r, float mu, float mu_s, float nu, int hitsGround)
{
    float H = hrt_sqrt(ATM_TOP * ATM_TOP - ATM_BOTTOM * ATM_BOTTOM);
    float rho = atm_safe_sqrt(r * r - ATM_BOTTOM * ATM_BOTTOM);
    float u_r = atm_texcoord(rho / H, 32);
    float r_mu = r * mu;
    float disc = r_mu * r_mu - r * r + ATM_BOTTOM * ATM_BOTTOM;
    float u_mu;
    if (hitsGround) {
        float d = -r_mu - atm_safe_sqrt(disc);
        float d_min = r - ATM_BOTTOM, d_max = rho;
        u_mu = 0.5f - 0.5f * atm_texcoord(d_max == d_min ? 0.0f : (d - d_min) / (d_max - d_min), 128 / 2);
    } else {
        float d = -r_mu + atm_safe_sqrt(disc + H * H);
        float d_min = ATM_TOP - r, d_max = rho + H;
        u_mu = 0.5f + 0.5f * atm_texcoord((d - d_min) / (d_max - d_min), 128 / 2);
    }
    float d = atm_dist_top(ATM_BOTTOM, mu_s);
    float d_min = ATM_TOP - ATM_BOTTOM, d_max = H;
    float a = (d - d_min) / (d_max - d_min);
    float D = atm_dist_top(ATM_BOTTOM, ATM_MU_S_MIN);
    float A = (D - d_min) / (d_max - d_min);
    float u_mu_s = atm_texcoord(hrt_max(1.0f - a / A, 0.0f) / (1.0f + a), 32);
    float u_nu = (nu + 1.0f) / 2.0f;
    v4 o = { u_nu, u_mu_s, u_mu, u_r };
    return o;
}
/* GetCombinedScattering :326-341 + GetExtrapolatedSingleMieScattering :305-314 */
static v3 atm_combined_scattering(const OrContext* c, float r, float mu, float mu_s, float nu, int hitsGround, v3* singleMie)
{
    v4 uvwz = atm_scattering_uvwz(r, mu, mu_s, nu, hitsGround);
    float tex_coord_x = uvwz.x * (float)(8 - 1);
    float tex_x = hrt_floor(tex_coord_x);
    float lerp_val = tex_coord_x - tex_x;
    float u0 = (tex_x + uvwz.y) / 8.0f, u1 = (tex_x + 1.0f + uvwz.y) / 8.0f;
    v4 s0 = sample_lut3d(c->lutScattering, 256, 128, 32, u0, uvwz.z, uvwz.w);
    v4 s1 = sample_lut3d(c->lutScattering, 256, 128, 32, u1, uvwz.z, uvwz.w);
    float w0 = 1.0f - lerp_val;
    v4 cs = { s0.x * w0 + s1.x * lerp_val, s0.y * w0 + s1.y * lerp_val, s0.z * w0 + s1.z * lerp_val, s0.w * w0 + s1.w * lerp_val };
    if (cs.x <= 0.0f) *singleMie = V3(0, 0, 0);
    else {
        /* scattering.rgb * scattering.a / scattering.r * (ray.r / mie.r) * (mie / ray) */
        float k = ATM_RAYLEIGH_SCATTERING.x / ATM_MIE_SCATTERING.x;
        v3 ratio = V3(ATM_MIE_SCATTERING.x / ATM_RAYLEIGH_SCATTERING.x, ATM_MIE_SCATTERING.y / ATM_RAYLEIGH_SCATTERING.y, ATM_MIE_SCATTERING.z / ATM_RAYLEIGH_SCATTERING.z);
        v3 t = V3(cs.x * cs.w / cs.x * k, cs.y * cs.w / cs.x * k, cs.z * cs.w / cs.x * k);
        *singleMie = mul3(t, ratio);
    }
    return V3(cs.x, cs.y, cs.z);
}
static float atm_rayleigh_phase(float nu) { float k = 3.0f / (16.0f * HRT_PI); return k * (1.0f + nu * nu); }   /* :349-353 */
static float atm_mie_phase(float g, float nu)                                                                    /* :355-359 */
{
    float k = 3.0f / (8.0f * HRT_PI) * (1.0f - g * g) / (2.0f + g * g);
    float b = hrt_max(1.0f + g * g - 2.0f * g * nu, 0.0001f);
    return k * (1.0f + nu * nu) / (b * hrt_sqrt(b));   /* pow(b, 1.5) == b*sqrt(b) */
}
static float smoothstep1(float a, float b, float x) { float t = hrt_saturate((x - a) / (b - a)); return (t * t) * (3.0f - 2.0f * t); }
/* GetTransmittanceToSun :414-420 */
static v3 atm_transmittance_to_sun(const OrContext* c, float r, float mu_s)
{
    float sin_theta_h = ATM_BOTTOM / r;
    float cos_theta_h = -hrt_sqrt(hrt_max(1.0f - sin_theta_h * sin_theta_h, 0.0f));
    float e = sin_theta_h * ATM_SUN_ANGULAR_RADIUS;
    float f = smoothstep1(-e, e, mu_s - cos_theta_h);
    return scale3(atm_transmittance_to_top(c, r, mu_s), f);
}
/* GetAtmospherePos :564-567, kEarthCenter Common.hlsli:202 */
static v3 atm_pos(v3 worldPos) { return div3s(sub3(worldPos, V3(0.0f, -6360000.0f, 0.0f)), 1000.0f); }
/* GetAtmosphereSunRadiance :569-574 */
static v3 atm_sun_radiance(const OrContext* c, v3 p_atmo, v3 sunDir, float sunIntensity)
{
    float r = length3(p_atmo);
    float mu_s = dot3(p_atmo, sunDir) / r;
    return scale3(mul3(ATM_SOLAR_IRRADIANCE, atm_transmittance_to_sun(c, r, mu_s)), sunIntensity);
}
/* GetSkyRadiance :458-500 with shadow_length == 0 */
static v3 atm_sky_radiance_core(const OrContext* c, v3 camera, v3 view_ray, v3 sun_direction, v3* transmittance)
{
    float r = length3(camera);
    float rmu = dot3(camera, view_ray);
    float dist_top = -rmu - atm_safe_sqrt(rmu * rmu - r * r + ATM_TOP * ATM_TOP);
    if (dist_top > 0.0f) {
        camera = add3(camera, scale3(view_ray, dist_top));
        r = ATM_TOP;
        rmu += dist_top;
    } else if (r > ATM_TOP) {
        *transmittance = V3(1, 1, 1);
        return V3(0, 0, 0);
    }
    float mu = rmu / r;
    float mu_s = dot3(camera, sun_direction) / r;
    float nu = dot3(view_ray, sun_direction);
    int hitsGround = atm_ray_hits_ground(r, mu);
    *transmittance = hitsGround ? V3(0, 0, 0) : atm_transmittance_to_top(c, r, mu);
    v3 mie;
    v3 scat = atm_combined_scattering(c, r, mu, mu_s, nu, hitsGround, &mie);
    return add3(scale3(scat, atm_rayleigh_phase(nu)), scale3(mie, atm_mie_phase(ATM_MIE_G, nu)));
}
/* GetAtmosphereSkyRadiance :583-601 */
static v3 atm_sky_radiance(const OrContext* c, v3 cameraPos, v3 viewRay, v3 sunDir, float sunIntensity, int addSunDisk)
{
    v3 cam = atm_pos(cameraPos);
    v3 transmittance;
    v3 sky = atm_sky_radiance_core(c, cam, viewRay, sunDir, &transmittance);
    if (addSunDisk) {
        float nu = dot3(viewRay, sunDir);
        float sar = ATM_SUN_ANGULAR_RADIUS;
        if (nu > hrt_cos(sar)) {
            float den = HRT_PI * sar * sar;
            v3 disk = V3(ATM_SOLAR_IRRADIANCE.x / den, ATM_SOLAR_IRRADIANCE.y / den, ATM_SOLAR_IRRADIANCE.z / den);
            sky = add3(sky, mul3(disk, transmittance));
        }
    }
    return scale3(sky, sunIntensity);
}
void or_sky_radiance(OrContext* c, const float cp[3], const float vr[3], const float sd[3], float si, int disk, float out[3])
{
    v3 r = atm_sky_radiance(c, V3(cp[0], cp[1], cp[2]), V3(vr[0], vr[1], vr[2]), V3(sd[0], sd[1], sd[2]), si, disk);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void or_sun_radiance(OrContext* c, const float wp[3], const float sd[3], float si, float out[3])
{
    v3 r = atm_sun_radiance(c, atm_pos(V3(wp[0], wp[1], wp[2])), V3(sd[0], sd[1], sd[2]), si);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------ shading (CommonLighting.hlsli) */
typedef struct {
    v3 N, V, L, baseColor; float roughness, metallic, ior; v3 worldPos;
    v3 sunRadiance, sunDirection;
    v3 F0, kD, F; float NdotV, NdotL, NdotH, VdotH, LdotV, LdotH;
} LightingInputs;
typedef struct { v3 diffuse, specular; } LightingComponents;

/* F_Schlick :123-129 */
static v3 f_schlick(v3 spec, float VdotH)
{
    float Fc = hrt_pow5(1.0f - VdotH);
    float s = hrt_saturate(50.0f * spec.y) * Fc;
    float k = 1.0f - Fc;
    return V3(s + k * spec.x, s + k * spec.y, s + k * spec.z);
}
/* ComputeF0 :64-68 (pow(x,2) = x*x) */
static v3 compute_f0(v3 baseColor, float metallic, float ior)
{
    float q = (ior - 1.0f) / (ior + 1.0f);
    float d = q * q;
    return V3(lerp1(d, baseColor.x, metallic), lerp1(d, baseColor.y, metallic), lerp1(d, baseColor.z, metallic));
}
/* PrepareLightingByproducts :316-334 */
static void prepare_byproducts(LightingInputs* in)
{
    in->NdotV = hrt_saturate(dot3(in->N, in->V));
    in->NdotL = hrt_saturate(dot3(in->N, in->L));
    v3 VpL = add3(in->V, in->L);
    float len = dot3(VpL, VpL);
    v3 H = (len > 1e-8f) ? scale3(VpL, hrt_rsqrt(len)) : in->N;
    in->NdotH = hrt_saturate(dot3(in->N, H));
    in->VdotH = hrt_saturate(dot3(in->V, H));
    in->LdotV = hrt_saturate(dot3(in->L, in->V));
    in->LdotH = hrt_saturate(dot3(in->L, H));
    in->F0 = compute_f0(in->baseColor, in->metallic, in->ior);
    float kd = 1.0f - in->metallic; in->kD = V3(kd, kd, kd);
    in->F = f_schlick(in->F0, in->VdotH);
}
/* D_GGX :80-86 */
static float d_ggx(float NdotH, float roughness)
{
    float alpha = roughness * roughness, alpha2 = alpha * alpha;
    float denom = NdotH * NdotH * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (HRT_PI * denom * denom);
}
/* DisneyBurleyDiffuse :148-168 */
static float burley_diffuse(float NdotL, float NdotV, float LdotH, float rough)
{
    if (NdotL <= 0.0f || NdotV <= 0.0f) return 0.0f;
    float rough2 = rough * rough;
    float FL = hrt_pow5(1.0f - NdotL), FV = hrt_pow5(1.0f - NdotV);
    float Fd90 = 0.5f + 2.0f * rough2 * LdotH * LdotH;
    float Fd = lerp1(1.0f, Fd90, FL) * lerp1(1.0f, Fd90, FV);
    return Fd * NdotL / HRT_PI;
}
/* ComputeSpecularBRDF :345-358 */
static v3 specular_brdf(v3 F, float NdotH, float NdotV, float NdotL, float roughness)
{
    float alpha = roughness * roughness, alpha2 = alpha * alpha;
    float D = d_ggx(NdotH, roughness);
    float g1 = NdotV * hrt_sqrt(alpha2 + (1.0f - alpha2) * NdotL * NdotL);
    float g2 = NdotL * hrt_sqrt(alpha2 + (1.0f - alpha2) * NdotV * NdotV);
    float G2 = 0.5f / hrt_max(g1 + g2, 1e-6f);
    float k = D * G2;
    return scale3(F, k);
}
/* EvaluateDirectLight :360-375 */
static LightingComponents evaluate_direct(const LightingInputs* in, v3 radiance, float shadow)
{
    LightingComponents o;
    float dt = burley_diffuse(in->NdotL, in->NdotV, in->LdotH, in->roughness);
    v3 diffuse = mul3(scale3(in->kD, dt), in->baseColor);
    v3 spec = specular_brdf(in->F, in->NdotH, in->NdotV, in->NdotL, in->roughness);
    o.diffuse = scale3(mul3(diffuse, radiance), shadow);
    o.specular = scale3(mul3(scale3(spec, in->NdotL), radiance), shadow);
    return o;
}
/* tangent frame shared by SampleHemisphereCosine :170-183, BuildTangentFrame :610-615, SampleConeSolidAngle :702-705 */
static void tangent_frame(v3 N, v3* T, v3* B)
{
    v3 up = hrt_abs(N.z) < 0.999f ? V3(0, 0, 1) : V3(1, 0, 0);
    *T = normalize3(cross3(up, N));
    *B = cross3(N, *T);
}
static v3 frame_combine(v3 T, v3 N, v3 B, v3 l) { return add3(add3(scale3(T, l.x), scale3(N, l.y)), scale3(B, l.z)); }
/* SampleHemisphereCosine :170-183 */
static v3 sample_hemisphere_cosine(float ux, float uy, v3 normal)
{
    float phi = 2.0f * HRT_PI * ux;
    float cosTheta = hrt_sqrt(uy);
    float sinTheta = hrt_sqrt(hrt_max(0.0f, 1.0f - cosTheta * cosTheta));
    float sp, cp; hrt_sincos(phi, &sp, &cp);
    v3 l = V3(sinTheta * cp, cosTheta, sinTheta * sp);
    v3 T, B; tangent_frame(normal, &T, &B);
    return frame_combine(T, normal, B, l);
}
/* SampleGGX_VNDF :622-655 */
static v3 sample_ggx_vndf(float ux, float uy, v3 N, v3 V, float roughness)
{
    float alpha = roughness * roughness;
    v3 T, B; tangent_frame(N, &T, &B);
    v3 Vl = V3(dot3(V, T), dot3(V, N), dot3(V, B));
    v3 Vh = normalize3(V3(alpha * Vl.x, Vl.y, alpha * Vl.z));
    float lensq = Vh.x * Vh.x + Vh.z * Vh.z;
    v3 T1 = lensq > 0.0f ? div3s(V3(-Vh.z, 0.0f, Vh.x), hrt_sqrt(lensq)) : V3(1, 0, 0);
    v3 T2 = cross3(Vh, T1);
    float r = hrt_sqrt(ux);
    float phi = 2.0f * HRT_PI * uy;
    float sp, cp; hrt_sincos(phi, &sp, &cp);
    float t1 = r * cp, t2 = r * sp;
    float s = 0.5f * (1.0f + Vh.y);
    t2 = lerp1(hrt_sqrt(hrt_max(0.0f, 1.0f - t1 * t1)), t2, s);
    float nz = hrt_sqrt(hrt_max(0.0f, (1.0f - t1 * t1) - t2 * t2));
    v3 Nh = add3(add3(scale3(T1, t1), scale3(T2, t2)), scale3(Vh, nz));
    v3 Ne = normalize3(V3(alpha * Nh.x, hrt_max(0.0f, Nh.y), alpha * Nh.z));
    return frame_combine(T, N, B, Ne);
}
/* EvalGGX_VNDF_Weight :671-688 */
static v3 eval_ggx_vndf_weight(v3 F0, v3 N, v3 V, v3 L, v3 H, float roughness)
{
    float alpha = roughness * roughness, alpha2 = alpha * alpha;
    float NdotV = hrt_saturate(dot3(N, V)), NdotL = hrt_saturate(dot3(N, L)), VdotH = hrt_saturate(dot3(V, H));
    if (NdotV <= 0.0f || NdotL <= 0.0f) return V3(0, 0, 0);
    v3 F = f_schlick(F0, VdotH);
    float G1L = 2.0f * NdotL / (NdotL + hrt_sqrt(alpha2 + (1.0f - alpha2) * NdotL * NdotL));
    return scale3(F, G1L);
}
/* SampleConeSolidAngle :693-708 */
static v3 sample_cone(v3 dir, float cosHalf, float ux, float uy)
{
    float cosTheta = 1.0f - ux * (1.0f - cosHalf);
    float sinTheta = hrt_sqrt(hrt_max(0.0f, 1.0f - cosTheta * cosTheta));
    float phi = 2.0f * HRT_PI * uy;
    float sp, cp; hrt_sincos(phi, &sp, &cp);
    v3 l = V3(sinTheta * cp, cosTheta, sinTheta * sp);
    v3 T, B; tangent_frame(dir, &T, &B);
    return frame_combine(T, dir, B, l);
}

/* ------------------------------------------------------------------ tracing with alpha */
typedef struct { OrContext* c; OrStats st; int brute; } Tls;

/* alpha of a candidate: mat.m_BaseColor.w [* albedo.a] */
static float candidate_alpha(const OrContext* c, const HrptMaterialConstants* mat, v2 uv)
{
    float alpha = mat->m_BaseColor[3];
    if (mat->m_TextureFlags & HRPT_TEXFLAG_ALBEDO) alpha *= sample_texture(c, mat->m_AlbedoTextureIndex, mat->m_AlbedoSamplerIndex, uv).w;
    return alpha;
}

/* TraceRayStandard, RaytracingCommon.hlsli:138-198 */
static int trace_ray_standard(Tls* tl, const Ray* ray, uint32_t* rng, Hit* out)
{
    const OrContext* c = tl->c;
    tl->st.closestRays++;
    TravCount tc = { 0, 0 };
    int haveLower = 0; float lt = 0; uint32_t li = 0, lp = 0;
    int result = 0;
    for (;;) {
        Hit h = closest_any(c, ray, haveLower, lt, li, lp, tl->brute, &tc);
        if (!h.valid) break;
        if (h.opaque) { *out = h; result = 1; break; }
        const HrptPerInstanceData* inst = &c->instances[h.inst];
        const HrptMeshData* mesh = &c->meshData[inst->m_MeshDataIndex];
        const HrptMaterialConstants* mat = &c->materials[inst->m_MaterialIndex];
        Vtx tv[3]; get_triangle_vertices(c, h.prim, 0, mesh, tv);
        v2 uv = interpolated_uv(tv, h.u, h.v);
        int commit = 0;
        if (mat->m_AlphaMode == HRPT_ALPHA_MODE_MASK) {
            commit = candidate_alpha(c, mat, uv) >= mat->m_AlphaCutoff;      /* AlphaTest :91-110 */
        } else if (mat->m_AlphaMode == HRPT_ALPHA_MODE_BLEND) {
            float alpha = candidate_alpha(c, mat, uv);
            if (mat->m_TransmissionFactor > 0.0f) commit = 1;
            else commit = hrt_rng_next(rng) < hrt_saturate(alpha);
        }
        if (commit) { *out = h; result = 1; break; }
        haveLower = 1; lt = h.t; li = h.inst; lp = h.prim;
        tl->st.retraces++;
    }
    tl->st.closestNodes += tc.nodes; tl->st.closestTris += tc.tris;
    return result;
}

/* CalculateRTShadow<true>, CommonLighting.hlsli:380-496. Candidates front to back. */
static float calculate_rt_shadow(Tls* tl, v3 worldPos, v3 L, float maxDist)
{
    const OrContext* c = tl->c;
    tl->st.shadowRays++;
    const float kShadowBias = 0.01f;
    Ray ray; ray.o = worldPos; ray.d = L; ray.tmin = kShadowBias; ray.tmax = hrt_max(kShadowBias, maxDist - kShadowBias * 2.0f);
    float transmission = 1.0f; int inVolume = 0; float inVolumeStartT = 0.0f; v3 sigmaT = V3(0, 0, 0);
    TravCount tc = { 0, 0 };
    int haveLower = 0; float lt = 0; uint32_t li = 0, lp = 0;
    int committed = 0;
    for (;;) {
        Hit h = closest_any(c, &ray, haveLower, lt, li, lp, tl->brute, &tc);
        if (!h.valid) break;
        if (h.opaque) { committed = 1; break; }
        const HrptPerInstanceData* inst = &c->instances[h.inst];
        const HrptMeshData* mesh = &c->meshData[inst->m_MeshDataIndex];
        const HrptMaterialConstants* mat = &c->materials[inst->m_MaterialIndex];
        if (mat->m_AlphaMode == HRPT_ALPHA_MODE_MASK) {
            /* AlphaTestGrad (RaytracingCommon.hlsli:112-130) with the gradients of GetShadowRayGradients (:207-240) */
            Vtx tv[3]; get_triangle_vertices(c, h.prim, inst->m_LODIndex, mesh, tv);
            v2 uv = interpolated_uv(tv, h.u, h.v);
            float alpha = mat->m_BaseColor[3];
            if (mat->m_TextureFlags & HRPT_TEXFLAG_ALBEDO) {
                v3 p0 = transform_point(tv[0].pos, inst->m_World), p1 = transform_point(tv[1].pos, inst->m_World), p2 = transform_point(tv[2].pos, inst->m_World);
                float w0 = (1.0f - h.u) - h.v;
                v3 hitPos = add3(add3(scale3(p0, w0), scale3(p1, h.u)), scale3(p2, h.v));
                float dist = length3(sub3(hitPos, ray.o));
                float triangleArea = length3(cross3(sub3(p1, p0), sub3(p2, p0))) * 0.5f;
                v2 uvRange;
                uvRange.x = hrt_max(tv[0].uv.x, hrt_max(tv[1].uv.x, tv[2].uv.x)) - hrt_min(tv[0].uv.x, hrt_min(tv[1].uv.x, tv[2].uv.x));
                uvRange.y = hrt_max(tv[0].uv.y, hrt_max(tv[1].uv.y, tv[2].uv.y)) - hrt_min(tv[0].uv.y, hrt_min(tv[1].uv.y, tv[2].uv.y));
                float gradientScale = triangleArea / hrt_max(dist, 0.1f);
                v2 grad = { uvRange.x * gradientScale, uvRange.y * gradientScale };
                alpha *= sample_texture_grad(c, mat->m_AlbedoTextureIndex, mat->m_AlbedoSamplerIndex, uv, grad, grad).w;
            }
            if (alpha >= mat->m_AlphaCutoff) { committed = 1; break; }
        } else if (mat->m_AlphaMode == HRPT_ALPHA_MODE_BLEND) {
            Vtx tv[3]; get_triangle_vertices(c, h.prim, inst->m_LODIndex, mesh, tv);
            v2 uv = interpolated_uv(tv, h.u, h.v);
            float alpha = candidate_alpha(c, mat, uv);
            float opacity = hrt_saturate(alpha * (1.0f - mat->m_TransmissionFactor));
            transmission *= (1.0f - opacity);
            if (mat->m_TransmissionFactor > 0.0f && mat->m_IsThinSurface == 0) {
                float w0 = (1.0f - h.u) - h.v;
                v3 ln = add3(add3(scale3(tv[0].normal, w0), scale3(tv[1].normal, h.u)), scale3(tv[2].normal, h.v));
                v3 wn = normalize3(transform_normal(ln, inst->m_World));
                int front = dot3(wn, ray.d) < 0.0f;
                if (front) {
                    inVolume = 1; inVolumeStartT = h.t;
                    sigmaT = V3(mat->m_SigmaA[0] + mat->m_SigmaS[0], mat->m_SigmaA[1] + mat->m_SigmaS[1], mat->m_SigmaA[2] + mat->m_SigmaS[2]);
                } else if (inVolume) {
                    float seg = hrt_max(0.0f, h.t - inVolumeStartT);
                    v3 tr = V3(hrt_exp(-sigmaT.x * seg), hrt_exp(-sigmaT.y * seg), hrt_exp(-sigmaT.z * seg));
                    transmission *= dot3(tr, V3(0.2126f, 0.7152f, 0.0722f));
                    inVolume = 0;
                }
            }
            if (transmission <= 1e-3f) { committed = 1; break; }
        } else { committed = 1; break; }
        haveLower = 1; lt = h.t; li = h.inst; lp = h.prim;
        tl->st.retraces++;
    }
    tl->st.shadowNodes += tc.nodes; tl->st.shadowTris += tc.tris;
    if (committed) return 0.0f;
    if (inVolume) {
        float seg = hrt_max(0.0f, ray.tmax - inVolumeStartT);
        v3 tr = V3(hrt_exp(-sigmaT.x * seg), hrt_exp(-sigmaT.y * seg), hrt_exp(-sigmaT.z * seg));
        transmission *= dot3(tr, V3(0.2126f, 0.7152f, 0.0722f));
    }
    return hrt_saturate(transmission);
}

/* ---- per-light NEE (RNG overloads), CommonLighting.hlsli:716-908 ---- */
static LightingComponents lc_zero(void) { LightingComponents o; o.diffuse = V3(0, 0, 0); o.specular = V3(0, 0, 0); return o; }

static LightingComponents directional_lighting(Tls* tl, LightingInputs in, float cosSun, uint32_t* rng)   /* :716-745 */
{
    LightingComponents res = lc_zero();
    if (dot3(in.N, in.sunDirection) <= 0.0f) return res;
    v3 radiance = in.sunRadiance;   /* useSunRadiance == true, PathTracer.hlsl:139 */
    float ux = hrt_rng_next(rng), uy = hrt_rng_next(rng);
    v3 Ls = sample_cone(in.sunDirection, cosSun, ux, uy);
    if (!(dot3(in.N, Ls) <= 0.0f)) {
        in.L = Ls; prepare_byproducts(&in);
        float shadow = calculate_rt_shadow(tl, in.worldPos, Ls, 1e10f);
        LightingComponents comp = evaluate_direct(&in, radiance, shadow);
        res.diffuse = add3(res.diffuse, comp.diffuse); res.specular = add3(res.specular, comp.specular);
    }
    res.diffuse = div3s(res.diffuse, 1.0f); res.specular = div3s(res.specular, 1.0f);   /* /float(LIGHT_SHADOW_SAMPLES) */
    return res;
}
static float distance_attenuation(const HrptGPULight* l, float distSq, float dist)
{
    float a = 1.0f / (distSq + 1.0f);
    if (l->m_Range > 0.0f) {
        float q = dist / l->m_Range; float q2 = q * q; float q4 = q2 * q2;   /* pow(x,4) */
        float s = hrt_saturate(1.0f - q4);
        a *= s * s;                                                            /* pow(x,2) */
    }
    return a;
}
/* shared sample loop of ComputePointLighting :773-800 / ComputeSpotLighting :843-869 */
static LightingComponents sphere_light_sample(Tls* tl, LightingInputs in, const HrptGPULight* l, v3 radiance, uint32_t* rng)
{
    LightingComponents res = lc_zero();
    float ux = hrt_rng_next(rng), uy = hrt_rng_next(rng);
    float cosT = 1.0f - 2.0f * ux;
    float sinT = hrt_sqrt(hrt_max(0.0f, 1.0f - cosT * cosT));
    float phi = 2.0f * HRT_PI * uy;
    float sp, cp; hrt_sincos(phi, &sp, &cp);
    v3 sphereDir = V3(sinT * cp, cosT, sinT * sp);
    v3 lp = V3(l->m_Position[0], l->m_Position[1], l->m_Position[2]);
    v3 samplePos = add3(lp, scale3(sphereDir, l->m_Radius));
    v3 toSample = sub3(samplePos, in.worldPos);
    float sampleDist = length3(toSample);
    v3 Ls = div3s(toSample, sampleDist);
    if (!(dot3(in.N, Ls) <= 0.0f)) {
        in.L = Ls; prepare_byproducts(&in);
        float shadow = calculate_rt_shadow(tl, in.worldPos, Ls, sampleDist);
        LightingComponents comp = evaluate_direct(&in, radiance, shadow);
        res.diffuse = add3(res.diffuse, comp.diffuse); res.specular = add3(res.specular, comp.specular);
    }
    res.diffuse = div3s(res.diffuse, 1.0f); res.specular = div3s(res.specular, 1.0f);
    return res;
}
static LightingComponents point_lighting(Tls* tl, const LightingInputs* in, const HrptGPULight* l, uint32_t* rng)   /* :752-804 */
{
    if (l->m_Intensity <= 0.0f) return lc_zero();
    v3 lp = V3(l->m_Position[0], l->m_Position[1], l->m_Position[2]);
    v3 toLight = sub3(lp, in->worldPos);
    float distSq = dot3(toLight, toLight);
    if (l->m_Range > 0.0f && distSq > l->m_Range * l->m_Range) return lc_zero();
    float dist = hrt_sqrt(distSq);
    float att = distance_attenuation(l, distSq, dist);
    v3 col = V3(l->m_Color[0], l->m_Color[1], l->m_Color[2]);
    v3 radiance = scale3(scale3(col, l->m_Intensity), att);
    return sphere_light_sample(tl, *in, l, radiance, rng);
}
static LightingComponents spot_lighting(Tls* tl, const LightingInputs* in, const HrptGPULight* l, uint32_t* rng)    /* :809-874 */
{
    if (l->m_Intensity <= 0.0f) return lc_zero();
    v3 lp = V3(l->m_Position[0], l->m_Position[1], l->m_Position[2]);
    v3 Lun = sub3(lp, in->worldPos);
    float distSq = dot3(Lun, Lun);
    if (l->m_Range > 0.0f && distSq > l->m_Range * l->m_Range) return lc_zero();
    float dist = hrt_sqrt(distSq);
    v3 Lc = div3s(Lun, dist);
    if (dot3(in->N, Lc) <= 0.0f) return lc_zero();
    v3 lightDir = normalize3(V3(l->m_Direction[0], l->m_Direction[1], l->m_Direction[2]));
    float cosTheta = dot3(neg3(Lc), lightDir);
    float cosOuter = hrt_cos(l->m_SpotOuterConeAngle);
    if (cosTheta < cosOuter) return lc_zero();
    float cosInner = hrt_cos(l->m_SpotInnerConeAngle);
    float spotAtt = hrt_saturate((cosTheta - cosOuter) / (cosInner - cosOuter));
    float att = distance_attenuation(l, distSq, dist);
    v3 col = V3(l->m_Color[0], l->m_Color[1], l->m_Color[2]);
    v3 radiance = scale3(scale3(scale3(col, l->m_Intensity), spotAtt), att);
    return sphere_light_sample(tl, *in, l, radiance, rng);
}
/* AccumulateDirectLighting :877-908 */
static LightingComponents accumulate_direct(Tls* tl, const LightingInputs* in, uint32_t lightCount, float cosSun, uint32_t* rng)
{
    LightingComponents total = lc_zero();
    const OrContext* c = tl->c;
    for (uint32_t i = 0; i < lightCount; i++) {
        HrptGPULight zero; memset(&zero, 0, sizeof zero);
        const HrptGPULight* l = (i < c->lightCount) ? &c->lights[i] : &zero;   /* OOB structured read = 0 */
        LightingComponents comp;
        if (l->m_Type == 0) comp = directional_lighting(tl, *in, cosSun, rng);
        else if (l->m_Type == 1) comp = point_lighting(tl, in, l, rng);
        else if (l->m_Type == 2) comp = spot_lighting(tl, in, l, rng);
        else continue;
        total.diffuse = add3(total.diffuse, comp.diffuse); total.specular = add3(total.specular, comp.specular);
    }
    return total;
}

/* ------------------------------------------------------------------ PathTracer.hlsl */
/* EvalFresnelDielectric :26-43 */
static float fresnel_dielectric(float eta, float cosThetaI, float* cosThetaT)
{
    if (cosThetaI < 0.0f) { eta = 1.0f / eta; cosThetaI = -cosThetaI; }
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq >= 1.0f) { *cosThetaT = 0.0f; return 1.0f; }
    *cosThetaT = hrt_sqrt(hrt_max(0.0f, 1.0f - sinThetaTSq));
    float Rs = (eta * cosThetaI - *cosThetaT) / (eta * cosThetaI + *cosThetaT);
    float Rp = (eta * *cosThetaT - cosThetaI) / (eta * *cosThetaT + cosThetaI);
    return 0.5f * (Rs * Rs + Rp * Rp);
}
float or_fresnel_dielectric(float eta, float c, float* ct) { return fresnel_dielectric(eta, c, ct); }
/* EvalTransmittance :47-50 */
static v3 eval_transmittance(v3 sa, v3 ss, float dist)
{
    return V3(hrt_exp(-(sa.x + ss.x) * dist), hrt_exp(-(sa.y + ss.y) * dist), hrt_exp(-(sa.z + ss.z) * dist));
}

typedef struct { v3 worldPos, worldNormal, worldTangent; float tangentSign; v2 uv; } FullHitAttributes;
typedef struct { v3 baseColor; float alpha, roughness, metallic; v3 emissive, normal; } PBRAttributes;

/* GetFullHitAttributes, RaytracingCommon.hlsli:52-77 */
static FullHitAttributes full_hit_attributes(const OrContext* c, const Hit* hit, const Ray* ray, const HrptPerInstanceData* inst, uint32_t lod, const HrptMeshData* mesh)
{
    Vtx tv[3]; get_triangle_vertices(c, hit->prim, lod, mesh, tv);
    float bx = (1.0f - hit->u) - hit->v, by = hit->u, bz = hit->v;
    FullHitAttributes a;
    a.worldPos = add3(ray->o, scale3(ray->d, hit->t));
    v3 ln = add3(add3(scale3(tv[0].normal, bx), scale3(tv[1].normal, by)), scale3(tv[2].normal, bz));
    a.worldNormal = transform_normal(ln, inst->m_World);
    v3 t0 = V3(tv[0].tangent.x, tv[0].tangent.y, tv[0].tangent.z), t1 = V3(tv[1].tangent.x, tv[1].tangent.y, tv[1].tangent.z), t2 = V3(tv[2].tangent.x, tv[2].tangent.y, tv[2].tangent.z);
    v3 lt = add3(add3(scale3(t0, bx), scale3(t1, by)), scale3(t2, bz));
    a.worldTangent = transform_normal(lt, inst->m_World);
    a.tangentSign = (tv[0].tangent.w * bx + tv[1].tangent.w * by) + tv[2].tangent.w * bz;
    a.uv.x = (tv[0].uv.x * bx + tv[1].uv.x * by) + tv[2].uv.x * bz;
    a.uv.y = (tv[0].uv.y * bx + tv[1].uv.y * by) + tv[2].uv.y * bz;
    return a;
}
/* TransformNormalWithTBN, Common.hlsli:183-200 */
static v3 normal_with_tbn(float nx, float ny, v3 normal, v3 tangent, float tangentSign)
{
    float x = 2.0f * nx - 1.0f, y = 2.0f * ny - 1.0f;
    float z = hrt_sqrt(hrt_saturate(1.0f - (x * x + y * y)));
    v3 n_w = normalize3(normal);
    v3 t_w = normalize3(tangent);
    t_w = normalize3(sub3(t_w, scale3(n_w, dot3(t_w, n_w))));
    v3 b_w = normalize3(scale3(cross3(n_w, t_w), tangentSign));
    /* mul(normalMap, float3x3(t,b,n)) */
    v3 o = V3((x * t_w.x + y * b_w.x) + z * n_w.x, (x * t_w.y + y * b_w.y) + z * n_w.y, (x * t_w.z + y * b_w.z) + z * n_w.z);
    return normalize3(o);
}
/* GetPBRAttributes, RaytracingCommon.hlsli:252-296 */
static PBRAttributes pbr_attributes(const OrContext* c, const FullHitAttributes* a, const HrptMaterialConstants* m)
{
    PBRAttributes p;
    p.baseColor = V3(m->m_BaseColor[0], m->m_BaseColor[1], m->m_BaseColor[2]);
    p.alpha = m->m_BaseColor[3];
    if (m->m_TextureFlags & HRPT_TEXFLAG_ALBEDO) {
        v4 s = sample_texture(c, m->m_AlbedoTextureIndex, m->m_AlbedoSamplerIndex, a->uv);
        p.baseColor = mul3(p.baseColor, V3(s.x, s.y, s.z)); p.alpha *= s.w;
    }
    p.roughness = m->m_RoughnessMetallic[0];
    if (m->m_TextureFlags & HRPT_TEXFLAG_ROUGHNESS_METALLIC)
        p.roughness = sample_texture(c, m->m_RoughnessMetallicTextureIndex, m->m_RoughnessSamplerIndex, a->uv).y;
    p.roughness = hrt_max(p.roughness, 0.04f);
    p.metallic = m->m_RoughnessMetallic[1];
    if (m->m_TextureFlags & HRPT_TEXFLAG_ROUGHNESS_METALLIC)
        p.metallic = sample_texture(c, m->m_RoughnessMetallicTextureIndex, m->m_RoughnessSamplerIndex, a->uv).z;
    p.emissive = V3(m->m_EmissiveFactor[0], m->m_EmissiveFactor[1], m->m_EmissiveFactor[2]);
    if (m->m_TextureFlags & HRPT_TEXFLAG_EMISSIVE) {
        v4 s = sample_texture(c, m->m_EmissiveTextureIndex, m->m_EmissiveSamplerIndex, a->uv);
        p.emissive = mul3(p.emissive, V3(s.x, s.y, s.z));
    }
    if (m->m_TextureFlags & HRPT_TEXFLAG_NORMAL) {
        v4 s = sample_texture(c, m->m_NormalTextureIndex, m->m_NormalSamplerIndex, a->uv);
        p.normal = normal_with_tbn(s.x, s.y, a->worldNormal, a->worldTangent, a->tangentSign);
    } else p.normal = normalize3(a->worldNormal);
    return p;
}

/* PathTracer_CSMain :53-340 for one pixel */
static void trace_pixel(Tls* tl, const HrptPathTracerConstants* cb, uint32_t px, uint32_t py, float* accumulation, float* output, uint32_t W)
{
    const OrContext* c = tl->c;
    tl->st.paths++;
    /* primary ray :61-72 */
    float u = (((float)px + 0.5f) + cb->m_Jitter[0]) * cb->m_View.m_ViewportSizeInv[0];
    float v = (((float)py + 0.5f) + cb->m_Jitter[1]) * cb->m_View.m_ViewportSizeInv[1];
    float cx = u * 2.0f + -1.0f, cy = v * -2.0f + 1.0f;                       /* UVToClipXY, Common.hlsli:50-53 */
    v4 farp = mul_v4_m(cx, cy, 0.9f, 1.0f, cb->m_View.m_MatClipToWorldNoOffset);
    v3 end = V3(farp.x / farp.w, farp.y / farp.w, farp.z / farp.w);
    Ray ray;
    ray.o = V3(cb->m_CameraPos[0], cb->m_CameraPos[1], cb->m_CameraPos[2]);
    ray.d = normalize3(sub3(end, ray.o));
    ray.tmin = 0.0f; ray.tmax = 1e10f;

    uint32_t rng = hrt_rng_seed(px, py, cb->m_AccumulationIndex);
    v3 throughput = V3(1, 1, 1), radianceAcc = V3(0, 0, 0);
    int inVolume = 0; float interiorIOR = 1.0f; v3 sigA = V3(0, 0, 0), sigS = V3(0, 0, 0);
    v3 sunDir = V3(cb->m_SunDirection[0], cb->m_SunDirection[1], cb->m_SunDirection[2]);
    float sunIntensity = c->lights[0].m_Intensity;   /* g_Lights[0], :137,:323 (quirk kept) */
    int maxBounces = (int)cb->m_MaxBounces;

    for (int bounce = 0; bounce < maxBounces; ++bounce) {
        Hit hit;
        int didHit = trace_ray_standard(tl, &ray, &rng, &hit);
        if (didHit) {
            const HrptPerInstanceData* inst = &c->instances[hit.inst];
            const HrptMeshData* mesh = &c->meshData[inst->m_MeshDataIndex];
            const HrptMaterialConstants* mat = &c->materials[inst->m_MaterialIndex];
            if (inVolume) throughput = mul3(throughput, eval_transmittance(sigA, sigS, hit.t));      /* :97-100 */
            FullHitAttributes attr = full_hit_attributes(c, &hit, &ray, inst, 0, mesh);              /* :103-104 LOD 0 */
            PBRAttributes pbr = pbr_attributes(c, &attr, mat);
            v3 p_atmo = atm_pos(attr.worldPos);
            v3 Ng = normalize3(attr.worldNormal);
            v3 N = pbr.normal;
            v3 V = neg3(ray.d);
            int isFrontFace = dot3(Ng, ray.d) < 0.0f;
            if (dot3(N, V) < 0.0f) N = neg3(N);

            LightingInputs in; memset(&in, 0, sizeof in);
            in.N = N; in.V = V; in.L = V3(0, 0, 0); in.worldPos = attr.worldPos; in.baseColor = pbr.baseColor;
            in.roughness = pbr.roughness; in.metallic = pbr.metallic; in.ior = mat->m_IOR;
            in.sunRadiance = atm_sun_radiance(c, p_atmo, sunDir, sunIntensity);
            in.sunDirection = sunDir;
            prepare_byproducts(&in);

            if (mat->m_TransmissionFactor > 0.0f || mat->m_AlphaMode == HRPT_ALPHA_MODE_BLEND) {       /* :149-255 */
                float effectiveAlpha = (mat->m_AlphaMode == HRPT_ALPHA_MODE_BLEND) ? pbr.alpha : 1.0f;
                float transmissionFactor = hrt_max(mat->m_TransmissionFactor, 1.0f - effectiveAlpha);
                float materialIOR = hrt_max(mat->m_IOR, 1.0001f);
                float outsideIOR = inVolume ? interiorIOR : 1.0f;
                float etaSurface = isFrontFace ? (outsideIOR / materialIOR) : (materialIOR / outsideIOR);
                float etaFresnel = etaSurface;
                float etaRefract = (mat->m_IsThinSurface != 0) ? 1.0f : etaFresnel;
                float cosT_geo;
                float F = fresnel_dielectric(etaFresnel, hrt_max(dot3(N, V), 0.0f), &cosT_geo);
                float probT = hrt_saturate((1.0f - F) * transmissionFactor);
                if (hrt_rng_next(&rng) < probT) {
                    v3 refractedDir, bsdfWeight;
                    if (pbr.roughness <= 0.08f) {
                        refractedDir = refract3(ray.d, N, etaRefract);
                        if (dot3(refractedDir, refractedDir) < 1e-8f) refractedDir = reflect3(ray.d, N);
                        bsdfWeight = pbr.baseColor;
                    } else {
                        float ux = hrt_rng_next(&rng), uy = hrt_rng_next(&rng);
                        v3 H = sample_ggx_vndf(ux, uy, N, V, pbr.roughness);
                        float VdotH = hrt_saturate(dot3(V, H));
                        float cosT_mf;
                        float F_mf = fresnel_dielectric(etaFresnel, VdotH, &cosT_mf);
                        float cosT_dir;
                        fresnel_dielectric(etaRefract, VdotH, &cosT_dir);
                        refractedDir = sub3(scale3(H, etaRefract * VdotH - cosT_dir), scale3(V, etaRefract));
                        if (dot3(refractedDir, refractedDir) < 1e-8f) refractedDir = reflect3(ray.d, H);
                        refractedDir = normalize3(refractedDir);
                        float alpha = pbr.roughness * pbr.roughness, alpha2 = alpha * alpha;
                        float NdotL_t = hrt_abs(dot3(N, refractedDir));
                        float G1_t = (NdotL_t > HRT_K_EPSILON) ? 2.0f * NdotL_t / (NdotL_t + hrt_sqrt(alpha2 + (1.0f - alpha2) * NdotL_t * NdotL_t)) : 0.0f;
                        bsdfWeight = scale3(scale3(scale3(pbr.baseColor, 1.0f - F_mf), G1_t), NdotL_t);
                    }
                    throughput = mul3(throughput, bsdfWeight);
                    if (mat->m_IsThinSurface == 0) {
                        if (isFrontFace) {
                            inVolume = 1; interiorIOR = materialIOR;
                            sigA = V3(mat->m_SigmaA[0], mat->m_SigmaA[1], mat->m_SigmaA[2]);
                            sigS = V3(mat->m_SigmaS[0], mat->m_SigmaS[1], mat->m_SigmaS[2]);
                        } else { inVolume = 0; interiorIOR = 1.0f; sigA = V3(0, 0, 0); sigS = V3(0, 0, 0); }
                    }
                    ray.o = sub3(attr.worldPos, scale3(N, 0.001f));
                    ray.d = normalize3(refractedDir);
                    ray.tmin = 1e-4f; ray.tmax = 1e10f;
                    continue;
                }
            }
            radianceAcc = add3(radianceAcc, mul3(throughput, pbr.emissive));                         /* :258 */
            LightingComponents direct = accumulate_direct(tl, &in, cb->m_LightCount, cb->m_CosSunAngularRadius, &rng);
            v3 dsum = add3(direct.diffuse, (bounce == 0) ? direct.specular : V3(0, 0, 0));
            radianceAcc = add3(radianceAcc, mul3(throughput, dsum));                                 /* :261 */
            if (bounce >= 2) {                                                                         /* :264-270 */
                float continuePr = hrt_saturate(hrt_max(throughput.x, hrt_max(throughput.y, throughput.z)));
                if (hrt_rng_next(&rng) > continuePr) break;
                throughput = div3s(throughput, continuePr);
            }
            float specProb = hrt_clamp(lerp1(in.F.x * 0.5f + 0.5f * pbr.metallic, 1.0f, pbr.metallic), 0.1f, 0.9f);   /* :275 */
            v3 newDir, brdfWeight;
            if (hrt_rng_next(&rng) < specProb) {
                float ux = hrt_rng_next(&rng), uy = hrt_rng_next(&rng);
                v3 H = sample_ggx_vndf(ux, uy, N, V, pbr.roughness);
                newDir = reflect3(neg3(V), H);
                if (dot3(N, newDir) <= 0.0f) break;
                brdfWeight = div3s(eval_ggx_vndf_weight(in.F0, N, V, newDir, H, pbr.roughness), specProb);
            } else {
                float ux = hrt_rng_next(&rng), uy = hrt_rng_next(&rng);
                newDir = sample_hemisphere_cosine(ux, uy, N);
                if (dot3(N, newDir) <= 0.0f) break;
                brdfWeight = div3s(scale3(pbr.baseColor, 1.0f - pbr.metallic), 1.0f - specProb);
            }
            throughput = mul3(throughput, brdfWeight);
            if (hrt_max(throughput.x, hrt_max(throughput.y, throughput.z)) < 0.01f) break;            /* :306 */
            ray.o = attr.worldPos; ray.d = newDir; ray.tmin = 1e-4f; ray.tmax = 1e10f;                /* :310-313 */
        } else {
            v3 sky = atm_sky_radiance(c, ray.o, ray.d, sunDir, sunIntensity, bounce == 0);            /* :315-328 */
            radianceAcc = add3(radianceAcc, mul3(throughput, sky));
            break;
        }
    }
    /* accumulate + resolve :332-339 */
    float* A = accumulation + ((size_t)py * W + px) * 4;
    float* O = output + ((size_t)py * W + px) * 4;
    v4 accum = { radianceAcc.x, radianceAcc.y, radianceAcc.z, 1.0f };
    if (cb->m_AccumulationIndex > 0) { accum.x += A[0]; accum.y += A[1]; accum.z += A[2]; accum.w += A[3]; }
    A[0] = accum.x; A[1] = accum.y; A[2] = accum.z; A[3] = accum.w;
    O[0] = accum.x / accum.w; O[1] = accum.y / accum.w; O[2] = accum.z / accum.w; O[3] = 1.0f;
}

/* ------------------------------------------------------------------ threaded dispatch */
typedef struct {
    OrContext* c; const HrptPathTracerConstants* cb; float* acc; float* out;
    uint32_t x0, y0, x1, y1, W; int brute; volatile uint32_t* nextRow; OrStats st;
} Job;
static void* worker(void* p)
{
    Job* j = (Job*)p;
    Tls tl; memset(&tl, 0, sizeof tl); tl.c = j->c; tl.brute = j->brute;
    for (;;) {
        uint32_t y = __sync_fetch_and_add(j->nextRow, 1u);
        if (y >= j->y1) break;
        for (uint32_t x = j->x0; x < j->x1; x++) trace_pixel(&tl, j->cb, x, y, j->acc, j->out, j->W);
    }
    j->st = tl.st;
    return NULL;
}
int or_render(OrContext* c, const HrptPathTracerConstants* cb, float* accumulation, float* output,
              uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, int nthreads, int bruteForce, OrStats* stats)
{
    if (!c || !cb || !accumulation || !output) return HRPT_ERR_INVALID_ARGUMENT;
    uint32_t W = (uint32_t)cb->m_View.m_ViewportSize[0], H = (uint32_t)cb->m_View.m_ViewportSize[1];
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = W; y1 = H; }
    if (x1 > W || y1 > H || x0 > x1 || y0 > y1) return HRPT_ERR_INVALID_ARGUMENT;
    if (nthreads <= 0) nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads > 256) nthreads = 256;
    if (nthreads < 1) nthreads = 1;
    volatile uint32_t nextRow = y0;
    Job* jobs = (Job*)calloc((size_t)nthreads, sizeof(Job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
        Job* j = &jobs[i]; j->c = c; j->cb = cb; j->acc = accumulation; j->out = output;
        j->x0 = x0; j->y0 = y0; j->x1 = x1; j->y1 = y1; j->W = W; j->brute = bruteForce; j->nextRow = &nextRow;
        if (i + 1 < nthreads) pthread_create(&th[i], NULL, worker, j);
    }
    worker(&jobs[nthreads - 1]);
    for (int i = 0; i + 1 < nthreads; i++) pthread_join(th[i], NULL);
    if (stats) for (int i = 0; i < nthreads; i++) {
        stats->closestRays += jobs[i].st.closestRays; stats->shadowRays += jobs[i].st.shadowRays; stats->paths += jobs[i].st.paths;
        stats->closestNodes += jobs[i].st.closestNodes; stats->closestTris += jobs[i].st.closestTris;
        stats->shadowNodes += jobs[i].st.shadowNodes; stats->shadowTris += jobs[i].st.shadowTris; stats->retraces += jobs[i].st.retraces;
    }
    free(jobs); free(th);
    return HRPT_OK;
}

/* ------------------------------------------------------------------ host logic */
float or_halton(uint32_t index, uint32_t base)   /* src/Utilities.cpp:67-79 */
{
    float result = 0.0f, f = 1.0f / (float)base; uint32_t i = index;
    while (i > 0) { result += f * (float)(i % base); i /= base; f /= (float)base; }
    return result;
}
/* src/PathTracerRenderer.cpp:58-75. cosf() of the host libm is replaced by the contract cosine. */
void or_fill_constants(HrptPathTracerConstants* cb, const HrptPlanarViewConstants* view, const float cameraPos[3],
                       uint32_t lightCount, uint32_t accumulationIndex, uint32_t frameIndex, uint32_t maxBounces,
                       const float sunDirection[3], float sunAngularSizeDeg)
{
    memset(cb, 0, sizeof *cb);
    cb->m_View = *view;
    cb->m_CameraPos[0] = cameraPos[0]; cb->m_CameraPos[1] = cameraPos[1]; cb->m_CameraPos[2] = cameraPos[2]; cb->m_CameraPos[3] = 1.0f;
    cb->m_LightCount = lightCount; cb->m_AccumulationIndex = accumulationIndex; cb->m_FrameIndex = frameIndex; cb->m_MaxBounces = maxBounces;
    cb->m_Jitter[0] = or_halton(accumulationIndex + 1, 2) - 0.5f;
    cb->m_Jitter[1] = or_halton(accumulationIndex + 1, 3) - 0.5f;
    cb->m_SunDirection[0] = sunDirection[0]; cb->m_SunDirection[1] = sunDirection[1]; cb->m_SunDirection[2] = sunDirection[2];
    float halfAngleRad = sunAngularSizeDeg * 0.5f * (3.141592654f / 180.0f);   /* DirectX::XM_PI */
    cb->m_CosSunAngularRadius = hrt_cos(halfAngleRad);
}

/* ------------------------------------------------------------------ probes */
uint32_t or_pcg_hash(uint32_t v) { return hrt_pcg_hash(v); }
uint32_t or_init_rng(uint32_t px, uint32_t py, uint32_t a) { return hrt_rng_seed(px, py, a); }
float or_next_float(uint32_t* s) { return hrt_rng_next(s); }
float or_sin(float x) { return hrt_sin(x); }
float or_cos(float x) { return hrt_cos(x); }
float or_exp(float x) { return hrt_exp(x); }
int or_trace_closest(OrContext* c, const float o[3], const float d[3], float tmin, float tmax, int brute,
                     uint32_t* inst, uint32_t* prim, float bary[2], float* t)
{
    Ray r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]); r.tmin = tmin; r.tmax = tmax;
    TravCount tc = { 0, 0 };
    Hit h = closest_any(c, &r, 0, 0, 0, 0, brute, &tc);
    if (!h.valid) return 0;
    *inst = h.inst; *prim = h.prim; bary[0] = h.u; bary[1] = h.v; *t = h.t;
    return 1;
}

/* TraceRayStandard (RaytracingCommon.hlsli:138-198) and CalculateRTShadow<true> (CommonLighting.hlsli:380-496) as stand-alone ray
 * queries: what the other inline-ray-tracing passes of the reference call. rng is the caller's RNG state (BLEND
 * candidates draw from it). Checkers for hrpt_trace_rays. */
int or_trace_standard(OrContext* c, const float o[3], const float d[3], float tmin, float tmax, uint32_t* rng,
                      uint32_t* inst, uint32_t* prim, float bary[2], float* t)
{
    Tls tl; memset(&tl, 0, sizeof tl); tl.c = c; tl.brute = 0;
    Ray r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]); r.tmin = tmin; r.tmax = tmax;
    Hit h;
    if (!trace_ray_standard(&tl, &r, rng, &h)) return 0;
    *inst = h.inst; *prim = h.prim; bary[0] = h.u; bary[1] = h.v; *t = h.t;
    return 1;
}
float or_shadow_query(OrContext* c, const float worldPos[3], const float L[3], float maxDist)
{
    Tls tl; memset(&tl, 0, sizeof tl); tl.c = c; tl.brute = 0;
    return calculate_rt_shadow(&tl, V3(worldPos[0], worldPos[1], worldPos[2]), V3(L[0], L[1], L[2]), maxDist);
}

/* ------------------------------------------------------------------ HDR post chain */
float or_log2(float x) { return hrt_log2(x); }
float or_exp2(float x) { return hrt_exp2(x); }
float or_pow(float x, float y) { return hrt_pow(x, y); }

/* PBRNeutralToneMapping, Tonemap.hlsl:13-33 */
static v3 pbr_neutral(v3 c)
{
    const float startCompression = 0.8f - 0.04f, desaturation = 0.15f;
    float x = hrt_min(c.x, hrt_min(c.y, c.z));
    float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    c = V3(c.x - offset, c.y - offset, c.z - offset);
    float peak = hrt_max(c.x, hrt_max(c.y, c.z));
    if (peak < startCompression) return c;
    const float d = 1.0f - startCompression;
    float newPeak = 1.0f - d * d / (peak + d - startCompression);
    float k = newPeak / peak;
    c = scale3(c, k);
    float g = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
    return V3(lerp1(c.x, newPeak * 1.0f, g), lerp1(c.y, newPeak * 1.0f, g), lerp1(c.z, newPeak * 1.0f, g));
}
/* sRGB_OETF, Tonemap.hlsl:35-42 */
static float srgb_oetf(float x)
{
    float v = (x <= 0.0031308f) ? x * 12.92f : 1.055f * hrt_pow(x, 1.0f / 2.4f) - 0.055f;
    return hrt_saturate(v);
}
/* HDRDisplayTonemap, Tonemap.hlsl:72-92 */
static v3 hdr_display_tonemap(v3 x, float maxNits)
{
    float maxSCRGB = maxNits / 80.0f;
    float lum = hrt_max(x.x, hrt_max(x.y, x.z));
    if (lum <= 1.0f) return x;
    float headroom = maxSCRGB - 1.0f, excess = lum - 1.0f;
    float compressed = excess * headroom / (excess + headroom);
    float newLum = 1.0f + compressed;
    float k = newLum / lum;
    return V3(hrt_min(x.x * k, maxSCRGB), hrt_min(x.y * k, maxSCRGB), hrt_min(x.z * k, maxSCRGB));
}
void or_post_process(const float* hdr, uint32_t W, uint32_t H, const HrptPostParams* p, float* exposure, uint32_t histogram[256], float* display)
{
    const float kMinLog = -10.0f, kMaxLog = 20.0f;   /* src/HDRRenderer.cpp:12-13 */
    uint32_t hist[256]; memset(hist, 0, sizeof hist);
    if (p->autoExposure) {
        /* LuminanceHistogram_CSMain */
        for (size_t i = 0; i < (size_t)W * H; i++) {
            v3 c = V3(hdr[i * 4], hdr[i * 4 + 1], hdr[i * 4 + 2]);
            float lum = dot3(c, V3(0.2126f, 0.7152f, 0.0722f));
            uint32_t bin = 0;
            if (!(lum < 0.0001f)) {
                float range = kMaxLog - kMinLog;
                float logLum = hrt_clamp((hrt_log2(lum) - kMinLog) / range, 0.0f, 1.0f);
                bin = (uint32_t)(logLum * 254.0f + 1.0f);
            }
            hist[bin]++;
        }
        /* ExposureAdaptation_CSMain: the 256-wide shared-memory tree, same order */
        float w[256]; float range = kMaxLog - kMinLog;
        for (uint32_t t = 0; t < 256; t++) {
            float logLum = t == 0 ? kMinLog : (kMinLog + ((float)(t - 1) / 254.0f) * range);
            w[t] = (float)hist[t] * logLum;
        }
        for (uint32_t i = 128; i > 0; i >>= 1) for (uint32_t t = 0; t < i; t++) w[t] += w[t + i];
        float avgLogLum = w[0] / hrt_max((float)(W * H), 1.0f);
        float avgLum = hrt_exp2(avgLogLum);
        float EV100 = hrt_log2(avgLum * 100.0f / 12.5f);
        EV100 = hrt_clamp(EV100, p->exposureValueMin, p->exposureValueMax);
        EV100 -= p->exposureCompensation;
        float target = 1.0f / (hrt_pow(2.0f, EV100) * 1.2f);
        float cur = *exposure;
        *exposure = cur + (target - cur) * (1.0f - hrt_exp(-p->deltaTimeSeconds * p->adaptationSpeed));
    } else *exposure = p->manualExposure;   /* writeBuffer(exposureBuffer, &m_Camera.m_Exposure), src/HDRRenderer.cpp:163 */
    if (histogram) memcpy(histogram, hist, sizeof hist);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        v3 c = scale3(V3(hdr[i * 4], hdr[i * 4 + 1], hdr[i * 4 + 2]), *exposure);
        v3 o;
        if (p->hdrDisplay) o = hdr_display_tonemap(c, p->maxDisplayNits);
        else { v3 t = pbr_neutral(c); o = V3(srgb_oetf(t.x), srgb_oetf(t.y), srgb_oetf(t.z)); }
        display[i * 4] = o.x; display[i * 4 + 1] = o.y; display[i * 4 + 2] = o.z; display[i * 4 + 3] = 1.0f;
    }
}
