// bvh_build.cpp -- binned-SAH BVH2 over world-space triangles; see bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <system_error>
#include <thread>
#include <atomic>
#include <limits>
#include <numeric>

namespace hrt {
namespace {

struct Prim {
    float bmin[3], bmax[3], c[3];
    uint32_t tri;
};
struct Box {
    float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
    void grow(const float* a, const float* b) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], a[k]); mx[k] = std::max(mx[k], b[k]); } }
    void grow(const Box& o) { grow(o.mn, o.mx); }
    float area() const
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Builder {
    uint32_t minLeaf = 2;       // ranges this small always become a leaf
    uint32_t maxLeaf = kMaxLeafTris;   // largest leaf the SAH termination may create (1 for the tree over instances)
    float triCost = 1.0f;       // SAH: cost of a triangle test relative to a node step
    std::vector<Prim> prims;
    const std::vector<HostTri>* src = nullptr;
    BuiltBvh* out = nullptr;

    static int32_t encode_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 2) | (count - 1)); }

    int32_t make_leaf(BuiltBvh& o, uint32_t first, uint32_t count)
    {
        uint32_t base = (uint32_t)o.tris.size();
        for (uint32_t i = 0; i < count; ++i) o.tris.push_back((*src)[prims[first + i].tri]);
        return encode_leaf(base, count);
    }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth, Box& bounds) { return build(*out, first, count, depth, bounds, kParallelLevels); }

    // returns child reference (>= 0 inner node, < 0 leaf) and its padded bounds. `forks`: levels below which the right subtree may still be built
    // by another thread (into its own arrays, spliced behind the left subtree: nodes and triangles end up exactly where the sequential build puts
    // them -- depth-first order -- so the tree does not depend on the thread count).
    static constexpr int kParallelLevels = 4;              // up to 16 threads
    static constexpr uint32_t kParallelMin = 8192;         // primitives in a range worth a thread
    int32_t build(BuiltBvh& o, uint32_t first, uint32_t count, uint32_t depth, Box& bounds, int forks)
    {
        o.maxDepth = std::max(o.maxDepth, depth);
        Box b, cb;
        for (uint32_t i = first; i < first + count; ++i) { b.grow(prims[i].bmin, prims[i].bmax); cb.grow(prims[i].c, prims[i].c); }
        bounds = b;
        if (count <= minLeaf) return make_leaf(o, first, count);

        uint32_t log2c = 0; while ((1u << log2c) < count) ++log2c;
        bool forceMedian = depth + log2c + 2 >= kHostBuilderDepthGoal;

        int bestAxis = -1; uint32_t bestSplit = 0; float bestCost = 1e30f;
        constexpr int NB = 16;
        float leafCost = triCost * (float)count * b.area();
        if (!forceMedian) {
            for (int axis = 0; axis < 3; ++axis) {
                float lo = cb.mn[axis], ext = cb.mx[axis] - cb.mn[axis];
                if (!(ext > 0.0f)) continue;
                Box bb[NB]; uint32_t bc[NB] = {};
                for (uint32_t i = first; i < first + count; ++i) {
                    int bi = std::min(NB - 1, (int)((prims[i].c[axis] - lo) / ext * NB));
                    bb[bi].grow(prims[i].bmin, prims[i].bmax); bc[bi]++;
                }
                float rarea[NB]; uint32_t rcount[NB]; Box acc; uint32_t n = 0;
                for (int i = NB - 1; i > 0; --i) { acc.grow(bb[i]); n += bc[i]; rarea[i] = acc.area(); rcount[i] = n; }
                Box lacc; uint32_t ln = 0;
                for (int i = 0; i < NB - 1; ++i) {
                    lacc.grow(bb[i]); ln += bc[i];
                    if (ln == 0 || rcount[i + 1] == 0) continue;
                    float cost = 1.0f * b.area() + lacc.area() * (float)ln + rarea[i + 1] * (float)rcount[i + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = (uint32_t)i; }
                }
            }
        }
        if (!forceMedian && count <= maxLeaf && (bestAxis < 0 || bestCost >= leafCost)) return make_leaf(o, first, count);

        uint32_t mid;
        if (bestAxis >= 0 && !forceMedian) {
            float lo = cb.mn[bestAxis], ext = cb.mx[bestAxis] - cb.mn[bestAxis];
            auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim& p) {
                int bi = std::min(NB - 1, (int)((p.c[bestAxis] - lo) / ext * NB));
                return (uint32_t)bi <= bestSplit;
            });
            mid = (uint32_t)(it - prims.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        } else {
            int axis = 0; float e = cb.mx[0] - cb.mn[0];
            if (cb.mx[1] - cb.mn[1] > e) { axis = 1; e = cb.mx[1] - cb.mn[1]; }
            if (cb.mx[2] - cb.mn[2] > e) axis = 2;
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count, [axis](const Prim& a, const Prim& c) {
                if (a.c[axis] != c.c[axis]) return a.c[axis] < c.c[axis];
                return a.tri < c.tri;
            });
        }
        int32_t id = (int32_t)o.nodes.size();
        o.nodes.emplace_back();
        Box lb, rb;
        int32_t l, r;
        if (forks > 0 && count >= kParallelMin) {
            BuiltBvh ro;
            std::future<int32_t> right;
            try { right = std::async(std::launch::async, [&]() { return build(ro, mid, first + count - mid, depth + 1, rb, forks - 1); }); }
            catch (const std::system_error&) {}                     // no thread to be had: this one builds both halves
            l = build(o, first, mid - first, depth + 1, lb, forks - 1);
            r = right.valid() ? right.get() : build(ro, mid, first + count - mid, depth + 1, rb, 0);
            const int32_t nodeOff = (int32_t)o.nodes.size(); const uint32_t triOff = (uint32_t)o.tris.size();
            auto moved = [&](int32_t ref) { if (ref >= 0) return ref + nodeOff; const uint32_t enc = (uint32_t)~ref; return ~(int32_t)((((enc >> 2) + triOff) << 2) | (enc & 3u)); };
            for (HostNode& rn : ro.nodes) { rn.left = moved(rn.left); rn.right = moved(rn.right); }
            o.nodes.insert(o.nodes.end(), ro.nodes.begin(), ro.nodes.end());
            o.tris.insert(o.tris.end(), ro.tris.begin(), ro.tris.end());
            o.maxDepth = std::max(o.maxDepth, ro.maxDepth);
            r = moved(r);
        } else {
            l = build(o, first, mid - first, depth + 1, lb, 0);
            r = build(o, mid, first + count - mid, depth + 1, rb, 0);
        }
        HostNode& n = o.nodes[id];
        for (int k = 0; k < 3; ++k) { n.lmin[k] = lb.mn[k]; n.lmax[k] = lb.mx[k]; n.rmin[k] = rb.mn[k]; n.rmax[k] = rb.mx[k]; }
        n.left = l; n.right = r; n.pad0 = 0; n.pad1 = 0;
        return id;
    }
};

// mul(float4(p,1), M).xyz in the row-vector convention of Common.hlsli:18-21, left-to-right, no FMA.
inline void transform_point(const float* p, const float* M, float* o)
{
    o[0] = ((p[0] * M[0] + p[1] * M[4]) + p[2] * M[8]) + M[12];
    o[1] = ((p[0] * M[1] + p[1] * M[5]) + p[2] * M[9]) + M[13];
    o[2] = ((p[0] * M[2] + p[1] * M[6]) + p[2] * M[10]) + M[14];
}

// ---- BVH2 -> BVH4 collapse: repeatedly replace the inner child with the largest box by its two children
struct ChildRef { int32_t ref; float mn[3], mx[3]; };
inline float box_area(const ChildRef& c)
{
    float dx = c.mx[0] - c.mn[0], dy = c.mx[1] - c.mn[1], dz = c.mx[2] - c.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
inline void children_of(const HostNode& n, ChildRef& l, ChildRef& r)
{
    l.ref = n.left; r.ref = n.right;
    for (int k = 0; k < 3; ++k) { l.mn[k] = n.lmin[k]; l.mx[k] = n.lmax[k]; r.mn[k] = n.rmin[k]; r.mx[k] = n.rmax[k]; }
}
int32_t collapse4(const std::vector<HostNode>& n2, int32_t root2, std::vector<HostNode4>& out, uint32_t depth, uint32_t& maxDepth)
{
    maxDepth = std::max(maxDepth, depth);
    ChildRef c[4]; int count = 2;
    children_of(n2[(size_t)root2], c[0], c[1]);
    while (count < 4) {
        int best = -1; float bestArea = -1.0f;
        for (int i = 0; i < count; ++i) if (c[i].ref >= 0) { float a = box_area(c[i]); if (a > bestArea) { bestArea = a; best = i; } }
        if (best < 0) break;
        ChildRef l, r; children_of(n2[(size_t)c[best].ref], l, r);
        c[best] = l; c[count++] = r;
    }
    int32_t id = (int32_t)out.size();
    out.emplace_back();
    int32_t refs[4];
    for (int i = 0; i < count; ++i) refs[i] = c[i].ref >= 0 ? collapse4(n2, c[i].ref, out, depth + 1, maxDepth) : c[i].ref;
    HostNode4& n = out[(size_t)id];
    // an unused slot holds a degenerate box far outside any ray interval (|inv| >= 1 and tmax = 1e10), so the slab test
    // rejects it without a special case
    const float far = 1e30f;
    for (int i = 0; i < 4; ++i) {
        bool used = i < count;
        n.minx[i] = used ? c[i].mn[0] : far; n.miny[i] = used ? c[i].mn[1] : far; n.minz[i] = used ? c[i].mn[2] : far;
        n.maxx[i] = used ? c[i].mx[0] : far; n.maxy[i] = used ? c[i].mx[1] : far; n.maxz[i] = used ? c[i].mx[2] : far;
        n.child[i] = used ? refs[i] : kEmptyChild;
        n.pad[i] = 0;
    }
    return id;
}

inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float* a, const float* b, float* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
inline float half_to_float(uint32_t h)
{
    uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    float f;
    if (e == 0) {
        if (m == 0) { u = s; memcpy(&f, &u, 4); return f; }
        u = 0x33800000u; float scale; memcpy(&scale, &u, 4);
        f = (float)m * scale;
        return s ? -f : f;
    }
    u = (e == 31) ? (s | 0x7f800000u | (m << 13)) : (s | ((e + 112u) << 23) | (m << 13));
    memcpy(&f, &u, 4);
    return f;
}
// UnpackVertex, MeshCommon.hlsli:9-22 (normal, uv) and DecodeOct, Common.hlsli:174-181 (tangent)
inline void unpack_normal(const HrptVertexQuantized& q, float* n)
{
    n[0] = (float)(q.m_Normal & 1023u) / 511.0f - 1.0f;
    n[1] = (float)((q.m_Normal >> 10) & 1023u) / 511.0f - 1.0f;
    n[2] = (float)((q.m_Normal >> 20) & 1023u) / 511.0f - 1.0f;
}
inline void unpack_uv(const HrptVertexQuantized& q, float* uv) { uv[0] = half_to_float(q.m_Uv & 0xFFFFu); uv[1] = half_to_float(q.m_Uv >> 16); }
inline void unpack_tangent(const HrptVertexQuantized& q, float* t)
{
    float ex = (float)(q.m_Tangent & 255u) / 127.0f - 1.0f, ey = (float)((q.m_Tangent >> 8) & 255u) / 127.0f - 1.0f;
    float v[3] = { ex, ey, (1.0f - std::fabs(ex)) - std::fabs(ey) };
    float neg = -v[2];
    float tt = (neg >= 0.0f) ? neg : 0.0f;      // max(-v.z, 0)
    v[0] += (v[0] >= 0.0f) ? -tt : tt;
    v[1] += (v[1] >= 0.0f) ? -tt : tt;
    float inv = 1.0f / std::sqrt(dot3(v, v));   // normalize(v) = v * (1/sqrt(dot(v,v)))
    t[0] = v[0] * inv; t[1] = v[1] * inv; t[2] = v[2] * inv;
    t[3] = (q.m_Normal & (1u << 30)) != 0 ? -1.0f : 1.0f;
}

} // namespace

void collapse_bvh2_on_host(const std::vector<HostNode>& nodes2, std::vector<HostNode4>& nodes4, uint32_t& maxDepth4)
{
    nodes4.clear(); nodes4.reserve(nodes2.size() / 2 + 1); maxDepth4 = 0;
    collapse4(nodes2, 0, nodes4, 0, maxDepth4);
}

// The largest side of a triangle's bounding box. The watertight test computes t from vertex coordinates relative to the ray origin: for a hit
// next to the origin (t ~ tmin) its rounding error is a few ulp of the triangle's SIZE, whatever the size of t -- measured: a 460-unit floor
// quad reports t = 1.01e-4 for a true 0.89e-4. A box padded by its own coordinates alone (1e-6 around a floor at y = 0) is then tighter than
// the test it guards, and which hits survive starts to depend on the boxes (fp32 against quantised nodes: 53 of 2.5 M surface-leaving rays).
// Hence 1e-6 (8 ulp) of this on every side, on top of the coordinate term.
static inline float triangle_extent(const HostTri& t)
{
    float e = 0.0f;
    for (int k = 0; k < 3; ++k) e = std::max(e, std::max(t.p0[k], std::max(t.p1[k], t.p2[k])) - std::min(t.p0[k], std::min(t.p1[k], t.p2[k])));
    return e;
}

bool validate_scene(const HrptSceneDesc& s, uint64_t& triCount, std::string& error, bool flatLimit)
{
    triCount = 0;
    if ((!s.vertices && s.vertexCount) || (!s.indices && s.indexCount) || (!s.meshData && s.meshDataCount) || (!s.instances && s.instanceCount) ||
        (!s.materials && s.materialCount) || !s.lights) { error = "null scene array with a non-zero count"; return false; }
    if (s.lightCount == 0) { error = "scene needs at least one light (the reference guarantees a directional light, src/Scene.cpp:635-666)"; return false; }
    for (uint32_t i = 0; i < s.indexCount; ++i)
        if (s.indices[i] >= s.vertexCount) { error = "index buffer references a vertex out of range"; return false; }
    for (uint32_t i = 0; i < s.instanceCount; ++i) {
        const HrptPerInstanceData& in = s.instances[i];
        if (in.m_MeshDataIndex >= s.meshDataCount) { error = "instance m_MeshDataIndex out of range"; return false; }
        if (in.m_MaterialIndex >= s.materialCount) { error = "instance m_MaterialIndex out of range"; return false; }
        if (in.m_LODIndex != 0) { error = "instance m_LODIndex must be 0 on the path-tracer path (TLASPatch does not run, PathTracer.hlsl:102)"; return false; }
        const HrptMeshData& md = s.meshData[in.m_MeshDataIndex];
        if ((uint64_t)md.m_IndexOffsets[0] + md.m_IndexCounts[0] > s.indexCount || md.m_IndexCounts[0] % 3 != 0) { error = "mesh LOD0 index range invalid"; return false; }
        triCount += md.m_IndexCounts[0] / 3;
    }
    if (flatLimit && triCount >= kMaxStructureTriangles) { error = "too many triangles for the flat structure (2^32 / 48)"; return false; }
    return true;
}

// per-instance adjugate rows (TransformNormal, Common.hlsli:33-47)
static inline void instance_shade(const float* M, HostInstShade& is)
{
    const float r0[3] = { M[0], M[1], M[2] }, r1[3] = { M[4], M[5], M[6] }, r2[3] = { M[8], M[9], M[10] };
    cross3(r1, r2, is.adj0); cross3(r2, r0, is.adj1); cross3(r0, r1, is.adj2);
    is.adj0[3] = is.adj1[3] = is.adj2[3] = 0.0f;
}
void build_instance_shade(const HrptSceneDesc& s, std::vector<HostInstShade>& out)
{
    out.resize(s.instanceCount);
    for (uint32_t i = 0; i < s.instanceCount; ++i) instance_shade(s.instances[i].m_World, out[i]);
}

bool scene_needs_tangents(const HrptSceneDesc& s)
{
    for (uint32_t m = 0; m < s.materialCount; ++m) if (s.materials[m].m_TextureFlags & HRPT_TEXFLAG_NORMAL) return true;
    return false;
}

bool build_scene_bvh(const HrptSceneDesc& s, BuiltBvh& out, std::string& error)
{
    out = BuiltBvh();
    uint64_t triCount = 0;
    if (!validate_scene(s, triCount, error)) return false;

    std::vector<HostTri> tris; tris.reserve((size_t)triCount);
    for (uint32_t i = 0; i < s.instanceCount; ++i) {
        const HrptPerInstanceData& in = s.instances[i];
        const HrptMeshData& md = s.meshData[in.m_MeshDataIndex];
        uint32_t opaque = triangle_flags_for_material(s.materials[in.m_MaterialIndex]);     // bit 0 opaque, bits 1-2 shading class
        for (uint32_t p = 0; p < md.m_IndexCounts[0] / 3; ++p) {
            const uint32_t* ix = s.indices + md.m_IndexOffsets[0] + 3 * (size_t)p;
            HostTri t;
            transform_point(s.vertices[ix[0]].m_Pos, in.m_World, t.p0);
            transform_point(s.vertices[ix[1]].m_Pos, in.m_World, t.p1);
            transform_point(s.vertices[ix[2]].m_Pos, in.m_World, t.p2);
            t.inst = i; t.prim = p; t.flags = opaque;
            tris.push_back(t);
        }
    }
    build_instance_shade(s, out.instShade);
    if (tris.empty()) return true;

    Builder b; b.src = &tris; b.out = &out;
    if (const char* e = getenv("HRPT_HOST_BVH_MIN_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 4) b.minLeaf = (uint32_t)v; }
    if (const char* e = getenv("HRPT_HOST_BVH_TRI_COST")) { float v = (float)atof(e); if (v > 0.0f) b.triCost = v; }
    b.prims.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) {
        Prim& p = b.prims[i]; const HostTri& t = tris[i];
        const float extPad = 1e-6f * triangle_extent(t);
        for (int k = 0; k < 3; ++k) {
            float mn = std::min(t.p0[k], std::min(t.p1[k], t.p2[k])), mx = std::max(t.p0[k], std::max(t.p1[k], t.p2[k]));
            if (!(mn == mn) || !(mx == mx) || std::isinf(mn) || std::isinf(mx)) { error = "non-finite vertex position"; return false; }
            // conservative padding: the fp32 watertight test can accept points a few ulp outside the triangle (see triangle_extent)
            float pad = 1e-5f * std::max(std::fabs(mn), std::fabs(mx)) + 1e-6f + extPad;
            p.bmin[k] = mn - pad; p.bmax[k] = mx + pad; p.c[k] = 0.5f * mn + 0.5f * mx;
        }
        p.tri = (uint32_t)i;
    }
    out.tris.reserve(tris.size());
    Box root;
    int32_t r = b.build(0, (uint32_t)tris.size(), 0, root);
    if (r < 0) { out.rootLeaf = r; out.nodes.clear(); }
    else {
        out.nodes4.reserve(out.nodes.size() / 2 + 1); collapse4(out.nodes, 0, out.nodes4, 0, out.maxDepth4);
        double sum = 0.0;
        auto area = [](const float* mn, const float* mx) { double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return dx * dy + dy * dz + dz * dx; };
        for (const HostNode& n : out.nodes)
            sum += area(n.lmin, n.lmax) * (n.left >= 0 ? 1.0 : (double)(((uint32_t)~n.left & 3u) + 1u)) + area(n.rmin, n.rmax) * (n.right >= 0 ? 1.0 : (double)(((uint32_t)~n.right & 3u) + 1u));
        double ra = root.area() * 0.5;
        out.sahCost = ra > 0.0 ? (float)(1.0 + sum / ra) : 0.0f;
    }

    // shading attributes in leaf order
    const bool needTangents = scene_needs_tangents(s);
    out.attrs.resize(out.tris.size());
    if (needTangents) out.tangents.resize(out.tris.size());
    for (size_t k = 0; k < out.tris.size(); ++k) {
        const HostTri& t = out.tris[k];
        const HrptPerInstanceData& in = s.instances[t.inst];
        const HrptMeshData& md = s.meshData[in.m_MeshDataIndex];
        const uint32_t* ix = s.indices + md.m_IndexOffsets[0] + 3 * (size_t)t.prim;
        HostTriAttr& a = out.attrs[k];
        unpack_normal(s.vertices[ix[0]], a.n0); unpack_normal(s.vertices[ix[1]], a.n1); unpack_normal(s.vertices[ix[2]], a.n2);
        unpack_uv(s.vertices[ix[0]], a.uv0); unpack_uv(s.vertices[ix[1]], a.uv1); unpack_uv(s.vertices[ix[2]], a.uv2);
        a.material = in.m_MaterialIndex; a.inst = t.inst; a.prim = t.prim; a.pad[0] = a.pad[1] = 0;
        if (needTangents) {
            HostTriTangent& tg = out.tangents[k];
            unpack_tangent(s.vertices[ix[0]], tg.t0); unpack_tangent(s.vertices[ix[1]], tg.t1); unpack_tangent(s.vertices[ix[2]], tg.t2);
        }
    }
    return true;
}

// ---------------------------------------------------------------- two-level structure (instanced scenes)
namespace {
// inverse of the affine map p' = p * M (row-vector, translation in row 3), computed in binary64 and rounded once
bool invert_affine(const float* M, float* inv12)
{
    const double a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double r[9] = { (e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                          (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                          (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det };
    const double t[3] = { M[12], M[13], M[14] };
    for (int k = 0; k < 9; ++k) inv12[k] = (float)r[k];
    for (int k = 0; k < 3; ++k) inv12[9 + k] = (float)-(t[0] * r[0 + k] + t[1] * r[3 + k] + t[2] * r[6 + k]);
    for (int k = 0; k < 12; ++k) if (!std::isfinite(inv12[k])) return false;
    return true;
}

// trees of the distinct meshes: one binned-SAH tree per MeshData entry that some instance uses
bool build_mesh_trees(const HrptSceneDesc& s, BuiltTwoLevel& out, std::vector<int32_t>& meshRoot, std::string& error)
{
    meshRoot.assign(s.meshDataCount, 0);
    std::vector<uint8_t> used(s.meshDataCount, 0);
    for (uint32_t i = 0; i < s.instanceCount; ++i) used[s.instances[i].m_MeshDataIndex] = 1;
    const bool needTangents = scene_needs_tangents(s);
    for (uint32_t m = 0; m < s.meshDataCount; ++m) {
        if (!used[m]) continue;
        ++out.distinctMeshes;
        const HrptMeshData& md = s.meshData[m];
        const uint32_t nt = md.m_IndexCounts[0] / 3;
        if (nt == 0) { meshRoot[m] = (int32_t)0x80000000; continue; }      // kTraversalDone: nothing to intersect
        std::vector<HostTri> tris(nt);
        for (uint32_t p = 0; p < nt; ++p) {
            const uint32_t* ix = s.indices + md.m_IndexOffsets[0] + 3 * (size_t)p;
            HostTri& t = tris[p];
            for (int k = 0; k < 3; ++k) { t.p0[k] = s.vertices[ix[0]].m_Pos[k]; t.p1[k] = s.vertices[ix[1]].m_Pos[k]; t.p2[k] = s.vertices[ix[2]].m_Pos[k]; }
            t.inst = m; t.prim = p; t.flags = 0;
        }
        BuiltBvh local;
        Builder b; b.src = &tris; b.out = &local;
        b.prims.resize(nt);
        for (uint32_t p = 0; p < nt; ++p) {
            Prim& pr = b.prims[p]; const HostTri& t = tris[p];
            const float extPad = 1e-6f * triangle_extent(t);
            for (int k = 0; k < 3; ++k) {
                float mn = std::min(t.p0[k], std::min(t.p1[k], t.p2[k])), mx = std::max(t.p0[k], std::max(t.p1[k], t.p2[k]));
                if (!(mn == mn) || !(mx == mx) || std::isinf(mn) || std::isinf(mx)) { error = "non-finite vertex position"; return false; }
                float pad = 1e-5f * std::max(std::fabs(mn), std::fabs(mx)) + 1e-6f + extPad;
                pr.bmin[k] = mn - pad; pr.bmax[k] = mx + pad; pr.c[k] = 0.5f * mn + 0.5f * mx;
            }
            pr.tri = p;
        }
        Box root;
        const int32_t r = b.build(0, nt, 0, root);
        const uint32_t triBase = (uint32_t)out.tris.size(), nodeBase = (uint32_t)out.nodes4.size();
        if ((uint64_t)triBase + nt >= kMaxStructureTriangles) { error = "too many distinct triangles for the two-level structure (2^32 / 48)"; return false; }
        auto fix_leaf = [&](int32_t ref) { const uint32_t enc = (uint32_t)~ref; return ~(int32_t)((((enc >> 2) + triBase) << 2) | (enc & 3u)); };
        if (r < 0) meshRoot[m] = fix_leaf(r);
        else {
            std::vector<HostNode4> n4; uint32_t d4 = 0;
            collapse4(local.nodes, 0, n4, 0, d4);
            out.maxDepth4Blas = std::max(out.maxDepth4Blas, d4 + 1);
            for (HostNode4& n : n4) for (int k = 0; k < 4; ++k) {
                if (n.child[k] == kEmptyChild) continue;
                n.child[k] = n.child[k] >= 0 ? n.child[k] + (int32_t)nodeBase : fix_leaf(n.child[k]);
            }
            out.nodes4.insert(out.nodes4.end(), n4.begin(), n4.end());
            meshRoot[m] = (int32_t)nodeBase;
        }
        // triangles + attributes of the mesh in leaf order
        for (const HostTri& t : local.tris) {
            out.tris.push_back(t);
            const uint32_t* ix = s.indices + md.m_IndexOffsets[0] + 3 * (size_t)t.prim;
            HostTriAttr a{};
            unpack_normal(s.vertices[ix[0]], a.n0); unpack_normal(s.vertices[ix[1]], a.n1); unpack_normal(s.vertices[ix[2]], a.n2);
            unpack_uv(s.vertices[ix[0]], a.uv0); unpack_uv(s.vertices[ix[1]], a.uv1); unpack_uv(s.vertices[ix[2]], a.uv2);
            a.material = 0; a.inst = 0; a.prim = t.prim;
            out.attrs.push_back(a);
            if (needTangents) {
                HostTriTangent tg;
                unpack_tangent(s.vertices[ix[0]], tg.t0); unpack_tangent(s.vertices[ix[1]], tg.t1); unpack_tangent(s.vertices[ix[2]], tg.t2);
                out.tangents.push_back(tg);
            }
        }
    }
    return true;
}
}

// [first, last) in chunks on up to 8 host threads (a rebuild per frame over 65 k instances is 4 ms on one thread); small ranges stay on the caller's
template <class F>
static void for_instance_ranges(uint32_t count, F&& body)
{
    unsigned threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (count < 8192u) threads = 1;
    if (threads == 1) { body(0u, count); return; }
    std::vector<std::thread> pool;
    const uint32_t chunk = (count + threads - 1) / threads;
    try {
        for (unsigned t = 1; t < threads; ++t) { const uint32_t f = std::min(count, t * chunk), l = std::min(count, f + chunk); if (f < l) pool.emplace_back([&body, f, l] { body(f, l); }); }
    } catch (const std::system_error&) { for (std::thread& th : pool) th.join(); body(0u, count); return; }       // no threads to be had: the whole range here (idempotent)
    body(0u, std::min(count, chunk));
    for (std::thread& th : pool) th.join();
}

bool rebuild_two_level_instances(const HrptSceneDesc& s, BuiltTwoLevel& out, std::string& error, std::vector<float>* worldBoxes)
{
    // mesh roots are recovered from the existing instance records (same mesh -> same root)
    std::vector<int32_t> meshRoot(s.meshDataCount, 0); std::vector<uint8_t> have(s.meshDataCount, 0);
    for (const HostInstance& hi : out.instances) if (hi.mesh < s.meshDataCount) { meshRoot[hi.mesh] = hi.blasRoot; have[hi.mesh] = 1; }
    for (uint32_t i = 0; i < s.instanceCount; ++i) if (!have[s.instances[i].m_MeshDataIndex]) { error = "instance uses a mesh without a tree"; return false; }
    // object-space bounds per mesh from its triangles (leaf-order array; inst = mesh)
    std::vector<Box> meshBox(s.meshDataCount);
    for (const HostTri& t : out.tris) { meshBox[t.inst].grow(t.p0, t.p0); meshBox[t.inst].grow(t.p1, t.p1); meshBox[t.inst].grow(t.p2, t.p2); }
    out.instances.resize(s.instanceCount);
    out.instShade.resize(s.instanceCount);
    std::vector<float> ownBoxes;
    std::vector<float>& boxes = worldBoxes ? *worldBoxes : ownBoxes;       // 6 floats per instance: padded world box, min then max
    boxes.resize((size_t)s.instanceCount * 6);
    std::atomic<uint32_t> firstBad{ 0xFFFFFFFFu };                          // lowest instance index that cannot be represented (the error names that one, whatever the thread count)
    auto note_bad = [&](uint32_t i) { uint32_t cur = firstBad.load(); while (i < cur && !firstBad.compare_exchange_weak(cur, i)) {} };
    for_instance_ranges(s.instanceCount, [&](uint32_t first, uint32_t last) {
        for (uint32_t i = first; i < last; ++i) {
            const HrptPerInstanceData& in = s.instances[i];
            HostInstance hi{};
            const float* M = in.m_World;
            for (int r = 0; r < 4; ++r) for (int k = 0; k < 3; ++k) hi.world[3 * r + k] = M[4 * r + k];
            if (!invert_affine(M, hi.inv)) { note_bad(i); continue; }
            hi.blasRoot = meshRoot[in.m_MeshDataIndex]; hi.flags = triangle_flags_for_material(s.materials[in.m_MaterialIndex]); hi.material = in.m_MaterialIndex;
            hi.mesh = in.m_MeshDataIndex;
            // world box: the eight corners of the mesh's object box through the flat path's transform, padded like every box
            const Box& ob = meshBox[in.m_MeshDataIndex];
            Box wb; float maxAbs = 0.0f; bool finite = true;
            if (ob.mn[0] <= ob.mx[0]) {
                for (int c = 0; c < 8; ++c) {
                    const float p[3] = { (c & 1) ? ob.mx[0] : ob.mn[0], (c & 2) ? ob.mx[1] : ob.mn[1], (c & 4) ? ob.mx[2] : ob.mn[2] };
                    float w[3]; transform_point(p, M, w);
                    for (int k = 0; k < 3; ++k) if (!std::isfinite(w[k])) finite = false;
                    wb.grow(w, w);
                }
                if (!finite) { note_bad(i); continue; }
                const float extPad = 1e-6f * std::max(wb.mx[0] - wb.mn[0], std::max(wb.mx[1] - wb.mn[1], wb.mx[2] - wb.mn[2]));      // as for triangles (triangle_extent): no triangle inside is larger
                for (int k = 0; k < 3; ++k) {
                    maxAbs = std::max(maxAbs, std::max(std::fabs(wb.mn[k]), std::fabs(wb.mx[k])));
                    const float pad = 1e-5f * std::max(std::fabs(wb.mn[k]), std::fabs(wb.mx[k])) + 1e-6f + extPad;
                    wb.mn[k] -= pad; wb.mx[k] += pad;
                }
            } else { for (int k = 0; k < 3; ++k) { wb.mn[k] = wb.mx[k] = M[12 + k]; } }
            // a world-space vertex is rounded to binary32 after the transform; the error is mapped back to object space through |Minv| and taken
            // eight times over
            float invNorm = 0.0f;
            for (int k = 0; k < 3; ++k) invNorm = std::max(invNorm, std::fabs(hi.inv[k]) + std::fabs(hi.inv[3 + k]) + std::fabs(hi.inv[6 + k]));
            hi.invNorm = invNorm;
            float om = 0.0f;
            if (ob.mn[0] <= ob.mx[0]) for (int k = 0; k < 3; ++k) om = std::max(om, std::max(std::fabs(ob.mn[k]), std::fabs(ob.mx[k])));
            hi.objMaxAbs = om * (1.0f + 1e-5f) + 1e-6f;
            // every partial sum of transform_point is bounded by |p|max * (column sum of |M|) + |T|: three roundings of at most half an ulp of that
            float fwdNorm = 0.0f, tMax = 0.0f;
            for (int k = 0; k < 3; ++k) { fwdNorm = std::max(fwdNorm, std::fabs(M[k]) + std::fabs(M[4 + k]) + std::fabs(M[8 + k])); tMax = std::max(tMax, std::fabs(M[12 + k])); }
            hi.boxEps = 8.0f * 2.4e-7f * std::max(maxAbs, om * fwdNorm + tMax) * invNorm;
            out.instances[i] = hi;
            float* bx = &boxes[(size_t)i * 6];
            for (int k = 0; k < 3; ++k) { bx[k] = wb.mn[k]; bx[3 + k] = wb.mx[k]; }
            instance_shade(M, out.instShade[i]);
        }
    });
    if (firstBad.load() != 0xFFFFFFFFu) {
        const uint32_t i = firstBad.load();
        float inv[12];
        if (!invert_affine(s.instances[i].m_World, inv)) error = "instance " + std::to_string(i) + " has a singular world matrix: the two-level structure cannot represent it";
        else error = "non-finite vertex position";
        return false;
    }
    // the tree over the instances occupies the first tlasNodeCount nodes: the array is put together again, mesh trees shifted, when that count changes
    std::vector<HostNode4> tlas; uint32_t dT = 0; int32_t rootLeaf = 0;
    if (worldBoxes) {           // tree built by the caller (GPU): the node range is reserved (one node per instance: an upper bound), its contents are the caller's
        if (out.tlasNodeCount == s.instanceCount && out.tlasRootLeaf == 0) { out.maxDepth4Tlas = 0; return true; }      // reserved already (a rebuild): nothing moves
        HostNode4 empty{};
        for (int k = 0; k < 4; ++k) { empty.child[k] = kEmptyChild; empty.minx[k] = empty.miny[k] = empty.minz[k] = empty.maxx[k] = empty.maxy[k] = empty.maxz[k] = 1e30f; }
        tlas.assign(s.instanceCount, empty);
    } else if (s.instanceCount > 0) {
        std::vector<HostTri> leafSrc(s.instanceCount);
        Builder b; BuiltBvh tl; b.src = &leafSrc; b.out = &tl; b.minLeaf = 1; b.maxLeaf = 1;
        b.prims.resize(s.instanceCount);
        for (uint32_t i = 0; i < s.instanceCount; ++i) {
            Prim& pr = b.prims[i]; const float* bx = &boxes[(size_t)i * 6];
            for (int k = 0; k < 3; ++k) { pr.bmin[k] = bx[k]; pr.bmax[k] = bx[3 + k]; pr.c[k] = 0.5f * bx[k] + 0.5f * bx[3 + k]; }
            pr.tri = i; leafSrc[i].inst = i; leafSrc[i].prim = 0; leafSrc[i].flags = 0;
        }
        Box root;
        const int32_t r = b.build(0, s.instanceCount, 0, root);
        auto inst_leaf = [&](int32_t ref) { const uint32_t first = ((uint32_t)~ref) >> 2; return ~(int32_t)(tl.tris[first].inst << 2); };
        if (r < 0) rootLeaf = inst_leaf(r);
        else {
            collapse4(tl.nodes, 0, tlas, 0, dT); dT += 1;
            for (HostNode4& n : tlas) for (int k = 0; k < 4; ++k) if (n.child[k] != kEmptyChild && n.child[k] < 0) n.child[k] = inst_leaf(n.child[k]);
        }
    }
    const int32_t shift = (int32_t)tlas.size() - (int32_t)out.tlasNodeCount;
    std::vector<HostNode4> all; all.reserve(out.nodes4.size() + (size_t)std::max(shift, 0));
    all.insert(all.end(), tlas.begin(), tlas.end());
    all.insert(all.end(), out.nodes4.begin() + out.tlasNodeCount, out.nodes4.end());
    if (shift != 0) {
        for (size_t k = tlas.size(); k < all.size(); ++k) for (int c = 0; c < 4; ++c) if (all[k].child[c] != kEmptyChild && all[k].child[c] >= 0) all[k].child[c] += shift;
        for (HostInstance& hi : out.instances) if (hi.blasRoot >= 0) hi.blasRoot += shift;
    }
    out.nodes4.swap(all); out.tlasNodeCount = (uint32_t)tlas.size(); out.tlasRootLeaf = rootLeaf; out.maxDepth4Tlas = dT;
    return true;
}

bool build_scene_two_level(const HrptSceneDesc& s, BuiltTwoLevel& out, std::string& error, std::vector<float>* worldBoxes)
{
    out = BuiltTwoLevel();
    uint64_t triCount = 0;
    if (!validate_scene(s, triCount, error, false)) return false;
    if (s.instanceCount >= kMaxStructureNodes) { error = "too many instances for the two-level structure (2^25)"; return false; }
    std::vector<int32_t> meshRoot;
    if (!build_mesh_trees(s, out, meshRoot, error)) return false;
    // seed instance records so that rebuild_two_level_instances finds the mesh roots
    out.instances.assign(s.instanceCount, HostInstance{});
    for (uint32_t i = 0; i < s.instanceCount; ++i) { out.instances[i].mesh = s.instances[i].m_MeshDataIndex; out.instances[i].blasRoot = meshRoot[s.instances[i].m_MeshDataIndex]; }
    return rebuild_two_level_instances(s, out, error, worldBoxes);
}

} // namespace hrt
