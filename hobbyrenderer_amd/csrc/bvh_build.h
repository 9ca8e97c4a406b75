// bvh_build.h -- host-side construction of the library-owned acceleration structure that stands in
// for the driver BLAS/TLAS of Scene::BuildAccelerationStructures (/root/reference/src/Scene.cpp:67-214).
//
// MI355X-first choice: no two-level structure. Instance transforms are static on this path
// (PathTracerRenderer::Render pauses animation, src/PathTracerRenderer.cpp:53), so every triangle is
// pre-transformed to world space once and a single BVH2 is built over all of them; rays are never
// re-transformed and a traversal step is one 64-byte read. The contract kept from the reference:
// triangle p of instance i is indices[m_IndexOffsets[0] + 3p + {0,1,2}] -> vertices[].m_Pos through
// the instance's world matrix; CommittedInstanceIndex = i, CommittedPrimitiveIndex = p; instances whose
// material alpha mode is OPAQUE are ForceOpaque, the others raise candidates (src/Scene.cpp:135,150-154).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/hobbyrt_pt.h"

namespace hrt {

// GpuTri::flags of every triangle of an instance with material m: bit 0 = the instance is ForceOpaque (alpha mode OPAQUE,
// src/Scene.cpp:150-154); bits 1-2 = shading class, which wf_extend copies into the hit record so that wf_shade can group the paths of
// a segment by the code path they will take (north_star: "ray compaction/sort by material"):
//   0 constants only   1 samples textures (RaytracingCommon.hlsli:252-296)   2 takes the transmission branch (PathTracer.hlsl:149-255)
inline uint32_t triangle_flags_for_material(const HrptMaterialConstants& m)
{
    const uint32_t opaque = m.m_AlphaMode == HRPT_ALPHA_MODE_OPAQUE ? 1u : 0u;
    const bool transmissive = m.m_TransmissionFactor > 0.0f || m.m_AlphaMode == HRPT_ALPHA_MODE_BLEND;
    const uint32_t cls = transmissive ? 2u : (m.m_TextureFlags != 0 ? 1u : 0u);
    return opaque | (cls << 1);
}

struct HostNode {           // mirrors hrt::GpuNode (pt_device.h), 64 B
    float lmin[3]; int32_t left;
    float lmax[3]; int32_t right;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};
struct HostTri {            // mirrors hrt::GpuTri, 48 B
    float p0[3]; uint32_t inst;
    float p1[3]; uint32_t prim;
    float p2[3]; uint32_t flags;
};
// 4-wide node for the wavefront kernels (collapsed from the BVH2): child boxes in SoA (x of 4 children, y, z, ...) so
// the four slab tests are data-parallel; 128 B = two cache lines fetched together. Empty slots: a degenerate box at 1e30.
// Row order min/max interleaved per axis: the traversal picks the NEAR row of an axis by the sign of the ray direction
// (byte offset axis * 32 + (d < 0 ? 16 : 0)) and gets the far row as near ^ 16 (pt_device.h inner_step).
struct HostNode4 {
    float minx[4], maxx[4], miny[4], maxy[4], minz[4], maxz[4];
    int32_t child[4];               // >= 0 inner node4 index, < 0 leaf (same encoding as HostNode), kEmptyChild = unused slot
    uint32_t pad[4];
};
constexpr int32_t kEmptyChild = 0x7fffffff;

// Shading attributes of one world triangle, unpacked ONCE at upload with exactly the arithmetic of UnpackVertex
// (src/shaders/MeshCommon.hlsli:9-22): what GetTriangleVertices + UnpackVertex (RaytracingCommon.hlsli:33-50)
// would produce per hit. 80 B, same (leaf) order as HostTri, so a hit is one index into both arrays.
struct HostTriAttr {
    float n0[3], n1[3], n2[3];      // decoded vertex normals (local space)
    float uv0[2], uv1[2], uv2[2];   // f16tof32 texture coordinates
    uint32_t material;              // PerInstanceData::m_MaterialIndex of the owning instance
    uint32_t inst;                  // CommittedInstanceIndex
    uint32_t prim;                  // CommittedPrimitiveIndex
    uint32_t pad[2];
};
struct HostTriTangent { float t0[4], t1[4], t2[4]; };   // DecodeOct tangents + sign, only built when a normal map exists
// MakeAdjugateMatrix(world) rows (Common.hlsli:33-41), computed once per instance with the same cross products.
struct HostInstShade { float adj0[4], adj1[4], adj2[4]; };
static_assert(sizeof(HostNode4) == 128, "GPU layouts");
static_assert(sizeof(HostNode) == 64 && sizeof(HostTri) == 48 && sizeof(HostTriAttr) == 80 && sizeof(HostInstShade) == 48, "GPU layouts");

struct BuiltBvh {
    std::vector<HostNode> nodes;    // empty when the scene fits one leaf
    std::vector<HostNode4> nodes4;  // the same tree collapsed to 4-wide nodes (root = 0); empty when nodes is empty
    uint32_t maxDepth4 = 0;         // depth of the 4-wide tree; a traversal stack needs at most 3 * maxDepth4 + 1 entries
    std::vector<HostTri> tris;      // leaf order
    std::vector<HostTriAttr> attrs; // parallel to tris
    std::vector<HostTriTangent> tangents; // parallel to tris, empty unless some material samples a normal map
    std::vector<HostInstShade> instShade; // per instance
    int32_t rootLeaf = 0;           // encoded leaf when nodes is empty and tris is not
    uint32_t maxDepth = 0;
    float sahCost = 0.0f;           // SAH cost of the 2-wide tree (node cost 1, triangle cost 1), root area = 1
};

// ---- two-level structure for instanced scenes (the reference's own form: one BLAS per mesh + a TLAS of instances, src/Scene.cpp:98-154) ----
// The flat tree above costs memory and build time proportional to instances x triangles. For scenes where that product is large the
// library keeps one object-space tree per DISTINCT mesh and a tree over the instances' world boxes instead. The hit definition does not
// change: at a leaf the traversal transforms the three object-space vertices with the instance's m_World exactly as the flat upload
// does (transform_point) and runs the same watertight test on the world-space ray, so (t, u, v) and the (t, instance, primitive) order
// are bit-identical to the flat path; only the culling runs on an object-space ray.
struct HostInstance {           // mirrors hrt::GpuInstance (pt_device.h), 128 B
    float world[12];            // rows 0..3 of m_World, xyz each (row-vector convention: p_world = p * M, translation in row 3)
    float inv[12];              // the inverse map in the same layout: p_object = p_world * Minv
    int32_t blasRoot;           // node4 index of the mesh's tree (>= 0) or an encoded leaf (< 0) when the whole mesh is one leaf
    uint32_t flags;             // triangle_flags_for_material of the instance's material (bit 0 opaque, bits 1-2 shading class)
    uint32_t material;
    float boxEps;               // object-space slack added to both sides of every BLAS slab: the fp32 rounding of the world-space vertices
                                // (half an ulp of the instance's largest world coordinate) mapped back through |Minv|, times a safety factor
    uint32_t mesh;
    float objMaxAbs;            // largest |coordinate| of the mesh's (padded) object-space box
    float invNorm;              // max column sum of |Minv| (bounds |v * Minv| by invNorm * max|v_k|)
    uint32_t pad;
};
static_assert(sizeof(HostInstance) == 128, "GPU layouts");
struct BuiltTwoLevel {
    std::vector<HostNode4> nodes4;        // [0, tlasNodeCount): the tree over instances (root 0, leaves = one instance: ~((instance << 2) | 0));
                                          // behind it the trees of the distinct meshes, child indices already offset into this array
    uint32_t tlasNodeCount = 0;
    int32_t tlasRootLeaf = 0;             // encoded instance leaf when the scene has a single instance (no TLAS nodes)
    std::vector<HostTri> tris;            // object-space triangles of all distinct meshes in leaf order (inst = mesh index, prim = primitive, flags = 0)
    std::vector<HostTriAttr> attrs;       // parallel to tris (material / inst fields unused: they come from the instance)
    std::vector<HostTriTangent> tangents;
    std::vector<HostInstance> instances;
    std::vector<HostInstShade> instShade;
    uint32_t maxDepth4Tlas = 0, maxDepth4Blas = 0;
    uint32_t distinctMeshes = 0;
};
// Stack need of the two-level traversal: 3 entries per 4-wide level of both trees + the BLAS exit marker + 2.
inline uint32_t two_level_stack_need(const BuiltTwoLevel& b) { return 3u * (b.maxDepth4Tlas + b.maxDepth4Blas) + 3u; }
bool build_scene_two_level(const HrptSceneDesc& scene, BuiltTwoLevel& out, std::string& error, std::vector<float>* worldBoxes = nullptr);
// The instance records + the tree over the instances only (moving objects: the per-frame TLAS rebuild of src/CommonRenderers.cpp:234-246);
// `out` keeps its mesh trees.
// worldBoxes != nullptr: the tree is NOT built -- the caller builds it on the GPU from the padded world boxes returned here (6 floats per
// instance: min xyz, max xyz) and writes it into the first scene.instanceCount nodes, which this call reserves (tlasNodeCount = instanceCount,
// the mesh trees behind them; maxDepth4Tlas is the caller's to set).
bool rebuild_two_level_instances(const HrptSceneDesc& scene, BuiltTwoLevel& out, std::string& error, std::vector<float>* worldBoxes = nullptr);

constexpr uint32_t kMaxLeafTris = 4;
constexpr uint32_t kTraversalStackDepth = 64;   // 2-wide trees: the builders guarantee depth + 2 <= this (private / LDS + overflow stacks of the kernels)
constexpr uint32_t kHostBuilderDepthGoal = 32;  // the host SAH builder switches to median splits early enough to stay below this

// Validates every index / range of the scene description (false + message); triCount = world triangles over all instances.
// (flatLimit: the traversal kernels address nodes and triangle records by 32-bit byte offsets from the array base (pt_device.h GlobalBvh4), so one
// structure holds fewer than 2^32 / 48 triangle records and 2^25 nodes of 128 bytes (checked after the build: a 4-wide tree over N triangles
// has between N / 12 and N nodes); the two-level structure only counts DISTINCT triangles, and instances as leaves of its upper tree)
constexpr uint64_t kMaxStructureTriangles = 0xFFFFFFFFull / 48ull, kMaxStructureNodes = 1ull << 25;
bool validate_scene(const HrptSceneDesc& scene, uint64_t& triCount, std::string& error, bool flatLimit = true);
void build_instance_shade(const HrptSceneDesc& scene, std::vector<HostInstShade>& out);
bool scene_needs_tangents(const HrptSceneDesc& scene);
// validate_scene + host build (binned SAH).
bool build_scene_bvh(const HrptSceneDesc& scene, BuiltBvh& out, std::string& error);

// Area-greedy 4-wide collapse of a 2-wide tree (depth-first node order); experiment hook for GPU-built trees (HRPT_GPU_BVH_HOST_COLLAPSE).
void collapse_bvh2_on_host(const std::vector<HostNode>& nodes2, std::vector<HostNode4>& nodes4, uint32_t& maxDepth4);

} // namespace hrt
