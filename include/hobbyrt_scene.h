/*
 * hobbyrt_scene.h -- C ABI of the scene-data formats on either side of the path tracer.
 * Host-only code (libhobbyrt_scene.so, plain g++): it produces / exchanges the arrays hrpt_upload_scene consumes
 * (include/hobbyrt_pt.h). No GPU work happens here.
 *
 *   Cooked-mesh cache "RLFY" v1: /root/reference/src/SceneCache.h:7-33 (format), src/SceneCache.cpp:22-146
 *   (SaveCookedMesh / LoadCookedMesh). Byte-compatible with files the reference writes next to a glTF as
 *   <scene_stem>_mesh.bin (src/SceneCache.cpp:155).
 */
#ifndef HOBBYRT_SCENE_H
#define HOBBYRT_SCENE_H

#include <stddef.h>
#include <stdint.h>

#include "hobbyrt_pt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HRSC_OK                 0
#define HRSC_ERR_INVALID_ARG   -1
#define HRSC_ERR_IO            -2   /* missing / unreadable / unwritable file */
#define HRSC_ERR_FORMAT        -3   /* wrong magic, wrong version, truncated or inconsistent payload */

#define HRSC_COOKED_MESH_MAGIC   0x59464C52u   /* "RLFY", src/SceneCache.h:26 */
#define HRSC_COOKED_MESH_VERSION 1u            /* src/SceneCache.h:33 */

/* Scene::Primitive as serialised (src/SceneCache.cpp:51-57): 16 bytes, no padding */
typedef struct HrscPrimitive {
    uint32_t m_VertexOffset;
    uint32_t m_VertexCount;
    int32_t  m_MaterialIndex;
    uint32_t m_MeshDataIndex;
} HrscPrimitive;

/* srrhi::Meshlet (src/shaders/Mesh.sr:27-35): 28 bytes */
typedef struct HrscMeshlet {
    uint32_t m_CenterRadius[2];
    uint32_t m_VertexOffset;
    uint32_t m_TriangleOffset;
    uint32_t m_VertexCount;
    uint32_t m_TriangleCount;
    uint32_t m_ConeAxisAndCutoff;
} HrscMeshlet;

/* Everything one cooked-mesh file holds. Mesh i owns primitives [meshPrimitiveOffsets[i], meshPrimitiveOffsets[i+1])
 * and the local bounding sphere meshSpheres[4*i .. 4*i+3] = (m_Center.xyz, m_Radius). */
typedef struct HrscCookedMesh {
    uint32_t meshCount;
    const uint32_t*            meshPrimitiveOffsets;   /* meshCount + 1 entries */
    const HrscPrimitive*       primitives;
    const float*               meshSpheres;            /* 4 * meshCount */
    uint64_t meshDataCount;        const HrptMeshData*        meshData;
    uint64_t meshletCount;         const HrscMeshlet*         meshlets;
    uint64_t meshletVertexCount;   const uint32_t*            meshletVertices;
    uint64_t meshletTriangleCount; const uint32_t*            meshletTriangles;
    uint64_t vertexCount;          const HrptVertexQuantized* vertices;
    uint64_t indexCount;           const uint32_t*            indices;
} HrscCookedMesh;

/* Thread-local message of the last failing hrsc_* call on this thread. */
const char* hrsc_last_error(void);

/* SceneCache::LoadCookedMesh (src/SceneCache.cpp:80-146). On success *out is a library-owned object (free it with
 * hrsc_cooked_mesh_free). Missing file -> HRSC_ERR_IO; magic/version mismatch -> HRSC_ERR_FORMAT, as the reference
 * returns false. Deliberate difference: a truncated file is HRSC_ERR_FORMAT here, while the reference's
 * `!is.good() && !is.eof()` test (:139) lets it through with zero-filled arrays. */
int  hrsc_cooked_mesh_load(const char* path, HrscCookedMesh** out);
void hrsc_cooked_mesh_free(HrscCookedMesh* mesh);
/* SceneCache::SaveCookedMesh (src/SceneCache.cpp:22-78): truncates and rewrites `path`. */
int  hrsc_cooked_mesh_save(const char* path, const HrscCookedMesh* mesh);
/* SceneCache::IsCacheValid (src/SceneCache.cpp:7-20): 1 when `cachePath` exists and is not older than `sourcePath`. */
int  hrsc_cache_is_valid(const char* cachePath, const char* sourcePath);

/* ---- glTF 2.0 scene ingestion --------------------------------------------------------------
 * Scene::LoadScene (src/Scene.cpp:9-65) without the GPU upload: SceneLoader::LoadGLTFScene (src/SceneLoader.cpp:2495-2570:
 * materials incl. KHR_materials_transmission / ior / volume / emissive_strength / pbrSpecularGlossiness, textures + samplers,
 * perspective cameras, KHR_lights_punctual, meshes -> VertexQuantized + LOD-0 indices, node hierarchy with RH -> LH
 * conversion), the cooked-mesh cache next to the file when asked for, FinalizeLoadedScene (instance order), texture decode
 * (PNG, DDS RGBA8/BC1-BC5) to RGBA8, MaterialConstantsFromMaterial, CreateAndUploadLightBuffer, first scene camera.
 * The result is exactly the set of arrays hrpt_upload_scene takes (the Bruneton LUTs are the caller's). */
typedef struct HrscScene HrscScene;
#define HRSC_LOAD_USE_MESH_CACHE 1u     /* read <stem>_mesh.bin when newer than the glTF, write it after cooking otherwise */

typedef struct HrscSceneView {
    const HrptVertexQuantized* vertices;   uint32_t vertexCount;
    const uint32_t*            indices;    uint32_t indexCount;
    const HrptMeshData*        meshData;   uint32_t meshDataCount;
    const HrptPerInstanceData* instances;  uint32_t instanceCount;
    const HrptMaterialConstants* materials; uint32_t materialCount;
    const HrptGPULight*        lights;     uint32_t lightCount;
    const HrptTextureDesc*     textures;   uint32_t textureCount;    /* bindless table: slots 0..10 are the default textures (null here) */
    float    sunDirection[3];              /* Scene::GetSunDirection */
    float    sunAngularSizeDeg;            /* Scene::Light::m_AngularSize of the directional light */
    uint32_t cameraCount;                  /* perspective cameras found; the view below is camera 0 or the default camera */
    float    cameraPosition[3], cameraYaw, cameraPitch, cameraFovY, cameraAspect, cameraNearZ;
    uint32_t nodeCount, meshCount, sceneTextureCount, warningCount;
    uint32_t loadedFromMeshCache;
} HrscSceneView;

int  hrsc_scene_load(const char* path, uint32_t flags, HrscScene** out);
void hrsc_scene_free(HrscScene* scene);
int  hrsc_scene_view(const HrscScene* scene, HrscSceneView* out);          /* pointers stay valid until hrsc_scene_free */
const char* hrsc_scene_warning(const HrscScene* scene, uint32_t index);   /* non-fatal findings of the load (skipped textures, ...) */
/* PNG / JPEG / DDS bytes -> level 0 as RGBA8 (malloc'ed; release with hrsc_free_pixels); files with float texels are refused here. */
int  hrsc_decode_image(const uint8_t* bytes, size_t byteCount, uint32_t* width, uint32_t* height, uint8_t** rgba);
/* The same with everything the file holds: HRPT_TEXTURE_FORMAT_* of the decoded texels (DDS *_SRGB formats -> RGBA8_SRGB, BC6H and the
 * float formats -> RGBA16_FLOAT / RGBA32_FLOAT) and all mip levels, level 0 first, tightly packed (src/TextureLoader.cpp:196-213). */
int  hrsc_decode_image_ex(const uint8_t* bytes, size_t byteCount, uint32_t* width, uint32_t* height, uint32_t* format, uint32_t* mipCount,
                          uint8_t** texels, size_t* texelBytes);
void hrsc_free_pixels(uint8_t* rgba);
/* 0 when the BC7 partition and anchor-index tables of the decoder agree with each other (every anchor lies in the subset it anchors). */
int  hrsc_selftest_bc7_tables(void);

#ifdef __cplusplus
}
#endif
#endif
