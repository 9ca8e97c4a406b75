/*
 * hobbyrt_pt.h -- C ABI of libhobbyrt_pt.so, the MI355X (gfx950) drop-in for the
 * reference path-tracer pass of lawfuyang/HobbyRenderer.
 *
 * What it replaces (reference file:line, all under /root/reference):
 *   - the GPU work of PathTracerRenderer::Render, src/PathTracerRenderer.cpp:31-106:
 *     writeBuffer(PathTracerCB) + PathTracerInputs binding + dispatch of
 *     PathTracer_CSMain (src/shaders/PathTracer.hlsl:53-340);
 *   - the acceleration structures of Scene::BuildAccelerationStructures,
 *     src/Scene.cpp:67-214 (driver BLAS/TLAS -> library-owned BVH);
 *   - the scene buffer uploads of SceneLoader::CreateAndUploadGpuBuffers /
 *     CreateAndUploadLightBuffer, src/SceneLoader.cpp:2319-2493;
 *   - the Bruneton LUT upload of CommonResources, src/CommonResources.cpp:519-569.
 *
 * Conventions: extern "C", plain pointers and sizes, no exceptions, no torch types.
 * Every call returns HRPT_OK (0) or a negative HrptStatus; the message of the last
 * failure on a context is available from hrpt_last_error(). The reference has no
 * return codes (SDL_assert + log, src/pch.h:77); a reference-side caller asserts on
 * != HRPT_OK. One context = one GPU = one HIP stream; calls on one context must be
 * serialised by the caller, different contexts may be driven from different threads
 * (mirrors: Render runs on a TaskScheduler worker with its own command list,
 * src/RenderGraph.cpp:329-349).
 *
 * All struct layouts are the structured-buffer / cbuffer layouts of the reference
 * shaders (the .sr files under src/shaders); sizes are checked with static asserts below.
 */
#ifndef HOBBYRT_PT_H
#define HOBBYRT_PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HRPT_ABI_VERSION 3

typedef enum HrptStatus {
    HRPT_OK = 0,
    HRPT_ERR_INVALID_ARGUMENT = -1,   /* null pointer, bad size, index out of range in scene data */
    HRPT_ERR_NO_DEVICE = -2,          /* no HIP device / device ordinal out of range */
    HRPT_ERR_HIP = -3,                /* a HIP runtime call failed; see hrpt_last_error */
    HRPT_ERR_NO_SCENE = -4,           /* render before upload_scene */
    HRPT_ERR_OUT_OF_MEMORY = -5,
    HRPT_ERR_UNSUPPORTED = -6
} HrptStatus;

/* ---- GPU struct layouts (src/shaders/Mesh.sr, Instance.sr, GPULight.sr, Common.sr, PathTracer.sr) ---- */

typedef struct HrptVertexQuantized {       /* Mesh.sr:9-15, 24 B */
    float    m_Pos[3];
    uint32_t m_Normal;                     /* 10:10:10 snorm+511, bit 30 = tangent sign (1 => -1) */
    uint32_t m_Uv;                         /* 2 x fp16 */
    uint32_t m_Tangent;                    /* 8:8 octahedral +127 */
} HrptVertexQuantized;

typedef struct HrptMeshData {              /* Mesh.sr:17-25, 164 B */
    uint32_t m_LODCount;
    uint32_t m_IndexOffsets[8];
    uint32_t m_IndexCounts[8];
    uint32_t m_MeshletOffsets[8];
    uint32_t m_MeshletCounts[8];
    float    m_LODErrors[8];
} HrptMeshData;

typedef struct HrptPerInstanceData {       /* Instance.sr:49-65, 160 B */
    float    m_World[16];                  /* row-major, row-vector convention: p_world = p * M, translation in row 3 */
    float    m_PrevWorld[16];
    uint32_t m_MaterialIndex;
    uint32_t m_MeshDataIndex;
    float    m_Radius;
    uint32_t m_LODIndex;
    float    m_Center[3];
    uint32_t m_FirstGeometryInstanceIndex;
} HrptPerInstanceData;

typedef struct HrptMaterialConstants {     /* Instance.sr:2-46, 180 B */
    float    m_BaseColor[4];
    float    m_EmissiveFactor[4];
    float    m_RoughnessMetallic[2];
    uint32_t m_TextureFlags;
    uint32_t m_AlbedoTextureIndex;
    uint32_t m_NormalTextureIndex;
    uint32_t m_RoughnessMetallicTextureIndex;
    uint32_t m_EmissiveTextureIndex;
    uint32_t m_AlbedoSamplerIndex;
    uint32_t m_NormalSamplerIndex;
    uint32_t m_RoughnessSamplerIndex;
    uint32_t m_EmissiveSamplerIndex;
    uint32_t m_AlbedoMinMipIndex;
    uint32_t m_NormalMinMipIndex;
    uint32_t m_RoughnessMinMipIndex;
    uint32_t m_EmissiveMinMipIndex;
    uint32_t m_AlbedoFeedbackIndex;
    uint32_t m_NormalFeedbackIndex;
    uint32_t m_RoughnessFeedbackIndex;
    uint32_t m_EmissiveFeedbackIndex;
    uint32_t m_MinMipDimsX;
    uint32_t m_MinMipDimsY;
    uint32_t m_AlphaMode;
    float    m_AlphaCutoff;
    float    m_IOR;
    float    m_TransmissionFactor;
    float    m_ThicknessFactor;
    float    m_AttenuationDistance;
    float    m_AttenuationColor[3];
    float    m_SigmaA[3];
    uint32_t m_IsThinSurface;
    float    m_SigmaS[3];
} HrptMaterialConstants;

typedef struct HrptGPULight {              /* GPULight.sr:1-13, 64 B */
    float    m_Position[3];
    float    m_Intensity;
    float    m_Direction[3];
    uint32_t m_Type;                       /* 0 directional, 1 point, 2 spot */
    float    m_Color[3];
    float    m_Range;
    float    m_SpotInnerConeAngle;
    float    m_SpotOuterConeAngle;
    float    m_Radius;
    float    m_CosSunAngularRadius;
} HrptGPULight;

typedef struct HrptPlanarViewConstants {   /* Common.sr:17-43, 704 B */
    float m_MatWorldToView[16];
    float m_MatViewToClip[16];
    float m_MatWorldToClip[16];
    float m_MatClipToView[16];
    float m_MatViewToWorld[16];
    float m_MatClipToWorld[16];
    float m_MatViewToClipNoOffset[16];
    float m_MatWorldToClipNoOffset[16];
    float m_MatClipToViewNoOffset[16];
    float m_MatClipToWorldNoOffset[16];
    float m_ViewportOrigin[2];
    float m_ViewportSize[2];
    float m_ViewportSizeInv[2];
    float m_PixelOffset[2];
    float m_ClipToWindowScale[2];
    float m_ClipToWindowBias[2];
    float m_CameraDirectionOrPosition[4];
} HrptPlanarViewConstants;

/* cbuffer PathTracerConstants, PathTracer.sr:6-17, 768 B (offsets follow HLSL cbuffer
 * packing; the generated srrhi header is not in the reference tree). */
typedef struct HrptPathTracerConstants {
    HrptPlanarViewConstants m_View;        /* @0   */
    float    m_CameraPos[4];               /* @704 */
    uint32_t m_LightCount;                 /* @720 */
    uint32_t m_AccumulationIndex;          /* @724 */
    uint32_t m_FrameIndex;                 /* @728 (written, never read by the shader) */
    uint32_t m_MaxBounces;                 /* @732 */
    float    m_Jitter[2];                  /* @736 */
    float    m_Pad0[2];                    /* @744 (float3 may not straddle a 16-byte row) */
    float    m_SunDirection[3];            /* @752 */
    float    m_CosSunAngularRadius;        /* @764 */
} HrptPathTracerConstants;

/* CommonConsts (Common.sr:47-140) used on this path */
enum {
    HRPT_TEXFLAG_ALBEDO = 1, HRPT_TEXFLAG_NORMAL = 2, HRPT_TEXFLAG_ROUGHNESS_METALLIC = 4, HRPT_TEXFLAG_EMISSIVE = 8,
    HRPT_ALPHA_MODE_OPAQUE = 0, HRPT_ALPHA_MODE_MASK = 1, HRPT_ALPHA_MODE_BLEND = 2,
    HRPT_TRANSMITTANCE_TEXTURE_WIDTH = 256, HRPT_TRANSMITTANCE_TEXTURE_HEIGHT = 64,
    HRPT_SCATTERING_TEXTURE_WIDTH = 256, HRPT_SCATTERING_TEXTURE_HEIGHT = 128, HRPT_SCATTERING_TEXTURE_DEPTH = 32,
    HRPT_IRRADIANCE_TEXTURE_WIDTH = 64, HRPT_IRRADIANCE_TEXTURE_HEIGHT = 16,
    HRPT_LIGHT_DIRECTIONAL = 0, HRPT_LIGHT_POINT = 1, HRPT_LIGHT_SPOT = 2
};

/* A bindless 2D texture. The stb path of the reference produces RGBA8_UNORM with one level (src/TextureLoader.cpp:215-250); its DDS
 * path keeps the file's format and mip chain (:66-135, :196-213): *_SRGB formats are linearised by the sampler BEFORE filtering,
 * BC1-5 / BC7 decode to 8-bit channels, BC6H and the float formats to float channels. The caller hands over DECODED texels
 * (hobbyrt::DecodeImage, include/hobbyrt_scene.h, does the block decompression): */
enum {
    HRPT_TEXTURE_FORMAT_RGBA8_UNORM = 0,   /* 4 bytes per texel */
    HRPT_TEXTURE_FORMAT_RGBA8_SRGB = 1,    /* 4 bytes per texel; r, g, b through the sRGB -> linear table (include/hobbyrt/srgb_table.h), a linear */
    HRPT_TEXTURE_FORMAT_RGBA16_FLOAT = 2,  /* 8 bytes per texel (binary16) */
    HRPT_TEXTURE_FORMAT_RGBA32_FLOAT = 3   /* 16 bytes per texel */
};
#define HRPT_TEXTURE_MAX_MIPS 16
typedef struct HrptTextureDesc {
    const void* texels;                    /* all levels, level 0 first, tightly packed, rows top to bottom; level l is max(1, width >> l) x
                                              max(1, height >> l) texels; NULL for an unused slot */
    uint32_t width, height;
    uint32_t format;                       /* HRPT_TEXTURE_FORMAT_* */
    uint32_t mipCount;                     /* 0 or 1: level 0 only; at most HRPT_TEXTURE_MAX_MIPS. Only the gradient-sampled alpha test of shadow
                                              rays (AlphaTestGrad, RaytracingCommon.hlsli:112-130,207-240) reads levels above 0 */
} HrptTextureDesc;

/* Everything the reference binds to PathTracerInputs (PathTracer.sr:19-32) plus the
 * global bindless tables it reads (src/Renderer.cpp:1834-1841). Host pointers; the
 * library copies during hrpt_upload_scene and the caller keeps ownership. */
typedef struct HrptSceneDesc {
    const HrptVertexQuantized*   vertices;   uint32_t vertexCount;     /* Scene::m_VertexBufferQuantized */
    const uint32_t*              indices;    uint32_t indexCount;      /* Scene::m_IndexBuffer (global vertex indices) */
    const HrptMeshData*          meshData;   uint32_t meshDataCount;   /* Scene::m_MeshData */
    const HrptPerInstanceData*   instances;  uint32_t instanceCount;   /* Scene::m_InstanceData; index == TLAS instanceID */
    const HrptMaterialConstants* materials;  uint32_t materialCount;   /* MaterialConstantsFromMaterial output */
    const HrptGPULight*          lights;     uint32_t lightCount;      /* CreateAndUploadLightBuffer order */
    /* bindless Texture2D table, index = MaterialConstants::m_*TextureIndex. Slots 0..10 are the
     * reference's default textures (Common.sr:103-113); entries with texels == NULL are unbound. */
    const HrptTextureDesc*       textures;   uint32_t textureCount;
    /* Bruneton LUTs in the file format of bin/bruneton/{transmittance,scattering,irradiance}.dat: raw float32 RGBA. The library
     * converts to RGBA16F like CommonResources.cpp:550-558. irradiance may be NULL (not read on this path). */
    const float* brunetonTransmittance;    /* 256*64*4 floats */
    const float* brunetonScattering;       /* 256*128*32*4 floats */
    const float* brunetonIrradiance;       /* 64*16*4 floats or NULL */
} HrptSceneDesc;

typedef struct HrptDeviceDesc {
    int32_t  deviceOrdinal;                /* HIP device index */
    uint32_t abiVersion;                   /* HRPT_ABI_VERSION */
} HrptDeviceDesc;

/* One dispatch of the reference == one accumulation index. `accumCount` > 1 renders
 * indices first..first+accumCount-1 in one call (the "spp" of BASELINE.json); the
 * per-index jitter / accumulation-index fields of `constants` are then recomputed
 * by the library exactly as PathTracerRenderer::Render does (:62,:65). */
typedef struct HrptFrameParams {
    HrptPathTracerConstants constants;     /* as filled by PathTracerRenderer::Render :58-75 for the FIRST index */
    uint32_t accumCount;                   /* >= 1 */
    /* pixel rectangle [x0,x1) x [y0,y1) rendered by this context (image-tile sharding); 0,0,0,0 = full viewport */
    uint32_t tileX0, tileY0, tileX1, tileY1;
    uint32_t flags;                        /* HRPT_FRAME_* */
    /* column interleaving inside the rectangle (multi-GPU load balance): the rectangle is cut into columns of 8 pixels
     * (the reference's thread-group width, PathTracer.hlsl:52) and only the columns k with k % stripeCount == stripeIndex are rendered;
     * every rank then works on all parts of the image. stripeCount 0 or 1 = the whole rectangle. Pixels outside are not touched. */
    uint32_t stripeCount, stripeIndex;
} HrptFrameParams;

enum {
    HRPT_FRAME_DEFAULT = 0,
    HRPT_FRAME_MEGAKERNEL = 1,             /* one-thread-per-pixel restatement kernel (validation path) */
    HRPT_FRAME_WAVEFRONT = 2,              /* persistent wavefront pipeline (default when available) */
    HRPT_FRAME_PROFILE = 4                 /* record HIP events around every kernel launch: fills HrptStats::*KernelMs (adds launch gaps) */
};

typedef struct HrptStats {
    uint64_t closestRays;                  /* TraceRayStandard queries launched */
    uint64_t shadowRays;                   /* CalculateRTShadow queries launched */
    uint64_t paths;                        /* pixel-paths started */
    float    lastRenderMs;                 /* device time of the last hrpt_render (HIP events on the context stream) */
    /* summed device time / launch count per kernel class of the wavefront pipeline since hrpt_reset_stats
     * (HIP events recorded on the context stream around every launch); zero in megakernel mode */
    float    traceKernelMs;                /* wf_extend: closest-hit traversal */
    uint32_t traceKernelLaunches;
    float    shadeKernelMs;                /* wf_shade: attributes, BSDF, NEE sample generation, compaction */
    uint32_t shadeKernelLaunches;
    float    shadowKernelMs;               /* wf_shadow: NEE visibility */
    uint32_t shadowKernelLaunches;
    uint32_t bvhNodeCount;
    uint32_t bvhTriangleCount;
    uint32_t bvhMaxDepth;
    /* ---- ABI 3: queue accounting of the wavefront pipeline (cumulative since hrpt_reset_stats) -------------------------------
     * Bytes each kernel class moves through its queue streams in HBM: records read / written x record size, from counters the
     * kernels keep (paths per bounce, shadow-queue entries, light samples, sampleRadiance updates) and the record layouts of the
     * configuration the LAST hrpt_render used. BVH nodes / triangles / attribute records / textures are not included: they are served
     * by LDS and L2 (rocprofv3 FETCH_SIZE / WRITE_SIZE give the HBM total). Zero in megakernel mode. */
    uint32_t megakernelFallbacks;          /* hrpt_render calls that asked for the wavefront pipeline but ran the validation megakernel */
    float    raygenKernelMs;               /* wf_raygen (HRPT_FRAME_PROFILE) */
    uint32_t raygenKernelLaunches;
    float    resolveKernelMs;              /* wf_resolve (HRPT_FRAME_PROFILE) */
    uint32_t resolveKernelLaunches;
    uint32_t pad0;
    uint64_t raygenQueueBytes;
    uint64_t traceQueueBytes;              /* ray records read + hit records written */
    uint64_t shadeQueueBytes;              /* path + hit records read, surviving paths + shadow-queue entries written, sampleRadiance updates */
    uint64_t shadowQueueBytes;             /* shadow-queue entries + light samples read, sampleRadiance updates (+ shadow-ray queue in the any-hit schedule) */
    uint64_t resolveQueueBytes;            /* sampleRadiance read, Accumulation read / written, Output written */
    uint64_t neeEntries;                   /* path vertices with at least one light sample */
    uint64_t neeSamples;                   /* light samples drawn (>= shadowRays: a sample below the horizon traces no ray) */
    uint64_t queuePoolBytes;               /* size of the context's queue pool in HBM */
} HrptStats;

typedef struct HrptContext HrptContext;

int  hrpt_create(const HrptDeviceDesc* desc, HrptContext** out);
void hrpt_destroy(HrptContext* ctx);
const char* hrpt_last_error(const HrptContext* ctx);      /* ctx may be NULL: last creation error */

/* Replaces scene buffer upload + BLAS/TLAS build. Validates every index in the scene data.
 * Size limit: one acceleration structure holds fewer than 2^32 / 48 (89 M) triangle records -- world-space triangles (instances x mesh triangles)
 * for the flat structure, distinct mesh triangles for the two-level one (hrpt_set_acceleration_structure) -- and fewer than 2^25 tree nodes,
 * because the traversal kernels address both arrays by 32-bit byte offsets; larger scenes fail with HRPT_ERR_INVALID_ARGUMENT. */
int  hrpt_upload_scene(HrptContext* ctx, const HrptSceneDesc* scene);

/* (Re)allocates the RGBA32F Accumulation (u1) and Output (u0) images, PathTracerRenderer::Setup :14-29. */
int  hrpt_resize(HrptContext* ctx, uint32_t width, uint32_t height);

/* The dispatch. Asynchronous on the context stream. */
int  hrpt_render(HrptContext* ctx, const HrptFrameParams* params);
int  hrpt_synchronize(HrptContext* ctx);
/* Run all further work of the context on the caller's HIP stream (the counterpart of recording into the caller's
 * command list, src/RenderGraph.cpp:329-349; also lets an RCCL all-gather follow the render without a host sync).
 * useCallerStream != 0: hipStream is used as is (NULL = the legacy default stream); 0: back to the context's own
 * stream. The previous stream is drained first. */
int  hrpt_set_stream(HrptContext* ctx, void* hipStream, int useCallerStream);

/* Device pointers of the two images (width*height float4, row-major) for zero-copy consumers
 * (the HDR post chain, RCCL all-gather). */
int  hrpt_get_device_images(HrptContext* ctx, void** accumulation, void** output);
/* ---- acceleration-structure builder --------------------------------------------------------------
 * Scene::BuildAccelerationStructures (src/Scene.cpp:67-214) is a driver BLAS/TLAS build in the reference. Here
 * hrpt_upload_scene builds the library's own structure either on the host (binned SAH on up to 16 host threads: the best tree, ~0.25 s per million
 * triangles) or on the GPU (Morton-order LBVH or PLOC clustering: milliseconds). Radiance is identical either way (the hit
 * definition is BVH-independent). The GPU builders fall back to the host one for scenes under 8 triangles or when their
 * tree is deeper than the traversal stacks allow. The default, HRPT_BVH_BUILDER_AUTO, takes the host builder below 65 536
 * triangles and PLOC above (measured on MI355X: same frame time as the SAH tree at 101 k and 1.17 M triangles, 2 ms / 5 ms
 * instead of 32 ms / 548 ms of build time; on small scenes the SAH tree still renders up to 20 % faster). */
#define HRPT_BVH_BUILDER_HOST_SAH 0
#define HRPT_BVH_BUILDER_GPU_LBVH 1      /* Morton radix tree (Karras 2012): fastest build */
#define HRPT_BVH_BUILDER_GPU_PLOC 2      /* Morton order + nearest-neighbour clustering (PLOC): better tree, a few times the LBVH build time;
                                            falls back to the LBVH hierarchy when its tree is too deep for the traversal stacks */
#define HRPT_BVH_BUILDER_AUTO     3      /* default: HOST_SAH below 65 536 triangles, GPU_PLOC from there on */
#define HRPT_BVH_BUILDER_REFITTED 0x100u /* ORed into HrptBuildInfo::usedBuilder when the last hrpt_refit_instances kept the hierarchy of an earlier GPU build */
int  hrpt_set_bvh_builder(HrptContext* ctx, int builder);       /* takes effect at the next hrpt_upload_scene / hrpt_update_instances */
typedef struct HrptBuildInfo {
    uint32_t requestedBuilder, usedBuilder;     /* HRPT_BVH_BUILDER_* */
    float    buildMs;                           /* host wall time of the last build inside hrpt_upload_scene / hrpt_update_instances (copies included) */
    float    deviceBuildMs;                     /* GPU builder: device time, instance-table copy to last kernel; else 0 */
    uint32_t triangleCount, nodeCount, node4Count, maxDepth, maxDepth4;
    uint32_t mortonBits;                        /* GPU builder: Morton bits of the hierarchy (63 unless the full-code tree was too deep) */
    float    sahCost;                           /* surface-area-heuristic cost of the 2-wide tree: 1 + sum(area(child) * (inner ? 1 : triangles)) / area(root) */
    uint32_t structure;                         /* HRPT_ACCEL_FLAT or HRPT_ACCEL_TWO_LEVEL: what the last build produced */
    uint32_t instanceNodeCount;                 /* two-level: 4-wide nodes of the tree over the instances (node4Count counts those + the mesh trees) */
    uint32_t distinctMeshes;                    /* two-level: meshes with a tree of their own (triangleCount counts THEIR triangles, not instances x triangles) */
    uint32_t leafAreaPermille;                  /* flat structure: surface area of the leaf boxes in the 64-byte quantised nodes / in the fp32 nodes, x 1000 (0: no tree) */
    uint32_t nodeFormat;                        /* what the wavefront kernels walk when the tree is in global memory: 1 = 128-byte fp32 nodes, 2 = 64-byte quantised nodes
                                                   (four instead of seven 16-byte requests per lane and step; chosen when leafAreaPermille <= 1100; HRPT_BVH_NODE_FORMAT=1|2 forces) */
} HrptBuildInfo;                                /* 64 B */
int  hrpt_get_build_info(HrptContext* ctx, HrptBuildInfo* out);

/* Shape of the acceleration structure. The reference builds one BLAS per mesh and a TLAS over the instances (src/Scene.cpp:98-154).
 * HRPT_ACCEL_FLAT (what AUTO picks for all but heavily instanced scenes) transforms every instance's triangles to world space once
 * and builds ONE tree: no ray transform, one tree walk -- the fastest traversal while the tree stays cache-resident, but memory and build
 * time grow with instances x triangles (~600 B per world triangle with the GPU builders' buffers). HRPT_ACCEL_TWO_LEVEL keeps one
 * object-space tree per distinct mesh plus a tree over the instances: memory grows with distinct triangles + instances, hrpt_update_instances rebuilds only the small instance tree (the
 * reference's per-frame TLAS build), traversal pays a ray transform per visited instance. Radiance is identical: hits are still
 * decided in world space on the world-space vertices the flat upload would produce, non-opaque instances (MASK / BLEND / transmissive
 * materials) included: their candidates are visited in the same front-to-back order. Every render / query entry point traverses it: the wavefront
 * pipeline, the validation megakernel (HRPT_FRAME_MEGAKERNEL) and both ray-query kernels (the last two with a private 64-entry stack: deeper
 * structures answer HRPT_ERR_INVALID_ARGUMENT there); hrpt_selftest_bvh is a flat-structure check and answers HRPT_ERR_INVALID_ARGUMENT
 * on such a scene. A scene that cannot be held
 * in this form (an instance whose world matrix has no inverse, now or after a later hrpt_update_instances) is built flat whatever was
 * asked -- HrptBuildInfo::structure tells. AUTO: two-level when the scene has at least 2 M world triangles (16 M if any instance is non-opaque: measured cross-over) and at least 8
 * instances per distinct mesh on average (measured on MI355X, opaque spheres / cylinders of ~400 triangles, 1920x1080, 8 spp, 4 bounces:
 * 4 096 instances 19.3 ms flat vs 16.2 ms two-level, 16 384: 27.7 vs 17.3 ms, 65 536: 44.4 vs 18.4 ms and 17.6 GB vs 30 MB -- the small trees stay in cache).
 * Takes effect at the next hrpt_upload_scene. */
#define HRPT_ACCEL_AUTO      0
#define HRPT_ACCEL_FLAT      1
#define HRPT_ACCEL_TWO_LEVEL 2
int  hrpt_set_acceleration_structure(HrptContext* ctx, int structure);

/* Moving objects: writes instances[0..count) over the scene's instances [firstInstance, firstInstance + count) -- the closed dirty range
 * Renderer::UploadDirtyInstanceTransforms copies into m_InstanceDataBuffer / m_RTInstanceDescBuffer (src/Renderer.cpp:924-967, fed by
 * Scene::Update, src/Scene.cpp:536-556) -- and rebuilds the acceleration structure, the job of TLASRenderer's per-frame
 * buildTopLevelAccelStructFromBuffer (src/CommonRenderers.cpp:234-246). Only m_World (and the unused m_PrevWorld / m_Center / m_Radius)
 * may differ from the uploaded instance; a changed mesh, material or LOD index is HRPT_ERR_INVALID_ARGUMENT. With a GPU builder selected
 * the geometry and all build buffers are already on the device: the call uploads count-independent O(instances) data and runs the build
 * kernels (hrpt_get_build_info reports the rebuild); with the host builder the tree is rebuilt on the host from the library's copy of
 * the scene. Waits for frames in flight, returns when the new tree is in place. The accumulation image is not touched: like
 * PathTracerRenderer::Render's reset on a changed view matrix (src/PathTracerRenderer.cpp:41-50), restarting accumulation
 * (firstAccumulationIndex = 0) is the caller's decision. If the rebuild fails the scene is unusable until hrpt_upload_scene. */
int  hrpt_update_instances(HrptContext* ctx, const HrptPerInstanceData* instances, uint32_t firstInstance, uint32_t count);
/* The same call for SMALL motions: where a GPU builder holds the hierarchy of the previous build (flat structure built by LBVH / PLOC; the
 * instance tree of a two-level structure from 1 024 instances on) the boxes are recomputed bottom-up on that hierarchy instead of the tree
 * being rebuilt -- the refit of a driver's acceleration-structure update (the reference always rebuilds its TLAS,
 * src/CommonRenderers.cpp:234-246; its BLASes are static). No sort and no hierarchy construction: a PLOC tree is refitted in a fraction of its
 * build time and keeps its topology, so its quality follows the motion (large moves: call hrpt_update_instances). Radiance is the same
 * either way (the hit definition does not depend on the tree). Everywhere else (host-built trees, the first call after an upload with the
 * host builder) it IS hrpt_update_instances. HrptBuildInfo::usedBuilder carries HRPT_BVH_BUILDER_REFITTED when the hierarchy was kept. */
int  hrpt_refit_instances(HrptContext* ctx, const HrptPerInstanceData* instances, uint32_t firstInstance, uint32_t count);
/* The other two per-frame uploads of the reference's main loop (src/Renderer.cpp:500-507):
 * hrpt_update_lights replaces the whole light buffer, like SceneLoader::CreateAndUploadLightBuffer when Scene::m_LightsDirty is set
 * (count may differ from the uploaded scene's; at least one light; HrptPathTracerConstants::m_LightCount of later frames must not
 * exceed it). hrpt_update_materials writes materials[0..count) over the material constants [firstMaterial, firstMaterial + count),
 * the closed dirty range of Renderer::UploadDirtyMaterialConstants (src/Renderer.cpp:976-1025; emissive animations mark it,
 * src/Scene.cpp:440-470). Texture indices keep referring to the uploaded texture table. A material that switches between OPAQUE and
 * MASK / BLEND, or the first material with a normal map, changes what the acceleration structure caches per triangle: the call then
 * rebuilds it (as hrpt_update_instances would); anything else is a buffer write. Both wait for frames in flight. */
int  hrpt_update_lights(HrptContext* ctx, const HrptGPULight* lights, uint32_t count);
int  hrpt_update_materials(HrptContext* ctx, const HrptMaterialConstants* materials, uint32_t firstMaterial, uint32_t count);

/* ---- in-process multi-GPU: one context per GPU inside ONE process ------------------------------------
 * Rank i of n has rendered the row band [i*H/n, (i+1)*H/n) of its accumulation image (HrptFrameParams::tile*; H must be a
 * multiple of n, every context the same size). hrpt_allgather sends every band to every other context with
 * hipMemcpyPeerAsync (xGMI between GPUs, a plain device copy when two contexts share a GPU), orders the copies against each
 * context's stream with events, and resolves Output = rgb / a on every context. Asynchronous: follow with hrpt_synchronize on
 * the contexts that are read. The one-process-per-GPU form of the same exchange (torch.distributed / RCCL) is
 * hobbyrenderer_amd/distributed.py. */
int  hrpt_allgather(HrptContext* const* ranks, int n);

/* ---- stand-alone ray queries ("other inline-RT consumers") ------------------------------------------
 * The two queries every inline-ray-tracing pass of the reference is built from, over the uploaded scene:
 *   HRPT_RAYS_CLOSEST  TraceRayStandard (src/shaders/RaytracingCommon.hlsli:138-198): closest hit, MASK candidates alpha-tested,
 *                      BLEND candidates committed stochastically from the ray's own RNG state (returned advanced in HrptRayHit::rng)
 *   HRPT_RAYS_SHADOW   CalculateRTShadow<true> (src/shaders/CommonLighting.hlsli:380-496): origin = shaded point, direction = L,
 *                      tmax = distance to the light; the visibility in [0, 1] comes back in HrptRayHit::t (tmin is ignored:
 *                      the query applies its own 0.01 bias)
 * rays / hits are host arrays unless HRPT_RAYS_DEVICE_POINTERS is set (then both are device pointers and the call is asynchronous
 * on the context stream). */
typedef struct HrptRay    { float origin[3]; float tmin; float direction[3]; float tmax; uint32_t rng; uint32_t pad[3]; } HrptRay;       /* 48 B */
typedef struct HrptRayHit { float t, u, v; uint32_t instance, primitive, hit, rng, pad; } HrptRayHit;                                  /* 32 B */
#define HRPT_RAYS_CLOSEST 0u
#define HRPT_RAYS_SHADOW  1u
#define HRPT_RAYS_DEVICE_POINTERS 0x100u
#define HRPT_RAYS_THREAD_PER_RAY  0x200u   /* testing: the one-thread-per-ray kernel instead of the persistent refilling traversal kernel (same results) */
int  hrpt_trace_rays(HrptContext* ctx, const HrptRay* rays, HrptRayHit* hits, uint64_t count, uint32_t flags);

/* Host read-back (synchronises). bytes must be width*height*16. */
int  hrpt_read_accumulation(HrptContext* ctx, float* rgba, size_t bytes);
int  hrpt_read_output(HrptContext* ctx, float* rgba, size_t bytes);
/* Host write of the accumulation image (resume a progressive render). */
int  hrpt_write_accumulation(HrptContext* ctx, const float* rgba, size_t bytes);
/* Output = accum.rgb / accum.a for every pixel (PathTracer.hlsl:339), e.g. after an all-gather of accumulation tiles. */
int  hrpt_resolve_output(HrptContext* ctx);
/* The same resolve over caller-owned DEVICE images (pixelCount float4 each) on the caller's stream: the consumer of a
 * gathered accumulation image that lives outside the context (pipelined multi-GPU frames, hobbyrenderer_amd/distributed.py).
 * Asynchronous; stream is a hipStream_t (NULL = the default stream). */
int  hrpt_resolve_device(HrptContext* ctx, const float* accumulationDevice, float* outputDevice, uint64_t pixelCount, void* stream);
/* The consumer of column-interleaved shards (HrptFrameParams::stripeCount): `shardsDevice` holds the all-gathered accumulation shards of
 * `ranks` ranks, rank-major -- rank r's block is [height][width / 8 / ranks][8] float4, its k-th column being image column k * ranks + r.
 * Writes Output = rgb / a in image order to outputDevice and, unless accumulationDevice is NULL, the re-assembled accumulation image:
 * one pass over the data instead of re-assembly followed by hrpt_resolve_device. width must be a multiple of 8 * ranks. Asynchronous. */
int  hrpt_resolve_columns_device(HrptContext* ctx, const float* shardsDevice, float* accumulationDevice, float* outputDevice,
                                 uint32_t width, uint32_t height, uint32_t ranks, void* stream);

/* ---- HDR post chain: the consumer of the pass -------------------------------------------------
 * HDRRenderer::Render (src/HDRRenderer.cpp:88-224) over Output (u0 = g_RG_HDRColor in path-tracer mode):
 * luminance histogram (src/shaders/LuminanceHistogram.hlsl), exposure adaptation (ExposureAdaptation.hlsl) or manual
 * exposure (Camera::m_Exposure, src/Camera.cpp:107-108), then Tonemap_PSMain (PBR-Neutral + sRGB OETF) or
 * TonemapHDR_PSMain (scRGB roll-off) of src/shaders/Tonemap.hlsl into a W x H float4 display image. */
typedef struct HrptPostParams {
    uint32_t autoExposure;          /* Renderer::m_EnableAutoExposure (src/Renderer.h:303) */
    float    manualExposure;        /* Camera::m_Exposure = 1 / (2^EV * 1.2); used when autoExposure == 0 */
    float    deltaTimeSeconds;      /* m_FrameTime / 1000 */
    float    adaptationSpeed;       /* Renderer::m_AdaptationSpeed, default 5 */
    float    exposureValueMin;      /* Camera::m_ExposureValueMin, default -7 */
    float    exposureValueMax;      /* Camera::m_ExposureValueMax, default 23 */
    float    exposureCompensation;  /* Camera::m_ExposureCompensation */
    uint32_t hdrDisplay;            /* GraphicRHI::m_bIsHDR: 0 = Tonemap_PSMain, 1 = TonemapHDR_PSMain */
    float    maxDisplayNits;        /* GraphicRHI::m_MaxDisplayNits */
} HrptPostParams;
int  hrpt_post_process(HrptContext* ctx, const HrptPostParams* params);
int  hrpt_read_display(HrptContext* ctx, float* rgba, size_t bytes);           /* W*H*16 bytes */
int  hrpt_get_exposure(HrptContext* ctx, float* exposure, uint32_t histogram256[256]);   /* histogram may be NULL */
int  hrpt_set_exposure(HrptContext* ctx, float exposure);                      /* the persistent exposure buffer, initially 1 */

/* Intra-frame overlap: by default the shadow stage of bounce b runs on a second, library-owned stream next to the traversal of bounce
 * b + 1 (they share no buffer). That fills the tails of a context that renders one frame at a time (-3 % per frame). A host that keeps
 * two frames in flight on two contexts already fills those tails with the other frame; there the fork / join events only cost
 * (-5 % per frame with the overlap off at full size, -13 % for an eighth of the picture): pass 0. Takes effect at the next hrpt_render. */
int  hrpt_set_shadow_overlap(HrptContext* ctx, int enabled);

int  hrpt_get_stats(HrptContext* ctx, HrptStats* out);      /* synchronises; ray counters are cumulative */
int  hrpt_reset_stats(HrptContext* ctx);
/* Device self-test of the acceleration structure: *violations = number of child boxes (2-wide tree and its 4-wide collapse) that do
 * not contain the boxes / triangle vertices below them. 0 for a sound tree; anything else means missed hits. Synchronises. */
int  hrpt_selftest_bvh(HrptContext* ctx, uint64_t* violations);
/* Device self-test: out65536[i] = the kernels' decode of the binary16 bit pattern i (RGBA16F LUT texels). */
int  hrpt_selftest_f16_decode(HrptContext* ctx, float* out65536);
/* out512[i] = the kernels' RGBA8_UNORM channel decode of byte i (i < 256); out512[256 + i] = (float)i / 255.0f computed on the device. */
int  hrpt_selftest_unorm8(HrptContext* ctx, float* out512);

/* Host-side helpers of PathTracerRenderer::Render, exported so that callers in other languages
 * produce the same constants: Halton (src/Utilities.cpp:67-79) and the CB fill (:58-75). */
float hrpt_halton(uint32_t index, uint32_t base);

/* Producer of stand-ins for bin/bruneton/{transmittance,scattering,irradiance}.dat, which the reference loads
 * (src/CommonResources.cpp:519-569) but does not ship: raw float32 RGBA tables of 256*64, 256*128*32 and 64*16 texels from
 * the constants of src/shaders/Atmosphere.hlsli:41-75, following Bruneton's 2017 precomputation: transmittance, single
 * scattering, and `orders` - 1 further scattering orders (scattering density, indirect ground irradiance, multiple
 * scattering; ground albedo 0.1). irradiance may be NULL. hrpt_precompute_atmosphere computes 4 orders (Bruneton's demo value;
 * what the reference's files hold is unknown) on the current HIP device when there is one, else on host threads;
 * the _ex form chooses: orders 1..8 (1 = single scattering only, zero irradiance table), device -1 = host threads, >= 0 = that HIP
 * device, -2 = automatic. The tables are bit-identical whichever executor computed them (one __host__ __device__ source in the
 * arithmetic of hobbyrt/detmath.h). Four orders: ~0.1 s on an MI355X, ~40 s on 8 host threads. */
int  hrpt_precompute_atmosphere(float* transmittance, float* scattering, float* irradiance, int nthreads);
int  hrpt_precompute_atmosphere_ex(float* transmittance, float* scattering, float* irradiance, int orders, int nthreads, int device);
/* Test hook: texels [first, first + count) of one pass (3 scattering density, 4 indirect irradiance, 5 multiple scattering) of order `order`
 * from tables in host memory (3 floats per texel; transmittance / scattering4: 4), by host threads (device < 0) or on HIP device `device`. */
int  hrpt_atmosphere_pass(int pass, int order, uint32_t first, uint32_t count, const float* transmittance, const float* deltaIrradiance,
                          const float* deltaRayleigh, const float* deltaMie, const float* deltaDensity, const float* deltaMultiple,
                          float* scattering4, float* out3, int nthreads, int device);

#ifdef __cplusplus
} /* extern "C" */
#endif

#if defined(__cplusplus)
static_assert(sizeof(HrptVertexQuantized) == 24, "VertexQuantized");
static_assert(sizeof(HrptMeshData) == 164, "MeshData");
static_assert(sizeof(HrptPerInstanceData) == 160, "PerInstanceData");
static_assert(sizeof(HrptMaterialConstants) == 180, "MaterialConstants");
static_assert(sizeof(HrptGPULight) == 64, "GPULight");
static_assert(sizeof(HrptPlanarViewConstants) == 704, "PlanarViewConstants");
static_assert(sizeof(HrptPathTracerConstants) == 768, "PathTracerConstants");
#else
_Static_assert(sizeof(HrptVertexQuantized) == 24, "VertexQuantized");
_Static_assert(sizeof(HrptMeshData) == 164, "MeshData");
_Static_assert(sizeof(HrptPerInstanceData) == 160, "PerInstanceData");
_Static_assert(sizeof(HrptMaterialConstants) == 180, "MaterialConstants");
_Static_assert(sizeof(HrptGPULight) == 64, "GPULight");
_Static_assert(sizeof(HrptPlanarViewConstants) == 704, "PlanarViewConstants");
_Static_assert(sizeof(HrptPathTracerConstants) == 768, "PathTracerConstants");
#endif

#endif /* HOBBYRT_PT_H */
