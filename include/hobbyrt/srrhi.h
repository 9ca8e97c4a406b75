// hobbyrt/srrhi.h -- source-compatible stand-ins for the srrhi-generated C++ headers the reference includes as
// "shaders/srrhi/cpp/{Common,Mesh,Instance,GPULight,PathTracer}.h" (generated from src/shaders/*.sr; the generator
// and its output are not in the reference tree, .gitignore:135). Field names, order and byte layout follow the .sr
// files; every struct is layout-identical to its Hrpt* twin in include/hobbyrt_pt.h (static_asserts below), so a
// Scene's vectors are handed to the C ABI without conversion.
#pragma once

#include "../hobbyrt_pt.h"
#include "Math.h"

namespace srrhi {

using hobbyrt::Matrix; using hobbyrt::Vector2; using hobbyrt::Vector3; using hobbyrt::Vector4;

struct CommonConsts {   // Common.sr:47-140 (the entries this path touches)
    static constexpr float PI = 3.14159265359f;
    static constexpr uint32_t MAX_LOD_COUNT = 8, kMaxMeshletVertices = 64, kMaxMeshletTriangles = 96;
    static constexpr uint32_t TEXFLAG_ALBEDO = 1, TEXFLAG_NORMAL = 2, TEXFLAG_ROUGHNESS_METALLIC = 4, TEXFLAG_EMISSIVE = 8;
    static constexpr int DEFAULT_TEXTURE_BLACK = 0, DEFAULT_TEXTURE_WHITE = 1, DEFAULT_TEXTURE_GRAY = 2, DEFAULT_TEXTURE_NORMAL = 3,
                         DEFAULT_TEXTURE_PBR = 4, BRUNETON_TRANSMITTANCE_TEXTURE = 8, BRUNETON_SCATTERING_TEXTURE = 9,
                         BRUNETON_IRRADIANCE_TEXTURE = 10, DEFAULT_TEXTURE_COUNT = 11;
    static constexpr int ALPHA_MODE_OPAQUE = 0, ALPHA_MODE_MASK = 1, ALPHA_MODE_BLEND = 2;
    static constexpr int RENDERING_MODE_PATH_TRACER = 2;
};

struct VertexQuantized { Vector3 m_Pos; uint32_t m_Normal = 0, m_Uv = 0, m_Tangent = 0; };                      // Mesh.sr:9-15
struct MeshData {                                                                                                // Mesh.sr:17-25
    uint32_t m_LODCount = 0, m_IndexOffsets[8] = {}, m_IndexCounts[8] = {}, m_MeshletOffsets[8] = {}, m_MeshletCounts[8] = {};
    float m_LODErrors[8] = {};
};
struct Meshlet { uint32_t m_CenterRadius[2] = {}, m_VertexOffset = 0, m_TriangleOffset = 0, m_VertexCount = 0, m_TriangleCount = 0, m_ConeAxisAndCutoff = 0; };   // Mesh.sr:27-35
struct PerInstanceData {                                                                                         // Instance.sr:49-65
    Matrix m_World, m_PrevWorld; uint32_t m_MaterialIndex = 0, m_MeshDataIndex = 0; float m_Radius = 0; uint32_t m_LODIndex = 0;
    Vector3 m_Center; uint32_t m_FirstGeometryInstanceIndex = 0;
};
struct MaterialConstants {                                                                                       // Instance.sr:2-46
    Vector4 m_BaseColor, m_EmissiveFactor; Vector2 m_RoughnessMetallic;
    uint32_t m_TextureFlags = 0, m_AlbedoTextureIndex = 0, m_NormalTextureIndex = 0, m_RoughnessMetallicTextureIndex = 0, m_EmissiveTextureIndex = 0;
    uint32_t m_AlbedoSamplerIndex = 0, m_NormalSamplerIndex = 0, m_RoughnessSamplerIndex = 0, m_EmissiveSamplerIndex = 0;
    uint32_t m_AlbedoMinMipIndex = 0, m_NormalMinMipIndex = 0, m_RoughnessMinMipIndex = 0, m_EmissiveMinMipIndex = 0;
    uint32_t m_AlbedoFeedbackIndex = 0, m_NormalFeedbackIndex = 0, m_RoughnessFeedbackIndex = 0, m_EmissiveFeedbackIndex = 0;
    uint32_t m_MinMipDimsX = 0, m_MinMipDimsY = 0, m_AlphaMode = 0;
    float m_AlphaCutoff = 0, m_IOR = 0, m_TransmissionFactor = 0, m_ThicknessFactor = 0, m_AttenuationDistance = 0;
    Vector3 m_AttenuationColor, m_SigmaA; uint32_t m_IsThinSurface = 0; Vector3 m_SigmaS;
};
struct GPULight {                                                                                                // GPULight.sr:1-13
    Vector3 m_Position; float m_Intensity = 0; Vector3 m_Direction; uint32_t m_Type = 0; Vector3 m_Color; float m_Range = 0;
    float m_SpotInnerConeAngle = 0, m_SpotOuterConeAngle = 0, m_Radius = 0, m_CosSunAngularRadius = 0;
};
struct PlanarViewConstants {                                                                                     // Common.sr:17-43
    Matrix m_MatWorldToView, m_MatViewToClip, m_MatWorldToClip, m_MatClipToView, m_MatViewToWorld, m_MatClipToWorld;
    Matrix m_MatViewToClipNoOffset, m_MatWorldToClipNoOffset, m_MatClipToViewNoOffset, m_MatClipToWorldNoOffset;
    Vector2 m_ViewportOrigin, m_ViewportSize, m_ViewportSizeInv, m_PixelOffset, m_ClipToWindowScale, m_ClipToWindowBias;
    Vector4 m_CameraDirectionOrPosition;
};
// cbuffer PathTracerConstants (PathTracer.sr:6-17) with the Set* accessors PathTracerRenderer::Render uses (:58-75)
struct PathTracerConstants {
    PlanarViewConstants m_View; Vector4 m_CameraPos; uint32_t m_LightCount = 0, m_AccumulationIndex = 0, m_FrameIndex = 0, m_MaxBounces = 0;
    Vector2 m_Jitter; float m_Pad0[2] = { 0, 0 }; Vector3 m_SunDirection; float m_CosSunAngularRadius = 0;
    void SetView(const PlanarViewConstants& v) { m_View = v; }
    void SetCameraPos(const Vector4& v) { m_CameraPos = v; }
    void SetLightCount(uint32_t v) { m_LightCount = v; }
    void SetAccumulationIndex(uint32_t v) { m_AccumulationIndex = v; }
    void SetFrameIndex(uint32_t v) { m_FrameIndex = v; }
    void SetMaxBounces(uint32_t v) { m_MaxBounces = v; }
    void SetJitter(const Vector2& v) { m_Jitter = v; }
    void SetSunDirection(const Vector3& v) { m_SunDirection = v; }
    void SetCosSunAngularRadius(float v) { m_CosSunAngularRadius = v; }
};

static_assert(sizeof(VertexQuantized) == sizeof(HrptVertexQuantized) && sizeof(MeshData) == sizeof(HrptMeshData), "layout");
static_assert(sizeof(PerInstanceData) == sizeof(HrptPerInstanceData) && sizeof(MaterialConstants) == sizeof(HrptMaterialConstants), "layout");
static_assert(sizeof(GPULight) == sizeof(HrptGPULight) && sizeof(PlanarViewConstants) == sizeof(HrptPlanarViewConstants), "layout");
static_assert(sizeof(PathTracerConstants) == sizeof(HrptPathTracerConstants), "layout");

} // namespace srrhi
