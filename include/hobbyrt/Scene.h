// hobbyrt/Scene.h -- the part of the reference's Scene (src/Scene.h:65-410, UTF-16 in the reference tree) that
// PathTracerRenderer and its data producers read, with the same member names and nested types so reference-side code
// keeps compiling. NVRHI buffer/AS handles become hobbyrt::BufferHandle / AccelStructHandle (host views: the C ABI
// copies at upload). Host logic mirrored: FinalizeLoadedScene (src/Scene.cpp:216-343), EnsureDefaultDirectionalLight
// (:635-666), GetSunDirection (src/Scene.h:336-346), MaterialConstantsFromMaterial (src/SceneLoader.cpp:1525-1545),
// CreateAndUploadLightBuffer (:2435-2493), BuildAccelerationStructures (src/Scene.cpp:67-214 -> hrpt_upload_scene).
#pragma once

#include <cfloat>
#include <string>
#include <utility>
#include <vector>

#include "../hobbyrt_pt.h"
#include "Camera.h"
#include "Math.h"
#include "srrhi.h"

namespace hobbyrt {

struct BufferHandle { const void* data = nullptr; size_t bytes = 0; explicit operator bool() const { return data != nullptr; } };
struct AccelStructHandle { HrptContext* context = nullptr; explicit operator bool() const { return context != nullptr; } };

class Scene {
public:
    struct Primitive { uint32_t m_VertexOffset = 0, m_VertexCount = 0; int m_MaterialIndex = -1; uint32_t m_MeshDataIndex = 0; };
    struct Mesh { std::vector<Primitive> m_Primitives; Vector3 m_Center; float m_Radius = 0; };
    struct Node {
        std::string m_Name; int m_MeshIndex = -1, m_Parent = -1; std::vector<int> m_Children;
        Matrix m_LocalTransform = Matrix::Identity(), m_WorldTransform = Matrix::Identity();
        Vector3 m_Translation; Quaternion m_Rotation{ 0, 0, 0, 1 }; Vector3 m_Scale{ 1, 1, 1 };
        bool m_IsAnimated = false, m_IsDynamic = false, m_IsDirty = false;
        Vector3 m_Center; float m_Radius = 0; int m_CameraIndex = -1, m_LightIndex = -1;
        std::vector<uint32_t> m_InstanceIndices;
    };
    struct Material {
        srrhi::MaterialConstants m_GPU = []() {   // defaults src/Scene.h:163-178
            srrhi::MaterialConstants c{};
            c.m_BaseColor = Vector4{ 1, 1, 1, 1 }; c.m_EmissiveFactor = Vector4{ 0, 0, 0, 1 }; c.m_RoughnessMetallic = Vector2{ 1, 0 };
            c.m_AlbedoTextureIndex = (uint32_t)srrhi::CommonConsts::DEFAULT_TEXTURE_WHITE;
            c.m_NormalTextureIndex = (uint32_t)srrhi::CommonConsts::DEFAULT_TEXTURE_NORMAL;
            c.m_RoughnessMetallicTextureIndex = (uint32_t)srrhi::CommonConsts::DEFAULT_TEXTURE_PBR;
            c.m_EmissiveTextureIndex = (uint32_t)srrhi::CommonConsts::DEFAULT_TEXTURE_BLACK;
            c.m_AlphaMode = (uint32_t)srrhi::CommonConsts::ALPHA_MODE_OPAQUE; c.m_AlphaCutoff = 0.5f; c.m_IOR = 1.5f;
            c.m_AttenuationDistance = FLT_MAX; c.m_AttenuationColor = Vector3{ 1, 1, 1 };
            return c;
        }();
        std::string m_Name;
        int m_BaseColorTexture = -1, m_NormalTexture = -1, m_MetallicRoughnessTexture = -1, m_EmissiveTexture = -1;
    };
    struct Texture {
        std::string m_Uri; std::vector<uint8_t> m_Pixels; uint32_t m_Width = 0, m_Height = 0;   // decoded texels, all levels (HrptTextureDesc)
        uint32_t m_Format = 0, m_MipCount = 1;   // HRPT_TEXTURE_FORMAT_* (stb images: RGBA8_UNORM, one level; DDS: the file's, src/TextureLoader.cpp:196-213)
        std::string m_SourceUri;            // the image's own URI when m_Uri was switched to a .dds sibling (fallback if that cannot be decoded)
        uint32_t m_BindlessIndex = UINT32_MAX;
        enum SamplerType { Clamp = 0, Wrap = 1 };
        SamplerType m_Sampler = Wrap;
    };
    struct Light {
        std::string m_Name;
        enum Type { Directional, Point, Spot };
        Type m_Type = Directional;
        Vector3 m_Color{ 1, 1, 1 }; float m_Intensity = 1.0f, m_Range = 0.0f, m_Radius = 0.0f;
        float m_SpotInnerConeAngle = 0.0f, m_SpotOuterConeAngle = XM_PIDIV4, m_AngularSize = 0.533f;
        int m_NodeIndex = -1;
    };

    struct Camera {             // src/Scene.h:219-232
        std::string m_Name; ProjectionParams m_Projection; int m_NodeIndex = -1;
        float m_ExposureValue = 10.0f, m_ExposureCompensation = 0.0f, m_ExposureValueMin = -7.0f, m_ExposureValueMax = 23.0f;
    };

    std::vector<Mesh> m_Meshes; std::vector<Node> m_Nodes; std::vector<Material> m_Materials; std::vector<Texture> m_Textures;
    std::vector<Camera> m_Cameras; std::vector<Light> m_Lights;
    int m_SelectedCameraIndex = -1;

    ::hobbyrt::Camera m_Camera;
    srrhi::PlanarViewConstants m_View, m_ViewPrev;

    struct BucketInfo { uint32_t m_BaseIndex = 0, m_Count = 0; };
    BucketInfo m_OpaqueBucket, m_MaskedBucket, m_TransparentBucket;

    // "GPU buffers created for the scene" (src/Scene.h:285-313): host views handed to hrpt_upload_scene
    BufferHandle m_VertexBufferQuantized, m_IndexBuffer, m_MaterialConstantsBuffer, m_MeshDataBuffer, m_LightBuffer, m_InstanceDataBuffer;
    uint32_t m_LightCount = 0;
    AccelStructHandle m_TLAS;

    std::vector<srrhi::VertexQuantized> m_Vertices;   // backing store of m_VertexBufferQuantized
    std::vector<uint32_t> m_Indices;                  // backing store of m_IndexBuffer (global vertex indices)
    std::vector<srrhi::PerInstanceData> m_InstanceData;
    std::vector<srrhi::MeshData> m_MeshData;
    std::vector<srrhi::Meshlet> m_Meshlets;             // carried through the cooked-mesh cache; the path tracer reads LOD-0 indices only
    std::vector<uint32_t> m_MeshletVertices, m_MeshletTriangles;
    std::vector<srrhi::MaterialConstants> m_MaterialConstants;
    std::vector<srrhi::GPULight> m_GPULights;
    // Bruneton LUTs in the float32 layout of bin/bruneton/*.dat (src/CommonResources.cpp:519-569)
    std::vector<float> m_BrunetonTransmittance, m_BrunetonScattering, m_BrunetonIrradiance;

    void FinalizeLoadedScene();
    void EnsureDefaultDirectionalLight();
    void UpdateMaterialsAndCreateConstants();   // MaterialConstantsFromMaterial for every material
    void CreateAndUploadLightBuffer();
    // Replaces BLAS/TLAS build + buffer uploads: validates and uploads everything through the C ABI. Returns an HrptStatus.
    int BuildAccelerationStructures(HrptContext* context);

    // Moving objects. The closed range of m_InstanceData entries whose transform changed since the last upload ({UINT32_MAX, 0} = clean),
    // grown by Scene::Update's "Sync instances" step (src/Scene.cpp:536-556) and consumed by Renderer::UploadDirtyInstanceTransforms
    // (src/Renderer.cpp:915-971). Animation evaluation itself is out of scope (the path tracer pauses animations,
    // src/PathTracerRenderer.cpp:53); SetNodeWorldTransform is the "manual scene mutation" route the reference mentions (:920-923).
    std::pair<uint32_t, uint32_t> m_InstanceDirtyRange{ UINT32_MAX, 0 };
    bool AreInstanceTransformsDirty() const { return m_InstanceDirtyRange.first <= m_InstanceDirtyRange.second; }
    void SetNodeWorldTransform(int nodeIndex, const Matrix& world);
    // Scene::m_LightsDirty (consumed at src/Renderer.cpp:500-504) and the closed material dirty range of the emissive animations
    // (src/Scene.cpp:440-470, consumed by Renderer::UploadDirtyMaterialConstants, src/Renderer.cpp:976-1025)
    bool m_LightsDirty = false;
    std::pair<uint32_t, uint32_t> m_MaterialDirtyRange{ UINT32_MAX, 0 };

    Vector3 GetSunDirection() const;
    void UpdateNodeBoundingSphere(int nodeIndex);   // node sphere = mesh sphere through the node's world transform
    // Renderer::SetCameraFromSceneCamera: position / yaw / pitch / projection of m_Camera from a scene camera's node
    void SetCameraFromSceneCamera(const Camera& sceneCamera);
};

} // namespace hobbyrt
