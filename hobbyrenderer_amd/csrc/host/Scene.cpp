// Scene.cpp -- host logic of the reference Scene that defines what the path tracer sees: instance order, light order,
// default sun, material constants, light packing, and the hand-over to the C ABI. See include/hobbyrt/Scene.h for the
// reference line ranges.
#include "../../../include/hobbyrt/Scene.h"

#include <algorithm>

namespace hobbyrt {

void Scene::EnsureDefaultDirectionalLight()
{
    // directional lights last: Spot(2) < Point(1) < Directional(0) under "a.type > b.type" (src/Scene.cpp:638-641)
    std::stable_sort(m_Lights.begin(), m_Lights.end(), [](const Light& a, const Light& b) { return a.m_Type > b.m_Type; });
    if (m_Lights.empty() || m_Lights.back().m_Type != Light::Directional) {
        Light light; light.m_Name = "Default Directional"; light.m_Type = Light::Directional; light.m_Intensity = 1.0f;
        light.m_NodeIndex = (int)m_Nodes.size();
        m_Lights.push_back(light);
        Node node; node.m_LightIndex = (int)m_Lights.size() - 1;
        node.m_LocalTransform = MatrixRotationPitchYaw(XM_PIDIV4, 0.0f);   // 45 degree pitch, src/Scene.cpp:656-663
        node.m_WorldTransform = node.m_LocalTransform;
        m_Nodes.push_back(node);
    }
}

void Scene::SetNodeWorldTransform(int nodeIndex, const Matrix& world)
{
    // the per-node tail of Scene::Update (src/Scene.cpp:527-556): world transform, bounding sphere, instance records, dirty range
    Node& node = m_Nodes.at((size_t)nodeIndex);
    node.m_WorldTransform = world;
    UpdateNodeBoundingSphere(nodeIndex);
    for (uint32_t instIdx : node.m_InstanceIndices) {
        srrhi::PerInstanceData& inst = m_InstanceData[instIdx];
        inst.m_World = node.m_WorldTransform; inst.m_Center = node.m_Center; inst.m_Radius = node.m_Radius;
        m_InstanceDirtyRange.first = std::min(m_InstanceDirtyRange.first, instIdx);
        m_InstanceDirtyRange.second = std::max(m_InstanceDirtyRange.second, instIdx);
    }
}

Vector3 Scene::GetSunDirection() const
{
    const Light& dirLight = m_Lights.back();
    const Node& node = m_Nodes.at((size_t)dirLight.m_NodeIndex);
    return Normalize(TransformNormal(Vector3(0.0f, 0.0f, -1.0f), node.m_WorldTransform));
}

void Scene::FinalizeLoadedScene()
{
    // bucket order: opaque static/dynamic, masked static/dynamic, transparent static/dynamic (src/Scene.cpp:266-322);
    // a node is dynamic when it carries a light or is animated (:252-256)
    m_InstanceData.clear();
    struct Info { srrhi::PerInstanceData data; int node; };
    std::vector<Info> bucket[6];
    for (int ni = 0; ni < (int)m_Nodes.size(); ++ni) {
        const Node& node = m_Nodes[ni];
        if (node.m_MeshIndex < 0) continue;
        for (const Primitive& prim : m_Meshes[(size_t)node.m_MeshIndex].m_Primitives) {
            srrhi::PerInstanceData inst{};
            inst.m_World = node.m_WorldTransform; inst.m_PrevWorld = node.m_WorldTransform;
            inst.m_MaterialIndex = (uint32_t)prim.m_MaterialIndex; inst.m_MeshDataIndex = prim.m_MeshDataIndex;
            inst.m_Center = node.m_Center; inst.m_Radius = node.m_Radius;
            uint32_t alpha = prim.m_MaterialIndex >= 0 ? m_Materials[(size_t)prim.m_MaterialIndex].m_GPU.m_AlphaMode : (uint32_t)srrhi::CommonConsts::ALPHA_MODE_OPAQUE;
            bool dynamic = node.m_IsAnimated || node.m_LightIndex != -1 || node.m_IsDynamic;
            int b = (alpha == (uint32_t)srrhi::CommonConsts::ALPHA_MODE_OPAQUE ? 0 : alpha == (uint32_t)srrhi::CommonConsts::ALPHA_MODE_MASK ? 2 : 4) + (dynamic ? 1 : 0);
            bucket[b].push_back({ inst, ni });
        }
    }
    m_OpaqueBucket = { 0, (uint32_t)(bucket[0].size() + bucket[1].size()) };
    m_MaskedBucket = { m_OpaqueBucket.m_Count, (uint32_t)(bucket[2].size() + bucket[3].size()) };
    m_TransparentBucket = { m_MaskedBucket.m_BaseIndex + m_MaskedBucket.m_Count, (uint32_t)(bucket[4].size() + bucket[5].size()) };
    for (auto& b : bucket)
        for (const Info& info : b) {
            m_Nodes[(size_t)info.node].m_InstanceIndices.push_back((uint32_t)m_InstanceData.size());
            m_InstanceData.push_back(info.data);
        }
    m_InstanceDataBuffer = { m_InstanceData.data(), m_InstanceData.size() * sizeof(srrhi::PerInstanceData) };
}

void Scene::UpdateMaterialsAndCreateConstants()
{
    m_MaterialConstants.clear();
    for (const Material& mat : m_Materials) {   // MaterialConstantsFromMaterial, src/SceneLoader.cpp:1525-1545
        srrhi::MaterialConstants mc = mat.m_GPU;
        mc.m_TextureFlags = 0;
        if (mat.m_BaseColorTexture != -1) mc.m_TextureFlags |= srrhi::CommonConsts::TEXFLAG_ALBEDO;
        if (mat.m_NormalTexture != -1) mc.m_TextureFlags |= srrhi::CommonConsts::TEXFLAG_NORMAL;
        if (mat.m_MetallicRoughnessTexture != -1) mc.m_TextureFlags |= srrhi::CommonConsts::TEXFLAG_ROUGHNESS_METALLIC;
        if (mat.m_EmissiveTexture != -1) mc.m_TextureFlags |= srrhi::CommonConsts::TEXFLAG_EMISSIVE;
        auto sampler = [&](int tex) { return tex != -1 ? (uint32_t)m_Textures[(size_t)tex].m_Sampler : (uint32_t)Texture::Wrap; };
        mc.m_AlbedoSamplerIndex = sampler(mat.m_BaseColorTexture); mc.m_NormalSamplerIndex = sampler(mat.m_NormalTexture);
        mc.m_RoughnessSamplerIndex = sampler(mat.m_MetallicRoughnessTexture); mc.m_EmissiveSamplerIndex = sampler(mat.m_EmissiveTexture);
        auto bindless = [&](int tex, uint32_t fallback) { return tex != -1 ? m_Textures[(size_t)tex].m_BindlessIndex : fallback; };
        mc.m_AlbedoTextureIndex = bindless(mat.m_BaseColorTexture, mc.m_AlbedoTextureIndex);
        mc.m_NormalTextureIndex = bindless(mat.m_NormalTexture, mc.m_NormalTextureIndex);
        mc.m_RoughnessMetallicTextureIndex = bindless(mat.m_MetallicRoughnessTexture, mc.m_RoughnessMetallicTextureIndex);
        mc.m_EmissiveTextureIndex = bindless(mat.m_EmissiveTexture, mc.m_EmissiveTextureIndex);
        m_MaterialConstants.push_back(mc);
    }
    m_MaterialConstantsBuffer = { m_MaterialConstants.data(), m_MaterialConstants.size() * sizeof(srrhi::MaterialConstants) };
}

void Scene::CreateAndUploadLightBuffer()
{
    m_GPULights.clear();
    for (const Light& light : m_Lights) {   // src/SceneLoader.cpp:2439-2469
        const Node& node = m_Nodes.at((size_t)light.m_NodeIndex);
        srrhi::GPULight gl;
        gl.m_Type = (uint32_t)light.m_Type; gl.m_Color = light.m_Color; gl.m_Intensity = light.m_Intensity; gl.m_Range = light.m_Range;
        gl.m_Radius = light.m_Radius; gl.m_SpotInnerConeAngle = light.m_SpotInnerConeAngle; gl.m_SpotOuterConeAngle = light.m_SpotOuterConeAngle;
        gl.m_CosSunAngularRadius = 1.0f;
        gl.m_Position = Vector3(node.m_WorldTransform._41, node.m_WorldTransform._42, node.m_WorldTransform._43);
        gl.m_Direction = Normalize(Vector3(node.m_WorldTransform._31, node.m_WorldTransform._32, node.m_WorldTransform._33));   // +Z forward
        if (light.m_Type == Light::Directional) gl.m_CosSunAngularRadius = (float)std::cos((double)(light.m_AngularSize * 0.5f * (XM_PI / 180.0f)));
        m_GPULights.push_back(gl);
    }
    m_LightCount = (uint32_t)m_GPULights.size();
    m_LightBuffer = { m_GPULights.data(), m_GPULights.size() * sizeof(srrhi::GPULight) };
}

} // namespace hobbyrt
