// SceneUpload.cpp -- the two places where the host-side Scene meets the GPU library (libhobbyrt_pt.so): the hand-over that replaces
// Scene::BuildAccelerationStructures + CreateAndUploadGpuBuffers (/root/reference/src/Scene.cpp:67-214, src/SceneLoader.cpp:2319-2433)
// and the Bruneton LUT producer. Kept apart from Scene.cpp so that the scene-format library (libhobbyrt_scene.so) stays host-only.
#include "../../../include/hobbyrt/ProceduralScenes.h"
#include "../../../include/hobbyrt/Scene.h"

namespace hobbyrt {

int Scene::BuildAccelerationStructures(HrptContext* context)
{
    m_VertexBufferQuantized = { m_Vertices.data(), m_Vertices.size() * sizeof(srrhi::VertexQuantized) };
    m_IndexBuffer = { m_Indices.data(), m_Indices.size() * 4 };
    m_MeshDataBuffer = { m_MeshData.data(), m_MeshData.size() * sizeof(srrhi::MeshData) };
    std::vector<HrptTextureDesc> table((size_t)srrhi::CommonConsts::DEFAULT_TEXTURE_COUNT);
    for (Texture& t : m_Textures) {
        if (t.m_BindlessIndex == UINT32_MAX) t.m_BindlessIndex = (uint32_t)table.size();
        if (table.size() <= t.m_BindlessIndex) table.resize((size_t)t.m_BindlessIndex + 1);
        table[t.m_BindlessIndex] = { t.m_Pixels.data(), t.m_Width, t.m_Height, t.m_Format, t.m_MipCount };
    }
    HrptSceneDesc d{};
    d.vertices = reinterpret_cast<const HrptVertexQuantized*>(m_Vertices.data()); d.vertexCount = (uint32_t)m_Vertices.size();
    d.indices = m_Indices.data(); d.indexCount = (uint32_t)m_Indices.size();
    d.meshData = reinterpret_cast<const HrptMeshData*>(m_MeshData.data()); d.meshDataCount = (uint32_t)m_MeshData.size();
    d.instances = reinterpret_cast<const HrptPerInstanceData*>(m_InstanceData.data()); d.instanceCount = (uint32_t)m_InstanceData.size();
    d.materials = reinterpret_cast<const HrptMaterialConstants*>(m_MaterialConstants.data()); d.materialCount = (uint32_t)m_MaterialConstants.size();
    d.lights = reinterpret_cast<const HrptGPULight*>(m_GPULights.data()); d.lightCount = (uint32_t)m_GPULights.size();
    d.textures = table.data(); d.textureCount = (uint32_t)table.size();
    d.brunetonTransmittance = m_BrunetonTransmittance.data(); d.brunetonScattering = m_BrunetonScattering.data();
    d.brunetonIrradiance = m_BrunetonIrradiance.empty() ? nullptr : m_BrunetonIrradiance.data();
    int rc = hrpt_upload_scene(context, &d);
    if (rc == HRPT_OK) m_TLAS.context = context;
    return rc;
}


int GenerateAtmosphereLuts(Scene& s, int nthreads)
{
    s.m_BrunetonTransmittance.assign(256u * 64u * 4u, 0.0f);
    s.m_BrunetonScattering.assign(256u * 128u * 32u * 4u, 0.0f);
    s.m_BrunetonIrradiance.assign(64u * 16u * 4u, 0.0f);
    return hrpt_precompute_atmosphere(s.m_BrunetonTransmittance.data(), s.m_BrunetonScattering.data(), s.m_BrunetonIrradiance.data(), nthreads);
}

} // namespace hobbyrt
