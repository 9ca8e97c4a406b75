// Renderer.cpp -- the slice of the reference's frame loop that reaches the path-tracer pass:
// renderer instantiation from the registry (/root/reference/src/Renderer.cpp:346-354), per-frame view update
// (:510-519) and ScheduleRenderer(PathTracerRenderer) -> Setup + Render (:1278-1281, src/RenderGraph.cpp:286,345).
#include "../../../include/hobbyrt/Renderer.h"

#include <cstring>

namespace hobbyrt {

Renderer g_Renderer;
RGTextureHandle g_RG_HDRColor{ 0 };

void RenderGraph::DeclarePersistentTexture(const RGTextureDesc& desc, RGTextureHandle& handle)
{
    if (handle.id >= 0 && handle.id < (int)m_Textures.size()) { m_Textures[(size_t)handle.id]->desc = desc.m_NvrhiDesc; return; }
    auto t = std::make_unique<nvrhi::Texture>(); t->desc = desc.m_NvrhiDesc;
    m_Textures.push_back(std::move(t));
    handle.id = (int)m_Textures.size() - 1;
}
nvrhi::TextureHandle RenderGraph::GetTexture(RGTextureHandle handle, RGResourceAccessMode) const
{
    return (handle.id >= 0 && handle.id < (int)m_Textures.size()) ? m_Textures[(size_t)handle.id].get() : nullptr;
}

int Renderer::Initialize(GraphicRHI* rhi)
{
    m_RHI = rhi;
    m_Renderers.clear();
    for (const auto& c : RendererRegistry::GetCreators()) m_Renderers.push_back(c.second());
    // HDR colour target (ClearRenderer declares it RGBA32_FLOAT in path-tracer mode, src/CommonRenderers.cpp:49-60)
    RGTextureDesc hdr; hdr.m_NvrhiDesc.width = rhi->m_SwapchainExtent.x; hdr.m_NvrhiDesc.height = rhi->m_SwapchainExtent.y; hdr.m_NvrhiDesc.debugName = "HDRColor";
    RGTextureHandle h; m_RenderGraph.DeclarePersistentTexture(hdr, h); g_RG_HDRColor = h;
    int rc = hrpt_resize(rhi->m_Context, rhi->m_SwapchainExtent.x, rhi->m_SwapchainExtent.y);
    if (rc != HRPT_OK) return rc;
    // first frame: m_ViewPrev differs from m_View so the pass starts at accumulation index 0
    m_Scene.m_ViewPrev = srrhi::PlanarViewConstants{};
    return HRPT_OK;
}

int Renderer::UploadDirtyInstanceTransforms()
{
    if (!m_Scene.AreInstanceTransformsDirty()) return HRPT_OK;
    const uint32_t startIdx = m_Scene.m_InstanceDirtyRange.first, count = m_Scene.m_InstanceDirtyRange.second - startIdx + 1;
    m_Scene.m_InstanceDirtyRange = { UINT32_MAX, 0 };       // "always reset after upload so the range never persists into the next frame"
    if ((size_t)startIdx + count > m_Scene.m_InstanceData.size()) return HRPT_ERR_INVALID_ARGUMENT;   // the reference asserts (:932-938)
    return hrpt_update_instances(m_RHI->m_Context, reinterpret_cast<const HrptPerInstanceData*>(m_Scene.m_InstanceData.data() + startIdx), startIdx, count);
}

int Renderer::UploadDirtyMaterialConstants()
{
    if (m_Scene.m_Materials.empty() || m_Scene.m_MaterialDirtyRange.first > m_Scene.m_MaterialDirtyRange.second) return HRPT_OK;
    const uint32_t firstMat = m_Scene.m_MaterialDirtyRange.first, count = m_Scene.m_MaterialDirtyRange.second - firstMat + 1;
    m_Scene.m_MaterialDirtyRange = { UINT32_MAX, 0 };
    if ((size_t)firstMat + count > m_Scene.m_Materials.size()) return HRPT_ERR_INVALID_ARGUMENT;     // the reference asserts (:996-1000)
    m_Scene.UpdateMaterialsAndCreateConstants();             // MaterialConstantsFromMaterial; the library gets only the dirty range
    return hrpt_update_materials(m_RHI->m_Context, reinterpret_cast<const HrptMaterialConstants*>(m_Scene.m_MaterialConstants.data() + firstMat), firstMat, count);
}

int Renderer::RunPathTracerFrame()
{
    if (int r = UploadDirtyInstanceTransforms()) { m_LastStatus = r; return r; }   // before the renderers run, src/Renderer.cpp:497-498
    if (m_Scene.m_LightsDirty) {                                                    // :500-504
        m_Scene.CreateAndUploadLightBuffer();
        m_Scene.m_LightsDirty = false;
        if (int r = hrpt_update_lights(m_RHI->m_Context, reinterpret_cast<const HrptGPULight*>(m_Scene.m_GPULights.data()), (uint32_t)m_Scene.m_GPULights.size())) { m_LastStatus = r; return r; }
    }
    if (int r = UploadDirtyMaterialConstants()) { m_LastStatus = r; return r; }     // :507
    m_Scene.m_ViewPrev = m_Scene.m_View;
    m_Scene.m_Camera.FillPlanarViewConstants(m_Scene.m_View, (float)m_RHI->m_SwapchainExtent.x, (float)m_RHI->m_SwapchainExtent.y);
    if (m_FrameNumber == 0) m_Scene.m_ViewPrev = srrhi::PlanarViewConstants{};
    IRenderer* pt = RendererRegistry::GetRenderer("PathTracerRenderer");   // looked up by name, src/Renderer.cpp:1280
    nvrhi::CommandList cmd; cmd.context = m_RHI->m_Context;
    m_LastStatus = HRPT_OK;
    if (pt->Setup(m_RenderGraph)) pt->Render(&cmd, m_RenderGraph);
    ++m_FrameNumber;
    return m_LastStatus;
}

} // namespace hobbyrt
