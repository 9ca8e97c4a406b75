// PathTracerRenderer.cpp -- the reference's path-tracer pass plugin (/root/reference/src/PathTracerRenderer.cpp:8-111)
// with its D3D12/NVRHI back end replaced by the C ABI of include/hobbyrt_pt.h. Same class shape, same registration
// key ("PathTracerRenderer"), same GetName() ("ReferencePathTracer"), same per-frame host logic:
//   Setup  :14-29   declares the persistent RGBA32F accumulation texture at swapchain size, registers the HDR write
//   Render :31-106  resets accumulation when world->clip changed (memcmp, :41-50), pauses animation (:53), fills
//                   PathTracerConstants (:58-75: view, camera position, light count, accumulation / frame index, max
//                   bounces, Halton(index+1,{2,3})-0.5 jitter, sun direction, cos of half the sun's angular size),
//                   then -- instead of writeBuffer + binding set + dispatch(ceil(W/8), ceil(H/8), 1) (:77-103) --
//                   hrpt_render for exactly this accumulation index; ++m_AccumulationIndex (:105).
#include <cmath>
#include <cstring>

#include "../../../include/hobbyrt/Renderer.h"

namespace hobbyrt {

float Halton(uint32_t index, uint32_t base) { return hrpt_halton(index, base); }   // src/Utilities.cpp:67-79

class PathTracerRenderer : public IRenderer
{
    RGTextureHandle m_AccumulationBuffer;
    uint32_t m_AccumulationIndex = 0;

public:
    bool Setup(RenderGraph& renderGraph) override
    {
        RGTextureDesc desc;
        desc.m_NvrhiDesc.width = g_Renderer.m_RHI->m_SwapchainExtent.x;
        desc.m_NvrhiDesc.height = g_Renderer.m_RHI->m_SwapchainExtent.y;
        desc.m_NvrhiDesc.format = nvrhi::Format::RGBA32_FLOAT;
        desc.m_NvrhiDesc.isUAV = true;
        desc.m_NvrhiDesc.debugName = "AccumulationBuffer";
        desc.m_NvrhiDesc.initialState = nvrhi::ResourceStates::UnorderedAccess;
        renderGraph.DeclarePersistentTexture(desc, m_AccumulationBuffer);
        renderGraph.WriteTexture(g_RG_HDRColor);
        return true;
    }

    void Render(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph) override
    {
        nvrhi::TextureHandle hdrColor = renderGraph.GetTexture(g_RG_HDRColor, RGResourceAccessMode::Write);
        nvrhi::TextureHandle accumBuffer = renderGraph.GetTexture(m_AccumulationBuffer, RGResourceAccessMode::Write);
        const nvrhi::TextureDesc& hdrDesc = hdrColor->getDesc();
        (void)accumBuffer;

        // camera change detection
        bool reset = std::memcmp(&g_Renderer.m_Scene.m_View.m_MatWorldToClipNoOffset, &g_Renderer.m_Scene.m_ViewPrev.m_MatWorldToClipNoOffset, sizeof(Matrix)) != 0;
        if (reset) m_AccumulationIndex = 0;

        g_Renderer.m_EnableAnimations = false;   // pause animations

        srrhi::PathTracerConstants cb;
        cb.SetView(g_Renderer.m_Scene.m_View);
        const Vector3 camPos = g_Renderer.m_Scene.m_Camera.GetPosition();
        cb.SetCameraPos(Vector4{ camPos.x, camPos.y, camPos.z, 1.0f });
        cb.SetLightCount(g_Renderer.m_Scene.m_LightCount);
        cb.SetAccumulationIndex(m_AccumulationIndex);
        cb.SetFrameIndex(g_Renderer.m_FrameNumber);
        cb.SetMaxBounces(g_Renderer.m_PathTracerMaxBounces);
        cb.SetJitter(Vector2{ Halton(m_AccumulationIndex + 1, 2) - 0.5f, Halton(m_AccumulationIndex + 1, 3) - 0.5f });
        cb.SetSunDirection(g_Renderer.m_Scene.GetSunDirection());
        {
            // the last light is the directional light (EnsureDefaultDirectionalLight); 0.533 degrees = the real sun
            const float angularSizeDeg = !g_Renderer.m_Scene.m_Lights.empty() ? g_Renderer.m_Scene.m_Lights.back().m_AngularSize : 0.533f;
            const float halfAngleRad = angularSizeDeg * 0.5f * (XM_PI / 180.0f);
            cb.SetCosSunAngularRadius(std::cos(halfAngleRad));
        }

        // writeBuffer(cb) + PathTracerInputs{TLAS, Lights, Instances, MeshData, Materials, Indices, Vertices, Output, Accumulation}
        // + dispatch: the scene side was bound by Scene::BuildAccelerationStructures (hrpt_upload_scene); one call renders
        // this accumulation index over the whole viewport.
        HrptFrameParams params{};
        static_assert(sizeof(params.constants) == sizeof(cb), "PathTracerConstants layout");
        std::memcpy(&params.constants, &cb, sizeof(cb));
        params.accumCount = 1;
        params.flags = HRPT_FRAME_DEFAULT;
        (void)hdrDesc;   // DivideAndRoundUp(hdrDesc.width, 8) x DivideAndRoundUp(hdrDesc.height, 8) groups in the reference
        g_Renderer.m_LastStatus = hrpt_render(commandList->context, &params);

        m_AccumulationIndex++;
    }

    const char* GetName() const override { return "ReferencePathTracer"; }
};

REGISTER_RENDERER(PathTracerRenderer);

} // namespace hobbyrt
