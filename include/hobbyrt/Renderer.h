// hobbyrt/Renderer.h -- the Renderer plugin surface of the reference (/root/reference/src/Renderer.h:17-86):
// IRenderer, RendererRegistry, REGISTER_RENDERER, and the slice of `struct Renderer` / RenderGraph / nvrhi handles
// PathTracerRenderer touches (src/PathTracerRenderer.cpp:14-106). D3D12/NVRHI objects become thin types owned here:
// a command list is the context's HIP stream, a texture is one of the two RGBA32F images of the HrptContext.
#pragma once

#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../hobbyrt_pt.h"
#include "Scene.h"

namespace hobbyrt {

namespace nvrhi {
enum class Format { RGBA32_FLOAT };
enum class ResourceStates { UnorderedAccess };
struct TextureDesc { uint32_t width = 0, height = 0; Format format = Format::RGBA32_FLOAT; bool isUAV = false; const char* debugName = ""; ResourceStates initialState = ResourceStates::UnorderedAccess; };
struct Texture { TextureDesc desc; void* devicePtr = nullptr; const TextureDesc& getDesc() const { return desc; } };
using TextureHandle = Texture*;
struct CommandList { HrptContext* context = nullptr; };
using CommandListHandle = CommandList*;
} // namespace nvrhi

struct RGTextureHandle { int id = -1; };
struct RGTextureDesc { nvrhi::TextureDesc m_NvrhiDesc; };
enum class RGResourceAccessMode { Read, Write };

// Two textures only: the HDR colour target (u0) and persistent textures declared by passes (the accumulation, u1).
class RenderGraph {
public:
    void DeclarePersistentTexture(const RGTextureDesc& desc, RGTextureHandle& handle);
    void WriteTexture(RGTextureHandle handle) { (void)handle; }
    nvrhi::TextureHandle GetTexture(RGTextureHandle handle, RGResourceAccessMode mode) const;
    std::vector<std::unique_ptr<nvrhi::Texture>> m_Textures;
};

class IRenderer {
public:
    virtual ~IRenderer() = default;
    virtual void Initialize() {}
    virtual void PostSceneLoad() {}
    virtual bool Setup(RenderGraph& renderGraph) { (void)renderGraph; return false; }
    virtual void Render(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph) { (void)commandList; (void)renderGraph; }
    virtual const char* GetName() const { return "Unnamed Renderer"; }
    virtual bool IsBasePassRenderer() const { return false; }
    float m_CPUTime = 0.0f, m_GPUTime = 0.0f;
    bool m_bPassEnabled = false, m_bClearOnNextRender = false;
};

class RendererRegistry {
public:
    using Creator = std::function<std::shared_ptr<IRenderer>()>;
    static void RegisterRenderer(const char* name, Creator creator) { Creators().push_back({ name, creator }); }
    static const std::vector<std::pair<const char*, Creator>>& GetCreators() { return Creators(); }
    static IRenderer* GetRenderer(const char* name) { return Renderers().at(name); }
    static void SetRenderer(const char* name, IRenderer* renderer) { Renderers()[name] = renderer; }
private:
    static std::vector<std::pair<const char*, Creator>>& Creators() { static std::vector<std::pair<const char*, Creator>> v; return v; }
    static std::unordered_map<std::string, IRenderer*>& Renderers() { static std::unordered_map<std::string, IRenderer*> m; return m; }
};

#define REGISTER_RENDERER(ClassName)                                             \
    static bool s_##ClassName##Registered = []() {                               \
        ::hobbyrt::RendererRegistry::RegisterRenderer(#ClassName, []() {         \
            auto renderer = std::make_shared<ClassName>();                        \
            ::hobbyrt::RendererRegistry::SetRenderer(#ClassName, renderer.get()); \
            return std::shared_ptr<::hobbyrt::IRenderer>(renderer);               \
        });                                                                       \
        return true;                                                              \
    }();

struct GraphicRHI { Vector2U m_SwapchainExtent; HrptContext* m_Context = nullptr; };

// The members of the reference's `struct Renderer` (src/Renderer.h:105-463) that the path-tracer pass reads.
struct Renderer {
    Scene m_Scene;
    GraphicRHI* m_RHI = nullptr;
    uint32_t m_FrameNumber = 0;
    uint32_t m_PathTracerMaxBounces = 8;     // src/Renderer.h:299
    bool m_EnableAnimations = true;
    RenderGraph m_RenderGraph;
    std::vector<std::shared_ptr<IRenderer>> m_Renderers;

    // InitializeGPUStack's renderer instantiation (src/Renderer.cpp:346-354) + resize of the two images
    int Initialize(GraphicRHI* rhi);
    // One frame of ReferencePathTracer mode (src/Renderer.cpp:1276-1281 + camera update :510-519): returns HrptStatus
    int RunPathTracerFrame();
    // src/Renderer.cpp:915-971: hands the scene's dirty instance range to the library (hrpt_update_instances: instance records +
    // acceleration-structure rebuild, the reference's writeBuffer pair + TLASRenderer) and resets the range. Returns an HrptStatus.
    int UploadDirtyInstanceTransforms();
    // src/Renderer.cpp:976-1025: MaterialConstantsFromMaterial for the dirty range -> hrpt_update_materials; resets the range
    int UploadDirtyMaterialConstants();
    int m_LastStatus = 0;
};
extern Renderer g_Renderer;
extern RGTextureHandle g_RG_HDRColor;

} // namespace hobbyrt
