// hobbyrt_pt_demo -- minimal driver that takes the reference's route into the pass: build a Scene, instantiate the
// registered renderers, run N frames of ReferencePathTracer mode, write the images. With --dump it also writes every
// input of the boundary (scene arrays, per-frame constants) so the Python tests can feed the SAME bytes to the oracle.
//   hobbyrt_pt_demo --scene cube|cornell | --gltf FILE [--mesh-cache]  --width W --height H --frames N --bounces B --out PREFIX [--dump] [--no-gpu]
//                   [--move-node N DX DY DZ]   translate node N AFTER the scene was uploaded: the frames (and the dump) see the moved scene,
//                                              the library gets it through Renderer::UploadDirtyInstanceTransforms -> hrpt_update_instances
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/hobbyrt/ProceduralScenes.h"
#include "../../../include/hobbyrt/Renderer.h"
#include "../../../include/hobbyrt/SceneLoader.h"

using namespace hobbyrt;

static bool write_file(const std::string& path, const void* data, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok;
}

int main(int argc, char** argv)
{
    std::string scene = "cube", out = "demo", gltf;
    bool meshCache = false;
    uint32_t width = 256, height = 256, frames = 1, bounces = 1;
    bool dump = false, noGpu = false;
    int moveNode = -1; float moveBy[3] = { 0.0f, 0.0f, 0.0f };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--scene") scene = next(); else if (a == "--out") out = next(); else if (a == "--gltf") gltf = next(); else if (a == "--mesh-cache") meshCache = true;
        else if (a == "--width") width = (uint32_t)std::atoi(next()); else if (a == "--height") height = (uint32_t)std::atoi(next());
        else if (a == "--frames") frames = (uint32_t)std::atoi(next()); else if (a == "--bounces") bounces = (uint32_t)std::atoi(next());
        else if (a == "--dump") dump = true; else if (a == "--no-gpu") noGpu = true;
        else if (a == "--move-node") { moveNode = std::atoi(next()); for (float& v : moveBy) v = (float)std::atof(next()); }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    Scene& s = g_Renderer.m_Scene;
    ProjectionParams proj;
    if (!gltf.empty()) {
        // Scene::LoadScene (src/Scene.cpp:9-65): glTF (+ cooked-mesh cache) -> instances, textures, material constants, lights, first camera
        if (!SceneLoader::LoadSceneFile(s, gltf, meshCache)) { std::fprintf(stderr, "%s\n", SceneLoader::LastError()); return 1; }
        for (const std::string& w : SceneLoader::Warnings()) std::fprintf(stderr, "warning: %s\n", w.c_str());
        proj = s.m_Camera.GetProjection();
        if (s.m_Cameras.empty()) proj.aspectRatio = (float)width / (float)height;
    } else if (scene == "cornell") {
        BuildCornellScene(s);
        s.m_Camera.SetPosition(Vector3(0.0f, 1.0f, -3.4f));
        proj.fovY = 40.0f * (XM_PI / 180.0f); proj.aspectRatio = 16.0f / 9.0f;
    } else {
        BuildDefaultCubeScene(s);
        proj.aspectRatio = (float)width / (float)height;
    }
    s.m_Camera.SetProjection(proj);
    // the Bruneton tables are needed by the upload only (four scattering orders: ~0.1 s on the GPU the context is about to use, ~40 s on host threads)
    if (!noGpu && GenerateAtmosphereLuts(s, 0) != HRPT_OK) { std::fprintf(stderr, "LUT generation failed\n"); return 1; }
    g_Renderer.m_PathTracerMaxBounces = bounces;

    HrptContext* ctx = nullptr;
    if (!noGpu) {
        HrptDeviceDesc dd{ 0, HRPT_ABI_VERSION };
        if (hrpt_create(&dd, &ctx) != HRPT_OK) { std::fprintf(stderr, "%s\n", hrpt_last_error(nullptr)); return 1; }
        if (s.BuildAccelerationStructures(ctx) != HRPT_OK) { std::fprintf(stderr, "%s\n", hrpt_last_error(ctx)); return 1; }
    }
    if (moveNode >= 0) {
        if (moveNode >= (int)s.m_Nodes.size()) { std::fprintf(stderr, "--move-node: the scene has %zu nodes\n", s.m_Nodes.size()); return 2; }
        Matrix w = s.m_Nodes[(size_t)moveNode].m_WorldTransform;
        w._41 += moveBy[0]; w._42 += moveBy[1]; w._43 += moveBy[2];
        s.SetNodeWorldTransform(moveNode, w);
        std::printf("node %d moved, dirty instance range [%u, %u]\n", moveNode, s.m_InstanceDirtyRange.first, s.m_InstanceDirtyRange.second);
    }
    if (dump) {
        write_file(out + "_vertices.bin", s.m_Vertices.data(), s.m_Vertices.size() * sizeof(srrhi::VertexQuantized));
        write_file(out + "_indices.bin", s.m_Indices.data(), s.m_Indices.size() * 4);
        write_file(out + "_meshdata.bin", s.m_MeshData.data(), s.m_MeshData.size() * sizeof(srrhi::MeshData));
        write_file(out + "_instances.bin", s.m_InstanceData.data(), s.m_InstanceData.size() * sizeof(srrhi::PerInstanceData));
        write_file(out + "_materials.bin", s.m_MaterialConstants.data(), s.m_MaterialConstants.size() * sizeof(srrhi::MaterialConstants));
        write_file(out + "_lights.bin", s.m_GPULights.data(), s.m_GPULights.size() * sizeof(srrhi::GPULight));
        srrhi::PlanarViewConstants v{};
        s.m_Camera.FillPlanarViewConstants(v, (float)width, (float)height);
        write_file(out + "_view.bin", &v, sizeof v);
        Vector3 sun = s.GetSunDirection(), cam = s.m_Camera.GetPosition();
        float misc[8] = { sun.x, sun.y, sun.z, cam.x, cam.y, cam.z, s.m_Lights.back().m_AngularSize, 0.0f };
        write_file(out + "_misc.bin", misc, sizeof misc);
    }
    if (noGpu) return 0;

    GraphicRHI rhi; rhi.m_SwapchainExtent = { width, height }; rhi.m_Context = ctx;
    if (g_Renderer.Initialize(&rhi) != HRPT_OK) { std::fprintf(stderr, "%s\n", hrpt_last_error(ctx)); return 1; }
    std::printf("renderer: %s\n", RendererRegistry::GetRenderer("PathTracerRenderer")->GetName());
    for (uint32_t f = 0; f < frames; ++f)
        if (g_Renderer.RunPathTracerFrame() != HRPT_OK) { std::fprintf(stderr, "frame %u: %s\n", f, hrpt_last_error(ctx)); return 1; }
    std::vector<float> acc((size_t)width * height * 4), img((size_t)width * height * 4);
    if (hrpt_read_accumulation(ctx, acc.data(), acc.size() * 4) != HRPT_OK || hrpt_read_output(ctx, img.data(), img.size() * 4) != HRPT_OK) {
        std::fprintf(stderr, "%s\n", hrpt_last_error(ctx)); return 1;
    }
    write_file(out + "_accumulation.bin", acc.data(), acc.size() * 4);
    write_file(out + "_output.bin", img.data(), img.size() * 4);
    HrptStats st{};
    hrpt_get_stats(ctx, &st);
    std::printf("frames %u, closest rays %llu, shadow rays %llu, last render %.3f ms\n", frames, (unsigned long long)st.closestRays,
                (unsigned long long)st.shadowRays, st.lastRenderMs);
    hrpt_destroy(ctx);
    return 0;
}
