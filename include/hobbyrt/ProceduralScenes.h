// hobbyrt/ProceduralScenes.h -- procedural Scene inputs on the C++ host side.
//  - GenerateDefaultCube: the fixture of /root/reference/src/ProceduralDefaultCube.h:9-21 (24 vertices, 36 indices, the
//    same vertex order, quantisation and MeshData; meshlet data is not produced: the path tracer never reads it).
//  - BuildDefaultCubeScene / BuildCornellScene: BASELINE.json configs 1 and 2 as hobbyrt::Scene objects (the same scenes
//    hobbyrenderer_amd/scenes.py builds for the Python harness).
#pragma once

#include <vector>

#include "Scene.h"

namespace hobbyrt {

struct ProceduralCubeData {
    std::vector<srrhi::VertexQuantized> m_Vertices;   // 4 per face x 6 faces
    std::vector<uint32_t> m_Indices;                  // 6 per face x 6 faces, zero-based
    srrhi::MeshData m_MeshData;                       // LODCount = 1, zero-based offsets
};
ProceduralCubeData GenerateDefaultCube();

// meshoptimizer's inline quantisers as used by the reference (src/ProceduralDefaultCube.cpp:67-82)
int QuantizeSnorm(float v, int bits);
unsigned short QuantizeHalf(float v);
srrhi::VertexQuantized QuantizeVertex(const float pos[3], const float normal[3], const float uv[2], const float tangent[3], float tangentW);

void BuildDefaultCubeScene(Scene& scene);                       // config 1
void BuildCornellScene(Scene& scene);                           // config 2
// fills scene.m_Bruneton* with hrpt_precompute_atmosphere
int GenerateAtmosphereLuts(Scene& scene, int nthreads);

} // namespace hobbyrt
