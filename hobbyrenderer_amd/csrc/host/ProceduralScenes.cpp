// ProceduralScenes.cpp -- see include/hobbyrt/ProceduralScenes.h.
#include "../../../include/hobbyrt/ProceduralScenes.h"

#include <cmath>
#include <cstring>

namespace hobbyrt {

int QuantizeSnorm(float v, int bits)   // meshopt_quantizeSnorm
{
    const float scale = (float)((1 << (bits - 1)) - 1);
    float round = (v >= 0.0f ? 0.5f : -0.5f);
    v = (v >= -1.0f) ? v : -1.0f;
    v = (v <= 1.0f) ? v : 1.0f;
    return (int)(v * scale + round);
}

unsigned short QuantizeHalf(float v)   // meshopt_quantizeHalf: round half up on the magnitude, flush denormals, saturate to inf
{
    uint32_t ui; std::memcpy(&ui, &v, 4);
    int s = (int)((ui >> 16) & 0x8000u);
    int em = (int)(ui & 0x7fffffffu);
    int h = (em - (112 << 23) + (1 << 12)) >> 13;
    h = (em < (113 << 23)) ? 0 : h;
    h = (em >= (143 << 23)) ? 0x7c00 : h;
    h = (em > (255 << 23)) ? 0x7e00 : h;
    return (unsigned short)(s | h);
}

srrhi::VertexQuantized QuantizeVertex(const float pos[3], const float normal[3], const float uv[2], const float tangent[3], float tangentW)
{
    srrhi::VertexQuantized vq{};
    vq.m_Pos = Vector3(pos[0], pos[1], pos[2]);
    for (int k = 0; k < 3; ++k) vq.m_Normal |= (uint32_t)(QuantizeSnorm(normal[k], 10) + 511) << (10 * k);
    if (tangentW < 0.0f) vq.m_Normal |= 1u << 30;
    vq.m_Uv = (uint32_t)QuantizeHalf(uv[0]) | ((uint32_t)QuantizeHalf(uv[1]) << 16);
    // octahedral tangent, 8 bits per component
    float sum = std::fabs(tangent[0]) + std::fabs(tangent[1]) + std::fabs(tangent[2]);
    if (sum > 1e-6f) {
        float ox, oy;
        if (tangent[2] >= 0.0f) { ox = tangent[0] / sum; oy = tangent[1] / sum; }
        else {
            ox = (1.0f - std::fabs(tangent[1] / sum)) * (tangent[0] >= 0.0f ? 1.0f : -1.0f);
            oy = (1.0f - std::fabs(tangent[0] / sum)) * (tangent[1] >= 0.0f ? 1.0f : -1.0f);
        }
        vq.m_Tangent = (uint32_t)(QuantizeSnorm(ox, 8) + 127) | ((uint32_t)(QuantizeSnorm(oy, 8) + 127) << 8);
    }
    return vq;
}

namespace {
// A cube face is spanned by its tangent (u direction) and a v direction: corner = N/2 + (u - 1/2) T + (v - 1/2) V.
// Side faces run v downwards (V = -Y) and start at (u,v) = (0,1); caps run along +-Z and start at (0,0). The order
// reproduces the 24 vertices of the reference fixture (LH, CCW seen from outside), src/ProceduralDefaultCube.cpp:19-57.
struct Face { float n[3], t[3], v[3]; bool cap; };
const Face kFaces[6] = {
    { { 1, 0, 0 }, { 0, 0, -1 }, { 0, -1, 0 }, false }, { { -1, 0, 0 }, { 0, 0, 1 }, { 0, -1, 0 }, false },
    { { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }, true },    { { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, -1 }, true },
    { { 0, 0, 1 }, { 1, 0, 0 }, { 0, -1, 0 }, false },  { { 0, 0, -1 }, { -1, 0, 0 }, { 0, -1, 0 }, false },
};
const float kSideUv[4][2] = { { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 } };
const float kCapUv[4][2] = { { 0, 0 }, { 0, 1 }, { 1, 1 }, { 1, 0 } };

void AppendFaces(const Face* faces, int count, float yShift, std::vector<srrhi::VertexQuantized>& verts, std::vector<uint32_t>& idx)
{
    for (int f = 0; f < count; ++f) {
        const Face& fc = faces[f];
        uint32_t base = (uint32_t)verts.size();
        for (int c = 0; c < 4; ++c) {
            const float* uv = fc.cap ? kCapUv[c] : kSideUv[c];
            float p[3];
            for (int k = 0; k < 3; ++k) p[k] = 0.5f * fc.n[k] + (uv[0] - 0.5f) * fc.t[k] + (uv[1] - 0.5f) * fc.v[k];
            p[1] += yShift;
            verts.push_back(QuantizeVertex(p, fc.n, uv, fc.t, 1.0f));
        }
        const uint32_t tri[6] = { 0, 1, 2, 0, 2, 3 };
        for (uint32_t t : tri) idx.push_back(base + t);
    }
}

uint32_t AddMesh(Scene& s, const std::vector<srrhi::VertexQuantized>& v, const std::vector<uint32_t>& idx)
{
    srrhi::MeshData md{};
    md.m_LODCount = 1; md.m_IndexOffsets[0] = (uint32_t)s.m_Indices.size(); md.m_IndexCounts[0] = (uint32_t)idx.size();
    uint32_t vbase = (uint32_t)s.m_Vertices.size();
    s.m_Vertices.insert(s.m_Vertices.end(), v.begin(), v.end());
    for (uint32_t i : idx) s.m_Indices.push_back(i + vbase);
    s.m_MeshData.push_back(md);
    Scene::Mesh mesh; Scene::Primitive prim; prim.m_VertexOffset = vbase; prim.m_VertexCount = (uint32_t)v.size();
    prim.m_MeshDataIndex = (uint32_t)s.m_MeshData.size() - 1;
    mesh.m_Primitives.push_back(prim);
    s.m_Meshes.push_back(mesh);
    return (uint32_t)s.m_Meshes.size() - 1;
}

int AddMaterial(Scene& s, float r, float g, float b, float er = 0, float eg = 0, float eb = 0)
{
    Scene::Material m; m.m_GPU.m_BaseColor = Vector4(r, g, b, 1.0f); m.m_GPU.m_EmissiveFactor = Vector4(er, eg, eb, 1.0f);
    s.m_Materials.push_back(m);
    return (int)s.m_Materials.size() - 1;
}

// node with world = S * R * T (row-vector convention); one mesh per node, material set on a private copy of the mesh
void AddNode(Scene& s, uint32_t meshTemplate, int material, const double scale[3], const double rot[3][3], const double trans[3])
{
    Scene::Mesh mesh = s.m_Meshes[meshTemplate];
    mesh.m_Primitives[0].m_MaterialIndex = material;
    s.m_Meshes.push_back(mesh);
    Scene::Node node; node.m_MeshIndex = (int)s.m_Meshes.size() - 1;
    Matrix w = Matrix::Identity();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) w.m[i][j] = (float)(scale[i] * rot[i][j]);
    for (int j = 0; j < 3; ++j) w.m[3][j] = (float)trans[j];
    node.m_LocalTransform = w; node.m_WorldTransform = w; node.m_Radius = 1.0f;
    s.m_Nodes.push_back(node);
}
const double kRotPY[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, kRotNY[3][3] = { { -1, 0, 0 }, { 0, -1, 0 }, { 0, 0, 1 } };
const double kRotPX[3][3] = { { 0, -1, 0 }, { 1, 0, 0 }, { 0, 0, 1 } }, kRotNX[3][3] = { { 0, 1, 0 }, { -1, 0, 0 }, { 0, 0, 1 } };
const double kRotPZ[3][3] = { { 1, 0, 0 }, { 0, 0, 1 }, { 0, -1, 0 } }, kRotNZ[3][3] = { { 1, 0, 0 }, { 0, 0, -1 }, { 0, 1, 0 } };

void Finish(Scene& s)
{
    s.EnsureDefaultDirectionalLight();
    s.FinalizeLoadedScene();
    s.UpdateMaterialsAndCreateConstants();
    s.CreateAndUploadLightBuffer();
}
} // namespace

ProceduralCubeData GenerateDefaultCube()
{
    ProceduralCubeData out;
    AppendFaces(kFaces, 6, 0.0f, out.m_Vertices, out.m_Indices);
    out.m_MeshData = {};
    out.m_MeshData.m_LODCount = 1; out.m_MeshData.m_IndexCounts[0] = (uint32_t)out.m_Indices.size();
    return out;
}

void BuildDefaultCubeScene(Scene& s)
{
    ProceduralCubeData cube = GenerateDefaultCube();
    uint32_t mesh = AddMesh(s, cube.m_Vertices, cube.m_Indices);
    s.m_Materials.push_back(Scene::Material());
    const double one[3] = { 1, 1, 1 }, zero[3] = { 0, 0, 0 };
    AddNode(s, mesh, 0, one, kRotPY, zero);
    Finish(s);
}

void BuildCornellScene(Scene& s)
{
    std::vector<srrhi::VertexQuantized> qv; std::vector<uint32_t> qi;
    AppendFaces(&kFaces[2], 1, -0.5f, qv, qi);                 // the +Y face lowered to y = 0: floor quad
    uint32_t quad = AddMesh(s, qv, qi);
    ProceduralCubeData cube = GenerateDefaultCube();
    uint32_t box = AddMesh(s, cube.m_Vertices, cube.m_Indices);
    int white = AddMaterial(s, 0.73f, 0.73f, 0.73f), red = AddMaterial(s, 0.65f, 0.05f, 0.05f), green = AddMaterial(s, 0.12f, 0.45f, 0.15f);
    int light = AddMaterial(s, 0.78f, 0.78f, 0.78f, 17.0f, 12.0f, 4.0f);
    const double wall[3] = { 2, 1, 5 }, end[3] = { 2, 1, 2 }, zc = -1.5;
    const double tFloor[3] = { 0, 0, zc }, tCeil[3] = { 0, 2, zc }, tLeft[3] = { -1, 1, zc }, tRight[3] = { 1, 1, zc }, tBack[3] = { 0, 1, 1 }, tFront[3] = { 0, 1, -4 };
    AddNode(s, quad, white, wall, kRotPY, tFloor); AddNode(s, quad, white, wall, kRotNY, tCeil);
    AddNode(s, quad, red, wall, kRotPX, tLeft);    AddNode(s, quad, green, wall, kRotNX, tRight);
    AddNode(s, quad, white, end, kRotNZ, tBack);   AddNode(s, quad, white, end, kRotPZ, tFront);
    const double c = 0.96, sn = 0.28;   // 7-24-25 rotation about Y
    const double rA[3][3] = { { c, 0, -sn }, { 0, 1, 0 }, { sn, 0, c } }, rB[3][3] = { { c, 0, sn }, { 0, 1, 0 }, { -sn, 0, c } };
    const double sA[3] = { 0.6, 0.6, 0.6 }, sB[3] = { 0.6, 1.2, 0.6 }, tA[3] = { 0.35, 0.3, -0.15 }, tB[3] = { -0.35, 0.6, 0.35 };
    AddNode(s, box, white, sA, rA, tA); AddNode(s, box, white, sB, rB, tB);
    const double sL[3] = { 0.5, 1, 0.5 }, tL[3] = { 0, 1.98, 0 };
    AddNode(s, quad, light, sL, kRotNY, tL);
    Finish(s);
}


} // namespace hobbyrt
